"""Time of one mvip_absmax_scale launch as a function of the grid (MVIP_ABSMAX_FPT floats per thread, MVIP_ABSMAX_MAXB most
workgroups) for the tensor sizes of the SDS step: 200 launches replayed from a hipGraph (what the step does), per launch."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops

dev = torch.device('cuda', 0)
sizes = [2 * 1280 * 64, 2 * 1280 * 256, 2 * 640 * 1024, 2 * 320 * 4096, 2 * 640 * 4096, 4 * 64 * 64 * 512, 128 * 512 * 512, 2 * 128 * 512 * 512]
out = []
for n in sizes:
    x = torch.randn(n, device=dev)
    for fpt, maxb in ((256, 256), (128, 256), (64, 256), (32, 256), (16, 256), (64, 64), (32, 64), (16, 64), (32, 128), (16, 128), (8, 128)):
        os.environ['MVIP_ABSMAX_FPT'], os.environ['MVIP_ABSMAX_MAXB'] = str(fpt), str(maxb)
        for _ in range(3):
            ops.absmax_scale(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(200):
                    sc = ops.absmax_scale(x)
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 1000 * 1e6
        out.append({'n': n, 'fpt': fpt, 'maxb': maxb, 'us': round(us, 2), 'scale': float(sc[0])})
        print(out[-1], flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/absmax_sweep.json', 'w'))
