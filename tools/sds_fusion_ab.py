"""A/B of the SDS step with round 6's two statistics fusions, each on / off: LayerNorm statistics left by the producing GEMM's
epilogue (ops.LN_STATS_FROM_GEMM, MVIP_LN_STATS_FROM_GEMM) and GroupNorm moments left by the unsplit convolution's epilogue
(ops.TILE_MOMENTS, MVIP_TILE_MOMENTS); both default to on.  Median of 15 steps, hipGraph replay and eager, kernel launches of
one eager step, each setting in its own child process, two rounds.
    python tools/sds_fusion_ab.py -> gpurun_out/r6_sds_fusion_ab.json"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from mvip_nerf_amd.guidance.sd_utils import StableDiffusion
    from mvip_nerf_amd import ops
    ops.LN_STATS_FROM_GEMM = os.environ.get('MVIP_LN_STATS_FROM_GEMM', '1') != '0'
    dev = torch.device('cuda', 0)
    out = {}
    g = torch.Generator(device=dev).manual_seed(2)
    pred = torch.rand(1, 3, 378, 504, device=dev, generator=g).requires_grad_(True)
    mask = torch.zeros(1, 1, 378, 504, device=dev)
    mask[:, :, 137:241, 196:307] = 1
    for graphs in (True, False):
        sd = StableDiffusion(dev, False, False, use_graphs=graphs)
        sd.seed_generator(11)

        def step(i):
            pred.grad = None
            (1e-4 * sd.train_step_sd(i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
        for k in range(3):
            step(k)
        torch.cuda.synchronize()
        ts = []
        for k in range(15):
            t0 = time.perf_counter()
            step(100 + k)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        out['graph_replay_ms' if graphs else 'eager_ms'] = round(sorted(ts)[7], 3)
        sd.seed_generator(11)
        step(5000)
        out['grad_checksum_' + ('graph' if graphs else 'eager')] = float(pred.grad.double().abs().sum())
        if not graphs:
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                step(200)
                torch.cuda.synchronize()
            out['kernel_launches_per_step'] = int(sum(e.count for e in prof.key_averages() if e.device_type is not None and 'Memcpy' not in e.key and 'hipLaunch' not in e.key and e.self_device_time_total > 0))
        del sd
    print('RESULT ' + json.dumps(out))


def main():
    res = {'what': __doc__.split('\n')[0]}
    for rnd in range(2):
        for ln, tile in (('1', '1'), ('0', '1'), ('1', '0'), ('0', '0')):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], capture_output=True, text=True,
                               env=dict(os.environ, MVIP_LN_STATS_FROM_GEMM=ln, MVIP_TILE_MOMENTS=tile), cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
            key = f'ln_stats_from_gemm={ln},tile_moments={tile},round{rnd}'
            res[key] = json.loads(line[-1][7:]) if line else {'error': r.stderr[-1500:]}
            print(key, res[key], flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'r6_sds_fusion_ab.json'), 'w'), indent=1)


if __name__ == '__main__':
    child() if '--child' in sys.argv else main()
