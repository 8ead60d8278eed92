"""A/B of the persistent form of the exact-fp32 forward (csrc/mlp_fwd16.hip::mlp_forward16_persistent_kernel: one workgroup
per CU looping over its tiles, the weight stream carried from tile to tile) against one workgroup per tile: the bench's
fine-pass launch (190,512 rays x 128 samples), a 32,768-ray chunk, and the whole 378x504 frame; bit-equality of the results.
The switch is read once per process: MVIP_MLP_PERSISTENT = 0 / 1, one child process per mode, interleaved."""
import json
import os
import subprocess
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import hashlib
    import torch
    sys.path.insert(0, os.path.dirname(here))
    import bench
    from mvip_nerf_amd import ops, run
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    tr, te, *_ = run.create_nerf(bench.make_args(), device=dev)
    net = te['network_fine']
    rows = ops.ray_rows_from_pose(bench.orbit_pose(0, dev), bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR)
    z = ops.stratified_z(rows, 128, True)
    zc = ops.stratified_z(rows, 64, True)

    def t(fn, n):
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    out = {}
    with torch.no_grad():
        raw = net.query_rays(rows, z)
        rawc = net.query_rays(rows[:33001], zc[:33001])                  # odd count, 64 samples per ray
        out['digest_fine'] = hashlib.sha256(raw.cpu().numpy().tobytes()).hexdigest()[:16]
        out['digest_coarse_odd'] = hashlib.sha256(rawc.cpu().numpy().tobytes()).hexdigest()[:16]
        out['fine_launch_ms'] = t(lambda: net.query_rays(rows, z), 4)
        out['chunk_32768_ms'] = t(lambda: net.query_rays(rows[:32768], z[:32768]), 10)
        out['frame_ms'] = t(lambda: run.render(bench.H, bench.W, bench.FOCAL, chunk=1 << 15, c2w=bench.orbit_pose(1, dev), near=bench.NEAR,
                                               far=bench.FAR, **te), 4)
    print(json.dumps(out))
else:
    res = {}
    for rnd in range(2):
        for mode in ('0', '1'):
            r = subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, MVIP_MLP_PERSISTENT=mode), capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            res[f'persistent_{mode}_round{rnd}'] = json.loads(line[-1]) if line else {'error': r.stderr[-400:]}
            print(f'persistent={mode} round {rnd}:', res[f'persistent_{mode}_round{rnd}'], flush=True)
    ok = [v for v in res.values() if 'digest_fine' in v]
    res['bit_identical'] = len({(v['digest_fine'], v['digest_coarse_odd']) for v in ok}) == 1 and len(ok) == 4
    print('bit identical:', res['bit_identical'])
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/r4_persistent_ab.json', 'w'), indent=1)
