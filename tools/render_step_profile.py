"""Per-kernel device time of ONE steady-state frame of the headline metric (378 x 504, 64 + 128 samples, chunk 32768, test
mode), from the torch profiler: what the frame spends outside the two fused MLP launches of each chunk."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                            # noqa: E402
from mvip_nerf_amd import run                                           # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    tr, te, start, grad_vars, opt = run.create_nerf(bench.make_args(), device=dev)

    def step(k):
        with torch.no_grad():
            return run.render(bench.H, bench.W, bench.FOCAL, chunk=1 << 15, c2w=bench.orbit_pose(k, dev), near=bench.NEAR,
                              far=bench.FAR, **te)[0]
    for k in range(2):
        step(k)
    torch.cuda.synchronize()
    ts = []
    for k in range(5):
        t0 = time.perf_counter()
        step(2 + k)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        step(10)
        torch.cuda.synchronize()
    ev = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
    total = sum(e.device_time_total for e in ev) / 1e3
    n = sum(e.count for e in ev if e.device_time_total > 0)
    mlp = sum(e.device_time_total for e in ev if 'mlp_forward' in e.key) / 1e3
    print(f'== frame: median wall {sorted(ts)[len(ts) // 2]:.2f} ms {[round(t, 1) for t in ts]}, device-busy {total:.2f} ms, '
          f'{n} kernels / copies; fused MLP launches {mlp:.2f} ms, everything else {total - mlp:.2f} ms')
    rows = []
    for e in ev[:30]:
        if e.device_time_total <= 0:
            continue
        print(f'  {e.device_time_total / 1e3:8.3f} ms  x{e.count:4d}  {e.key[:120]}')
        rows.append([round(e.device_time_total / 1e3, 3), e.count, e.key[:120]])
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump({'median_wall_ms': sorted(ts)[len(ts) // 2], 'wall_ms': ts, 'device_busy_ms': total, 'kernels': n,
               'mlp_forward_ms': mlp, 'top': rows}, open('gpurun_out/render_step_profile.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
