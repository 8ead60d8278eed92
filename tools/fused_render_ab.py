"""render_rays as two launches per chunk (run.FUSED_RENDER, csrc/mlp_fwd16.hip FUSE = 1 / 2) vs the six-launch chain: the
1,024-ray supervision-size render and the 378x504 frame, test-mode kwargs, same weights; launches counted by the profiler.
Three variants, interleaved: `fused` (two launches at every size: FUSED_RENDER_MAX_RAYS lifted), `six_launch_chain`, and
`default` (the product's choice by chunk size: two launches up to run.FUSED_RENDER_MAX_RAYS rays, the chain above)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from mvip_nerf_amd import run, ops                             # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    tr, te, _, _, _ = run.create_nerf(bench.make_args(), device=dev)
    H, W, F, NEAR, FAR = bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR
    rows_all = ops.ray_rows_from_pose(bench.orbit_pose(0, dev), H, W, F, NEAR, FAR)
    rows_1k = rows_all[torch.linspace(0, rows_all.shape[0] - 1, 1024).long()].contiguous()
    kw = dict(lindisp=True, perturb=0., N_importance=64, network_fine=te['network_fine'], white_bkgd=True, raw_noise_std=0.)
    out = {}
    order = ('fused', 'six_launch_chain', 'default') * 2
    if '--swap' in sys.argv:
        order = order[::-1]
    default_max = run.FUSED_RENDER_MAX_RAYS
    for rnd, variant in enumerate(order):               # interleaved: a sustained run drifts by a few per cent (clocks)
        run.FUSED_RENDER = variant != 'six_launch_chain'
        run.FUSED_RENDER_MAX_RAYS = (1 << 30) if variant == 'fused' else default_max
        rec = {}
        with torch.no_grad():
            def small():
                return run.render_rays(rows_1k, te['network_fn'], te['network_query_fn'], 64, **kw)

            def frame():
                return run.render(H, W, F, chunk=1 << 15, c2w=bench.orbit_pose(1, dev), near=NEAR, far=FAR, **te)
            for name, fn, reps in (('render_rays_1024', small, 200), ('frame_378x504', frame, 5)):
                fn(); fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                rec[name + '_ms'] = (time.perf_counter() - t0) / reps * 1e3
                with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                    fn()
                    torch.cuda.synchronize()
                ev = [e for e in prof.key_averages() if e.device_time_total > 0]
                rec[name + '_launches'] = sum(e.count for e in ev)
                rec[name + '_device_ms'] = sum(e.device_time_total for e in ev) / 1e3
        ev_top = sorted(ev, key=lambda e: -e.device_time_total)[:3]
        rec['frame_top_kernels'] = [[e.key[:60], e.count, round(e.device_time_total / 1e3, 2)] for e in ev_top]
        out[variant + f'_round{rnd // 3}'] = rec
    run.FUSED_RENDER, run.FUSED_RENDER_MAX_RAYS = True, default_max
    print(json.dumps(out, indent=1))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/r5_fused_render_ab.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
