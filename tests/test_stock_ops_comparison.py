"""SURVEY.md 8(d) "GPU comparison point": the CPU oracle's restatement of render_rays executed on the SAME MI355X
through stock PyTorch-ROCm ops (unfused: ~150 library kernels per chunk, encodings and every activation
materialised in HBM) next to the hand-written HIP path, on the same 32,768 rays of bench frame 0 with the same
weights.  Checks parity of the two results and that the HIP path is the faster one; the measured numbers are
written to gpurun_out/stock_ops_comparison.json (copied to profiles/ when refreshed)."""
import json
import os
import time

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from oracle.weights import seeded_state_dict
from test_render import build

pytestmark = pytest.mark.gpu

H, W, FOCAL, NEAR, FAR = 378, 504, 383.65, 1.2, 7.74
N_RAYS = 32768


def _timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps


def test_hip_path_vs_stock_rocm_ops(cuda):
    from mvip_nerf_amd import run
    ro, rd = O.get_rays(H, W, FOCAL, O.bench_poses(1)[0])
    rows = O.assemble_ray_batch(ro, rd, NEAR, FAR)
    rows = rows[torch.linspace(0, rows.shape[0] - 1, N_RAYS).long()].to(cuda)
    pc = {k: torch.from_numpy(v).to(cuda) for k, v in seeded_state_dict(0).items()}
    pf = {k: torch.from_numpy(v).to(cuda) for k, v in seeded_state_dict(1).items()}
    tr, te, _, _ = build(0, 1, cuda)

    def stock():
        with torch.no_grad(), torch.device(cuda):          # the oracle's factory calls follow the default device
            return O.render_rays(rows, pc, pf, 64, 64, lindisp=True, white_bkgd=True)

    def hip():
        with torch.no_grad():
            return run.render_rays(rows, te['network_fn'], te['network_query_fn'], 64, lindisp=True, perturb=0.,
                                   N_importance=64, network_fine=te['network_fine'], white_bkgd=True,
                                   raw_noise_std=0.)

    r_stock, t_stock = _timed(stock)
    r_hip, t_hip = _timed(hip)
    # same tolerance as the golden render test (fp32 sums in a different order; a few fine depths displaced by the
    # ill-conditioned inverse CDF)
    for k in ('rgb_map', 'acc_map', 'depth_map', 'rgb0'):
        a, b = r_hip[k].cpu().numpy(), r_stock[k].cpu().numpy()
        bad = np.abs(a - b) > 1e-4 + 1e-4 * np.abs(b)
        assert bad.mean() < 0.01, (k, bad.mean(), np.abs(a - b).max())
    res = {'rays': N_RAYS, 'points_per_ray': 192, 'stock_rocm_ops_rays_per_sec': N_RAYS / t_stock,
           'hip_path_rays_per_sec': N_RAYS / t_hip, 'speedup': t_stock / t_hip,
           'what': 'oracle/nerf_oracle.render_rays on cuda:0 (stock PyTorch-ROCm fp32 ops, netchunk 65536) vs '
                   'mvip_nerf_amd.run.render_rays (HIP kernels, exact fp32), test mode, 64+128 samples'}
    print(json.dumps(res))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(res, open(os.path.join(out, 'stock_ops_comparison.json'), 'w'), indent=1)
    except OSError:
        pass
    assert t_hip < t_stock, res


def test_hip_training_vs_stock_rocm_ops(cuda):
    """Same comparison for forward + backward to the parameter gradients of both MLPs (8192 rays, test-mode sampling
    so that both sides see the same depths; loss = mean(rgb_map^2) + mean(rgb0^2))."""
    from mvip_nerf_amd import run
    n = 8192
    ro, rd = O.get_rays(H, W, FOCAL, O.bench_poses(1)[0])
    rows = O.assemble_ray_batch(ro, rd, NEAR, FAR)
    rows = rows[torch.linspace(0, rows.shape[0] - 1, n).long()].to(cuda)
    pc = {k: torch.from_numpy(v).to(cuda).requires_grad_(True) for k, v in seeded_state_dict(0).items()}
    pf = {k: torch.from_numpy(v).to(cuda).requires_grad_(True) for k, v in seeded_state_dict(1).items()}
    tr, te, grad_vars, _ = build(0, 1, cuda)

    def stock():
        for p in list(pc.values()) + list(pf.values()):
            p.grad = None
        with torch.device(cuda):
            r = O.render_rays(rows, pc, pf, 64, 64, lindisp=True, white_bkgd=True)
        ((r['rgb_map'] ** 2).mean() + (r['rgb0'] ** 2).mean()).backward()
        return [pc[k].grad for k in pc] + [pf[k].grad for k in pf]

    def hip():
        for p in grad_vars:
            p.grad = None
        r = run.render_rays(rows, tr['network_fn'], tr['network_query_fn'], 64, lindisp=True, perturb=0.,
                            N_importance=64, network_fine=tr['network_fine'], white_bkgd=True, raw_noise_std=0.)
        ((r['rgb_map'] ** 2).mean() + (r['rgb0'] ** 2).mean()).backward()
        return [p.grad for p in grad_vars]

    g_stock, t_stock = _timed(stock)
    g_hip, t_hip = _timed(hip)
    a = torch.cat([g.flatten() for g in g_hip]).double()
    b = torch.cat([g.flatten() for g in g_stock]).double()
    rel = float((a - b).norm() / b.norm())
    assert rel < 2e-3, rel            # a few displaced fine depths (ill-conditioned inverse CDF) move their samples
    res = {'rays': n, 'stock_rocm_ops_rays_per_sec': n / t_stock, 'hip_path_rays_per_sec': n / t_hip,
           'speedup': t_stock / t_hip, 'grad_rel_l2_diff': rel,
           'what': 'forward + backward to all 2 x 595,844 parameter gradients, exact fp32 kernels'}
    print(json.dumps(res))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(res, open(os.path.join(out, 'stock_ops_comparison_training.json'), 'w'), indent=1)
    except OSError:
        pass
    assert t_hip < t_stock, res
