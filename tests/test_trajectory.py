"""The north star's PSNR clause pinned to the REFERENCE'S OWN OPTIMISER TRAJECTORY (VERDICT r5 task 3).

tests/golden/trajectory_100.npz (oracle/gen_golden_trajectory.py, run in the build container) holds 100 iterations of the
reference's render(...) + img2mse(rgb) + img2mse(rgb0) + Adam + lr decay (DS_NeRF/run.py:1143, :1000-1039) on real pixels of
SPIn-NeRF scene 1 (1/16 resolution), 256 rays per iteration, with every random draw an input:

  det     perturb = 0, raw_noise_std = 0 (no draw inside render_rays);
  pytest  perturb = 1, raw_noise_std = 1 through the reference's pytest=True hooks (np.random.seed(0) + np.random.rand),

plus the held-out view rendered by the reference from ITS trained weights.  Here the same 100 iterations run through the drop-in
API (run.create_nerf / run.render / img2mse, HIP kernels, torch.optim.Adam), the same selections, the same draws.

Tolerances.  A ReLU MLP under Adam amplifies rounding differences (Adam's update g / sqrt(v) is scale-free: a parameter whose
gradient is at rounding level moves by ~lr whatever the size of the difference).  The fixture therefore also holds the reference
AGAINST ITSELF: each trajectory on ONE BLAS thread instead of eight (another summation order, nothing else).  Asserted:
  * the first 10 losses within 2e-4 relative (before amplification: same forward, same gradients, same Adam; measured 2e-7 ..
    8e-7), the first 50 within 1e-3 (measured <= 2e-4);
  * every loss within max(3e-3, 10 x the reference's own 1-thread-vs-8-thread deviation up to that iteration) relative, and the
    mean deviation over iterations 50-99 within max(1e-3, 6 x the reference's own).  Measured (three runs each, one box): the
    reference drifts from itself by at most 1.1e-3 / 2.2e-3 (mean over 50-99: 2.0e-4 / 4.9e-4) on the det / pytest trajectory;
    the HIP path from the 8-thread reference by at most 2.2-2.9e-3 / 7.1-12.5e-3 (mean 2.7e-4 / 1.5-1.7e-3), and from ITSELF,
    run to run (the weight gradients are flushed with fp32 atomics: the order of the additions is not fixed), by 0.6-1.3e-3 /
    6.7-7.8e-3 (mean 1e-4 / 1e-3) -- the distance to the reference is the distance between two runs of the same code;
  * held-out PSNR of the HIP-trained weights within 0.05 dB of the reference's (the north-star clause), or within
    2 x the reference's own 1-thread-vs-8-thread PSNR difference when that is larger (it is printed);
  * the HIP render of the held-out view vs the CPU ORACLE's render of the SAME HIP-trained weights: |dPSNR| < 0.05 dB, > 60 dB apart.
"""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from oracle.weights import seeded_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _args(g, **over):
    import types
    a = types.SimpleNamespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=64, alpha_model_path=None,
        netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=float(g['lrate']),
        basedir='/tmp/mvip_traj', expname='none', ft_path=None, no_reload=True, perturb=0., N_samples=64,
        white_bkgd=False, raw_noise_std=0., dataset_type='llff', no_ndc=True, lindisp=False, sigma_loss=False,
        N_rand=int(g['n_rays']), chunk=1 << 15, lrate_decay=int(g['lrate_decay']), no_coarse=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def _selections(g, n_views, hw):
    rs = np.random.RandomState(int(g['seed_sel']))
    held = int(g['held'])
    i_train = [i for i in range(n_views) if i != held]
    out = []
    for _ in range(int(g['n_iters'])):
        v = i_train[int(rs.randint(0, len(i_train)))]
        out.append((v, rs.randint(0, hw, size=int(g['n_rays'])).astype(np.int64)))
    return out


def _run(cuda, g, mode, precision=0):
    from mvip_nerf_amd import run
    from mvip_nerf_amd.run_nerf_helpers import get_rays, img2mse
    d = np.load(os.path.join(GOLD, 'scene1_small.npz'))
    images = torch.from_numpy(d['images'].astype(np.float32) / 255.).to(cuda)
    poses = torch.from_numpy(d['poses'][:, :, :4].astype(np.float32)).to(cuda)
    Nv, H, W, _ = images.shape
    focal, near, far = float(g['focal']), float(g['near']), float(g['far'])
    a = _args(g, perturb=1., raw_noise_std=1.) if mode == 'pytest' else _args(g)
    kw_tr, kw_te, start, grad_vars, optimizer = run.create_nerf(a, device=cuda)
    for key, seed in (('network_fn', int(g['seed_coarse'])), ('network_fine', int(g['seed_fine']))):
        kw_tr[key].load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(seed).items()})
        kw_tr[key].train_precision = kw_tr[key].inference_precision = precision
    for kw in (kw_tr, kw_te):
        kw.update(near=near, far=far)
    if mode == 'pytest':
        kw_tr['pytest'] = True
    sel = _selections(g, Nv, H * W)
    assert [s[0] for s in sel] == list(g['sel_views']) and [int(s[1].sum()) for s in sel] == list(g['sel_checksum'])
    losses, lrs = [], []
    global_step = start
    for it in range(int(g['n_iters'])):
        v, pix = sel[it]
        rays_o, rays_d = get_rays(H, W, focal, poses[v])
        pix_t = torch.from_numpy(pix).to(cuda)
        batch_rays = torch.stack([rays_o.reshape(-1, 3)[pix_t], rays_d.reshape(-1, 3)[pix_t]], 0)
        target = images[v].reshape(-1, 3)[pix_t]
        rgb, disp, acc, depth, extras = run.render(H, W, focal, chunk=a.chunk, rays=batch_rays, verbose=False, retraw=True, **kw_tr)
        optimizer.zero_grad()
        loss = img2mse(rgb, target) + img2mse(extras['rgb0'], target)
        loss.backward()
        optimizer.step()
        new_lrate = a.lrate * (0.1 ** (global_step / (a.lrate_decay * 1000)))
        for pg in optimizer.param_groups:
            pg['lr'] = new_lrate
        global_step += 1
        losses.append(float(loss.detach()))
        lrs.append(new_lrate)
    held = int(g['held'])
    with torch.no_grad():
        rgb, _, _, _, _ = run.render(H, W, focal, chunk=a.chunk, c2w=poses[held][:3, :4], **kw_te)
    rgb = rgb.reshape(-1, 3)
    gt = images[held].reshape(-1, 3)
    psnr = float(-10 * torch.log10(((rgb - gt) ** 2).mean()))
    nets = (kw_tr['network_fn'], kw_tr['network_fine'])
    return np.array(losses), np.array(lrs), psnr, rgb.cpu(), gt.cpu(), nets, (H, W, focal, near, far, poses[held].cpu())


@pytest.mark.parametrize('mode,precision', [('det', 0), ('pytest', 0), ('det', 1)], ids=['det-f32', 'pytest-f32', 'det-f16x3'])
def test_hundred_iterations_follow_the_reference_trajectory(cuda, mode, precision):
    g = np.load(os.path.join(GOLD, 'trajectory_100.npz'))
    losses, lrs, psnr, rgb, gt, nets, cam = _run(cuda, g, mode, precision)
    want = g[f'{mode}/losses']
    np.testing.assert_allclose(lrs, g[f'{mode}/lrs'], rtol=1e-12)
    rel = np.abs(losses - want) / np.abs(want)
    if os.environ.get('MVIP_TRAJECTORY_DUMP'):               # calibration runs: the whole curve, one file per run
        np.save(os.path.join(os.environ['MVIP_TRAJECTORY_DUMP'], f'traj_{mode}_{precision}_{os.getpid()}.npy'), losses)
    # the reference against itself (1 BLAS thread vs 8) on THIS trajectory: the amplification any implementation
    # sees; cumulative maximum so the bound never tightens after a divergence has happened
    self_rel = np.abs(g[f'{mode}_1thread/losses'] - g[f'{mode}/losses']) / np.abs(g[f'{mode}/losses'])
    bound = np.maximum(3e-3, 10.0 * np.maximum.accumulate(self_rel))
    print(f'[{mode}/precision {precision}] loss deviation: first 10 max {rel[:10].max():.2e}, first 50 max {rel[:50].max():.2e}, overall max '
          f'{rel.max():.2e}, mean over iterations 50-99 {rel[50:].mean():.2e} (reference vs itself: max {self_rel.max():.2e}, mean 50-99 '
          f'{self_rel[50:].mean():.2e}); every 10th: {np.round(rel[::10], 5).tolist()}')
    assert rel[:10].max() < 2e-4, rel[:10]
    assert rel[:50].max() < 1e-3, rel[:50].max()
    assert (rel <= bound).all(), (np.nonzero(rel > bound)[0], rel.max())
    assert rel[50:].mean() <= max(1e-3, 6.0 * self_rel[50:].mean()), (rel[50:].mean(), self_rel[50:].mean())
    # held-out PSNR: HIP-trained weights vs the reference's own trained weights
    p_ref = float(g[f'{mode}/heldout_psnr'])
    p_self = abs(float(g[f'{mode}_1thread/heldout_psnr']) - float(g[f'{mode}/heldout_psnr']))
    print(f'[{mode}/precision {precision}] held-out PSNR {psnr:.4f} dB, reference {p_ref:.4f} dB (reference vs itself: {p_self:.4f} dB)')
    assert abs(psnr - p_ref) < max(0.05, 2.0 * p_self), (psnr, p_ref, p_self)
    # pixels: the two renders of independently trained weights agree far better than either agrees with the photograph
    ref_rgb = torch.from_numpy(g[f'{mode}/heldout_rgb_every3'].astype(np.float32))
    assert float(-10 * torch.log10(((rgb[::3] - ref_rgb) ** 2).mean())) > psnr + 15.0
    # the SAME (HIP-trained) weights through the CPU oracle: the clause as a same-weights comparison, on every 7th pixel
    H, W, focal, near, far, pose = cam
    pc = {k: p.detach().cpu() for k, p in nets[0].named_parameters()}
    pf = {k: p.detach().cpu() for k, p in nets[1].named_parameters()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ro, rd = O.get_rays(H, W, focal, pose)
    sel = torch.arange(0, H * W, 7)
    rows = O.assemble_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], near, far)
    with torch.no_grad():
        ora = O.render_rays(rows, pc, pf, 64, 64, lindisp=False, white_bkgd=False)['rgb_map']
    ps = lambda x: float(-10 * torch.log10(((x - gt[sel]) ** 2).mean()))
    assert abs(ps(rgb[sel]) - ps(ora)) < 0.05
    assert float(-10 * torch.log10(((rgb[sel] - ora) ** 2).mean())) > 60.0
