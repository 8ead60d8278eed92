"""BASELINE.json configs at their REAL sizes through the trainer and the full-size diffusion networks
(SD-1.5-inpaint shapes, random weights), on the real poses / masks / bounds of SPIn-NeRF scene 1:
the rasters of tests/golden/scene1_small.npz resampled to the config's resolution (the GPU box has no dataset).

  configs[0]  factor 8 (283 x 504), 64 coarse samples, no fine network, no guidance   -> coarse-only render vs oracle
  configs[1]  factor 4 (567 x 1008), 64 + 128 samples, RGB SDS                         -> one iteration + properties
  configs[2]  factor 4, RGB + normal SDS, normalmap_render_factor = 2                  -> one iteration
  f1          render_path + held-out PSNR: HIP render vs oracle render of the SAME trained weights, |dPSNR| < 0.05 dB
"""
import os
import types

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from oracle.weights import seeded_state_dict

pytestmark = pytest.mark.gpu
FIXTURE = os.path.join(os.path.dirname(__file__), 'golden', 'scene1_small.npz')


def N(t):
    return t.detach().cpu().numpy()


def cfg_args(**over):
    a = types.SimpleNamespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=64, alpha_model_path=None,
        netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=3e-3,
        basedir='/tmp/mvip_cfg', expname='none', ft_path=None, no_reload=True, perturb=1., N_samples=64,
        white_bkgd=True, raw_noise_std=1., dataset_type='llff', no_ndc=True, lindisp=True, sigma_loss=False,
        N_rand=1024, chunk=1 << 15, lrate_decay=10, depth_lambda=0.1, sds_loss_weight=1e-4, no_coarse=False,
        is_normal_guidance=False, is_colla_guidance=False, normalmap_render_factor=2)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def guidance(dev, sd, **flags):
    from mvip_nerf_amd.nerf.utils import Pretrain_Model
    opt = types.SimpleNamespace(is_rgb_guidance=True, is_colla_guidance=False, is_normal_guidance=False,
                                text='a stone bench in a park', text_normal='a normal map of a stone bench in a park',
                                rgb_guidance_scale=7.5, colla_guidance_scale=7.5, normal_guidance_scale=1.5,
                                normal_start=500, lambda_guidance=1, uniform_sphere_rate=0)
    for k, v in flags.items():
        setattr(opt, k, v)
    return Pretrain_Model(opt, dev, {'SD': sd})


@pytest.fixture(scope='module')
def full_sd(cuda):
    from mvip_nerf_amd.guidance.sd_utils import StableDiffusion
    return StableDiffusion(cuda, False, False)                  # SD-1.5-inpaint shapes, random weights, fp32


@pytest.fixture(scope='module')
def scene_f4(cuda):
    from mvip_nerf_amd.scene import LLFFScene
    return LLFFScene.from_fixture(FIXTURE, size=(567, 1008), device=cuda, views=[0, 6, 12, 18, 24, 29])


def load(tr, sc, sf):
    for net, seed in ((tr.kw_train['network_fn'], sc), (tr.kw_train['network_fine'], sf)):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(seed).items()})


def _config1_gradient_vs_oracle(cuda, full_sd, sc, precision):
    """The config-level autograd check of configs[1] (DS_NeRF/run.py:948-974, :1000-1031): masked-set render driven by the image-space
    gradient the real prior produced + colour / depth supervision batches, backward through both MLPs, every parameter
    gradient against the CPU oracle's autograd on the same rays -- with the NeRF kernels in `precision`
    (0 = exact fp32 MFMA, the default; 1 = split-precision fp16 MFMA, train_precision = inference_precision = 1)."""
    from mvip_nerf_amd.trainer import SecondStageTrainer
    # -- the image-space gradient of the real prior w.r.t. the assembled frame, as the upstream of an autograd check
    tr2 = SecondStageTrainer(cfg_args(perturb=0., raw_noise_std=0.), sc, cuda, guidance=None)
    load(tr2, 71, 72)
    for net in (tr2.kw_train['network_fn'], tr2.kw_train['network_fine']):
        net.train_precision = net.inference_precision = precision
    pose, midx = sc.poses[2], sc.masked_idx_of(2)
    with torch.no_grad():
        rgb_m = tr2._render_pixels(pose, midx, retraw=True, **tr2.kw_test)['rgb_map']
    combin = sc.images[2].reshape(-1, 3).index_put((midx,), rgb_m).reshape(sc.H, sc.W, 3).permute(2, 0, 1)[None]
    mask = sc.mask_of(2).float().reshape(1, 1, sc.H, sc.W)
    d_img = full_sd.image_grad('rgb', 1000, mask, 'a stone bench in a park', combin, 7.5, weight=1.0, seed=3)
    assert d_img.shape == combin.shape and torch.isfinite(d_img).all()
    G = d_img[0].permute(1, 2, 0).reshape(-1, 3)
    assert float(G[midx].abs().max()) > 0
    sub = midx[::97]                                            # strided subset of the masked rays
    kw = {k: v for k, v in tr2.kw_train.items()}
    rec_c, rec_d = sc.next_batch('rays_rgb_clf', 64), sc.next_batch('rays_inp', 64)
    for p in tr2.grad_vars:
        p.grad = None
    r1 = tr2._render_pixels(pose, sub, retraw=True, **kw)
    r2 = tr2._render_records(rec_c[0], retraw=True, **kw)
    r3 = tr2._render_records(rec_d[0], retraw=True, **kw)
    Gs = G[sub] / (float(G[sub].abs().max()) + 1e-30)
    loss_h = ((r1['rgb_map'] * Gs).sum() + ((r2['rgb_map'] - rec_c[1].float()) ** 2).mean()
              + ((r2['rgb0'] - rec_c[1].float()) ** 2).mean() + 0.1 * ((r3['disp_map'] - rec_d[2].float()) ** 2).mean())
    loss_h.backward()
    # oracle: the same rays (the records' fp16 rays assembled the reference's way), the same loss
    pc = {k: torch.from_numpy(v).requires_grad_(True) for k, v in seeded_state_dict(71).items()}
    pf = {k: torch.from_numpy(v).requires_grad_(True) for k, v in seeded_state_dict(72).items()}
    ro, rd = O.get_rays(sc.H, sc.W, sc.focal, pose.cpu())
    rows1 = O.assemble_ray_batch(ro.reshape(-1, 3)[sub.cpu()], rd.reshape(-1, 3)[sub.cpu()], sc.near, sc.far)

    def rows_of(rays):
        o, d = rays[0].cpu(), rays[1].cpu()
        v = (d / torch.norm(d, dim=-1, keepdim=True)).float()                      # run.py:1186-1187 on fp16 rays
        o, d = o.float(), d.float()
        return torch.cat([o, d, torch.full_like(d[:, :1], sc.near), torch.full_like(d[:, :1], sc.far), v], -1)
    rr = lambda rows: O.render_rays(rows, pc, pf, 64, 64, lindisp=True, white_bkgd=True)
    o1, o2, o3 = rr(rows1), rr(rows_of(rec_c[0])), rr(rows_of(rec_d[0]))
    loss_o = ((o1['rgb_map'] * Gs.cpu()).sum() + ((o2['rgb_map'] - rec_c[1].float().cpu()) ** 2).mean()
              + ((o2['rgb0'] - rec_c[1].float().cpu()) ** 2).mean() + 0.1 * ((o3['disp_map'] - rec_d[2].float().cpu()) ** 2).mean())
    np.testing.assert_allclose(float(loss_h.detach()), float(loss_o.detach()), rtol=5e-4)
    loss_o.backward()
    worst = []
    for net, ref in ((tr2.kw_train['network_fn'], pc), (tr2.kw_train['network_fine'], pf)):
        for k, q in net.named_parameters():
            want, got = ref[k].grad.numpy().astype(np.float64), N(q.grad).astype(np.float64)
            # per tensor: relative L2 error 4 %, no entry off by more than 8 % of the tensor's largest gradient.  The
            # supervision batches here are 64 rays: a hidden unit whose pre-activation sits at rounding distance from
            # zero for a few of the 4,096 points switches its ReLU gate between the two implementations, which moves
            # that unit's row by a few per cent (seen: one row of pts_linears.0 off by 4.5 %, its neighbours by
            # 0.1 %); the small goldens (tests/test_render.py) pin the autograd to 5e-3
            rel = np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30)
            worst.append((rel, k))
            assert rel <= 4e-2, (k, rel)
            assert np.abs(got - want).max() <= 8e-2 * np.abs(want).max() + 1e-12, k
    print(f'config1 gradient check (precision {precision}): worst relative L2 per tensor', sorted(worst)[-3:], 'median', float(np.median([w[0] for w in worst])))
    return tr2, pose, pc, pf, ro, rd


def test_config1_iteration_at_567x1008(cuda, full_sd, scene_f4):
    """configs[1]: one second-stage iteration with the real prior at the real frame size, then the renderer's
    properties at that size and its autograd against the oracle on a strided ray subset, driven by the image-space
    gradient the diffusion prior actually produced."""
    from mvip_nerf_amd import run, ops
    from mvip_nerf_amd.trainer import SecondStageTrainer
    sc = scene_f4
    assert (sc.H, sc.W) == (567, 1008) and abs(sc.focal - 3069.17 / 4) < 1.0 and 1.0 < sc.near < 2.0 < sc.far
    frac = float(sc.masks.float().mean())
    assert 0.04 < frac < 0.08                                   # scene 1's masks cover ~6 % of the frame
    tr = SecondStageTrainer(cfg_args(), sc, cuda, guidance=guidance(cuda, full_sd))
    load(tr, 71, 72)
    torch.manual_seed(0)
    loss, n_rays = tr.step(1000, img_i=2)
    assert torch.isfinite(loss) and n_rays == sc.masked_idx_of(2).numel() + 2 * 1024
    assert 25000 < sc.masked_idx_of(2).numel() < 45000
    grads = [p.grad for p in tr.grad_vars]
    assert len(grads) == 48 and all(g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0 for g in grads)
    del loss

    tr2, pose, pc, pf, ro, rd = _config1_gradient_vs_oracle(cuda, full_sd, sc, precision=0)

    # -- properties at 567 x 1008: chunk invariance bit-exact, strided sample == oracle
    with torch.no_grad():
        a = run.render(sc.H, sc.W, sc.focal, chunk=1 << 15, c2w=pose, near=sc.near, far=sc.far, **tr2.kw_test)
        b = run.render(sc.H, sc.W, sc.focal, chunk=sc.H * sc.W, c2w=pose, near=sc.near, far=sc.far, **tr2.kw_test)
    for k in range(4):
        assert torch.equal(a[k], b[k]), k
    sel = torch.arange(0, sc.H * sc.W, 2999)
    rows = O.assemble_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], sc.near, sc.far)
    with torch.no_grad():
        ref = O.render_rays(rows, {k: v.detach() for k, v in pc.items()}, {k: v.detach() for k, v in pf.items()}, 64, 64,
                            lindisp=True, white_bkgd=True)
    np.testing.assert_allclose(N(a[0].reshape(-1, 3))[sel.numpy()], N(ref['rgb_map']), rtol=1e-4, atol=1e-5)


def test_split_precision_config1_gradients_vs_oracle(cuda, full_sd, scene_f4):
    """VERDICT r5 task 1b: the SAME config-level gradient check with train_precision = inference_precision = 1 (what bench.py's
    *_f16x3 legs run): per-tensor relative L2 <= 4 %, no entry off by more than 8 % of the tensor's largest gradient -- the
    bounds of the fp32 test (the mode's own error, ~3e-7 relative per layer output, is far inside the ReLU-gate noise those
    bounds exist for) -- and one whole iteration with the real prior in that mode gives finite, non-zero gradients."""
    from mvip_nerf_amd.trainer import SecondStageTrainer
    sc = scene_f4
    _config1_gradient_vs_oracle(cuda, full_sd, sc, precision=1)
    tr = SecondStageTrainer(cfg_args(), sc, cuda, guidance=guidance(cuda, full_sd))
    load(tr, 71, 72)
    for net in (tr.kw_train['network_fn'], tr.kw_train['network_fine']):
        net.train_precision = net.inference_precision = 1
    torch.manual_seed(0)
    loss, n_rays = tr.step(1000, img_i=2)
    g1 = [p.grad.clone() for p in tr.grad_vars]
    assert torch.isfinite(loss) and all(torch.isfinite(g).all() and float(g.abs().max()) > 0 for g in g1)
    assert n_rays == sc.masked_idx_of(2).numel() + 2 * 1024


def test_config2_iteration_normal_sds_factor2(cuda, full_sd, scene_f4):
    """configs[2]: RGB + normal SDS with normalmap_render_factor = 2: a 283 x 504 depth frame rendered WITH grad
    (142,632 rays x 192 points = 271 GB of activation stash if kept: the device-sized budget must fall back to
    recomputation for what does not fit), 31x31 plane-fit normals, two prior evaluations."""
    from mvip_nerf_amd import ops
    from mvip_nerf_amd.trainer import SecondStageTrainer
    sc = scene_f4
    pm = guidance(cuda, full_sd, is_normal_guidance=True)
    tr = SecondStageTrainer(cfg_args(is_normal_guidance=True), sc, cuda, guidance=pm)
    load(tr, 73, 74)
    torch.cuda.reset_peak_memory_stats(cuda)
    torch.manual_seed(1)
    loss, n_rays = tr.step(501, img_i=4)                        # i > normal_start: the normal term is active
    assert torch.isfinite(loss)
    assert n_rays == sc.masked_idx_of(4).numel() + (567 // 2) * (1008 // 2) + 2 * 1024
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0 for p in tr.grad_vars)
    total = torch.cuda.get_device_properties(cuda).total_memory
    assert torch.cuda.max_memory_allocated(cuda) < 0.9 * total
    nm, _ = tr._normal_map(sc.poses[4])
    assert nm.shape == (1, 3, 283, 504) and torch.isfinite(nm).all()
    # gated before normal_start, as cal_loss does (nerf/utils.py:298)
    calls = []
    orig = full_sd.train_step_sd_normal
    full_sd.train_step_sd_normal = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        tr.step(500, img_i=4)
    finally:
        full_sd.train_step_sd_normal = orig
    assert calls == []


def test_config0_coarse_only_render_vs_oracle(cuda):
    """configs[0] geometry on the GPU: factor 8 (283 x 504), 64 coarse samples, N_importance = 0 (no fine network)."""
    from mvip_nerf_amd import run
    from mvip_nerf_amd.scene import LLFFScene
    sc = LLFFScene.from_fixture(FIXTURE, size=(283, 504), device=cuda, views=[3], build_sets=False)
    tr, te, _, grad_vars, _ = run.create_nerf(cfg_args(N_importance=0), device=cuda)
    assert te['network_fine'] is None and len(grad_vars) == 24
    te['network_fn'].load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(75).items()})
    with torch.no_grad():
        out = run.render(sc.H, sc.W, sc.focal, chunk=1 << 15, c2w=sc.poses[0], near=sc.near, far=sc.far, **te)
    assert out[0].shape == (283, 504, 3) and 'rgb0' not in out[4] and 'z_std' not in out[4]
    ro, rd = O.get_rays(sc.H, sc.W, sc.focal, sc.poses[0].cpu())
    sel = torch.arange(0, sc.H * sc.W, 211)
    rows = O.assemble_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], sc.near, sc.far)
    pc = {k: torch.from_numpy(v) for k, v in seeded_state_dict(75).items()}
    with torch.no_grad():
        ref = O.render_rays(rows, pc, None, 64, 0, lindisp=True, white_bkgd=True)
    for k, idx in (('rgb_map', 0), ('disp_map', 1), ('acc_map', 2), ('depth_map', 3)):
        got = N(out[idx].reshape(sc.H * sc.W, -1))[sel.numpy()].reshape(ref[k].shape)
        np.testing.assert_allclose(got, ref[k].numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(N(out[4]['z_vals'].reshape(sc.H * sc.W, -1))[sel.numpy()], ref['z_vals'].numpy(), rtol=2e-6)
    # the training form: gradients reach the one network
    rows_d = rows[:64].to(cuda)
    r = run.render_rays(rows_d, tr['network_fn'], tr['network_query_fn'], 64, lindisp=True, perturb=0., N_importance=0,
                        white_bkgd=True)
    r['rgb_map'].sum().backward()
    assert all(p.grad is not None for p in grad_vars)


@pytest.mark.parametrize('precision', [0, 1], ids=['f32', 'f16x3'])
def test_heldout_psnr_hip_vs_oracle_within_0p05_dB(cuda, tmp_path, precision):
    """The north star's PSNR clause as a test: train on REAL data (SPIn-NeRF scene 1 at 1/16 resolution), render a
    held-out view through render_path (with the reference's on-disk layout), and compare PSNR-vs-ground-truth of the
    HIP render with that of the CPU-oracle render of the SAME weights on a fixed pixel subset: |dPSNR| < 0.05 dB."""
    from mvip_nerf_amd import run, ops
    from mvip_nerf_amd.run_nerf_helpers import img2mse
    d = np.load(FIXTURE)
    images = torch.from_numpy(d['images'].astype(np.float32) / 255.).to(cuda)
    poses = torch.from_numpy(d['poses'][:, :, :4]).to(cuda)
    Nv, H, W, _ = images.shape
    focal = float(d['poses'][0, 2, 4]) * (H / float(d['poses'][0, 0, 4]))
    near, far = float(d['bds'].min() * .9), float(d['bds'].max() * 1.)
    held = 14
    i_train = [i for i in range(Nv) if i != held]
    args = cfg_args(lrate=5e-4, white_bkgd=False, lindisp=False)
    torch.manual_seed(0)
    tr, te, _, grad_vars, opt = run.create_nerf(args, device=cuda)
    for net in (tr['network_fn'], tr['network_fine']):          # precision 1: training AND the held-out render in split precision
        net.train_precision = net.inference_precision = precision
    assert te['network_fn'] is tr['network_fn'] and te['network_fine'] is tr['network_fine']
    kw_tr = {k: v for k, v in tr.items() if k not in ('ndc', 'use_viewdirs')}
    g = torch.Generator(device=cuda).manual_seed(0)
    for it in range(1500):
        v = i_train[int(torch.randint(0, len(i_train), (1,), generator=g, device=cuda))]
        sel = torch.randint(0, H * W, (4096,), generator=g, device=cuda)
        rows = ops.ray_rows_from_pose(poses[v], H, W, focal, near, far, sel=sel)
        r = run.batchify_rays(rows, 1 << 15, **kw_tr)
        tgt = images[v].reshape(-1, 3)[sel]
        loss = img2mse(r['rgb_map'], tgt) + img2mse(r['rgb0'], tgt)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    rgbs, disps, _ = run.render_path(poses[held:held + 1], (H, W, focal), 1 << 15, dict(te, near=near, far=far),
                                     gt_imgs=images[held:held + 1], savedir=str(tmp_path))
    for sub in ('rgb/000000.png', 'depth/000000.npy', 'disp/000000.npy', 'weight/000000.npy', 'z/000000.npy',
                'pose/000000.txt', 'images/000000.png', 'intrinsics.txt'):
        assert (tmp_path / sub).exists(), sub
    hip = torch.from_numpy(rgbs[0]).reshape(-1, 3)
    pc = {k: p.detach().cpu() for k, p in tr['network_fn'].named_parameters()}
    pf = {k: p.detach().cpu() for k, p in tr['network_fine'].named_parameters()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ro, rd = O.get_rays(H, W, focal, poses[held].cpu())
    sel = torch.arange(0, H * W, 3)
    rows = O.assemble_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], near, far)
    with torch.no_grad():
        ora = torch.cat([O.render_rays(rows[i:i + 4096], pc, pf, 64, 64, lindisp=False, white_bkgd=False)['rgb_map']
                         for i in range(0, rows.shape[0], 4096)], 0)
    gt = images[held].cpu().reshape(-1, 3)[sel]
    psnr = lambda x: float(-10 * torch.log10(((x - gt) ** 2).mean()))
    p_h, p_o = psnr(hip[sel]), psnr(ora)
    assert p_h > 20.0                                          # 1,500 iterations of 4,096 rays: past the coarse fit (3,000: 24.7 dB,
                                                               # profiles/r3_real_scene1.json)
    assert abs(p_h - p_o) < 0.05, (p_h, p_o)
    assert float(-10 * torch.log10(((hip[sel] - ora) ** 2).mean())) > 60.0        # HIP vs oracle, same weights


FIXTURE_F8 = os.path.join(os.path.dirname(__file__), 'golden', 'scene1_f8.npz')


@pytest.mark.slow
@pytest.mark.parametrize('precision', [0, 1], ids=['f32', 'f16x3'])
def test_heldout_psnr_at_factor8_hip_vs_oracle_within_0p05_dB(cuda, precision):
    """The PSNR clause at a BASELINE size (VERDICT r3 task 9): factor 8 = 283 x 504, configs[0]'s geometry, REAL pixels of
    SPIn-NeRF scene 1 (tests/golden/scene1_f8.npz: the factor-4 rasters of the reference's own loader box-filtered 2x, 15
    training views + view 30 held out; oracle/gen_golden_llff.py).  6,000 photometric iterations of 4,096 rays (3,000 reach
    21.6 dB on this 15-view subset, 6,000 reach 23.0: profiles/r4_real_scene1_f8.json)
    (DS_NeRF/run.py:1000-1039's loss form without the prior), the held-out view rendered by the HIP path
    (DS_NeRF/run.py:1222-1362 render_path) and, from the SAME trained weights, by the CPU oracle on every 5th pixel:
    PSNR(HIP) >= 22 dB against the ground truth (DS_NeRF/run_nerf_helpers.py:17 mse2psnr), |PSNR(HIP) - PSNR(oracle)| < 0.05 dB
    on the same pixels, HIP vs oracle > 60 dB."""
    from mvip_nerf_amd import run, ops
    from mvip_nerf_amd.run_nerf_helpers import img2mse
    d = np.load(FIXTURE_F8)
    images = torch.from_numpy(d['images'].astype(np.float32) / 255.).to(cuda)
    poses = torch.from_numpy(d['poses'][:, :, :4]).to(cuda)
    Nv, H, W, _ = images.shape
    assert (H, W) == (283, 504) and int(d['factor']) == 8
    focal = float(d['poses'][0, 2, 4]) * (W / float(d['poses'][0, 1, 4]))          # 383.65: BASELINE's focal
    assert abs(focal - 383.65) < 0.01
    near, far = float(d['bds'].min() * .9), float(d['bds'].max() * 1.)
    held = int(np.nonzero(d['views'] == int(d['held_out_view']))[0][0])
    i_train = [i for i in range(Nv) if i != held]
    args = cfg_args(lrate=5e-4, white_bkgd=False, lindisp=False)
    torch.manual_seed(0)
    tr, te, _, grad_vars, opt = run.create_nerf(args, device=cuda)
    for net in (tr['network_fn'], tr['network_fine']):          # precision 1: training AND the held-out render in split precision
        net.train_precision = net.inference_precision = precision
    assert te['network_fn'] is tr['network_fn'] and te['network_fine'] is tr['network_fine']
    kw_tr = {k: v for k, v in tr.items() if k not in ('ndc', 'use_viewdirs')}
    g = torch.Generator(device=cuda).manual_seed(0)
    for it in range(6000):
        v = i_train[int(torch.randint(0, len(i_train), (1,), generator=g, device=cuda))]
        sel = torch.randint(0, H * W, (4096,), generator=g, device=cuda)
        rows = ops.ray_rows_from_pose(poses[v], H, W, focal, near, far, sel=sel)
        r = run.batchify_rays(rows, 1 << 15, **kw_tr)
        tgt = images[v].reshape(-1, 3)[sel]
        loss = img2mse(r['rgb_map'], tgt) + img2mse(r['rgb0'], tgt)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    rgbs, _, _ = run.render_path(poses[held:held + 1], (H, W, focal), 1 << 15, dict(te, near=near, far=far))
    hip = torch.from_numpy(rgbs[0]).reshape(-1, 3)
    pc = {k: p.detach().cpu() for k, p in tr['network_fn'].named_parameters()}
    pf = {k: p.detach().cpu() for k, p in tr['network_fine'].named_parameters()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ro, rd = O.get_rays(H, W, focal, poses[held].cpu())
    sel = torch.arange(0, H * W, 5)
    rows = O.assemble_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], near, far)
    with torch.no_grad():
        ora = torch.cat([O.render_rays(rows[i:i + 4096], pc, pf, 64, 64, lindisp=False, white_bkgd=False)['rgb_map']
                         for i in range(0, rows.shape[0], 4096)], 0)
    gt = images[held].cpu().reshape(-1, 3)[sel]
    psnr = lambda x: float(-10 * torch.log10(((x - gt) ** 2).mean()))
    p_h, p_o = psnr(hip[sel]), psnr(ora)
    assert p_h >= 22.0, (p_h, p_o)
    assert abs(p_h - p_o) < 0.05, (p_h, p_o)
    assert float(-10 * torch.log10(((hip[sel] - ora) ** 2).mean())) > 60.0        # HIP vs oracle, same weights
