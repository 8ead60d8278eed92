"""Pins the CPU oracle (oracle/nerf_oracle.py) to outputs of the real reference captured in
tests/golden/ by oracle/gen_golden.py.  CPU only.

Tolerances: the oracle and the reference are both torch-CPU fp32 and mostly issue the same ops in
the same order, so most comparisons are exact or within a few ulp; integer outputs are exact."""
import numpy as np
import torch
import pytest

from oracle import nerf_oracle as O
from oracle.weights import seeded_state_dict


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def params(seed):
    return {k: T(v) for k, v in seeded_state_dict(int(seed)).items()}


@pytest.mark.parametrize('tag', ['identity', 'rot'])
def test_get_rays(golden, tag):
    g = golden(f'rays_{tag}')
    ro, rd = O.get_rays(int(g['H']), int(g['W']), float(g['focal']), T(g['c2w']))
    np.testing.assert_array_equal(ro.numpy(), g['rays_o'])
    np.testing.assert_allclose(rd.numpy(), g['rays_d'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(rd.numpy(), g['rays_d_np'], rtol=0, atol=1e-6)


@pytest.mark.parametrize('name,L', [('posenc_pts', 10), ('posenc_dirs', 4)])
def test_posenc(golden, name, L):
    g = golden(name)
    np.testing.assert_array_equal(O.posenc(T(g['x']), L).numpy(), g['y'])


def test_mlp_forward_backward(golden):
    g = golden('mlp_fwd_bwd')
    p = {k: v.requires_grad_(True) for k, v in params(g['seed']).items()}
    emb = torch.cat([O.posenc(T(g['pts']), 10), O.posenc(T(g['dirs']), 4)], -1)
    np.testing.assert_array_equal(emb.numpy(), g['emb'])
    out = O.mlp_forward(p, emb)
    np.testing.assert_allclose(out.detach().numpy(), g['out'], rtol=1e-5, atol=1e-6)
    (out * T(g['gout'])).sum().backward()
    for k, v in p.items():
        gr = v.grad.double().numpy().ravel()
        stat = g[f'gstat/{k}']
        np.testing.assert_allclose(np.sqrt((gr * gr).sum()), stat[2], rtol=1e-5)
        np.testing.assert_allclose(gr[g[f'gidx/{k}']], g[f'gval/{k}'], rtol=1e-4, atol=1e-5 * stat[2])


@pytest.mark.parametrize('tag', ['train', 'test', 'white', 'detach'])
def test_raw2outputs(golden, tag):
    g = golden(f'composite_{tag}')
    raw = T(g['raw']).requires_grad_(True)
    rgb, disp, acc, w, depth, alpha = O.raw2outputs(raw, T(g['z']), T(g['rays_d']), T(g['noise']),
                                                    bool(g['white']), bool(g['detach']))
    for name, val in (('rgb', rgb), ('acc', acc), ('weights', w), ('depth', depth), ('alpha', alpha)):
        np.testing.assert_allclose(val.detach().numpy(), g[name], rtol=1e-6, atol=1e-7, err_msg=name)
    np.testing.assert_allclose(disp.detach().numpy(), g['disp'], rtol=1e-6, equal_nan=True)
    assert np.isnan(g['disp'][0])          # the empty ray keeps the reference's 0/0
    ok = torch.isfinite(disp)
    loss = ((rgb * T(g['g_rgb'])).sum() + (acc * T(g['g_acc'])).sum() + (depth * T(g['g_depth'])).sum()
            + (w * T(g['g_w'])).sum() + (torch.where(ok, disp, torch.zeros_like(disp)) * T(g['g_disp'])).sum())
    loss.backward()
    np.testing.assert_allclose(raw.grad.numpy(), g['d_raw'], rtol=1e-5, atol=1e-6, equal_nan=True)


def test_sample_pdf(golden):
    g = golden('sample_pdf')
    for mode in ('det', 'pytest'):
        s, inds = O.sample_pdf(T(g['bins']), T(g['weights']), T(g[f'u_{mode}']))
        np.testing.assert_array_equal(inds.numpy(), g[f'inds_{mode}'])
        np.testing.assert_array_equal(s.numpy(), g[f'samples_{mode}'])


def _pytest_randoms(B, Nc, Nf):
    """What the reference's pytest=True hooks draw (each re-seeds numpy with 0)."""
    np.random.seed(0)
    t_rand = np.random.rand(B, Nc).astype(np.float32)
    np.random.seed(0)
    noise0 = np.random.rand(B, Nc).astype(np.float32)
    np.random.seed(0)
    u = np.random.rand(B, Nf).astype(np.float32)
    np.random.seed(0)
    noise1 = np.random.rand(B, Nc + Nf).astype(np.float32)
    return T(t_rand), T(noise0), T(u), T(noise1)


def test_render_rays_test_mode(golden):
    g = golden('render_rays_test')
    with torch.no_grad():
        r = O.render_rays(T(g['rays']), params(g['seed_coarse']), params(g['seed_fine']), 64, 64,
                          lindisp=True, white_bkgd=True, retraw=True, need_alpha=True)
    np.testing.assert_array_equal(r['z_vals'].numpy(), g['z_vals'])
    for k in ('rgb_map', 'disp_map', 'acc_map', 'depth_map', 'weights', 'raw', 'alpha', 'alpha0', 'rgb0',
              'disp0', 'acc0', 'z_std'):
        np.testing.assert_allclose(r[k].numpy(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)


def test_render_rays_train_mode(golden):
    g = golden('render_rays_pytest_train')
    pc = {k: v.requires_grad_(True) for k, v in params(g['seed_coarse']).items()}
    pf = {k: v.requires_grad_(True) for k, v in params(g['seed_fine']).items()}
    t_rand, n0, u, n1 = _pytest_randoms(64, 64, 64)
    r = O.render_rays(T(g['rays']), pc, pf, 64, 64, lindisp=True, white_bkgd=True, t_rand=t_rand,
                      noise0=n0 * 1.0, u=u, noise1=n1 * 1.0, retraw=True, need_alpha=True)
    np.testing.assert_allclose(r['z_vals'].detach().numpy(), g['z_vals'], rtol=0, atol=1e-6)
    for k in ('rgb_map', 'disp_map', 'acc_map', 'depth_map', 'weights', 'rgb0', 'disp0', 'acc0', 'z_std'):
        np.testing.assert_allclose(r[k].detach().numpy(), g[k], rtol=5e-5, atol=5e-6, err_msg=k)
    loss = ((r['rgb_map'] * T(g['g_rgb'])).sum() + (r['rgb0'] * T(g['g_rgb0'])).sum()
            + (r['disp_map'] * T(g['g_disp'])).sum() + (r['depth_map'] * T(g['g_depth'])).sum())
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-5)
    loss.backward()
    for net, p in (('coarse', pc), ('fine', pf)):
        for k, v in p.items():
            gr = v.grad.double().numpy().ravel()
            stat = g[f'gstat/{net}.{k}']
            np.testing.assert_allclose(np.sqrt((gr * gr).sum()), stat[2], rtol=2e-4, err_msg=k)
            np.testing.assert_allclose(gr[g[f'gidx/{net}.{k}']], g[f'gval/{net}.{k}'], rtol=1e-3,
                                       atol=2e-5 * stat[2], err_msg=k)


def test_render_fullframe(golden):
    g = golden('render_fullframe_15x20')
    H, W, f = int(g['H']), int(g['W']), float(g['focal'])
    ro, rd = O.get_rays(H, W, f, T(g['c2w']))
    rays = O.assemble_ray_batch(ro, rd, float(g['near']), float(g['far']))
    with torch.no_grad():
        r = O.render_rays(rays, params(g['seed_coarse']), params(g['seed_fine']), 64, 64, lindisp=True,
                          white_bkgd=True, retraw=True)
    np.testing.assert_allclose(r['rgb_map'].reshape(H, W, 3).numpy(), g['rgb'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(r['depth_map'].reshape(H, W).numpy(), g['depth'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(r['disp_map'].reshape(H, W).numpy(), g['disp'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(r['z_vals'].reshape(H, W, 128).numpy(), g['extras/z_vals'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r['raw'].reshape(H, W, 128, 4).numpy(), g['extras/raw'], rtol=2e-5, atol=2e-5)


def test_normal_fit(golden):
    g = golden('normal_fit_54x72')
    depth = T(g['depth']).requires_grad_(True)
    pts = O.depth2xyz(depth, T(g['K']))
    np.testing.assert_allclose(pts.detach().numpy(), g['points'], rtol=1e-6, atol=1e-7)
    P = pts.permute(2, 0, 1)[None]
    n_unfold = O.normal_fit_unfold(P.detach())
    np.testing.assert_allclose(n_unfold.numpy(), g['normals'], rtol=1e-3, atol=1e-4)
    # the box-sum formulation (what the HIP kernel implements) agrees with the reference to the
    # conditioning of the reference's own fp32 inverse (~1e-4 relative)
    n_box = O.normal_fit_boxsum(P)
    scale = np.abs(g['normals']).max()
    assert np.abs(n_box.detach().numpy() - g['normals']).max() < 2e-3 * scale
    (n_box * T(g['g'])).sum().backward()
    gscale = np.abs(g['d_depth']).max()
    assert np.abs(depth.grad.numpy() - g['d_depth']).max() < 2e-2 * gscale


def test_misc(golden):
    g = golden('misc')
    mse = O.img2mse(T(g['a']), T(g['b']))
    np.testing.assert_allclose(mse.numpy(), g['mse'], rtol=1e-6)
    np.testing.assert_allclose(O.mse2psnr(mse).numpy(), g['psnr'].reshape(()), rtol=1e-6)


# ---- the SDS wrapper oracle (oracle/sds_oracle.py) vs the reference's own wrapper on the stand-in networks --------
def _standin_nets():
    import types
    from oracle.sds_standin import TinyVAE, TinyUNet, TinyScheduler, prompt_embedding
    return types.SimpleNamespace(vae=TinyVAE(), unet=TinyUNet(), encode_prompt=prompt_embedding,
                                 alphas_cumprod=TinyScheduler().alphas_cumprod)


@pytest.mark.parametrize('name', ['sds_rgb_i0', 'sds_rgb_i100', 'sds_rgb_i5000', 'sds_rgb_i20000', 'sds_normal'])
def test_sds_wrapper_oracle(golden, name):
    from oracle import sds_oracle as S
    g = golden(name)
    nets = _standin_nets()
    torch.manual_seed(int(g['seed']))
    draws = iter([torch.randn(1, 4, 64, 64) for _ in range(4)])            # the reference's CPU draws, in its order
    nets.vae.randn = lambda s: next(draws)
    pred = T(g['pred']).requires_grad_(True)
    kw = dict(guidance_scale=float(g['guidance_scale']), randn=lambda s: next(draws))
    if name == 'sds_normal':
        loss = S.train_step_sd_normal(nets, int(g['i']), T(g['mask']), 'a normal map of a stone bench', pred,
                                      normal_start=int(g['normal_start']), **kw)
    else:
        loss = S.train_step_sd(nets, int(g['i']), T(g['mask']), 'a stone bench in a park', pred, **kw)
    assert float(loss) == 1.0
    (float(g['upstream']) * loss).sum().backward()
    np.testing.assert_allclose(pred.grad.numpy(), g['d_pred'], rtol=1e-4, atol=1e-6 * np.abs(g['d_pred']).max())


@pytest.mark.parametrize('mode', ['det', 'pytest'])
def test_oracle_follows_the_reference_trajectory(golden, mode):
    """tests/golden/trajectory_100.npz (oracle/gen_golden_trajectory.py: 100 iterations of the reference's own
    render + img2mse(rgb) + img2mse(rgb0) + Adam + lr decay, DS_NeRF/run.py:1000-1039, on real pixels): the oracle's
    render_rays + torch.optim.Adam reproduce the first 8 losses of the reference's trajectory (2e-5 relative: same
    torch-CPU arithmetic, a different grouping of a few sums), with the stochastic inputs of the 'pytest' trajectory drawn
    as the reference's pytest=True hooks draw them (np.random.seed(0) + np.random.rand)."""
    import os
    g = golden('trajectory_100')
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'scene1_small.npz'))
    images = T(d['images'].astype(np.float32) / 255.)
    poses = T(d['poses'][:, :, :4].astype(np.float32))
    Nv, H, W, _ = images.shape
    focal, near, far = float(g['focal']), float(g['near']), float(g['far'])
    pc = {k: v.requires_grad_(True) for k, v in params(g['seed_coarse']).items()}
    pf = {k: v.requires_grad_(True) for k, v in params(g['seed_fine']).items()}
    opt = torch.optim.Adam(list(pc.values()) + list(pf.values()), lr=float(g['lrate']), betas=(0.9, 0.999))
    rs = np.random.RandomState(int(g['seed_sel']))
    i_train = [i for i in range(Nv) if i != int(g['held'])]
    n_rays = int(g['n_rays'])
    torch.set_num_threads(8)
    for it in range(8):
        v = i_train[int(rs.randint(0, len(i_train)))]
        pix = T(rs.randint(0, H * W, size=n_rays).astype(np.int64))
        assert v == int(g['sel_views'][it]) and int(pix.sum()) == int(g['sel_checksum'][it])
        ro, rd = O.get_rays(H, W, focal, poses[v])
        rows = O.assemble_ray_batch(ro.reshape(-1, 3)[pix], rd.reshape(-1, 3)[pix], near, far)
        kw = {}
        if mode == 'pytest':                                     # the hooks' draws, in the reference's order (run.py:1776, helpers :319, :378)
            np.random.seed(0)
            kw['t_rand'] = T(np.random.rand(n_rays, 64).astype(np.float32))
            np.random.seed(0)
            kw['noise0'] = T((np.random.rand(n_rays, 64) * 1.0).astype(np.float32))            # raw_noise_std = 1
            np.random.seed(0)
            kw['u'] = T(np.random.rand(n_rays, 64).astype(np.float32))
            np.random.seed(0)
            kw['noise1'] = T((np.random.rand(n_rays, 128) * 1.0).astype(np.float32))
        r = O.render_rays(rows, pc, pf, 64, 64, lindisp=False, white_bkgd=False, **kw)
        target = images[v].reshape(-1, 3)[pix]
        loss = ((r['rgb_map'] - target) ** 2).mean() + ((r['rgb0'] - target) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        for pg in opt.param_groups:
            pg['lr'] = float(g['lrate']) * (0.1 ** (it / (int(g['lrate_decay']) * 1000)))
        np.testing.assert_allclose(float(loss), float(g[f'{mode}/losses'][it]), rtol=2e-5, err_msg=f'iteration {it}')
