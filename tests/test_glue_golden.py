"""The glue rows of the hot path against fixtures produced by the REAL reference (oracle/gen_golden_glue.py):
`render_path_4view` (DS_NeRF/run.py:1365-1401), `Pretrain_Model.cal_loss` (DS_NeRF/nerf/utils.py:222-311) and two
iterations of the second-stage loop body (DS_NeRF/run.py:798-1041, loss composition :1000-1027, lr :1035-1039),
the latter two with the stand-in diffusion networks of oracle/sds_standin.py on both sides."""
import types

import numpy as np
import pytest
import torch

from oracle.sds_standin import TinyVAE, TinyUNet, TinyScheduler, prompt_embedding
from oracle.weights import seeded_state_dict

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def nerf_args(**over):
    a = types.SimpleNamespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=64, alpha_model_path=None,
        netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=3e-3,
        basedir='/tmp/mvip_glue', expname='none', ft_path=None, no_reload=True, perturb=0., N_samples=64,
        white_bkgd=True, raw_noise_std=0., dataset_type='llff', no_ndc=True, lindisp=True, sigma_loss=False,
        N_rand=16, chunk=1 << 15, lrate_decay=10, depth_lambda=0.1, sds_loss_weight=1e-4, no_coarse=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def guidance_opt(**over):
    o = types.SimpleNamespace(
        text='a stone bench in a park', text_normal='a normal map of a stone bench', is_rgb_guidance=True,
        is_colla_guidance=False, is_normal_guidance=False, rgb_guidance_scale=7.5, colla_guidance_scale=7.5,
        normal_guidance_scale=1.5, normal_start=500, lambda_guidance=1, uniform_sphere_rate=0)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def standin_sd(dev, draws):
    """Our StableDiffusion wrapper on the stand-in networks, replaying the reference's recorded CPU draws."""
    from mvip_nerf_amd.guidance.sd_utils import StableDiffusion
    cache = {}

    def encode_prompt(p, cfg):
        if (p, cfg) not in cache:
            cache[(p, cfg)] = prompt_embedding(p, cfg).to(dev)
        return cache[(p, cfg)]
    nets = types.SimpleNamespace(vae=TinyVAE().to(dev), unet=TinyUNet().to(dev), encode_prompt=encode_prompt,
                                 alphas_cumprod=TinyScheduler().alphas_cumprod)
    sd = StableDiffusion(dev, False, False, networks=nets)
    it = iter(draws)
    sd._randn = lambda shape, dtype=torch.float32: next(it).to(dev)
    return sd


def load_nets(kw, sc, sf):
    for net, seed in ((kw['network_fn'], sc), (kw['network_fine'], sf)):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(seed).items()})


def test_render_path_4view_golden(golden, cuda):
    from mvip_nerf_amd import run
    g = golden('render_path_4view')
    tr, te, _, _, _ = run.create_nerf(nerf_args(), device=cuda)
    load_nets(te, int(g['seed_coarse']), int(g['seed_fine']))
    poses = T(g['poses'][:, :3, :4], cuda)
    hwf = [int(g['hwf'][0]), int(g['hwf'][1]), float(g['hwf'][2])]
    kw = dict(te, near=float(g['near']), far=float(g['far']))
    for it in g['iters']:
        with torch.no_grad():
            rgbs, disps, msel = run.render_path_4view(int(it), g['masks'], poses, hwf, 1 << 15, kw, render_factor=2,
                                                      need_alpha=True)
        assert rgbs.shape == g[f'rgbs_{it}'].shape and disps.shape == g[f'disps_{it}'].shape
        np.testing.assert_array_equal(np.asarray(msel), g[f'masks_{it}'])          # the same neighbour selection
        np.testing.assert_allclose(N(rgbs), g[f'rgbs_{it}'], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(N(disps), g[f'disps_{it}'], rtol=2e-4, atol=2e-5)
    assert g['rgbs_2'].shape[0] == 4 and g['rgbs_65'].shape[0] == 5 and g['rgbs_59'].shape[0] == 5


@pytest.mark.parametrize('tag', ['rgb', 'rgb_normal', 'rgb_normal_gated', 'all', 'colla_gated', 'normal_only'])
def test_cal_loss_golden(golden, cuda, tag):
    """Dispatch, gates on i, term order, returned value and the gradients reaching the three inputs."""
    from mvip_nerf_amd.nerf.utils import Pretrain_Model
    g = golden('cal_loss')
    flags = g[f'{tag}/flags']
    i = int(g[f'{tag}/i'])
    torch.manual_seed(int(g[f'{tag}/seed']))
    for _ in range(3):
        torch.rand(1)                                  # rand_poses' draws precede the terms' draws on the reference side
    draws = [torch.randn(1, 4, 64, 64) for _ in range(4 * 5)]
    sd = standin_sd(cuda, draws)
    opt = guidance_opt(is_rgb_guidance=bool(flags[0]), is_colla_guidance=bool(flags[1]), is_normal_guidance=bool(flags[2]))
    pm = Pretrain_Model(opt, cuda, {'SD': sd})
    p = T(g['pred'], cuda).requires_grad_(True)
    nm = T(g['normal'], cuda).requires_grad_(True)
    r4 = T(g['rgbs4'], cuda).requires_grad_(True)
    loss = pm.cal_loss(i, r4, nm, None, p, None, T(g['mask'], cuda), T(g['mask4'], cuda), 1)
    np.testing.assert_allclose(N(loss).reshape(-1), g[f'{tag}/loss'], rtol=1e-6)
    assert pm.global_step == int(g[f'{tag}/global_step'])
    (float(g['upstream']) * loss).sum().backward()
    for name, t in (('d_pred', p), ('d_normal', nm), ('d_rgbs4', r4)):
        want = g[f'{tag}/{name}']
        got = np.zeros_like(want) if t.grad is None else N(t.grad)
        scale = np.abs(want).max()
        if scale == 0:
            assert np.abs(got).max() == 0, name
        else:
            np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4 * scale, err_msg=f'{tag} {name}')


def test_cal_loss_consumes_reference_random_stream(cuda):
    """rand_poses' draws (DS_NeRF/nerf/utils.py:119-135) are consumed from the device generator: 3 x rand(B)."""
    from mvip_nerf_amd.nerf.utils import Pretrain_Model

    class NoSD(torch.nn.Module):
        reference_rng = True

        def train_step_sd(self, *a, **k):
            return torch.ones(1, device=cuda)
    pm = Pretrain_Model(guidance_opt(), cuda, {'SD': NoSD()})
    torch.manual_seed(5)
    pm.cal_loss(1, None, None, None, None, None, None, None, 1)
    after = torch.rand(1, device=cuda)
    torch.manual_seed(5)
    for _ in range(3):
        torch.rand(1, device=cuda)
    assert torch.equal(after, torch.rand(1, device=cuda))


def test_trainer_two_steps_golden(golden, cuda):
    """Two iterations of the reference's second-stage loop (configs[3] shape: RGB + normal + collaborative SDS) vs
    SecondStageTrainer on an LLFFScene built from the same arrays, replaying the same ray batches and draws."""
    from mvip_nerf_amd.nerf.utils import Pretrain_Model
    from mvip_nerf_amd.scene import LLFFScene
    from mvip_nerf_amd.trainer import SecondStageTrainer
    g = golden('trainer_two_steps')
    args = nerf_args(N_rand=16, is_normal_guidance=True, is_colla_guidance=True, normalmap_render_factor=2)
    scene = LLFFScene(g['images'], g['poses'], g['bds'], g['masks'], g['inpainted_depths'], device=cuda, build_sets=False)
    torch.manual_seed(int(g['torch_seed']))
    draws = []
    for _ in range(2):
        for _ in range(3):
            torch.rand(1)
        draws += [torch.randn(1, 4, 64, 64) for _ in range(20)]
    sd = standin_sd(cuda, draws)
    pm = Pretrain_Model(guidance_opt(is_normal_guidance=True, is_colla_guidance=True, normal_start=0), cuda, {'SD': sd})
    tr = SecondStageTrainer(args, scene, cuda, guidance=pm)
    load_nets(tr.kw_train, int(g['seed_coarse']), int(g['seed_fine']))
    losses, lrs = [], []
    for k in range(2):
        rec = (T(g['clf_batches'][k], cuda), T(g['inp_batches'][k], cuda))
        loss, n = tr.step(1 + k, img_i=int(g['img_i'][k]), records=rec)
        losses.append(float(loss))
        lrs.append(tr.optimizer.param_groups[0]['lr'])
    np.testing.assert_allclose(losses, g['losses'], rtol=2e-3)
    np.testing.assert_allclose(lrs, g['lrs'], rtol=1e-12)
    assert tr.global_step == int(g['global_step']) == 2
    # gradients of the second iteration (taken with the weights the first Adam step produced)
    for prefix, net in (('coarse.', tr.kw_train['network_fn']), ('fine.', tr.kw_train['network_fine'])):
        for k, p in net.named_parameters():
            gr = N(p.grad).astype(np.float64).ravel()
            stat = g[f'gstat/{prefix}{k}']
            np.testing.assert_allclose(np.sqrt((gr * gr).sum()), stat[2], rtol=2e-2, err_msg=prefix + k)
            # parameters after two Adam steps (lr = 3e-3): Adam's normalised update lr * m / sqrt(v) turns a relative
            # gradient difference into an absolute parameter difference of that fraction of lr, and entries whose
            # gradient is at rounding level may land up to 2 lr apart per step; bound both
            v = N(p).astype(np.float64).ravel()
            want = g[f'pval/{prefix}{k}']
            d = np.abs(v[g[f'pidx/{prefix}{k}']] - want)
            assert np.mean(d > 1e-3) <= 0.05 and d.max() <= 4.1 * 3e-3, (prefix + k, d.max(), np.mean(d > 1e-3))
