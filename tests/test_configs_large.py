"""BASELINE.json configs[3] and configs[4] at their own sizes (the GPU box has no dataset: the real poses / masks /
bounds of SPIn-NeRF scene 1 from tests/golden/scene1_small.npz, rasters resampled to the config's resolution).

  configs[3]  factor 2 (1134 x 2016 = 2,286,144 rays per frame), RGB + normal + multi-view collaborative SDS,
              normalmap_render_factor = 2 (567 x 1008 normal / neighbour-view frames), ray- and term-sharded:
                * the frame renderer at 1134 x 2016: chunk invariance bit-exact, strided ray sample == CPU oracle;
                * ONE iteration with all three guidance terms and the full-size prior: finite non-zero gradients on all
                  48 tensors, every SDS term evaluated exactly once, peak memory < 0.9 of the device;
                * its world-2 twin (two ranks on this one GPU over gloo, rays and SDS terms sharded) gives the same
                  parameter gradients.
  configs[4]  factor 4 (567 x 1008), full guidance, one independent replica per GPU: one replica alone, then two
              replicas started concurrently on this one device -- each reproduces its own solo run (nothing shared:
              no port, no path, no cache) and the two differ from each other (different scenes).
"""
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nerf_oracle as O
from oracle.weights import seeded_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, 'tests', 'golden', 'scene1_small.npz')
VIEWS = list(range(0, 30, 3))                       # 10 views: iteration i with i % 60 == 4 has 5 neighbour views
H2, W2 = 1134, 2016


def N(t):
    return t.detach().cpu().numpy()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene_f2(dev, build_sets=True):
    from mvip_nerf_amd.scene import LLFFScene
    return LLFFScene.from_fixture(FIXTURE, size=(H2, W2), device=dev, views=VIEWS, build_sets=build_sets)


def test_config3_frame_render_1134x2016_vs_oracle(cuda):
    """The renderer's properties at the configs[3] frame: 2,286,144 rays x (64 + 128) samples."""
    from mvip_nerf_amd import run
    from mvip_nerf_amd.replicas import config_args
    sc = _scene_f2(cuda, build_sets=False)
    assert (sc.H, sc.W) == (H2, W2) and abs(sc.focal - 3069.17 / 2) < 1.0
    frac = float(sc.masks.float().mean())
    assert 0.04 < frac < 0.08
    _, te, _, _, _ = run.create_nerf(config_args(), device=cuda)
    for net, seed in ((te['network_fn'], 71), (te['network_fine'], 72)):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(seed).items()})
    pose = sc.poses[3]
    with torch.no_grad():
        a = run.render(sc.H, sc.W, sc.focal, chunk=1 << 15, c2w=pose, near=sc.near, far=sc.far, **te)
        b = run.render(sc.H, sc.W, sc.focal, chunk=sc.H * sc.W, c2w=pose, near=sc.near, far=sc.far, **te)
    assert a[0].shape == (H2, W2, 3)
    bits = lambda t: t.contiguous().view(torch.int32)                   # bit patterns: NaN-safe (an empty ray's disparity)
    for k in range(4):
        assert torch.equal(bits(a[k]), bits(b[k])), k                   # chunk invariance, bit-exact
    for key in ('z_std', 'rgb0'):
        assert torch.equal(bits(a[4][key]), bits(b[4][key])), key
    assert torch.isfinite(a[0]).all() and float(a[2].min()) > 0           # the seeded field is not empty and float(a[2].min()) >= 0 and float(a[2].max()) <= 1 + 1e-5
    ro, rd = O.get_rays(sc.H, sc.W, sc.focal, pose.cpu())
    sel = torch.arange(0, sc.H * sc.W, 9973)                             # 230 rays across the whole frame
    rows = O.assemble_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], sc.near, sc.far)
    pc = {k: torch.from_numpy(v) for k, v in seeded_state_dict(71).items()}
    pf = {k: torch.from_numpy(v) for k, v in seeded_state_dict(72).items()}
    with torch.no_grad():
        ref = O.render_rays(rows, pc, pf, 64, 64, lindisp=True, white_bkgd=True)
    for k, idx in (('rgb_map', 0), ('disp_map', 1), ('acc_map', 2), ('depth_map', 3)):
        got = N(a[idx].reshape(sc.H * sc.W, -1))[sel.numpy()].reshape(ref[k].shape)
        np.testing.assert_allclose(got, ref[k].numpy(), rtol=2e-4, atol=2e-5, err_msg=k)


def _config3_rank(rank, world, port, out):
    """One configs[3] iteration (i = 1024: five neighbour views, normal term active), SDS terms owned per rank.
    world 1 runs it twice: with the configuration's own stochastic render flags (perturb = 1, raw_noise_std = 1), and
    with both off -- the form whose world-2 twin must reproduce it (a rank draws the jitter of ITS rays, so the
    stochastic renders of one and of two ranks are different samples of the same estimator)."""
    from mvip_nerf_amd.guidance.sd_utils import StableDiffusion
    from mvip_nerf_amd.nerf.utils import Pretrain_Model
    from mvip_nerf_amd.replicas import config_args, guidance_opt
    from mvip_nerf_amd.trainer import SecondStageTrainer
    dev = torch.device('cuda', 0)
    d = None
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        d = dist
    try:
        torch.manual_seed(0)
        sc = _scene_f2(dev)
        # SD-1.5-inpaint shapes, random weights (seeded), fp32; eager so that the forward hook counts evaluations, not the
        # warm-up and capture passes of a hipGraph (the graphed path runs in tests/test_configs.py and the replica test)
        sd = StableDiffusion(dev, False, False, use_graphs=False)
        calls, unet_calls = [], []
        for name in ('image_grad', 'colla_view_share', 'colla_last_view_image_grad'):
            f = getattr(sd, name)
            setattr(sd, name, (lambda f, name: (lambda *a, **k: (calls.append(name), f(*a, **k))[1]))(f, name))
        sd.unet.register_forward_hook(lambda m, i, o: unet_calls.append(tuple(i[0].shape)))
        rec = (sc.sets['rays_rgb_clf'][:1024].clone(), sc.sets['rays_inp'][:1024].clone())
        for tag, flags in ((('stochastic', {}),) if world == 1 else ()) + (('deterministic', dict(perturb=0., raw_noise_std=0.)),):
            del calls[:], unet_calls[:]
            tr = SecondStageTrainer(config_args(**flags), sc, dev, guidance=Pretrain_Model(guidance_opt(), dev, {'SD': sd}),
                                    world=world, rank=rank, dist=d, view_shard=True)
            for net, seed in ((tr.kw_train['network_fn'], 71), (tr.kw_train['network_fine'], 72)):
                net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(seed).items()})
            tr.optimizer.step = lambda: None
            torch.cuda.reset_peak_memory_stats(dev)
            loss, n = tr.step(1024, img_i=4, records=rec)
            torch.cuda.synchronize()
            torch.save({'grads': [p.grad.detach().cpu() for p in tr.grad_vars], 'rays': n, 'calls': list(calls),
                        'unet': list(unet_calls), 'loss': float(loss), 'peak': int(torch.cuda.max_memory_allocated(dev)),
                        'total': int(torch.cuda.get_device_properties(dev).total_memory),
                        'masked': int(sc.masked_idx_of(4).numel())}, os.path.join(out, f'c3w{world}r{rank}_{tag}.pt'))
            del tr, loss
            torch.cuda.empty_cache()
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_config3_iteration_rgb_normal_colla_and_world2_twin(tmp_path, cuda):
    out = str(tmp_path)
    torch.cuda.empty_cache()                      # earlier tests' cached blocks are not part of this iteration's footprint
    _config3_rank(0, 1, 0, out)
    for tag in ('stochastic', 'deterministic'):
        ref = torch.load(os.path.join(out, f'c3w1r0_{tag}.pt'))
        assert np.isfinite(ref['loss'])
        assert len(ref['grads']) == 48
        assert all(torch.isfinite(g).all() and float(g.abs().max()) > 0 for g in ref['grads'])
        # every SDS term once: RGB + normal (image_grad x2), four forward-only neighbour views, the last view with its gradient
        assert sorted(ref['calls']) == ['colla_last_view_image_grad'] + ['colla_view_share'] * 4 + ['image_grad'] * 2
        assert len(ref['unet']) == 7 and all(s == (2, 9, 64, 64) for s in ref['unet'])
        assert ref['peak'] < 0.9 * ref['total']
        # rays rendered WITH grad: the masked set at 1134 x 2016, the 567 x 1008 normal frame, the last neighbour view, 2 x 1024
        assert 100_000 < ref['masked'] < 180_000
        assert ref['rays'] == ref['masked'] + 2 * 567 * 1008 + 2 * 1024
    torch.cuda.empty_cache()
    mp.spawn(_config3_rank, args=(2, _free_port(), out), nprocs=2, join=True)
    parts = [torch.load(os.path.join(out, f'c3w2r{r}_deterministic.pt')) for r in range(2)]
    assert sorted(parts[0]['calls'] + parts[1]['calls']) == sorted(ref['calls'])          # each term ran exactly once
    assert len(parts[0]['calls']) <= 4 and len(parts[1]['calls']) <= 4                    # round-robin ownership
    assert len(parts[0]['unet']) + len(parts[1]['unet']) == 7
    assert parts[0]['rays'] + parts[1]['rays'] == ref['rays']
    worst = 0.0
    for k, gr in enumerate(ref['grads']):
        assert torch.equal(parts[0]['grads'][k], parts[1]['grads'][k])                    # identical after the all-reduce
        err = float((parts[0]['grads'][k] - gr).abs().max() / (gr.abs().max() + 1e-30))
        worst = max(worst, err)
        assert err < 3e-3, (k, err)
    print(f'config3 world-2 twin: worst per-tensor gradient deviation {worst:.2e} of the tensor maximum; '
          f'peak memory {ref["peak"] / 2 ** 30:.1f} GiB single, {parts[0]["peak"] / 2 ** 30:.1f} GiB per rank')


def _replica_cmd(scene, basedir, iters=2):
    return [sys.executable, '-m', 'mvip_nerf_amd.replicas', '--child', str(scene), '--iters', str(iters), '--basedir', basedir,
            '--fixture', FIXTURE, '--size', '567x1008']


def test_config4_full_guidance_replicas_share_nothing(tmp_path, cuda):
    """configs[4] = replicas only.  Scene 0 alone, then scenes 0 and 1 CONCURRENTLY on this one device (separate
    processes, no process group): both finish, write their own checkpoints under their own basedir/expname, scene 0's
    result equals its solo run (losses and parameter checksum: nothing leaked between the two), and the scenes differ."""
    from mvip_nerf_amd.replicas import launch
    torch.cuda.empty_cache()                      # this process's cached blocks are not the replicas' to work around
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    solo_dir, pair_dir = str(tmp_path / 'solo'), str(tmp_path / 'pair')
    r = subprocess.run(_replica_cmd(0, solo_dir), env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    solo = json.load(open(os.path.join(solo_dir, 'scene_0', 'replica.json')))
    assert len(solo['losses']) == 2 and all(np.isfinite(solo['losses'])) and os.path.exists(solo['checkpoint'])
    ck = torch.load(solo['checkpoint'], map_location='cpu')
    assert set(ck) == {'global_step', 'network_fn_state_dict', 'network_fine_state_dict', 'optimizer_state_dict'}
    assert all(k.startswith('module.') for k in ck['network_fn_state_dict'])
    recs = launch(2, 2, pair_dir, devices=[0, 0], fixture=FIXTURE, timeout=2400)
    assert all('error' not in r for r in recs), recs
    assert recs[0]['pid'] != recs[1]['pid'] and recs[0]['checkpoint'] != recs[1]['checkpoint']
    assert all(os.path.exists(r['checkpoint']) for r in recs)
    # the concurrent replica of scene 0 reproduces the solo one (atomic-add ordering only)
    np.testing.assert_allclose(recs[0]['losses'], solo['losses'], rtol=2e-3)
    np.testing.assert_allclose(recs[0]['param_checksum'], solo['param_checksum'], rtol=1e-4, atol=0.1)
    assert abs(recs[0]['losses'][0] - recs[1]['losses'][0]) > 1e-6              # another scene, another result
    total = torch.cuda.get_device_properties(cuda).total_memory
    assert recs[0]['peak_bytes'] + recs[1]['peak_bytes'] < 0.9 * total
    print('config4: solo', solo['iterations_per_sec'], 'it/s; two concurrent replicas on one GPU',
          [round(r['iterations_per_sec'], 3) for r in recs])
