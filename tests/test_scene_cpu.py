"""Host-side pieces of the real-scene second stage on the CPU: the pre-baked fp16 ray records
(`scene.build_ray_sets`, DS_NeRF/run.py:613-711) against the records produced by executing the reference's own
statements (tests/golden/ray_sets.npz), the device-side epoch sampler, and the view-sharded SDS evaluation
(`sds_shard.evaluate`) under gloo with world 2 and 3 against the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_ray_sets_match_reference_records(golden):
    from mvip_nerf_amd.scene import build_ray_sets
    g = golden('ray_sets')
    H, W, focal = 9, 13, float(g['poses'][0, 2, 4])
    sets = build_ray_sets(g['images'], g['poses'], g['masks'], g['inpainted_depths'], (H, W, focal), g['i_train'], 'all')
    for k, want in (('rays_rgb', g['rays_rgb']), ('rays_rgb_clf', g['rays_rgb_clf']), ('rays_rgb_sds', g['rays_rgb_sds']),
                    ('rays_inp', g['rays_inp_all'])):
        assert sets[k].dtype == np.float16 and sets[k].shape == want.shape, k
        np.testing.assert_array_equal(sets[k].view(np.uint16), want.view(np.uint16), err_msg=k)     # bit-exact
    # the run.py:712-713 defect, decided: depth records restricted to unmasked (default) or masked pixels
    n_mask = int((g['masks'][g['i_train']] == 1).sum())
    un = build_ray_sets(g['images'], g['poses'], g['masks'], g['inpainted_depths'], (H, W, focal), g['i_train'])
    ma = build_ray_sets(g['images'], g['poses'], g['masks'], g['inpainted_depths'], (H, W, focal), g['i_train'], 'masked')
    assert ma['rays_inp'].shape[0] == n_mask and un['rays_inp'].shape[0] == want.shape[0] - n_mask
    assert un['rays_rgb_clf'].shape[0] == un['rays_inp'].shape[0]


def test_llff_scene_fields_and_epoch_sampler(golden):
    from mvip_nerf_amd.scene import LLFFScene
    g = golden('ray_sets')
    bds = np.array([[1.4, 7.0]] * 5, np.float32)
    sc = LLFFScene(g['images'], g['poses'], bds, g['masks'], g['inpainted_depths'], device='cpu', i_train=g['i_train'])
    assert (sc.H, sc.W) == (9, 13) and abs(sc.near - 1.4 * .9) < 1e-6 and sc.far == 7.0
    assert sc.poses.shape == (5, 3, 4) and sc.masks.dtype == torch.bool
    for v in range(5):
        assert torch.equal(sc.masked_idx_of(v), torch.nonzero(torch.from_numpy(g['masks'][v] == 1).reshape(-1))[:, 0])
    n = sc.sets['rays_rgb_clf'].shape[0]
    seen = []
    for _ in range((n + 15) // 16):
        rays, rgb, label = sc.next_batch('rays_rgb_clf', 16)
        assert rays.shape[0] == 2 and rays.shape[2] == 3 and rays.dtype == torch.float16 and (label == 0).all()
        seen.append(rays.shape[1])
    assert sum(seen) == n                                # one epoch = every record exactly once
    rays, rgb, dep = sc.next_batch('rays_inp', 7)        # next epoch starts with a fresh permutation
    assert rays.shape == (2, 7, 3) and dep.shape == (7,)
    small = LLFFScene.from_fixture(os.path.join(os.path.dirname(__file__), 'golden', 'scene1_small.npz'), device='cpu',
                                   views=[0, 1], build_sets=False)
    assert (small.H, small.W) == (141, 252) and abs(small.focal - 3069.17 / 16) < 0.5 and 1.0 < small.near < small.far


# ---- view-sharded SDS terms --------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_terms(w, x, n_latent=3):
    """Frames rendered from parameters w (each rank holds the full frames here; the test is about ownership and
    result exchange), three image terms (one needing the latent sum) and `n_latent` latent terms (0: the collaborative
    term with a single neighbour view, whose last-view term still asks for the -- then all-zero -- latent sum)."""
    from mvip_nerf_amd.sds_shard import Term
    frames = [torch.tanh(x[k] @ w).reshape(1, 3, 4, 5) for k in range(3)]
    g = torch.Generator().manual_seed(0)
    K = [torch.randn(1, 3, 4, 5, generator=g) for _ in range(3)]
    Lt = [torch.randn(1, 4, 8, 8, generator=g) for _ in range(max(3, n_latent))]

    def image_term(k, share):
        f = frames[k].detach().clone().requires_grad_(True)
        loss = (f * K[k]).sum() + (f ** 2).sum() * 0.5
        if share is not None:
            loss = loss + f.mean() * (share.sum() + 1.0)
        loss.backward()
        return f.grad
    terms = [Term('rgb', 2, lambda s: image_term(0, None), image=frames[0]),
             Term('normal', 2, lambda s: image_term(1, None), image=frames[1]),
             Term('colla_last', 2, lambda s: image_term(2, s), image=frames[2], latent_shape=(1, 4, 8, 8), needs_latent_sum=True)]
    terms += [Term(f'colla_{k}', 1, lambda k=k: Lt[k] * (k + 1), latent_shape=(1, 4, 8, 8)) for k in range(n_latent)]
    return terms


def _shard_worker(rank, world, port, out, n_latent=3):
    from mvip_nerf_amd import sds_shard
    d = None
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        d = dist
    try:
        g = torch.Generator().manual_seed(1)
        w = torch.randn(6, 60, generator=g).requires_grad_(True)
        x = torch.randn(3, 1, 6, generator=g)
        calls = []
        terms = _toy_terms(w, x, n_latent)
        for t in terms:
            t.run = (lambda f, name: (lambda *a: (calls.append(name), f(*a))[1]))(t.run, t.name)
        loss = sds_shard.evaluate(terms, rank, world, d, torch.device('cpu'))
        loss.backward()
        torch.save({'grad': w.grad.clone(), 'calls': calls}, os.path.join(out, f'w{world}r{rank}.pt'))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_view_sharded_terms_equal_single_process(tmp_path, world):
    _shard_worker(0, 1, 0, str(tmp_path))
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = torch.load(os.path.join(str(tmp_path), 'w1r0.pt'))
    assert sorted(ref['calls']) == ['colla_0', 'colla_1', 'colla_2', 'colla_last', 'normal', 'rgb']
    owned = []
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f'w{world}r{r}.pt'))
        # here every rank holds the full frames, so each rank's surrogate gradient is the complete one
        np.testing.assert_allclose(got['grad'].numpy(), ref['grad'].numpy(), rtol=1e-5, atol=1e-6)
        owned += got['calls']
    assert sorted(owned) == sorted(ref['calls'])          # every term evaluated exactly once across the ranks


def test_view_sharded_terms_single_neighbour_view(tmp_path):
    """V == 1: no phase-1 term exists, the last-view term still needs the latent sum -> zeros, not None (it raised
    AttributeError in share_sum.detach()); world 1 and world 2 agree."""
    _shard_worker(0, 1, 0, str(tmp_path), 0)
    mp.spawn(_shard_worker, args=(2, _free_port(), str(tmp_path), 0), nprocs=2, join=True)
    ref = torch.load(os.path.join(str(tmp_path), 'w1r0.pt'))
    assert sorted(ref['calls']) == ['colla_last', 'normal', 'rgb']
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), f'w2r{r}.pt'))
        np.testing.assert_allclose(got['grad'].numpy(), ref['grad'].numpy(), rtol=1e-5, atol=1e-6)


def test_seven_terms_over_eight_ranks(tmp_path):
    """configs[3] on a full node: 7 SDS terms (RGB, normal, the collaborative term's last view + 4 latent shares) over 8
    ranks (VERDICT r4 task 4c).  Ownership is round-robin, so rank 7 owns NOTHING: it contributes zeros to the latent
    all_reduce, receives the three image-gradient broadcasts, and ends with the same surrogate gradient as everybody
    else; every term is evaluated exactly once across the node."""
    from mvip_nerf_amd import sds_shard
    world = 8
    _shard_worker(0, 1, 0, str(tmp_path), 4)
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path), 4), nprocs=world, join=True)
    ref = torch.load(os.path.join(str(tmp_path), 'w1r0.pt'))
    assert len(ref['calls']) == 7
    owned = []
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f'w{world}r{r}.pt'))
        np.testing.assert_allclose(got['grad'].numpy(), ref['grad'].numpy(), rtol=1e-5, atol=1e-6)
        assert len(got['calls']) == (1 if r < 7 else 0), (r, got['calls'])      # one term per rank, the eighth rank idle
        owned += got['calls']
    assert sorted(owned) == sorted(ref['calls'])
    assert [sds_shard.owner_of(k, world) for k in range(7)] == list(range(7))
