"""Host-side mirror of the renderer half of the reference's `DS_NeRF/run.py`: `batchify`,
`run_network`, `batchify_rays`, `render`, `render_path_4view`, `create_nerf`, `render_rays`,
`depth2xyz_torch`, `depth2normal_geo` -- same names, positional orders, defaults and return
structures, executed by the HIP kernels behind `mvip_nerf_amd.ops`.

Randomness: the reference draws `torch.rand(z_vals.shape)`, `torch.randn(raw[...,3].shape)` and
`torch.rand(cdf.shape[:-1] + [N])` from the default device generator, in that order, once per
ray chunk.  This module draws the same shapes in the same order from the same generator, so a
seeded run consumes the RNG stream exactly as the reference does on the same device; the draws
are passed to the kernels as inputs (which is also what the reference's `pytest=True` hooks do).
"""
import os

import numpy as np
import torch

from . import ops
from .run_nerf_helpers import (NeRF, get_embedder, get_rays, ndc_rays, _density_noise, _uniforms)

DEBUG = False


def batchify(fn, chunk):
    """DS_NeRF/run.py:1096-1105."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)

    return ret


def run_network(inputs2, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """DS_NeRF/run.py:1108-1124.  With the HIP-backed NeRF the encoding is fused into the MLP
    kernel, so nothing is materialised; any other `fn` gets the reference's embed/cat/chunk flow."""
    if isinstance(fn, NeRF) and viewdirs is not None:
        flat = torch.reshape(inputs2, [-1, inputs2.shape[-1]])
        dirs = viewdirs[:, None].expand(inputs2.shape).reshape(-1, 3)
        out = fn.query_points(flat, dirs)
        return torch.reshape(out, list(inputs2.shape[:-1]) + [out.shape[-1]])
    inputs_flat = torch.reshape(inputs2, [-1, inputs2.shape[-1]])
    embedded = embed_fn(inputs_flat)
    if viewdirs is not None:
        input_dirs = viewdirs[:, None].expand(inputs2.shape)
        input_dirs_flat = torch.reshape(input_dirs, [-1, input_dirs.shape[-1]])
        embedded = torch.cat([embedded, embeddirs_fn(input_dirs_flat)], -1)
    outputs_flat = batchify(fn, netchunk)(embedded)
    return torch.reshape(outputs_flat, list(inputs2.shape[:-1]) + [outputs_flat.shape[-1]])


def batchify_rays(rays_flat, chunk=1024 * 32, need_alpha=False, detach_weights=False, **kwargs):
    """DS_NeRF/run.py:1127-1140."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], need_alpha=need_alpha, detach_weights=detach_weights, **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, depths=None, need_alpha=False, detach_weights=False, patch=None, **kwargs):
    """DS_NeRF/run.py:1143-1219.  Returns [rgb_map, disp_map, acc_map, depth_map, extras]."""
    scalar_bounds = not torch.is_tensor(near) and not torch.is_tensor(far)
    fast = use_viewdirs and not ndc and depths is None and c2w_staticcam is None and scalar_bounds
    if rays is not None and c2w is None and torch.is_tensor(rays[1]) and rays[1].dtype != torch.float32:
        # the reference's pre-baked fp16 ray batches (run.py:639, :978): view directions are normalised in the
        # rays' own precision before the cast to float (run.py:1186-1187) -- the general assembly does exactly that
        fast = False
    if c2w is not None:
        c2w = torch.as_tensor(c2w)
        if fast and patch is None:
            sh = (H, W, 3)
            rows = ops.ray_rows_from_pose(c2w, H, W, focal, near, far)
        else:
            rays_o, rays_d = ops.get_rays(H, W, focal, c2w, patch)
    else:
        rays_o, rays_d = rays
    if not (c2w is not None and fast and patch is None):
        sh = tuple(rays_d.shape)
        if fast:
            rows = ops.ray_rows(rays_o, rays_d, near, far)
        else:
            rows = _assemble_rows_general(H, W, focal, rays_o, rays_d, ndc, near, far, use_viewdirs,
                                          c2w_staticcam, depths)
    all_ret = batchify_rays(rows, chunk, need_alpha=need_alpha, detach_weights=detach_weights, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map', 'depth_map']
    ret_list = [all_ret[k] for k in k_extract]
    ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
    return ret_list + [ret_dict]


def render_sharded(H, W, focal, c2w, rank, world, dist, chunk=1024 * 32, ndc=True, near=0., far=1., use_viewdirs=False,
                   c2w_staticcam=None, depths=None, row_fn=None, **kwargs):
    """ONE frame over all ranks (strong scaling; extension, no reference counterpart -- the reference's only multi-GPU
    mechanism is nn.DataParallel around the MLPs, DS_NeRF/run.py:1491, :1527): rank r renders the contiguous block
    dist_utils.block_bounds(H*W, r, world) of the frame's rays through batchify_rays (DS_NeRF/run.py:1127-1140, the loop
    being split), then ONE all_gather of [rays, 6] = (rgb, disp, acc, depth) blocks assembles the maps on every rank.
    Returns [rgb_map [H,W,3], disp_map [H,W], acc_map [H,W], depth_map [H,W]] -- render()'s first four outputs; the
    per-sample extras stay sharded (nobody reads them across ranks).  Rays of different blocks are independent, so the
    values equal render()'s (bit for bit when the chunk boundaries coincide; the kernels are chunk-invariant anyway,
    tests/test_configs_large.py).  ndc / near / far / use_viewdirs / c2w_staticcam / depths mean what they mean in
    render() (same defaults): the view-direction, non-NDC, scalar-bounds case builds its block of ray rows in one launch,
    every other case assembles the frame's rows as render() does (DS_NeRF/run.py:1182-1207) and takes the block.
    `row_fn(lo, hi)` overrides the ray-row source (CPU tests of the collective)."""
    from .dist_utils import block_bounds, all_gather_blocks
    n = H * W
    lo, hi = block_bounds(n, rank, world)
    if row_fn is not None:
        rows = row_fn(lo, hi)
    else:
        c2w = torch.as_tensor(c2w)
        scalar_bounds = not torch.is_tensor(near) and not torch.is_tensor(far)
        if use_viewdirs and not ndc and depths is None and c2w_staticcam is None and scalar_bounds:      # render()'s `fast`
            sel = torch.arange(lo, hi, device=c2w.device, dtype=torch.int64)
            rows = ops.ray_rows_from_pose(c2w, H, W, focal, near, far, sel=sel)
        else:
            rays_o, rays_d = ops.get_rays(H, W, focal, c2w)
            if torch.is_tensor(near):
                near = near.reshape(-1, 1)
            if torch.is_tensor(far):
                far = far.reshape(-1, 1)
            rows = _assemble_rows_general(H, W, focal, rays_o, rays_d, ndc, near, far, use_viewdirs, c2w_staticcam,
                                          depths)[lo:hi].contiguous()
    if rows.shape[0] > 0:
        ret = batchify_rays(rows, chunk, **kwargs)
        local = torch.cat([ret['rgb_map'], ret['disp_map'][:, None], ret['acc_map'][:, None], ret['depth_map'][:, None]], -1)
    else:
        local = rows.new_zeros((0, 6))
    full = all_gather_blocks(local, n, rank, world, dist)
    return [full[:, 0:3].reshape(H, W, 3), full[:, 3].reshape(H, W), full[:, 4].reshape(H, W), full[:, 5].reshape(H, W)]


def _assemble_rows_general(H, W, focal, rays_o, rays_d, ndc, near, far, use_viewdirs, c2w_staticcam, depths):
    """The uncommon branches of render()'s row assembly (ndc, static camera, per-ray bounds,
    depth column), DS_NeRF/run.py:1182-1207, as tensor algebra."""
    viewdirs = None
    if use_viewdirs:
        viewdirs = rays_d
        if c2w_staticcam is not None:
            rays_o, rays_d = ops.get_rays(H, W, focal, torch.as_tensor(c2w_staticcam))
        viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
        viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d)
    rays_o = torch.reshape(rays_o, [-1, 3]).float()
    rays_d = torch.reshape(rays_d, [-1, 3]).float()
    near_t = near * torch.ones_like(rays_d[..., :1])
    far_t = far * torch.ones_like(rays_d[..., :1])
    cols = [rays_o, rays_d, near_t, far_t]
    if depths is not None:
        cols.append(depths.reshape(-1, 1))
    if use_viewdirs:
        cols.append(viewdirs)
    return torch.cat(cols, -1).contiguous()


# render_rays as two launches per chunk for no-grad renders of the native networks (64 coarse + <= 64 fine samples);
# MVIP_FUSED_RENDER=0 restores the six-launch chain (A/B switch; the outputs are bit-identical either way).
# The two-launch form is chosen BY CHUNK SIZE: it exists to cut launches where launches matter (small renders); at frame
# size its compositing tail -- one wave of the workgroup at work, seven gone -- costs 0.5 % of the frame
# (profiles/r3_fused_render_ab.json: 303.3-303.8 vs 301.5-301.9 ms), so chunks above FUSED_RENDER_MAX_RAYS take the chain.
FUSED_RENDER = bool(int(os.environ.get('MVIP_FUSED_RENDER', '1')))
FUSED_RENDER_MAX_RAYS = int(os.environ.get('MVIP_FUSED_RENDER_MAX_RAYS', '4096'))


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., pytest=False,
                sigma_loss=None, verbose=False, need_alpha=False, detach_weights=False, coarse_grad=True):
    """DS_NeRF/run.py:1703-1847: stratified depths -> coarse MLP -> compositing -> inverse-CDF
    resampling + merge -> fine MLP -> compositing.  Five kernel launches per chunk on the native
    path (z, MLP, composite, sample+merge, MLP, composite) instead of ~150 torch ops.

    `coarse_grad` (extension, default = the reference's behaviour): with N_importance > 0 the fine outputs depend on
    the coarse network only through the DETACHED resampled depths (run.py:1812), so a caller that uses only
    rgb_map / disp_map / depth_map / acc_map of a render (the masked set, the normal frame, the neighbour views of
    run.py:919-974) gets exactly zero gradient from it into the coarse network; coarse_grad=False runs that pass
    without autograd (no activation stash, no backward: a third of the points), changing no value and no gradient."""
    ray_batch = ray_batch.float() if ray_batch.dtype != torch.float32 else ray_batch
    ray_batch = ray_batch.contiguous()
    N_rays, ncols = ray_batch.shape
    dev = ray_batch.device
    has_dirs = ncols > 9
    if ncols == 11:
        rows = ray_batch
    elif has_dirs:
        rows = torch.cat([ray_batch[:, :8], ray_batch[:, -3:]], -1).contiguous()
    else:
        rows = ray_batch

    def query(z, net):
        if has_dirs and isinstance(net, NeRF) and getattr(network_query_fn, '_mvip_native', False):
            return net.query_rays(rows, z)
        pts = rows[:, None, 0:3] + rows[:, None, 3:6] * z[:, :, None]
        return network_query_fn(pts, rows[:, 8:11] if has_dirs else None, net)

    t_rand = None
    if perturb > 0.:
        t_rand = torch.rand((N_rays, N_samples), device=dev)
        if pytest:
            np.random.seed(0)
            t_rand = torch.tensor(np.random.rand(N_rays, N_samples), dtype=torch.float32, device=dev)

    coarse_net = network_fn if network_fn is not None else (
        network_fine.alpha_model if getattr(network_fine, 'alpha_model', None) is not None else network_fine)
    fine_net = network_fn if network_fine is None else network_fine
    if (FUSED_RENDER and not torch.is_grad_enabled() and 0 < N_rays <= FUSED_RENDER_MAX_RAYS and N_samples == 64 and 0 < N_importance <= 64
            and ncols == 11 and sigma_loss is None and getattr(network_query_fn, '_mvip_native', False)
            and isinstance(coarse_net, NeRF) and isinstance(fine_net, NeRF)):
        c16, f16 = coarse_net._infer16(), fine_net._infer16()
        if c16 is not None and f16 is not None and N_samples + N_importance == 128:
            # TWO launches for the whole chunk (csrc/mlp_fwd16.hip, FUSE = 1 / 2): the random draws are made in the order of
            # the unfused path below, every output is bit-identical to it (tests/test_render.py)
            noise0 = _density_noise((N_rays, N_samples), raw_noise_std, pytest, dev)
            u = _uniforms((N_rays,), N_importance, perturb == 0., pytest, dev)
            rgb0, disp0, acc0, alpha0, z_vals, z_std = ops.render_coarse_fused(c16, rows, lindisp, t_rand, noise0, u,
                                                                               white_bkgd, need_alpha)
            noise = _density_noise((N_rays, N_samples + N_importance), raw_noise_std, pytest, dev)
            rgb_map, disp_map, acc_map, weights, depth_map, alpha, raw = ops.render_fine_fused(
                f16, rows, z_vals, noise, white_bkgd, need_alpha, retraw)
            ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'depth_map': depth_map,
                   'weights': weights, 'z_vals': z_vals}
            if retraw:
                ret['raw'] = raw
            if need_alpha:
                ret['alpha'], ret['alpha0'] = alpha, alpha0
            ret['rgb0'], ret['disp0'], ret['acc0'], ret['z_std'] = rgb0, disp0, acc0, z_std
            return ret
    z_vals = ops.stratified_z(rows, N_samples, lindisp, t_rand)

    with torch.set_grad_enabled(torch.is_grad_enabled() and (coarse_grad or N_importance <= 0)):
        raw = query(z_vals, coarse_net)
        noise = _density_noise((N_rays, N_samples), raw_noise_std, pytest, dev)
        rgb_map, disp_map, acc_map, weights, depth_map, alpha = ops.composite(
            raw, z_vals, rows, noise, white_bkgd, detach_weights, need_alpha)

    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0, alpha0 = rgb_map, disp_map, acc_map, alpha
        u = _uniforms((N_rays,), N_importance, perturb == 0., pytest, dev)
        z_samples, z_vals, z_std, _, _ = ops.sample_pdf_merge(z_vals, weights, u)
        run_fn = network_fn if network_fine is None else network_fine
        raw = query(z_vals, run_fn)
        noise = _density_noise((N_rays, N_samples + N_importance), raw_noise_std, pytest, dev)
        rgb_map, disp_map, acc_map, weights, depth_map, alpha = ops.composite(
            raw, z_vals, rows, noise, white_bkgd, detach_weights, need_alpha)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'depth_map': depth_map,
           'weights': weights, 'z_vals': z_vals}
    if retraw:
        ret['raw'] = raw
    if need_alpha:
        ret['alpha'] = alpha
        ret['alpha0'] = alpha0          # NameError without N_importance, like the reference (run.py:1831)
    if N_importance > 0:
        ret['rgb0'] = rgb_map_0
        ret['disp0'] = disp_map_0
        ret['acc0'] = acc_map_0
        ret['z_std'] = z_std
    if sigma_loss is not None and ncols > 11:
        depths = ray_batch[:, 8]
        ret['sigma_loss'] = sigma_loss.calculate_loss(rows[:, 0:3], rows[:, 3:6], rows[:, 8:11],
                                                      rows[:, 6:7], rows[:, 7:8], depths, network_query_fn,
                                                      network_fine)
    if DEBUG:
        for k in ret:
            if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret


def _write_png(path, rgb8):
    """Minimal 8-bit RGB PNG writer (the reference uses imageio, which is not a dependency here)."""
    import struct
    import zlib
    h, w, _ = rgb8.shape
    raw = b''.join(b'\x00' + rgb8[y].tobytes() for y in range(h))
    def chunk(tag, data):
        c = struct.pack('>I', len(data)) + tag + data
        return c + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0))
                + chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def render_path(render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0,
                disp_require_grad=False, need_alpha=False, rgb_require_grad=False, detach_weights=False,
                patch_len=None, masks=None):
    """DS_NeRF/run.py:1222-1362: render a list of poses; returns (rgbs, disps, (Xs, Ys)) -- numpy
    stacks unless *_require_grad, in which case torch stacks with autograd history.  With `savedir`
    the reference's on-disk layout is produced (rgb/*.png, depth|disp|weight|z|alpha/*.npy,
    pose/*.txt, intrinsics.txt, images/*.png for gt)."""
    import random
    from .run_nerf_helpers import to8b
    H, W, focal = hwf
    if render_factor != 0:
        H = H // render_factor
        W = W // render_factor
        focal = focal / render_factor
    K = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]])
    if savedir is not None:
        os.makedirs(savedir, exist_ok=True)
        np.savetxt(os.path.join(savedir, 'intrinsics.txt'), K)
    rgbs, disps, Xs, Ys = [], [], [], []
    for i, c2w in enumerate(render_poses):
        if disp_require_grad or rgb_require_grad:
            patch = None
            if patch_len is not None:
                masked = np.where(masks[i] != 0)
                masked = (masked[0] // render_factor, masked[1] // render_factor)
                Xs.append(random.randint(masked[0].min(), max(masked[0].max() - patch_len[0], masked[0].min())))
                Ys.append(random.randint(masked[1].min(), max(masked[1].max() - patch_len[1], masked[1].min())))
                patch = (Xs[-1], Ys[-1], patch_len[0], patch_len[1])
            rgb, disp, acc, depth, extras = render(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], retraw=True,
                                                   need_alpha=need_alpha, detach_weights=detach_weights,
                                                   patch=patch, **render_kwargs)
        else:
            with torch.no_grad():
                rgb, disp, acc, depth, extras = render(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], retraw=True,
                                                       need_alpha=need_alpha, **render_kwargs)
        disps.append(disp if disp_require_grad else disp.detach().cpu().numpy())
        rgbs.append(rgb if rgb_require_grad else rgb.detach().cpu().numpy())
        if savedir is not None:
            dirs = {k: os.path.join(savedir, k) for k in ('rgb', 'depth', 'disp', 'weight', 'z', 'pose', 'images')}
            if need_alpha:
                dirs['alpha'] = os.path.join(savedir, 'alpha')
            for d in dirs.values():
                os.makedirs(d, exist_ok=True)
            last = rgbs[-1] if not rgb_require_grad else rgbs[-1].detach().cpu().numpy()
            _write_png(os.path.join(dirs['rgb'], '{:06d}.png'.format(i)), to8b(np.nan_to_num(last)))
            if gt_imgs is not None:
                gt = gt_imgs[i]
                gt = gt.detach().cpu().numpy() if torch.is_tensor(gt) else np.asarray(gt)
                _write_png(os.path.join(dirs['images'], '{:06d}.png'.format(i)), to8b(gt))
            np.save(os.path.join(dirs['depth'], '{:06d}.npy'.format(i)), depth.detach().cpu().numpy())
            np.save(os.path.join(dirs['disp'], '{:06d}.npy'.format(i)), disp.detach().cpu().numpy())
            np.save(os.path.join(dirs['weight'], '{:06d}.npy'.format(i)), extras['weights'].detach().cpu().numpy())
            np.save(os.path.join(dirs['z'], '{:06d}.npy'.format(i)), extras['z_vals'].detach().cpu().numpy())
            if need_alpha:
                np.save(os.path.join(dirs['alpha'], '{:06d}.npy'.format(i)), extras['alpha'].detach().cpu().numpy())
            pose = torch.as_tensor(render_poses[i])[:3, :4].detach().cpu().numpy()
            np.savetxt(os.path.join(dirs['pose'], '{:06d}.txt'.format(i)),
                       np.concatenate([pose, np.array([[0, 0, 0, 1]])], axis=0))
    disps = torch.stack(disps, 0) if disp_require_grad else np.stack(disps, 0)
    rgbs = torch.stack(rgbs, 0) if rgb_require_grad else np.stack(rgbs, 0)
    return rgbs, disps, (Xs, Ys)


def save_checkpoint(path, global_step, render_kwargs_train, optimizer, module_prefix=True):
    """The reference's checkpoint format (DS_NeRF/run.py:1043-1053).  Its state dicts carry the
    `module.` prefix of nn.DataParallel; written the same way by default so either side can load
    the other's files."""
    def sd(net):
        if net is None:
            return None
        d = net.state_dict()
        return {('module.' + k if module_prefix else k): v for k, v in d.items()}
    torch.save({'global_step': global_step, 'network_fn_state_dict': sd(render_kwargs_train['network_fn']),
                'network_fine_state_dict': sd(render_kwargs_train['network_fine']),
                'optimizer_state_dict': optimizer.state_dict()}, path)


def render_path_4view(iter, all_masks, render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None,
                      render_factor=0, disp_require_grad=False, need_alpha=False, rgb_require_grad=False,
                      detach_weights=False, patch_len=None, masks=None):
    """DS_NeRF/run.py:1365-1401: the <=5 neighbour views [max(0,it-4) : it+5 : 2], it = iter % 60."""
    H, W, focal = hwf
    if render_factor != 0:
        H = H // render_factor
        W = W // render_factor
        focal = focal / render_factor
    neighborhood_size = 4
    iter = iter % 60
    lo, hi = max(0, iter - neighborhood_size), min(len(render_poses), iter + neighborhood_size + 1)
    selected_poses = render_poses[lo:hi:2]
    selected_masks = all_masks[max(0, iter - neighborhood_size):min(len(all_masks), iter + neighborhood_size + 1):2]
    rgbs, disps = [], []
    for c2w in selected_poses:
        rgb, disp, _, _, _ = render(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], retraw=True, need_alpha=need_alpha,
                                    **render_kwargs)
        disps.append(disp)
        rgbs.append(rgb)
    return torch.stack(rgbs, 0), torch.stack(disps, 0), selected_masks


def create_nerf(args, device=None):
    """DS_NeRF/run.py:1474-1599 for the `--no_tcnn`, `alpha_model_path is None` configuration:
    returns (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer).
    The reference wraps both MLPs in nn.DataParallel; here scaling is one process per GPU
    (mvip_nerf_amd.trainer), so the modules are bare and checkpoints are read with or without
    the `module.` key prefix."""
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    if getattr(args, 'alpha_model_path', None) is not None:
        raise NotImplementedError('alpha_model_path (NeRF_RGB) is outside the hot-path scope')
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views = 0
    embeddirs_fn = None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                          skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
        grad_vars += list(model_fine.parameters())

    def network_query_fn(inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                           netchunk=args.netchunk)
    network_query_fn._mvip_native = True

    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start = 0
    basedir, expname = args.basedir, args.expname
    if getattr(args, 'ft_path', None) is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        d = os.path.join(basedir, expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f] if os.path.isdir(d) else []
    if len(ckpts) > 0 and not args.no_reload:
        ckpt = torch.load(ckpts[-1], map_location=device)
        start = ckpt['global_step']
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        model.load_state_dict(_strip_module_prefix(ckpt['network_fn_state_dict']))
        if model_fine is not None:
            model_fine.load_state_dict(_strip_module_prefix(ckpt['network_fine_state_dict']))

    render_kwargs_train = {
        'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
        'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model,
        'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    else:
        render_kwargs_train['ndc'] = True
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    if getattr(args, 'sigma_loss', False):
        raise NotImplementedError('--sigma_loss (DS_NeRF/loss.py) is off in the shipped config and out of scope')
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer


def create_nerf_tcnn(args, device=None):
    """DS_NeRF/run.py:1602-1700: the hash-grid model (`no_tcnn = False`, the shipped config's choice).
    Identity embedders, one NeRF_TCNN per level of the hierarchy, Adam over all parameters; same return
    tuple and checkpoint keys as create_nerf."""
    from .run_nerf_helpers_tcnn import NeRF_TCNN
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    if getattr(args, 'alpha_model_path', None) is not None:
        raise NotImplementedError('alpha_model_path is outside the hot-path scope')
    embed_fn = lambda inp: inp
    embeddirs_fn = (lambda inp: inp) if args.use_viewdirs else None
    model = NeRF_TCNN(encoding="hashgrid", seed=0).to(device)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF_TCNN(encoding="hashgrid", seed=1).to(device)
        grad_vars += list(model_fine.parameters())

    def network_query_fn(inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                           netchunk=args.netchunk)

    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start = 0
    if getattr(args, 'ft_path', None) is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        d = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f] if os.path.isdir(d) else []
    if len(ckpts) > 0 and not args.no_reload:
        ckpt = torch.load(ckpts[-1], map_location=device)
        start = ckpt['global_step']
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        model.load_state_dict(_strip_module_prefix(ckpt['network_fn_state_dict']))
        if model_fine is not None:
            model_fine.load_state_dict(_strip_module_prefix(ckpt['network_fine_state_dict']))
    render_kwargs_train = {
        'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
        'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model,
        'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    else:
        render_kwargs_train['ndc'] = True
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer


def depth2xyz_torch(depth_map, depth_cam_matrix, depth_scale=1.0):
    """DS_NeRF/run.py:1909-1922: tensor(h,w), 3x3 intrinsics -> tensor(h,w,3)."""
    K = depth_cam_matrix
    z = depth_map if depth_scale == 1.0 else depth_map / depth_scale
    return ops.depth2xyz(z, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))


def depth2normal_geo(depth, k=31):
    """DS_NeRF/run.py:1924-1940: tensor(b,3,h,w) of points -> tensor(b,3,h,w) of plane-fit normals."""
    return torch.stack([ops.normal_fit(depth[b], k) for b in range(depth.shape[0])], 0)


def _strip_module_prefix(sd):
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
