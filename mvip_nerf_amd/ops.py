"""Thin tensor-level wrappers over the C ABI (allocation + autograd plumbing only).

Every function of the NeRF render / training path here launches HIP kernels from libmvipnerf.so on the
current torch stream; inputs must be dense fp32 tensors on the GPU and nothing computes on the CPU.
No library contraction runs anywhere on this path: the hash-grid model's small layers take `csrc/skinny_gemm.hip` for the
forward, the data gradient and the weight gradient (`_LinearCM`; `MVIP_SKINNY_LINEAR=0` switches back to `W @ X` for A/B
timing only), the SDS networks the split-precision kernels.  What stock torch GPU ops remain are tensor allocation,
concatenations and a few elementwise glue ops; they are named where they occur.
"""
import weakref

import torch

from . import _lib
from ._lib import ptr, stream, call

_F32 = torch.float32


def _f32c(t):
    """Dense fp32 view/copy of a device tensor."""
    if t.dtype != _F32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------
# rays
# ------------------------------------------------------------------------------------------------

def get_rays(H, W, focal, c2w, patch=None):
    """(rays_o, rays_d), each [h, w, 3]; `patch` = (i, j, len1, len2) crops rows i.., cols j..
    (DS_NeRF/run_nerf_helpers.py:249-260, DS_NeRF/run.py:1174-1177)."""
    c = _f32c(c2w[:3, :4])
    y0, x0, h, w = (0, 0, H, W) if patch is None else [int(v) for v in patch]
    ro = torch.empty((h, w, 3), device=c.device, dtype=_F32)
    rd = torch.empty_like(ro)
    call('mvip_get_rays', ptr(c), int(H), int(W), float(focal), y0, x0, h, w, ptr(ro), ptr(rd), stream())
    return ro, rd


def ray_rows(rays_o, rays_d, near, far, viewdirs_src=None):
    """[B,3] x2 -> [B,11] rows (o, d, near, far, d/|d|)  (DS_NeRF/run.py:1182-1207)."""
    o, d = _f32c(rays_o.reshape(-1, 3)), _f32c(rays_d.reshape(-1, 3))
    v = None if viewdirs_src is None else _f32c(viewdirs_src.reshape(-1, 3))
    rows = torch.empty((o.shape[0], 11), device=o.device, dtype=_F32)
    call('mvip_ray_rows', ptr(o), ptr(d), ptr(v), float(near), float(far), o.shape[0], ptr(rows), stream())
    return rows


def ray_rows_from_pose(c2w, H, W, focal, near, far, sel=None):
    """Rows for the pixels `sel` (int64 flat y*W+x indices; None = whole frame, raster order)."""
    c = _f32c(c2w[:3, :4])
    if sel is not None:
        sel = sel.to(torch.int64).contiguous()
    B = H * W if sel is None else sel.numel()
    rows = torch.empty((B, 11), device=c.device, dtype=_F32)
    call('mvip_ray_rows_from_pose', ptr(c), int(H), int(W), float(focal), float(near), float(far),
         ptr(sel, torch.int64), B, ptr(rows), stream())
    return rows


_T_VALS = {}


def _t_vals(S, device):
    key = (S, device)
    if key not in _T_VALS:
        _T_VALS[key] = torch.linspace(0., 1., steps=S, device=device, dtype=_F32)
    return _T_VALS[key]


def stratified_z(rows, S, lindisp, t_rand=None):
    """[B,S] sample depths (DS_NeRF/run.py:1759-1781); t_rand [B,S] uniforms or None."""
    B = rows.shape[0]
    z = torch.empty((B, S), device=rows.device, dtype=_F32)
    tr = None if t_rand is None else _f32c(t_rand)
    call('mvip_stratified_z', ptr(rows), rows.shape[1], B, int(S), ptr(_t_vals(S, rows.device)),
         int(bool(lindisp)), ptr(tr), ptr(z), stream())
    return z


def posenc(x, L):
    """Embedder.embed: [..,3] -> [.., 3+6L]."""
    xs = _f32c(x.reshape(-1, 3))
    y = torch.empty((xs.shape[0], 3 + 6 * L), device=xs.device, dtype=_F32)
    call('mvip_posenc', ptr(xs), xs.shape[0], int(L), ptr(y), stream())
    return y.reshape(*x.shape[:-1], 3 + 6 * L)


# ------------------------------------------------------------------------------------------------
# fused MLP
# ------------------------------------------------------------------------------------------------

PARAM_ORDER = tuple([f'pts_linears.{i}.{k}' for i in range(8) for k in ('weight', 'bias')]
                    + [f'{n}.{k}' for n in ('views_linears.0', 'feature_linear', 'alpha_linear', 'rgb_linear')
                       for k in ('weight', 'bias')])
PARAM_SHAPES = tuple([(256, 63), (256,)] + [(256, 256), (256,)] * 4 + [(256, 319), (256,)]
                     + [(256, 256), (256,)] * 2 + [(128, 283), (128,), (256, 256), (256,), (1, 256), (1,),
                                                     (3, 128), (3,)])

_PACKED_FLOATS = None


def packed_floats():
    global _PACKED_FLOATS
    if _PACKED_FLOATS is None:
        _PACKED_FLOATS = int(_lib.load().mvip_mlp_packed_floats())
    return _PACKED_FLOATS


def mlp_pack(params):
    """24 parameter tensors (state-dict order) -> a NEW packed image tensor."""
    ps = [_f32c(p.detach()) for p in params]
    for p, shp in zip(ps, PARAM_SHAPES):
        if tuple(p.shape) != shp:
            raise _lib.MvipError(f'fused MLP is built for the 8x256 NeRF; got parameter shape {tuple(p.shape)} '
                                 f'where {shp} is expected')
    packed = torch.empty(packed_floats(), device=ps[0].device, dtype=_F32)
    call('mvip_mlp_pack', _lib.ptr_array(ps), ptr(packed), stream())
    return packed


def mlp_pack16(params, packed_f32):
    """Image for the two-waves-per-SIMD inference kernel (16 points per wave, csrc/mlp_fwd16.hip)."""
    ps = [_f32c(p.detach()) for p in params]
    img = torch.empty(packed_floats(), device=ps[0].device, dtype=_F32)
    call('mvip_mlp_pack16', _lib.ptr_array(ps), ptr(packed_f32), ptr(img), stream())
    return img


def mlp_pack_f16x3(params, packed_f32):
    """Image for the split-precision forward (precision=1); same size as the fp32 image."""
    ps = [_f32c(p.detach()) for p in params]
    img = torch.empty(packed_floats(), device=ps[0].device, dtype=_F32)
    call('mvip_mlp_pack_f16x3', _lib.ptr_array(ps), ptr(img), ptr(packed_f32), stream())
    return img


def mlp_pack_f16x3_w16(params, packed_f32):
    """Image for the two-waves-per-SIMD split-precision forward (csrc/mlp_fwd16_f16x3.hip); same size as the fp32 image."""
    ps = [_f32c(p.detach()) for p in params]
    img = torch.empty(packed_floats(), device=ps[0].device, dtype=_F32)
    call('mvip_mlp_pack_f16x3_w16', _lib.ptr_array(ps), ptr(packed_f32), ptr(img), stream())
    return img


def mlp_unpack_grads(grad_packed, like):
    grads = [torch.empty(shp, device=grad_packed.device, dtype=_F32) for shp in PARAM_SHAPES]
    call('mvip_mlp_unpack_grads', ptr(grad_packed), _lib.ptr_array(grads), 0, stream())
    return grads


_WORKSPACE = {}
import os as _os
# points per backward tile: delta + weight-gradient launches per tile, 20 KB of G/activation workspace per point (5.2 GB at
# 262,144).  Measured in round 5 on configs[2] / configs[3] iterations (one box, ms): 16,384 points 221 / -, 32,768 175,
# 65,536 156, 131,072 146, 262,144 143.0 / 399.9, 524,288 141.9 / 397.9, 1,048,576 140.9 / 397.0 (the weight-gradient launches
# pay a ramp and a 256 x 256 atomic flush per workgroup and tile); the configs[1] iteration measures the same from 262,144 up.
# The default stays at 262,144: as the default, the 21-GB workspace of 1,048,576 ran tests/test_configs_large.py's replicas
# (several processes on one device beside a 116-GB parent) out of memory.  MVIP_BWD_TILE_POINTS=1048576 buys the 1.5 % on a
# device the job owns.
BWD_TILE_POINTS = int(_os.environ.get('MVIP_BWD_TILE_POINTS', 262144))


def _zero_grads(device):
    """24 zeroed gradient tensors carved from one flat allocation (one memset)."""
    flat = torch.zeros(sum(_numel(s) for s in PARAM_SHAPES), device=device, dtype=_F32)
    out, o = [], 0
    for shp in PARAM_SHAPES:
        n = _numel(shp)
        out.append(flat[o:o + n].view(shp))
        o += n
    return out


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def _workspace(device, tile_points):
    key = (device, tile_points)
    if key not in _WORKSPACE:
        n = int(_lib.load().mvip_mlp_backward_workspace_bytes(tile_points))
        _WORKSPACE[key] = torch.empty(max(n, 16) // 4 + 4, device=device, dtype=_F32)
    return _WORKSPACE[key]


# Keep the activations of a training forward for its backward (9.9 KB per point) while they fit the DEVICE: the
# budget (for ALL live stashes together) is computed per call from what is free right now (hipMemGetInfo + the allocator's cached-but-unused blocks),
# so two ranks sharing one device, a resident fp32 SD UNet + VAE, or a smaller-HBM part shrink it by themselves.
# Beyond it -- or if the allocation fails -- the backward recomputes the activations tile by tile.
# MVIP_STASH_BUDGET_BYTES overrides the computed budget (0 = always recompute).
STASH_FREE_FRACTION = 0.6          # of the currently free bytes, after the backward workspace is set aside
_stash_live = {}                   # device index -> bytes of live stashes


_FREE_CACHE = {}                   # device index -> (bytes reserved by torch at query time, free bytes from the driver, time)
# The cached answer also EXPIRES: another process on the device (mvip_nerf_amd/replicas.py with --devices 0,0, any other
# tenant) moves the free memory without touching this process's counters.  MVIP_SHARED_DEVICE=1 (set by replicas.launch
# when several replicas share a device) shortens the lifetime to every query.
FREE_CACHE_SECONDS = 0.0 if _os.environ.get('MVIP_SHARED_DEVICE') == '1' else 0.25


def _device_free_bytes(device):
    """hipMemGetInfo through torch, asked again only when torch's own reservation has changed since the last answer or the
    answer is older than FREE_CACHE_SECONDS: the driver call costs milliseconds once tens of GB are mapped (five stash
    allocations per iteration turned a 54 ms training iteration into 75-94 ms of wall time: tools/train_step_profile.py
    --after-hashgrid), and between two queries the device's free memory moves only when torch itself maps or unmaps memory
    -- which `memory_reserved` (a host-side counter) shows -- or when ANOTHER process does, which only time can cover (the
    allocation itself is guarded as well, see _take_stash)."""
    import time
    key = device.index if device.index is not None else torch.cuda.current_device()
    reserved = torch.cuda.memory_reserved(device)
    now = time.monotonic()
    hit = _FREE_CACHE.get(key)
    if hit is None or hit[0] != reserved or now - hit[2] > FREE_CACHE_SECONDS:
        hit = (reserved, torch.cuda.mem_get_info(device)[0], now)
        _FREE_CACHE[key] = hit
    return hit[1]


def _stash_budget(device):
    env = _os.environ.get('MVIP_STASH_BUDGET_BYTES')
    if env is not None:
        return int(env)
    free = _device_free_bytes(device)
    free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    ws_bytes = 0 if (device, BWD_TILE_POINTS) in _WORKSPACE else int(
        _lib.load().mvip_mlp_backward_workspace_bytes(BWD_TILE_POINTS))
    # all live stashes together may hold STASH_FREE_FRACTION of what is available to them (free now + already held)
    key = device.index if device.index is not None else torch.cuda.current_device()
    live = _stash_live.get(key, 0)
    return int(STASH_FREE_FRACTION * max(free + live - ws_bytes, 0)) - live


def _take_stash(P, device):
    n = int(_lib.load().mvip_mlp_stash_floats(P))
    if 4 * n > _stash_budget(device):
        return None
    try:
        t = torch.empty(n, device=device, dtype=_F32)
    except torch.OutOfMemoryError:
        return None                                  # fragmentation or a concurrent process: recompute instead
    key = device.index if device.index is not None else torch.cuda.current_device()
    _stash_live[key] = _stash_live.get(key, 0) + 4 * n
    weakref.finalize(t, _stash_freed, key, 4 * n)   # also covers graphs that are dropped without backward
    return t


def _stash_freed(key, nbytes):
    _stash_live[key] = _stash_live.get(key, 0) - nbytes


class _MLPRays(torch.autograd.Function):
    """raw[B,S,4] = MLP(enc(o + d z), enc(viewdir)); gradients flow to the 24 parameters only."""

    @staticmethod
    def forward(ctx, rows, z, packed, precision, packed16, *params):
        """`packed` is the weight image of `precision` (0: fp32 image, 1: f16x3 image); `packed16` (optional) the image of
        the two-waves-per-SIMD kernel of that precision (mlp_pack16 / mlp_pack_f16x3_w16), which then runs the stash-writing
        forward; the backward always works from `packed`."""
        B, S = z.shape
        raw = torch.empty((B, S, 4), device=z.device, dtype=_F32)
        stash = _take_stash(B * S, z.device)
        if stash is None:
            call('mvip_mlp_forward_rays', ptr(packed), ptr(rows), ptr(z), B, S, ptr(raw), precision, stream())
        elif packed16 is not None and precision == 0:
            call('mvip_mlp_forward_rays_stash16', ptr(packed16), ptr(rows), ptr(z), B, S, ptr(raw), ptr(stash), stream())
        elif packed16 is not None and precision == 1:       # the two-wave split-precision kernel (its own image; the backward keeps `packed`)
            call('mvip_mlp_forward_rays_stash_f16x3_w16', ptr(packed16), ptr(rows), ptr(z), B, S, ptr(raw), ptr(stash), stream())
        else:
            call('mvip_mlp_forward_rays_stash', ptr(packed), ptr(rows), ptr(z), B, S, ptr(raw), ptr(stash), precision,
                 stream())
        ctx.save_for_backward(rows, z, packed)
        ctx.stash, ctx.precision = stash, precision
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        rows, z, packed = ctx.saved_tensors
        B, S = z.shape
        grads = _zero_grads(z.device)
        ws = _workspace(z.device, BWD_TILE_POINTS)
        stash, ctx.stash = ctx.stash, None
        if stash is None:
            call('mvip_mlp_backward_rays', ptr(packed), ptr(rows), ptr(z), B, S, ptr(_f32c(d_raw)),
                 _lib.ptr_array(grads), ptr(ws), BWD_TILE_POINTS, ctx.precision, stream())
        else:
            call('mvip_mlp_backward_stash', ptr(packed), ptr(stash), B * S, ptr(_f32c(d_raw)),
                 _lib.ptr_array(grads), ptr(ws), BWD_TILE_POINTS, ctx.precision, stream())
            del stash
        return (None, None, None, None, None, *grads)


class _MLPPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, dirs, packed, precision, *params):
        P = pts.shape[0]
        raw = torch.empty((P, 4), device=pts.device, dtype=_F32)
        call('mvip_mlp_forward_points', ptr(packed), ptr(pts), ptr(dirs), P, ptr(raw), precision, stream())
        ctx.save_for_backward(pts, dirs, packed)
        ctx.precision = precision
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        pts, dirs, packed = ctx.saved_tensors
        grads = _zero_grads(pts.device)
        ws = _workspace(pts.device, BWD_TILE_POINTS)
        call('mvip_mlp_backward_points', ptr(packed), ptr(pts), ptr(dirs), pts.shape[0], ptr(_f32c(d_raw)),
             _lib.ptr_array(grads), ptr(ws), BWD_TILE_POINTS, ctx.precision, stream())
        return (None, None, None, None, *grads)


def mlp_rays(rows, z, packed, params, packed_f16x3=None, train_f16x3=None, packed16=None, train16=None, f16x3_w16=None,
             train_f16x3_w16=None):
    """Fused forward from ray rows + depths.  `params` (the 24 tensors) are passed so autograd
    routes the gradients back to them; with no grad needed the Function is skipped.
    `packed_f16x3` selects the split-precision kernel (precision = 1) for no-grad calls,
    `train_f16x3` (the same kind of image) for calls that will be back-propagated, `packed16` the exact-fp32
    two-waves-per-SIMD inference kernel (no-grad calls at precision 0), `train16` the same image for the stash-writing
    training forward at precision 0, `f16x3_w16` (ops.mlp_pack_f16x3_w16) the two-waves-per-SIMD split-precision kernel
    for no-grad calls (takes precedence over `packed_f16x3`), `train_f16x3_w16` the same kind of image for the stash-writing
    training forward at precision 1 (the backward keeps `train_f16x3`)."""
    rows, z = _f32c(rows), _f32c(z)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        if train_f16x3 is not None:
            return _MLPRays.apply(rows, z, train_f16x3, 1, train_f16x3_w16, *params)
        return _MLPRays.apply(rows, z, packed, 0, train16, *params)
    B, S = z.shape
    raw = torch.empty((B, S, 4), device=z.device, dtype=_F32)
    if f16x3_w16 is not None:
        call('mvip_mlp_forward_rays_f16x3_w16', ptr(f16x3_w16), ptr(rows), ptr(z), B, S, ptr(raw), stream())
    elif packed_f16x3 is not None:
        call('mvip_mlp_forward_rays', ptr(packed_f16x3), ptr(rows), ptr(z), B, S, ptr(raw), 1, stream())
    elif packed16 is not None:
        call('mvip_mlp_forward_rays16', ptr(packed16), ptr(rows), ptr(z), B, S, ptr(raw), stream())
    else:
        call('mvip_mlp_forward_rays', ptr(packed), ptr(rows), ptr(z), B, S, ptr(raw), 0, stream())
    return raw


def mlp_points(pts, dirs, packed, params, packed_f16x3=None, train_f16x3=None, packed16=None, f16x3_w16=None):
    pts, dirs = _f32c(pts), _f32c(dirs)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        if train_f16x3 is not None:
            return _MLPPoints.apply(pts, dirs, train_f16x3, 1, *params)
        return _MLPPoints.apply(pts, dirs, packed, 0, *params)
    raw = torch.empty((pts.shape[0], 4), device=pts.device, dtype=_F32)
    if f16x3_w16 is not None:
        call('mvip_mlp_forward_points_f16x3_w16', ptr(f16x3_w16), ptr(pts), ptr(dirs), pts.shape[0], ptr(raw), stream())
    elif packed_f16x3 is not None:
        call('mvip_mlp_forward_points', ptr(packed_f16x3), ptr(pts), ptr(dirs), pts.shape[0], ptr(raw), 1, stream())
    elif packed16 is not None:
        call('mvip_mlp_forward_points16', ptr(packed16), ptr(pts), ptr(dirs), pts.shape[0], ptr(raw), stream())
    else:
        call('mvip_mlp_forward_points', ptr(packed), ptr(pts), ptr(dirs), pts.shape[0], ptr(raw), 0, stream())
    return raw


# ------------------------------------------------------------------------------------------------
# compositing
# ------------------------------------------------------------------------------------------------

COMP_WHITE, COMP_DETACHW = 1, 2


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z, rows, noise, flags, need_alpha):
        B, S = z.shape
        dev = z.device
        rgb = torch.empty((B, 3), device=dev, dtype=_F32)
        disp = torch.empty((B,), device=dev, dtype=_F32)
        acc = torch.empty_like(disp)
        depth = torch.empty_like(disp)
        weights = torch.empty((B, S), device=dev, dtype=_F32)
        alpha = torch.empty((B, S), device=dev, dtype=_F32) if need_alpha else None
        call('mvip_composite_forward', ptr(raw), ptr(z), ptr(rows), rows.shape[1], ptr(noise), B, S, flags,
             ptr(rgb), ptr(disp), ptr(acc), ptr(depth), ptr(weights), ptr(alpha), stream())
        ctx.save_for_backward(raw, z, rows, noise)
        ctx.flags = flags
        ctx.set_materialize_grads(False)
        if need_alpha:
            return rgb, disp, acc, depth, weights, alpha
        return rgb, disp, acc, depth, weights

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_depth, g_w, g_alpha=None):
        raw, z, rows, noise = ctx.saved_tensors
        B, S = z.shape
        d_raw = torch.empty_like(raw)
        c = lambda g: None if g is None else _f32c(g)
        call('mvip_composite_backward', ptr(raw), ptr(z), ptr(rows), rows.shape[1], ptr(noise), B, S, ctx.flags,
             ptr(c(g_rgb)), ptr(c(g_disp)), ptr(c(g_acc)), ptr(c(g_depth)), ptr(c(g_w)), ptr(c(g_alpha)),
             ptr(d_raw), stream())
        return d_raw, None, None, None, None, None


def composite(raw, z, rows, noise=None, white_bkgd=False, detach_weights=False, need_alpha=False):
    """raw2outputs (DS_NeRF/run_nerf_helpers.py:350-404) -> (rgb, disp, acc, weights, depth, alpha|None).
    `rows` carries the ray direction in columns 3..5; `noise` is already scaled by raw_noise_std."""
    flags = (COMP_WHITE if white_bkgd else 0) | (COMP_DETACHW if detach_weights else 0)
    raw = _f32c(raw)
    z = _f32c(z)
    rows = _f32c(rows)
    noise = None if noise is None else _f32c(noise)
    out = _Composite.apply(raw, z, rows, noise, flags, bool(need_alpha))
    rgb, disp, acc, depth, weights = out[:5]
    return rgb, disp, acc, weights, depth, (out[5] if need_alpha else None)


# ------------------------------------------------------------------------------------------------
# hierarchical sampling
# ------------------------------------------------------------------------------------------------

def sample_pdf_merge(z, weights, u, want_inds=False, want_cdf=False):
    """Fused `sample_pdf(mids, weights[:,1:-1]) -> sort(cat[z, samples])` + z_std
    (DS_NeRF/run.py:1809-1816, :1836).  u: [B,Nf] uniforms, or a 1-D [Nf] row shared by all rays.
    Outputs carry no gradient (the reference detaches z_samples, run.py:1812)."""
    z = _f32c(z.detach())
    w = _f32c(weights.detach())
    u = _f32c(u)
    B, Nc = z.shape
    Nf = u.shape[-1]
    dev = z.device
    zs = torch.empty((B, Nf), device=dev, dtype=_F32)
    zm = torch.empty((B, Nc + Nf), device=dev, dtype=_F32)
    zstd = torch.empty((B,), device=dev, dtype=_F32)
    inds = torch.empty((B, Nf), device=dev, dtype=torch.int64) if want_inds else None
    cdf = torch.empty((B, Nc - 1), device=dev, dtype=_F32) if want_cdf else None
    call('mvip_sample_pdf_merge', ptr(z), ptr(w), ptr(u), int(u.dim() == 1), B, Nc, Nf, ptr(zs), ptr(zm),
         ptr(zstd), ptr(inds, torch.int64), ptr(cdf), stream())
    return zs, zm, zstd, inds, cdf


def sample_pdf(bins, weights, u, want_inds=False, want_cdf=False):
    """Standalone sample_pdf on explicit bins [B,Nb] / weights [B,Nb-1] / uniforms."""
    bins = _f32c(bins)
    w = _f32c(weights)
    u = _f32c(u)
    B, Nb = bins.shape
    Nf = u.shape[-1]
    dev = bins.device
    s = torch.empty((B, Nf), device=dev, dtype=_F32)
    inds = torch.empty((B, Nf), device=dev, dtype=torch.int64) if want_inds else None
    cdf = torch.empty((B, Nb), device=dev, dtype=_F32) if want_cdf else None
    call('mvip_sample_pdf', ptr(bins), ptr(w), ptr(u), int(u.dim() == 1), B, Nb, Nf, ptr(s),
         ptr(inds, torch.int64), ptr(cdf), stream())
    return s, inds, cdf


# ------------------------------------------------------------------------------------------------
# depth -> points -> plane-fit normals
# ------------------------------------------------------------------------------------------------

class _Depth2XYZ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, fx, fy, cx, cy):
        H, W = depth.shape
        pts = torch.empty((H, W, 3), device=depth.device, dtype=_F32)
        call('mvip_depth2xyz', ptr(depth), H, W, fx, fy, cx, cy, ptr(pts), stream())
        ctx.k = (H, W, fx, fy, cx, cy)
        return pts

    @staticmethod
    def backward(ctx, g):
        H, W, fx, fy, cx, cy = ctx.k
        d = torch.empty((H, W), device=g.device, dtype=_F32)
        call('mvip_depth2xyz_backward', ptr(_f32c(g)), H, W, fx, fy, cx, cy, ptr(d), stream())
        return d, None, None, None, None


def depth2xyz(depth, fx, fy, cx, cy):
    """depth [H,W] -> [H,W,3] (DS_NeRF/run.py:1909-1922), differentiable in depth."""
    return _Depth2XYZ.apply(_f32c(depth), float(fx), float(fy), float(cx), float(cy))


class _NormalFit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, k):
        _, H, W = points.shape
        dev = points.device
        moments = torch.empty((9, H, W), device=dev, dtype=_F32)
        scratch = torch.empty((9, H, W), device=dev, dtype=_F32)
        normals = torch.empty((3, H, W), device=dev, dtype=_F32)
        call('mvip_normal_fit_forward', ptr(points), H, W, k, ptr(moments), ptr(scratch), ptr(normals), stream())
        ctx.save_for_backward(points, moments, normals)
        ctx.k = k
        return normals

    @staticmethod
    def backward(ctx, g):
        points, moments, normals = ctx.saved_tensors
        _, H, W = points.shape
        scratch = torch.empty((18, H, W), device=g.device, dtype=_F32)
        d = torch.empty_like(points)
        call('mvip_normal_fit_backward', ptr(points), ptr(moments), ptr(normals), ptr(_f32c(g)), H, W, ctx.k,
             ptr(scratch), ptr(d), stream())
        return d, None


def normal_fit(points, k=31):
    """points [3,H,W] planar -> least-squares plane normals [3,H,W] (DS_NeRF/run.py:1924-1940)."""
    return _NormalFit.apply(_f32c(points), int(k))


# GroupNorm (+ SiLU) of the SDS networks -------------------------------------------------------------

_GN_DTYPES = {torch.float32: 0, torch.float16: 1}


class _ResizeBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW):
        xc = _f32c(x)
        N, C, H, W = xc.shape
        y = torch.empty((N, C, OH, OW), device=xc.device, dtype=torch.float32)
        call('mvip_resize_bilinear', ptr(xc), N * C, H, W, OH, OW, ptr(y), stream())
        ctx.shape = (N, C, H, W, OH, OW)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, OH, OW = ctx.shape
        dyc = _f32c(dy)
        dx = torch.empty((N, C, H, W), device=dyc.device, dtype=torch.float32)
        call('mvip_resize_bilinear_backward', ptr(dyc), N * C, H, W, OH, OW, ptr(dx), stream())
        return dx, None, None


def resize_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) for [N, C, H, W] fp32 device tensors (the resize in
    front of vae.encode, DS_NeRF/guidance/sd_utils.py:282-284), differentiable w.r.t. x."""
    return _ResizeBilinear.apply(x, int(size[0]), int(size[1]))


def _gn_workspace(N, C, HW, device):
    nbytes = int(_lib.load().mvip_groupnorm_workspace_bytes(N, C, HW))
    return torch.empty(max(nbytes // 8, 1), device=device, dtype=torch.float64)


class _GroupNorm(torch.autograd.Function):
    """y = act(group_norm(x)) in two HIP launches (csrc/group_norm.hip); dx only: the SDS networks are
    frozen, so a weight or bias that requires grad is an error, not a silent zero."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu):
        if x.dtype not in _GN_DTYPES:
            raise _lib.MvipError(f'group_norm: unsupported dtype {x.dtype}')
        if (weight is not None and weight.requires_grad) or (bias is not None and bias.requires_grad):
            raise NotImplementedError('group_norm: parameter gradients are not implemented (frozen networks only)')
        dt = x.dtype
        xc = x.contiguous()
        N, C = xc.shape[0], xc.shape[1]
        HW = xc.numel() // max(N * C, 1)
        w = None if weight is None else weight.detach().to(dt).contiguous()
        b = None if bias is None else bias.detach().to(dt).contiguous()
        y = torch.empty_like(xc)
        mean = torch.empty((N, groups), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ws = _gn_workspace(N, C, HW, x.device)
        call('mvip_groupnorm_forward', ptr(xc, dt), ptr(w, dt), ptr(b, dt), N, C, HW, int(groups), float(eps),
             int(bool(silu)), _GN_DTYPES[dt], ptr(y, dt), ptr(mean), ptr(rstd), ptr(ws, torch.float64), stream())
        ctx.save_for_backward(xc, w, b, mean, rstd)
        ctx.cfg = (int(groups), int(bool(silu)))
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, w, b, mean, rstd = ctx.saved_tensors
        groups, silu = ctx.cfg
        dt = xc.dtype
        dyc = dy.contiguous().to(dt)
        N, C = xc.shape[0], xc.shape[1]
        HW = xc.numel() // max(N * C, 1)
        dx = torch.empty_like(xc)
        ws = _gn_workspace(N, C, HW, xc.device)
        call('mvip_groupnorm_backward', ptr(xc, dt), ptr(dyc, dt), ptr(w, dt), ptr(b, dt), ptr(mean), ptr(rstd), N, C,
             HW, groups, silu, _GN_DTYPES[dt], ptr(dx, dt), ptr(ws, torch.float64), stream())
        return dx, None, None, None, None, None


def group_norm(x, weight, bias, groups, eps=1e-5, silu=False):
    """act(F.group_norm(x, groups, weight, bias, eps)) for x [N, C, *]; act = SiLU when `silu`."""
    return _GroupNorm.apply(x, weight, bias, groups, eps, silu)


# 3x3 convolution of the SDS networks (split-precision implicit GEMM, csrc/conv3x3.hip) ---------------------

def conv3x3_supported(conv, x, hw=None):
    """True when `conv` (an nn.Conv2d) applied to x [N, Cin, H, W] fp32 can run on the HIP kernel (hw = (H, W) overrides
    x's spatial size: the up-sampled image of Upsample2D, which is never materialised)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    if not (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32):
        return False
    lib = _lib.load()
    H, W = (x.shape[2], x.shape[3]) if hw is None else hw
    ok = bool(lib.mvip_conv3x3_supported(conv.out_channels, conv.in_channels, H, W))
    if ok and x.requires_grad and torch.is_grad_enabled():        # the data gradient runs the transposed operator
        ok = bool(lib.mvip_conv3x3_supported(conv.in_channels, conv.out_channels, H, W))
    return ok


# Arithmetic of the SDS networks' contractions (csrc/conv3x3.hip, attention.hip): 0 = split precision (fp16 hi + lo halves of
# both operands, three products: fp32-grade, the fp32 networks), 1 = the reference's --fp16 mode (DS_NeRF/guidance/
# sd_utils.py:66): one fp16 product, hi halves only.  Set for the duration of a network's forward by `precision(...)`
# (guidance/sd_nets.py); autograd Functions remember it for their backward.
PREC = 0


class precision:
    def __init__(self, prec):
        self.prec = int(prec)

    def __enter__(self):
        global PREC
        self.old, PREC = PREC, self.prec

    def __exit__(self, *exc):
        global PREC
        PREC = self.old


def _prec():
    return int(PREC)


# Two products instead of three for weights that are exact fp16 values (prec = 2 of the C ABI; csrc/conv3x3.hip NP = 2).
# The reference always loads `revision="fp16"` weights and, in its default fp32 mode, casts them UP
# (DS_NeRF/guidance/sd_utils.py:69-74): the lo half of every frozen UNet / VAE weight is zero, the third product
# W_lo . x_hi adds exact zeros, and skipping it (and the lo fragments' fetch) changes no bit of any result.  Decided per
# packed image at PACK time (one word read back from the packer: `_note_two_product`); images of anything else -- weights
# with a non-zero lo half, activations packed as an A operand -- keep the three-product kernels.
TWO_PRODUCT = bool(int(_os.environ.get('MVIP_TWO_PRODUCT', '1')))          # A/B switch


def _note_two_product(packed, fragment_bytes):
    """Ask the library whether the packed WEIGHT image's lo fragments are all zero and remember it on the tensor."""
    import ctypes
    flag = ctypes.c_int(0)
    call('mvip_packed_weights_two_product', ptr(packed, torch.uint8), int(fragment_bytes), ctypes.byref(flag), stream())
    packed._mvip_two_product = bool(flag.value)
    return packed


def _prec_w(packed):
    """`prec` for a contraction whose A operand is the packed weight image `packed`."""
    p = _prec()
    return 2 if (p == 0 and TWO_PRODUCT and getattr(packed, '_mvip_two_product', False)) else p


def conv3x3_pack(weight, transpose=False):
    """Packed split-precision image of a [Cout, Cin, 3, 3] weight (transpose: the data-gradient operator)."""
    Cout, Cin = weight.shape[0], weight.shape[1]
    nbytes = int(_lib.load().mvip_conv3x3_packed_bytes(Cout, Cin))
    packed = torch.empty(nbytes, device=weight.device, dtype=torch.uint8)
    w = weight.detach().contiguous()
    call('mvip_conv3x3_pack', ptr(w), Cout, Cin, int(bool(transpose)), ptr(packed, torch.uint8), stream())
    return _note_two_product(packed, Cout * Cin * 36)


def _conv_packed(conv, transpose):
    """Per-module cache of the packed images (the SDS networks are frozen; re-packed if the weight changes)."""
    cache = conv.__dict__.setdefault('_mvip_packed', {})
    key = (conv.weight.data_ptr(), conv.weight._version)
    if cache.get('key') != key:
        cache.clear()
        cache['key'] = key
    if transpose not in cache:
        cache[transpose] = conv3x3_pack(conv.weight, transpose)
    return cache[transpose]


def _split_buffer(N, C, HW, device):
    return torch.empty(N * C * HW * 2, device=device, dtype=torch.float16)


def _with_saved_prec(backward):
    """The backward of an SDS contraction runs in the arithmetic of its forward (ctx.prec), whatever is current."""
    def wrapped(ctx, *grads):
        with precision(getattr(ctx, 'prec', 0)):
            return backward(ctx, *grads)
    return wrapped


def _gn_stats(xc, N, C, H, W, G, eps, ws):
    """mean, rstd [N, G] of xc by a pass over it (fp64 moments, csrc/group_norm.hip)."""
    mean = torch.empty((N, G), device=xc.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    call('mvip_groupnorm_stats', ptr(xc), N, C, H * W, G, float(eps), 0, ptr(mean), ptr(rstd), ptr(ws, torch.float64), stream())
    return mean, rstd


class ShortcutLink:
    """Joins the two convolutions of a ResNet block whose shortcut is the identity: the gradient arriving over the shortcut
    (dy of the block) is added to the gradient through the block inside the GroupNorm backward of the FIRST convolution
    (`dx_add` of mvip_groupnorm_backward_fused) instead of by autograd in a pass of its own.  The second convolution's
    backward leaves dy here and reports no gradient for its `residual` input; the first one's picks it up."""
    __slots__ = ('dy',)

    def __init__(self):
        self.dy = None


# The last gradient written by mvip_groupnorm_backward_fused with its per-workgroup maxima: (dx, version, maxima).  The
# backward that receives exactly this tensor as its dy takes the scale from the maxima instead of re-reading it.  The
# strong reference keeps the memory from being reused under the same address; one entry, dropped at the next lookup.
_LAST_DX = [None]


def _scale_of_gradient(dyc):
    """scale2 of dyc: from the maxima its producer left (same power of two) or by the absmax pass."""
    dev = dyc.device
    scale2 = torch.empty(4, device=dev, dtype=torch.float32)
    last, _LAST_DX[0] = _LAST_DX[0], None
    if (last is not None and last[0].data_ptr() == dyc.data_ptr() and last[0].numel() == dyc.numel()
            and last[0]._version == last[1] and dyc._version == last[1] and dyc.dtype == torch.float32):
        call('mvip_absmax_scale_from_maxima', ptr(last[2]), last[2].numel(), ptr(scale2), stream())
    else:
        call('mvip_absmax_scale', ptr(dyc), dyc.numel(), ptr(scale2), ptr(_zero_words(dev)[32:34], torch.int32), stream())
    return scale2


class _NormActConv3x3(torch.autograd.Function):
    """conv3x3(act(group_norm(x))) + bias [+ chan_add[:, :, None, None]] [+ residual]: statistics, split-plane
    writer (normalise + SiLU fused) and the MFMA convolution.  Backward: dY -> split planes (scaled by a power of
    two from its absolute maximum) -> the same kernel with the transposed weight image -> GroupNorm backward.
    Parameter gradients are not produced (frozen networks)."""

    @staticmethod
    def forward(ctx, x, chan_add, residual, norm, conv, silu, link=None):
        ctx.prec = _prec()
        for p in (norm.weight, norm.bias, conv.weight, conv.bias):
            if p is not None and p.requires_grad:
                raise NotImplementedError('conv3x3: parameter gradients are not implemented (frozen networks only)')
        xc = x.contiguous()
        N, C, H, W = xc.shape
        HW, G, Cout = H * W, norm.num_groups, conv.out_channels
        dev = xc.device
        ws = _gn_workspace(N, C, HW, dev)
        gw = None if norm.weight is None else norm.weight.detach().contiguous()
        gb = None if norm.bias is None else norm.bias.detach().contiguous()
        xs = _split_buffer(N, C, HW, dev)
        keep_stats = ctx.needs_input_grad[0] or C // G < 4
        if keep_stats and C // G >= 4 and STATS_FROM_PLANE_WRITER:
            # the backward needs mean / rstd: the plane writer reduces the moment partials (as below) AND writes them out --
            # the same bits gn_finalize would write, without its launch
            rm = _row_moments_of(xc)
            if rm is None:
                call('mvip_groupnorm_stats', ptr(xc), N, C, HW, G, float(norm.eps), 0, None, None, ptr(ws, torch.float64), stream())
                rm = ws
            mean = torch.empty((N, G), device=dev, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            call('mvip_groupnorm_split_planes_moments_out', ptr(xc), ptr(gw), ptr(gb), ptr(rm, torch.float64), float(norm.eps),
                 N, C, HW, G, int(bool(silu)), ptr(xs, torch.float16), ptr(mean), ptr(rstd), _prec(), stream())
            ctx.save_for_backward(xc, gw, gb, mean, rstd)
        elif keep_stats:
            mean, rstd = _gn_stats(xc, N, C, H, W, G, norm.eps, ws)
            call('mvip_groupnorm_split_planes', ptr(xc), ptr(gw), ptr(gb), ptr(mean), ptr(rstd), N, C, HW, G,
                 int(bool(silu)), ptr(xs, torch.float16), _prec(), stream())
            ctx.save_for_backward(xc, gw, gb, mean, rstd)
        else:       # nobody needs mean / rstd afterwards: the plane writer reduces the moment partials itself (one launch less)
            rm = _row_moments_of(xc)                  # ... and the producing convolution may have left them already
            if rm is None:
                call('mvip_groupnorm_stats', ptr(xc), N, C, HW, G, float(norm.eps), 0, None, None, ptr(ws, torch.float64), stream())
                rm = ws
            call('mvip_groupnorm_split_planes_moments', ptr(xc), ptr(gw), ptr(gb), ptr(rm, torch.float64), float(norm.eps),
                 N, C, HW, G, int(bool(silu)), ptr(xs, torch.float16), _prec(), stream())
        y = torch.empty((N, Cout, H, W), device=dev, dtype=torch.float32)
        bias = None if conv.bias is None else conv.bias.detach().contiguous()
        ca = None if chan_add is None else chan_add.detach().contiguous()
        rs = None if residual is None else residual.detach().contiguous()
        _conv3x3_launch(xs, _conv_packed(conv, False), bias, ca, rs, None, N, C, Cout, H, W, y, moments=True)
        ctx.mods = (norm, conv, bool(silu))
        ctx.link = link
        ctx.link_tail = link is not None and residual is not None      # the block's second convolution
        return y

    @staticmethod
    @_with_saved_prec
    def backward(ctx, dy):
        norm, conv, silu = ctx.mods
        dyc = dy.contiguous().float()
        dx = None
        link = ctx.link
        if ctx.needs_input_grad[0]:
            xc, gw, gb, mean, rstd = ctx.saved_tensors
            N, C, H, W = xc.shape
            HW, Cout, dev = H * W, conv.out_channels, xc.device
            if not _lib.load().mvip_conv3x3_supported(C, Cout, H, W):
                raise NotImplementedError(f'conv3x3 data gradient: unsupported shape Cin={C} Cout={Cout}')
            scale2 = _scale_of_gradient(dyc)
            dys = _split_buffer(N, Cout, HW, dev)
            call('mvip_split_planes', ptr(dyc), N, Cout, HW, ptr(scale2), ptr(dys, torch.float16), _prec(), stream())
            dact = torch.empty_like(xc)
            _conv3x3_launch(dys, _conv_packed(conv, True), None, None, None, scale2, N, Cout, C, H, W, dact)
            del dys
            dx = torch.empty_like(xc)
            ws = _gn_workspace(N, C, HW, dev)
            add = None
            if link is not None and not ctx.link_tail:           # first convolution of a linked block: + the shortcut's gradient
                add, link.dy = link.dy, None
            maxima = torch.empty(int(_lib.load().mvip_groupnorm_backward_maxima(N, C, HW)), device=dev, dtype=torch.float32)
            call('mvip_groupnorm_backward_fused', ptr(xc), ptr(dact), ptr(gw), ptr(gb), ptr(mean), ptr(rstd), N, C, HW,
                 norm.num_groups, int(silu), 0, ptr(add), ptr(dx), ptr(maxima), ptr(ws, torch.float64), stream())
            _LAST_DX[0] = (dx, dx._version, maxima)
        d_ca = dyc.sum((2, 3)) if ctx.needs_input_grad[1] else None
        d_rs = dyc if ctx.needs_input_grad[2] else None
        if ctx.link_tail and d_rs is not None:
            link.dy, d_rs = dyc, None                            # handed to the first convolution's GroupNorm backward
        return dx, d_ca, d_rs, None, None, None, None


# The last convolution output whose channel-split reduction also left its GroupNorm row moments: (y, version, moments).
# A forward-only GroupNorm that is handed exactly this tensor skips its pass over y (mvip_groupnorm_split_planes_moments
# reads the moments directly).  One entry; the strong reference keeps the address from being reused.
_LAST_Y = [None]
ROW_MOMENTS = True             # A/B switch: False = every GroupNorm computes its moments from its input
TILE_MOMENTS = _os.environ.get('MVIP_TILE_MOMENTS', '1') != '0'   # A/B switch: unsplit convolutions leave moment partials too (round 6)
STATS_FROM_PLANE_WRITER = True # A/B switch: False = mean / rstd of a forward with grad come from gn_finalize (one launch more)


def _row_moments_of(xc):
    last, _LAST_Y[0] = _LAST_Y[0], None
    if (last is not None and last[0].data_ptr() == xc.data_ptr() and last[0].shape == xc.shape and xc.dtype == torch.float32
            and xc._version == last[1] and last[0]._version == last[1]):
        return last[2]
    return None


def _conv3x3_launch(xs, packed, bias, chan_add, residual, scale2, N, Cin, Cout, H, W, y, moments=False):
    """mvip_conv3x3_f16x3_ws with the split-K workspace the library asks for this shape (none for most).  moments: a
    channel-split launch also leaves y's GroupNorm row moments for the next layer (registered in _LAST_Y)."""
    nbytes = int(_lib.load().mvip_conv3x3_workspace_bytes(N, Cin, Cout, H, W))
    ws = torch.empty(nbytes // 4, device=y.device, dtype=torch.float32) if nbytes else None
    if moments and ROW_MOMENTS and nbytes:
        nd = int(_lib.load().mvip_conv3x3_row_moments_doubles(N, Cin, Cout, H, W))
        if nd:
            rm = torch.empty(nd, device=y.device, dtype=torch.float64)
            call('mvip_conv3x3_f16x3_ws_moments', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(chan_add),
                 ptr(residual), ptr(scale2), N, Cin, Cout, H, W, ptr(y), ptr(ws), ptr(rm, torch.float64), _prec_w(packed), stream())
            _LAST_Y[0] = (y, y._version, rm)
            return
    if moments and ROW_MOMENTS and TILE_MOMENTS and not nbytes:
        # an unsplit launch: moment partials per (pixel tile, wave) from the epilogue + a small reduction (round 6), instead of
        # the next GroupNorm's pass over y
        sb = int(_lib.load().mvip_conv3x3_tile_moments_scratch_bytes(N, Cin, Cout, H, W))
        if sb:
            tp = torch.empty(sb // 4, device=y.device, dtype=torch.float32)
            rm = torch.empty(int(_lib.load().mvip_groupnorm_workspace_bytes(N, Cout, H * W)) // 8, device=y.device, dtype=torch.float64)
            call('mvip_conv3x3_f16x3_tile_moments', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(chan_add),
                 ptr(residual), ptr(scale2), N, Cin, Cout, H, W, ptr(y), ptr(tp), ptr(rm, torch.float64), _prec_w(packed), stream())
            _LAST_Y[0] = (y, y._version, rm)
            return
    call('mvip_conv3x3_f16x3_ws', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(chan_add),
         ptr(residual), ptr(scale2), N, Cin, Cout, H, W, ptr(y), ptr(ws), _prec_w(packed), stream())


def conv3x3_plain(x, conv, upsample2=False):
    """conv(x) (+ bias) for a 3x3 / stride 1 / padding 1 convolution on the split-precision MFMA kernel, without a
    preceding GroupNorm and without autograd: the UNet's up-sampling convolutions (the UNet runs under no_grad).  The
    input is scaled by a power of two from its absmax before the fp16 hi/lo split, as the data gradient is.
    upsample2: the convolution runs on the 2x nearest-neighbour up-sampled image, which is never materialised (the plane
    writer reads x[oy / 2][ox / 2]; the absolute maximum is that of x itself)."""
    xc = _f32c(x.detach())
    N, C, H, W = xc.shape
    Cout, dev = conv.out_channels, xc.device
    scale2 = unit_scale(dev) if forward_unit_scale() else absmax_scale(xc)
    if upsample2:
        xs = _split_buffer(N, C, 4 * H * W, dev)
        call('mvip_split_planes_upsample2', ptr(xc), N, C, H, W, ptr(scale2), ptr(xs, torch.float16), _prec(), stream())
        H, W = 2 * H, 2 * W
    else:
        xs = _split_buffer(N, C, H * W, dev)
        call('mvip_split_planes', ptr(xc), N, C, H * W, ptr(scale2), ptr(xs, torch.float16), _prec(), stream())
    y = torch.empty((N, Cout, H, W), device=dev, dtype=torch.float32)
    bias = None if conv.bias is None else _f32c(conv.bias.detach())
    _conv3x3_launch(xs, _conv_packed(conv, False), bias, None, None, scale2, N, C, Cout, H, W, y)
    return y


def norm_act_conv3x3(x, norm, conv, silu=True, chan_add=None, residual=None, link=None):
    """conv(act(norm(x))) [+ chan_add[:, :, None, None]] [+ residual] on the HIP kernels; the caller checks
    conv3x3_supported first.  link: a ShortcutLink shared by the two convolutions of an identity-shortcut block."""
    return _NormActConv3x3.apply(x, chan_add, residual, norm, conv, silu, link)


# Hash-grid model (NeRF_TCNN) -----------------------------------------------------------------------------

class _HashGrid(torch.autograd.Function):
    """Multiresolution hash-grid features [32, P] (level-major) of points x [P, 3]; gradient w.r.t. the table
    by fp32 atomics (positions never need a gradient on this path).  half2=True: the scattered fine-level
    contributions go out as half-precision pair atomics (tiny-cuda-nn's arithmetic; half the atomic count)."""

    @staticmethod
    def forward(ctx, x, table, levels, bound, half2):
        xc = _f32c(x.detach())
        tc = table.detach().contiguous()
        P = xc.shape[0]
        out = torch.empty((32, P), device=xc.device, dtype=torch.float32)
        call('mvip_hashgrid_forward', ptr(xc), ptr(tc), ptr(levels, torch.int32), P, float(bound), ptr(out), stream())
        ctx.save_for_backward(xc, levels)
        ctx.meta = (float(bound), tc.numel(), bool(half2))
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, levels = ctx.saved_tensors
        bound, n, half2 = ctx.meta
        d = dout.contiguous().float()
        dtable = torch.zeros(n, device=xc.device, dtype=torch.float32)
        if half2:
            scale2 = absmax_scale(d)
            dtable_h = torch.zeros(n, device=xc.device, dtype=torch.float16)
            call('mvip_hashgrid_backward_half2', ptr(xc), ptr(d), ptr(levels, torch.int32), xc.shape[0], bound,
                 ptr(scale2), ptr(dtable), ptr(dtable_h, torch.float16), stream())
            dtable += dtable_h.float() * (16.0 * scale2[1])
        else:
            call('mvip_hashgrid_backward', ptr(xc), ptr(d), ptr(levels, torch.int32), xc.shape[0], bound, ptr(dtable),
                 stream())
        return None, dtable, None, None, None


def hashgrid_encode(x, table, levels, bound=0.0, half2_atomics=False):
    """x [P,3] (raw coordinates in [-bound, bound], or already in [0,1] when bound == 0), table flat
    [n_entries*2], levels [16,4] int32 -> [32, P]."""
    return _HashGrid.apply(x, table, levels, bound, half2_atomics)


def sh4(dirs):
    """Degree-4 spherical harmonics [16, P] of unit directions [P, 3] (NeRF_TCNN's input convention)."""
    dc = _f32c(dirs.detach())
    out = torch.empty((16, dc.shape[0]), device=dc.device, dtype=torch.float32)
    call('mvip_sh4', ptr(dc), dc.shape[0], ptr(out), stream())
    return out


def hashgrid_mlp_pack(sigma_params, colour_params):
    """Operand image of NeRF_TCNN's five bias-free matrices for `hashgrid_nerf_forward`."""
    sp, cp = _f32c(sigma_params.detach()), _f32c(colour_params.detach())
    assert sp.numel() == 3072 and cp.numel() == 7168
    img = torch.empty(int(_lib.load().mvip_hashgrid_mlp_packed_floats()), device=sp.device, dtype=torch.float32)
    call('mvip_hashgrid_mlp_pack', ptr(sp), ptr(cp), ptr(img), stream())
    return img


def hashgrid_nerf_forward(x, dirs, table, levels, img, bound):
    """Fused no-grad NeRF_TCNN.forward: x [P,3], dirs [P,3] -> [P,4] = (colour, sigma)."""
    xc, dc = _f32c(x.detach()), _f32c(dirs.detach())
    P = xc.shape[0]
    out = torch.empty((P, 4), device=xc.device, dtype=torch.float32)
    call('mvip_hashgrid_nerf_forward', ptr(xc), ptr(dc), ptr(table.detach().contiguous()), ptr(levels, torch.int32),
         ptr(img), P, float(bound), ptr(out), stream())
    return out


def skinny_wgrad(dY, X):
    """dW [M, N] = dY [M, P] @ X [N, P]^T for the hash-grid model's small layers (M, N <= 64, P % 64 == 0)."""
    dYc, Xc = _f32c(dY), _f32c(X)
    M, P = dYc.shape
    N = Xc.shape[0]
    slabs = torch.empty((int(_lib.load().mvip_skinny_wgrad_slabs(P)), M, N), device=dYc.device, dtype=torch.float32)
    call('mvip_skinny_wgrad', ptr(dYc), ptr(Xc), M, N, P, ptr(slabs), stream())
    return slabs.sum(0)


# Forward / data gradient of the hash-grid model's small layers run on skinny_fwd_kernel / skinny_fwd16_kernel (default since
# round 4; MVIP_SKINNY_LINEAR=0 puts torch matmuls back for these two products as the A/B alternative; the weight gradient
# is skinny_wgrad_kernel either way).  Measured on the training iteration
# (tools/hashgrid_train_profile.py, same box, alternating) in round 3 with 32-row tiles only: 15.6 ms with the kernel, 15.2 ms
# with the library, whose 16-row MFMA tiles did half the matrix work on the 16-row layers -- those layers now have a 16-row
# kernel of their own.
SKINNY_LINEAR = bool(int(_os.environ.get('MVIP_SKINNY_LINEAR', '1')))


def _skinny_ok(W, X):
    return (SKINNY_LINEAR and X.is_cuda and X.dtype == torch.float32 and W.dtype == torch.float32 and W.shape[0] <= 64 and W.shape[1] <= 64
            and X.shape[1] > 0)


def _pad_points(X, multiple):
    """X [C, P] with P rounded up to `multiple` by zero columns (ragged last chunks only: the kernels read 16-byte quads)."""
    pad = (-X.shape[1]) % multiple
    return X if pad == 0 else torch.nn.functional.pad(X, (0, pad))


def skinny_linear(W, X, relu=False, transpose=False):
    """act(W X) (or act(W^T X) with transpose) for the hash-grid model's small layers: X [N, P] channel-major,
    csrc/skinny_gemm.hip::skinny_fwd_kernel / skinny_fwd16_kernel (exact fp32, one streaming pass)."""
    Wc = _f32c(W)
    P0 = X.shape[1]
    Xc = _f32c(_pad_points(X, 4))
    M, N = (Wc.shape[1], Wc.shape[0]) if transpose else (Wc.shape[0], Wc.shape[1])
    P = Xc.shape[1]
    Y = torch.empty((M, P), device=Xc.device, dtype=torch.float32)
    sm, sn = (1, Wc.shape[1]) if transpose else (Wc.shape[1], 1)
    call('mvip_skinny_linear', ptr(Wc), sm, sn, ptr(Xc), M, N, P, int(bool(relu)), ptr(Y), stream())
    return Y if P == P0 else Y[:, :P0].contiguous()


class _LinearCM(torch.autograd.Function):
    """Y [M, P] = act(W [M, N] @ X [N, P]) (channel-major activations, act = ReLU or identity).  Forward and data gradient:
    csrc/skinny_gemm.hip::skinny_fwd_kernel (16-row variant for M <= 16); the weight gradient -- a [M, N] result contracted
    over millions of points, which a BLAS library runs on a handful of workgroups -- is skinny_wgrad_kernel.  Library
    matmuls only for layers wider than 64 (none in NeRF_TCNN) or with MVIP_SKINNY_LINEAR=0."""

    @staticmethod
    def forward(ctx, W, X, relu):
        if _skinny_ok(W, X):
            Y = skinny_linear(W, X, relu)
        else:
            Y = W @ X
            if relu:
                Y = torch.relu(Y)
        ctx.save_for_backward(W, X, Y if relu else None)
        ctx.relu = relu
        return Y

    @staticmethod
    def backward(ctx, dY):
        W, X, Y = ctx.saved_tensors
        dZ = dY * (Y > 0) if ctx.relu else dY
        dW = dX = None
        if ctx.needs_input_grad[1]:
            dX = skinny_linear(W, dZ, False, transpose=True) if _skinny_ok(W.t(), dZ) else W.t() @ dZ
        if ctx.needs_input_grad[0]:
            if X.is_cuda and X.dtype == torch.float32 and W.shape[0] <= 64 and W.shape[1] <= 64 and X.shape[1] > 0:
                dW = skinny_wgrad(_pad_points(dZ, 64), _pad_points(X, 64))      # zero columns add nothing
            else:
                dW = dZ @ X.t()
        return dW, dX, None


def linear_cm(W, X, relu=False):
    return _LinearCM.apply(W, X, bool(relu))


# Split-precision GEMM building blocks and the VAE mid-block attention built from them ------------------------

ABSMAX_THREE_LAUNCHES = bool(int(_os.environ.get('MVIP_ABSMAX_THREE_LAUNCHES', '0')))     # A/B switch: zero + reduce + scale


def absmax_scale(t):
    """Device-side {s, 1/s, scratch, scratch}: power of two with |t|max * s in [2^9, 2^10)."""
    tc = t.contiguous()
    scale2 = torch.empty(4, device=t.device, dtype=torch.float32)
    zw = None if ABSMAX_THREE_LAUNCHES else _zero_words(t.device)[32:34]
    call('mvip_absmax_scale', ptr(tc), tc.numel(), ptr(scale2), ptr(zw, torch.int32), stream())
    return scale2


def gemm_pack_a(src, M, K, sm, sk, weights=False):
    """A[m][k] = src.flatten()[m*sm + k*sk] (src a dense block of M*K floats) -> packed split-precision image.
    weights=True: a frozen layer's weights, packed once -- the packer's "lo fragments are zero" word is read back (a host
    synchronisation, which is why activations packed per step never ask) so that its contractions can run two products."""
    s = src.contiguous()
    assert s.numel() == M * K
    packed = torch.empty(int(_lib.load().mvip_gemm_packed_bytes(M, K)), device=s.device, dtype=torch.uint8)
    call('mvip_gemm_pack_a', ptr(s), M, K, sm, sk, ptr(packed, torch.uint8), stream())
    return _note_two_product(packed, M * K * 4) if weights else packed


def split_planes_strided(x, N, K, P, sn, sc, sp, scale2=None):
    """X[n][k][p] = x.flatten()[n*sn + k*sc + p*sp] (* scale2[0]) -> fp16 hi/lo split planes."""
    xc = x.contiguous()
    xs = _split_buffer(N, K, P, xc.device)
    call('mvip_split_planes_strided', ptr(xc), N, K, P, sn, sc, sp, ptr(scale2), ptr(xs, torch.float16), _prec(), stream())
    return xs


GEMM_CFG = 0       # 0: tile chosen by shape; 1..4 force a workgroup tile (A/B timing switch, see include/mvip_nerf.h)


LN_STATS_FROM_GEMM = True      # A/B switch: False = every LayerNorm computes its statistics from its input (one launch more)


def gemm_f16x3(xs, packed, N, K, M, P, bias=None, chan_add=None, residual=None, x_scale2=None, out=None, ln_stats=False):
    """Y[n][m][p] = sum_k A[m][k] X[n][k][p] (+ bias[m] + chan_add[n][m] + residual[n][m][p]), fp32 [N, M, P].
    out: a contiguous fp32 tensor of N * M * P elements to write instead of a new one (e.g. a slice of a larger result).
    ln_stats: returns (Y, stats) -- stats = (fp64 partial moments of Y over its rows, segments) for `layernorm_split(...,
    stats=)` when this shape's launch can leave them in its epilogue (round 6), else None."""
    if out is not None:
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == N * M * P
        y = out
    else:
        y = torch.empty((N, M, P), device=xs.device, dtype=torch.float32)
    if GEMM_CFG:
        call('mvip_gemm_f16x3_cfg', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(chan_add),
             ptr(residual), ptr(x_scale2), N, K, M, P, ptr(y), int(GEMM_CFG), _prec_w(packed), stream())
        return (y, None) if ln_stats else y
    nbytes = int(_lib.load().mvip_gemm_workspace_bytes(N, K, M, P))            # split-K partial sums (few shapes)
    ws = torch.empty(nbytes // 4, device=y.device, dtype=torch.float32) if nbytes else None
    if ln_stats:
        S = int(_lib.load().mvip_gemm_ln_segments(N, K, M, P, _prec_w(packed))) if (LN_STATS_FROM_GEMM and not GEMM_CFG) else 0
        if S:
            part = torch.empty(N * S * 2 * P, device=y.device, dtype=torch.float64)
            call('mvip_gemm_f16x3_ws_ln', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(chan_add), ptr(residual),
                 ptr(x_scale2), N, K, M, P, ptr(y), ptr(ws), ptr(part, torch.float64), _prec_w(packed), stream())
            return y, (part, S)
    call('mvip_gemm_f16x3_ws', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(chan_add), ptr(residual),
         ptr(x_scale2), N, K, M, P, ptr(y), ptr(ws), _prec_w(packed), stream())
    return (y, None) if ln_stats else y


# Opt-in: split the forward ACTIVATIONS of the UNet (evaluated under no_grad) at a fixed scale of 1 instead of a
# measured power of two.  They live well inside fp16's range (the reference's own --fp16 mode runs the whole UNet in
# fp16, DS_NeRF/guidance/sd_utils.py:66), and every absmax pass disappears from the step -- but the result is then
# fp32-grade only relative to a tensor scale of O(0.1 .. 1e4): for a tensor of magnitude 1e-4 the fp16 lo terms are
# subnormal and the error grows to fp16-grade (tests/test_sds.py::test_plain_conv3x3_on_mfma_kernel shows it).  The
# default keeps the measured scale, which is magnitude-invariant.
FORWARD_UNIT_SCALE = False         # opt-in (34.5 vs 35.x ms per SDS step): see the comment above for what it gives up
_UNIT = {}


_PROB = {}


def prob_scale(device):
    """scale2 = {2^9, 2^-9} for softmax probabilities (values in [0, 1]: the bound IS the scale, no pass over the [L, L] matrix
    for its maximum; a flat row's 1/L = 2^-12 still splits into normal fp16 hi + lo terms at this scale)."""
    if device not in _PROB:
        _PROB[device] = torch.tensor([512.0, 1.0 / 512.0, 0.0, 0.0], device=device, dtype=_F32)
    return _PROB[device]


def unit_scale(device):
    if device not in _UNIT:
        _UNIT[device] = torch.tensor([1.0, 1.0, 0.0, 0.0], device=device, dtype=_F32)
    return _UNIT[device]


def forward_unit_scale():
    """True when FORWARD activations are split at scale 1: the opt-in above, and always in the reference's --fp16 mode
    (PREC = 1) -- there the single hi plane fp16(x) IS the fp16 tensor the reference's half-precision networks hold
    (DS_NeRF/guidance/sd_utils.py:66), so a measured scale would buy nothing the mode promises.  Gradients keep their
    measured scale in both modes (they are ~1e-5 and would underflow)."""
    return FORWARD_UNIT_SCALE or _prec() == 1


def _scaled_planes(x, N, K, P, sn, sc, sp, forward_activation=False):
    s2 = unit_scale(x.device) if (forward_activation and forward_unit_scale()) else absmax_scale(x)
    return split_planes_strided(x, N, K, P, sn, sc, sp, s2), s2


def vae_attention_supported(x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 32 == 0
            and (x.shape[2] * x.shape[3]) % 256 == 0)


def _attn_weights(mod):
    """Cached packed images of the attention block's four linears (frozen): [Wq;Wk;Wv] and Wo with their
    transposes for the backward."""
    ws = (mod.to_q.weight, mod.to_k.weight, mod.to_v.weight, mod.to_out[0].weight)
    key = tuple((w.data_ptr(), w._version) for w in ws)
    cache = mod.__dict__.setdefault('_mvip_packed', {})
    if cache.get('key') != key:
        C = ws[0].shape[0]
        wqkv = torch.cat([w.detach() for w in ws[:3]], 0).contiguous()          # [3C, C]
        wo = ws[3].detach().contiguous()
        cache.clear()
        cache.update(key=key,
                     qkv=gemm_pack_a(wqkv, 3 * C, C, C, 1, weights=True), qkv_t=gemm_pack_a(wqkv, C, 3 * C, 1, C, weights=True),
                     o=gemm_pack_a(wo, C, C, C, 1, weights=True), o_t=gemm_pack_a(wo, C, C, 1, C, weights=True),
                     bqkv=torch.cat([mod.to_q.bias.detach(), mod.to_k.bias.detach(), mod.to_v.bias.detach()]).contiguous(),
                     bo=mod.to_out[0].bias.detach().contiguous())
    return cache


class _VAEAttention(torch.autograd.Function):
    """x + proj(softmax(q^T k / sqrt(C)) v) of the VAE mid block (single head over H*W tokens, channel-first),
    every product on the split-precision MFMA GEMM; the [L, L] score matrix is materialised (67 MB at 64x64).
    Backward: the six transposed products + GroupNorm backward.  Parameter gradients are not produced."""

    @staticmethod
    def forward(ctx, x, mod):
        ctx.prec = _prec()
        for p in mod.parameters():
            if p.requires_grad:
                raise NotImplementedError('vae_attention: parameter gradients are not implemented (frozen networks only)')
        xc = x.contiguous()
        N, C, H, W = xc.shape
        L, dev, norm = H * W, xc.device, mod.group_norm
        G = norm.num_groups
        wts = _attn_weights(mod)
        ws = _gn_workspace(N, C, L, dev)
        gw, gb = norm.weight.detach().contiguous(), norm.bias.detach().contiguous()
        mean, rstd = _gn_stats(xc, N, C, H, W, G, norm.eps, ws)
        hs = _split_buffer(N, C, L, dev)
        call('mvip_groupnorm_split_planes', ptr(xc), ptr(gw), ptr(gb), ptr(mean), ptr(rstd), N, C, L, G, 0,
             ptr(hs, torch.float16), _prec(), stream())
        qkv = gemm_f16x3(hs, wts['qkv'], N, C, 3 * C, L, bias=wts['bqkv'])           # [N, 3C, L]
        del hs
        probs, O = [], torch.empty((N, C, L), device=dev, dtype=torch.float32)
        for n in range(N):
            q, k, v = qkv[n, :C], qkv[n, C:2 * C], qkv[n, 2 * C:]
            ks, s2 = _scaled_planes(k, 1, C, L, 0, L, 1, forward_activation=_prec() == 1)
            S = gemm_f16x3(ks, gemm_pack_a(q, L, C, 1, L), 1, C, L, L, x_scale2=s2)[0]        # S[i][j] = q_i . k_j
            Pm = softmax_rows(S, C ** -0.5)
            del S
            s2 = unit_scale(dev) if _prec() == 1 else prob_scale(dev)
            pts = split_planes_strided(Pm, 1, L, L, 0, 1, L, s2)                              # X[k=j][p=i] = P[i][j]
            gemm_f16x3(pts, gemm_pack_a(v, C, L, L, 1), 1, L, C, L, x_scale2=s2, out=O[n])
            probs.append(Pm)
        os_, s2 = _scaled_planes(O, N, C, L, C * L, L, 1, forward_activation=_prec() == 1)
        out = gemm_f16x3(os_, wts['o'], N, C, C, L, bias=wts['bo'], residual=xc.reshape(N, C, L), x_scale2=s2)
        ctx.save_for_backward(xc, gw, gb, mean, rstd, qkv, *probs)
        ctx.mod = mod
        return out.reshape(N, C, H, W)

    @staticmethod
    @_with_saved_prec
    def backward(ctx, dout):
        xc, gw, gb, mean, rstd, qkv, *probs = ctx.saved_tensors
        mod = ctx.mod
        N, C, H, W = xc.shape
        L, dev, norm = H * W, xc.device, mod.group_norm
        wts = _attn_weights(mod)
        d = dout.contiguous().float().reshape(N, C, L)
        ds, s2 = _scaled_planes(d, N, C, L, C * L, L, 1)
        dO = gemm_f16x3(ds, wts['o_t'], N, C, C, L, x_scale2=s2)                               # Wo^T dOut
        dqkv = torch.empty_like(qkv)
        for n in range(N):
            q, k, v, Pm = qkv[n, :C], qkv[n, C:2 * C], qkv[n, 2 * C:], probs[n]
            s2 = prob_scale(dev)
            ps = split_planes_strided(Pm, 1, L, L, 0, L, 1, s2)                                # X[k=i][p=j] = P[i][j]
            gemm_f16x3(ps, gemm_pack_a(dO[n], C, L, L, 1), 1, L, C, L, x_scale2=s2, out=dqkv[n, 2 * C:])        # dV
            vs, s2 = _scaled_planes(v, 1, C, L, 0, L, 1)
            dP = gemm_f16x3(vs, gemm_pack_a(dO[n], L, C, 1, L), 1, C, L, L, x_scale2=s2)[0]    # dP[i][j] = dO_i . v_j
            dS = softmax_rows_backward(Pm, dP, C ** -0.5)
            del dP
            s2 = absmax_scale(dS)
            dst = split_planes_strided(dS, 1, L, L, 0, 1, L, s2)                               # X[k=j][p=i] = dS[i][j]
            gemm_f16x3(dst, gemm_pack_a(k, C, L, L, 1), 1, L, C, L, x_scale2=s2, out=dqkv[n, :C])               # dQ
            dsn = split_planes_strided(dS, 1, L, L, 0, L, 1, s2)                               # X[k=i][p=j] = dS[i][j]
            gemm_f16x3(dsn, gemm_pack_a(q, C, L, L, 1), 1, L, C, L, x_scale2=s2, out=dqkv[n, C:2 * C])          # dK
        dqs, s2 = _scaled_planes(dqkv, N, 3 * C, L, 3 * C * L, L, 1)
        dh = gemm_f16x3(dqs, wts['qkv_t'], N, 3 * C, C, L, x_scale2=s2)                         # Wqkv^T dqkv
        dx = torch.empty_like(xc)
        ws = _gn_workspace(N, C, L, dev)
        # + the gradient arriving over the block's residual connection, and the maxima of the sum for the next backward's scale
        maxima = torch.empty(int(_lib.load().mvip_groupnorm_backward_maxima(N, C, L)), device=dev, dtype=torch.float32)
        call('mvip_groupnorm_backward_fused', ptr(xc), ptr(dh.reshape(N, C, H, W)), ptr(gw), ptr(gb), ptr(mean), ptr(rstd), N,
             C, L, norm.num_groups, 0, 0, ptr(d), ptr(dx), ptr(maxima), ptr(ws, torch.float64), stream())
        _LAST_DX[0] = (dx, dx._version, maxima)
        return dx, None


def vae_attention(x, mod):
    return _VAEAttention.apply(x, mod)


# 1x1 convolutions of the SDS networks on the split-precision GEMM ------------------------------------------------

def conv1x1_supported(conv, x, tokens=False):
    """x: [N, Cin, H, W] (or [N, L, Cin] token-major when tokens=True), fp32 on the device."""
    if not (x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.weight.dtype == torch.float32):
        return False
    L = x.shape[1] if tokens else x.shape[2] * x.shape[3]
    return conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0 and L % 256 == 0


def _conv1x1_packed(conv, transpose):
    cache = conv.__dict__.setdefault('_mvip_packed', {})
    key = (conv.weight.data_ptr(), conv.weight._version)
    if cache.get('key') != key:
        cache.clear()
        cache['key'] = key
    if transpose not in cache:
        Cout, Cin = conv.out_channels, conv.in_channels
        w = conv.weight.detach().reshape(Cout, Cin).contiguous()
        cache[transpose] = gemm_pack_a(w, Cin, Cout, 1, Cin, weights=True) if transpose else gemm_pack_a(w, Cout, Cin, Cin, 1, weights=True)
    return cache[transpose]


class _Conv1x1(torch.autograd.Function):
    """conv1x1(x) + bias [+ residual] for x [N, Cin, H, W]; the data gradient runs the transposed image."""

    @staticmethod
    def forward(ctx, x, residual, conv):
        ctx.prec = _prec()
        if conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad):
            raise NotImplementedError('conv1x1: parameter gradients are not implemented (frozen networks only)')
        xc = x.contiguous()
        N, C, H, W = xc.shape
        L, Cout = H * W, conv.out_channels
        xs, s2 = _scaled_planes(xc, N, C, L, C * L, L, 1, forward_activation=(not x.requires_grad) or _prec() == 1)
        bias = None if conv.bias is None else conv.bias.detach().contiguous()
        rs = None if residual is None else residual.detach().contiguous()
        y = gemm_f16x3(xs, _conv1x1_packed(conv, False), N, C, Cout, L, bias=bias, residual=rs, x_scale2=s2)
        ctx.conv, ctx.shape = conv, (N, C, H, W)
        return y.reshape(N, Cout, H, W)

    @staticmethod
    @_with_saved_prec
    def backward(ctx, dy):
        conv = ctx.conv
        N, C, H, W = ctx.shape
        L, Cout = H * W, conv.out_channels
        d = dy.contiguous().float()
        dx = None
        if ctx.needs_input_grad[0]:
            ds, s2 = _scaled_planes(d, N, Cout, L, Cout * L, L, 1)
            dx = gemm_f16x3(ds, _conv1x1_packed(conv, True), N, Cout, C, L, x_scale2=s2).reshape(N, C, H, W)
        return dx, (d if ctx.needs_input_grad[1] else None), None


def conv1x1(x, conv, residual=None):
    return _Conv1x1.apply(x, residual, conv)


# General convolution as im2col planes + the split-precision GEMM: the layers the 3x3 stride-1 kernel cannot take --------
def conv_gemm_supported(conv, x):
    """Stride-1 / stride-2 square kernels (1x1, 3x3), any channel counts, fp32 device tensors: the UNet's and the VAE
    encoder's down-samplers, conv_in / conv_out (3, 4, 8, 9 channels), quant_conv."""
    if not (isinstance(conv, torch.nn.Conv2d) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    if torch.is_grad_enabled() and (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad)):
        return False                                  # frozen networks only: no parameter gradients here
    kh, kw = conv.kernel_size
    return (kh == kw and kh in (1, 3) and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32
            and isinstance(conv.padding, tuple) and conv.padding_mode == 'zeros')


def _up(v, m):
    return (v + m - 1) // m * m


def _conv_gemm_packed(conv):
    """(A forward [MP x KP], A transposed [KP x MP], bias padded to MP): the weight as [Cout][Cin*KH*KW], zero padded
    to the GEMM's multiples of 32, packed once per (frozen) layer."""
    cache = conv.__dict__.setdefault('_mvip_packed_gemm', {})
    key = (conv.weight.data_ptr(), conv.weight._version)
    if cache.get('key') != key:
        cache.clear()
        cache['key'] = key
        Cout, K = conv.out_channels, conv.in_channels * conv.kernel_size[0] * conv.kernel_size[1]
        MP, KP = _up(Cout, 32), _up(K, 32)
        w = torch.zeros((MP, KP), device=conv.weight.device, dtype=torch.float32)
        w[:Cout, :K] = conv.weight.detach().reshape(Cout, K)
        cache['fwd'] = gemm_pack_a(w, MP, KP, KP, 1, weights=True)
        cache['bwd'] = gemm_pack_a(w, KP, MP, 1, KP, weights=True)
        b = torch.zeros(MP, device=w.device, dtype=torch.float32)
        if conv.bias is not None:
            b[:Cout] = conv.bias.detach()
        cache['bias'] = b
    return cache['fwd'], cache['bwd'], cache['bias']


class _ConvGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, conv, pads):
        ctx.prec = _prec()
        if conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad):
            raise NotImplementedError('conv_gemm: parameter gradients are not implemented (frozen networks only)')
        xc = _f32c(x)
        N, Cin, H, W = xc.shape
        k, st = conv.kernel_size[0], conv.stride[0]
        pt, pl, pb, pr = pads
        OH, OW = (H + pt + pb - k) // st + 1, (W + pl + pr - k) // st + 1
        Cout, K = conv.out_channels, Cin * k * k
        MP, KP, P, PP = _up(Cout, 32), _up(K, 32), OH * OW, _up(OH * OW, 256)
        a_fwd, _, bias = _conv_gemm_packed(conv)
        s2 = unit_scale(xc.device) if forward_unit_scale() else absmax_scale(xc)
        xs = _split_buffer(N, KP, PP, xc.device)
        call('mvip_im2col_split_planes', ptr(xc), N, Cin, H, W, k, k, st, pt, pl, OH, OW, KP, PP, ptr(s2),
             ptr(xs, torch.float16), _prec(), stream())
        y = gemm_f16x3(xs, a_fwd, N, KP, MP, PP, bias=bias, x_scale2=s2)
        del xs
        ctx.conv, ctx.geom = conv, (N, Cin, H, W, k, st, pt, pl, OH, OW, Cout, MP, KP, P, PP)
        if MP != Cout or PP != P:
            y = y[:, :Cout, :P]
        return y.reshape(N, Cout, OH, OW)

    @staticmethod
    @_with_saved_prec
    def backward(ctx, dy):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        N, Cin, H, W, k, st, pt, pl, OH, OW, Cout, MP, KP, P, PP = ctx.geom
        d = _f32c(dy).reshape(N, Cout, P)
        if MP != Cout or PP != P:
            dp = torch.zeros((N, MP, PP), device=d.device, dtype=torch.float32)
            dp[:, :Cout, :P] = d
            d = dp
        _, a_bwd, _ = _conv_gemm_packed(ctx.conv)
        ds, s2 = _scaled_planes(d, N, MP, PP, MP * PP, PP, 1)
        col = gemm_f16x3(ds, a_bwd, N, MP, KP, PP, x_scale2=s2)                 # [N, KP, PP]
        del ds
        dx = torch.empty((N, Cin, H, W), device=d.device, dtype=torch.float32)
        call('mvip_col2im', ptr(col), N, Cin, H, W, k, k, st, pt, pl, OH, OW, KP, PP, ptr(dx), stream())
        return dx, None, None


def conv_gemm(x, conv, pads=None):
    """conv(x) + bias for the layers conv_gemm_supported accepts; `pads` = (top, left, bottom, right) overrides the
    module's symmetric padding (the VAE down-samplers pad bottom / right only).  Differentiable w.r.t. x."""
    if pads is None:
        pads = (conv.padding[0], conv.padding[1], conv.padding[0], conv.padding[1])
    return _ConvGemm.apply(x, conv, tuple(int(v) for v in pads))


def norm_conv1x1(x, norm, conv, ln_stats=False):
    """conv1x1(group_norm(x)) + bias, forward only (the UNet's transformer proj_in; runs under no_grad).
    ln_stats: as gemm_f16x3's (returns (y, stats))."""
    xc = x.detach().contiguous()
    N, C, H, W = xc.shape
    L, G, dev = H * W, norm.num_groups, xc.device
    ws = _gn_workspace(N, C, L, dev)
    xs = _split_buffer(N, C, L, dev)
    gw, gb = norm.weight.detach().contiguous(), norm.bias.detach().contiguous()
    if C // G < 4:
        mean, rstd = _gn_stats(xc, N, C, H, W, G, norm.eps, ws)
        call('mvip_groupnorm_split_planes', ptr(xc), ptr(gw), ptr(gb), ptr(mean), ptr(rstd), N, C, L, G, 0,
             ptr(xs, torch.float16), _prec(), stream())
    else:           # forward only: the plane writer reduces the moment partials itself
        rm = _row_moments_of(xc)
        if rm is None:
            call('mvip_groupnorm_stats', ptr(xc), N, C, L, G, float(norm.eps), 0, None, None, ptr(ws, torch.float64), stream())
            rm = ws
        call('mvip_groupnorm_split_planes_moments', ptr(xc), ptr(gw), ptr(gb), ptr(rm, torch.float64), float(norm.eps),
             N, C, L, G, 0, ptr(xs, torch.float16), _prec(), stream())
    bias = None if conv.bias is None else conv.bias.detach().contiguous()
    return gemm_f16x3(xs, _conv1x1_packed(conv, False), N, C, conv.out_channels, L, bias=bias, ln_stats=ln_stats)     # [N, Cout, L]


def tokens_conv1x1(h, conv, residual):
    """conv1x1 of token-major activations h [N, L, Cin] back to channel-first, + bias + residual [N, Cout, H, W];
    forward only (the UNet's transformer proj_out).  The token -> channel-first permute is folded into the
    split-plane writer's strides."""
    hc = h.detach().contiguous()
    N, L, C = hc.shape
    xs, s2 = _scaled_planes(hc, N, C, L, L * C, 1, C)
    bias = None if conv.bias is None else conv.bias.detach().contiguous()
    rs = residual.detach().contiguous()
    return gemm_f16x3(xs, _conv1x1_packed(conv, False), N, C, conv.out_channels, L, bias=bias, residual=rs,
                      x_scale2=s2).reshape(rs.shape)


# Transformer blocks of the SDS UNet: attention and token-side kernels (csrc/attention.hip, csrc/transformer.hip) -----

_ZERO_WORDS = {}
ZERO_SCOPE = None      # set by a hipGraph capture (guidance/sd_utils._GraphedStep) to a token of its own: every capture runs on
                       # torch's ONE shared capture stream, so without it all captured steps would bake in the SAME scratch words
                       # -- harmless while graphs replay one after the other, a race once two of them replay on different streams


def _zero_words(device):
    """64 scratch words per (device, stream[, capturing graph]) that are zero between kernel launches: the absmax collectors
    use them with atomics and the kernel that reads the maximum re-zeroes them (saves a zeroing launch per use)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, ZERO_SCOPE)
    if key not in _ZERO_WORDS:
        _ZERO_WORDS[key] = torch.zeros(64, device=device, dtype=torch.int32)
    return _ZERO_WORDS[key]


def absmax_scale_sections(x, outer, sections, length):
    """x viewed as [outer][sections][length] -> flat [sections * 4] device floats: {2^k, 2^-k, -, -} per section."""
    sc = torch.empty(sections * 4, device=x.device, dtype=_F32)
    call('mvip_absmax_scale_sections', ptr(x), int(outer), int(sections), int(length), ptr(sc),
         ptr(_zero_words(x.device), torch.int32), stream())
    return sc


def layernorm_split(x, weight, bias, eps, N, C, L, LP, out_scale, stats=None):
    """LayerNorm over the channel axis of channel-major x [N, C, LP] (tokens < L), times the power of two
    `out_scale`, as fp16 hi/lo split planes (the B operand of gemm_f16x3 with P = LP).  stats: the (partial moments,
    segments) pair the GEMM that produced x left (gemm_f16x3(..., ln_stats=True)): the statistics launch is skipped."""
    xs = _split_buffer(N, C, LP, x.device)
    if stats is not None:
        part, S = stats
        call('mvip_layernorm_split_planes_stats', ptr(x), ptr(weight), ptr(bias), ptr(part, torch.float64), int(S), int(N), int(C),
             int(L), int(LP), float(eps), float(out_scale), ptr(xs, torch.float16), _prec(), stream())
        return xs
    ws = torch.empty(int(_lib.load().mvip_layernorm_workspace_bytes(N, C, LP)) // 8, device=x.device, dtype=torch.float64)
    call('mvip_layernorm_split_planes', ptr(x), ptr(weight), ptr(bias), int(N), int(C), int(L), int(LP), float(eps),
         float(out_scale), ptr(ws, torch.float64), ptr(xs, torch.float16), _prec(), stream())
    return xs


def attention_pack_v(v, N, heads, D, DP, Lk, LkP, sn, sr, sk, scale2):
    """v.flatten()[n*sn + (h*DP + d)*sr + key*sk] -> the attention kernel's V operand (A fragments, fp16 hi/lo)."""
    nbytes = int(_lib.load().mvip_attention_v_bytes(N, heads, D, LkP))
    vp = torch.empty(nbytes, device=v.device, dtype=torch.uint8)
    call('mvip_attention_pack_v', ptr(v), int(N), int(heads), int(D), int(DP), int(Lk), int(LkP), int(sn), int(sr),
         int(sk), ptr(scale2), ptr(vp, torch.uint8), _prec(), stream())
    return vp


ATTENTION_FLAGS = 0      # bit 0: 64-key LDS tiles for 40-channel heads (A/B timing switch)


def attention_f16x3(qs, ks, vp, q_scale2, k_scale2, v_scale2, N, heads, D, Lq, LqP, Lk, LkP, out=None):
    """softmax(q k^T / sqrt(D)) v per head on the split-precision flash kernel -> fp32 [N, heads*D, LqP]
    (columns < Lq written; allocate `out` zeroed when LqP > Lq)."""
    if out is None:
        mk = torch.empty if LqP == Lq else torch.zeros
        out = mk((N, heads * D, LqP), device=qs.device, dtype=_F32)
    call('mvip_attention_f16x3', ptr(qs, torch.float16), ptr(ks, torch.float16), ptr(vp, torch.uint8), ptr(q_scale2),
         ptr(k_scale2), ptr(v_scale2), int(N), int(heads), int(D), int(Lq), int(LqP), int(Lk), int(LkP),
         float(D) ** -0.5, int(ATTENTION_FLAGS), ptr(out), _prec(), stream())
    return out


def geglu(y, N, R, L, LP):
    """y [N, 2R, LP] -> (y[:, :R] * gelu(y[:, R:]) [N, R, LP] zero beyond L, its power-of-two scale2)."""
    out = torch.empty((N, R, LP), device=y.device, dtype=_F32)
    scale2 = torch.empty(4, device=y.device, dtype=_F32)
    call('mvip_geglu', ptr(y), int(N), int(R), int(L), int(LP), ptr(out), ptr(scale2),
         ptr(_zero_words(y.device), torch.int32), stream())
    return out, scale2


def geglu_interleave(weight, bias):
    """[2R, K] weight / [2R] bias of a GEGLU projection (value rows, then gate rows) -> the same rows interleaved in
    32-row tiles (value tile t, gate tile t), the order `gemm_geglu_f16x3` expects."""
    R = weight.shape[0] // 2
    w = torch.stack([weight[:R].reshape(R // 32, 32, -1), weight[R:].reshape(R // 32, 32, -1)], 1).reshape(2 * R, -1)
    b = torch.stack([bias[:R].reshape(R // 32, 32), bias[R:].reshape(R // 32, 32)], 1).reshape(2 * R)
    return w.contiguous(), b.contiguous()


def gemm_geglu_f16x3(xs, packed, bias, N, K, M2, P, L, x_scale2=None):
    """(value * gelu(gate)) of the interleaved projection `packed` applied to split planes xs -> ([N, M2/2, P] fp32,
    its power-of-two scale2); columns >= L are zero."""
    out = torch.empty((N, M2 // 2, P), device=xs.device, dtype=_F32)
    scale2 = torch.empty(4, device=xs.device, dtype=_F32)
    call('mvip_gemm_geglu_f16x3', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(x_scale2), int(N),
         int(K), int(M2), int(P), int(L), ptr(out), ptr(scale2), ptr(_zero_words(xs.device), torch.int32), _prec_w(packed), stream())
    return out, scale2


def linear_small(x, W, b, act_in=0):
    """act(x) @ W^T + b for x [NB <= 8, K] in exact fp32 (act_in: 0 identity, 1 SiLU)."""
    xc, Wc = _f32c(x), _f32c(W)
    NB, K = xc.shape
    M = Wc.shape[0]
    y = torch.empty((NB, M), device=xc.device, dtype=_F32)
    call('mvip_linear_small', ptr(xc), ptr(Wc), ptr(None if b is None else _f32c(b)), NB, M, K, int(act_in), ptr(y),
         stream())
    return y


# Contractions that hand each other operands (csrc/plane_sink.h): the producer's epilogue writes the consumer's operand
# format at a power-of-two scale fixed before the launch ------------------------------------------------------------------

def pow2_scale_for_bound(bound, top=32768.0):
    """Largest power of two s with bound * s <= top (fp16 overflows at 65504; the hi/lo pair keeps ~2^-25 of `top`
    absolutely, i.e. fp32-grade accuracy relative to the tensor's maximum while the bound is < ~2^15 too wide)."""
    import math
    if not (bound > 0.0) or not math.isfinite(bound):
        return 1.0
    k = int(math.floor(math.log2(top / bound)))
    return float(2.0 ** max(-60, min(60, k)))


def scale2_tensor(s, device):
    return torch.tensor([s, 1.0 / s, 0.0, 0.0], device=device, dtype=_F32)


def gemm_f16x3_sinks(xs, packed, N, K, P, sections, bias=None, x_scale2=None, v_dt=1):
    """(W X + bias) with the rows cut into `sections` = [(rows, kind, scale), ...]: kind 'planes' -> fp16 hi/lo split
    planes [N][rows/16][2][2][P][8], kind 'vfrag' (last section only) -> attention V fragments
    [N][rows/(32 v_dt)][v_dt][P/16][2][64][8]; returns the list of operand buffers (uint8 / fp16 storage)."""
    import ctypes
    n = len(sections)
    M = sum(r for r, _, _ in sections)
    rows = (ctypes.c_int64 * n)(*[int(r) for r, _, _ in sections])
    kinds = (ctypes.c_int * n)(*[1 if k == 'planes' else 2 for _, k, _ in sections])
    scales = (ctypes.c_float * n)(*[float(sc) for _, _, sc in sections])
    bufs = []
    for r, k, _ in sections:
        if k == 'planes':
            bufs.append(_split_buffer(N, r, P, xs.device))                          # 4 B per element: fp16 hi + lo
        else:
            bufs.append(torch.empty(N * r * P * 4, device=xs.device, dtype=torch.uint8))
    ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bufs])
    call('mvip_gemm_f16x3_sinks', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(x_scale2), int(N), int(K),
         int(M), int(P), n, rows, kinds, ptrs, scales, int(v_dt), _prec_w(packed), stream())
    return bufs


def gemm_geglu_f16x3_sink(xs, packed, bias, N, K, M2, P, L, out_scale, x_scale2=None):
    """(value * gelu(gate)) * out_scale of the interleaved projection as the second projection's operand planes
    [N][(M2/2)/16][2][2][P][8]; columns >= L are zero."""
    out = _split_buffer(N, M2 // 2, P, xs.device)
    call('mvip_gemm_geglu_f16x3_sink', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(x_scale2), int(N),
         int(K), int(M2), int(P), int(L), ptr(out, torch.float16), float(out_scale), _prec_w(packed), stream())
    return out


def attention_f16x3_sink(qs, ks, vp, q_scale2, k_scale2, v_scale2, N, heads, D, Lq, LqP, Lk, LkP, q_stride, k_stride,
                         v_groups):
    """mvip_attention_f16x3 on operands written by GEMM epilogues; the result leaves as the output projection's operand
    planes [N][heads*D/16][2][2][LqP][8], scaled by V's scale (zero beyond Lq when LqP > Lq)."""
    mk = torch.empty if LqP == Lq else torch.zeros
    out = mk(N * heads * D * LqP * 2, device=qs.device, dtype=torch.float16)
    call('mvip_attention_f16x3_sink', ptr(qs, torch.float16), ptr(ks, torch.float16), ptr(vp, torch.uint8), ptr(q_scale2),
         ptr(k_scale2), ptr(v_scale2), int(N), int(heads), int(D), int(Lq), int(LqP), int(Lk), int(LkP), int(q_stride),
         int(k_stride), int(v_groups), float(D) ** -0.5, int(ATTENTION_FLAGS), ptr(out, torch.float16), _prec(), stream())
    return out


def gemm_f16x3_planes(xs, packed, N, K, M, P, out_scale, bias=None, residual=None, x_scale2=None):
    """(W X + bias + residual) * out_scale as split planes [N][M/16][2][2][P][8] (the next GEMM's operand); split-K as
    gemm_f16x3 for the small grids."""
    out = _split_buffer(N, M, P, xs.device)
    nbytes = int(_lib.load().mvip_gemm_workspace_bytes(N, K, M, P))
    ws = torch.empty(nbytes // 4, device=xs.device, dtype=torch.float32) if nbytes else None
    call('mvip_gemm_f16x3_planes_ws', ptr(xs, torch.float16), ptr(packed, torch.uint8), ptr(bias), ptr(residual), ptr(x_scale2),
         int(N), int(K), int(M), int(P), ptr(out, torch.float16), float(out_scale), ptr(ws), _prec_w(packed), stream())
    return out


def linear_small_grouped(x, Wcat, bcat, offsets_dev, sizes, act_in=0):
    """act(x) @ W_l^T + b_l for several layers sharing x [NB <= 8, K] in ONE launch; returns one contiguous [NB, C_l]
    tensor per layer (views of one buffer).  Wcat [sum C_l, K], offsets_dev int32 [L + 1] on the device."""
    xc = _f32c(x)
    NB, K = xc.shape
    M = Wcat.shape[0]
    y = torch.empty(NB * M, device=xc.device, dtype=_F32)
    call('mvip_linear_small_grouped', ptr(xc), ptr(Wcat), ptr(bcat), NB, M, K, int(act_in), ptr(offsets_dev, torch.int32),
         len(sizes), ptr(y), stream())
    out, o = [], 0
    for c in sizes:
        out.append(y[o * NB:(o + c) * NB].view(NB, c))
        o += c
    return out


# render_rays in two launches (csrc/mlp_fwd16.hip, FUSE = 1 / 2): no-grad renders of the native networks ----------------

def render_coarse_fused(packed16, rows, lindisp, t_rand, noise, u, white_bkgd, need_alpha=False):
    """Coarse pass of render_rays (64 samples) behind ONE launch: stratified depths -> network -> raw2outputs -> inverse-CDF
    resampling (u: [B, Nf] or a shared row [Nf], Nf <= 64) -> merged depths.  Returns (rgb0, disp0, acc0, alpha0 | None,
    z_merged [B, 64 + Nf], z_std), bit-identical to stratified_z -> mlp_rays -> composite -> sample_pdf_merge."""
    rows = _f32c(rows)
    B, dev = rows.shape[0], rows.device
    u = _f32c(u)
    Nf = u.shape[-1]
    rgb = torch.empty((B, 3), device=dev, dtype=_F32)
    disp = torch.empty((B,), device=dev, dtype=_F32)
    acc = torch.empty((B,), device=dev, dtype=_F32)
    alpha = torch.empty((B, 64), device=dev, dtype=_F32) if need_alpha else None
    zm = torch.empty((B, 64 + Nf), device=dev, dtype=_F32)
    zstd = torch.empty((B,), device=dev, dtype=_F32)
    tr = None if t_rand is None else _f32c(t_rand)
    nz = None if noise is None else _f32c(noise)
    call('mvip_render_coarse_fused', ptr(packed16), ptr(rows), B, ptr(_t_vals(64, dev)), int(bool(lindisp)), ptr(tr), ptr(nz),
         ptr(u), int(u.dim() == 1), int(Nf), COMP_WHITE if white_bkgd else 0, ptr(rgb), ptr(disp), ptr(acc), ptr(None),
         ptr(None), ptr(alpha), ptr(zm), ptr(zstd), stream())
    return rgb, disp, acc, alpha, zm, zstd


def render_fine_fused(packed16, rows, z, noise, white_bkgd, need_alpha=False, want_raw=False):
    """Fine pass of render_rays (128 samples at depths z) behind ONE launch: network -> raw2outputs.  Returns (rgb, disp,
    acc, weights, depth, alpha | None, raw | None), bit-identical to mlp_rays -> composite."""
    rows, z = _f32c(rows), _f32c(z)
    B, dev = rows.shape[0], rows.device
    rgb = torch.empty((B, 3), device=dev, dtype=_F32)
    disp = torch.empty((B,), device=dev, dtype=_F32)
    acc = torch.empty((B,), device=dev, dtype=_F32)
    depth = torch.empty((B,), device=dev, dtype=_F32)
    weights = torch.empty((B, 128), device=dev, dtype=_F32)
    alpha = torch.empty((B, 128), device=dev, dtype=_F32) if need_alpha else None
    raw = torch.empty((B, 128, 4), device=dev, dtype=_F32) if want_raw else None
    nz = None if noise is None else _f32c(noise)
    call('mvip_render_fine_fused', ptr(packed16), ptr(rows), ptr(z), B, ptr(nz), COMP_WHITE if white_bkgd else 0, ptr(raw),
         ptr(rgb), ptr(disp), ptr(acc), ptr(depth), ptr(weights), ptr(alpha), stream())
    return rgb, disp, acc, weights, depth, alpha, raw


def softmax_rows(S, scale):
    """softmax(scale * S, -1) of a 2-D fp32 matrix on the HIP row kernel (the VAE mid-block attention's scores)."""
    Sc = _f32c(S)
    P = torch.empty_like(Sc)
    call('mvip_softmax_rows', ptr(Sc), Sc.shape[0], Sc.shape[1], float(scale), ptr(P), stream())
    return P


def softmax_rows_backward(P, dP, scale):
    """scale * P * (dP - rowsum(dP * P)): the adjoint of softmax_rows."""
    Pc, dc = _f32c(P), _f32c(dP)
    dS = torch.empty_like(Pc)
    call('mvip_softmax_rows_backward', ptr(Pc), ptr(dc), Pc.shape[0], Pc.shape[1], float(scale), ptr(dS), stream())
    return dS
