"""Thin tensor-level wrappers over the C ABI (allocation + autograd plumbing only).

Every function here launches HIP kernels from libmvipnerf.so on the current torch stream.
Nothing falls back to torch ops or the CPU: inputs must be dense fp32 tensors on the GPU.
"""
import ctypes
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


def mlp_pack_f16x3(params, packed_f32):
    """Image for the split-precision forward (precision=1); same size as the fp32 image."""
    ps = [_f32c(p.detach()) for p in params]
    img = torch.empty(packed_floats(), device=ps[0].device, dtype=_F32)
    call('mvip_mlp_pack_f16x3', _lib.ptr_array(ps), ptr(img), ptr(packed_f32), stream())
    return img


def mlp_unpack_grads(grad_packed, like):
    grads = [torch.empty(shp, device=grad_packed.device, dtype=_F32) for shp in PARAM_SHAPES]
    call('mvip_mlp_unpack_grads', ptr(grad_packed), _lib.ptr_array(grads), 0, stream())
    return grads


_WORKSPACE = {}
BWD_TILE_POINTS = 65536          # points per recompute/backward tile (1.3 GB of stash workspace)


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


# Keep the activations of a training forward for its backward when the live stashes stay under
# this many bytes (9.9 KB per point); beyond it the backward recomputes them tile by tile.
STASH_BUDGET_BYTES = 96 << 30
_stash_live = [0]


def _take_stash(P, device):
    n = int(_lib.load().mvip_mlp_stash_floats(P))
    if _stash_live[0] + 4 * n > STASH_BUDGET_BYTES:
        return None
    _stash_live[0] += 4 * n
    t = torch.empty(n, device=device, dtype=_F32)
    weakref.finalize(t, _stash_freed, 4 * n)        # also covers graphs that are dropped without backward
    return t


def _stash_freed(nbytes):
    _stash_live[0] -= nbytes


class _MLPRays(torch.autograd.Function):
    """raw[B,S,4] = MLP(enc(o + d z), enc(viewdir)); gradients flow to the 24 parameters only."""

    @staticmethod
    def forward(ctx, rows, z, packed, precision, *params):
        """`packed` is the weight image of `precision` (0: fp32 image, 1: f16x3 image)."""
        B, S = z.shape
        raw = torch.empty((B, S, 4), device=z.device, dtype=_F32)
        stash = _take_stash(B * S, z.device)
        if stash is None:
            call('mvip_mlp_forward_rays', ptr(packed), ptr(rows), ptr(z), B, S, ptr(raw), precision, stream())
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
        return (None, None, None, None, *grads)


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


def mlp_rays(rows, z, packed, params, packed_f16x3=None, train_f16x3=None):
    """Fused forward from ray rows + depths.  `params` (the 24 tensors) are passed so autograd
    routes the gradients back to them; with no grad needed the Function is skipped.
    `packed_f16x3` selects the split-precision kernel (precision = 1) for no-grad calls,
    `train_f16x3` (the same kind of image) for calls that will be back-propagated."""
    rows, z = _f32c(rows), _f32c(z)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        if train_f16x3 is not None:
            return _MLPRays.apply(rows, z, train_f16x3, 1, *params)
        return _MLPRays.apply(rows, z, packed, 0, *params)
    B, S = z.shape
    raw = torch.empty((B, S, 4), device=z.device, dtype=_F32)
    if packed_f16x3 is not None:
        call('mvip_mlp_forward_rays', ptr(packed_f16x3), ptr(rows), ptr(z), B, S, ptr(raw), 1, stream())
    else:
        call('mvip_mlp_forward_rays', ptr(packed), ptr(rows), ptr(z), B, S, ptr(raw), 0, stream())
    return raw


def mlp_points(pts, dirs, packed, params, packed_f16x3=None, train_f16x3=None):
    pts, dirs = _f32c(pts), _f32c(dirs)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        if train_f16x3 is not None:
            return _MLPPoints.apply(pts, dirs, train_f16x3, 1, *params)
        return _MLPPoints.apply(pts, dirs, packed, 0, *params)
    raw = torch.empty((pts.shape[0], 4), device=pts.device, dtype=_F32)
    if packed_f16x3 is not None:
        call('mvip_mlp_forward_points', ptr(packed_f16x3), ptr(pts), ptr(dirs), pts.shape[0], ptr(raw), 1, stream())
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
