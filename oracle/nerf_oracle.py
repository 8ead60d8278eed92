"""CPU oracle for the NeRF volume-rendering hot path (TEST INFRASTRUCTURE ONLY).

This module is a plain PyTorch-CPU fp32 restatement of the reference algorithm.  It exists to
*check* the HIP path; nothing in the product package (`mvip_nerf_amd/`) may import it.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it.

Parity status: PINNED.  Every function here is compared against outputs of the real reference
(`/root/reference/DS_NeRF`, imported in the build container by `oracle/gen_golden.py`) through the
fixtures committed under `tests/golden/` (`tests/test_oracle_golden.py`).

All random inputs (stratified jitter `t_rand`, density noise, inverse-CDF uniforms `u`) are
explicit arguments, which is what the reference's own `pytest=True` hooks do
(DS_NeRF/run.py:1776-1779, DS_NeRF/run_nerf_helpers.py:319-327, :378-381).

Reference citations are relative to /root/reference/.
"""
import math
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# a1  ray generation                                   DS_NeRF/run_nerf_helpers.py:249-260
# ----------------------------------------------------------------------------------------------

def get_rays(H, W, focal, c2w):
    """Pinhole rays, pixel centres at integer coordinates (no half-pixel shift), camera looks
    down -z, directions NOT normalised.  Returns (rays_o, rays_d) each [H, W, 3] fp32."""
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    xs = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)
    ys = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
    cam = torch.stack([(xs - W * .5) / focal, -(ys - H * .5) / focal, -torch.ones(H, W)], -1)
    # world = R @ cam, written as the reference writes it: sum_k cam[k] * R[row, k]
    rays_d = (cam[..., None, :] * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, 3].expand(H, W, 3)
    return rays_o, rays_d


def assemble_ray_batch(rays_o, rays_d, near, far, use_viewdirs=True):
    """The [B, 8|11] row layout render() hands to render_rays(): o, d, near, far, (viewdirs).
    DS_NeRF/run.py:1182-1207."""
    o = rays_o.reshape(-1, 3).float()
    d = rays_d.reshape(-1, 3).float()
    cols = [o, d, near * torch.ones_like(d[:, :1]), far * torch.ones_like(d[:, :1])]
    if use_viewdirs:
        v = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
        cols.append(v.reshape(-1, 3).float())
    return torch.cat(cols, -1)

# ----------------------------------------------------------------------------------------------
# a3  stratified sampling                              DS_NeRF/run.py:1759-1783
# ----------------------------------------------------------------------------------------------

def stratified_z(near, far, n_samples, lindisp, t_rand=None):
    """near, far: [B, 1].  t_rand: [B, n_samples] uniforms or None (no jitter)."""
    t = torch.linspace(0., 1., steps=n_samples)
    if lindisp:
        z = 1. / (1. / near * (1. - t) + 1. / far * t)
    else:
        z = near * (1. - t) + far * t
    z = z.expand(near.shape[0], n_samples)
    if t_rand is not None:
        mid = .5 * (z[:, 1:] + z[:, :-1])
        hi = torch.cat([mid, z[:, -1:]], -1)
        lo = torch.cat([z[:, :1], mid], -1)
        z = lo + (hi - lo) * t_rand
    return z

# ----------------------------------------------------------------------------------------------
# a4  sinusoidal encoding                              DS_NeRF/run_nerf_helpers.py:22-70
# ----------------------------------------------------------------------------------------------

def posenc(x, n_freqs):
    """[.., 3] -> [.., 3 + 6*n_freqs]: x, then per octave k: sin(x*2^k) (xyz), cos(x*2^k) (xyz)."""
    freqs = 2. ** torch.linspace(0., n_freqs - 1, steps=n_freqs)
    out = [x]
    for f in freqs:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)

# ----------------------------------------------------------------------------------------------
# a5  the 8x256 MLP                                    DS_NeRF/run_nerf_helpers.py:74-127
# ----------------------------------------------------------------------------------------------

def mlp_param_names(D=8):
    names = []
    for i in range(D):
        names += [f'pts_linears.{i}.weight', f'pts_linears.{i}.bias']
    for n in ('views_linears.0', 'feature_linear', 'alpha_linear', 'rgb_linear'):
        names += [n + '.weight', n + '.bias']
    return names


def mlp_init(seed, D=8, W=256, in_pts=63, in_dirs=27, skips=(4,)):
    """Default nn.Linear initialisation, layer creation order identical to the reference
    constructor (same distribution; fixtures carry explicit seeds, see oracle/weights.py)."""
    g = torch.Generator().manual_seed(seed)
    def lin(fan_in, fan_out):
        bound = 1. / math.sqrt(fan_in)
        w = (torch.rand(fan_out, fan_in, generator=g) * 2 - 1) * bound
        b = (torch.rand(fan_out, generator=g) * 2 - 1) * bound
        return w, b
    p = {}
    fan = in_pts
    for i in range(D):
        p[f'pts_linears.{i}.weight'], p[f'pts_linears.{i}.bias'] = lin(fan, W)
        fan = W + in_pts if i in skips else W
    p['views_linears.0.weight'], p['views_linears.0.bias'] = lin(in_dirs + W, W // 2)
    p['feature_linear.weight'], p['feature_linear.bias'] = lin(W, W)
    p['alpha_linear.weight'], p['alpha_linear.bias'] = lin(W, 1)
    p['rgb_linear.weight'], p['rgb_linear.bias'] = lin(W // 2, 3)
    return p


def mlp_forward(p, emb, in_pts=63, D=8, skips=(4,)):
    """emb: [P, 90] = cat[posenc(pts,10), posenc(dirs,4)] -> [P, 4] = (rgb raw, sigma raw)."""
    e_pts, e_dir = emb[:, :in_pts], emb[:, in_pts:]
    h = e_pts
    for i in range(D):
        h = F.relu(F.linear(h, p[f'pts_linears.{i}.weight'], p[f'pts_linears.{i}.bias']))
        if i in skips:
            h = torch.cat([e_pts, h], -1)
    sigma = F.linear(h, p['alpha_linear.weight'], p['alpha_linear.bias'])
    feat = F.linear(h, p['feature_linear.weight'], p['feature_linear.bias'])
    v = F.relu(F.linear(torch.cat([feat, e_dir], -1), p['views_linears.0.weight'],
                        p['views_linears.0.bias']))
    rgb = F.linear(v, p['rgb_linear.weight'], p['rgb_linear.bias'])
    return torch.cat([rgb, sigma], -1)


def query_network(p, pts, viewdirs, chunk=65536):
    """run_network (DS_NeRF/run.py:1108-1124): flatten, encode, broadcast dirs, chunk, reshape."""
    B, S, _ = pts.shape
    flat = pts.reshape(-1, 3)
    dirs = viewdirs[:, None].expand(B, S, 3).reshape(-1, 3)
    emb = torch.cat([posenc(flat, 10), posenc(dirs, 4)], -1)
    out = torch.cat([mlp_forward(p, emb[i:i + chunk]) for i in range(0, emb.shape[0], chunk)], 0)
    return out.reshape(B, S, 4)

# ----------------------------------------------------------------------------------------------
# a7  alpha compositing                                DS_NeRF/run_nerf_helpers.py:350-404
# ----------------------------------------------------------------------------------------------

def raw2outputs(raw, z, rays_d, noise=None, white_bkgd=False, detach_weights=False):
    """raw [B,S,4], z [B,S], rays_d [B,3], noise [B,S] (already scaled by raw_noise_std) or None.
    Returns rgb_map, disp_map, acc_map, weights, depth_map, alpha."""
    dz = z[:, 1:] - z[:, :-1]
    dz = torch.cat([dz, torch.full_like(dz[:, :1], 1e10)], -1)
    dz = dz * torch.norm(rays_d[:, None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sig = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1. - torch.exp(-F.relu(sig) * dz)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * trans
    wc = w.detach() if detach_weights else w
    rgb_map = (wc[..., None] * rgb).sum(-2)
    depth_map = (w * z).sum(-1)
    acc_map = w.sum(-1)
    disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[:, None])
    return rgb_map, disp_map, acc_map, w, depth_map, alpha

# ----------------------------------------------------------------------------------------------
# a8  inverse-CDF sampling                             DS_NeRF/run_nerf_helpers.py:304-347
# ----------------------------------------------------------------------------------------------

def sample_pdf(bins, weights, u):
    """bins [B,Nb], weights [B,Nb-1], u [B,Ns] -> (samples [B,Ns], inds int64 [B,Ns]).
    inds = #{cdf <= u} (searchsorted right=True)."""
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    lo = (inds - 1).clamp(min=0)
    hi = inds.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    t = (u - c_lo) / den
    return b_lo + t * (b_hi - b_lo), inds


def det_u(n, B):
    return torch.linspace(0., 1., steps=n).expand(B, n)

# ----------------------------------------------------------------------------------------------
# a10  render_rays                                      DS_NeRF/run.py:1703-1847
# ----------------------------------------------------------------------------------------------

def render_rays(ray_batch, p_coarse, p_fine, N_samples, N_importance=0, lindisp=False,
                white_bkgd=False, t_rand=None, noise0=None, u=None, noise1=None,
                retraw=False, need_alpha=False, detach_weights=False):
    """Random inputs: t_rand [B,Nc] or None (perturb==0); noise0 [B,Nc] / noise1 [B,Nc+Nf]
    (raw_noise_std-scaled) or None; u [B,Nf] or None (deterministic linspace, i.e. perturb==0)."""
    B = ray_batch.shape[0]
    o, d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    vd = ray_batch[:, -3:] if ray_batch.shape[-1] > 9 else None
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z = stratified_z(near, far, N_samples, lindisp, t_rand)
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    raw = query_network(p_coarse, pts, vd)
    rgb, disp, acc, w, depth, alpha = raw2outputs(raw, z, d, noise0, white_bkgd, detach_weights)
    ret = {}
    if N_importance > 0:
        rgb0, disp0, acc0, alpha0 = rgb, disp, acc, alpha
        mids = .5 * (z[:, 1:] + z[:, :-1])
        uu = det_u(N_importance, B) if u is None else u
        zs, inds = sample_pdf(mids, w[:, 1:-1], uu)
        zs = zs.detach()
        z, _ = torch.sort(torch.cat([z, zs], -1), -1)
        pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
        raw = query_network(p_fine if p_fine is not None else p_coarse, pts, vd)
        rgb, disp, acc, w, depth, alpha = raw2outputs(raw, z, d, noise1, white_bkgd, detach_weights)
        ret.update(rgb0=rgb0, disp0=disp0, acc0=acc0, z_std=torch.std(zs, dim=-1, unbiased=False),
                   _inds=inds, _z_samples=zs)
        if need_alpha:
            ret['alpha0'] = alpha0
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth, weights=w, z_vals=z)
    if retraw:
        ret['raw'] = raw
    if need_alpha:
        ret['alpha'] = alpha
    return ret

# ----------------------------------------------------------------------------------------------
# a12  depth -> points -> least-squares plane normal   DS_NeRF/run.py:1909-1940
# ----------------------------------------------------------------------------------------------

def depth2xyz(depth, K):
    """depth [H,W], K 3x3 -> [H,W,3] camera-space points (x right, y down, z = depth)."""
    H, W = depth.shape
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    hh = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
    ww = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)
    x = (ww - cx) * depth / fx
    y = (hh - cy) * depth / fy
    return torch.stack([x, y, depth], -1)


def normal_fit_unfold(points, k=31):
    """The reference's formulation: per pixel, A = the k*k window of points (zero padded),
    n = (A^T A)^-1 A^T 1.  points [1,3,H,W] -> [1,3,H,W].  O(k^2) memory; small inputs only."""
    B, C, H, W = points.shape
    cols = F.unfold(points, (k, k), padding=(k - 1) // 2)
    A = cols.transpose(1, 2).reshape(B, H, W, C, k * k).transpose(-1, -2)
    At = A.transpose(-1, -2)
    n = torch.linalg.inv(At @ A) @ At @ torch.ones(B, H, W, k * k, 1)
    return n.squeeze(-1).permute(0, 3, 1, 2)


def normal_fit_boxsum(points, k=31):
    """Same quantity through nine zero-padded box sums + a closed-form symmetric 3x3 solve
    (the formulation the HIP kernel uses).  fp64 accumulation for use as a checker."""
    P = points.double()
    x, y, z = P[:, 0:1], P[:, 1:2], P[:, 2:3]
    mom = torch.cat([x * x, x * y, x * z, y * y, y * z, z * z, x, y, z], 1)
    box = F.avg_pool2d(mom, k, stride=1, padding=(k - 1) // 2, count_include_pad=True) * (k * k)
    sxx, sxy, sxz, syy, syz, szz, sx, sy, sz = [box[:, i] for i in range(9)]
    c00 = syy * szz - syz * syz
    c01 = sxz * syz - sxy * szz
    c02 = sxy * syz - sxz * syy
    c11 = sxx * szz - sxz * sxz
    c12 = sxy * sxz - sxx * syz
    c22 = sxx * syy - sxy * sxy
    det = sxx * c00 + sxy * c01 + sxz * c02
    nx = (c00 * sx + c01 * sy + c02 * sz) / det
    ny = (c01 * sx + c11 * sy + c12 * sz) / det
    nz = (c02 * sx + c12 * sy + c22 * sz) / det
    return torch.stack([nx, ny, nz], 1).float()

# ----------------------------------------------------------------------------------------------
# misc                                                  DS_NeRF/run_nerf_helpers.py:15-18
# ----------------------------------------------------------------------------------------------

def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    return -10. * torch.log(x) / math.log(10.)


def bench_poses(n=60):
    """SURVEY.md §8(d) synthetic orbit: c2w_k = [R_y(theta_k) | (0.3 sin, 0, 0.3 cos)], 6 deg steps."""
    out = []
    for k in range(n):
        th = math.radians(6.0 * k)
        c, s = math.cos(th), math.sin(th)
        out.append(torch.tensor([[c, 0., s, 0.3 * s], [0., 1., 0., 0.], [-s, 0., c, 0.3 * c]],
                                dtype=torch.float32))
    return torch.stack(out, 0)
