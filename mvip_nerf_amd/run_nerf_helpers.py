"""Host-side mirror of the reference's `DS_NeRF/run_nerf_helpers.py` public names, backed by the HIP
kernels of libmvipnerf.so.  Same names, argument order, defaults and return structures, so
`from run_nerf_helpers import *` call sites (DS_NeRF/run.py:24) keep working.

Nothing here computes on the CPU or through stock torch kernels except where noted
(`get_rays_np`/`get_rays_by_coord_np` are numpy by definition in the reference; `ndc_rays` is off
the LLFF-no_ndc path and is plain tensor algebra).
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops

# Misc (DS_NeRF/run_nerf_helpers.py:15-18)
img2mse = lambda x, y: torch.mean((x - y) ** 2)
img2l1 = lambda x, y: torch.mean(torch.abs(x - y))
mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device))
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


class Embedder:
    """Positional encoding (DS_NeRF/run_nerf_helpers.py:22-52).  Only the configuration the
    reference ever builds is supported by the kernel: include_input, log-sampled octaves
    2^0..2^(L-1), [sin, cos]."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        if not (kwargs.get('include_input', True) and kwargs.get('log_sampling', True)
                and kwargs.get('input_dims', 3) == 3
                and kwargs.get('max_freq_log2') == kwargs.get('num_freqs') - 1):
            raise NotImplementedError('HIP encoder implements the reference configuration only')
        self.num_freqs = int(kwargs['num_freqs'])
        self.out_dim = 3 + 6 * self.num_freqs

    def embed(self, inputs):
        return ops.posenc(inputs, self.num_freqs)


def get_embedder(multires, i=0):
    if i == -1:
        return nn.Identity(), 3
    eo = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    embed = lambda x, eo=eo: eo.embed(x)
    return embed, eo.out_dim


class NeRF(nn.Module):
    """The 8x256 NeRF MLP (DS_NeRF/run_nerf_helpers.py:74-127): same constructor, same parameter
    names (`pts_linears.i`, `views_linears.0`, `feature_linear`, `alpha_linear`, `rgb_linear`), so
    reference checkpoints load.  The arithmetic runs in the fused HIP kernel; the nn.Linear
    modules only own the parameters."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.skips, self.use_viewdirs = skips, use_viewdirs
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W)
                                        for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        self._packed = None
        self._packed_key = None
        self._packed16 = None
        self._packed16_key = None
        self._packed_w16 = None
        self._packed_w16_key = None
        self._packed_hw16 = None
        self._packed_hw16_key = None
        self.two_wave_f16x3 = True        # no-grad split-precision forwards use csrc/mlp_fwd16_f16x3.hip (two waves per SIMD)
        self.two_wave_inference = True    # no-grad fp32 forwards use csrc/mlp_fwd16.hip (two waves per SIMD)
        self.two_wave_training = True     # ... and so does the stash-writing fp32 training forward from ray rows
        # 0: exact fp32 MFMA.  1: split-precision fp16 MFMA ("f16x3", ~1e-6 relative, fp32 accumulate).
        self.inference_precision = 0  # forward passes that need no gradient (rendering)
        self.train_precision = 0      # stash-writing forward, delta and weight-gradient kernels

    def _check_supported(self):
        if not (self.D == 8 and self.W == 256 and self.input_ch == 63 and self.input_ch_views == 27
                and list(self.skips) == [4] and self.use_viewdirs):
            raise NotImplementedError(
                'the fused HIP kernel is built for the north-star model: D=8, W=256, skips=[4], '
                'multires=10, multires_views=4, use_viewdirs=True')

    def invalidate_packed(self):
        """Drop the packed weight images.  They are keyed on (data_ptr, tensor version), which in-place writes through
        `.data` do NOT bump (p.data.copy_(), an all-reduce or broadcast on p.data): call this after any such write.
        load_state_dict() and the trainer's initial weight broadcast do."""
        self._packed = self._packed_key = None
        self._packed16 = self._packed16_key = None
        self._packed_w16 = self._packed_w16_key = None
        self._packed_hw16 = self._packed_hw16_key = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed()
        return out

    def param_list(self):
        """The 24 tensors in state-dict order (what mvip_mlp_pack expects)."""
        sd = dict(self.named_parameters())
        return [sd[k] for k in ops.PARAM_ORDER]

    def packed(self):
        """Packed weight image, rebuilt only when a parameter changed (tensor version counters)."""
        self._check_supported()
        ps = self.param_list()
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or key != self._packed_key:
            self._packed = ops.mlp_pack(ps)
            self._packed_key = key
        return self._packed

    def forward(self, x):
        """x: [P, 90] = cat[embed(pts), embed(dirs)] as run_network builds it.  Both encodings
        carry their raw input in the first three channels (include_input=True), which is all the
        fused kernel needs: it re-encodes on chip."""
        self._check_supported()
        pts, dirs = x[..., 0:3], x[..., self.input_ch:self.input_ch + 3]
        out = ops.mlp_points(pts.reshape(-1, 3), dirs.reshape(-1, 3), self.packed(), self.param_list())
        return out.reshape(*x.shape[:-1], 4)

    def packed_f16x3(self):
        packed = self.packed()
        if self._packed16 is None or self._packed16_key != self._packed_key:
            self._packed16 = ops.mlp_pack_f16x3(self.param_list(), packed)
            self._packed16_key = self._packed_key
        return self._packed16

    def packed_w16(self):
        """Image of the two-waves-per-SIMD exact-fp32 inference kernel (csrc/mlp_fwd16.hip)."""
        packed = self.packed()
        if self._packed_w16 is None or self._packed_w16_key != self._packed_key:
            self._packed_w16 = ops.mlp_pack16(self.param_list(), packed)
            self._packed_w16_key = self._packed_key
        return self._packed_w16

    def packed_f16x3_w16(self):
        """Image of the two-waves-per-SIMD split-precision inference kernel (csrc/mlp_fwd16_f16x3.hip)."""
        packed = self.packed()
        if self._packed_hw16 is None or self._packed_hw16_key != self._packed_key:
            self._packed_hw16 = ops.mlp_pack_f16x3_w16(self.param_list(), packed)
            self._packed_hw16_key = self._packed_key
        return self._packed_hw16

    def _infer16(self):
        no_grad = not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
        return self.packed_w16() if (self.two_wave_inference and no_grad and self.inference_precision == 0) else None

    def _train16(self):
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return self.packed_w16() if (self.two_wave_training and grad and self.train_precision == 0) else None

    def _fast_image(self):
        return self.packed_f16x3() if self.inference_precision == 1 else None

    def _fast_image_w16(self):
        no_grad = not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
        return self.packed_f16x3_w16() if (self.inference_precision == 1 and self.two_wave_f16x3 and no_grad) else None

    def _train_image(self):
        return self.packed_f16x3() if self.train_precision == 1 else None

    def _train_image_w16(self):
        return self.packed_f16x3_w16() if (self.train_precision == 1 and self.two_wave_f16x3) else None

    def query_points(self, pts, dirs):
        return ops.mlp_points(pts, dirs, self.packed(), self.param_list(), self._fast_image(), self._train_image(),
                              self._infer16(), self._fast_image_w16())

    def query_rays(self, rows, z):
        return ops.mlp_rays(rows, z, self.packed(), self.param_list(), self._fast_image(), self._train_image(),
                            self._infer16(), self._train16(), self._fast_image_w16(), self._train_image_w16())


# Ray helpers -------------------------------------------------------------------------------------

def get_rays(H, W, focal, c2w):
    """DS_NeRF/run_nerf_helpers.py:249-260."""
    return ops.get_rays(H, W, focal, torch.as_tensor(c2w))


def get_rays_np(H, W, focal, c2w):
    """DS_NeRF/run_nerf_helpers.py:263-272 (host/numpy in the reference too)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def get_rays_by_coord_np(H, W, focal, c2w, coords):
    """DS_NeRF/run_nerf_helpers.py:275-280."""
    i, j = (coords[:, 0] - W * 0.5) / focal, -(coords[:, 1] - H * 0.5) / focal
    dirs = np.stack([i, j, -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """DS_NeRF/run_nerf_helpers.py:283-300.  Not on the no_ndc hot path; tensor algebra only."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# Hierarchical sampling ----------------------------------------------------------------------------

def _uniforms(shape, n, det, pytest, device):
    """The `u` the reference draws (DS_NeRF/run_nerf_helpers.py:312-327); a 1-D row when det."""
    if pytest:
        np.random.seed(0)
        if det:
            return torch.tensor(np.linspace(0., 1., n), dtype=torch.float32, device=device)
        return torch.tensor(np.random.rand(*shape, n), dtype=torch.float32, device=device)
    if det:
        return torch.linspace(0., 1., steps=n, device=device)
    return torch.rand(list(shape) + [n], device=device)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """DS_NeRF/run_nerf_helpers.py:304-347."""
    lead = bins.shape[:-1]
    u = _uniforms(lead, N_samples, det, pytest, bins.device)
    b2 = bins.reshape(-1, bins.shape[-1])
    w2 = weights.reshape(-1, weights.shape[-1])
    u2 = u if u.dim() == 1 else u.reshape(-1, N_samples)
    s, _, _ = ops.sample_pdf(b2, w2, u2)
    return s.reshape(*lead, N_samples)


# Compositing --------------------------------------------------------------------------------------

def _density_noise(shape, raw_noise_std, pytest, device):
    """DS_NeRF/run_nerf_helpers.py:373-381 (note: the pytest hook draws np.random.rand, uniform)."""
    if not raw_noise_std > 0.:
        return None
    if pytest:
        np.random.seed(0)
        return torch.tensor(np.random.rand(*shape) * raw_noise_std, dtype=torch.float32, device=device)
    return torch.randn(shape, device=device) * raw_noise_std


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, need_alpha=False,
                detach_weights=False):
    """DS_NeRF/run_nerf_helpers.py:350-404 -> (rgb_map, disp_map, acc_map, weights, depth_map, alpha|None)."""
    B = z_vals.shape[0]
    rows = torch.zeros((B, 6), device=z_vals.device, dtype=torch.float32)
    rows[:, 3:6] = rays_d
    noise = _density_noise(tuple(raw[..., 3].shape), raw_noise_std, pytest, raw.device)
    return ops.composite(raw, z_vals, rows, noise, white_bkgd, detach_weights, need_alpha)
