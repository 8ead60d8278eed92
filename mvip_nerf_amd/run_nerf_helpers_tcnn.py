"""Mirror of `DS_NeRF/run_nerf_helpers_tcnn.py::NeRF_TCNN` (the model the reference's shipped config selects,
`no_tcnn = False`): 16-level multiresolution hash grid (2 features, 2^19 entries, base 16) -> 32->64->16
sigma MLP; degree-4 spherical harmonics of the view direction + 15 geometry features -> 32->64->64->16 colour
MLP; output cat[colour(3), sigma] (run_nerf_helpers_tcnn.py:88-112).

The reference obtains the encodings and the bias-free "FullyFusedMLP"s from tiny-cuda-nn (NVIDIA-only, fp16
tensor-core arithmetic, not in the reference tree and not installable here), so this model is restated from
the published algorithm and is **parity unpinned** (DESIGN.md): the hash-grid gather/scatter and the SH basis
are HIP kernels (csrc/hashgrid.hip), the five small GEMMs are library fp32 matmuls on level-major [C, P]
activations with the weight gradients on csrc/skinny_gemm.hip; no-grad passes run one fused kernel
(csrc/hashgrid_fused.hip).  Parameter names follow the tiny-cuda-nn torch binding (`encoder.params`, `sigma_net.params`,
`color_net.params`, `encoder_dir.params` (empty)); matrices are [out, in] row-major, in layer order.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops

N_LEVELS, N_FEATURES, LOG2_T, BASE_RES = 16, 2, 19, 16


def level_table(bound=100, n_levels=N_LEVELS, log2_hashmap_size=LOG2_T, base_resolution=BASE_RES):
    """[n_levels, 4] int32 words {scale (fp32 bits), resolution, offset, size} and the total entry count, as
    tiny-cuda-nn's GridEncoding constructor derives them (fp32 arithmetic for scale/resolution; level sizes are
    the dense vertex count rounded up to 8 and capped at 2^log2_hashmap_size)."""
    per_level_scale = np.exp2(np.log2(2048 * bound / 16) / (16 - 1))
    log2s = np.float32(np.log2(np.float32(per_level_scale)))
    rows, offset = [], 0
    for lvl in range(n_levels):
        scale = np.float32(np.exp2(np.float32(lvl) * log2s)) * np.float32(base_resolution) - np.float32(1.0)
        res = int(np.ceil(scale)) + 1
        n = min(res ** 3, (2 ** 32 - 1) // 2)
        n = (n + 7) // 8 * 8
        n = min(n, 1 << log2_hashmap_size)
        rows.append((int(np.float32(scale).view(np.int32)), res, offset, n))
        offset += n
    return np.array(rows, dtype=np.int64).astype(np.uint32).view(np.int32).reshape(n_levels, 4), offset


class _Params(nn.Module):
    def __init__(self, values):
        super().__init__()
        self.params = nn.Parameter(values)


def _xavier(gen, out_f, in_f):
    a = (6.0 / (in_f + out_f)) ** 0.5
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * a


class NeRF_TCNN(nn.Module):
    def __init__(self, encoding="HashGrid", encoding_dir="SphericalHarmonics", num_layers=2, hidden_dim=64,
                 geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, bound=100, seed=None, **kwargs):
        super().__init__()
        if not (num_layers == 2 and hidden_dim == 64 and geo_feat_dim == 15 and num_layers_color == 3
                and hidden_dim_color == 64):
            raise NotImplementedError('NeRF_TCNN: only the reference configuration is implemented')
        self.bound = bound
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.in_dim_color = 16 + geo_feat_dim
        tab, n_entries = level_table(bound)
        self.register_buffer('levels', torch.from_numpy(tab.copy()), persistent=False)
        self.n_entries = n_entries
        gen = torch.Generator().manual_seed(0 if seed is None else seed)
        self.encoder = _Params((torch.rand(n_entries * N_FEATURES, generator=gen) * 2 - 1) * 1e-4)
        self.sigma_net = _Params(torch.cat([_xavier(gen, 64, 32).reshape(-1), _xavier(gen, 16, 64).reshape(-1)]))
        self.encoder_dir = _Params(torch.zeros(0))
        self.color_net = _Params(torch.cat([_xavier(gen, 64, 32).reshape(-1), _xavier(gen, 64, 64).reshape(-1),
                                            _xavier(gen, 16, 64).reshape(-1)]))

    def mlp_matrices(self):
        s, c = self.sigma_net.params, self.color_net.params
        return (s[:2048].view(64, 32), s[2048:3072].view(16, 64),
                c[:2048].view(64, 32), c[2048:6144].view(64, 64), c[6144:7168].view(16, 64))

    fused_inference = True          # no-grad forwards run the fused gather + MFMA kernel (csrc/hashgrid_fused.hip)
    table_grad_atomics = 'float32'  # or 'half2': tiny-cuda-nn's half-pair atomics for the scattered contributions

    def _packed_mlps(self):
        key = (self.sigma_net.params._version, self.color_net.params._version, self.sigma_net.params.data_ptr())
        if getattr(self, '_packed_key', None) != key:
            self._packed_img = ops.hashgrid_mlp_pack(self.sigma_net.params, self.color_net.params)
            self._packed_key = key
        return self._packed_img

    def forward(self, input):
        """input [N, 6] = cat[x in [-bound, bound], d in [-1, 1]] -> [N, 4] = cat[colour, sigma]."""
        x = input[:, :3].contiguous()
        d = input[:, 3:].contiguous()
        if self.fused_inference and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return ops.hashgrid_nerf_forward(x, d, self.encoder.params, self.levels, self._packed_mlps(), float(self.bound))
        W1, W2, C1, C2, C3 = self.mlp_matrices()
        feats = ops.hashgrid_encode(x, self.encoder.params, self.levels, float(self.bound),
                                    self.table_grad_atomics == 'half2')      # [32, N]
        h = ops.linear_cm(W2, ops.linear_cm(W1, feats, relu=True))                                # [16, N]
        sh = ops.sh4(d)                                                                           # [16, N]
        # the colour network's 31 inputs are padded to 32 with ones (tiny-cuda-nn pads network inputs to a
        # multiple of 16 with 1.0)
        cin = torch.cat([sh, h[1:16], torch.ones_like(h[:1])], 0)
        c = ops.linear_cm(C3, ops.linear_cm(C2, ops.linear_cm(C1, cin, relu=True), relu=True))
        return torch.stack([c[0], c[1], c[2], h[0]], -1)
