"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch, fp32) of the reference's NeRF_TCNN forward
(DS_NeRF/run_nerf_helpers_tcnn.py:88-112).  The encodings and MLPs are tiny-cuda-nn's (third-party, NVIDIA-only,
absent from the reference tree, unpinned in requirements_df.txt): restated from the published algorithm
(Mueller et al. 2022; tiny-cuda-nn include/tiny-cuda-nn/encodings/grid.h and spherical_harmonics.h).
**Parity unpinned**: there is no reference output to pin it to; the HIP kernels are checked against THIS file.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it."""
import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)
M32 = 0xFFFFFFFF


def grid_encode(x01, table, levels):
    """x01 [P,3] in [0,1]; table [n_entries, 2]; levels [16,4] int32 words -> [P, 32] (level-major, feature-minor)."""
    P = x01.shape[0]
    out = []
    lv = levels.astype(np.int64) & M32
    for scale_bits, res, off, size in lv:
        scale = np.array([scale_bits], dtype=np.uint32).view(np.float32)[0]
        pos = x01 * float(scale) + 0.5
        fl = torch.floor(pos)
        w = pos - fl
        cell = fl.to(torch.int64) & M32
        acc = torch.zeros(P, 2, dtype=torch.float32)
        for k in range(8):
            bits = [(k >> d) & 1 for d in range(3)]
            c = [(cell[:, d] + bits[d]) & M32 for d in range(3)]
            wk = torch.ones(P, dtype=torch.float32)
            for d in range(3):
                wk = wk * (w[:, d] if bits[d] else 1.0 - w[:, d])
            stride, index = 1, torch.zeros(P, dtype=torch.int64)
            for d in range(3):
                if stride <= size:
                    index = (index + c[d] * stride) & M32
                    stride = (stride * int(res)) & M32 if stride * int(res) <= M32 else stride * int(res)
            if size < stride:
                index = ((c[0] * PRIMES[0]) & M32) ^ ((c[1] * PRIMES[1]) & M32) ^ ((c[2] * PRIMES[2]) & M32)
            index = index % int(size)
            acc = acc + wk[:, None] * table[int(off) + index]
        out.append(acc)
    return torch.cat(out, -1)


def sh4(d01):
    """tiny-cuda-nn SphericalHarmonics degree 4 on inputs in [0,1]."""
    v = d01 * 2 - 1
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z,
        -0.48860251190291987 * x, 1.0925484305920792 * xy, -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2, 0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], -1)


def nerf_tcnn_forward(inp, table, levels, mats, bound=100.0):
    """inp [N,6]; mats = (W1[64,32], W2[16,64], C1[64,32], C2[64,64], C3[16,64]) -> [N,4]."""
    W1, W2, C1, C2, C3 = mats
    x = (inp[:, :3] + bound) / (2 * bound)
    f = grid_encode(x, table.reshape(-1, 2), levels)
    h = torch.relu(f @ W1.T) @ W2.T
    d = (inp[:, 3:] + 1) / 2
    cin = torch.cat([sh4(d), h[:, 1:16], torch.ones_like(h[:, :1])], -1)
    c = torch.relu(torch.relu(cin @ C1.T) @ C2.T) @ C3.T
    return torch.cat([c[:, :3], h[:, :1]], -1)
