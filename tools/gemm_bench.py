"""Timing of the split-precision GEMM's workgroup-tile variants on the UNet transformer's linear-layer shapes
(channel-major: M = output features, K = input features, P = tokens of both CFG samples)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops                                 # noqa: E402


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device('cuda', 0)
    shapes = []
    for C, L in ((320, 4096), (640, 1024), (1280, 256)):
        R = 8 * ((C // 8 + 15) // 16 * 16)
        shapes += [('qkv', 3 * R, C, L), ('out', C, C, L), ('q2', R, C, L), ('ff1', 8 * C, C, L), ('ff2', C, 4 * C, L)]
    out = {}
    for name, M, K, L in shapes:
        Nb = 2
        x = torch.randn(Nb, K, L, device=dev)
        W = torch.randn(M, K, device=dev) / K ** 0.5
        xs, s2 = ops._scaled_planes(x, Nb, K, L, K * L, L, 1)
        pk = ops.gemm_pack_a(W, M, K, K, 1)
        row = {}
        for cfg in (1, 2, 3, 4, 5, 0):
            if (cfg in (2, 3) and M % 128) or (cfg == 4 and M % 64):
                continue
            ops.GEMM_CFG = cfg
            ms = timed(lambda: ops.gemm_f16x3(xs, pk, Nb, K, M, L, x_scale2=s2))
            row[f'cfg{cfg}'] = round(ms * 1e3, 1)
        ops.GEMM_CFG = 0
        ms = timed(lambda: torch.matmul(W, x))
        row['torch_fp32'] = round(ms * 1e3, 1)
        row['GFLOP'] = round(2.0 * Nb * M * K * L / 1e9, 2)
        out[f'{name}_M{M}_K{K}_L{L}'] = row
        print(name, M, K, L, row, flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/gemm_bench.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
