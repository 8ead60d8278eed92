"""Timing-only experiment on the 32/64-row GEMM kernel (build with MVIP_EXTRA_FLAGS=-DMVIP_EXPERIMENT_GEMM):
full kernel vs no epilogue stores vs no MFMAs, on three UNet shapes."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops
from tools.gemm_bench import timed
dev = torch.device('cuda', 0)
for name, M, K, L in (('qkv', 1152, 320, 4096), ('out', 320, 320, 4096), ('ff2', 320, 1280, 4096), ('ff1_l3', 10240, 1280, 256)):
    x = torch.randn(2, K, L, device=dev); W = torch.randn(M, K, device=dev) / K ** 0.5
    xs, s2 = ops._scaled_planes(x, 2, K, L, K * L, L, 1); pk = ops.gemm_pack_a(W, M, K, K, 1)
    row = {}
    for tag, dbg in (('full', 0), ('no_store', 1), ('no_mfma', 2), ('no_mfma_no_store', 3)):
        ops.GEMM_CFG = 1 | (dbg << 8)
        row[tag] = round(timed(lambda: ops.gemm_f16x3(xs, pk, 2, K, M, L, x_scale2=s2)) * 1e3, 1)
    print(name, M, K, L, row, flush=True)
