"""Timing of the UNet transformer kernels on the GPU box: the flash attention at the three UNet levels (both tile
variants for 40-channel heads), and the whole UNet forward with the HIP transformer path on / off."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops                                 # noqa: E402
from mvip_nerf_amd.guidance import sd_nets                    # noqa: E402


def timed(fn, reps=10):
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
    out = {}
    for heads, D, L, Lk in ((8, 40, 4096, 4096), (8, 80, 1024, 1024), (8, 160, 256, 256), (8, 40, 4096, 77)):
        Nb, DP = 2, (D + 15) // 16 * 16
        R = heads * DP
        LkP = (Lk + 63) // 64 * 64
        q = torch.randn(Nb, R, L, device=dev)
        k = torch.randn(Nb, R, LkP, device=dev)
        v = torch.randn(Nb, R, LkP, device=dev)
        sq, sk, sv = (ops.absmax_scale(t) for t in (q, k, v))
        qs = ops.split_planes_strided(q, Nb, R, L, R * L, L, 1, sq)
        ks = ops.split_planes_strided(k, Nb, R, LkP, R * LkP, LkP, 1, sk)
        vp = ops.attention_pack_v(v, Nb, heads, D, DP, Lk, LkP, R * LkP, LkP, 1, sv)
        o = torch.empty(Nb, heads * D, L, device=dev)
        flops = 4.0 * Nb * heads * L * Lk * D
        for flags in ((0, 2, 3) if D == 40 else (0,)):
            ops.ATTENTION_FLAGS = flags
            ms = timed(lambda: ops.attention_f16x3(qs, ks, vp, sq, sk, sv, Nb, heads, D, L, L, Lk, LkP, out=o))
            out[f'attn_D{D}_L{L}_Lk{Lk}_flags{flags}'] = {'ms': round(ms, 4), 'TFLOPs_fp32_equiv': round(flops / ms / 1e9, 1)}
        ops.ATTENTION_FLAGS = 0
        if Lk == L:
            qq = q.view(Nb, heads, DP, L).transpose(2, 3).contiguous()
            kk = k.view(Nb, heads, DP, L).transpose(2, 3).contiguous()
            vv = v.view(Nb, heads, DP, L).transpose(2, 3).contiguous()
            ms = timed(lambda: torch.nn.functional.scaled_dot_product_attention(qq, kk, vv))
            out[f'attn_D{D}_L{L}_library_sdpa_fp32_padded_heads'] = round(ms, 4)
    torch.manual_seed(0)
    unet = sd_nets.UNet2DConditionModel().to(dev).eval()
    for p in unet.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 9, 64, 64, device=dev)
    ctx = torch.randn(2, 77, 768, device=dev)
    t = torch.tensor(417, device=dev)
    with torch.no_grad():
        for name, tr, tl, c1 in (('library', False, False, False), ('hip_transformer', True, True, False),
                                 ('hip_transformer+conv1x1', True, True, True)):
            sd_nets.USE_HIP_TRANSFORMER, sd_nets.USE_HIP_TIME_LINEARS, sd_nets.USE_MFMA_CONV1X1 = tr, tl, c1
            for _ in range(2):
                unet(x, t, encoder_hidden_states=ctx)
            out[f'unet_forward_ms_{name}'] = round(timed(lambda: unet(x, t, encoder_hidden_states=ctx), 5), 3)
    print(json.dumps(out, indent=1))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/attn_bench.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
