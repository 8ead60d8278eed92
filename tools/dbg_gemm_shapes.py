"""Per-shape device time of every GEMM-family launch of one SDS step (eager): events around each C-ABI call."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops                                            # noqa: E402
from mvip_nerf_amd.guidance.sd_utils import StableDiffusion          # noqa: E402

# name -> positions of (N, K, M, P) in the argument list
POS = {'mvip_gemm_f16x3_ws': (6, 7, 8, 9), 'mvip_gemm_f16x3_cfg': (6, 7, 8, 9), 'mvip_gemm_f16x3_sinks': (4, 5, 6, 7),
       'mvip_gemm_geglu_f16x3_sink': (4, 5, 6, 7), 'mvip_gemm_geglu_f16x3': None, 'mvip_gemm_f16x3_planes_ws': None,
       'mvip_conv3x3_f16x3_ws': (6, 7, 8, 9, 10), 'mvip_attention_f16x3_sink': (6, 7, 8, 9, 11), 'mvip_attention_f16x3': None}
recs = []
on = [False]
orig = ops.call


def call(name, *a):
    if on[0] and name in POS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(name, *a)
        e1.record()
        pos = POS[name]
        shape = tuple(int(a[i]) for i in pos) if pos else tuple(int(x) for x in a if isinstance(x, int) and 0 < x < 1 << 20)[:5]
        recs.append((name, shape, e0, e1))
        return r
    return orig(name, *a)


ops.call = call


def main():
    dev = torch.device('cuda', 0)
    sd = StableDiffusion(dev, '--fp16' in sys.argv, False, use_graphs=False)
    g = torch.Generator(device=dev).manual_seed(2)
    pred = torch.rand(1, 3, 378, 504, device=dev, generator=g).requires_grad_(True)
    mask = torch.zeros(1, 1, 378, 504, device=dev)
    mask[:, :, 137:241, 196:307] = 1

    def step(i):
        pred.grad = None
        (1e-4 * sd.train_step_sd(i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
    step(0)
    step(1)
    tally = collections.defaultdict(lambda: [0, 0.0])
    for rep in range(3):
        recs.clear()
        on[0] = True
        step(2 + rep)
        on[0] = False
        torch.cuda.synchronize()
        for name, shape, e0, e1 in recs:
            t = tally[(name, shape)]
            t[0] += 1
            t[1] += e0.elapsed_time(e1)
    tot = collections.Counter()
    for (name, shape), (n, ms) in sorted(tally.items(), key=lambda kv: -kv[1][1]):
        n3, ms3 = n / 3, ms / 3
        fl = ''
        if name.startswith('mvip_gemm') and len(shape) == 4:
            N, K, M, P = shape
            fl = f'{2 * N * K * M * P / (ms3 / n3 * 1e-3) / 1e12:7.1f} TF'
        if name.startswith('mvip_conv3x3'):
            N, Ci, Co, H, W = shape
            fl = f'{2 * N * Ci * Co * 9 * H * W / (ms3 / n3 * 1e-3) / 1e12:7.1f} TF'
        print(f'{ms3:7.3f} ms x{n3:5.1f} {1e3 * ms3 / n3:8.1f} us  {name:28s} {shape} {fl}')
        tot[name] += ms3
    print(dict(tot))


if __name__ == '__main__':
    main()
