"""Which call sites launch the per-tensor absmax / plane-writer / reduce passes of one SDS step (eager), with sizes."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops                                            # noqa: E402
from mvip_nerf_amd.guidance.sd_utils import StableDiffusion          # noqa: E402

WATCH = ('mvip_absmax_scale', 'mvip_split_planes', 'mvip_split_planes_strided', 'mvip_im2col_split_planes',
         'mvip_groupnorm_stats', 'mvip_groupnorm_split_planes', 'mvip_col2im', 'mvip_groupnorm_backward',
         'mvip_layernorm_split_planes', 'mvip_absmax_scale_sections')
tally = collections.Counter()
on = [False]
orig = ops.call


def call(name, *a):
    if on[0] and name in WATCH:
        st = traceback.extract_stack(limit=7)[:-1]
        site = ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in reversed(st[-4:]))
        tally[(name, site)] += 1
    return orig(name, *a)


ops.call = call


def main():
    dev = torch.device('cuda', 0)
    sd = StableDiffusion(dev, False, False, use_graphs=False)
    g = torch.Generator(device=dev).manual_seed(2)
    pred = torch.rand(1, 3, 378, 504, device=dev, generator=g).requires_grad_(True)
    mask = torch.zeros(1, 1, 378, 504, device=dev)
    mask[:, :, 137:241, 196:307] = 1

    def step(i):
        pred.grad = None
        (1e-4 * sd.train_step_sd(i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
    step(0)
    on[0] = True
    step(1)
    torch.cuda.synchronize()
    for (name, site), n in sorted(tally.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print(f'{n:4d}  {name:32s} {site}')


if __name__ == '__main__':
    main()
