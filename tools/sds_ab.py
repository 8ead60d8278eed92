"""Same-box A/B of the SDS step: 1x1 shortcut convolutions on the split-precision GEMM or on the library, absmax as
one launch or three, fixed unit scale for the UNet's forward activations.  Median of 15 steps each, one process."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops
from mvip_nerf_amd.guidance import sd_nets
from mvip_nerf_amd.guidance.sd_utils import StableDiffusion
dev = torch.device('cuda', 0)
sd = StableDiffusion(dev, False, False)
g = torch.Generator(device=dev).manual_seed(2)
pred = torch.rand(1, 3, 378, 504, device=dev, generator=g).requires_grad_(True)
mask = torch.zeros(1, 1, 378, 504, device=dev); mask[:, :, 137:241, 196:307] = 1
def step(i):
    pred.grad = None
    (1e-4 * sd.train_step_sd(i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
def med(n=15):
    for k in range(3): step(k)
    torch.cuda.synchronize(); ts = []
    for k in range(n):
        t0 = time.perf_counter(); step(100 + k); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(sorted(ts)[n // 2], 2)
out = {}
for rep in range(2):
    for c1 in (True, False):
        for three in (False, True):
            for unit in (False, True):
                sd_nets.USE_MFMA_CONV1X1, ops.ABSMAX_THREE_LAUNCHES, ops.FORWARD_UNIT_SCALE = c1, three, unit
                out.setdefault(f'conv1x1={int(c1)} absmax3={int(three)} unit={int(unit)}', []).append(med())
print(json.dumps(out, indent=1))
