"""N eager train_step_sd steps (full-size networks) for the PMC passes: python tools/sds_profile_steps.py [--fp16] [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd.guidance.sd_utils import StableDiffusion
dev = torch.device('cuda', 0)
n = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 2
sd = StableDiffusion(dev, '--fp16' in sys.argv, False, use_graphs=False)
g = torch.Generator(device=dev).manual_seed(2)
pred = torch.rand(1, 3, 378, 504, device=dev, generator=g).requires_grad_(True)
mask = torch.zeros(1, 1, 378, 504, device=dev)
mask[:, :, 137:241, 196:307] = 1
for i in range(n):
    pred.grad = None
    (1e-4 * sd.train_step_sd(1000 + i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
torch.cuda.synchronize()
print('done', n)
