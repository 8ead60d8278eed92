"""Time one SDS step (train_step_sd forward + backward to pred_rgb.grad) at SD-1.5-inpaint shapes."""
import sys, time, json
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd.guidance.sd_utils import StableDiffusion

GRAPHS = False
CHANNELS_LAST = False


def run(fp16, steps=5):
    dev = torch.device('cuda', 0)
    sd = StableDiffusion(dev, fp16, False, use_graphs=GRAPHS)
    if CHANNELS_LAST:
        sd.vae.to(memory_format=torch.channels_last)
        sd.unet.to(memory_format=torch.channels_last)
    g = torch.Generator(device=dev).manual_seed(2)
    pred = torch.rand(1, 3, 378, 504, device=dev, generator=g).requires_grad_(True)
    mask = torch.zeros(1, 1, 378, 504, device=dev); mask[:, :, 137:241, 196:307] = 1
    def step(i):
        pred.grad = None
        loss = sd.train_step_sd(i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)
        ((1.0 if fp16 else 1e-4) * loss).sum().backward()
    step(1000); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps): step(1000 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {'fp16': fp16, 'ms_per_step': dt * 1e3, 'steps_per_sec': 1 / dt, 'grad_abs_max': float(pred.grad.abs().max()),
            'mem_GB': torch.cuda.max_memory_allocated() / 1e9}

if __name__ == '__main__':
    from mvip_nerf_amd.guidance import sd_nets
    for c3, c1, at in ((True, True, True), (True, False, True), (True, False, False), (False, False, False), (True, True, True)):
        sd_nets.USE_MFMA_CONV3X3, sd_nets.USE_MFMA_CONV1X1, sd_nets.USE_MFMA_VAE_ATTENTION = c3, c1, at
        r = run(False)
        r.update(conv3x3=c3, conv1x1=c1, vae_attention=at)
        print(json.dumps(r), flush=True)
        torch.cuda.empty_cache()
    sd_nets.USE_MFMA_CONV3X3 = sd_nets.USE_MFMA_CONV1X1 = sd_nets.USE_MFMA_VAE_ATTENTION = True
    print(json.dumps(run(True)), flush=True)
