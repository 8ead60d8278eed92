"""Golden vectors for the SDS wrapper arithmetic, produced by the REAL reference
`DS_NeRF/guidance/sd_utils.py::StableDiffusion.train_step_sd / _sd_normal / _colla_sds` running on
CPU with the tiny stand-in networks of oracle/sds_standin.py attached.

    python oracle/gen_golden_sds.py        # build container only; writes tests/golden/sds_*.npz

Randomness: every draw of the reference goes through torch.randn on the CPU default generator
after torch.manual_seed(seed); the fixtures store the seed, and tests replay the same CPU draws.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')


def main():
    from oracle.gen_golden import import_reference
    import_reference()
    from guidance import sd_utils
    from oracle.sds_standin import TinyVAE, TinyUNet, TinyScheduler, FakePipe
    torch.set_num_threads(8)

    sd = sd_utils.StableDiffusion.__new__(sd_utils.StableDiffusion)
    torch.nn.Module.__init__(sd)
    vae, unet, sched = TinyVAE(), TinyUNet(), TinyScheduler()
    pipe = FakePipe(vae, sched, lambda s: torch.randn(s))
    sd.device = torch.device('cpu')
    sd.vae, sd.unet, sd.scheduler, sd.pipe = vae, unet, sched, pipe
    sd.strength = 0.75
    sd.timesteps = torch.arange(999, -1, -1)
    sd.min_step, sd.max_step = 20, 980

    captured = {}
    orig_apply = sd_utils.SpecifyGradient.apply

    def spy(latents, grad, mask):
        captured['latents'] = latents.detach().clone()
        captured['grad'] = grad.detach().clone()
        captured['mask'] = mask.detach().clone()
        return orig_apply(latents, grad, mask)
    sd_utils.SpecifyGradient.apply = staticmethod(spy)

    rs = np.random.RandomState(77)
    H, W = 48, 64
    yy, xx = np.mgrid[0:H, 0:W]
    mask_np = ((yy > 14) & (yy < 34) & (xx > 20) & (xx < 46)).astype(np.float32)[None, None]

    def save(name, **kw):
        path = os.path.join(OUT, name + '.npz')
        np.savez_compressed(path, **kw)
        print(f'{name}: {os.path.getsize(path) / 1024:.1f} KB')

    # ---- rgb SDS ----
    for i in (0, 100, 5000, 20000):
        seed = 900 + i
        pred = torch.from_numpy(rs.uniform(0, 1, size=(1, 3, H, W)).astype(np.float32)).requires_grad_(True)
        torch.manual_seed(seed)
        loss = sd.train_step_sd(i, torch.from_numpy(mask_np), 'a stone bench in a park', pred, guidance_scale=7.5,
                                as_latent=True, grad_scale=1)
        (1e-4 * loss).sum().backward()
        save(f'sds_rgb_i{i}', i=i, seed=seed, pred=pred.detach().numpy(), mask=mask_np, guidance_scale=7.5,
             loss=loss.detach().numpy(), latents=captured['latents'].numpy(), grad=captured['grad'].numpy(),
             mask64=captured['mask'].numpy(), d_pred=pred.grad.numpy(), upstream=1e-4)

    # ---- normal SDS ----
    i, ns, seed = 1700, 500, 4242
    pred = torch.from_numpy(rs.uniform(0, 1, size=(1, 3, 27, 36)).astype(np.float32)).requires_grad_(True)
    torch.manual_seed(seed)
    loss = sd.train_step_sd_normal(i, torch.from_numpy(mask_np), 'a normal map of a stone bench', pred,
                                   guidance_scale=1.5, normal_start=ns, as_latent=True, grad_scale=1)
    (1e-4 * loss).sum().backward()
    save('sds_normal', i=i, normal_start=ns, seed=seed, pred=pred.detach().numpy(), mask=mask_np, guidance_scale=1.5,
         loss=loss.detach().numpy(), latents=captured['latents'].numpy(), grad=captured['grad'].numpy(),
         mask64=captured['mask'].numpy(), d_pred=pred.grad.numpy(), upstream=1e-4)

    # ---- collaborative SDS over 3 neighbour views (reproduces the reference's quirks) ----
    seed = 777
    NN = 3
    preds = torch.from_numpy(rs.uniform(0, 1, size=(NN, 3, 27, 36)).astype(np.float32)).requires_grad_(True)
    masks = np.repeat(mask_np, NN, axis=0)
    masks[1] = np.roll(masks[1], 5, axis=-1)
    torch.manual_seed(seed)
    loss = sd.train_step_colla_sds(1234, torch.from_numpy(masks), 'a stone bench in a park', preds,
                                   guidance_scale=7.5, as_latent=True, grad_scale=1)
    (1e-4 * loss).sum().backward()
    save('sds_colla', seed=seed, preds=preds.detach().numpy(), masks=masks, guidance_scale=7.5,
         loss=loss.detach().numpy(), latents=captured['latents'].numpy(), grad=captured['grad'].numpy(),
         mask64=captured['mask'].numpy(), d_preds=preds.grad.numpy(), upstream=1e-4)


if __name__ == '__main__':
    main()
