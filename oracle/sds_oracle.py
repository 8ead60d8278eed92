"""CPU oracle of the SDS step's WRAPPER arithmetic (TEST INFRASTRUCTURE ONLY; nothing in `mvip_nerf_amd/` may import
it): a plain-torch restatement of `StableDiffusion.train_step_sd / train_step_sd_normal`
(DS_NeRF/guidance/sd_utils.py:275-429, :120-272) around ANY network bundle (vae.encode -> latent_dist with
mean / std, unet(x, t, encoder_hidden_states) -> (eps,), encode_prompt(prompt, cfg), alphas_cumprod).

Parity status: PINNED for the wrapper -- tests/test_oracle_golden.py compares it with the fixtures the reference's
own wrapper produced on the stand-in networks (tests/golden/sds_*.npz).  The SD-1.5 network BODIES stay parity
unpinned (diffusers is not in the reference tree).  bench.py's `cpu_baseline` times this function with
the SD-1.5-shaped modules of `mvip_nerf_amd/guidance/sd_nets.py` on the host cores.

Steps, as the reference performs them (useless work of the reference -- the unused encode, the decode, the PNG --
is left out exactly as in the product path):
  pred, |mask| -> bilinear 512^2 (align_corners=False); masked = pred * (mask < 0.5); mask64 = nearest 64^2;
  masked latents = 0.18215 * sample(vae.encode(masked)); image latents likewise (carries grad);
  t = int(980 - 960 * sqrt(i / 20000)); noisy = sqrt(abar_t) z + sqrt(1 - abar_t) eps;
  eps_hat = unet(cat[noisy x2, mask64 x2, masked x2]) -> e_u + s (e_c - e_u);
  grad = nan_to_num((1 - abar_t) (eps_hat - eps));  loss = SpecifyGradient(latents, grad, mask64[0]).
"""
import numpy as np
import torch
import torch.nn.functional as F


class SpecifyGradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gt_grad, mask):
        ctx.save_for_backward(gt_grad, mask)
        return torch.ones([1], device=x.device, dtype=x.dtype)

    @staticmethod
    def backward(ctx, g):
        gt_grad, mask = ctx.saved_tensors
        return gt_grad * g * mask, None, None


def train_step_sd(nets, i, mask, prompt, pred_rgb, guidance_scale=100., randn=torch.randn, frac=None, size=512,
                  reference_rng=True, min_step=20, max_step=980):
    """Returns the [1] loss whose backward injects the SDS gradient into pred_rgb.  `frac` overrides sqrt(i/20000)
    (the normal variant passes sqrt((i - normal_start)/20000)); `size` = 512 in the reference."""
    pred = F.interpolate(pred_rgb, (size, size), mode='bilinear', align_corners=False)
    m = F.interpolate(torch.abs(mask), (size, size), mode='bilinear', align_corners=False)
    cfg = guidance_scale > 1.0
    emb = nets.encode_prompt(prompt, cfg)
    sf = 0.18215

    def encode(img):
        d = nets.vae.encode(img).latent_dist
        return sf * (d.mean + d.std * randn(d.mean.shape))
    masked_latents = encode(pred[:, :3] * (m < 0.5))
    mask64 = F.interpolate(m, size=(size // 8, size // 8))
    if cfg:
        mask64, masked_latents = torch.cat([mask64] * 2), torch.cat([masked_latents] * 2)
    if reference_rng:
        randn((1, 4, size // 8, size // 8))
    t = int(max_step - (max_step - min_step) * (np.sqrt(i / 20000) if frac is None else frac))
    abar = float(nets.alphas_cumprod[t])
    z = encode(pred[:, :3])
    noise = randn(z.shape)
    latents = abar ** 0.5 * z + (1.0 - abar) ** 0.5 * noise
    with torch.no_grad():
        x = torch.cat([latents] * 2) if cfg else latents
        eps = nets.unet(torch.cat([x, mask64, masked_latents], 1), t, encoder_hidden_states=emb)[0]
        if cfg:
            e_u, e_c = eps.chunk(2)
            eps = e_u + guidance_scale * (e_c - e_u)
        grad = torch.nan_to_num((1.0 - abar) * (eps - noise))
    return SpecifyGradient.apply(latents, grad, mask64[0])


def train_step_sd_normal(nets, i, mask, prompt, pred_normal, guidance_scale=100., normal_start=0, **kw):
    return train_step_sd(nets, i, mask, prompt, pred_normal, guidance_scale, frac=np.sqrt((i - normal_start) / 20000), **kw)
