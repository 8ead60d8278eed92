"""Tiny stand-in diffusion networks + pipeline shim (TEST INFRASTRUCTURE ONLY).

The reference's SDS wrapper (DS_NeRF/guidance/sd_utils.py) calls into `diffusers`, which is neither in
the reference tree nor installed here, and no Stable-Diffusion weights exist offline.  To pin the
WRAPPER ARITHMETIC (resize, masking, VAE-sample, add_noise, CFG, w(t), nan_to_num, SpecifyGradient,
the colla quirks) we attach these small deterministic networks to the reference's `StableDiffusion`
object (oracle/gen_golden_sds.py) and to ours (tests/test_sds.py) and compare.  The network BODIES of
the real SD-1.5-inpaint model remain "parity unpinned" (DESIGN.md).

`FakePipe` restates what the wrapper needs from diffusers' StableDiffusionInpaintPipeline, following
the vendored scratch copy DS_NeRF/guidance/pipeline_sd_inpainting.py:631-748 (prepare_latents,
_encode_vae_image, prepare_mask_latents).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _seeded(shape, seed, scale):
    return torch.from_numpy((np.random.RandomState(seed).normal(size=shape) * scale).astype(np.float32))


class DiagGaussian:
    """diffusers' DiagonalGaussianDistribution: moments [B,8,h,w] -> mean/logvar halves, logvar clamped."""

    def __init__(self, moments, randn):
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self._randn = randn

    def sample(self, generator=None):
        return self.mean + self.std * self._randn(self.mean.shape)


class EncOut:
    def __init__(self, dist):
        self.latent_dist = dist


class TinyVAE(nn.Module):
    """[B,3,512,512] -> moments [B,8,64,64] (two strided convs) and a decoder stub."""

    class Cfg:
        scaling_factor = 0.18215

    def __init__(self, seed=5):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 4, stride=4)
        self.c2 = nn.Conv2d(8, 8, 2, stride=2)
        self.d = nn.Conv2d(4, 3, 1)
        self.config = self.Cfg()
        with torch.no_grad():
            for k, p in enumerate(self.parameters()):
                p.copy_(_seeded(tuple(p.shape), seed + k, 0.3 / np.sqrt(max(p[0].numel(), 1))))
        self.randn = torch.randn          # replaced by tests to replay recorded draws

    @staticmethod
    def _patch_conv(conv, x):
        """conv (kernel == stride, no padding) as pixel_unshuffle + channel contraction: the same arithmetic without
        the convolution library.  Used on the GPU, where MIOpen's backward-data solver for these unusual stride-4 /
        stride-2 stand-in layers faulted once the library's tuning database had seen the full-size VAE."""
        r = conv.stride[0]
        w = conv.weight.reshape(conv.out_channels, -1)
        return torch.einsum('ok,bkhw->bohw', w, F.pixel_unshuffle(x, r)) + conv.bias[None, :, None, None]

    def encode(self, x):
        if x.is_cuda:
            h = self._patch_conv(self.c2, torch.tanh(self._patch_conv(self.c1, x)))
        else:
            h = self.c2(torch.tanh(self.c1(x)))
        return EncOut(DiagGaussian(h, lambda s: self.randn(s)))

    def decode(self, z, return_dict=False):
        return (F.interpolate(self.d(z), scale_factor=8),)


class TinyUNet(nn.Module):
    """[B,9,64,64], t, text [B,77,768] -> [B,4,64,64]: conv + time and text conditioning."""

    def __init__(self, seed=40):
        super().__init__()
        self.c1 = nn.Conv2d(9, 16, 3, padding=1)
        self.c2 = nn.Conv2d(16, 4, 3, padding=1)
        self.t = nn.Linear(1, 16)
        self.txt = nn.Linear(768, 16)
        with torch.no_grad():
            for k, p in enumerate(self.parameters()):
                p.copy_(_seeded(tuple(p.shape), seed + k, 0.5 / np.sqrt(max(p[0].numel(), 1))))

    def forward(self, x, t, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=False):
        tt = torch.as_tensor(t, dtype=torch.float32, device=x.device).reshape(1, 1) / 1000.0
        h = self.c1(x) + self.t(tt)[:, :, None, None] + self.txt(encoder_hidden_states.mean(1))[:, :, None, None]
        return (self.c2(torch.tanh(h)),)


class TinyScheduler:
    """scaled_linear betas 0.00085 -> 0.012 over 1000 steps (SD v1), PNDM-style add_noise,
    identity scale_model_input."""

    def __init__(self):
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.init_noise_sigma = 1.0
        self.order = 1

        class C:
            num_train_timesteps = 1000
        self.config = C()

    def scale_model_input(self, x, t):
        return x

    def add_noise(self, x0, noise, t):
        a = self.alphas_cumprod.to(x0.device)[t]
        return (a ** 0.5) * x0 + ((1 - a) ** 0.5) * noise


def prompt_embedding(prompt, cfg):
    """Deterministic stand-in for CLIP: [2,77,768] (uncond first) or [1,77,768]."""
    seed = sum(ord(c) * (k + 1) for k, c in enumerate(prompt)) % 100003
    cond = _seeded((1, 77, 768), 1000 + seed, 1.0)
    if not cfg:
        return cond
    return torch.cat([_seeded((1, 77, 768), 999, 1.0), cond], 0)


class FakePipe:
    """What sd_utils.py uses of the inpaint pipeline (pipeline_sd_inpainting.py:631-748)."""
    vae_scale_factor = 8

    def __init__(self, vae, scheduler, randn):
        self.vae, self.scheduler, self.randn = vae, scheduler, randn
        self.unet = None

        class IP:
            @staticmethod
            def postprocess(image, output_type='pil', do_denormalize=None):
                return image
        self.image_processor = IP()

    def check_inputs(self, *a, **k):
        return None

    def _encode_prompt(self, prompt, device, n, cfg, *a, **k):
        return prompt_embedding(prompt, cfg).to(device)

    def prepare_extra_step_kwargs(self, generator, eta):
        return {}

    def run_safety_checker(self, image, device, dtype):
        return image, None

    def _encode_vae_image(self, image, generator=None):
        return self.vae.config.scaling_factor * self.vae.encode(image).latent_dist.sample(generator=generator)

    def prepare_mask_latents(self, mask, masked_image, batch_size, height, width, dtype, device, generator, cfg):
        mask = F.interpolate(mask, size=(height // 8, width // 8))
        mask = mask.to(device=device, dtype=dtype)
        masked_image_latents = self._encode_vae_image(masked_image.to(device=device, dtype=dtype), generator)
        mask = torch.cat([mask] * 2) if cfg else mask
        masked_image_latents = torch.cat([masked_image_latents] * 2) if cfg else masked_image_latents
        return mask, masked_image_latents

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None,
                        image=None, timestep=None, is_strength_max=True, return_noise=False,
                        return_image_latents=False):
        shape = (batch_size, num_channels_latents, height // 8, width // 8)
        image_latents = self._encode_vae_image(image.to(device=device, dtype=dtype), generator)
        noise = self.randn(shape)
        latents = self.scheduler.add_noise(image_latents, noise, timestep)
        out = (latents,)
        if return_noise:
            out += (noise,)
        if return_image_latents:
            out += (image_latents,)
        return out

    def get_timesteps(self, n, strength, device):
        return torch.arange(999, -1, -1)[int(n * (1 - strength)):], n
