"""Algorithmic FLOPs of one SDS step, counted from the layer shapes of `sd_nets` themselves (SURVEY.md 8(d)
"Roofline, SDS": count from the build's own layer shapes, not from literature numbers).

The modules are instantiated on the `meta` device (no memory, no arithmetic) and run once with forward hooks:
  Conv2d : 2 * Cout * (Cin / groups) * kh * kw * Hout * Wout * N
  Linear : 2 * in * out * rows
  attention products (q k^T and p v): 4 * N * heads * Lq * Lk * d   (hooked on the attention modules' inputs)
Normalisations, activations and the elementwise wrapper arithmetic are not counted (they are not contraction work;
the roofline this feeds is the matrix-pipe one).

One `train_step_sd` (guidance/sd_utils.py; DS_NeRF/guidance/sd_utils.py:275-429 minus its unused encode / decode):
    2 x VAE-encoder forward (masked image, image)  +  1 x VAE-encoder data-gradient pass (same MACs as a forward:
    the networks are frozen, so no weight gradients)  +  1 x UNet forward at batch 2 (CFG).
"""
import torch
import torch.nn as nn


def _count(module, inputs_fn):
    from . import sd_nets
    total = {'conv': 0, 'linear': 0, 'attention': 0, 'bytes': 0}

    def conv_hook(m, inp, out):
        n, co, ho, wo = out.shape
        total['conv'] += 2 * co * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1] * ho * wo * n
        total['bytes'] += 4 * (m.weight.numel() + inp[0].numel() + out.numel())        # fp32: weights + input + output, once each

    def lin_hook(m, inp, out):
        total['linear'] += 2 * m.in_features * m.out_features * (out.numel() // m.out_features)
        total['bytes'] += 4 * (m.weight.numel() + inp[0].numel() + out.numel())

    def attn_hook(m, inp, kwargs, out):
        x = inp[0]
        ctx = inp[1] if len(inp) > 1 and inp[1] is not None else kwargs.get('ctx')
        ctx = x if ctx is None else ctx
        total['attention'] += 4 * x.shape[0] * x.shape[1] * ctx.shape[1] * x.shape[2]

    def vae_attn_hook(m, inp, out):
        n, c, h, w = inp[0].shape
        total['attention'] += 4 * n * (h * w) ** 2 * c

    hs = []
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            hs.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, nn.Linear):
            hs.append(m.register_forward_hook(lin_hook))
        elif isinstance(m, sd_nets.Attention):
            hs.append(m.register_forward_hook(attn_hook, with_kwargs=True))
        elif isinstance(m, sd_nets.VAEAttention):
            hs.append(m.register_forward_hook(vae_attn_hook))
    with torch.no_grad():
        inputs_fn(module)
    for h in hs:
        h.remove()
    return total


def sds_step_flops(image_size=512, batch_unet=2, ctx_tokens=77):
    """{'unet_forward', 'vae_encoder_forward', 'per_step', 'breakdown'} in FLOPs for an SDS step at `image_size`^2."""
    from . import sd_nets
    with torch.device('meta'):
        unet = sd_nets.UNet2DConditionModel()
        enc = sd_nets.Encoder()
        quant = nn.Conv2d(8, 8, 1)
    lat = image_size // 8
    u = _count(unet, lambda m: m(torch.empty(batch_unet, 9, lat, lat, device='meta'), torch.zeros(1, device='meta'),
                                 encoder_hidden_states=torch.empty(batch_unet, ctx_tokens, 768, device='meta')))
    e = _count(nn.Sequential(enc, quant), lambda m: m(torch.empty(1, 3, image_size, image_size, device='meta')))
    ub, eb = u.pop('bytes'), e.pop('bytes')
    uf, ef = sum(u.values()), sum(e.values())
    return {'unet_forward': uf, 'vae_encoder_forward': ef, 'per_step': uf + 3 * ef,
            'breakdown': {'unet': u, 'vae_encoder': e},
            # algorithmic HBM bytes: every convolution / linear layer reads its fp32 weights and input and writes its output
            # once (normalisations, attention internals and elementwise glue not counted)
            'bytes_per_step': ub + 3 * eb, 'unet_forward_bytes': ub, 'vae_encoder_forward_bytes': eb,
            'composition': '1 x UNet forward (batch 2) + 2 x VAE-encoder forward + 1 x VAE-encoder data-gradient pass'}
