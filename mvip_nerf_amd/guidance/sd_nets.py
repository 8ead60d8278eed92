"""Stable-Diffusion-1.5-inpainting-shaped networks (UNet2DCondition 9->4, AutoencoderKL, CLIP ViT-L/14
text tower, scaled-linear scheduler constants) written from the published architecture
(SURVEY.md Appendix A.8), on PyTorch-ROCm.

The reference obtains these from `diffusers` / `transformers` with the
`runwayml/stable-diffusion-inpainting@fp16` weights (DS_NeRF/guidance/sd_utils.py:69-74).  Neither the
library source nor the weights exist offline, so: the module/parameter names follow diffusers'
state-dict layout (a real checkpoint can be loaded with `load_state_dict` once available), the
weights are randomly initialised unless a checkpoint DIRECTORY is given (`SDNetworks.load_checkpoint`,
guidance/sd_checkpoint.py: strict manifest of names and shapes), and the network bodies are "parity
unpinned" (DESIGN.md).  They exist so the SDS step can be executed and timed at its true shapes and FLOPs.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# Which layers take the hand-written split-precision kernels (csrc/conv3x3.hip) instead of the library ones;
# switches exist for A/B timing (tools/sds_bench.py), every combination computes the same function.
USE_MFMA_CONV3X3 = True
USE_MFMA_CONV1X1 = True       # 1x1 shortcut convolutions on the split-precision GEMM: +0.2 ms per UNet forward against the
                              # library's GEMM-based 1x1 path (tools/attn_bench.py), kept on so that no library GEMM runs in the step
USE_MFMA_VAE_ATTENTION = True
USE_HIP_TRANSFORMER = True    # UNet transformer blocks on csrc/attention.hip + csrc/transformer.hip (guidance/transformer_cm.py)
USE_HIP_TIME_LINEARS = True   # timestep-embedding MLP and the ResNet blocks' time projections on mvip_linear_small
USE_GEMM_CONV = True          # stride-2 down-samplers, conv_in / conv_out, quant_conv: im2col planes + the split-precision GEMM
                              # (ops.conv_gemm) instead of the library convolution -- with it no library contraction is left in the step


def conv_any(conv, x, pads=None):
    """conv(x) for the layers outside the 3x3 stride-1 kernel's shapes.  pads = (top, left, bottom, right) replaces the
    module's own padding (the VAE down-samplers pad bottom / right by one).  fp32 device tensors run ops.conv_gemm,
    anything else (fp16 mode, host tensors) the library convolution."""
    if USE_GEMM_CONV and x.is_cuda:
        from .. import ops
        if ops.conv_gemm_supported(conv, x):
            return ops.conv_gemm(x, conv, pads)
    if pads is not None:
        x = F.pad(x, (pads[1], pads[3], pads[0], pads[2]))
    return conv(x)


# ---------------------------------------------------------------------------------------------- blocks
class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm (same parameters and state-dict keys) whose device path is the HIP kernel pair of
    csrc/group_norm.hip, optionally fused with the SiLU that follows it in every ResNet block.
    Host tensors (only the CPU shape/name checks build these modules on the host) take torch's op."""

    def forward(self, x, silu=False):
        if x.is_cuda:
            from .. import ops
            return ops.group_norm(x, self.weight, self.bias, self.num_groups, self.eps, silu)
        y = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return F.silu(y) if silu else y


def norm_act_conv(norm, conv, x, chan_add=None, residual=None, link=None):
    """conv(silu(norm(x))) [+ chan_add[:, :, None, None]] [+ residual].  Device fp32 tensors whose shape the
    split-precision MFMA convolution covers (csrc/conv3x3.hip: 3x3/s1/p1, Cout % 32 == 0, Cin % 16 == 0,
    H % 8 == 0 and W % 32 == 0, or H % 16 == 0 and W % 16 == 0) run statistics -> normalise+SiLU -> convolution (+ the additions) in four HIP
    launches; every other shape takes the GroupNorm kernel pair followed by the library convolution."""
    if x.is_cuda:
        from .. import ops
        if USE_MFMA_CONV3X3 and ops.conv3x3_supported(conv, x):
            return ops.norm_act_conv3x3(x, norm, conv, True, chan_add, residual, link)
    assert link is None         # a ShortcutLink needs BOTH convolutions of the block on the HIP path (ResnetBlock2D checks)
    h = conv_any(conv, norm(x, silu=True))
    if chan_add is not None:
        h = h + chan_add[:, :, None, None]
    return h if residual is None else residual + h


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb=None, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout) if temb else None
        self.norm2 = GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    _t_ready = None        # set by UNet2DConditionModel for one forward: this block's time projection, already computed

    def forward(self, x, temb=None):
        if self.time_emb_proj is None:
            t = None
        elif self._t_ready is not None:
            t, self._t_ready = self._t_ready, None
        elif (USE_HIP_TIME_LINEARS and temb.is_cuda and temb.dtype == torch.float32 and temb.shape[0] <= 8
              and not (torch.is_grad_enabled() and temb.requires_grad)):
            from .. import ops
            t = ops.linear_small(temb, self.time_emb_proj.weight, self.time_emb_proj.bias, act_in=1)     # proj(silu(temb))
        else:
            t = self.time_emb_proj(F.silu(temb))
        link = None
        if (self.conv_shortcut is None and x.is_cuda and USE_MFMA_CONV3X3 and torch.is_grad_enabled() and x.requires_grad):
            from .. import ops
            if ops.conv3x3_supported(self.conv1, x) and ops.conv3x3_supported(self.conv2, x):
                link = ops.ShortcutLink()        # identity shortcut: its gradient is added inside conv1's GroupNorm backward
        h = norm_act_conv(self.norm1, self.conv1, x, chan_add=t, link=link)
        if self.conv_shortcut is None:
            sc = x
        else:
            sc = None
            if x.is_cuda:
                from .. import ops
                if USE_MFMA_CONV1X1 and ops.conv1x1_supported(self.conv_shortcut, x):
                    sc = ops.conv1x1(x, self.conv_shortcut)
            if sc is None:
                sc = conv_any(self.conv_shortcut, x)            # 8 x 8 level (64 pixels): the GEMM path pads the columns
        return norm_act_conv(self.norm2, self.conv2, h, residual=sc, link=link)


class Attention(nn.Module):
    def __init__(self, dim, ctx_dim=None, heads=8, bias=False):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=bias)
        self.to_k = nn.Linear(ctx_dim or dim, dim, bias=bias)
        self.to_v = nn.Linear(ctx_dim or dim, dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])

    def forward(self, x, ctx=None, mask=None):
        ctx = x if ctx is None else ctx
        B, N, C = x.shape
        q = self.to_q(x).view(B, N, self.heads, -1).transpose(1, 2)
        k = self.to_k(ctx).view(B, ctx.shape[1], self.heads, -1).transpose(1, 2)
        v = self.to_v(ctx).view(B, ctx.shape[1], self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, C))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm1, self.attn1 = nn.LayerNorm(dim), Attention(dim, None, heads)
        self.norm2, self.attn2 = nn.LayerNorm(dim), Attention(dim, ctx_dim, heads)
        self.norm3, self.ff = nn.LayerNorm(dim), FeedForward(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    def __init__(self, ch, heads, ctx_dim):
        super().__init__()
        self.norm = GroupNorm(32, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, ctx_dim)])
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        fast = False
        if x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad):
            from .. import ops
            if USE_HIP_TRANSFORMER:
                from . import transformer_cm
                if transformer_cm.supported(self, x):
                    return transformer_cm.transformer2d_forward(self, x, ctx)
            fast = USE_MFMA_CONV1X1 and ops.conv1x1_supported(self.proj_in, x)       # 1x1 projections on the split-precision GEMM
        if fast:
            h = ops.norm_conv1x1(x, self.norm, self.proj_in).permute(0, 2, 1)
        else:
            h = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        if fast:
            return ops.tokens_conv1x1(h, self.proj_out, x)
        return x + self.proj_out(h.reshape(B, H, W, C).permute(0, 3, 1, 2))


class Downsample2D(nn.Module):
    def __init__(self, ch, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=pad)
        self.pad = pad

    def forward(self, x):
        return conv_any(self.conv, x, (0, 0, 1, 1) if self.pad == 0 else None)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        if x.is_cuda and USE_MFMA_CONV3X3 and not (torch.is_grad_enabled() and x.requires_grad) and x.dim() == 4:
            from .. import ops
            if ops.conv3x3_supported(self.conv, x, hw=(2 * x.shape[2], 2 * x.shape[3])):    # 1280 @ 32x32, 640 @ 64x64 in the UNet
                return ops.conv3x3_plain(x, self.conv, upsample2=True)   # the up-sampling is folded into the plane writer
        x = F.interpolate(x, scale_factor=2.0, mode='nearest')
        return conv_any(self.conv, x)


# ---------------------------------------------------------------------------------------------- UNet
class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, heads, ctx_dim, attn, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb) for i in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, ctx_dim) for _ in range(2)]) if attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if down else None

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class UpBlock(nn.Module):
    def __init__(self, cprev, cout, skips, temb, heads, ctx_dim, attn, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D((cprev if i == 0 else cout) + skips[i], cout, temb)
                                      for i in range(3)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, ctx_dim) for _ in range(3)]) if attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, x, skips, temb, ctx):
        for i, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], 1), temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class MidBlock(nn.Module):
    def __init__(self, ch, temb, heads, ctx_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb), ResnetBlock2D(ch, ch, temb)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, ctx_dim)])

    def forward(self, x, temb, ctx):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, temb), ctx), temb)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(cin, cout), nn.Linear(cout, cout)

    def forward(self, x):
        if (USE_HIP_TIME_LINEARS and x.is_cuda and x.dtype == torch.float32 and x.shape[0] <= 8
                and not (torch.is_grad_enabled() and x.requires_grad)):
            from .. import ops
            h = ops.linear_small(x, self.linear_1.weight, self.linear_1.bias)
            return ops.linear_small(h, self.linear_2.weight, self.linear_2.bias, act_in=1)
        return self.linear_2(F.silu(self.linear_1(x)))


_FREQS = {}


def timestep_sinusoid(t, dim):
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0).  The frequencies depend on `dim` only and
    are computed once per device; on the device the product, cosine, sine and concatenation are one launch
    (csrc/sds_elem.hip::timestep_sincos_kernel) instead of nine tiny ones per step."""
    half = dim // 2
    key = (half, t.device)
    if key not in _FREQS:
        _FREQS[key] = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    freqs = _FREQS[key]
    if t.is_cuda and t.dim() == 1:
        from .._lib import call, ptr, stream
        tc = t.float().contiguous()
        out = torch.empty((tc.shape[0], 2 * half), device=t.device, dtype=torch.float32)
        call('mvip_timestep_sincos', ptr(tc), ptr(freqs), tc.shape[0], half, ptr(out), stream())
        return out
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], -1)


class UNet2DConditionModel(nn.Module):
    def __init__(self, in_channels=9, out_channels=4, block_out=(320, 640, 1280, 1280), heads=8, ctx_dim=768):
        super().__init__()
        temb = block_out[0] * 4
        self.in_channels = in_channels
        self.conv_in = nn.Conv2d(in_channels, block_out[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(block_out[0], temb)
        self.down_blocks = nn.ModuleList()
        c = block_out[0]
        skip_ch = [c]
        for i, co in enumerate(block_out):
            last = i == len(block_out) - 1
            self.down_blocks.append(DownBlock(c, co, temb, heads, ctx_dim, attn=not last, down=not last))
            skip_ch += [co, co] + ([] if last else [co])
            c = co
        self.mid_block = MidBlock(c, temb, heads, ctx_dim)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(block_out))
        for i, co in enumerate(rev):
            skips = [skip_ch.pop() for _ in range(3)]
            self.up_blocks.append(UpBlock(c, co, skips, temb, heads, ctx_dim, attn=i > 0, up=i < len(rev) - 1))
            c = co
        self.conv_norm_out = GroupNorm(32, block_out[0])
        self.conv_out = nn.Conv2d(block_out[0], out_channels, 3, padding=1)
        self._t_dim = block_out[0]

    mfma_prec = 0          # 1: the reference's --fp16 mode on the hand-written kernels (one fp16 product; SDNetworks sets it)

    def forward(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=False):
        if sample.is_cuda:
            from .. import ops
            with ops.precision(self.mfma_prec):
                return self._forward(sample.to(self.conv_in.weight.dtype), timestep, encoder_hidden_states)
        return self._forward(sample, timestep, encoder_hidden_states)

    def _time_projections(self, temb):
        """All 22 ResNet blocks project the SAME silu(temb): one grouped launch (ops.linear_small_grouped) whose per-block
        results are handed to the blocks, instead of 22 launches of a few microseconds of work each."""
        if not (USE_HIP_TIME_LINEARS and temb.is_cuda and temb.dtype == torch.float32 and temb.shape[0] <= 8
                and not (torch.is_grad_enabled() and temb.requires_grad)):
            return
        from .. import ops
        blocks = [m for m in self.modules() if isinstance(m, ResnetBlock2D) and m.time_emb_proj is not None]
        key = tuple((m.time_emb_proj.weight.data_ptr(), m.time_emb_proj.weight._version, m.time_emb_proj.bias._version)
                    for m in blocks)
        cache = self.__dict__.get('_tcat')
        if cache is None or cache[0] != key:
            sizes = [m.time_emb_proj.out_features for m in blocks]
            offs = torch.tensor([sum(sizes[:k]) for k in range(len(sizes) + 1)], dtype=torch.int32, device=temb.device)
            cache = (key, torch.cat([m.time_emb_proj.weight.detach() for m in blocks], 0).contiguous(),
                     torch.cat([m.time_emb_proj.bias.detach() for m in blocks], 0).contiguous(), offs, sizes)
            self.__dict__['_tcat'] = cache
        outs = ops.linear_small_grouped(temb, cache[1], cache[2], cache[3], cache[4], act_in=1)
        for m, o in zip(blocks, outs):
            m._t_ready = o

    def _forward(self, sample, timestep, encoder_hidden_states):
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(sample.shape[0])
        temb = self.time_embedding(timestep_sinusoid(t, self._t_dim).to(sample.dtype))
        self._time_projections(temb)
        x = conv_any(self.conv_in, sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, temb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states)
        return (conv_any(self.conv_out, self.conv_norm_out(x, silu=True)),)


# ---------------------------------------------------------------------------------------------- VAE
class VAEAttention(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.group_norm = GroupNorm(32, ch, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Identity()])

    def forward(self, x):
        if x.is_cuda:
            from .. import ops
            if USE_MFMA_VAE_ATTENTION and ops.vae_attention_supported(x):          # split-precision MFMA products (csrc/conv3x3.hip GEMM)
                return ops.vae_attention(x, self)
        B, C, H, W = x.shape
        h = self.group_norm(x).reshape(B, C, H * W).transpose(1, 2)
        o = F.scaled_dot_product_attention(self.to_q(h)[:, None], self.to_k(h)[:, None], self.to_v(h)[:, None])[:, 0]
        return x + self.to_out[0](o).transpose(1, 2).reshape(B, C, H, W)


class VAEMid(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, eps=1e-6), ResnetBlock2D(ch, ch, None, eps=1e-6)])
        self.attentions = nn.ModuleList([VAEAttention(ch)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, block_out=(128, 256, 512, 512), latent=4):
        super().__init__()
        self.conv_in = nn.Conv2d(3, block_out[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = block_out[0]
        for i, co in enumerate(block_out):
            blk = nn.Module()
            blk.resnets = nn.ModuleList([ResnetBlock2D(c if k == 0 else co, co, None, eps=1e-6) for k in range(2)])
            blk.downsamplers = nn.ModuleList([Downsample2D(co, pad=0)]) if i < len(block_out) - 1 else None
            self.down_blocks.append(blk)
            c = co
        self.mid_block = VAEMid(c)
        self.conv_norm_out = GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)

    def forward(self, x):
        x = conv_any(self.conv_in, x)
        for blk in self.down_blocks:
            for r in blk.resnets:
                x = r(x)
            if blk.downsamplers is not None:
                x = blk.downsamplers[0](x)
        return conv_any(self.conv_out, self.conv_norm_out(self.mid_block(x), silu=True))


class Decoder(nn.Module):
    def __init__(self, block_out=(128, 256, 512, 512), latent=4):
        super().__init__()
        rev = list(reversed(block_out))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = VAEMid(rev[0])
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            blk = nn.Module()
            blk.resnets = nn.ModuleList([ResnetBlock2D(c if k == 0 else co, co, None, eps=1e-6) for k in range(3)])
            blk.upsamplers = nn.ModuleList([Upsample2D(co)]) if i < len(rev) - 1 else None
            self.up_blocks.append(blk)
            c = co
        self.conv_norm_out = GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for blk in self.up_blocks:
            for r in blk.resnets:
                x = r(x)
            if blk.upsamplers is not None:
                x = blk.upsamplers[0](x)
        return self.conv_out(self.conv_norm_out(x, silu=True))


class LatentDist:
    """diffusers' DiagonalGaussianDistribution: mean / logvar / std of the encoder's moments, formed on first use (the SDS step
    never reads them: it draws its sample through `scaled_sample`, one launch)."""

    def __init__(self, moments):
        self.moments = moments

    @property
    def mean(self):
        return torch.chunk(self.moments, 2, dim=1)[0]

    @property
    def logvar(self):
        return torch.clamp(torch.chunk(self.moments, 2, dim=1)[1], -30.0, 20.0)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    def scaled_sample(self, noise, scaling_factor):
        """scaling_factor * (mean + std * noise) (the pipeline's _encode_vae_image), with autograd to the moments."""
        m = self.moments
        if m.is_cuda and m.dtype == torch.float32 and noise.dtype == torch.float32 and m.dim() == 4:
            return _VaeSample.apply(m, noise, float(scaling_factor))
        return scaling_factor * (self.mean + self.std * noise)


class _VaeSample(torch.autograd.Function):
    """csrc/sds_elem.hip::vae_sample_kernel / vae_sample_bwd_kernel."""

    @staticmethod
    def forward(ctx, moments, noise, sf):
        from .._lib import call, ptr, stream
        mc, nc = moments.contiguous(), noise.contiguous()
        N, C2, H, W = mc.shape
        out = torch.empty((N, C2 // 2, H, W), device=mc.device, dtype=torch.float32)
        call('mvip_vae_sample', ptr(mc), ptr(nc), sf, N, C2 // 2, H * W, ptr(out), stream())
        ctx.save_for_backward(mc, nc)
        ctx.sf = sf
        return out

    @staticmethod
    def backward(ctx, g):
        from .._lib import call, ptr, stream
        mc, nc = ctx.saved_tensors
        N, C2, H, W = mc.shape
        gc = g.contiguous().float()
        dm = torch.empty_like(mc)
        call('mvip_vae_sample_backward', ptr(mc), ptr(nc), ptr(gc), ctx.sf, N, C2 // 2, H * W, ptr(dm), stream())
        return dm, None, None


class _EncOut:
    def __init__(self, d):
        self.latent_dist = d


class AutoencoderKL(nn.Module):
    class Cfg:
        scaling_factor = 0.18215

    def __init__(self):
        super().__init__()
        self.encoder, self.decoder = Encoder(), Decoder()
        self.quant_conv, self.post_quant_conv = nn.Conv2d(8, 8, 1), nn.Conv2d(4, 4, 1)
        self.config = self.Cfg()

    mfma_prec = 0          # as UNet2DConditionModel.mfma_prec

    def encode(self, x):
        if x.is_cuda:
            from .. import ops
            with ops.precision(self.mfma_prec):
                x = x.to(self.quant_conv.weight.dtype)
                return _EncOut(LatentDist(conv_any(self.quant_conv, self.encoder(x))))
        return _EncOut(LatentDist(conv_any(self.quant_conv, self.encoder(x))))

    def decode(self, z, return_dict=False):
        return (self.decoder(self.post_quant_conv(z)),)


# ---------------------------------------------------------------------------------------------- CLIP text
class CLIPLayer(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.layer_norm1, self.layer_norm2 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))
        self.fc1, self.fc2 = nn.Linear(d, 4 * d), nn.Linear(4 * d, d)
        self.heads = heads

    def forward(self, x):
        B, N, C = x.shape
        h = self.layer_norm1(x)
        sp = lambda t: t.view(B, N, self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(self.q_proj(h)), sp(self.k_proj(h)), sp(self.v_proj(h)), is_causal=True)
        x = x + self.out_proj(o.transpose(1, 2).reshape(B, N, C))
        h = self.fc1(self.layer_norm2(x))
        return x + self.fc2(h * torch.sigmoid(1.702 * h))          # quick_gelu


class CLIPTextModel(nn.Module):
    def __init__(self, vocab=49408, d=768, layers=12, heads=12, ctx=77):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, d)
        self.position_embedding = nn.Embedding(ctx, d)
        self.layers = nn.ModuleList([CLIPLayer(d, heads) for _ in range(layers)])
        self.final_layer_norm = nn.LayerNorm(d)

    def forward(self, ids):
        x = self.token_embedding(ids) + self.position_embedding.weight[None, :ids.shape[1]]
        for l in self.layers:
            x = l(x)
        return self.final_layer_norm(x)


class ByteTokenizer:
    """Offline stand-in for CLIP's BPE tokenizer (its vocabulary files are not available here):
    BOS, one id per UTF-8 byte (offset into the vocab), EOS, EOS-padding to 77 like CLIPTokenizer."""
    BOS, EOS, CTX = 49406, 49407, 77

    def __call__(self, prompt):
        ids = [self.BOS] + [1000 + b for b in prompt.encode('utf-8')][:self.CTX - 2] + [self.EOS]
        ids += [self.EOS] * (self.CTX - len(ids))
        return torch.tensor([ids], dtype=torch.long)


def scaled_linear_alphas_cumprod(n=1000, b0=0.00085, b1=0.012):
    betas = torch.linspace(b0 ** 0.5, b1 ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class SDNetworks:
    """Bundle used by guidance.sd_utils.StableDiffusion: vae, unet, prompt encoder, scheduler table."""

    def __init__(self, device, dtype=torch.float32, seed=3, fp16_weights=True):
        """fp16_weights (default): the random parameters are rounded to fp16-REPRESENTABLE values, kept in fp32 containers
        -- what the reference's networks hold in its default mode: it always loads the `revision="fp16"` checkpoint and
        casts it up to fp32 (DS_NeRF/guidance/sd_utils.py:69-74).  The split-precision kernels then run two products per
        contraction step instead of three with bit-identical results (ops.TWO_PRODUCT, csrc/conv3x3.hip NP = 2).  False
        keeps full fp32 random values (the three-product path: tests of both)."""
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        # dtype float16 = the reference's --fp16 mode (DS_NeRF/guidance/sd_utils.py:66: every network in half).  On the
        # device it runs on the hand-written kernels in their single-product arithmetic (mfma_prec = 1: fp16 operands,
        # fp32 accumulate): the parameters are ROUNDED to fp16 values and kept in fp32 containers, so that the operand
        # packers, GroupNorm / LayerNorm and the epilogues read them as they read the fp32 networks' parameters, and
        # tensors between kernels stay fp32 (wider than the reference's fp16 tensors, never narrower).  On the host (shape
        # checks only) the modules are plain half modules.
        on_kernels = dtype == torch.float16 and torch.device(device).type == 'cuda'
        pdt = torch.float32 if on_kernels else dtype
        self.vae = AutoencoderKL().to(device=device, dtype=pdt).eval()
        self.unet = UNet2DConditionModel().to(device=device, dtype=pdt).eval()
        self.text_encoder = CLIPTextModel().to(device=device, dtype=pdt).eval()
        torch.random.set_rng_state(g)
        for m in (self.vae, self.unet, self.text_encoder):
            for p in m.parameters():
                p.requires_grad_(False)
                if on_kernels or (fp16_weights and p.dtype == torch.float32):
                    p.data = p.data.half().float()
        self.fp16_weights = bool(fp16_weights) or dtype == torch.float16
        if on_kernels:
            self.vae.mfma_prec = self.unet.mfma_prec = 1
        self.tokenizer = ByteTokenizer()
        self.alphas_cumprod = scaled_linear_alphas_cumprod()
        self.device, self.dtype = device, dtype
        self._cache = {}

    def load_checkpoint(self, root):
        """Replace the random weights by those of the diffusers-layout checkpoint directory `root` (strict key / shape check
        against guidance/sd_checkpoint_manifest.json; `fp16_weights` = every UNet / VAE tensor an exact fp16 value; the CLIP BPE
        tokenizer when its files are there) -- what the reference's from_pretrained does (DS_NeRF/guidance/sd_utils.py:69-74).
        In --fp16 mode the values are rounded to fp16 like the reference's half modules."""
        from . import sd_checkpoint
        report = sd_checkpoint.load_into(self, root)
        if self.dtype == torch.float16 and torch.device(self.device).type == 'cuda':
            for m in (self.vae, self.unet, self.text_encoder):
                for p in m.parameters():
                    p.data = p.data.half().float()
            self.fp16_weights = True
        return report

    @torch.no_grad()
    def encode_prompt(self, prompt, cfg):
        """[2,77,768] (uncond first) when cfg else [1,77,768]; cached per prompt (the reference
        re-encodes every step, DS_NeRF/guidance/sd_utils.py:317 -- same values, wasted work)."""
        key = (prompt, bool(cfg))
        if key not in self._cache:
            cond = self.text_encoder(self.tokenizer(prompt).to(self.device))
            if cfg:
                cond = torch.cat([self.text_encoder(self.tokenizer('').to(self.device)), cond], 0)
            self._cache[key] = cond
        return self._cache[key]
