"""The SD UNet's Transformer2DModel (GroupNorm -> 1x1 proj_in -> BasicTransformerBlock -> 1x1 proj_out + residual;
the unet(...) call of DS_NeRF/guidance/sd_utils.py:390-403, block structure from the published SD-1.5
architecture) executed entirely by the hand-written split-precision kernels, forward only (the UNet runs under
no_grad in every SDS step: no UNet backward ever).

Activations stay CHANNEL-MAJOR [N, C, LP] from proj_in to proj_out (LP = tokens, padded to 256 for the 8x8 level), so
  * every linear layer is `ops.gemm_f16x3` (weights = the packed A operand, tokens = columns): no NCHW <-> NHWC
    transposes and no library GEMM;
  * LayerNorm writes the GEMM's fp16 hi/lo operand planes directly (`ops.layernorm_split`);
  * the q / k / v projections are one GEMM with the heads' rows padded to a multiple of 16 channels (40 -> 48,
    zero rows), whose three thirds get their power-of-two scales in one launch;
  * attention is the flash-style kernel of csrc/attention.hip (nothing of size [Lq, Lk] in memory);
  * the prompt's key / value projections depend only on the (cached) prompt embedding: computed once per prompt;
  * (round 3, USE_SINKS) the contractions hand each other OPERANDS: the q / k / v projection's epilogue writes the attention
    kernel's Q / K planes and V fragments, attention writes the output projection's operand planes, the GEGLU projection
    writes the second feed-forward projection's planes -- each at a power-of-two scale fixed at pack time from a rigorous
    bound of the tensor (|LayerNorm| <= sqrt(C) |gamma|max + |beta|max, |W x| <= |x|max max_row ||W||_1, |softmax V| <=
    |V|max, |a gelu(g)| <= |a| |g|).  No fp32 intermediate, no absolute-maximum pass, no split pass inside a block:
    11 launches fewer per block (30 -> 19).
Stock torch ops left in here: tensor allocation, the zero-padding copy of the 8x8 level, and nothing else.
"""
import math

import torch

from .. import ops


def _pad_head_rows(W, heads, D, DP):
    """[heads*D, K] -> [heads*DP, K] with each head's rows at h*DP .. h*DP + D - 1 (zero rows between)."""
    if DP == D:
        return W.detach().contiguous()
    out = torch.zeros((heads * DP, W.shape[1]), device=W.device, dtype=W.dtype)
    out.view(heads, DP, -1)[:, :D] = W.detach().view(heads, D, -1)
    return out


def _ln_scale(ln, C):
    """Power of two s with |LayerNorm(x)| * s < 2^15 for every input (|y| <= sqrt(C) max|gamma| + max|beta|)."""
    bound = math.sqrt(C) * float(ln.weight.detach().abs().max()) + float(ln.bias.detach().abs().max())
    k = int(math.floor(math.log2(30000.0 / max(bound, 1e-30))))
    return float(2.0 ** max(-20, min(4, k)))


def _scale_tensor(s, device):
    return torch.tensor([s, 1.0 / s, 0.0, 0.0], device=device, dtype=torch.float32)


USE_SINKS = True      # A/B switch: False restores the round-2 chain (fp32 intermediates + measured scales)


def _ln_bound(ln, C):
    """|LayerNorm(x)_c| <= sqrt(C) |gamma|max + |beta|max for EVERY input (|x_c - mean| <= sqrt(C) std)."""
    return math.sqrt(C) * float(ln.weight.detach().abs().max()) + float(ln.bias.detach().abs().max())


def _row_l1(W):
    """max over rows of sum_k |W[row][k]|: |W x|max <= |x|max * this."""
    return float(W.detach().abs().sum(1).max())


def _bound_after_ln(W, bias, ln, C):
    """max_i |(W LayerNorm(x) + bias)_i| over EVERY input x: LayerNorm(x) = gamma * xhat + beta with ||xhat||_2 <= sqrt(C),
    so |W_i . LN(x)| <= ||W_i * gamma||_2 sqrt(C) + |W_i . beta| (Cauchy-Schwarz; ~sqrt(C) tighter than the row-L1 form)."""
    Wd, g, b = W.detach().double(), ln.weight.detach().double(), ln.bias.detach().double()
    v = (Wd * g[None, :]).norm(dim=1) * math.sqrt(C) + (Wd @ b).abs()
    if bias is not None:
        v = v + bias.detach().double().abs()
    return float(v.max())


class _Packed:
    """Packed weight images of one Transformer2DModel (frozen network: rebuilt only if a weight's version moves)."""

    def __init__(self, mod):
        blk = mod.transformer_blocks[0]
        a1, a2, ff = blk.attn1, blk.attn2, blk.ff
        C = mod.proj_in.in_channels
        heads = a1.heads
        D = C // heads
        DP = (D + 15) // 16 * 16
        R = heads * DP
        dev = mod.proj_in.weight.device
        self.C, self.heads, self.D, self.DP, self.R = C, heads, D, DP, R
        self.key = self.version_key(mod)
        pack = lambda *a: ops.gemm_pack_a(*a, weights=True)     # frozen weights: two-product contractions when fp16-exact
        wqkv = torch.cat([_pad_head_rows(w.weight, heads, D, DP) for w in (a1.to_q, a1.to_k, a1.to_v)], 0).contiguous()
        self.qkv1 = pack(wqkv, 3 * R, C, C, 1)
        self.o1 = pack(a1.to_out[0].weight.detach().contiguous(), C, C, C, 1)
        self.bo1 = a1.to_out[0].bias.detach().contiguous()
        self.q2 = pack(_pad_head_rows(a2.to_q.weight, heads, D, DP), R, C, C, 1)
        ctx_dim = a2.to_k.weight.shape[1]
        wkv = torch.cat([_pad_head_rows(a2.to_k.weight, heads, D, DP), _pad_head_rows(a2.to_v.weight, heads, D, DP)], 0)
        self.kv2 = pack(wkv.contiguous(), 2 * R, ctx_dim, ctx_dim, 1)
        self.ctx_dim = ctx_dim
        self.o2 = pack(a2.to_out[0].weight.detach().contiguous(), C, C, C, 1)
        self.bo2 = a2.to_out[0].bias.detach().contiguous()
        w1 = ff.net[0].proj
        wi, bi = ops.geglu_interleave(w1.weight.detach(), w1.bias.detach())       # value / gate rows in 32-row tiles
        self.ff1 = pack(wi, 8 * C, C, C, 1)
        self.b1 = bi
        self.ff2 = pack(ff.net[2].weight.detach().contiguous(), C, 4 * C, 4 * C, 1)
        self.b2 = ff.net[2].bias.detach().contiguous()
        self.pin = pack(mod.proj_in.weight.detach().reshape(C, C).contiguous(), C, C, C, 1)
        self.bin = mod.proj_in.bias.detach().contiguous()
        self.pout = pack(mod.proj_out.weight.detach().reshape(C, C).contiguous(), C, C, C, 1)
        self.bout = mod.proj_out.bias.detach().contiguous()
        self.ln = []
        for ln in (blk.norm1, blk.norm2, blk.norm3):
            s = _ln_scale(ln, C)
            self.ln.append((ln.weight.detach().contiguous(), ln.bias.detach().contiguous(), float(ln.eps), s,
                            _scale_tensor(s, dev)))
        # ---- operand sinks: weight images and the power-of-two scales of every intermediate, fixed here ----
        DT = (D + 31) // 32
        self.DT, self.RV = DT, heads * DT * 32
        # V rows padded per head to the attention kernel's row tiles (DT * 32), so that a 32-row GEMM tile is one
        # (head, row tile) block of the V fragments
        wqkv_s = torch.cat([_pad_head_rows(a1.to_q.weight, heads, D, DP), _pad_head_rows(a1.to_k.weight, heads, D, DP),
                            _pad_head_rows(a1.to_v.weight, heads, D, DT * 32)], 0).contiguous()
        self.qkv1_s = pack(wqkv_s, 2 * R + self.RV, C, C, 1)
        sc = ops.pow2_scale_for_bound
        bq1, bk1, bv1 = (_bound_after_ln(w.weight, None, blk.norm1, C) for w in (a1.to_q, a1.to_k, a1.to_v))
        self.s_q1, self.s_k1, self.s_v1 = sc(bq1), sc(bk1), sc(bv1)
        self.s_q2 = sc(_bound_after_ln(a2.to_q.weight, None, blk.norm2, C))
        wa, wg = w1.weight.detach()[:4 * C], w1.weight.detach()[4 * C:]
        ba, bg = w1.bias.detach()[:4 * C], w1.bias.detach()[4 * C:]
        bound_act = _bound_after_ln(wa, ba, blk.norm3, C) * _bound_after_ln(wg, bg, blk.norm3, C)   # |a gelu(g)| <= |a| |g|
        self.s_act = sc(bound_act)
        # the residual stream after the block, |h3| <= |h0| + |o1 term| + |o2 term| + |ff term|: the pieces that do not
        # depend on the call (h0's bound depends on the token count through GroupNorm, o2's on the prompt)
        self.h_static = (bv1 * _row_l1(a1.to_out[0].weight) + float(a1.to_out[0].bias.detach().abs().max())
                         + bound_act * _row_l1(ff.net[2].weight) + float(ff.net[2].bias.detach().abs().max()))
        self.o2_l1, self.o2_b = _row_l1(a2.to_out[0].weight), float(a2.to_out[0].bias.detach().abs().max())
        gn = mod.norm
        self.gn_gb = (float(gn.weight.detach().abs().max()), float(gn.bias.detach().abs().max()), C // gn.num_groups)
        self.pin_l1, self.pin_b = _row_l1(mod.proj_in.weight.detach().reshape(C, C)), float(mod.proj_in.bias.detach().abs().max())
        self.h3_scales = {}
        self.t_q1, self.t_k1, self.t_v1, self.t_q2, self.t_act = (_scale_tensor(v, dev) for v in
                                                                  (self.s_q1, self.s_k1, self.s_v1, self.s_q2, self.s_act))
        self.ctx_cache = {}             # (ptr, version, shape) -> (ctx, planes ...): one entry PER PROMPT, never cleared

    @staticmethod
    def version_key(mod):
        return tuple((p.data_ptr(), p._version) for p in mod.parameters())


def supported(mod, x):
    """fp32 device tensors, one transformer block, a head size the attention kernel is built for."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and len(mod.transformer_blocks) == 1):
        return False
    blk = mod.transformer_blocks[0]
    # _Packed packs the q / k / v WEIGHTS only and addresses the feed-forward by position: anything else (biased
    # projections, another feed-forward layout) takes the module's own forward
    for a in (blk.attn1, blk.attn2):
        if any(getattr(lin, 'bias', None) is not None for lin in (a.to_q, a.to_k, a.to_v)):
            return False
        if not isinstance(a.to_out[0], torch.nn.Linear):
            return False
    net = blk.ff.net
    if not (len(net) == 3 and hasattr(net[0], 'proj') and isinstance(net[0].proj, torch.nn.Linear)
            and type(net[0]).__name__ == 'GEGLU' and isinstance(net[2], torch.nn.Linear)
            and net[0].proj.bias is not None and net[2].bias is not None):
        return False
    C = mod.proj_in.in_channels
    heads = mod.transformer_blocks[0].attn1.heads
    L = x.shape[2] * x.shape[3]
    return (C % 64 == 0 and C % heads == 0 and bool(ops._lib.load().mvip_attention_supported(C // heads))
            and L % 64 == 0 and mod.proj_in.weight.dtype == torch.float32)


def _packed(mod):
    pk = mod.__dict__.get('_mvip_cm')
    if pk is None or pk.key != _Packed.version_key(mod):
        pk = _Packed(mod)
        mod.__dict__['_mvip_cm'] = pk
    return pk


_UNIT_SECTIONS = {}


def _scales(x, outer, sections, length):
    """Power-of-two scales of the q / k / v sections: 1 for every section when the UNet's forward activations are
    split at a fixed scale (ops.forward_unit_scale()), else measured in one launch."""
    if ops.forward_unit_scale():
        key = (x.device, sections)
        if key not in _UNIT_SECTIONS:
            _UNIT_SECTIONS[key] = ops.unit_scale(x.device).repeat(sections)
        return _UNIT_SECTIONS[key]
    return ops.absmax_scale_sections(x, outer, sections, length)


MAX_PROMPTS = 16


def prompt_entries(unet):
    """Every cached prompt entry of every transformer of `unet` (what a captured graph must keep alive)."""
    out = []
    for m in unet.modules():
        pk = m.__dict__.get('_mvip_cm')
        if pk is not None:
            out.extend(pk.ctx_cache.values())
    return out


def _prompt_kv(pk, ctx, want_vmax=False):
    """Key planes / value fragments of the prompt tokens for the cross-attention (constant per prompt).  want_vmax: also
    the power of two >= |v|max (read back ONCE per prompt, when the entry is made: 1024 / scale of the measured split)."""
    key = (ctx.data_ptr(), ctx._version, tuple(ctx.shape), ops._prec())     # fp16-mode entries carry no lo halves
    hit = pk.ctx_cache.get(key)
    if hit is not None:
        return hit[1:] if want_vmax else hit[1:-1]
    N, T, E = ctx.shape
    TP, GP = 128, 256                          # key padding of the attention kernel / column padding of the GEMM
    assert T <= TP and E == pk.ctx_dim
    cpad = torch.zeros((N, GP, E), device=ctx.device, dtype=torch.float32)
    cpad[:, :T] = ctx.detach().float()
    xs, s2 = ops._scaled_planes(cpad, N, E, GP, GP * E, 1, E, forward_activation=True)   # X[n][k][p] = ctx[n][p][k]
    kv = ops.gemm_f16x3(xs, pk.kv2, N, E, 2 * pk.R, GP, x_scale2=s2)                # [N, 2R, GP]
    sc = _scales(kv, N, 2, pk.R * GP)
    flat = kv.reshape(-1)
    ks = ops.split_planes_strided(flat, N, pk.R, TP, 2 * pk.R * GP, GP, 1, sc[0:4])
    vp = ops.attention_pack_v(flat[pk.R * GP:], N, pk.heads, pk.D, pk.DP, T, TP, 2 * pk.R * GP, GP, 1, sc[4:8])
    # The power of two above |v|max, MEASURED whatever scale the split used (under ops.forward_unit_scale() -- always in
    # fp16 mode -- `sc` is 1 and says nothing about the values; the residual-stream bound of _h3_scale must not rest on an
    # assumed range).  With the measured split scale this equals 1024 / sc[4] bit for bit.  Host read-back once per prompt
    # (not in a capture: the graphed step fills this cache in its eager warm-up).
    import math
    v_abs = float(kv.reshape(N, 2, -1)[:, 1].abs().max())
    vmax = 2.0 ** math.frexp(v_abs)[1] if v_abs > 0 and math.isfinite(v_abs) else 1.0
    hit = (ks, vp, sc[0:4], sc[4:8], T, TP, vmax)
    # One entry per prompt, kept for the life of the module: a captured hipGraph replays against the addresses of the
    # entry it was captured with, so an entry must never be freed while another prompt runs (RGB text / text_normal
    # alternate inside one iteration).  The entry holds `ctx` itself: the key is an address, and a live tensor keeps it
    # from being handed to another one.  Callers that build a fresh embedding tensor per call (not this repo's
    # SDNetworks.encode_prompt, which caches) are bounded by dropping the oldest entry; a graph keeps its own
    # references (sd_utils._GraphedStep.pinned), so eviction never frees memory a graph replays against.
    while len(pk.ctx_cache) >= MAX_PROMPTS:
        pk.ctx_cache.pop(next(iter(pk.ctx_cache)))
    pk.ctx_cache[key] = (ctx,) + hit
    return hit if want_vmax else hit[:-1]


def _block_sinks(h, pk, ctx, N, L, LP, st_h=None):
    """The block with every contraction writing the next one's operands (module docstring, USE_SINKS).  st_h: the LayerNorm
    statistics of h its producer left (ops.gemm_f16x3 with ln_stats); round 6: each projection that writes the residual
    stream leaves the next LayerNorm's statistics from its epilogue."""
    C, R, heads, D = pk.C, pk.R, pk.heads, pk.D
    # ---- self-attention: LayerNorm planes -> {Q planes, K planes, V fragments} -> attention -> o planes -> projection ----
    g, b, eps, s, st = pk.ln[0]
    xs = ops.layernorm_split(h, g, b, eps, N, C, L, LP, s, stats=st_h)
    qs, ks, vp = ops.gemm_f16x3_sinks(xs, pk.qkv1_s, N, C, LP, [(R, 'planes', pk.s_q1), (R, 'planes', pk.s_k1),
                                                                (pk.RV, 'vfrag', pk.s_v1)], x_scale2=st, v_dt=pk.DT)
    op = ops.attention_f16x3_sink(qs, ks, vp, pk.t_q1, pk.t_k1, pk.t_v1, N, heads, D, L, LP, L, L, LP, LP, LP // 16)
    h, st_h = ops.gemm_f16x3(op, pk.o1, N, C, C, LP, bias=pk.bo1, residual=h, x_scale2=pk.t_v1, ln_stats=True)
    # ---- cross-attention onto the prompt tokens (their K planes / V fragments are cached per prompt) ----
    g, b, eps, s, st = pk.ln[1]
    xs = ops.layernorm_split(h, g, b, eps, N, C, L, LP, s, stats=st_h)
    (qs,) = ops.gemm_f16x3_sinks(xs, pk.q2, N, C, LP, [(R, 'planes', pk.s_q2)], x_scale2=st)
    ks2, vp2, sk2, sv2, T, TP, vmax2 = _prompt_kv(pk, ctx, want_vmax=True)
    op = ops.attention_f16x3_sink(qs, ks2, vp2, pk.t_q2, sk2, sv2, N, heads, D, L, LP, T, TP, LP, TP, TP // 16)
    h, st_h = ops.gemm_f16x3(op, pk.o2, N, C, C, LP, bias=pk.bo2, residual=h, x_scale2=sv2, ln_stats=True)
    # ---- GEGLU feed-forward: the product leaves the first projection as the second one's operand planes ----
    g, b, eps, s, st = pk.ln[2]
    xs = ops.layernorm_split(h, g, b, eps, N, C, L, LP, s, stats=st_h)
    ap = ops.gemm_geglu_f16x3_sink(xs, pk.ff1, pk.b1, N, C, 8 * C, LP, L, pk.s_act, x_scale2=st)
    # the finished residual stream leaves as proj_out's operand planes (its fp32 form has no other reader)
    s_h3, t_h3 = _h3_scale(pk, L, vmax2)
    hp = ops.gemm_f16x3_planes(ap, pk.ff2, N, 4 * C, C, LP, s_h3, bias=pk.b2, residual=h, x_scale2=pk.t_act)
    return hp, t_h3


def _h3_scale(pk, L, vmax2):
    """Power-of-two scale of the block's output from a bound of the residual stream:
    |h0| <= |GroupNorm(x)|max ||W_in||_1 + |b_in| with |GroupNorm(x)| <= sqrt(m - 1) |gamma|max + |beta|max (m = elements of
    a group = channels per group x tokens), plus the three sub-layers' bounds."""
    key = (L, vmax2)
    hit = pk.h3_scales.get(key)
    if hit is None:
        gmax, bmax, cpg = pk.gn_gb
        b_gn = math.sqrt(max(cpg * L - 1, 1)) * gmax + bmax
        bound = b_gn * pk.pin_l1 + pk.pin_b + pk.h_static + vmax2 * pk.o2_l1 + pk.o2_b
        s = ops.pow2_scale_for_bound(bound)
        hit = (s, _scale_tensor(s, pk.bin.device))
        pk.h3_scales[key] = hit
    return hit


def _block(h, pk, ctx, N, L, LP):
    C, R, heads, D, DP = pk.C, pk.R, pk.heads, pk.D, pk.DP
    # ---- self-attention ----
    g, b, eps, s, st = pk.ln[0]
    xs = ops.layernorm_split(h, g, b, eps, N, C, L, LP, s)
    qkv = ops.gemm_f16x3(xs, pk.qkv1, N, C, 3 * R, LP, x_scale2=st)                  # [N, 3R, LP]
    sc = _scales(qkv, N, 3, R * LP)
    flat = qkv.reshape(-1)
    qs = ops.split_planes_strided(flat, N, R, L, 3 * R * LP, LP, 1, sc[0:4])
    ks = ops.split_planes_strided(flat[R * LP:], N, R, L, 3 * R * LP, LP, 1, sc[4:8])
    vp = ops.attention_pack_v(flat[2 * R * LP:], N, heads, D, DP, L, L, 3 * R * LP, LP, 1, sc[8:12])
    o = ops.attention_f16x3(qs, ks, vp, sc[0:4], sc[4:8], sc[8:12], N, heads, D, L, LP, L, L)
    os_ = ops.split_planes_strided(o, N, C, LP, C * LP, LP, 1, sc[8:12])            # |o| <= max|v|: v's scale fits
    h = ops.gemm_f16x3(os_, pk.o1, N, C, C, LP, bias=pk.bo1, residual=h, x_scale2=sc[8:12])
    # ---- cross-attention onto the prompt tokens ----
    g, b, eps, s, st = pk.ln[1]
    xs = ops.layernorm_split(h, g, b, eps, N, C, L, LP, s)
    q = ops.gemm_f16x3(xs, pk.q2, N, C, R, LP, x_scale2=st)
    sq = _scales(q, 1, 1, N * R * LP)
    qs = ops.split_planes_strided(q, N, R, L, R * LP, LP, 1, sq)
    ks2, vp2, sk2, sv2, T, TP = _prompt_kv(pk, ctx)
    o = ops.attention_f16x3(qs, ks2, vp2, sq, sk2, sv2, N, heads, D, L, LP, T, TP)
    os_ = ops.split_planes_strided(o, N, C, LP, C * LP, LP, 1, sv2)
    h = ops.gemm_f16x3(os_, pk.o2, N, C, C, LP, bias=pk.bo2, residual=h, x_scale2=sv2)
    # ---- GEGLU feed-forward ----
    g, b, eps, s, st = pk.ln[2]
    xs = ops.layernorm_split(h, g, b, eps, N, C, L, LP, s)
    act, sa = ops.gemm_geglu_f16x3(xs, pk.ff1, pk.b1, N, C, 8 * C, LP, L, x_scale2=st)   # GEGLU in the GEMM's epilogue
    as_ = ops.split_planes_strided(act, N, 4 * C, LP, 4 * C * LP, LP, 1, sa)
    return ops.gemm_f16x3(as_, pk.ff2, N, 4 * C, C, LP, bias=pk.b2, residual=h, x_scale2=sa)


def transformer2d_forward(mod, x, ctx):
    """Transformer2DModel.forward(x [N, C, H, W], ctx [N, 77, 768]) -> [N, C, H, W]; caller checks `supported`."""
    pk = _packed(mod)
    N, C, H, W = x.shape
    L = H * W
    xc = x.detach().contiguous()
    if L % 256 == 0:
        LP = L
        h, st_h = ops.norm_conv1x1(xc, mod.norm, mod.proj_in, ln_stats=True)             # GroupNorm + 1x1 conv: [N, C, L]
        res = xc.reshape(N, C, L)
    else:                                                                            # the 8x8 level: pad the token axis
        LP = (L + 255) // 256 * 256
        hn = ops.group_norm(xc, mod.norm.weight, mod.norm.bias, mod.norm.num_groups, mod.norm.eps, False)
        hp = torch.zeros((N, C, LP), device=x.device, dtype=torch.float32)
        hp[:, :, :L] = hn.reshape(N, C, L)
        xs, s2 = ops._scaled_planes(hp, N, C, LP, C * LP, LP, 1, forward_activation=True)
        h, st_h = ops.gemm_f16x3(xs, pk.pin, N, C, C, LP, bias=pk.bin, x_scale2=s2, ln_stats=True)
        res = torch.zeros((N, C, LP), device=x.device, dtype=torch.float32)
        res[:, :, :L] = xc.reshape(N, C, L)
    if USE_SINKS:
        xs, s2 = _block_sinks(h, pk, ctx, N, L, LP, st_h)
    else:
        h = _block(h, pk, ctx, N, L, LP)
        xs, s2 = ops._scaled_planes(h, N, C, LP, C * LP, LP, 1, forward_activation=True)
    y = ops.gemm_f16x3(xs, pk.pout, N, C, C, LP, bias=pk.bout, residual=res, x_scale2=s2)
    if LP != L:
        y = y[:, :, :L].contiguous()
    return y.reshape(N, C, H, W)
