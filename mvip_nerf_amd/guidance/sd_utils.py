"""Score-distillation guidance: mirror of `DS_NeRF/guidance/sd_utils.py::StableDiffusion` (the three
step methods the live path calls through Pretrain_Model.cal_loss) and `SpecifyGradient`.

Same method names, argument orders and defaults.  What is kept from the reference, bit for bit in
arithmetic: bilinear resize to 512^2 (align_corners=False), |mask|, masked_image = rgb*(mask<0.5)
with NO [0,1]->[-1,1] rescale (sd_utils.py:329-330), nearest 64^2 mask, VAE posterior SAMPLE (not
mean) scaled by 0.18215, t(i) = int(980 - 960*sqrt(i/20000)) (sd_utils.py:363-365), scheduler.add_noise,
CFG batch [uncond, cond], w = 1 - abar_t, nan_to_num, SpecifyGradient with the 64^2 mask, and the
colla variant's quirks (t from the VIEW index, accumulated grad, only the last view's graph gets
gradient, doubled by the CFG-duplicated mask; sd_utils.py:442, :527, :575, :597).

What is dropped because it cannot change any returned value or gradient: the unused VAE encode of
init_image (sd_utils.py:354), the unused VAE decode + PIL conversion (:418-425), the per-step PNG
(:416), per-step prompt re-encoding (:317, cached instead).  The dropped encode DID consume one
randn draw; with reference_rng=True (default) that draw is still consumed so a seeded run sees
the same random stream as the reference.

Elementwise cores (add_noise; CFG + w*(eps-noise) + nan_to_num) are single HIP kernels
(csrc/sds_elem.hip); the networks are PyTorch-ROCm modules (guidance/sd_nets.py) or any injected
object with the diffusers call signatures.
"""
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import ptr, stream, call

img2mse = lambda x, y: torch.mean((x - y) ** 2)


class SpecifyGradient(torch.autograd.Function):
    """DS_NeRF/guidance/sd_utils.py:21-37: forward returns ones([1]); backward injects
    gt_grad * upstream * mask as the gradient of the latents."""

    @staticmethod
    def forward(ctx, input_tensor, gt_grad, mask):
        ctx.save_for_backward(gt_grad, mask)
        return torch.ones([1], device=input_tensor.device, dtype=input_tensor.dtype)

    @staticmethod
    def backward(ctx, grad_scale):
        gt_grad, mask = ctx.saved_tensors
        return gt_grad * grad_scale * mask, None, None


class _AddNoise(torch.autograd.Function):
    """scheduler.add_noise: sqrt(abar) x0 + sqrt(1-abar) noise, one HIP kernel; d/dx0 = sqrt(abar)."""

    @staticmethod
    def forward(ctx, x0, noise, sa, sb):
        x0c, nc = x0.contiguous().float(), noise.contiguous().float()
        out = torch.empty_like(x0c)
        call('mvip_sds_add_noise', ptr(x0c), ptr(nc), sa, sb, x0c.numel(), ptr(out), stream())
        ctx.sa = sa
        return out.to(x0.dtype)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.sa, None, None, None


def _resize_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) (DS_NeRF/guidance/sd_utils.py:282-284): the HIP
    kernel pair for fp32 device tensors, the torch op for anything else (CPU tensors in host-side tests)."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4:
        from .. import ops
        return ops.resize_bilinear(x, size)
    return F.interpolate(x, size, mode='bilinear', align_corners=False)


def sds_grad(eps_uncond, eps_cond, noise, guidance_scale, w, accumulate_into=None):
    """nan_to_num([accumulate_into +] w * (e_u + s (e_c - e_u) - noise)) in one HIP kernel."""
    eu = eps_uncond.contiguous().float()
    ec = None if eps_cond is None else eps_cond.contiguous().float()
    nz = noise.contiguous().float()
    out = accumulate_into if accumulate_into is not None else torch.empty_like(eu)
    call('mvip_sds_grad', ptr(eu), ptr(ec), ptr(nz), float(guidance_scale), float(w), eu.numel(),
         int(accumulate_into is not None), ptr(out), stream())
    return out


class _AddNoiseDev(torch.autograd.Function):
    """add_noise with {sqrt(abar), sqrt(1-abar)} read from a device tensor (graph-replayable)."""

    @staticmethod
    def forward(ctx, x0, noise, scal):
        x0c, nc = x0.contiguous().float(), noise.contiguous().float()
        out = torch.empty_like(x0c)
        call('mvip_sds_add_noise_dev', ptr(x0c), ptr(nc), ptr(scal), x0c.numel(), ptr(out), stream())
        ctx.save_for_backward(scal)
        return out.to(x0.dtype)

    @staticmethod
    def backward(ctx, g):
        (scal,) = ctx.saved_tensors
        return g * scal[0], None, None


def sds_grad_dev(eps_uncond, eps_cond, noise, guidance_scale, scal, accumulate_into=None):
    eu = eps_uncond.contiguous().float()
    ec = None if eps_cond is None else eps_cond.contiguous().float()
    nz = noise.contiguous().float()
    out = accumulate_into if accumulate_into is not None else torch.empty_like(eu)
    call('mvip_sds_grad_dev', ptr(eu), ptr(ec), ptr(nz), float(guidance_scale), ptr(scal), eu.numel(),
         int(accumulate_into is not None), ptr(out), stream())
    return out


class _InjectGrad(torch.autograd.Function):
    """ones([1]) forward; backward hands `d_pred * upstream` to the image (the graph already holds
    J^T (grad . mask), i.e. SpecifyGradient's backward chained through the VAE encoder and resize)."""

    @staticmethod
    def forward(ctx, pred, d_pred):
        ctx.save_for_backward(d_pred)
        return torch.ones([1], device=pred.device, dtype=pred.dtype)

    @staticmethod
    def backward(ctx, upstream):
        (d_pred,) = ctx.saved_tensors
        return d_pred * upstream, None


class _NoiseFeed:
    """What StableDiffusion._randn hands out while a step is being captured / warmed up.  A captured step contains NO random
    number generation (round 5): a hipGraph that draws reads the generator's seed / offset from device words the GENERATOR owns,
    written by every replay's prologue -- two captured steps replaying side by side on different streams (nerf/utils.cal_loss)
    then race on those words and draw from each other's offsets.  Instead the draws of a step are recorded once (shapes, order),
    kept as static input buffers of the graph, and filled EAGERLY before each replay, in the recorded order, from the same
    generator: the values an eager step would have drawn."""

    def __init__(self):
        self.recording, self.specs, self.buffers, self.pos = True, [], [], 0

    def take(self, sd, shape, dtype):
        if self.recording:
            self.specs.append((tuple(shape), dtype))
            return torch.randn(tuple(shape), device=sd.device, dtype=dtype, generator=sd.generator)
        buf = self.buffers[self.pos]
        assert tuple(buf.shape) == tuple(shape) and buf.dtype == dtype, 'the captured step draws what the recorded step drew'
        self.pos += 1
        return buf

    def freeze(self, device):
        self.recording = False
        self.buffers = [torch.zeros(shape, device=device, dtype=dt) for shape, dt in self.specs]

    def refill(self, generator):
        for b in self.buffers:
            b.normal_(generator=generator)


class _OffDefaultStream:
    """Runs a block on a pool stream instead of the device's DEFAULT stream (the default stream waits for it afterwards).
    Used by nerf/utils.Pretrain_Model.cal_loss for the single-term iterations of a model that is CONFIGURED for several terms.
    Why (round 6, same-box experiments, profiles/r6_stream_experiments.json): once ONE captured step had been captured / replayed
    from the default stream, the replays of two LATER graphs on two other streams no longer overlapped for the rest of the process
    -- the BASELINE configs[2] iteration took 149.3 ms instead of 142.5, whatever was released afterwards and with 4 or 8 hardware
    queues -- while the same earlier step issued from a pool stream left the later concurrency intact (142.7 ms).  That is what a
    configs[2] / configs[3] run does: iterations up to `normal_start` evaluate ONE term in line, every later one two or three on
    streams of their own.  The hop itself costs a single-term step 0.4-0.9 ms (21.3 -> 21.7-22.2 ms: a replay from a pool stream
    is slower than from the default stream), so a model configured for ONE term (configs[1], the metric's configuration) and direct
    callers of train_step_sd stay on the default stream.  MVIP_SDS_OFF_DEFAULT_STREAM=0 switches the hop off (A/B)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.hop = None

    def __enter__(self):
        import os
        if self.device.type != 'cuda' or os.environ.get('MVIP_SDS_OFF_DEFAULT_STREAM', '1') == '0':      # A/B switch
            return self
        self.cur = torch.cuda.current_stream(self.device)
        if self.cur.cuda_stream == torch.cuda.default_stream(self.device).cuda_stream:
            from .. import streams as _streams
            self.hop = _streams.get(self.device, 'term', 0)
            self.hop.wait_stream(self.cur)
            self._ctx = torch.cuda.stream(self.hop)
            self._ctx.__enter__()
        return self

    def keep(self, *tensors):
        """Results made on the hop stream and read by the caller's stream afterwards."""
        if self.hop is not None:
            for t in tensors:
                if torch.is_tensor(t):
                    t.record_stream(self.cur)

    def __exit__(self, *exc):
        if self.hop is not None:
            self._ctx.__exit__(*exc)
            self.cur.wait_stream(self.hop)
        return False


class _GraphedStep:
    """The whole single-view SDS step -- resize, masking, two VAE encodes, add_noise, UNet (CFG batch),
    SDS gradient, and the backward through the VAE encoder to the image -- captured once as a hipGraph.
    ~1400 kernel launches per step otherwise leave the GPU idle a third of the time (profiles/)."""

    def __init__(self, sd, pred_shape, mask_shape, prompt, guidance_scale, mode='single'):
        """mode: 'single' = train_step_sd / train_step_sd_normal; 'share' = a non-final neighbour view of
        train_step_colla_sds (forward only: its w (eps_hat - eps) is ADDED to the running latent gradient `acc`, nothing
        flows back to the image, DS_NeRF/guidance/sd_utils.py:575); 'last' = the final neighbour view (its share joins `acc`
        and the sum is injected through SpecifyGradient with the CFG-duplicated mask, :597-599)."""
        self.sd, self.prompt, self.gs, self.mode = sd, prompt, guidance_scale, mode
        dev = sd.device
        self.pred = torch.zeros(pred_shape, device=dev, requires_grad=mode != 'share')
        self.mask = torch.zeros(mask_shape, device=dev)
        self.acc = torch.zeros(1, 4, 64, 64, device=dev) if mode != 'single' else None
        self.scal = torch.zeros(4, device=dev)                  # sqrt(abar), sqrt(1-abar), 1-abar, t
        sd.networks.encode_prompt(prompt, guidance_scale > 1.0)    # cached constant, outside the capture
        # The first warm-up run DRAWS (and records what it draws); the generator's state is put back afterwards, so that the
        # first replay consumes exactly the draws an eager first step would have (measured: replays equal eager steps to
        # atomics-level, 2e-6 relative, draw for draw -- profiles/r4_graph_vs_eager_draws.json).
        # the default generator of the device the draws are made on: a bare 'cuda' means the CURRENT device, not device 0
        dev_index = torch.device(dev).index
        if dev_index is None:
            dev_index = torch.cuda.current_device()
        gen = sd.generator if sd.generator is not None else torch.cuda.default_generators[dev_index]
        gen_state = gen.get_state()
        # this graph's OWN scratch words (ops.ZERO_SCOPE): two captured steps may replay side by side on different streams
        # (nerf/utils.Pretrain_Model.cal_loss runs the RGB / normal / collaborative terms concurrently)
        from .. import ops as _ops
        scope_before, _ops.ZERO_SCOPE = _ops.ZERO_SCOPE, ('graph', id(self))
        cur = torch.cuda.current_stream()
        from .. import streams as _streams
        side = _streams.get(dev, 'capture')                     # one warm-up stream per process, not one per captured graph (streams.py)
        side.wait_stream(cur)
        self.noise = _NoiseFeed()                               # first warm-up run: records what the step draws
        sd._noise_feed = self.noise
        try:
            with torch.cuda.stream(side):                       # warm-up (library autotuning) before capture
                self._body()
                self.noise.freeze(dev)                          # from here on the draws are static input buffers
                self.noise.pos = 0
                self._body()
            cur.wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            self.generator = sd.generator
            self.noise.pos = 0
            # thread_local: another thread of this process (an RCCL watchdog, a data loader) may touch the device meanwhile
            with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                self.d_pred = self._body()
        finally:
            # also on a failed warm-up / capture (out of memory, capture error): later eager kernels and later captures must not
            # keep pointing at this graph's scratch words, and the warm-up's draws must not stay consumed
            sd._noise_feed = None
            _ops.ZERO_SCOPE = scope_before
            gen.set_state(gen_state)
        self._scope = ('graph', id(self))
        # everything the capture read from per-prompt caches (the cross-attention's key / value planes, filled by the
        # warm-up above) lives as long as this graph, whatever another prompt does to those caches afterwards
        from . import transformer_cm
        self.pinned = transformer_cm.prompt_entries(sd.unet) if isinstance(sd.unet, nn.Module) else []
        self._done = None                                       # event after the last replay's output clones (see run)

    def __del__(self):
        try:                                                    # this graph's scratch words go with it
            from .. import ops as _ops
            for k in [k for k in _ops._ZERO_WORDS if k[2] == getattr(self, '_scope', None)]:
                del _ops._ZERO_WORDS[k]
        except Exception:                                       # noqa: BLE001 -- interpreter shutdown
            pass

    def _body(self):
        sd = self.sd
        with torch.set_grad_enabled(self.mode != 'share'):
            init_image, mask64, masked_latents, emb, cfg = sd._prepare(self.pred, self.mask, self.prompt, self.gs)
            image_latents = sd._encode_vae_image(init_image)
            noise = sd._randn(image_latents.shape, image_latents.dtype)
            latents = _AddNoiseDev.apply(image_latents, noise, self.scal)
        with torch.no_grad():
            x = torch.cat([latents] * 2) if cfg else latents
            join, sd._join = sd._join, None
            if join is not None:                                 # the masked-image encode (second stream, see _prepare) ends here
                torch.cuda.current_stream(latents.device).wait_stream(join)
            x = torch.cat([x, mask64, masked_latents], dim=1)
            eps = sd.unet(x.to(sd.precision_t), self.scal[3:4], encoder_hidden_states=emb,
                          cross_attention_kwargs=None, return_dict=False)[0]
            e_u, e_c = eps.chunk(2) if cfg else (eps, None)
            grad = sds_grad_dev(e_u, e_c, noise, self.gs, self.scal, accumulate_into=self.acc)
            if self.mode == 'share':
                return None
            if self.mode == 'single':
                inject = grad * mask64[0, :, :, :]
            else:       # SpecifyGradient is handed the CFG-duplicated [2, 1, 64, 64] mask whole: autograd sums the two copies
                inject = (grad * mask64).sum(0, keepdim=True)
        (d_pred,) = torch.autograd.grad(latents, self.pred, grad_outputs=inject)
        return d_pred

    def run(self, pred, mask, t, acc=None):
        """Replays the step for this image / mask / timestep.  'single': d step / d pred.  'share': the updated running latent
        gradient.  'last': (d step / d pred, updated running latent gradient)."""
        abar = self.sd._alphas_host[t]
        # One graph = one set of static input / output buffers: a run on ANOTHER stream (cal_loss replays the terms on streams
        # of their own) must not rewrite them while the previous replay or its output clones are still in flight.
        # _graph_for keys graphs by stream, so this wait is a no-op in the shipped callers; it keeps any other caller correct.
        cur = torch.cuda.current_stream(self.pred.device)
        if self._done is not None:
            cur.wait_event(self._done)
        # scalars travel as kernel arguments of four fill launches: an asynchronous copy from a temporary host
        # tensor can execute after that tensor's memory has been reused (seen as sporadic garbage timesteps)
        for k, v in enumerate((abar ** 0.5, (1.0 - abar) ** 0.5, 1.0 - abar, float(t))):
            self.scal[k].fill_(v)
        self.pred.data.copy_(pred.detach())
        self.mask.copy_(mask)
        if self.acc is not None:
            self.acc.copy_(acc.reshape(1, 4, 64, 64))
        self.noise.refill(self.sd.generator)                    # this step's draws, eagerly, in the order an eager step makes them
        self.graph.replay()
        if self.mode == 'single':
            out = self.d_pred.clone()
        elif self.mode == 'share':
            out = self.acc.clone()
        else:
            out = (self.d_pred.clone(), self.acc.clone())
        self._done = torch.cuda.Event()
        self._done.record(cur)
        return out


def seed_everything(seed):
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)


class StableDiffusion(nn.Module):
    def __init__(self, device, fp16, vram_O, sd_version='2.1', hf_key=None, t_range=[0.02, 0.98], networks=None,
                 reference_rng=True, use_graphs=None):
        """Signature of DS_NeRF/guidance/sd_utils.py:46 plus two keyword extensions: `networks`
        (an object with .vae, .unet, .encode_prompt(prompt, cfg), .alphas_cumprod; default = the
        SD-1.5-inpaint-shaped modules of sd_nets -- filled from the diffusers-layout checkpoint directory `hf_key`
        when one is given (guidance/sd_checkpoint.py), with random weights otherwise) and `reference_rng`."""
        super().__init__()
        self.device = device
        self.sd_version = sd_version
        self.precision_t = torch.float16 if fp16 else torch.float32
        builtin = networks is None
        if networks is None:
            from .sd_nets import SDNetworks
            networks = SDNetworks(device, self.precision_t)
            if hf_key is not None:
                # the reference resolves hf_key / sd_version to a hub name and downloads it (sd_utils.py:52-74); offline, hf_key is
                # a local diffusers-layout DIRECTORY: loaded strictly (wrong key set or shape -> CheckpointError), fp16-exactness
                # of the weights decides the two-product contractions, CLIP BPE tokenizer when its files are present
                networks.load_checkpoint(hf_key)
        self.networks = networks
        self.vae, self.unet = networks.vae, networks.unet
        self.num_train_timesteps = 1000
        self.min_step = int(self.num_train_timesteps * t_range[0])
        self.max_step = int(self.num_train_timesteps * t_range[1])
        self.alphas = networks.alphas_cumprod.to(device)
        self._alphas_host = [float(a) for a in networks.alphas_cumprod.cpu()]   # no device sync per step
        self.strength = 0.75
        self.reference_rng = reference_rng
        # capture the single-view steps as hipGraphs (same arithmetic; ~1250 launches per step otherwise leave the device
        # idle between kernels: fp32 26.1 -> 25.4 ms, fp16 mode 21.2 -> 17.9 ms, profiles/r3_sds_step_*.json).  Default: on
        # for the built-in networks on the device; injected networks (tests replaying recorded draws through _randn,
        # library modules that may synchronise) keep the eager path unless asked.
        if use_graphs is None:
            import os
            use_graphs = builtin and torch.device(device).type == 'cuda' and os.environ.get('MVIP_SDS_GRAPHS', '1') != '0'
            # (MVIP_SDS_GRAPHS=0: launch-by-launch steps, e.g. under rocprofv3 --pmc, whose per-dispatch counter collection next
            # to replayed graphs of the multi-view step did not finish in 49 minutes on this pool)
            # next to a live multi-rank process group the default stays eager: a capture beside RCCL's own threads and
            # streams has not been exercised on multi-GPU hardware (the build boxes have one GPU).  use_graphs=True or
            # MVIP_GRAPHS_WITH_DIST=1 turn the replay on there too (thread-local capture; the step itself contains no
            # collective -- bench.py reports which mode ran, profiles/r4_bench_2rank_single_device.json has the 2-rank run)
            import torch.distributed as dist
            if (use_graphs and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                    and os.environ.get('MVIP_GRAPHS_WITH_DIST', '0') != '1'):
                use_graphs = False
        self.use_graphs = bool(use_graphs)
        self._graphs = {}
        self.scaling_factor = float(getattr(getattr(self.vae, 'config', None), 'scaling_factor', 0.18215))

    def release_graphs(self):
        """Drop every captured step (and the private memory pool each one holds: the activations of a UNet forward and of the
        VAE encoder's forward + backward at 512^2, GBs per graph).  Graphs are keyed by (mode, shapes, prompt, scale, generator,
        stream) and kept for the life of the object; a caller that moves on to other shapes / prompts / streams -- bench.py between
        its legs -- returns the memory to the NeRF backward's activation stash, whose budget is a fraction of what is free
        (ops._stash_budget).  The next step of a released key captures again."""
        import gc
        self._graphs.clear()
        gc.collect()
        if torch.device(self.device).type == 'cuda':
            torch.cuda.empty_cache()

    # -- hooks (tests replay recorded draws through _randn) --------------------------------------
    generator = None      # optional torch.Generator (default: the device's global generator, like the reference)

    def seed_generator(self, seed):
        """Draw this object's noise from ONE persistent private generator, re-seeded in place.  Multi-GPU runs seed it
        per (iteration, term) so a term's noise does not depend on which rank evaluates it; a captured hipGraph
        registers this generator (`_GraphedStep`), so re-seeding between replays is honoured."""
        if self.__dict__.get('_private_gen') is None:
            self._private_gen = torch.Generator(device=self.device)
        self._private_gen.manual_seed(int(seed))
        self.generator = self._private_gen
        return self._private_gen

    _noise_feed = None    # a _NoiseFeed while a step is being warmed up / captured (see _GraphedStep)

    def _randn(self, shape, dtype=torch.float32):
        if self._noise_feed is not None:
            return self._noise_feed.take(self, shape, dtype)
        return torch.randn(tuple(shape), device=self.device, dtype=dtype, generator=self.generator)

    def _encode_vae_image(self, image):
        """pipeline _encode_vae_image: scaling_factor * posterior.sample()."""
        d = self.vae.encode(image.to(self.precision_t)).latent_dist
        if hasattr(d, 'scaled_sample'):                      # the built-in networks: one launch (and one in the backward)
            n, c2, h, w = d.moments.shape
            return d.scaled_sample(self._randn((n, c2 // 2, h, w), d.moments.dtype), self.scaling_factor)
        return self.scaling_factor * (d.mean + d.std * self._randn(d.mean.shape, d.mean.dtype))

    def _timestep(self, frac):
        return int(self.max_step - (self.max_step - self.min_step) * frac)

    def _prepare(self, pred, mask, prompt, guidance_scale):
        """Steps 0-4 of the reference step methods up to (not including) prepare_latents."""
        latent_size = 512
        pred = _resize_bilinear(pred, (latent_size, latent_size))
        mask = _resize_bilinear(torch.abs(mask), (latent_size, latent_size))
        cfg = guidance_scale > 1.0
        prompt_embeds = self.networks.encode_prompt(prompt, cfg)
        masked_image = pred[:, :3, :, :] * (mask < 0.5)
        init_image = pred[:, :3, :, :]
        # F.interpolate(mask, size=(64, 64)) is mode='nearest': source index floor(dst * 512 / 64) = 8 * dst
        mask64 = mask[:, :, ::8, ::8].contiguous().to(prompt_embeds.dtype)
        # Its only consumer is the UNet input inside torch.no_grad() below (DS_NeRF/guidance/sd_utils.py:375-380: the
        # gradient is applied to `latents` alone), so no gradient ever flows back through this encode: run it without a
        # graph -- same values, no saved activations, and the GroupNorm statistics need not be kept (one launch less each).
        # ... and on a SECOND STREAM (round 5): the two VAE encodes of a step are independent until the UNet's input is
        # assembled, each is ~170 launches, many of them too small to fill the chip -- side by side their kernel boundaries
        # and tails overlap.  Fork here, join in _noise_and_predict right before the concatenation; inside a captured step
        # the fork / join become graph edges.  The random draws keep their order (they are ordered by the host calls).
        # MVIP_SDS_TWO_STREAMS=0: one stream (A/B switch, same values).
        side = self._side_stream() if masked_image.is_cuda else None
        if side is not None:
            cur = torch.cuda.current_stream(masked_image.device)
            side.wait_stream(cur)
            masked_image.record_stream(side)
            with torch.cuda.stream(side), torch.no_grad():
                masked_image_latents = self._encode_vae_image(masked_image)          # randn draw 1
                if cfg:
                    masked_image_latents = torch.cat([masked_image_latents] * 2)
            masked_image_latents.record_stream(cur)
            self._join = side
        else:
            with torch.no_grad():
                masked_image_latents = self._encode_vae_image(masked_image)          # randn draw 1
            if cfg:
                masked_image_latents = torch.cat([masked_image_latents] * 2)
        if cfg:
            mask64 = torch.cat([mask64] * 2)
        if self.reference_rng:
            self._randn((1, 4, latent_size // 8, latent_size // 8))                  # draw 2: the unused encode
        return init_image, mask64, masked_image_latents, prompt_embeds, cfg

    _join = None

    def _side_stream(self):
        """The stream of the masked-image encode (one per StableDiffusion object), or None when switched off."""
        import os
        if os.environ.get('MVIP_SDS_TWO_STREAMS', '1') == '0':
            return None
        if self.__dict__.get('_side') is None:
            from .. import streams as _streams
            self._side = _streams.get(self.device, 'encode')    # shared by every wrapper of the process (streams.py)
        return self._side

    def _noise_and_predict(self, init_image, mask64, masked_image_latents, prompt_embeds, cfg, t, guidance_scale,
                           accumulate_into=None):
        abar = self._alphas_host[t]
        image_latents = self._encode_vae_image(init_image)                           # draw 3 (carries grad)
        noise = self._randn(image_latents.shape, image_latents.dtype)                # draw 4
        latents = _AddNoise.apply(image_latents, noise, abar ** 0.5, (1.0 - abar) ** 0.5)
        with torch.no_grad():
            x = torch.cat([latents] * 2) if cfg else latents
            join, self._join = self._join, None
            if join is not None:                                                     # the masked-image encode (second stream) ends here
                torch.cuda.current_stream(latents.device).wait_stream(join)
            x = torch.cat([x, mask64, masked_image_latents], dim=1)
            noise_pred = self.unet(x.to(self.precision_t), t, encoder_hidden_states=prompt_embeds,
                                   cross_attention_kwargs=None, return_dict=False)[0]
            if cfg:
                e_u, e_c = noise_pred.chunk(2)
            else:
                e_u, e_c = noise_pred, None
            grad = sds_grad(e_u, e_c, noise, guidance_scale, 1.0 - abar, accumulate_into)
        return latents, grad

    def _graph_for(self, mode, mask, prompt, pred, guidance_scale):
        # ... and by the stream it will replay on: the RGB and the normal term use the same mode, and with the reference's defaults
        # (--text == --text_normal, both scales 7.5, normalmap_render_factor 1) the same shapes, prompt and scale -- replayed
        # side by side on two streams (nerf/utils.cal_loss) they must not resolve to ONE graph's static buffers
        stream = torch.cuda.current_stream(pred.device).cuda_stream if pred.is_cuda else 0
        key = (mode, tuple(pred.shape), tuple(mask.shape), prompt, float(guidance_scale), id(self.generator), stream)
        if key not in self._graphs:
            self._graphs[key] = _GraphedStep(self, pred.shape, mask.shape, prompt, guidance_scale, mode)
        return self._graphs[key]

    def _graphed(self, t, mask, prompt, pred, guidance_scale):
        d_pred = self._graph_for('single', mask, prompt, pred, guidance_scale).run(pred, mask, t)
        return _InjectGrad.apply(pred, d_pred)

    # -- the three step methods ---------------------------------------------------------------------
    def train_step_sd(self, i, mask, prompt, pred_rgb, guidance_scale=100, as_latent=False, grad_scale=1,
                      save_guidance_path: Path = None):
        """DS_NeRF/guidance/sd_utils.py:275-429."""
        if self.use_graphs:
            return self._graphed(self._timestep(np.sqrt(i / 20000)), mask, prompt, pred_rgb, guidance_scale)
        prep = self._prepare(pred_rgb, mask, prompt, guidance_scale)
        t = self._timestep(np.sqrt(i / 20000))
        latents, grad = self._noise_and_predict(*prep, t, guidance_scale)
        return SpecifyGradient.apply(latents, grad, prep[1][0, :, :, :])

    def train_step_sd_normal(self, i, mask, prompt, pred_normal_map, guidance_scale=100, normal_start=0,
                             as_latent=False, grad_scale=1, save_guidance_path: Path = None):
        """DS_NeRF/guidance/sd_utils.py:120-272."""
        if self.use_graphs:
            return self._graphed(self._timestep(np.sqrt((i - normal_start) / 20000)), mask, prompt, pred_normal_map,
                                 guidance_scale)
        prep = self._prepare(pred_normal_map, mask, prompt, guidance_scale)
        t = self._timestep(np.sqrt((i - normal_start) / 20000))
        latents, grad = self._noise_and_predict(*prep, t, guidance_scale)
        return SpecifyGradient.apply(latents, grad, prep[1][0, :, :, :])

    def train_step_colla_sds(self, i, mask_nn, prompt, pred_rgb_nn, guidance_scale=100, as_latent=False,
                             grad_scale=1, save_guidance_path: Path = None):
        """DS_NeRF/guidance/sd_utils.py:432-599, reproduced as written: the loop variable shadows `i`
        (so t = 980, 979, ... by view index), `grad` accumulates over views, `loss` is overwritten
        each pass (only the last view receives gradient) and the CFG-duplicated [2,1,64,64] mask is
        passed whole, so autograd sums two copies."""
        NN = pred_rgb_nn.size(0)
        grad = torch.zeros(1, 4, 64, 64, device=self.device)
        loss = None
        for k in range(NN):
            pred_k, mask_k = pred_rgb_nn[k].unsqueeze(0), mask_nn[k].unsqueeze(0)
            t = self._timestep(k / 10000)
            last = k == NN - 1
            if self.use_graphs:
                g = self._graph_for('last' if last else 'share', mask_k, prompt, pred_k, guidance_scale)
                if last:
                    d_pred, grad = g.run(pred_k, mask_k, t, grad)
                    loss = _InjectGrad.apply(pred_k, d_pred)
                else:
                    grad = g.run(pred_k, mask_k, t, grad)
                continue
            # `loss` of a non-final view is overwritten before anything reads it, so no gradient ever flows through that view:
            # it runs without an autograd graph (same draws, same values; no saved activations)
            with torch.set_grad_enabled(last and torch.is_grad_enabled()):
                prep = self._prepare(pred_k, mask_k, prompt, guidance_scale)
                latents, grad = self._noise_and_predict(*prep, t, guidance_scale, accumulate_into=grad)
            if last:
                loss = SpecifyGradient.apply(latents, grad.clone(), prep[1])
        return loss

    # `train_step` exists by name only in the reference's unused guidance/sd.py (:162, :988)
    train_step = train_step_sd

    # -- per-term entry points of the view-sharded multi-GPU path (mvip_nerf_amd/sds_shard.py) ---------------------
    # The reference runs every term on one device (DS_NeRF/nerf/utils.py:280-302); these expose the same arithmetic
    # term by term so that different ranks can own different terms.  `seed` re-seeds the private generator, making a
    # term's noise a function of (iteration, term) instead of the evaluation order.
    def image_grad(self, kind, i, mask, prompt, pred, guidance_scale, weight=1.0, normal_start=0, seed=None):
        """weight * d train_step_sd{,_normal}(...) / d pred for kind in {'rgb', 'normal'}; pred is not touched."""
        if seed is not None:
            self.seed_generator(seed)
        x = pred.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            if kind == 'rgb':
                loss = self.train_step_sd(i, mask, prompt, x, guidance_scale=guidance_scale)
            elif kind == 'normal':
                loss = self.train_step_sd_normal(i, mask, prompt, x, guidance_scale=guidance_scale,
                                                 normal_start=normal_start)
            else:
                raise ValueError(kind)
            (weight * loss).sum().backward()
        return x.grad

    def colla_view_share(self, k, mask_k, prompt, pred_k, guidance_scale, seed=None):
        """Neighbour view k's share nan_to_num(w_k (eps_hat - eps)) of train_step_colla_sds' accumulated latent
        gradient (DS_NeRF/guidance/sd_utils.py:575), forward only; t comes from the VIEW index as in the reference."""
        if seed is not None:
            self.seed_generator(seed)
        if self.use_graphs:
            g = self._graph_for('share', mask_k, prompt, pred_k, guidance_scale)
            return g.run(pred_k, mask_k, self._timestep(k / 10000), torch.zeros(1, 4, 64, 64, device=self.device))
        with torch.no_grad():
            prep = self._prepare(pred_k.detach(), mask_k, prompt, guidance_scale)
            _, grad = self._noise_and_predict(*prep, self._timestep(k / 10000), guidance_scale)
        return grad

    def colla_last_view_image_grad(self, k, mask_k, prompt, pred_k, guidance_scale, share_sum, weight=1.0, seed=None):
        """The LAST neighbour view of train_step_colla_sds: its own share is added to `share_sum` (the other views'
        shares), and the accumulated gradient is injected through SpecifyGradient with the CFG-duplicated mask, as
        the reference does (sd_utils.py:597-599); returns weight * d term / d pred_k."""
        if seed is not None:
            self.seed_generator(seed)
        if self.use_graphs:
            g = self._graph_for('last', mask_k, prompt, pred_k, guidance_scale)
            d_pred, _ = g.run(pred_k, mask_k, self._timestep(k / 10000), share_sum.detach().float())
            return weight * d_pred
        x = pred_k.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            prep = self._prepare(x, mask_k, prompt, guidance_scale)
            acc = share_sum.detach().clone().float().reshape(1, 4, 64, 64).contiguous()
            latents, grad = self._noise_and_predict(*prep, self._timestep(k / 10000), guidance_scale,
                                                    accumulate_into=acc)
            loss = SpecifyGradient.apply(latents, grad.clone(), prep[1])
            (weight * loss).sum().backward()
        return x.grad
