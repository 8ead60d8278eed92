"""A per-leg TIME MODEL of the multi-GPU legs of `bench.py`, written down BEFORE any run on more than one MI355X exists
(VERDICT r4 task 4a), so that the first real N = 2 / 4 / 8 run confirms or refutes it.  Host arithmetic only.

What is modelled (one node, one process per GPU, RCCL over point-to-point xGMI; the traffic replaces the reference's
nn.DataParallel scatter / gather, DS_NeRF/run.py:1131, :1491, :1527):

  render (headline, strong scaling)   one frame's rays in N contiguous blocks (run.render_sharded) + ONE all_gather of
                                      [H*W, 6] fp32.   t(N) = t_frame / N + c_render + all_gather(H*W*24 B)
  train (no prior)                    rays of every set strided over the ranks; ONE gradient bucket of 1,191,688 floats
                                      reduced as two asynchronous halves, the coarse one under the backward
                                      (dist_utils.OverlappedGradBuckets).
                                      t(N) = c_iter + (t_train - c_iter) / N + all_gather(masked colours)
                                             + all_reduce(fine half, exposed)
  iterations with the prior           + the SDS terms, each owned by ONE rank (sds_shard.evaluate, round-robin): phase 1 =
  (configs[1] / [2] / [3])            forward-only "latent" terms, one 64 KB all_reduce; phase 2 = "image" terms, the one
                                      that needs the latent sum after it; <= 3 broadcasts of d term / d image.
                                      The critical path is found by replaying that ownership, not by a formula.

Inputs are the ONE-GPU legs measured by the same `bench.py` (frame, train, SDS step, configs[2] / [3] iterations).
Constants (every one an assumption the SCALE run tests; they are returned with the prediction):
  link_GBps        153    one xGMI link, one direction (task statement: 7 links x ~153 GB/s per GPU)
  link_efficiency  0.7    what a ring step sustains of it at these message sizes (MBs, not GBs)
  alpha_us         12     fixed cost of a collective (launch + first hop), plus
  hop_us           3      per ring step: all_gather N-1 steps, all_reduce 2(N-1), broadcast 1 (direct link)
  c_render_ms      0.3    per-frame work that does not shrink with the block: row generation, launches, the maps' cat
  c_iter_ms        5.0    per-iteration work that does not shrink with the shard: ~10^2 launches at their floor, Adam on
                          1.19 M parameters, the python between them
  fwd_share               a forward-only SDS term as a fraction of a full step, from the FLOP count (2 encoder forwards +
                          UNet over 3 encoder passes + UNet: guidance/flops.py)
"""

DEFAULTS = {'link_GBps': 153.0, 'link_efficiency': 0.7, 'alpha_us': 12.0, 'hop_us': 3.0, 'c_render_ms': 0.3, 'c_iter_ms': 5.0}

GRAD_FLOATS = 1191688            # both 8x256 networks (ops.PARAM_SHAPES x 2)
COARSE_FLOATS = GRAD_FLOATS // 2


def collective_ms(kind, nbytes, n, c=DEFAULTS):
    """Ring all_gather / all_reduce, direct broadcast, over point-to-point links (per-link bound, not switch-bound)."""
    if n <= 1 or nbytes <= 0:
        return 0.0
    bw = c['link_GBps'] * 1e9 * c['link_efficiency']
    if kind == 'all_gather':
        steps, wire = n - 1, (n - 1) / n * nbytes
    elif kind == 'all_reduce':
        steps, wire = 2 * (n - 1), 2 * (n - 1) / n * nbytes
    elif kind == 'broadcast':
        steps, wire = 1, nbytes
    else:
        raise ValueError(kind)
    return (c['alpha_us'] + c['hop_us'] * steps) * 1e-3 + wire / bw * 1e3


def sds_critical_path_ms(terms, n, t_full, t_fwd, image_bytes, c=DEFAULTS):
    """Replay sds_shard.evaluate's ownership for `terms` = list of (phase, needs_latent_sum) in the trainer's order
    (trainer._sds_view_sharded: image terms first): returns the time from the first term's start to the last broadcast."""
    owners = [k % n for k in range(len(terms))]
    t_rank = [0.0] * n
    for k, (phase, _) in enumerate(terms):                       # phase 1: latent shares, forward only
        if phase == 1:
            t_rank[owners[k]] += t_fwd
    has_latent = any(p == 1 for p, _ in terms)
    t_sum = (max(t_rank) + collective_ms('all_reduce', 4 * 64 * 64 * 4, n, c)) if has_latent else 0.0
    order = sorted((k for k, (p, _) in enumerate(terms) if p == 2), key=lambda k: terms[k][1])
    for k in order:                                              # phase 2: image terms; the one that needs the sum waits for it
        r = owners[k]
        start = max(t_rank[r], t_sum) if terms[k][1] else t_rank[r]
        t_rank[r] = start + t_full
    end = max(t_rank + [t_sum])
    n_img = sum(1 for p, _ in terms if p == 2)
    return end + n_img * collective_ms('broadcast', image_bytes, n, c)


def config_terms(config):
    """(phase, needs_latent_sum) per SDS term of BASELINE configs[config] in trainer order (5 neighbour views)."""
    if config == 1:
        return [(2, False)]
    if config == 2:
        return [(2, False), (2, False)]
    if config == 3:
        return [(2, False), (2, False), (2, True)] + [(1, False)] * 4
    raise ValueError(config)


def predict(measured, ns=(2, 4, 8), H=378, W=504, constants=None, fwd_share=None, sds_one_gpu_ms=None):
    """measured: {'frame_ms', 'train_ms', 'sds_ms', 'config2_ms', 'config3_ms'} from ONE GPU (any may be None: its legs are
    skipped).  Returns {'inputs', 'constants', 'N': {n: {...}}} with value (rays/s), strong_efficiency, train_ms,
    train_with_sds_ms, config2_ms, config3_ms and each leg's scaling ceiling (N -> infinity).

    `sds_ms` is the step AS THE MULTI-RANK RUN EXECUTES IT (eager launches next to a live process group unless
    MVIP_GRAPHS_WITH_DIST=1: bench.py's graphs_ok); `sds_one_gpu_ms` (default: the same) is the step as the ONE-GPU
    config legs executed it (hipGraph replay) -- it is what gets subtracted from those legs to isolate their NeRF part."""
    c = dict(DEFAULTS)
    c.update(constants or {})
    if fwd_share is None:
        try:
            from .guidance.flops import sds_step_flops
            fl = sds_step_flops(512)
            fwd_share = (fl['unet_forward'] + 2 * fl['vae_encoder_forward']) / fl['per_step']
        except Exception:                                        # noqa: BLE001 -- the model must not depend on torch's meta device
            fwd_share = 0.775
    c['fwd_share'] = round(float(fwd_share), 4)
    m = {k: (None if measured.get(k) is None else float(measured[k])) for k in ('frame_ms', 'train_ms', 'sds_ms', 'config2_ms', 'config3_ms')}
    rays = H * W
    out = {'inputs': m, 'constants': c, 'N': {},
           'what': 'predicted from the one-GPU legs by mvip_nerf_amd/scaling_model.py (ring collectives over point-to-point xGMI, '
                   'SDS-term ownership replayed); written before any multi-GPU run existed -- SCALE runs test it'}
    t_full = m['sds_ms']
    t_fwd = None if t_full is None else t_full * fwd_share
    t_full_1 = t_full if sds_one_gpu_ms is None else float(sds_one_gpu_ms)
    t_fwd_1 = None if t_full_1 is None else t_full_1 * fwd_share
    out['inputs']['sds_one_gpu_ms'] = t_full_1
    # the NeRF part of an iteration with the prior = the measured iteration minus its terms (one GPU runs them back to back)
    nerf = {}
    if m['train_ms'] is not None:
        nerf[1] = m['train_ms']
    for cfg, key in ((2, 'config2_ms'), (3, 'config3_ms')):
        if m[key] is not None and t_full is not None:
            terms = config_terms(cfg)
            nerf[cfg] = m[key] - sum(t_full_1 if p == 2 else t_fwd_1 for p, _ in terms)
    img_bytes = {1: 3 * H * W * 4, 2: 3 * H * W * 4, 3: 3 * H * W * 4}
    for n in ns:
        row = {}
        if m['frame_ms'] is not None:
            t = m['frame_ms'] / n + c['c_render_ms'] + collective_ms('all_gather', rays * 24, n, c)
            row['ms_per_step'] = round(t, 3)
            row['value'] = round(rays / (t * 1e-3), 1)
            row['strong_efficiency'] = round((m['frame_ms'] + 0.0) / (n * t), 4)
        shard = lambda t1: c['c_iter_ms'] + max(t1 - c['c_iter_ms'], 0.0) / n
        comm_iter = (collective_ms('all_gather', 11544 * 3 * 4, n, c)                       # the masked colours (104 x 111 pixels)
                     + collective_ms('all_reduce', (GRAD_FLOATS - COARSE_FLOATS) * 4, n, c))   # the fine half; the coarse half is hidden
        if 1 in nerf:
            row['train_ms'] = round(shard(nerf[1]) + comm_iter, 2)
            if t_full is not None:
                row['train_with_sds_ms'] = round(shard(nerf[1]) + comm_iter
                                                 + sds_critical_path_ms(config_terms(1), n, t_full, t_fwd, img_bytes[1], c), 2)
        for cfg, key in ((2, 'config2_ms'), (3, 'config3_ms')):
            if cfg in nerf:
                row[key] = round(shard(nerf[cfg]) + comm_iter * (3 if cfg == 3 else 2)      # + the frames' all_gathers
                                 + sds_critical_path_ms(config_terms(cfg), n, t_full, t_fwd, img_bytes[cfg], c), 2)
        out['N'][int(n)] = row
    ceil = {}
    if t_full is not None:
        for cfg, key in ((1, 'train_with_sds_ms'), (2, 'config2_ms'), (3, 'config3_ms')):
            if cfg in nerf:
                t_inf = c['c_iter_ms'] + sds_critical_path_ms(config_terms(cfg), 64, t_full, t_fwd, 0, c)
                t_one = nerf[cfg] + sum(t_full_1 if p == 2 else t_fwd_1 for p, _ in config_terms(cfg))
                ceil[key] = {'ms_at_infinite_ranks': round(t_inf, 2), 'max_speedup': round(t_one / t_inf, 2)}
    out['scaling_ceiling'] = ceil
    return out
