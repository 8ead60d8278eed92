"""bench.py -- throughput of the MVIP-NeRF hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path over one batch of synthetic input: one 378x504 frame
(190,512 rays), coarse 64 + fine 128 samples, test-mode kwargs.
  N = 1: `value` = rays/s of whole frames on the one GPU.
  N > 1: `value` = STRONG-scaling rays/s of the SAME frame sequence -- each frame's rays cut into N contiguous blocks
         (run.render_sharded: rank r renders block r, ONE all_gather of [rays, 6] = (rgb, disp, acc, depth) leaves the
         maps on every rank), `scaling: "strong"`; a scaling curve that CAN fail.  Beside it: `weak_rays_per_sec` (every
         rank renders its own whole frames, no collective: N x by construction), `value_1_same_run` (one rank's
         whole-frame rate from that weak leg) and `strong_efficiency = value / (N * value_1_same_run)`.
Further legs in the same JSON line: the
second-stage training iteration (rays sharded, ONE 4.77 MB gradient all-reduce), the SDS step with its own roofline
(FLOPs counted from the layer shapes), the full BASELINE configs[1]/[2]/[3] iterations (SDS terms owned by different
ranks), and the CPU baseline (oracle on the host cores: render, train and SDS legs, >= 3 warm-ups, median of >= 5).

Launch: with N > 1 and no WORLD_SIZE in the environment this process only SPAWNS the N ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py`, rendezvous on 127.0.0.1) before touching any
GPU, relays rank 0's JSON line and exits with the children's status; under an external launcher (WORLD_SIZE set)
--gpus must equal the world size.  Rank 0 prints ONE JSON line.

Exit status: 0 only if every leg ran.  The legs after the headline measurement exercise the collectives (sharded
training iteration, view-sharded SDS terms, the gradient all-reduce); they run under a watchdog whose deadline lies well
below the RCCL timeout.  If one of them raises or the deadline passes, rank 0 still prints the line -- with what was
measured so far and an `extra_legs_error` -- and every rank leaves with status 3 (a plain exit of the rank process).
The line's `multi_gpu` object records what the process group really was (`rccl_world`, `backend`) and the legs that
do NOT scale by construction: the gradient bucket's all-reduce alone, the sharded iterations' times.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, FOCAL, NEAR, FAR = 378, 504, 383.65, 1.2, 7.74
N_SAMPLES, N_IMPORTANCE = 64, 64
FLOP_PER_POINT = 2 * 593408            # algorithmic MACs of the 8x256 MLP, forward
PEAK_F32_TFLOPS = 157.3                # MI355X_MICROARCH.md: fp32 MFMA / vector peak
PEAK_F16_TFLOPS = 2500.0               # MI355X_MICROARCH.md: fp16 / bf16 dense MFMA peak


def make_args():
    import types
    return types.SimpleNamespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=N_IMPORTANCE,
        alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536,
        lrate=3e-3, basedir='/tmp/mvip_bench', expname='none', ft_path=None, no_reload=True, perturb=1.,
        N_samples=N_SAMPLES, white_bkgd=True, raw_noise_std=1., dataset_type='llff', no_ndc=True, lindisp=True,
        sigma_loss=False, N_rand=1024, chunk=1 << 15, lrate_decay=10, depth_lambda=0.1, sds_loss_weight=1e-4,
        no_coarse=False)


def orbit_pose(k, device):
    th = math.radians(6.0 * (k % 60))
    c, s = math.cos(th), math.sin(th)
    return torch.tensor([[c, 0., s, 0.3 * s], [0., 1., 0., 0.], [-s, 0., c, 0.3 * c]], dtype=torch.float32,
                        device=device)


def _median_time(fn, warmup, reps):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def cpu_baseline(warmup=3, reps=5, threads=None):
    """The oracle (a torch-CPU restatement of the reference: oracle/nerf_oracle.py, oracle/sds_oracle.py) timed on the
    host cores over BOUNDED samples of the same workloads (>= 3 warm-ups, median of >= 5):
      render : n rays of frame 0 through render_rays, no_grad (metric: rays/s);
      train  : one second-stage iteration without the prior on a reduced batch: forward + backward + Adam (rays/s);
      sds    : train_step_sd forward + backward with the SD-1.5-shaped networks at the real 512^2 (latents 64^2, CFG
               batch 2): 1 warm-up, median of 3 (steps/s)."""
    from oracle import nerf_oracle as O
    from oracle import sds_oracle as S
    cores = threads or min(os.cpu_count() or 1, 32)   # more threads than this only adds sync overhead here
    torch.set_num_threads(cores)
    pc, pf = O.mlp_init(0), O.mlp_init(1)
    ro, rd = O.get_rays(H, W, FOCAL, O.bench_poses(1)[0])
    rows = O.assemble_ray_batch(ro, rd, NEAR, FAR)
    n_render = 1024
    r_rows = rows[torch.linspace(0, rows.shape[0] - 1, n_render).long()]

    def render():
        with torch.no_grad():
            O.render_rays(r_rows, pc, pf, N_SAMPLES, N_IMPORTANCE, lindisp=True, white_bkgd=True)
    t_r, all_r = _median_time(render, warmup, reps)

    n_m, n_b = 192, 64                                   # masked set / each supervision batch of the reduced iteration
    g = torch.Generator().manual_seed(1)
    params = [p.requires_grad_(True) for p in list(pc.values()) + list(pf.values())]
    opt = torch.optim.Adam(params, lr=3e-3)
    tgt = torch.rand(n_m + 2 * n_b, 3, generator=g)
    t_rows = rows[torch.randint(0, rows.shape[0], (n_m + 2 * n_b,), generator=g)]

    def train():
        opt.zero_grad(set_to_none=True)
        r = O.render_rays(t_rows, pc, pf, N_SAMPLES, N_IMPORTANCE, lindisp=True, white_bkgd=True,
                          t_rand=torch.rand(t_rows.shape[0], N_SAMPLES, generator=g),
                          u=torch.rand(t_rows.shape[0], N_IMPORTANCE, generator=g))
        a, b = n_m, n_m + n_b
        loss = (1e-4 * O.img2mse(r['rgb_map'][:a], tgt[:a]) + O.img2mse(r['rgb_map'][a:b], tgt[a:b])
                + O.img2mse(r['rgb0'][a:b], tgt[a:b]) + 0.1 * O.img2mse(r['disp_map'][b:], tgt[b:, 0]))
        loss.backward()
        opt.step()
    t_t, all_t = _median_time(train, warmup, reps)
    for p in params:
        p.requires_grad_(False)

    legs = {'render': {'value': n_render / t_r, 'unit': 'rays/s', 'sample': f'{n_render} evenly spaced rays of frame 0, test mode',
                       'seconds': [round(t, 3) for t in all_r]},
            'train': {'value': (n_m + 2 * n_b) / t_t, 'unit': 'rays/s',
                      'sample': f'{n_m} masked + {n_b} colour + {n_b} depth rays, forward + backward + Adam',
                      'seconds': [round(t, 3) for t in all_t]}}
    try:
        from mvip_nerf_amd.guidance.sd_nets import SDNetworks
        nets = SDNetworks(torch.device('cpu'), torch.float32)
        pred = torch.rand(1, 3, H, W, generator=g).requires_grad_(True)
        mask = torch.zeros(1, 1, H, W)
        mask[:, :, (H - 104) // 2:(H - 104) // 2 + 104, (W - 111) // 2:(W - 111) // 2 + 111] = 1

        def sds():
            pred.grad = None
            (1e-4 * S.train_step_sd(nets, 1000, mask, 'a stone bench in a park', pred, guidance_scale=7.5, size=512)).sum().backward()
        t_s, all_s = _median_time(sds, 1, 3)
        legs['sds'] = {'value': 1.0 / t_s, 'unit': 'steps/s',
                       'sample': 'train_step_sd forward + backward at the real size (504x378 -> 512^2, latents 64^2, CFG batch 2), '
                                 'measured, not modelled; 1 warm-up, median of 3',
                       'seconds': [round(t, 3) for t in all_s]}
        del nets
    except Exception as e:                                  # a reported baseline, never fatal for the bench line
        legs['sds'] = {'error': f'{type(e).__name__}: {e}'}
    return {'value': legs['render']['value'], 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': legs['render']['sample'] + f'; {warmup} warm-ups, median of {reps}', 'legs': legs}


def kernel_roofline(run_mod, nets, device, reps=3):
    """Dominant kernel = the fused MLP forward at the fine-pass shape (190,512 rays x 128
    samples): events on the launch stream around `reps` launches."""
    from mvip_nerf_amd import ops
    rows = ops.ray_rows_from_pose(orbit_pose(0, device), H, W, FOCAL, NEAR, FAR)
    z = ops.stratified_z(rows, N_SAMPLES + N_IMPORTANCE, True)
    net = nets['network_fine']
    with torch.no_grad():
        net.query_rays(rows, z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net.query_rays(rows, z)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    points = rows.shape[0] * (N_SAMPLES + N_IMPORTANCE)
    tflops = points * FLOP_PER_POINT / (ms * 1e-3) / 1e12
    traffic, src = None, None
    for name in ('r6_pmc_mlp_forward.json', 'r5_pmc_mlp_forward.json', 'r4_pmc_mlp_forward.json', 'r3_pmc_mlp_forward.json', 'r2_pmc_mlp_forward.json'):     # separate --pmc passes of this launch, newest first
        pmc = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get('hbm_bytes_per_launch')
            src = f'profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per the gfx950 note)'
            break
    return {'bound': 'mfma', 'kernel': 'mlp_forward16_kernel<rays> (csrc/mlp_fwd16.hip)', 'achieved': round(tflops, 2),
            'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tflops / PEAK_F32_TFLOPS, 4),
            'traffic': traffic, 'traffic_unit': 'HBM bytes per launch', 'traffic_source': src,
            'algorithmic_hbm_bytes': points * 20 + rows.shape[0] * 44, 'launch_ms': round(ms, 3),
            'points_per_launch': points, 'flop_per_point': FLOP_PER_POINT}


def spawn_ranks(n, argv):
    """Start n ranks of this script (one per GPU) under torch.distributed.run and relay rank 0's JSON line.
    Runs BEFORE anything touches a GPU in this process, never re-executes this process, and returns the
    launcher's exit status (non-zero if any rank failed or no JSON line came back)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')           # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if proc.returncode != 0:
        return proc.returncode
    return 0 if line is not None else 1


EXIT_LEG_FAILED = 3
RCCL_TIMEOUT_S = 1800            # far above the extra-leg deadline: the watchdog below must fire first


class Reporter:
    """Owns the ONE JSON line and the exit status.  `guarded(fn)` runs the legs that follow the headline measurement:
    under a watchdog THREAD when there is more than one rank (a rank stuck inside a collective sits in a C call that
    never returns to the interpreter, so a signal handler would not run).  Whatever happens, rank 0 prints the line
    exactly once; a failed or timed-out leg makes every rank leave with EXIT_LEG_FAILED.

    A rank that fails tells the others out of band -- a flag file named after the rendezvous port (the ranks share one
    node) -- instead of just exiting: the launcher answers a dead rank by terminating the rest at once, which would
    take rank 0 down inside its collective before it printed.  Rank 0 prints and leaves as soon as it sees the flag (or
    its own failure / deadline); the other ranks leave a few seconds later."""
    GRACE_S = 5.0

    def __init__(self, result, rank, world, before_print=None):
        import tempfile
        import threading
        self.result, self.rank, self.world = result, rank, world
        self.printed = False
        self.before_print = before_print
        self.lock = threading.Lock()
        self.done = threading.Event()
        self.flag = os.path.join(tempfile.gettempdir(), f"mvip_bench_failed_{os.environ.get('MASTER_PORT', 'single')}")
        if rank == 0 and os.path.exists(self.flag):        # a stale flag of an earlier job on this port; every rank passes
            os.remove(self.flag)                            # a barrier after this point before it looks at the flag

    def print_line(self, ok):
        if self.rank != 0 or self.printed:
            return
        self.printed = True
        if ok and self.before_print is not None:
            self.before_print(self.result)
        line = None
        for _ in range(3):
            try:
                line = json.dumps(dict(self.result))
                break
            except RuntimeError:                # the main thread added a key while the watchdog was serialising
                time.sleep(0.05)
        print(line, flush=True)

    def fail(self, why, tell_others=True):
        self.lock.acquire()                     # first caller wins (main thread vs watchdog); it never releases
        self.result['extra_legs_error'] = why
        print(f'[bench] rank {self.rank}: {why}', file=sys.stderr, flush=True)
        if tell_others and self.world > 1:
            try:
                with open(self.flag, 'w') as f:
                    f.write(f'rank {self.rank}: {why}')
            except OSError:
                pass
        if self.rank == 0:
            self.print_line(ok=False)
            sys.stdout.flush()
        else:
            time.sleep(self.GRACE_S)            # rank 0 prints before the launcher tears the job down
        os._exit(EXIT_LEG_FAILED)               # not sys.exit: a stuck collective / other threads must not be joined

    def _watch(self, deadline):
        t_end = time.monotonic() + deadline
        while not self.done.wait(0.5):
            if os.path.exists(self.flag):
                try:
                    who = open(self.flag).read()
                except OSError:
                    who = 'another rank'
                self.fail(f'a leg failed elsewhere ({who}); legs reported so far are complete', tell_others=False)
            if time.monotonic() > t_end:
                self.fail(f'deadline ({deadline:.0f} s): a multi-rank leg did not finish; legs reported so far are complete')

    def guarded(self, fn):
        if self.world > 1:
            import threading
            deadline = float(os.environ.get('MVIP_BENCH_EXTRA_DEADLINE_S', 600))
            threading.Thread(target=self._watch, args=(deadline,), daemon=True).start()
        try:
            fn()
        except Exception as e:
            import traceback
            traceback.print_exc()
            self.fail(f'{type(e).__name__}: {e}')
        self.done.set()
        self.print_line(ok=True)


def add_scaling_prediction(result, world):
    """multi_gpu.predicted: the per-leg time model of mvip_nerf_amd/scaling_model.py evaluated on ONE-GPU
    legs -- this run's own at N = 1 (predictions for N = 2, 4, 8), the newest committed one-GPU line at N > 1 (prediction for
    this N, with measured / predicted beside it): the first real multi-GPU run confirms or refutes the model."""
    from mvip_nerf_amd.scaling_model import predict
    mg = result.setdefault('multi_gpu', {})
    get = lambda d, *ks: (None if d is None else (d.get(ks[0]) if len(ks) == 1 else get(d.get(ks[0]), *ks[1:]))) if isinstance(d, dict) else None
    if world == 1:
        src, line = 'this run', result
    else:
        src, line = None, None
        for name in ('r6_bench_line.json', 'r5_bench_line.json', 'r4_bench_line.json'):
            path = os.path.join(ROOT, 'profiles', name)
            if os.path.exists(path):
                src, line = f'profiles/{name}', json.load(open(path))
                break
        if line is None:
            return
    # The SDS step enters the model in the MODE the multi-rank legs run it in: next to a live process group the step is
    # launched eagerly unless MVIP_GRAPHS_WITH_DIST=1 (extra_legs: graphs_ok), so the eager time is the model's input
    # there, the hipGraph replay only when the multi-rank run will replay too.
    graphs_with_dist = os.environ.get('MVIP_GRAPHS_WITH_DIST', '0') == '1'
    sds_key = 'ms_per_step' if graphs_with_dist else 'ms_per_step_eager'
    sds_ms = get(line, 'sds', sds_key) or get(line, 'sds', 'ms_per_step')

    def cfg_ms(name, mode):
        v = get(line, name, mode, 'ms_per_step')
        if v is None and mode == 'f16x3' and get(line, name, 'f32') is None:
            v = get(line, name, 'ms_per_step')           # lines of rounds <= 5 printed only the split-precision leg
        return v
    measured = {'frame_ms': get(line, 'ms_per_step'), 'train_ms': get(line, 'train', 'ms_per_step'), 'sds_ms': sds_ms,
                'config2_ms': cfg_ms('config2_rgb_normal_sds', 'f32'), 'config3_ms': cfg_ms('config3_rgb_normal_colla_sds', 'f32')}
    measured_split = {'frame_ms': get(line, 'render_f16x3', 'ms_per_step'), 'train_ms': get(line, 'train_f16x3', 'ms_per_step'),
                      'sds_ms': sds_ms, 'config2_ms': cfg_ms('config2_rgb_normal_sds', 'f16x3'),
                      'config3_ms': cfg_ms('config3_rgb_normal_colla_sds', 'f16x3')}
    if measured['frame_ms'] is None:
        return
    ns = (2, 4, 8) if world == 1 else (world,)
    # on one GPU the NeRF part of a measured iteration = iteration - its terms AS THAT RUN EXECUTED THEM (graph replays)
    sds_one_gpu = get(line, 'sds', 'ms_per_step')
    pred = predict(measured, ns=ns, sds_one_gpu_ms=sds_one_gpu)
    pred['dtype'] = 'f32 NeRF kernels (the default path)'
    pred['sds_mode'] = ('hipGraph replay (MVIP_GRAPHS_WITH_DIST=1)' if graphs_with_dist else
                        'eager launches (what bench.py runs next to a live multi-rank process group)')
    pred['one_gpu_legs_from'] = src
    if measured_split['frame_ms'] is not None:
        ps = predict(measured_split, ns=ns, sds_one_gpu_ms=sds_one_gpu)
        ps['dtype'] = 'f16x3 NeRF kernels (opt-in split precision)'
        ps['sds_mode'] = pred['sds_mode']
        mg['predicted_f16x3'] = ps
    if world > 1:
        row = pred['N'].get(world, {})
        got = {'ms_per_step': result.get('ms_per_step'), 'value': result.get('value'), 'strong_efficiency': result.get('strong_efficiency'),
               'train_ms': mg.get('train_ms'), 'train_with_sds_ms': mg.get('train_with_sds_ms'),
               'config2_ms': mg.get('config2_ms'), 'config3_ms': mg.get('config3_ms')}
        pred['measured'] = got
        pred['measured_over_predicted'] = {k: round(got[k] / row[k], 3) for k in row if got.get(k) and row.get(k)}
        if 'predicted_f16x3' in mg:
            rs = mg['predicted_f16x3']['N'].get(world, {})
            gs = {'ms_per_step': get(result, 'render_f16x3', 'ms_per_step'), 'train_ms': mg.get('train_f16x3_ms'),
                  'train_with_sds_ms': mg.get('train_with_sds_f16x3_ms'), 'config2_ms': mg.get('config2_f16x3_ms'),
                  'config3_ms': mg.get('config3_f16x3_ms')}
            mg['predicted_f16x3']['measured'] = gs
            mg['predicted_f16x3']['measured_over_predicted'] = {k: round(gs[k] / rs[k], 3) for k in rs if gs.get(k) and rs.get(k)}
    mg['predicted'] = pred


def dry_run(world, rank, args):
    """MVIP_BENCH_DRYRUN=1: the launch / rendezvous / watchdog path without GPU work (CPU tests of the launcher).
    MVIP_BENCH_DRYRUN_HANG_RANK=r makes rank r sleep instead of joining the extra leg's collective."""
    import torch.distributed as dist
    result = {}
    rep = Reporter(result, rank, world)          # clears a stale failure flag BEFORE the first collective
    if world > 1:
        dist.init_process_group('gloo')
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        assert float(t) == world * (world - 1) / 2
    # the strong-scaling headline's collective on CPU tensors: a 7 x 5 "frame" of ray rows rendered by a stand-in
    from mvip_nerf_amd import run as run_mod
    Hd, Wd = 7, 5
    rows_all = torch.arange(Hd * Wd * 11, dtype=torch.float32).reshape(Hd * Wd, 11)

    def fake_render_rays(ray_batch, **kw):
        v = ray_batch[:, 0]
        return {'rgb_map': torch.stack([v, v + 1, v + 2], -1), 'disp_map': v * 2, 'acc_map': v * 3, 'depth_map': v * 4}
    real_rr, run_mod.render_rays = run_mod.render_rays, fake_render_rays
    try:
        maps = run_mod.render_sharded(Hd, Wd, 1.0, None, rank, world, dist if world > 1 else None, chunk=4,
                                      row_fn=lambda lo, hi: rows_all[lo:hi])
    finally:
        run_mod.render_rays = real_rr
    v = rows_all[:, 0]
    assert torch.equal(maps[0], torch.stack([v, v + 1, v + 2], -1).reshape(Hd, Wd, 3)) and torch.equal(maps[3], (v * 4).reshape(Hd, Wd))
    result.update({'metric': 'dry-run (launcher only)', 'value': 0.0, 'unit': 'rays/s', 'n_gpus': world,
              'steps': args.steps, 'warmup': args.warmup, 'scaling': 'strong' if world > 1 else 'weak',
              'weak_rays_per_sec': 0.0, 'strong_efficiency': None, 'sharded_frame_assembled': True,
              'multi_gpu': {'rccl_world': dist.get_world_size() if world > 1 else 1,
                            'backend': dist.get_backend() if world > 1 else None}})

    def extra():
        if os.environ.get('MVIP_BENCH_DRYRUN_HANG_RANK') == str(rank):
            time.sleep(3600)
        if os.environ.get('MVIP_BENCH_DRYRUN_RAISE_RANK') == str(rank):
            raise RuntimeError('simulated leg failure')
        if world > 1:
            t = torch.tensor([1.0])
            dist.all_reduce(t)
        result['extra_leg'] = 'done'
    rep.guarded(extra)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--train-steps', type=int, default=3, help='second-stage iterations timed after the render leg')
    ap.add_argument('--sds-steps', type=int, default=5, help='SDS / full-iteration steps timed in the third leg')
    ap.add_argument('--no-hashgrid', dest='hashgrid', action='store_false')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-warmup', type=int, default=3)
    ap.add_argument('--cpu-reps', type=int, default=5)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    if os.environ.get('MVIP_BENCH_DRYRUN') == '1':
        return dry_run(world, rank, args)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    # debug hook for 1-GPU boxes: all ranks on device 0 over gloo (never set by the driver)
    single = os.environ.get('MVIP_BENCH_SINGLE_DEVICE') == '1'
    if single:
        local_rank = 0
    backend = os.environ.get('MVIP_DIST_BACKEND', 'gloo' if single else 'nccl')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    result = {}
    rep = Reporter(result, rank, world)          # clears a stale failure flag BEFORE the first collective / barrier
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            import datetime
            dist.init_process_group('nccl', device_id=device, timeout=datetime.timedelta(seconds=RCCL_TIMEOUT_S))   # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from mvip_nerf_amd import run
    torch.manual_seed(0)
    tr, te, start, grad_vars, opt = run.create_nerf(make_args(), device=device)

    def step(k):                                    # one WHOLE frame on this rank (N = 1 headline; the weak leg at N > 1)
        with torch.no_grad():
            out = run.render(H, W, FOCAL, chunk=1 << 15, c2w=orbit_pose(k * world + rank, device), near=NEAR,
                             far=FAR, **te)
        return out[0]

    def step_strong(k):                             # ONE frame over all ranks: rays in contiguous blocks, maps all-gathered
        with torch.no_grad():
            return run.render_sharded(H, W, FOCAL, orbit_pose(k, device), rank, world, dist, chunk=1 << 15, near=NEAR,
                                      far=FAR, **te)[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize; max over ranks."""
        for k in range(args.warmup):
            fn(k)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            fn(args.warmup + k)
        barrier()
        dt_ = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt_], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_

    rays_per_step = H * W * world                    # of the weak form (one frame per rank)
    workload = ('render 378x504 frame, 64 coarse + 128 fine samples, 8x256 MLP x2, lindisp, '
                'white_bkgd, test-mode kwargs (BASELINE configs[1] geometry at the metric resolution)')
    if world == 1:
        dt = timed(step)
        value, scaling, extra = H * W * args.steps / dt, 'weak', {}
        par = 'rays x1 (whole frames on one GPU)'
    else:
        dt_weak = timed(step)                        # every rank its own frames, no collective: N x by construction
        dt = timed(step_strong)                      # the headline at N > 1: fixed work, one frame over all ranks
        value, scaling = H * W * args.steps / dt, 'strong'
        # The one-GPU baseline of this same run, measured with the OTHER ranks idle (they wait at the barrier): rank 0 renders
        # whole frames alone, so neither a shared device nor contention for host / fabric halves it.
        for k in range(args.warmup):
            if rank == 0:
                step(k)
        barrier()
        t0 = time.perf_counter()
        if rank == 0:
            for k in range(args.steps):
                step(args.warmup + k)
        barrier()
        dt_solo = time.perf_counter() - t0
        t = torch.tensor([dt_solo], device=device, dtype=torch.float64)
        dist.broadcast(t, 0)
        dt_solo = float(t.item())
        v1 = H * W * args.steps / dt_solo
        shared_device = os.environ.get('MVIP_BENCH_SINGLE_DEVICE') == '1'
        extra = {'weak_rays_per_sec': rays_per_step * args.steps / dt_weak, 'weak_ms_per_step': dt_weak / args.steps * 1e3,
                 'value_1_same_run': v1,
                 # ranks that time-share ONE device scale nothing: no efficiency is claimed for such a run
                 'strong_efficiency': None if shared_device else value / (world * v1),
                 'ranks_share_one_device': shared_device,
                 'strong_what': 'each frame cut into N contiguous ray blocks (run.render_sharded), ONE all_gather of '
                                f'[{H * W}, 6] fp32 = {H * W * 24 / 1e6:.2f} MB per frame so every rank holds rgb / disp / acc / depth; '
                                'value_1_same_run = whole frames on rank 0 ALONE (the other ranks idle at a barrier)'}
        par = (f'one frame over {world} ranks: contiguous ray blocks of {-(-H * W // world)} rays, one all_gather of the maps '
               f'per frame ({"RCCL over xGMI" if backend == "nccl" else backend})')
    result.update({
        'metric': 'rays_per_sec (coarse+fine, 64+128 samples, 504x378)', 'value': value,
        'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': workload,
                   'rays_per_step_per_gpu': H * W if world == 1 else -(-H * W // world), 'points_per_ray': 192, 'chunk': 1 << 15,
                   'parallelism': par},
    })
    result.update(extra)
    mg = {'rccl_world': dist.get_world_size() if dist is not None else 1,
          'backend': dist.get_backend() if dist is not None else None}
    result['multi_gpu'] = mg
    if rank == 0:
        result['roofline'] = kernel_roofline(run, te, device)          # local to rank 0: no collective

    def add_cpu_baseline(res):
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args.cpu_warmup, args.cpu_reps)
    rep.before_print = add_cpu_baseline

    def extra_legs():
        # ---- the gradient bucket's all-reduce ALONE (1,191,688 floats = 4.77 MB, the only per-iteration data-path
        #      collective of the ray-sharded training step), 20 repetitions between barriers
        if dist is not None:
            bucket = torch.zeros(sum(p.numel() for p in grad_vars), device=device, dtype=torch.float32)
            for _ in range(3):
                dist.all_reduce(bucket)
            barrier()
            ta = time.perf_counter()
            for _ in range(20):
                dist.all_reduce(bucket)
            barrier()
            t = torch.tensor([(time.perf_counter() - ta) / 20], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mg['allreduce_ms'] = float(t.item()) * 1e3
            mg['allreduce_bytes'] = bucket.numel() * 4
            del bucket
        # ---- extra leg: the reference's second model (hash grid + tiny MLPs, the shipped config's `no_tcnn = False`),
        # same frames / same training iteration; SURVEY.md 8(f) row 4
        if args.hashgrid:
            from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene
            a_h = make_args()
            a_h.no_tcnn, a_h.netchunk, a_h.lrate = False, 1 << 20, 1e-2
            scene_h = SyntheticScene(H, W, FOCAL, NEAR, FAR, device=device)
            tr_h = SecondStageTrainer(a_h, scene_h, device, guidance=None, world=world, rank=rank, dist=dist)

            def render_h(k):
                with torch.no_grad():
                    return run.render(H, W, FOCAL, chunk=1 << 15, c2w=orbit_pose(rank * 7 + k, device), near=NEAR, far=FAR,
                                      **tr_h.kw_test)
            render_h(0)
            barrier()
            th = time.perf_counter()
            for k in range(args.steps):
                render_h(k + 1)
            barrier()
            dt_h = time.perf_counter() - th
            tr_h.step(0)
            barrier()
            th = time.perf_counter()
            for k in range(args.train_steps):
                tr_h.step(1 + k)
            barrier()
            dt_ht = time.perf_counter() - th
            if dist is not None:
                t = torch.tensor([dt_h, dt_ht], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_h, dt_ht = float(t[0]), float(t[1])
            result['hashgrid_model'] = {
                'render_rays_per_sec': H * W * world * args.steps / dt_h, 'render_ms_per_frame': dt_h / args.steps * 1e3,
                'train_ms_per_step': dt_ht / max(args.train_steps, 1) * 1e3,
                'what': 'NeRF_TCNN (16-level hash grid + 64-wide MLPs, coarse+fine), same frames and same training '
                        'iteration as the 8x256 legs; renders through the fused gather + fp32-MFMA kernel '
                        '(csrc/hashgrid_fused.hip); parity unpinned (tiny-cuda-nn absent)'}
            del tr_h, scene_h
            torch.cuda.empty_cache()       # the hash-grid leg's 1M-point chunks leave ~40 GB of cached blocks behind

        # ---- extra leg: the same frames with the split-precision forward (precision = 1, "f16x3": fp16 MFMA on
        #      hi/lo splits of both operands, fp32 accumulate).  Reported separately; `value` stays exact fp32. ----
        with torch.no_grad():
            ref_img = step(0)
            for net in (te['network_fn'], te['network_fine']):
                net.inference_precision = 1
            fast_img = step(0)
            barrier()
            tf = time.perf_counter()
            for k in range(args.steps):
                step(args.warmup + k)
            barrier()
            dt_fast = time.perf_counter() - tf
            for net in (te['network_fn'], te['network_fine']):
                net.inference_precision = 0
        if dist is not None:
            t = torch.tensor([dt_fast], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_fast = float(t.item())
        mse_fast = float(((fast_img - ref_img) ** 2).mean())
        result['render_f16x3'] = {
            'rays_per_sec': rays_per_step * args.steps / dt_fast, 'ms_per_step': dt_fast / args.steps * 1e3,
            'dtype': 'f16x3 (fp16 MFMA, both operands split hi+lo, 3 products, fp32 accumulate)',
            'kernel': 'mvip::f16h::mlp_forward_f16x3_w16_kernel (csrc/mlp_fwd16_f16x3.hip: 16 points per wave on v_mfma_f32_16x16x32_f16, '
                      'two waves per SIMD; round 4 ran the 32-point one-wave kernel at 102-104 ms per frame)',
            'fp16_product_TFLOPs': round(3 * rays_per_step * args.steps * 192 * FLOP_PER_POINT / dt_fast / 1e12, 1),
            'psnr_vs_f32_render_dB': -10 * math.log10(max(mse_fast, 1e-30)),
            'max_abs_pixel_diff': float((fast_img - ref_img).abs().max())}

        # ---- second leg: the training iteration (masked render + 2 supervision batches, fwd+bwd+Adam) ----
        if args.train_steps > 0:
            from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene
            torch.manual_seed(1)
            scene = SyntheticScene(H, W, FOCAL, NEAR, FAR, device=device)
            trainer = SecondStageTrainer(make_args(), scene, device, guidance=None, world=world, rank=rank, dist=dist)
            # three untimed iterations: after the hash-grid leg's empty_cache() the first iterations rebuild the allocator's
            # block pool (GB-sized stash buffers: hipMalloc calls of milliseconds each) -- with one warm-up the three timed
            # steps read 54 ... 70 ms on the same box for the same kernels (device time 54.4 ms throughout)
            for w in range(3):
                trainer.step(w)
            barrier()
            t1 = time.perf_counter()
            n_rays = 0
            for k in range(args.train_steps):
                _, nr = trainer.step(10 + k)
                n_rays += nr
            barrier()
            dt_tr = time.perf_counter() - t1
            if dist is not None:
                t = torch.tensor([dt_tr], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_tr = float(t.item())
            # the same iteration with the split-precision kernels (train_precision = 1: f16x3 forward, delta and
            # weight-gradient kernels; same stash, same atomics, fp32 accumulation everywhere)
            for n in (trainer.kw_train['network_fn'], trainer.kw_train['network_fine']):
                n.train_precision = 1
            trainer.step(100)
            barrier()
            t1b = time.perf_counter()
            for k in range(args.train_steps):
                trainer.step(101 + k)
            barrier()
            dt_tr16 = time.perf_counter() - t1b
            for n in (trainer.kw_train['network_fn'], trainer.kw_train['network_fine']):
                n.train_precision = 0
            if dist is not None:
                t = torch.tensor([dt_tr16], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_tr16 = float(t.item())
            mg['train_f16x3_ms'] = dt_tr16 / args.train_steps * 1e3
            result['train_f16x3'] = {'rays_per_sec': n_rays * world / dt_tr16, 'ms_per_step': dt_tr16 / args.train_steps * 1e3,
                                     'dtype': 'f16x3 (fp16 MFMA, both operands split hi+lo, 3 products, fp32 accumulate)',
                                     'what': 'the same iteration with train_precision=1 (split-precision MFMA kernels)'}
            mg['train_ms'] = dt_tr / args.train_steps * 1e3
            result['train'] = {'rays_per_sec': n_rays * world / dt_tr, 'ms_per_step': dt_tr / args.train_steps * 1e3, 'dtype': 'f32',
                               'steps': args.train_steps, 'rays_per_step': n_rays * world // args.train_steps,
                               'what': 'second-stage iteration without the diffusion prior: masked-set render '
                                       '(11,544 rays) + 1024 colour rays + 1024 depth rays, losses, backward through '
                                       'both MLPs, gradient all-reduce, Adam'}
        # ---- third leg: the diffusion prior (BASELINE configs[1]: RGB SDS only) ----
        if args.sds_steps > 0:
            import types
            from mvip_nerf_amd.guidance.sd_utils import StableDiffusion
            from mvip_nerf_amd.nerf.utils import Pretrain_Model
            from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene
            sd = StableDiffusion(device, False, False)                  # SD-1.5-inpaint shapes, random weights, fp32
            g2 = torch.Generator(device=device).manual_seed(2)
            pred = torch.rand(1, 3, H, W, device=device, generator=g2).requires_grad_(True)
            mask = torch.zeros(1, 1, H, W, device=device)
            mask[:, :, (H - 104) // 2:(H - 104) // 2 + 104, (W - 111) // 2:(W - 111) // 2 + 111] = 1

            def timed_steps(sd_x, scale, use_graphs, n):
                """median wall time of n train_step_sd forward + backward steps (per-step barrier), after 2 warm-ups"""
                sd_x.use_graphs = use_graphs
                pg = pred.detach().clone().requires_grad_(True)

                def one(i):
                    pg.grad = None
                    (scale * sd_x.train_step_sd(i, mask, 'a stone bench in a park', pg, guidance_scale=7.5)).sum().backward()
                one(1000)
                one(1001)
                barrier()
                ts = []
                for k in range(n):
                    tg = time.perf_counter()
                    one(1002 + k)
                    barrier()
                    ts.append(time.perf_counter() - tg)
                return ts
            # the step as the product runs it by default: fp32 networks (split-precision MFMA kernels), ONE captured hipGraph
            # (StableDiffusion.use_graphs defaults to True for the built-in networks; thread-local capture, so a live RCCL
            # communicator's watchdog thread does not disturb it) -- and the same step launch by launch
            # ---- BASELINE configs[2] / configs[3] FIRST: their SDS terms replay on term streams, and a captured step that has been
            #      replayed from the DEFAULT stream -- which is where the one-term legs below run, because that is faster for them --
            #      would cost these legs' term streams their concurrency for the rest of the process (guidance/sd_utils._OffDefaultStream,
            #      profiles/r6_stream_experiments.json: 152 instead of 144.6 ms for configs[2] f16x3, 280 instead of 273 in f32, when these
            #      legs ran last; in this order every leg reads what the same iteration reads in a process of its own,
            #      tools/config_step_profile.py).  A training run is one configuration; only this script is several.
            import types
            from mvip_nerf_amd.nerf.utils import Pretrain_Model
            from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene
            graphs_ok = world == 1 or os.environ.get('MVIP_GRAPHS_WITH_DIST', '0') == '1'    # eager next to a live multi-rank group unless asked
            sd.use_graphs = graphs_ok
            opt = types.SimpleNamespace(is_rgb_guidance=True, is_colla_guidance=False, is_normal_guidance=False,
                                        text='a stone bench in a park', text_normal='', rgb_guidance_scale=7.5,
                                        colla_guidance_scale=7.5, normal_guidance_scale=1.5, normal_start=500,
                                        lambda_guidance=1)
            scene = SyntheticScene(H, W, FOCAL, NEAR, FAR, device=device)
            # BASELINE configs[2] (RGB + normal SDS, normalmap_render_factor=2) and configs[3] (+ multi-view
            # collaborative SDS over <=5 neighbour views), EACH IN BOTH ARITHMETICS of the NeRF kernels: "f32" = the package
            # default (train_precision = inference_precision = 0, exact fp32 MFMA; what every config-level test runs and
            # what the reference computes in) and "f16x3" = the opt-in split-precision mode (fp16 MFMA on hi + lo halves,
            # three products, fp32 accumulate; config-level parity: tests/test_configs.py::test_split_precision_*).
            # The top-level ms_per_step of each leg is the f32 one.
            for name, colla, nsteps in (('config3_rgb_normal_colla_sds', True, 2), ('config2_rgb_normal_sds', False, args.sds_steps)):
                a2 = make_args()
                a2.is_normal_guidance, a2.is_colla_guidance, a2.normalmap_render_factor = True, colla, 2
                opt.is_normal_guidance, opt.is_colla_guidance, opt.normal_start = True, colla, 500
                opt.text_normal = 'a normal map of a stone bench in a park'
                legs = {}
                for mode, prec in (('f32', 0), ('f16x3', 1)):
                    # a trainer and an allocator pool of its own per leg: the legs must not time each other's leftovers
                    tr2 = None
                    torch.cuda.empty_cache()
                    if os.environ.get('MVIP_BENCH_PER_STEP') == '1':
                        fr, tot = torch.cuda.mem_get_info(device)
                        print(f'[bench] before {name} {mode}: allocated {torch.cuda.memory_allocated(device) / 2**30:.1f} GiB, reserved '
                              f'{torch.cuda.memory_reserved(device) / 2**30:.1f} GiB, driver-free {fr / 2**30:.1f} of {tot / 2**30:.1f} GiB', file=sys.stderr)
                    tr2 = SecondStageTrainer(a2, scene, device, guidance=Pretrain_Model(opt, device, {'SD': sd}), world=world,
                                             rank=rank, dist=dist)
                    for n in (tr2.kw_train['network_fn'], tr2.kw_train['network_fine']):
                        n.train_precision = n.inference_precision = prec
                    tr2.step(999)
                    tr2.step(1000)
                    barrier()
                    t5 = time.perf_counter()
                    rays = 0
                    _per = []
                    for k in range(nsteps):
                        _t = time.perf_counter()
                        rays += tr2.step(1001 + k)[1]
                        if os.environ.get('MVIP_BENCH_PER_STEP') == '1':        # diagnostic: per-step wall times (adds a sync per step)
                            torch.cuda.synchronize()
                            _per.append(round((time.perf_counter() - _t) * 1e3, 1))
                    barrier()
                    dt5 = time.perf_counter() - t5
                    if _per:
                        print(f'[bench] {name} {mode} per-step ms {_per}', file=sys.stderr)
                    if os.environ.get('MVIP_BENCH_PROFILE_LEG') == f'{name}:{mode}':     # diagnostic: one step of this leg per kernel
                        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                            tr2.step(1100)
                            torch.cuda.synchronize()
                        ev = sorted([e for e in prof.key_averages() if e.device_time_total > 0], key=lambda e: -e.device_time_total)
                        print(f'[bench] {name} {mode}: device-busy {sum(e.device_time_total for e in ev) / 1e3:.1f} ms, {sum(e.count for e in ev)} launches', file=sys.stderr)
                        for e in ev[:16]:
                            print(f'[bench]   {e.device_time_total / 1e3:8.3f} ms x{e.count:4d}  {e.key[:100]}', file=sys.stderr)
                    if dist is not None:
                        t = torch.tensor([dt5], device=device, dtype=torch.float64)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        dt5 = float(t.item())
                    legs[mode] = {'ms_per_step': dt5 / nsteps * 1e3, 'steps': nsteps,
                                  'dtype': ('f32 (exact fp32 MFMA NeRF kernels: the package default)' if prec == 0 else
                                            'f16x3 (NeRF kernels on fp16 MFMA, both operands split hi+lo, 3 products, fp32 accumulate; '
                                            'opt-in train_precision = inference_precision = 1)'),
                                  'sds_dtype': 'f32 tensors, contractions in split precision (see sds.dtype)'}
                    mg[('config3' if colla else 'config2') + ('_ms' if prec == 0 else '_f16x3_ms')] = dt5 / nsteps * 1e3
                result[name] = {'ms_per_step': legs['f32']['ms_per_step'], 'dtype': legs['f32']['dtype'],
                                'f32': legs['f32'], 'f16x3': legs['f16x3'],
                                'rays_with_grad_per_step_per_gpu': rays // nsteps,
                                'sds_evaluations_per_step': 2 + (5 if colla else 0)}
                del tr2
            opt.is_normal_guidance = opt.is_colla_guidance = False
            opt.text_normal = ''
            n_sds = max(args.sds_steps, 3)
            graphs_ok = world == 1 or os.environ.get('MVIP_GRAPHS_WITH_DIST', '0') == '1'    # eager next to a live multi-rank group unless asked
            sds_times = timed_steps(sd, 1e-4, graphs_ok, n_sds)
            dt_sds = float(np.median(sds_times)) * args.sds_steps
            ms_eager32 = float(np.median(timed_steps(sd, 1e-4, False, n_sds))) * 1e3
            # the reference's --fp16 mode (DS_NeRF/guidance/sd_utils.py:66) on the SAME hand-written kernels in their
            # single-product instantiations (round 3; it used to fall onto MIOpen / CK / AOTriton kernels)
            ms_graph16 = ms_eager16 = None
            if world == 1:
                try:
                    sd16 = StableDiffusion(device, True, False)
                    ms_graph16 = float(np.median(timed_steps(sd16, 1.0, True, n_sds))) * 1e3
                    ms_eager16 = float(np.median(timed_steps(sd16, 1.0, False, n_sds))) * 1e3
                    del sd16
                except Exception as e:                            # reported, never fatal for the bench line
                    print(f'[bench] fp16-mode leg skipped: {type(e).__name__}: {e}', file=sys.stderr)
                torch.cuda.empty_cache()
            sd.use_graphs = graphs_ok
            opt = types.SimpleNamespace(is_rgb_guidance=True, is_colla_guidance=False, is_normal_guidance=False,
                                        text='a stone bench in a park', text_normal='', rgb_guidance_scale=7.5,
                                        colla_guidance_scale=7.5, normal_guidance_scale=1.5, normal_start=500,
                                        lambda_guidance=1)
            scene = SyntheticScene(H, W, FOCAL, NEAR, FAR, device=device)
            full = SecondStageTrainer(make_args(), scene, device, guidance=Pretrain_Model(opt, device, {'SD': sd}),
                                      world=world, rank=rank, dist=dist)
            full.step(999)                 # two untimed iterations (allocator pool after the SDS leg's empty_cache)
            full.step(1000)
            barrier()
            t3 = time.perf_counter()
            for k in range(args.sds_steps):
                full.step(1001 + k)
            barrier()
            dt_full = time.perf_counter() - t3
            for n in (full.kw_train['network_fn'], full.kw_train['network_fine']):
                n.train_precision = 1
            full.step(2000)
            barrier()
            t4 = time.perf_counter()
            for k in range(args.sds_steps):
                full.step(2001 + k)
            barrier()
            dt_full16 = time.perf_counter() - t4
            if dist is not None:
                t = torch.tensor([dt_sds, dt_full, dt_full16], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_sds, dt_full, dt_full16 = float(t[0]), float(t[1]), float(t[2])
            opt.is_normal_guidance = opt.is_colla_guidance = False
            mg['train_with_sds_f16x3_ms'] = dt_full16 / args.sds_steps * 1e3
            result['train_with_sds_f16x3'] = {'ms_per_step': dt_full16 / args.sds_steps * 1e3,
                                              'dtype': 'f16x3 NeRF kernels (3 fp16 products, fp32 accumulate); SDS as in sds.dtype',
                                              'iterations_per_sec': args.sds_steps / dt_full16,
                                              'what': 'the same iteration with train_precision=1 for the NeRF kernels'}
            from mvip_nerf_amd.guidance.flops import sds_step_flops
            fl = sds_step_flops(512)
            sds_ms = dt_sds / args.sds_steps * 1e3
            ach = fl['per_step'] / (sds_ms * 1e-3) / 1e12
            traffic, traffic_src = None, None
            for name in ('r6_pmc_sds_traffic.json', 'r5_pmc_sds_traffic.json', 'r4_pmc_sds_traffic.json', 'r3_pmc_sds_traffic.json'):      # newest first
                pmc_sds = os.path.join(ROOT, 'profiles', name)
                if os.path.exists(pmc_sds):
                    traffic = json.load(open(pmc_sds)).get('hbm_bytes_per_step')
                    traffic_src = f'profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over eager steps, per kernel)'
                    break
            # The ceiling of the work the step ISSUES: a contraction with an fp16-exact weight operand runs two fp16 products per
            # algorithmic multiply-add (ops.TWO_PRODUCT), activation x activation products (attention) three, so the
            # fp32-equivalent peak is 2500 x FLOPs / (2 x weight FLOPs + 3 x attention FLOPs); three everywhere otherwise.
            from mvip_nerf_amd import ops as _ops
            two_prod = bool(getattr(sd.networks, 'fp16_weights', False)) and bool(_ops.TWO_PRODUCT)
            fl_w = fl_a = 0
            for net, mult in (('unet', 1), ('vae_encoder', 3)):
                fl_w += mult * (fl['breakdown'][net]['conv'] + fl['breakdown'][net]['linear'])
                fl_a += mult * fl['breakdown'][net]['attention']
            products_per_flop = ((2 if two_prod else 3) * fl_w + 3 * fl_a) / max(fl_w + fl_a, 1)
            peak_issued = PEAK_F16_TFLOPS / products_per_flop
            sds_roof = {'bound': 'mfma', 'flops_per_step': fl['per_step'], 'composition': fl['composition'],
                        'algorithmic_hbm_bytes': fl['bytes_per_step'], 'traffic': traffic, 'traffic_unit': 'HBM bytes per step',
                        'traffic_source': traffic_src,
                        'unet_forward_flops': fl['unet_forward'], 'vae_encoder_forward_flops': fl['vae_encoder_forward'],
                        'achieved': round(ach, 1), 'unit': 'TFLOP/s (fp32-equivalent: algorithmic FLOPs of the step / median step time)',
                        'peak': round(peak_issued, 1),
                        'products_per_algorithmic_flop': round(products_per_flop, 4),
                        'peak_is': 'fp16 dense MFMA 2500 TFLOP/s / the fp16 products ISSUED per algorithmic multiply-add, weighted over the '
                                   'step: two for contractions whose weight operand is an exact fp16 value (every convolution and linear '
                                   'layer when two_product_weights is true), three for attention and for everything when it is false.  '
                                   'The split-precision kernels (3x3 convolutions, GEMMs, '
                                   'attention) carry every contraction of the step.  Under dense fp16 MFMA load the shader clock of '
                                   'this part settles at ~1.67 GHz (tools/micro/mfma_clock.hip, profiles/r2_micro_mfma_clock_and_lds.json), '
                                   'i.e. a sustained peak of ~580 TFLOP/s fp32-equivalent; with operands that change from one MFMA '
                                   'to the next, as in a real contraction, the power limit holds the part at ~23 ns per 32x32x16 MFMA '
                                   'per SIMD (tools/micro/conv_loop.hip: ~1.44 GHz at 33 cycles, sustained), i.e. ~490 TFLOP/s fp32-equivalent',
                        'frac_of_sustained_f16x3_peak': round(ach / (PEAK_F16_TFLOPS / 3 * 1.67 / 2.4), 4),
                        'frac': round(ach / peak_issued, 4), 'frac_of_exact_fp32_mfma_peak': round(ach / PEAK_F32_TFLOPS, 4),
                        'peak_three_product': round(PEAK_F16_TFLOPS / 3, 1),
                        'frac_of_three_product_peak': round(ach / (PEAK_F16_TFLOPS / 3), 4),
                        # with fp16-exact weights the weight contractions (all but attention's ~3 % of the FLOPs) need TWO fp16
                        # products for the same bits, so the stricter denominator for this step is 2500 / 2
                        'peak_two_product': round(PEAK_F16_TFLOPS / 2, 1),
                        'frac_of_two_product_peak': round(ach / (PEAK_F16_TFLOPS / 2), 4)}
            result['sds'] = {'steps_per_sec': args.sds_steps * world / dt_sds, 'ms_per_step': sds_ms, 'roofline': sds_roof,
                             'dtype': 'f32 tensors; every convolution, linear layer and attention product on fp16 MFMA in split precision (f16x3, ~1e-6 '
                                      'relative).  The random UNet / VAE weights are fp16-REPRESENTABLE values in fp32 containers, like the '
                                      'reference\'s revision="fp16" checkpoint cast up (DS_NeRF/guidance/sd_utils.py:69-74): contractions with a '
                                      'weight operand run TWO products (W_hi x_hi + W_hi x_lo; bit-identical to three, the lo half of such a '
                                      'weight is zero), activation x activation products (attention) three',
                             'two_product_weights': bool(getattr(sd.networks, 'fp16_weights', False)),
                             'ms_per_step_all': [round(t * 1e3, 2) for t in sds_times],
                             'mode': ('one captured hipGraph per (shape, prompt): the default of StableDiffusion for the built-in networks'
                                      + ('' if world == 1 else ' (MVIP_GRAPHS_WITH_DIST=1: replay beside a live multi-rank process group)')
                                      if graphs_ok else 'eager (the default next to a live multi-rank process group)'),
                             'ms_per_step_eager': ms_eager32,
                             'ms_per_step_fp16_hipgraph': ms_graph16, 'ms_per_step_fp16_eager': ms_eager16,
                             'fp16_what': "the reference's --fp16 mode on the same hand-written kernels, single fp16 product per "
                                          'step, fp32 accumulate (csrc/conv3x3.hip, attention.hip: <..., true> instantiations)',
                             'what': 'median step; train_step_sd at 504x378 -> 512^2, SD-1.5-inpaint-shaped UNet (B=2, '
                             '9ch, 64x64) + VAE encoder x2 fwd / x1 bwd, random weights; one independent step per rank'}
            mg['train_with_sds_ms'] = dt_full / args.sds_steps * 1e3
            mg['what'] = ('STRONG-scaling legs (fixed work, rays and SDS terms sharded over the ranks): train_ms = second-stage '
                          'iteration without the prior, train_with_sds_ms = configs[1] iteration, config2_ms / config3_ms = '
                          'configs[2] / configs[3] iterations at the metric resolution; allreduce_ms = the 4.77 MB gradient '
                          'bucket alone.  The headline `value` at N > 1 is strong scaling too (one frame over all ranks)')
            result['train_with_sds'] = {'ms_per_step': dt_full / args.sds_steps * 1e3, 'dtype': 'f32 NeRF kernels; SDS as in sds.dtype',
                                        'iterations_per_sec': args.sds_steps / dt_full,
                                        'what': 'full BASELINE configs[1] second-stage iteration: masked render + RGB SDS '
                                                '+ colour/depth batches, backward, all-reduce, Adam (rays sharded over ranks)'}

    def legs_and_prediction():
        extra_legs()
        add_scaling_prediction(result, world)

    rep.guarded(legs_and_prediction)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
