"""Per-kernel device time of ONE steady-state `train_step_sd` (forward + backward to the image) at SD-1.5-inpaint
shapes, from the torch profiler: top kernels with call counts, device-busy total and the wall time of the step.
  --fp16          the reference's --fp16 mode
  --graphs        also time the hipGraph replay of the same step
  --sequence=F    also write the step's launches IN ORDER to gpurun_out/F: [start_us, dur_us, name] per device activity
  --three         MVIP_TWO_PRODUCT off (the three-product kernels on the same fp16-exact weights: A/B of round 4's task 1a)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd.guidance.sd_utils import StableDiffusion          # noqa: E402


def main():
    if os.environ.get('MVIP_ROW_MOMENTS') == '0':          # A/B: every GroupNorm computes its moments from its input
        from mvip_nerf_amd import ops
        ops.ROW_MOMENTS = False
    if '--three' in sys.argv:
        from mvip_nerf_amd import ops
        ops.TWO_PRODUCT = False
    dev = torch.device('cuda', 0)
    fp16 = '--fp16' in sys.argv            # the reference's --fp16 mode on the single-product kernels
    graphs = '--graphs' in sys.argv        # also time the captured-hipGraph replay of the same step
    out_name = next((a.split('=', 1)[1] for a in sys.argv if a.startswith('--out=')), 'sds_step_profile.json')
    sd = StableDiffusion(dev, fp16, False, use_graphs=False)    # eager for the per-kernel table; graphs timed below
    g = torch.Generator(device=dev).manual_seed(2)
    H, W = 378, 504
    pred = torch.rand(1, 3, H, W, device=dev, generator=g).requires_grad_(True)
    mask = torch.zeros(1, 1, H, W, device=dev)
    mask[:, :, 137:241, 196:307] = 1

    def step(i):
        pred.grad = None
        (1e-4 * sd.train_step_sd(i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
    for k in range(3):
        step(1000 + k)
    torch.cuda.synchronize()
    ts = []
    for k in range(7):
        t0 = time.perf_counter()
        step(1010 + k)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        step(1100)
        torch.cuda.synchronize()
    seq_name = next((a.split('=', 1)[1] for a in sys.argv if a.startswith('--sequence=')), None)
    if seq_name:
        evs = [e for e in prof.events() if getattr(e, 'device_type', None) is not None and str(e.device_type).endswith('CUDA')]
        evs.sort(key=lambda e: e.time_range.start)
        t0 = evs[0].time_range.start if evs else 0
        os.makedirs('gpurun_out', exist_ok=True)
        json.dump([[round(e.time_range.start - t0, 2), round(e.time_range.end - e.time_range.start, 2), e.name[:140]] for e in evs],
                  open(os.path.join('gpurun_out', seq_name), 'w'))
    ev = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
    total = sum(e.device_time_total for e in ev) / 1e3
    n = sum(e.count for e in ev if e.device_time_total > 0)        # device activities only (the averages also list the launch calls)
    n_all = sum(e.count for e in ev)
    print(f'== train_step_sd: median wall {sorted(ts)[len(ts) // 2]:.2f} ms {[round(t, 1) for t in ts]}, device-busy {total:.2f} ms, {n} kernels / copies ({n_all} profiler events)')
    rows = []
    for e in ev[:70]:
        print(f'  {e.device_time_total / 1e3:8.3f} ms  x{e.count:4d}  {e.key[:120]}')
        rows.append([round(e.device_time_total / 1e3, 3), e.count, e.key[:120]])
    own = sum(e.device_time_total for e in ev if 'mvip::' in e.key) / 1e3
    print(f'   hand-written (mvip::) kernels: {own:.2f} ms of {total:.2f} ms device-busy')
    graph_ms = None
    if graphs:
        sd.use_graphs = True
        for k in range(3):
            step(2000 + k)
        torch.cuda.synchronize()
        tg = []
        for k in range(9):
            t0 = time.perf_counter()
            step(2010 + k)
            torch.cuda.synchronize()
            tg.append((time.perf_counter() - t0) * 1e3)
        graph_ms = sorted(tg)[len(tg) // 2]
        print(f'   hipGraph replay of the same step: median {graph_ms:.2f} ms {[round(t, 1) for t in tg]}')
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump({'fp16_mode': fp16, 'median_wall_ms': sorted(ts)[len(ts) // 2], 'wall_ms': ts, 'hipgraph_replay_ms': graph_ms,
               'device_busy_ms': total, 'kernels': n, 'mvip_kernels_ms': own, 'top': rows},
              open(os.path.join('gpurun_out', out_name), 'w'), indent=1)


if __name__ == '__main__':
    main()
