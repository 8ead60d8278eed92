"""A/B of the second-stage iteration (no prior, bench shapes): coarse pass of the masked render with / without
autograd, stash budget queried from the device / fixed by MVIP_STASH_BUDGET_BYTES, plus the per-kernel device time of
one iteration (torch profiler)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                         # noqa: E402
from mvip_nerf_amd import run                                        # noqa: E402
from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene  # noqa: E402

dev = torch.device('cuda', 0)
scene = SyntheticScene(bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR, device=dev)
out = {}
orig = run.render_rays


def timed(tag, coarse_grad_forced=None, env=None):
    if env is None:
        os.environ.pop('MVIP_STASH_BUDGET_BYTES', None)
    else:
        os.environ['MVIP_STASH_BUDGET_BYTES'] = env
    if coarse_grad_forced is not None:
        run.render_rays = lambda *a, **k: orig(*a, **dict(k, coarse_grad=coarse_grad_forced))
    else:
        run.render_rays = orig
    torch.manual_seed(1)
    tr = SecondStageTrainer(bench.make_args(), scene, dev)
    tr.step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(5):
        tr.step(1 + k)
    torch.cuda.synchronize()
    out[tag] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
    return tr


timed('coarse_grad_default(False for masked)_budget_query')
timed('coarse_grad_forced_True_budget_query', True)
timed('coarse_grad_default_budget_env_96GiB', None, str(96 << 30))
tr = timed('coarse_grad_forced_True_budget_env_96GiB', True, str(96 << 30))
run.render_rays = orig
os.environ.pop('MVIP_STASH_BUDGET_BYTES', None)
tr = timed('again_default')
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    tr.step(50)
    torch.cuda.synchronize()
ev = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in ev if e.device_type == torch.autograd.DeviceType.CUDA) / 1e3
print(json.dumps(out, indent=1))
print(f'device-busy {tot:.2f} ms')
for e in ev[:16]:
    if e.device_time_total > 0:
        print(f'  {e.device_time_total / 1e3:8.3f} ms  x{e.count:4d}  {e.key[:100]}')
cpu = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
for e in cpu[:8]:
    print(f'  cpu {e.self_cpu_time_total / 1e3:8.3f} ms  x{e.count:4d}  {e.key[:80]}')
