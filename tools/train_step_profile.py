"""Per-kernel device time of one steady second-stage training iteration without the prior (the bench's `train` leg)."""
import sys, os, json, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene
dev = torch.device('cuda', 0)
scene = SyntheticScene(bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR, device=dev)
if '--after-hashgrid' in sys.argv:          # what bench.py runs in front of its `train` leg
    from mvip_nerf_amd import run
    a_h = bench.make_args()
    a_h.no_tcnn, a_h.netchunk, a_h.lrate = False, 1 << 20, 1e-2
    tr_h = SecondStageTrainer(a_h, scene, dev)
    with torch.no_grad():
        run.render(bench.H, bench.W, bench.FOCAL, chunk=1 << 15, c2w=bench.orbit_pose(3, dev), near=bench.NEAR, far=bench.FAR, **tr_h.kw_test)
    for k in range(4):
        tr_h.step(k)
    torch.cuda.synchronize()
    del tr_h
torch.manual_seed(1)
tr = SecondStageTrainer(bench.make_args(), scene, dev)
for k in range(3):
    tr.step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(5):
    tr.step(10 + k)
torch.cuda.synchronize()
print('ms per iteration', (time.perf_counter() - t0) / 5 * 1e3)
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    tr.step(100)
    torch.cuda.synchronize()
ev = sorted([e for e in prof.key_averages() if e.device_time_total > 0], key=lambda e: -e.device_time_total)
print('device ms', sum(e.device_time_total for e in ev) / 1e3, 'launches', sum(e.count for e in ev))
for e in ev[:14]:
    print(f'  {e.device_time_total / 1e3:8.3f} ms x{e.count:4d}  {e.key[:110]}')
