"""Second-stage training iterations of the hash-grid model (the bench's `hashgrid_model` training leg) on their own,
for `rocprofv3 --kernel-trace --stats`:  rocprofv3 ... -- python3 tools/hashgrid_train_profile.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene

dev = torch.device('cuda', 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
a = bench.make_args()
a.no_tcnn, a.netchunk, a.lrate = False, 1 << 20, 1e-2
scene = SyntheticScene(bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR, device=dev)
tr = SecondStageTrainer(a, scene, dev, guidance=None)
if os.environ.get('MVIP_HG_HALF2') == '1':
    for net in (tr.kw_train['network_fn'], tr.kw_train['network_fine']):
        net.table_grad_atomics = 'half2'
for k in range(3):
    tr.step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    tr.step(3 + k)
torch.cuda.synchronize()
print('ms_per_step', (time.perf_counter() - t0) / steps * 1e3)
