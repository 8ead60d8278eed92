import sys, json, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene
dev = torch.device('cuda', 0)
scene = SyntheticScene(bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR, device=dev)
out = {}
for prec in (0, 1):
    torch.manual_seed(1)
    tr = SecondStageTrainer(bench.make_args(), scene, dev)
    for n in (tr.kw_train['network_fn'], tr.kw_train['network_fine']):
        n.train_precision = prec
    tr.step(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(5): loss, _ = tr.step(1 + k)
    torch.cuda.synchronize()
    out[f'train_precision{prec}_ms'] = (time.perf_counter() - t0) / 5 * 1e3
    out[f'loss{prec}'] = float(loss)
print(json.dumps(out))
