"""32-point (one wave per SIMD) vs 16-point (two waves per SIMD) exact-fp32 forward on the bench's fine pass."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvip_nerf_amd import ops, run
dev = torch.device('cuda', 0)
tr, te, *_ = run.create_nerf(bench.make_args(), device=dev)
net = te['network_fine']
rows = ops.ray_rows_from_pose(bench.orbit_pose(0, dev), bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR)
z = ops.stratified_z(rows, 128, True)
ps = net.param_list()
packed, p16 = net.packed(), net.packed_w16()


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    a = t(lambda: ops.mlp_rays(rows, z, packed, ps))
    b = t(lambda: ops.mlp_rays(rows, z, packed, ps, packed16=p16))
    a2 = t(lambda: ops.mlp_rays(rows, z, packed, ps))
    b2 = t(lambda: ops.mlp_rays(rows, z, packed, ps, packed16=p16))
flop = rows.shape[0] * 128 * 1186816
print(json.dumps({'fwd32_ms': [a, a2], 'fwd16_ms': [b, b2], 'fwd32_TFLOPs': flop / a2 / 1e9, 'fwd16_TFLOPs': flop / b2 / 1e9}))
