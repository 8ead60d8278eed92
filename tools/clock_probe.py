"""Sustained shader clock and cycles per workgroup of the fused MLP forward (diagnostic entry point)."""
import ctypes, sys, json
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops, _lib
import bench
dev = torch.device('cuda', 0)
tr, te, *_ = __import__('mvip_nerf_amd.run', fromlist=['x']).create_nerf(bench.make_args(), device=dev)
net = te['network_fine']
rows = ops.ray_rows_from_pose(bench.orbit_pose(0, dev), bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR)
z = ops.stratified_z(rows, 128, True)
B, S = z.shape
raw = torch.empty(B, S, 4, device=dev)
nwg = (B * S + 127) // 128
clk = torch.zeros(nwg * 10, device=dev, dtype=torch.int64)
lib = _lib.load()
lib.mvip_debug_forward_clock.restype = ctypes.c_int
lib.mvip_debug_forward_clock.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 3
for _ in range(6):   # the probe launches run back to back for > 1 s so DVFS has settled
    rc = lib.mvip_debug_forward_clock(_lib.ptr(net.packed()), _lib.ptr(rows), _lib.ptr(z), B, S, _lib.ptr(raw),
                                      ctypes.c_void_p(clk.data_ptr()), _lib.stream())
    assert rc == 0
torch.cuda.synchronize()
c = clk.cpu().numpy().reshape(-1, 10).astype(np.float64)
cyc, ticks = c[:, 0], c[:, 1]
ghz = cyc / ticks * 0.1
print(json.dumps({'workgroups': int(nwg), 'median_cycles_per_wg': float(np.median(cyc)), 'median_us_per_wg': float(np.median(ticks) / 100),
                  'median_clock_GHz': float(np.median(ghz)), 'p10_clock': float(np.percentile(ghz, 10)), 'p90_clock': float(np.percentile(ghz, 90)),
                  'mfma_cycles_per_wg': 9280 * 64, 'mfma_share_of_cycles': 9280 * 64 / float(np.median(cyc))}))
names = ['inputs+encode', 'ring primed', 'layer0 (256 mfma)', 'layers1-4 (4096)', 'layer5 (1280)', 'layers6-7 (2048)',
         'sigma+feature (1024)', 'views (576)', 'rgb+store']
st = np.median(np.concatenate([c[:, 2:10], c[:, 0:1]], 1), 0)
prev = 0
mf = [0, 0, 256, 4096, 1280, 2048, 1024, 576, 0]
for n, t, m in zip(names, st, mf):
    d = t - prev
    print(f'{n:24s} {d:10.0f} cycles   mfma {m * 64:8d}   overhead {d - m * 64:8.0f}')
    prev = t
