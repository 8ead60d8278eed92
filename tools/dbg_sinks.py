import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvip_nerf_amd import ops
sys.path.insert(0, 'tests')
from test_transformer import decode_planes, decode_vfrag
cuda = torch.device('cuda:0')
Nb, K, P, rq, rv, v_dt = 2, 320, 256, 384, 512, 2
gen = torch.Generator().manual_seed(K + P)
M = 2 * rq + rv
x = torch.randn(Nb, K, P, generator=gen) * 1.7
W = torch.randn(M, K, generator=gen) / K ** 0.5
b = torch.randn(M, generator=gen) * 0.3
ref = torch.einsum('mk,nkp->nmp', W.double(), x.double()) + b.double()[None, :, None]
s2 = ops.absmax_scale(x.to(cuda))
xs = ops.split_planes_strided(x.to(cuda), Nb, K, P, K * P, P, 1, s2)
secs = [(rq, 'planes', 0.5), (rq, 'planes', 64.0), (rv, 'vfrag', 4.0)]
bufs = ops.gemm_f16x3_sinks(xs, ops.gemm_pack_a(W.to(cuda), M, K, K, 1), Nb, K, P, secs, bias=b.to(cuda), x_scale2=s2, v_dt=v_dt)
torch.cuda.synchronize()
row = 0
for (rows, kind, sc), buf in zip(secs, bufs):
    want = ref[:, row:row + rows] * sc
    got = decode_planes(buf, Nb, rows, P) if kind == 'planes' else decode_vfrag(buf, Nb, rows // 32, P)
    bad = (got - want).abs() > 1e-4 * float(want.abs().max())
    print(kind, row, 'bad frac', float(bad.double().mean()))
    if bad.any():
        n, r, p = np.nonzero(bad.numpy())
        print('  bad n', np.unique(n), 'rows%32', np.unique(r % 32), 'rows//32', np.unique(r // 32)[:12], 'cols%64', np.unique(p % 64)[:70])
        print('  example', n[0], r[0], p[0], float(got[n[0], r[0], p[0]]), float(want[n[0], r[0], p[0]]))
        # is the bad value some other element of want?
        g = float(got[n[0], r[0], p[0]])
        w = want[n[0]].numpy()
        loc = np.argwhere(np.abs(w - g) < 1e-4 * np.abs(w).max())
        print('  value found at (row, col):', loc[:5])
    row += rows
