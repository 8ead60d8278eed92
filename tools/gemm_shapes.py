"""Which GEMM shapes does one SDS step launch (through ops.gemm_f16x3), how often, and how long does each take alone?"""
import collections, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvip_nerf_amd import ops
from mvip_nerf_amd.guidance import sd_utils
dev = torch.device('cuda', 0)
calls = collections.Counter()
orig = ops.gemm_f16x3
def rec(xs, packed, N, K, M, P, **kw):
    calls[(N, K, M, P, kw.get('residual') is not None)] += 1
    return orig(xs, packed, N, K, M, P, **kw)
ops.gemm_f16x3 = rec
sd = sd_utils.StableDiffusion(dev, False, False)
pred = torch.rand(1, 3, bench.H, bench.W, device=dev, requires_grad=True)
mask = torch.zeros(1, 1, bench.H, bench.W, device=dev)
mask[:, :, 137:241, 196:307] = 1
def step(i):
    pred.grad = None
    (1e-4 * sd.train_step_sd(i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
step(1000)
calls.clear()
step(1001)
torch.cuda.synchronize()
ops.gemm_f16x3 = orig
rows = []
for (N, K, M, P, res), cnt in calls.items():
    x = torch.randn(N, K, P, device=dev); W = torch.randn(M, K, device=dev) / K ** 0.5
    xs, s2 = ops._scaled_planes(x, N, K, P, K * P, P, 1); pk = ops.gemm_pack_a(W, M, K, K, 1)
    r = torch.randn(N, M, P, device=dev) if res else None
    for _ in range(3): ops.gemm_f16x3(xs, pk, N, K, M, P, residual=r, x_scale2=s2)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): ops.gemm_f16x3(xs, pk, N, K, M, P, residual=r, x_scale2=s2)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    rows.append({'N': N, 'K': K, 'M': M, 'P': P, 'residual': res, 'calls': cnt, 'us_each': round(us, 1), 'us_total': round(us * cnt, 1),
                 'TFLOPs_equiv': round(2.0 * N * K * M * P / us / 1e6, 1)})
rows.sort(key=lambda r: -r['us_total'])
for r in rows: print(json.dumps(r))
print('total ms', round(sum(r['us_total'] for r in rows) / 1e3, 2))
