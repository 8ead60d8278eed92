"""Isolated timing of the HBM-bound kernels at bench scale: algorithmic bytes / time vs the 8 TB/s
HBM3E peak (6.3 TB/s achievable per MI355X_MICROARCH.md).  One JSON object per kernel."""
import json, sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops

PEAK = 8000.0  # GB/s
dev = torch.device('cuda', 0)
B = 190512
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def report(name, bytes_, t):
    gbs = bytes_ / t / 1e9
    print(json.dumps({'kernel': name, 'ms': round(t * 1e3, 4), 'algorithmic_MB': round(bytes_ / 1e6, 2),
                      'GBps': round(gbs, 1), 'frac_of_8TBps': round(gbs / PEAK, 3)}), flush=True)


c2w = torch.tensor([[1., 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0.3]], device=dev)
rows = ops.ray_rows_from_pose(c2w, 378, 504, 383.65, 1.2, 7.74)
report('ray_rows_from_pose (44 B/ray out)', B * 44, timeit(lambda: ops.ray_rows_from_pose(c2w, 378, 504, 383.65, 1.2, 7.74)))
report('get_rays (24 B/ray out)', B * 24, timeit(lambda: ops.get_rays(378, 504, 383.65, c2w)))
for S in (64, 128):
    t_rand = torch.rand(B, S, device=dev, generator=g)
    report(f'stratified_z S={S} (4 in + 4 out B/sample)', B * S * 8, timeit(lambda: ops.stratified_z(rows, S, True, t_rand)))
    z = ops.stratified_z(rows, S, True, t_rand)
    raw = torch.randn(B, S, 4, device=dev, generator=g)
    noise = torch.randn(B, S, device=dev, generator=g)
    fwd_bytes = B * S * (16 + 4 + 4 + 4) + B * (44 + 24)
    report(f'composite_fwd S={S} (28 B/sample + 68 B/ray)', fwd_bytes,
           timeit(lambda: ops.composite(raw, z, rows, noise, True)))
    raw_g = raw.clone().requires_grad_(True)
    out = ops.composite(raw_g, z, rows, noise, True)
    g_rgb = torch.randn(B, 3, device=dev, generator=g)
    def bwd():
        raw_g.grad = None
        out[0].backward(g_rgb, retain_graph=True)
    report(f'composite_bwd S={S} (24 in + 16 out B/sample)', B * S * 40 + B * 56, timeit(bwd))
zc = ops.stratified_z(rows, 64, True)
w = torch.rand(B, 64, device=dev, generator=g)
u = torch.rand(B, 64, device=dev, generator=g)
report('sample_pdf_merge Nc=Nf=64 (768 in + 772 out B/ray)', B * (768 + 772), timeit(lambda: ops.sample_pdf_merge(zc, w, u)))
u_det = torch.linspace(0., 1., 64, device=dev)          # test-mode renders (perturb = 0): one sorted row shared by all rays, no sort of the new samples
report('sample_pdf_merge Nc=Nf=64, deterministic u row (512 in + 772 out B/ray)', B * (512 + 772), timeit(lambda: ops.sample_pdf_merge(zc, w, u_det)))
# a PEAKED pdf (a surface: every new sample lands in two or three depth intervals) -- what trained scenes look like
kpk = torch.randint(2, 62, (B, 1), device=dev, generator=g).float()
w_pk = torch.exp(-0.5 * ((torch.arange(64, device=dev)[None].float() - kpk) / 0.7) ** 2) + 1e-7
report('sample_pdf_merge Nc=Nf=64, peaked pdf, random u (768 in + 772 out B/ray)', B * (768 + 772), timeit(lambda: ops.sample_pdf_merge(zc, w_pk, u)))
# the same kernels on working sets PAST the 256 MB Infinity Cache (8 frames' rays: 2.3 GB for the merge): the 190,512-ray sets above
# (97-293 MB) fit it, so their fractions are flattered
BL = 8 * B
rows_l = rows.repeat(8, 1)
t_l = torch.rand(BL, 64, device=dev, generator=g)
report(f'LARGE stratified_z S=64, {BL} rays (4 in + 4 out B/sample)', BL * 64 * 8, timeit(lambda: ops.stratified_z(rows_l, 64, True, t_l), reps=5))
z_l = ops.stratified_z(rows_l, 64, True, t_l)
raw_l = torch.randn(BL, 64, 4, device=dev, generator=g)
noise_l = torch.randn(BL, 64, device=dev, generator=g)
report(f'LARGE composite_fwd S=64, {BL} rays (28 B/sample + 68 B/ray)', BL * 64 * 28 + BL * 68,
       timeit(lambda: ops.composite(raw_l, z_l, rows_l, noise_l, True), reps=5))
del raw_l, noise_l
w_l = torch.rand(BL, 64, device=dev, generator=g)
u_l = torch.rand(BL, 64, device=dev, generator=g)
zs_l = torch.sort(z_l, -1)[0]
report(f'LARGE sample_pdf_merge Nc=Nf=64, {BL} rays (768 in + 772 out B/ray)', BL * (768 + 772), timeit(lambda: ops.sample_pdf_merge(zs_l, w_l, u_l), reps=5))
report(f'LARGE sample_pdf_merge Nc=Nf=64, {BL} rays, deterministic u row (512 in + 772 out B/ray)', BL * (512 + 772),
       timeit(lambda: ops.sample_pdf_merge(zs_l, w_l, u_det), reps=5))
del w_l, u_l, zs_l, z_l, t_l
x = torch.randn(B * 16, 3, device=dev, generator=g)
report('posenc L=10 (12 in + 252 out B/point)', x.shape[0] * 264, timeit(lambda: ops.posenc(x, 10)))
x_l = torch.randn(B * 64, 3, device=dev, generator=g)
report(f'LARGE posenc L=10, {x_l.shape[0]} points (12 in + 252 out B/point)', x_l.shape[0] * 264, timeit(lambda: ops.posenc(x_l, 10), reps=5))
del x_l
# reference point: a plain device copy of 1 GB
big = torch.empty(256 << 20, device=dev); dst = torch.empty_like(big)
report('torch copy 1 GiB (read+write)', 2 * big.numel() * 4, timeit(lambda: dst.copy_(big), reps=5))
