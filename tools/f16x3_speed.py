import sys, json, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvip_nerf_amd import ops, run
dev = torch.device('cuda', 0)
tr, te, *_ = run.create_nerf(bench.make_args(), device=dev)
net = te['network_fine']
rows = ops.ray_rows_from_pose(bench.orbit_pose(0, dev), bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR)
z = ops.stratified_z(rows, 128, True)
pts = rows.shape[0] * 128
out = {}
ref = None
for prec in (0, 1):
    net.inference_precision = prec
    with torch.no_grad():
        r = net.query_rays(rows, z); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): r = net.query_rays(rows, z)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    out[f'precision{prec}'] = {'ms': ms, 'algorithmic_TFLOPs': pts * bench.FLOP_PER_POINT / ms / 1e9}
    if prec == 0: ref = r
    else:
        d = (r - ref).abs()
        out['max_abs_diff_vs_fp32'] = float(d.max()); out['max_abs_raw'] = float(ref.abs().max())
        out['rel_rms'] = float((d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
# full frame render both ways
for prec in (0, 1):
    for n in (te['network_fn'], te['network_fine']): n.inference_precision = prec
    with torch.no_grad():
        run.render(bench.H, bench.W, bench.FOCAL, chunk=1 << 15, c2w=bench.orbit_pose(1, dev), near=bench.NEAR, far=bench.FAR, **te)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(3): img = run.render(bench.H, bench.W, bench.FOCAL, chunk=1 << 15, c2w=bench.orbit_pose(2 + k, dev), near=bench.NEAR, far=bench.FAR, **te)[0]
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    out[f'render_precision{prec}'] = {'ms_per_frame': ms, 'rays_per_sec': bench.H * bench.W / ms * 1e3}
    if prec == 0: img0 = img
    else: out['render_psnr_vs_fp32_dB'] = float(-10 * torch.log10(((img - img0) ** 2).mean().clamp_min(1e-30)))
print(json.dumps(out))
