"""A/B of the split-precision no-grad forward: the 32-points-per-wave kernel (csrc/mlp_fwd_f16x3.hip, one wave per SIMD) against
the 16-points-per-wave kernel (csrc/mlp_fwd16_f16x3.hip, two waves per SIMD, round 5), on the bench's fine-pass launch
(190,512 rays x 128 samples) and on whole frames; interleaved rounds on one box.  MVIP_F16W16_RING selects the new kernel's ring
(0: 4 x 16 KB, 1: 3 x 32 KB) -- read once per process, so each geometry runs in its own child process.
    python tools/f16x3_w16_ab.py            -> gpurun_out/r5_f16x3_w16_ab.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    import bench
    from mvip_nerf_amd import ops, run
    dev = torch.device('cuda', 0)
    tr, te, *_ = run.create_nerf(bench.make_args(), device=dev)
    net = te['network_fine']
    rows = ops.ray_rows_from_pose(bench.orbit_pose(0, dev), bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR)
    z = ops.stratified_z(rows, 128, True)
    pts = rows.shape[0] * 128
    out = {}
    with torch.no_grad():
        net.inference_precision = 0
        ref = net.query_rays(rows, z)
        net.inference_precision = 1
        res = {}
        for name, two in (('w32_one_wave', False), ('w16_two_waves', True), ('w32_one_wave_again', False), ('w16_two_waves_again', True)):
            net.two_wave_f16x3 = two
            r = net.query_rays(rows, z)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                r = net.query_rays(rows, z)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            d = (r - ref).abs()
            res[name] = {'launch_ms': round(ms, 3), 'fp32_equivalent_TFLOPs': round(pts * bench.FLOP_PER_POINT / ms / 1e9, 1),
                         'fp16_product_TFLOPs': round(3 * pts * bench.FLOP_PER_POINT / ms / 1e9, 1),
                         'max_abs_diff_vs_fp32': float(d.max()), 'rel_rms_vs_fp32': float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())}
        out['fine_pass_launch'] = res
        frames = {}
        img0 = None
        for name, prec, two in (('fp32', 0, True), ('f16x3_w32', 1, False), ('f16x3_w16', 1, True)):
            for n in (te['network_fn'], te['network_fine']):
                n.inference_precision, n.two_wave_f16x3 = prec, two
            run.render(bench.H, bench.W, bench.FOCAL, chunk=1 << 15, c2w=bench.orbit_pose(1, dev), near=bench.NEAR, far=bench.FAR, **te)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(3):
                img = run.render(bench.H, bench.W, bench.FOCAL, chunk=1 << 15, c2w=bench.orbit_pose(2, dev), near=bench.NEAR, far=bench.FAR, **te)[0]
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            frames[name] = {'ms_per_frame': round(ms, 2), 'rays_per_sec': round(bench.H * bench.W / ms * 1e3)}
            if img0 is None:
                img0 = img
            else:
                frames[name]['psnr_vs_fp32_dB'] = float(-10 * torch.log10(((img - img0) ** 2).mean().clamp_min(1e-30)))
        out['frame'] = frames
    print('RESULT ' + json.dumps(out))


def main():
    res = {'what': __doc__.split('\n')[0], 'head': subprocess.run(['git', 'rev-parse', 'HEAD'], capture_output=True, text=True, cwd=ROOT).stdout.strip()}
    for ring in ('0', '1'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], capture_output=True, text=True,
                           env=dict(os.environ, MVIP_F16W16_RING=ring), cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
        res[f'ring{ring}_' + ('4x16KB' if ring == '0' else '3x32KB')] = json.loads(line[-1][7:]) if line else {'error': r.stderr[-1500:]}
        print(ring, res[f'ring{ring}_' + ('4x16KB' if ring == '0' else '3x32KB')], flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'r5_f16x3_w16_ab.json'), 'w'), indent=1)


if __name__ == '__main__':
    child() if '--child' in sys.argv else main()
