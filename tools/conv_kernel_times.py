"""Kernel-only time of the 3x3 convolution (csrc/conv3x3.hip) at the VAE-encoder / UNet shapes of the SDS step, split
precision (prec 0) and fp16 mode (prec 1): profiler device time of the conv kernel alone, algorithmic TFLOP/s and the share
of the fp16-MFMA peak (2500 / 3 resp. 2500 TFLOP/s)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops
from mvip_nerf_amd.guidance.sd_nets import GroupNorm, norm_act_conv

dev = torch.device('cuda', 0)
shapes = [(1, 128, 128, 512, 512), (1, 128, 256, 256, 256), (1, 256, 256, 256, 256), (1, 256, 512, 128, 128), (1, 512, 512, 128, 128),
          (1, 512, 512, 64, 64), (2, 320, 320, 64, 64), (2, 640, 640, 32, 32), (2, 1280, 1280, 16, 16), (2, 1280, 1280, 8, 8),
          (2, 960, 320, 64, 64), (2, 2560, 1280, 16, 16)]
out = []
for (N, cin, cout, H, W) in shapes:
    torch.manual_seed(0)
    norm = GroupNorm(32, cin).to(dev)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    for p in list(norm.parameters()) + list(conv.parameters()):
        p.requires_grad_(False)
    x = torch.randn(N, cin, H, W, device=dev)
    rec = {'shape': [N, cin, cout, H, W], 'GFLOP': round(2.0 * N * H * W * cin * cout * 9 / 1e9, 2)}
    for prec in (0, 1):
        with torch.no_grad(), ops.precision(prec):
            for _ in range(3):
                norm_act_conv(norm, conv, x)
            torch.cuda.synchronize()
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                for _ in range(5):
                    norm_act_conv(norm, conv, x)
                torch.cuda.synchronize()
        us = sum(e.device_time_total for e in prof.key_averages() if 'conv3x3_f16x3_kernel' in e.key) / 5
        red = sum(e.device_time_total for e in prof.key_averages() if 'split_reduce' in e.key) / 5
        tf = rec['GFLOP'] / us * 1e-3 * 1e3 / 1e3 * 1e3 if us else 0
        tf = rec['GFLOP'] / (us * 1e-6) / 1e3
        rec[f'prec{prec}'] = {'conv_us': round(us, 1), 'split_reduce_us': round(red, 1), 'TFLOPs': round(tf, 1),
                              'frac_of_peak': round(tf / (2500 / 3 if prec == 0 else 2500), 3)}
    out.append(rec)
    print(json.dumps(rec), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/r3_conv_kernel_times.json', 'w'), indent=1)
