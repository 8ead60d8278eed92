"""Split-precision MFMA 3x3 convolution vs the library convolution at the VAE-encoder / UNet shapes."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops
from mvip_nerf_amd.guidance.sd_nets import GroupNorm, norm_act_conv

dev = torch.device('cuda', 0)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for (N, cin, cout, H, W) in [(1, 128, 128, 512, 512), (1, 256, 256, 256, 256), (1, 512, 512, 128, 128),
                             (1, 512, 512, 64, 64), (2, 640, 640, 32, 32), (2, 320, 320, 64, 64)]:
    torch.manual_seed(0)
    norm = GroupNorm(32, cin).to(dev)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    for p in list(norm.parameters()) + list(conv.parameters()):
        p.requires_grad_(False)
    x = torch.randn(N, cin, H, W, device=dev)
    with torch.no_grad():
        fused = norm_act_conv(norm, conv, x)
        lib = conv(norm(x, silu=True))
        err = float((fused - lib).abs().max() / lib.abs().max())
        t_f = timeit(lambda: norm_act_conv(norm, conv, x))
        t_l = timeit(lambda: conv(norm(x, silu=True)))
        act = norm(x, silu=True)
        t_conv_lib = timeit(lambda: conv(act))
    flop = 2.0 * N * H * W * cin * cout * 9
    print(json.dumps({'shape': [N, cin, cout, H, W], 'fused_ms': round(t_f, 4), 'gn_kernels+lib_conv_ms': round(t_l, 4),
                      'lib_conv_only_ms': round(t_conv_lib, 4), 'rel_diff_vs_lib': err,
                      'fused_TFLOPs_equiv': round(flop / t_f / 1e9, 1)}), flush=True)
