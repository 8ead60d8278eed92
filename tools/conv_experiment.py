"""Timing-only experiment on the 3x3 convolution kernel: which part of a launch is epilogue traffic, matrix work, operand
staging?  Needs a library built with -DMVIP_EXPERIMENT_CONV (results are WRONG in every mode but 0):
  MVIP_EXTRA_FLAGS=-DMVIP_EXPERIMENT_CONV python -m mvip_nerf_amd.csrc.build -f
MVIP_CONV_DBG bits: 1 no epilogue, 2 no MFMAs, 4 no input DMA, 8 no weight DMA, 16 no barrier, 32 wave-linear B fragment reads.  One process per mode (the switch is read
once)."""
import json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    sys.path.insert(0, os.path.dirname(here))
    from mvip_nerf_amd import ops
    from mvip_nerf_amd._lib import ptr, stream, call
    dev = torch.device('cuda', 0)
    out = {}
    for (N, cin, cout, H, W) in [(1, 128, 128, 512, 512), (1, 256, 256, 256, 256), (1, 512, 512, 128, 128), (2, 640, 640, 32, 32),
                                 (2, 320, 320, 64, 64)]:
        conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
        if os.environ.get('MVIP_CONV_EXACT_FP16_WEIGHTS', '1') == '1':     # as the reference's revision="fp16" weights: TWO products (NP = 2)
            with torch.no_grad():
                conv.weight.copy_(conv.weight.half().float())
        x = torch.randn(N, cin, H, W, device=dev)
        rs = torch.randn(N, cout, H, W, device=dev)
        s2 = ops.absmax_scale(x)
        xs = ops._split_buffer(N, cin, H * W, dev)
        call('mvip_split_planes', ptr(x), N, cin, H * W, ptr(s2), ptr(xs, torch.float16), 0, stream())
        y = torch.empty(N, cout, H, W, device=dev)
        pk = ops._conv_packed(conv, False)
        bias = conv.bias.detach()
        def run():
            ops._conv3x3_launch(xs, pk, bias, None, rs, s2, N, cin, cout, H, W, y)
        # MVIP_CONV_SUSTAIN=n: n untimed launches first (~100 ms: the clock governor needs tens of milliseconds to settle
        # under dense fp16 MFMA load, and a 5 ms measurement catches it in transit), then the 20 timed ones
        for _ in range(max(3, int(os.environ.get('MVIP_CONV_SUSTAIN', '3')))): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        out[f'{N}x{cin}->{cout}@{H}x{W}'] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print(json.dumps(out))
else:
    for mode, tag in ((0, 'full'), (1, 'no_epilogue'), (2, 'no_mfma'), (3, 'no_mfma_no_epilogue'), (12, 'no_dma'),
                      (13, 'no_dma_no_epilogue'), (14, 'no_dma_no_mfma'), (15, 'loop_skeleton_only'),
                      (31, 'skeleton_no_barrier'), (29, 'mfma_lds_only_no_barrier'), (16, 'full_no_barrier'), (17, 'no_barrier_no_epilogue'),
                      (29 + 32, 'mfma_lds_only_linear_b_reads'), (32, 'full_linear_b_reads')):
        r = subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, MVIP_CONV_DBG=str(mode)),
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        print(tag.ljust(22), line[-1] if line else r.stderr[-300:], flush=True)
