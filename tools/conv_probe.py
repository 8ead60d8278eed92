"""In-kernel clock probe of the 3x3 convolution (library built with -DMVIP_EXPERIMENT_CONV): per workgroup, shader
cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of the whole kernel, the prologue, the per-stage barrier wait, the per-stage DMA-issue
section, the per-stage MFMA / fragment-read section and the epilogue -> sustained clock and cycles per stage."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops
from mvip_nerf_amd._lib import ptr, stream, call
dev = torch.device('cuda', 0)
for (N, cin, cout, H, W) in [(1, 128, 128, 512, 512), (1, 512, 512, 128, 128), (2, 640, 640, 32, 32)]:
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    x = torch.randn(N, cin, H, W, device=dev)
    rs = torch.randn(N, cout, H, W, device=dev)
    s2 = ops.absmax_scale(x)
    xs = ops._split_buffer(N, cin, H * W, dev)
    call('mvip_split_planes', ptr(x), N, cin, H * W, ptr(s2), ptr(xs, torch.float16), 0, stream())
    y = torch.empty(N, cout, H, W, device=dev)
    pk = ops._conv_packed(conv, False)
    bias = conv.bias.detach()
    for _ in range(3):
        ops._conv3x3_launch(xs, pk, bias, None, rs, s2, N, cin, cout, H, W, y)
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 65536, device=dev, dtype=torch.int64)
    os.environ['MVIP_CONV_PROBE'] = str(buf.data_ptr())
    ops._conv3x3_launch(xs, pk, bias, None, rs, s2, N, cin, cout, H, W, y)
    torch.cuda.synchronize()
    os.environ.pop('MVIP_CONV_PROBE')
    b = buf.view(-1, 8).cpu().double()
    b = b[b[:, 0] > 0]
    med = b.median(0).values
    start, dur = b[:, 7], b[:, 1]
    span_us = float((start + dur).max() - start.min()) / 100.0
    late_us = float(start.max() - start.min()) / 100.0
    nstage = (cin // 16) * 3
    print(json.dumps({'shape': f'{N}x{cin}->{cout}@{H}x{W}', 'workgroups': int(b.shape[0]), 'stages_unsplit': nstage,
                      'cycles_total': med[0].item(), 'MHz': round(med[0].item() / (med[1].item() / 100.0), 0),
                      'prologue': med[2].item(), 'barrier_wait': med[6].item(), 'dma_issue': med[3].item(), 'compute': med[4].item(),
                      'epilogue': med[5].item(), 'wg_us_min_med_max': [round(float(dur.min()) / 100, 1), round(float(dur.median()) / 100, 1), round(float(dur.max()) / 100, 1)],
                      'first_start_to_last_end_us': round(span_us, 1), 'last_start_after_first_us': round(late_us, 1),
                      'mfma_cycles_ideal': 'stages_of_this_workgroup x 36 x 32 (MT = 2)'}))
