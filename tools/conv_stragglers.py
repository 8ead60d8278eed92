import json, os, sys, torch
sys.path.insert(0, '/root/repo')
from mvip_nerf_amd import ops
from mvip_nerf_amd._lib import ptr, stream, call
dev = torch.device('cuda', 0)
N, cin, cout, H, W = 1, 512, 512, 128, 128
conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
x = torch.randn(N, cin, H, W, device=dev); rs = torch.randn(N, cout, H, W, device=dev)
s2 = ops.absmax_scale(x); xs = ops._split_buffer(N, cin, H * W, dev)
call('mvip_split_planes', ptr(x), N, cin, H * W, ptr(s2), ptr(xs, torch.float16), 0, stream())
y = torch.empty(N, cout, H, W, device=dev); pk = ops._conv_packed(conv, False); bias = conv.bias.detach()
for _ in range(3): ops._conv3x3_launch(xs, pk, bias, None, rs, s2, N, cin, cout, H, W, y)
torch.cuda.synchronize()
for rep in range(2):
    buf = torch.zeros(8 * 65536, device=dev, dtype=torch.int64)
    os.environ['MVIP_CONV_PROBE'] = str(buf.data_ptr())
    ops._conv3x3_launch(xs, pk, bias, None, rs, s2, N, cin, cout, H, W, y)
    torch.cuda.synchronize()
    b = buf.view(-1, 8).cpu()
    b = b[b[:, 0] > 0]
    dur = b[:, 1].double() / 100.0
    hw = b[:, 5] & 0xffffffff; xcc = (b[:, 5] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3; wave = hw & 0xf
    ids = torch.arange(b.shape[0])
    total = b.shape[0]
    sw = (ids & 7) * (total >> 3) + (ids >> 3)
    mb = sw % 8; tile = sw // 8
    print('rep', rep, 'dur min/med/max', float(dur.min()), float(dur.median()), float(dur.max()))
    for name, key in (('xcc', xcc), ('se', se), ('cu', cu), ('mb', mb), ('tile_row', tile // 4), ('wave_slot', wave)):
        vals = sorted(set(key.tolist()))
        print(' ', name, {int(v): round(float(dur[key == v].mean()), 1) for v in vals})
os.environ.pop('MVIP_CONV_PROBE')
