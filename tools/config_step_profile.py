"""Per-kernel device time of ONE steady BASELINE configs[2] / configs[3] iteration as bench.py runs it (378x504, RGB + normal
[+ collaborative] SDS, NeRF kernels in split precision): python tools/config_step_profile.py [2|3] [--f32]"""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                      # noqa: E402
from mvip_nerf_amd.guidance.sd_utils import StableDiffusion                       # noqa: E402
from mvip_nerf_amd.nerf.utils import Pretrain_Model                               # noqa: E402
from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene              # noqa: E402

cfg = 3 if '3' in sys.argv[1:2] else 2
dev = torch.device('cuda', 0)
torch.manual_seed(0)
sd = StableDiffusion(dev, False, False)
opt = types.SimpleNamespace(is_rgb_guidance=True, is_colla_guidance=cfg == 3, is_normal_guidance=True, text='a stone bench in a park',
                            text_normal='a normal map of a stone bench in a park', rgb_guidance_scale=7.5, colla_guidance_scale=7.5,
                            normal_guidance_scale=1.5, normal_start=500, lambda_guidance=1)
a2 = bench.make_args()
a2.is_normal_guidance, a2.is_colla_guidance, a2.normalmap_render_factor = True, cfg == 3, 2
scene = SyntheticScene(bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR, device=dev)
tr = SecondStageTrainer(a2, scene, dev, guidance=Pretrain_Model(opt, dev, {'SD': sd}))
if '--f32' not in sys.argv:
    for n in (tr.kw_train['network_fn'], tr.kw_train['network_fine']):
        n.train_precision = n.inference_precision = 1
for k in range(3):
    tr.step(999 + k)
torch.cuda.synchronize()
ts = []
for k in range(3):
    t0 = time.perf_counter()
    tr.step(1010 + k)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    tr.step(1100)
    torch.cuda.synchronize()
ev = sorted([e for e in prof.key_averages() if e.device_time_total > 0], key=lambda e: -e.device_time_total)
busy = sum(e.device_time_total for e in ev) / 1e3
sds = sum(e.device_time_total for e in ev if any(t in e.key for t in ('conv3x3', 'gemm5', 'attn', 'cv_', 'gn_', 'tok::', 'gm_', 'softmax_rows', 'resize_bilinear', 'sds_'))) / 1e3
print(f'configs[{cfg}] iteration: wall {sorted(ts)[1]:.1f} ms {[round(t, 1) for t in ts]}, device-busy {busy:.1f} ms of which SD-network kernels {sds:.1f} ms, {sum(e.count for e in ev)} launches')
rows = []
for e in ev[:28]:
    print(f'  {e.device_time_total / 1e3:8.3f} ms x{e.count:4d}  {e.key[:120]}')
    rows.append([round(e.device_time_total / 1e3, 3), e.count, e.key[:120]])
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'config': cfg, 'wall_ms': ts, 'device_busy_ms': busy, 'sd_network_kernels_ms': sds, 'top': rows},
          open(f"gpurun_out/{os.environ.get('MVIP_ROUND', 'r6')}_config{cfg}_step_kernels.json", 'w'), indent=1)
