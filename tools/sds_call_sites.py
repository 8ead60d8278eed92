"""Which Python call sites launch the plane-writing / scale passes of one SDS step (mvip_split_planes*, mvip_absmax_scale,
mvip_gemm_pack_a, mvip_im2col_split_planes, mvip_col2im), how often and with how many elements: the passes a producer-side
operand format would remove.  Eager step (graphs off)."""
import collections, json, os, sys, traceback
os.environ['MVIP_SDS_GRAPHS'] = '0'
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvip_nerf_amd import ops
from mvip_nerf_amd.guidance import sd_utils
dev = torch.device('cuda', 0)
WATCH = ('mvip_split_planes', 'mvip_split_planes_strided', 'mvip_split_planes_upsample2', 'mvip_absmax_scale', 'mvip_gemm_pack_a',
         'mvip_im2col_split_planes', 'mvip_col2im', 'mvip_groupnorm_split_planes', 'mvip_groupnorm_split_planes_moments',
         'mvip_groupnorm_split_planes_moments_out', 'mvip_group_norm_moments')
calls = collections.Counter()
orig = ops.call
def rec(name, *a):
    if name in WATCH:
        st = traceback.extract_stack(limit=8)[:-1]
        site = ' < '.join(f'{os.path.basename(f.filename)}:{f.name}:{f.lineno}' for f in reversed(st) if 'mvip_nerf_amd' in f.filename)[:200]
        calls[(name, site)] += 1
    return orig(name, *a)
sd = sd_utils.StableDiffusion(dev, False, False)
pred = torch.rand(1, 3, bench.H, bench.W, device=dev, requires_grad=True)
mask = torch.zeros(1, 1, bench.H, bench.W, device=dev)
mask[:, :, 137:241, 196:307] = 1
def step(i):
    pred.grad = None
    (1e-4 * sd.train_step_sd(i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
step(1000)
ops.call = rec
for m in list(sys.modules.values()):
    if m is not None and getattr(m, '__name__', '').startswith('mvip_nerf_amd') and getattr(m, 'call', None) is orig:
        m.call = rec
step(1001)
torch.cuda.synchronize()
rows = [{'entry': k[0], 'site': k[1], 'calls': v} for k, v in calls.most_common()]
json.dump(rows, open('gpurun_out/r5_sds_call_sites.json', 'w'), indent=1)
tot = collections.Counter()
for r in rows:
    tot[r['entry']] += r['calls']
print(dict(tot))
for r in rows[:60]:
    print(r['calls'], r['entry'], '|', r['site'])
