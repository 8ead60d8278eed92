"""Soak: N second-stage iterations with the diffusion prior (BASELINE configs[1] shapes, random SD weights) and N of
the hash-grid model; prints loss finiteness, iteration time drift and device-memory growth."""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvip_nerf_amd.trainer import SecondStageTrainer, SyntheticScene
from mvip_nerf_amd.guidance.sd_utils import StableDiffusion
from mvip_nerf_amd.nerf.utils import Pretrain_Model

dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scene = SyntheticScene(bench.H, bench.W, bench.FOCAL, bench.NEAR, bench.FAR, device=dev)
sd = StableDiffusion(dev, False, False)
opt = types.SimpleNamespace(is_rgb_guidance=True, is_colla_guidance=False, is_normal_guidance=False,
                            text='a stone bench in a park', text_normal='', rgb_guidance_scale=7.5,
                            colla_guidance_scale=7.5, normal_guidance_scale=1.5, normal_start=500, lambda_guidance=1)
for name, args_mod, guidance in (('8x256 + SDS', {}, Pretrain_Model(opt, dev, {'SD': sd})),
                                 ('hash grid', {'no_tcnn': False, 'netchunk': 1 << 20, 'lrate': 1e-2}, None)):
    a = bench.make_args()
    for k, v in args_mod.items():
        setattr(a, k, v)
    tr = SecondStageTrainer(a, scene, dev, guidance=guidance)
    tr.step(1000)
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    times, finite = [], True
    for k in range(n):
        t0 = time.perf_counter()
        out = tr.step(1001 + k)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        finite &= all(bool(torch.isfinite(p).all()) for p in tr.grad_vars) if k % 10 == 0 else True
    print(name, 'finite params', finite, 'first10 ms %.2f' % (sum(times[:10]) * 100), 'last10 ms %.2f' % (sum(times[-10:]) * 100),
          'mem growth MB %.1f' % ((torch.cuda.memory_allocated() - m0) / 1e6), 'peak GB %.2f' % (torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del tr
