"""Hash-grid model (NeRF_TCNN) timings: encode kernel fwd/bwd against its algorithmic HBM bytes, module forward,
full-frame render through run.render (generic network path)."""
import json, os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd import ops, run
from mvip_nerf_amd.run_nerf_helpers_tcnn import NeRF_TCNN

dev = torch.device('cuda', 0)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


net = NeRF_TCNN(seed=0).to(dev)
P = 1 << 23
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(P, 3, device=dev, generator=g) * 2 - 1) * 4.0
tab = net.encoder.params.detach().clone().requires_grad_(True)
t_f = timeit(lambda: ops.hashgrid_encode(x, tab.detach(), net.levels, 100.0))
f = ops.hashgrid_encode(x, tab, net.levels, 100.0)
dout = torch.randn_like(f)
t_b = timeit(lambda: torch.autograd.grad(ops.hashgrid_encode(x, tab, net.levels, 100.0), tab, dout)) - t_f
alg_f = P * (12 * 16 + 16 * 8 * 8 + 128)         # x re-read per level, 8 corners x 8 B x 16 levels, 128 B out
alg_b = P * (12 * 16 + 128 + 16 * 8 * 8 * 2)     # atomics: read-modify-write of 8 B per corner
print(json.dumps({'kernel': 'hg_forward', 'points': P, 'ms': t_f * 1e3, 'algorithmic_GBps': alg_f / t_f / 1e9,
                  'frac_of_8TBps': alg_f / t_f / 8e12, 'points_per_sec': P / t_f}), flush=True)
print(json.dumps({'kernel': 'hg_backward', 'points': P, 'ms': t_b * 1e3, 'algorithmic_GBps': alg_b / t_b / 1e9,
                  'frac_of_8TBps': alg_b / t_b / 8e12}), flush=True)
inp = torch.cat([x[:1 << 20], torch.nn.functional.normalize(torch.randn(1 << 20, 3, device=dev), dim=-1)], -1)
with torch.no_grad():
    net.fused_inference = False
    t_m = timeit(lambda: net(inp))
    net.fused_inference = True
    t_u = timeit(lambda: net(inp))
    # ray-ordered points (64 consecutive samples along each ray), the order the renderer produces
    o = (torch.rand(1 << 14, 1, 3, device=dev) * 2 - 1) * 0.3
    dd = torch.nn.functional.normalize(torch.randn(1 << 14, 1, 3, device=dev), dim=-1)
    zz = torch.linspace(1.2, 7.7, 64, device=dev)[None, :, None]
    inp_r = torch.cat([(o + dd * zz).reshape(-1, 3), dd.expand(-1, 64, -1).reshape(-1, 3)], -1)
    t_ur = timeit(lambda: net(inp_r))
print(json.dumps({'module_forward_points_per_sec': (1 << 20) / t_m, 'ms_per_1M_points': t_m * 1e3,
                  'fused_points_per_sec': (1 << 20) / t_u, 'fused_ms_per_1M_points': t_u * 1e3,
                  'fused_ray_ordered_ms_per_1M_points': t_ur * 1e3}), flush=True)

args = types.SimpleNamespace(use_viewdirs=True, N_importance=64, alpha_model_path=None, netchunk=1 << 20, lrate=1e-2,
                             basedir='/tmp/x', expname='none', ft_path=None, no_reload=True, perturb=0., N_samples=64,
                             white_bkgd=True, raw_noise_std=0., dataset_type='llff', no_ndc=True, lindisp=True)
kw_train, kw_test, _, grad_vars, opt = run.create_nerf_tcnn(args, dev)
H, W, focal = 378, 504, 383.65
pose = torch.tensor([[1., 0, 0, 0], [0, 1., 0, 0], [0, 0, 1., 0.3]], device=dev)
with torch.no_grad():
    t_r = timeit(lambda: run.render(H, W, focal, chunk=1 << 15, c2w=pose, near=1.2, far=7.74, **kw_test), n=3)
print(json.dumps({'render_hashgrid_rays_per_sec': H * W / t_r, 'ms_per_frame': t_r * 1e3,
                  'workload': '378x504, 64+128 samples, NeRF_TCNN coarse+fine'}), flush=True)
