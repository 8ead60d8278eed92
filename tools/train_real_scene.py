"""End-to-end on REAL data (SPIn-NeRF scene 1: 30 views at 1/16 resolution from tests/golden/scene1_small.npz, or with
--fixture f8 the factor-8 fixture tests/golden/scene1_f8.npz -- 283 x 504, BASELINE configs[0]'s geometry, 15 training views
and view 30 held out):
photometric NeRF training with the HIP renderer, held-out PSNR, and the north-star parity clause --
PSNR against ground truth of the HIP render vs the CPU-oracle render of the SAME trained weights."""
import argparse, json, os, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvip_nerf_amd import run, ops
from mvip_nerf_amd.run_nerf_helpers import img2mse, mse2psnr

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=3000)
ap.add_argument('--rays', type=int, default=4096)
ap.add_argument('--train-precision', type=int, default=0)
ap.add_argument('--tcnn', action='store_true', help='train the hash-grid model (NeRF_TCNN) instead of the 8x256 MLPs')
ap.add_argument('--half2-atomics', action='store_true', help='hash-grid model: half-pair atomics for the table gradient')
ap.add_argument('--oracle-view', type=int, default=1, help='render this many held-out views with the CPU oracle')
ap.add_argument('--fixture', default='small', choices=['small', 'f8'])
ap.add_argument('--oracle-stride', type=int, default=1, help='the oracle renders every k-th pixel of the held-out view')
a = ap.parse_args()
dev = torch.device('cuda', 0)
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'scene1_small.npz' if a.fixture == 'small' else 'scene1_f8.npz'))
images = torch.from_numpy(d['images'].astype(np.float32) / 255.).to(dev)            # [30,141,252,3] / [16,283,504,3]
poses = torch.from_numpy(d['poses'][:, :, :4]).to(dev)
N, H, W, _ = images.shape
focal = float(d['poses'][0, 2, 4]) * (W / float(d['poses'][0, 1, 4])) if a.fixture == 'f8' else float(d['poses'][0, 2, 4]) * (H / float(d['poses'][0, 0, 4]))
near, far = float(d['bds'].min() * .9), float(d['bds'].max() * 1.)
i_test = [4, 14, 24] if a.fixture == 'small' else [int(np.nonzero(d['views'] == int(d['held_out_view']))[0][0])]
i_train = [i for i in range(N) if i not in i_test]
args = types.SimpleNamespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=64,
                             alpha_model_path=None, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                             netchunk=65536, lrate=5e-4, basedir='/tmp/mvip_real', expname='none', ft_path=None,
                             no_reload=True, perturb=1., N_samples=64, white_bkgd=False, raw_noise_std=1.,
                             dataset_type='llff', no_ndc=True, lindisp=False, sigma_loss=False)
torch.manual_seed(0)
if a.tcnn:
    args.netchunk, args.lrate = 1 << 20, 1e-2
    tr, te, _, grad_vars, opt = run.create_nerf_tcnn(args, device=dev)
else:
    tr, te, _, grad_vars, opt = run.create_nerf(args, device=dev)
kw_tr = {k: v for k, v in tr.items() if k not in ('ndc', 'use_viewdirs')}
for _n in (tr['network_fn'], tr['network_fine']):
    _n.train_precision = a.train_precision
    if a.tcnn and a.half2_atomics:
        _n.table_grad_atomics = 'half2'
g = torch.Generator(device=dev).manual_seed(0)
t0 = time.perf_counter()
log = []
for it in range(a.iters):
    v = i_train[int(torch.randint(0, len(i_train), (1,), generator=g, device=dev))]
    sel = torch.randint(0, H * W, (a.rays,), generator=g, device=dev)
    rows = ops.ray_rows_from_pose(poses[v], H, W, focal, near, far, sel=sel)
    r = run.batchify_rays(rows, 1 << 15, **kw_tr)
    tgt = images[v].reshape(-1, 3)[sel]
    loss = img2mse(r['rgb_map'], tgt) + img2mse(r['rgb0'], tgt)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    for pg in opt.param_groups:
        pg['lr'] = args.lrate * (0.1 ** (it / 250000))
    if it % 500 == 0 or it == a.iters - 1:
        log.append((it, float(loss)))
        print(it, float(loss), flush=True)
torch.cuda.synchronize()
train_s = time.perf_counter() - t0
res = {'train_precision': a.train_precision, 'iters': a.iters, 'rays_per_iter': a.rays, 'train_seconds': train_s, 'ms_per_iter': train_s / a.iters * 1e3,
       'train_rays_per_sec': a.iters * a.rays / train_s, 'H': H, 'W': W, 'views_train': len(i_train), 'loss_log': log}
kw_te = dict(te, near=near, far=far)
psnr_hip, renders = [], []
with torch.no_grad():
    for v in i_test:
        rgb = run.render(H, W, focal, chunk=1 << 15, c2w=poses[v], **kw_te)[0]
        renders.append(rgb)
        psnr_hip.append(float(mse2psnr(img2mse(rgb, images[v]))))
res['psnr_heldout_hip'] = psnr_hip
if a.tcnn:
    res['model'] = 'NeRF_TCNN (hash grid)' + (', half2 table-gradient atomics' if a.half2_atomics else '')
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'real_scene_r1_tcnn_half2.json' if a.half2_atomics else
                                     'real_scene_r1_tcnn.json'), 'w'), indent=1)
    sys.exit(0)
# the same trained weights rendered by the CPU oracle (the reference restatement)
from oracle import nerf_oracle as O
pc = {k: p.detach().cpu() for k, p in tr['network_fn'].named_parameters()}
pf = {k: p.detach().cpu() for k, p in tr['network_fine'].named_parameters()}
torch.set_num_threads(min(os.cpu_count() or 1, 32))
cmp = []
for k in range(a.oracle_view):
    v = i_test[k]
    ro, rd = O.get_rays(H, W, focal, poses[v].cpu())
    sel = torch.arange(0, H * W, a.oracle_stride)
    rows = O.assemble_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], near, far)
    with torch.no_grad():
        parts = [O.render_rays(rows[i:i + 4096], pc, pf, 64, 64, lindisp=False, white_bkgd=False)['rgb_map']
                 for i in range(0, rows.shape[0], 4096)]
    rgb_o = torch.cat(parts, 0)
    gt = images[v].cpu().reshape(-1, 3)[sel]
    hip_sel = renders[k].cpu().reshape(-1, 3)[sel]
    p_o = float(O.mse2psnr(O.img2mse(rgb_o, gt)))
    p_h = float(O.mse2psnr(O.img2mse(hip_sel, gt)))          # on the oracle's pixel subset (stride 1: the whole view)
    mse_ho = float(O.img2mse(rgb_o, hip_sel))
    cmp.append({'view': int(d['views'][v]), 'psnr_gt_oracle': p_o, 'psnr_gt_hip': p_h, 'delta_dB': p_h - p_o,
                'psnr_hip_vs_oracle': float(-10 * np.log10(max(mse_ho, 1e-20)))})
res['hip_vs_oracle_same_weights'] = cmp
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', f'real_scene_{a.fixture}_prec{a.train_precision}.json'), 'w'), indent=1)
