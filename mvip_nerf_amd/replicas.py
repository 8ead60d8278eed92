"""BASELINE configs[4]: N SPIn-NeRF scenes trained concurrently, ONE independent replica per GPU ("throughput mode").

The reference has no such launcher (a user starts `python run.py --config ...` once per scene, DS_NeRF/run.py:1975-1981);
the path does not shard here, so this is "replicas only" (DESIGN.md section 8): no process group, no collective, no shared
state.  Every replica is its own process with its own device (HIP_VISIBLE_DEVICES), its own seeds, its own
`basedir/expname` (checkpoints, DS_NeRF/run.py:1046-1057) and its own in-process caches (packed weights, hipGraphs,
scratch words) -- nothing is keyed by a path, a port or a name that two replicas could both pick.

    python -m mvip_nerf_amd.replicas --scenes 8 --iters 20 --basedir /tmp/mvip_replicas [--datadirs d1,d2,...]

`--devices 0,0` places several replicas on one GPU (what the 1-GPU test box can exercise).  The parent prints ONE JSON
line: per-replica iterations/s and their sum.  A replica is started BEFORE anything touches a GPU in the parent.
"""
import argparse
import json
import os
import subprocess
import sys
import time
import types


def config_args(**over):
    """Second-stage arguments of the shipped configuration at factor 4 with every guidance term on (configs[4])."""
    a = types.SimpleNamespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=64, alpha_model_path=None,
        netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=3e-3,
        basedir='/tmp/mvip_replicas', expname='scene', ft_path=None, no_reload=True, perturb=1., N_samples=64,
        white_bkgd=True, raw_noise_std=1., dataset_type='llff', no_ndc=True, lindisp=True, sigma_loss=False,
        N_rand=1024, chunk=1 << 15, lrate_decay=10, depth_lambda=0.1, sds_loss_weight=1e-4, no_coarse=False,
        is_normal_guidance=True, is_colla_guidance=True, normalmap_render_factor=2)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def guidance_opt(**over):
    o = types.SimpleNamespace(is_rgb_guidance=True, is_colla_guidance=True, is_normal_guidance=True,
                              text='a stone bench in a park', text_normal='a normal map of a stone bench in a park',
                              rgb_guidance_scale=7.5, colla_guidance_scale=7.5, normal_guidance_scale=1.5,
                              normal_start=500, lambda_guidance=1, uniform_sphere_rate=0)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def run_replica(scene_id, iters, basedir, datadir=None, fixture=None, size=(567, 1008), start_iter=1000, seed=None,
                n_views=10, sd=None):
    """One replica on the CURRENT device: builds its scene (a real LLFF directory, or the committed scene-1 raster
    fixture resampled to `size` with a scene-specific view subset), its own networks and prior, runs `iters` full-guidance
    iterations, writes its checkpoint under basedir/scene_<id>/ and returns its record."""
    import numpy as np
    import torch
    from . import run
    from .guidance.sd_utils import StableDiffusion
    from .nerf.utils import Pretrain_Model
    from .scene import LLFFScene
    from .trainer import SecondStageTrainer
    dev = torch.device('cuda', torch.cuda.current_device())
    seed = 1000 + scene_id if seed is None else seed
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    if datadir is not None:
        scene = LLFFScene.from_llff(datadir, 4, device=dev, seed=seed)
    else:
        views = [(scene_id + 3 * k) % 30 for k in range(n_views)]
        scene = LLFFScene.from_fixture(fixture, size=size, device=dev, views=views, seed=seed)
    args = config_args(basedir=basedir, expname=f'scene_{scene_id}')
    os.makedirs(os.path.join(basedir, args.expname), exist_ok=True)
    sd = sd if sd is not None else StableDiffusion(dev, False, False)
    tr = SecondStageTrainer(args, scene, dev, guidance=Pretrain_Model(guidance_opt(), dev, {'SD': sd}))
    tr.rng = np.random.RandomState(seed)
    losses = []
    n_poses = len(scene.poses)

    def iteration(k):
        # the neighbour views of iteration i are poses [i % 60 - 4 : i % 60 + 5 : 2] (DS_NeRF/run.py:1365-1401): with a
        # full 60-view scene any i works; a reduced fixture scene needs i % 60 inside its view range
        if n_poses >= 60:
            return start_iter + k
        return (start_iter // 60 + 1 + k) * 60 + min(4, n_poses - 1)
    tr.step(iteration(-1), img_i=min(4, n_poses - 1))             # warm-up: weight images, prompt embeddings, workspaces
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(iters):
        loss, _ = tr.step(iteration(k), img_i=int(tr.rng.randint(0, n_poses)))
        losses.append(float(loss))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    path = os.path.join(basedir, args.expname, f'{start_iter + iters:06d}.tar')
    run.save_checkpoint(path, tr.global_step, tr.kw_train, tr.optimizer)
    rec = {'scene': scene_id, 'iters': iters, 'seconds': dt, 'iterations_per_sec': iters / dt, 'losses': losses,
           'checkpoint': path, 'pid': os.getpid(), 'device': os.environ.get('HIP_VISIBLE_DEVICES', ''),
           'param_checksum': float(sum(p.detach().double().sum() for p in tr.grad_vars)),
           'peak_bytes': int(torch.cuda.max_memory_allocated(dev))}
    with open(os.path.join(basedir, args.expname, 'replica.json'), 'w') as f:
        json.dump(rec, f)
    return rec


def launch(n_scenes, iters, basedir, devices=None, datadirs=None, fixture=None, size=(567, 1008), timeout=None):
    """Start n_scenes child processes (one replica each), wait, return their records.  No GPU call in this process."""
    devices = devices if devices is not None else list(range(n_scenes))
    procs = []
    for s in range(n_scenes):
        stale = os.path.join(basedir, f'scene_{s}', 'replica.json')
        if os.path.exists(stale):
            os.remove(stale)
        env = dict(os.environ)
        env['HIP_VISIBLE_DEVICES'] = str(devices[s % len(devices)])
        if len(set(devices[i % len(devices)] for i in range(n_scenes))) < n_scenes:
            env['MVIP_SHARED_DEVICE'] = '1'          # several replicas on one GPU: never trust a cached free-memory answer (ops.py)
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):       # a replica is not a rank
            env.pop(k, None)
        cmd = [sys.executable, '-m', 'mvip_nerf_amd.replicas', '--child', str(s), '--iters', str(iters), '--basedir', basedir,
               '--size', f'{size[0]}x{size[1]}']
        if datadirs:
            cmd += ['--datadirs', datadirs[s % len(datadirs)]]
        if fixture:
            cmd += ['--fixture', fixture]
        procs.append(subprocess.Popen(cmd, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=timeout))
        except subprocess.TimeoutExpired:
            p.kill()                                 # the exact child this launcher started
            codes.append(-9)
    recs = []
    for s in range(n_scenes):
        path = os.path.join(basedir, f'scene_{s}', 'replica.json')
        recs.append(json.load(open(path)) if codes[s] == 0 and os.path.exists(path) else {'scene': s, 'error': codes[s]})
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=8)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--basedir', default='/tmp/mvip_replicas')
    ap.add_argument('--devices', default=None, help='comma list of device indices, one per replica (default 0..scenes-1)')
    ap.add_argument('--datadirs', default=None, help='comma list of LLFF scene directories (default: the raster fixture)')
    ap.add_argument('--fixture', default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests',
                                                      'golden', 'scene1_small.npz'))
    ap.add_argument('--size', default='567x1008')
    ap.add_argument('--child', type=int, default=None)
    a = ap.parse_args()
    size = tuple(int(v) for v in a.size.split('x'))
    if a.child is not None:
        run_replica(a.child, a.iters, a.basedir, datadir=a.datadirs, fixture=None if a.datadirs else a.fixture, size=size)
        return 0
    devices = None if a.devices is None else [int(v) for v in a.devices.split(',')]
    recs = launch(a.scenes, a.iters, a.basedir, devices, a.datadirs.split(',') if a.datadirs else None, a.fixture, size)
    ok = [r for r in recs if 'error' not in r]
    print(json.dumps({'metric': 'second-stage iterations/s, full guidance, factor 4, independent replicas', 'replicas': len(recs),
                      'value': sum(r['iterations_per_sec'] for r in ok), 'unit': 'iterations/s', 'per_replica': recs}))
    return 0 if len(ok) == len(recs) else 3


if __name__ == '__main__':
    sys.exit(main())
