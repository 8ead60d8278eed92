"""Second-stage training iteration of MVIP-NeRF (the loop body of DS_NeRF/run.py:862-1039) on the
HIP renderer, one process per GPU.

Per iteration, as in the reference:
  1. pick a training view; render its MASKED pixels with grad (run.py:864-921) and scatter them
     into a copy of the image -> combin_rgb [1,3,H,W] (run.py:924-931);
  2. optionally render a reduced-resolution frame -> depth -> normal map (run.py:948-965) and
     <=5 neighbour views (run.py:968-974) for the normal / collaborative SDS terms;
  3. render a batch of unmasked rays (colour supervision, rgb_map and rgb0) and a batch of
     "inpainted-depth" rays (disparity supervision)  (run.py:978-984);
  4. loss = mse(rgb2) + depth_lambda*mse(disp2) + mse(rgb0) + sds_loss_weight*SDS  (run.py:1000-1027);
  5. backward, Adam step, lr = lrate * 0.1^(step / (lrate_decay*1000))  (run.py:1030-1039).

Two kinds of scene feed it:
  * `scene.LLFFScene` (real data, or arrays shaped like `load_llff_data`'s return values): per-view masks, and the
    supervision batches are the reference's pre-baked fp16 ray records drawn ACROSS views (run.py:613-717, :887-908)
    -- with it two iterations reproduce the reference's own loop (tests/golden/trainer_two_steps.npz);
  * `SyntheticScene` (SURVEY.md 8(d) bench inputs): one shared mask, supervision pixels of the chosen view with
    rays generated on the GPU in fp32.
Deliberate differences: the masked rays come from (pose, pixel index) on the GPU instead of a full-frame
get_rays + 94 % discard (run.py:869-883); the neighbour views that receive no gradient in the reference's
collaborative term are rendered without autograd, and so is the COARSE pass of every render of which only the fine
outputs are used (masked set, normal frame, last neighbour view: the coarse network gets exactly zero gradient from
them, run.py:1812 detaches the resampled depths); the Tk GUI thread, PNG dumps and host syncs are gone.

Multi-GPU (world > 1, torch.distributed over RCCL): every per-step ray set is sharded by index
across ranks (rank r takes rays r::world), frames are assembled with all_gather (the local shard keeps its autograd
history), the <= 7 SDS terms are evaluated by different ranks (`sds_shard`: one async 64 KB all_reduce + <= 3 frame
broadcasts), and the two MLPs' gradients are summed with ONE all_reduce of a flat 1,191,688-float bucket (4.77 MB),
after which every rank applies the identical Adam update.  Shard losses are weighted n_local / n_global so the
reduced gradient is the single-process one for any shard sizes.
"""
import math
import os

import numpy as np
import torch

from . import ops, run, sds_shard
from .dist_utils import shard, all_gather_ragged, FlatGradBucket, OverlappedGradBuckets
from .run_nerf_helpers import img2mse


class SyntheticScene:
    """SURVEY.md §8(d) synthetic stand-in for SPIn-NeRF scene 1: 60 orbit poses, random target
    images, a centred rectangular inpainting mask (~6 % of the frame), constant near/far."""
    sets = None                                     # no pre-baked ray records: supervision pixels of the chosen view

    def __init__(self, H=378, W=504, focal=383.65, near=1.2, far=7.74, n_views=60, mask_hw=(104, 111),
                 device='cuda', seed=0):
        self.H, self.W, self.focal, self.near, self.far = H, W, focal, near, far
        g = torch.Generator(device='cpu').manual_seed(seed)
        self.poses = torch.stack([self._pose(k) for k in range(n_views)], 0).to(device)
        self.images = torch.rand((n_views, H, W, 3), generator=g).to(device)
        self.depths = (torch.rand((n_views, H, W), generator=g) * 0.5 + 0.2).to(device)   # disparity targets
        mh, mw = mask_hw
        mask = torch.zeros((H, W), dtype=torch.bool)
        y0, x0 = (H - mh) // 2, (W - mw) // 2
        mask[y0:y0 + mh, x0:x0 + mw] = True
        self.mask = mask.to(device)
        self.masks = self.mask[None].expand(n_views, H, W)
        flat = self.mask.reshape(-1)
        self.masked_idx = torch.nonzero(flat, as_tuple=False).reshape(-1).to(device)       # int64, raster order
        self.unmasked_idx = torch.nonzero(~flat, as_tuple=False).reshape(-1).to(device)
        self.i_train = np.arange(n_views)

    def masked_idx_of(self, view):
        return self.masked_idx

    def mask_of(self, view):
        return self.mask

    @staticmethod
    def _pose(k):
        th = math.radians(6.0 * k)
        c, s = math.cos(th), math.sin(th)
        return torch.tensor([[c, 0., s, 0.3 * s], [0., 1., 0., 0.], [-s, 0., c, 0.3 * c]], dtype=torch.float32)


class SecondStageTrainer:
    def __init__(self, args, scene, device, guidance=None, world=1, rank=0, dist=None, view_shard=None):
        self.args, self.scene, self.device = args, scene, device
        self.world, self.rank, self.dist = world, rank, dist
        make = run.create_nerf if getattr(args, 'no_tcnn', True) else run.create_nerf_tcnn      # run.py:541-546
        (self.kw_train, self.kw_test, self.start, self.grad_vars, self.optimizer) = make(args, device)
        self.global_step = self.start
        self.guidance = guidance                       # Pretrain_Model-like object with cal_loss(), or None
        self.rng = np.random.RandomState(1234)         # same draw on every rank (view choice must agree)
        self.N_rand = args.N_rand
        # one flat bucket, reduced as two asynchronous halves [coarse network | fine network] (dist_utils.OverlappedGradBuckets);
        # MVIP_OVERLAP_ALLREDUCE=0 restores the single blocking all_reduce after the backward (A/B switch, same values)
        n_coarse = len(list(self.kw_train['network_fn'].parameters())) if self.kw_train.get('network_fn') is not None else 0
        if (world > 1 and os.environ.get('MVIP_OVERLAP_ALLREDUCE', '1') != '0' and 0 < n_coarse < len(self.grad_vars)
                and hasattr(torch.nn.Parameter, 'register_post_accumulate_grad_hook')):
            self.bucket = OverlappedGradBuckets(self.grad_vars, [n_coarse, len(self.grad_vars) - n_coarse])
        else:
            self.bucket = FlatGradBucket(self.grad_vars)
        # SDS terms owned by different ranks (sds_shard); None = whenever there is more than one rank
        self.view_shard = (world > 1) if view_shard is None else bool(view_shard)
        if world > 1:                                  # identical initial weights on every rank
            with torch.no_grad():
                for p in self.grad_vars:
                    dist.broadcast(p.detach(), src=0)  # shares p's version counter: the packed images are rebuilt
            for key in ('network_fn', 'network_fine'):
                net = self.kw_train.get(key)
                if net is not None and hasattr(net, 'invalidate_packed'):
                    net.invalidate_packed()

    # -- helpers ---------------------------------------------------------------------------------
    def _shard(self, idx):
        return shard(idx, self.rank, self.world)

    def _kw(self, kw):
        return {k: v for k, v in kw.items() if k not in ('ndc', 'use_viewdirs', 'near', 'far')}

    def _render_pixels(self, pose, sel, hwf=None, **kw):
        sc = self.scene
        H, W, focal = hwf if hwf is not None else (sc.H, sc.W, sc.focal)
        rows = ops.ray_rows_from_pose(pose, H, W, focal, sc.near, sc.far, sel=sel)
        return run.batchify_rays(rows, self.args.chunk, **self._kw(kw))

    def _render_records(self, rays, **kw):
        """render(H, W, focal, rays=[2, B, 3] fp16 records) of run.py:978-984: the reference's own row assembly
        (view directions normalised in the records' precision), this rank's strided share of the batch."""
        sc = self.scene
        rows = run._assemble_rows_general(sc.H, sc.W, sc.focal, rays[0], rays[1], False, sc.near, sc.far, True, None, None)
        return run.batchify_rays(rows, self.args.chunk, **self._kw(kw))

    def _render_frame(self, pose, hwf, key, **kw):
        """One full frame at (H, W, focal) = hwf, rays sharded over the ranks; returns the assembled
        [H*W, ...] map `key` (the local shard keeps its autograd history) and the local ray count."""
        H, W, _ = hwf
        idx = torch.arange(H * W, device=self.device)
        sel = self._shard(idx)
        ret = self._render_pixels(pose, sel, hwf=hwf, **kw)
        return all_gather_ragged(ret[key], H * W, self.rank, self.world, self.dist), sel.numel()

    def _normal_map(self, pose):
        """run.py:948-965: reduced-resolution depth (train kwargs, with grad) -> back-projection -> 31x31
        plane-fit normals -> (n + 1) / 2, [1, 3, H_r, W_r]."""
        sc, f = self.scene, self.args.normalmap_render_factor
        H_r, W_r, focal_r = sc.H // f, sc.W // f, sc.focal / f
        depth, n = self._render_frame(pose, (H_r, W_r, focal_r), 'depth_map', retraw=True, coarse_grad=False,
                                      **self.kw_train)
        K = torch.tensor([[focal_r, 0, W_r / 2], [0, focal_r, H_r / 2], [0, 0, 1]], dtype=torch.float32)
        points = run.depth2xyz_torch(depth.reshape(H_r, W_r), K)
        points = points.unsqueeze(0).transpose(2, 3).transpose(1, 2)
        return (run.depth2normal_geo(points) + 1) / 2, n

    def _colla_views(self, i):
        """run.py:968-974 / render_path_4view (:1365-1401): the <=5 neighbour views [it-4 : it+5 : 2] of
        it = i % 60 at the reduced resolution, TEST kwargs, with grad; masks stay at full resolution."""
        sc, f = self.scene, self.args.normalmap_render_factor
        hwf = (sc.H // f, sc.W // f, sc.focal / f)
        it = i % 60
        lo, hi = max(0, it - 4), min(len(sc.poses), it + 5)
        rgbs, n = [], 0
        views = list(range(lo, hi, 2))
        for k in views:
            # train_step_colla_sds overwrites its loss on every view (sd_utils.py:575-597): only the LAST view's
            # graph ever receives a gradient, the earlier views enter as constants.  Rendering them without
            # autograd changes no value and no gradient and skips their activation stash and backward.
            with torch.set_grad_enabled(k == views[-1]):
                rgb, m = self._render_frame(sc.poses[k], hwf, 'rgb_map', retraw=True, need_alpha=True,
                                            coarse_grad=False, **self.kw_test)
            rgbs.append(rgb.reshape(hwf[0], hwf[1], 3))
            n += m if k == views[-1] else 0
        masks = sc.masks[lo:min(len(sc.masks), it + 5):2]
        return torch.stack(rgbs, 0).permute(0, 3, 1, 2), masks.float().unsqueeze(1), n

    def _allreduce_grads(self):
        self.bucket.all_reduce(self.dist, self.world)      # the 4.77 MB bucket over RCCL/xGMI (waits for the async halves)

    # -- the SDS terms, owned by different ranks ----------------------------------------------------------------
    def _sds_view_sharded(self, i, combin_rgb, mask, normal_map, rgbs4, mask4):
        """The terms `Pretrain_Model.cal_loss` would sum (DS_NeRF/nerf/utils.py:280-302: RGB; collaborative if i > 0;
        normal if i > normal_start), each evaluated by ONE rank (sds_shard.evaluate); returns
        sds_loss_weight * (sum of terms) as a surrogate whose gradient w.r.t. the frames is the terms' gradient."""
        pm = self.guidance
        opt, sd = pm.opt, pm.guidance['SD']
        w = float(self.args.sds_loss_weight)
        pm.global_step += 1
        base = 7919 * (i + 1)                           # a term's noise depends on (iteration, term), not on its owner
        terms = []
        if opt.is_rgb_guidance:
            terms.append(sds_shard.Term('rgb', 2, lambda _, b=base: sd.image_grad(
                'rgb', i, mask, opt.text, combin_rgb, opt.rgb_guidance_scale, w, seed=b), image=combin_rgb))
        if opt.is_normal_guidance and i > opt.normal_start:
            terms.append(sds_shard.Term('normal', 2, lambda _, b=base + 1: sd.image_grad(
                'normal', i, mask, opt.text_normal, normal_map, opt.normal_guidance_scale, w,
                normal_start=opt.normal_start, seed=b), image=normal_map))
        if opt.is_colla_guidance and i > 0:
            V = rgbs4.shape[0]
            last = V - 1
            terms.append(sds_shard.Term('colla_last', 2, lambda share, b=base + 2 + last: sd.colla_last_view_image_grad(
                last, mask4[last:last + 1], opt.text, rgbs4[last:last + 1], opt.colla_guidance_scale, share, w, seed=b),
                image=rgbs4[last:last + 1], needs_latent_sum=True))
            for k in range(last):
                terms.append(sds_shard.Term(f'colla_{k}', 1, lambda k=k, b=base + 2 + k: sd.colla_view_share(
                    k, mask4[k:k + 1], opt.text, rgbs4[k:k + 1], opt.colla_guidance_scale, seed=b)))
        if not terms:
            return None
        return sds_shard.evaluate(terms, self.rank, self.world, self.dist, self.device)

    # -- one iteration -----------------------------------------------------------------------------
    def step(self, i, img_i=None, records=None):
        """One iteration.  `img_i` overrides the random view choice and `records` = (clf [B,3,4], inp [B,3,4]) the
        drawn supervision batches (tests replay the reference's draws with them)."""
        args, sc = self.args, self.scene
        if img_i is None:
            img_i = int(self.rng.choice(sc.i_train))
        pose = sc.poses[img_i]
        # 1. masked pixels of the chosen view, with grad
        masked_idx = sc.masked_idx_of(img_i)
        sel = self._shard(masked_idx)
        r1 = self._render_pixels(pose, sel, retraw=True, coarse_grad=False, **self.kw_train)     # only rgb_map is used
        rgb_masked = r1['rgb_map']
        rays_rendered = sel.numel()

        loss_sds = None
        if self.guidance is not None:
            # assemble the frame on every rank (local shard keeps its autograd history)
            rgb_all = all_gather_ragged(rgb_masked, masked_idx.numel(), self.rank, self.world, self.dist)
            combin = sc.images[img_i].detach().clone().reshape(-1, 3)
            combin = combin.index_put((masked_idx,), rgb_all).reshape(sc.H, sc.W, 3)
            combin_rgb = combin.permute(2, 0, 1).unsqueeze(0)
            mask = sc.mask_of(img_i).float().reshape(1, 1, sc.H, sc.W)
            normal_map = rgbs4 = mask4 = None
            if getattr(args, 'is_normal_guidance', False):      # run.py:948 (colla without normal is a NameError
                normal_map, n = self._normal_map(pose)          # in the reference, run.py:1003; here it passes None)
                rays_rendered += n
            if getattr(args, 'is_colla_guidance', False):
                rgbs4, mask4, n = self._colla_views(i)
                rays_rendered += n
            if self.view_shard and hasattr(self.guidance, 'guidance') and 'SD' in getattr(self.guidance, 'guidance', {}) \
                    and hasattr(self.guidance.guidance['SD'], 'image_grad'):
                loss_sds = self._sds_view_sharded(i, combin_rgb, mask, normal_map, rgbs4, mask4)
            else:
                if self.world > 1:                     # replicated SDS terms: identical noise on every rank
                    for sd in self.guidance.guidance.values():
                        if hasattr(sd, 'seed_generator'):
                            sd.seed_generator(777 + i)
                loss_sds = args.sds_loss_weight * self.guidance.cal_loss(i, rgbs4, normal_map, None, combin_rgb, None,
                                                                         mask, mask4, 1)

        # 3. supervision batches: unmasked colour rays and inpainted-depth rays
        if getattr(sc, 'sets', None) is not None or records is not None:
            rays_c, target_clf, _ = sc.next_batch('rays_rgb_clf', self.N_rand, None if records is None else records[0])
            rays_d, _, target_inp = sc.next_batch('rays_inp', self.N_rand, None if records is None else records[1])
            n_clf, n_inp = rays_c.shape[1], rays_d.shape[1]
            # (min: a batch with fewer rays than ranks leaves the last ranks an EMPTY shard -- torch.arange refuses start > end;
            #  found by tests/test_distributed_gpu.py::test_eight_ranks_issue_one_collective_sequence)
            take = lambda t, dim: t if self.world == 1 else t.index_select(
                dim, torch.arange(min(self.rank, t.shape[dim]), t.shape[dim], self.world, device=t.device))
            rays_c, target_clf = take(rays_c, 1), take(target_clf, 0)
            rays_d, target_inp = take(rays_d, 1), take(target_inp, 0)
            r2 = self._render_records(rays_c, retraw=True, **self.kw_train)
            r3 = self._render_records(rays_d, retraw=True, **self.kw_train)
            n_clf_local, n_inp_local = rays_c.shape[1], rays_d.shape[1]
        else:
            g = torch.Generator(device=self.device).manual_seed(10007 * (i + 1))
            pick = sc.unmasked_idx[torch.randint(0, sc.unmasked_idx.numel(), (self.N_rand,), device=self.device,
                                                  generator=g)]
            n_clf = n_inp = self.N_rand
            pick = self._shard(pick)
            r2 = self._render_pixels(pose, pick, retraw=True, **self.kw_train)
            target_clf = sc.images[img_i].reshape(-1, 3)[pick]
            pick_d = sc.masked_idx[torch.randint(0, sc.masked_idx.numel(), (self.N_rand,), device=self.device,
                                                 generator=g)]
            pick_d = self._shard(pick_d)
            r3 = self._render_pixels(pose, pick_d, retraw=True, **self.kw_train)
            target_inp = sc.depths[img_i].reshape(-1)[pick_d]
            n_clf_local, n_inp_local = pick.numel(), pick_d.numel()
        rays_rendered += n_clf_local + n_inp_local

        # 4. losses (run.py:1000-1027); means over the GLOBAL batch: a shard's mean enters with n_local / n_global
        self.optimizer.zero_grad(set_to_none=True)
        wc, wi = n_clf_local / max(n_clf, 1), n_inp_local / max(n_inp, 1)
        img_loss = img2mse(r2['rgb_map'], target_clf.float()) * wc if n_clf_local else 0.
        depth_loss = img2mse(r3['disp_map'], target_inp.float()) * wi if n_inp_local else 0.
        loss = img_loss + args.depth_lambda * depth_loss
        if 'rgb0' in r2 and not getattr(args, 'no_coarse', False) and n_clf_local:
            loss = loss + img2mse(r2['rgb0'], target_clf.float()) * wc
        if loss_sds is not None:
            loss = loss + loss_sds
        elif self.guidance is None:
            # without a diffusion prior the masked render still has to be back-propagated for the
            # iteration to have the reference's cost structure: a plain colour loss stands in
            wm = sel.numel() / max(masked_idx.numel(), 1)
            loss = loss + args.sds_loss_weight * img2mse(rgb_masked, sc.images[img_i].reshape(-1, 3)[sel]) * wm
        if isinstance(self.bucket, OverlappedGradBuckets):
            self.bucket.begin(self.dist, self.world)       # the coarse half is reduced while the masked render's backward runs
        try:
            loss.backward()
        except BaseException:
            # peers that did not fail are (or will be) waiting in finish(): issue this rank's share of the collectives
            # so the error surfaces as an error here instead of a hang there
            if isinstance(self.bucket, OverlappedGradBuckets):
                self.bucket.abort()
            raise
        self._allreduce_grads()
        self.optimizer.step()

        # 5. lr schedule (run.py:1035-1039)
        new_lrate = args.lrate * (0.1 ** (self.global_step / (args.lrate_decay * 1000)))
        for pg in self.optimizer.param_groups:
            pg['lr'] = new_lrate
        self.global_step += 1
        return loss.detach(), rays_rendered
