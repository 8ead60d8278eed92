"""Real-scene input of the second stage: what `train()` builds between `load_llff_data` and its loop
(DS_NeRF/run.py:382-418 bounds / intrinsics, :613-717 the pre-baked ray sets) as an object the trainer consumes.

`build_ray_sets` restates run.py:613-711 in numpy, bit for bit (pinned by tests/golden/ray_sets.npz, produced by
executing the reference's own statements): every pixel of every training view becomes a [3, 4] fp16 record
(origin | label, direction | label, colour | label); `rays_rgb_clf` keeps the records with label 0 (unmasked
pixels: colour supervision), `rays_rgb` those with label 1, and `rays_inp` carries the inpainted depth as its label.

The reference then filters `rays_inp` with a mask computed from the ALREADY FILTERED `rays_rgb` (run.py:712-713),
which raises IndexError.  Here `rays_inp` is restricted to the UNMASKED pixels, which is what
the surrounding comments of the reference say the term is ("unmasked RGBD supervision", "compute the unmasked depth
loss", run.py:977, :1017); `inp_pixels='masked'` selects the other reading (SPIn-NeRF's: inpainted depth inside the
mask), `'all'` no filter.

`LLFFScene` keeps everything on the GPU (60 views x 567 x 1008 x 12 fp16 = 0.8 GB of records: sized for 288 GB of
HBM, no host DataLoader in the loop) and draws batches the way `DataLoader(shuffle=True)` does: a fresh random
permutation per epoch, consecutive slices of N_rand records.
"""
import numpy as np
import torch

from .run_nerf_helpers import get_rays_np


def build_ray_sets(images, poses, masks, inpainted_depths, hwf, i_train, inp_pixels='unmasked'):
    """images [N,H,W,3], poses [N,3,4+], masks [N,H,W], inpainted_depths [N,H,W] (numpy), hwf = (H, W, focal).
    Returns dict(rays_rgb [*,3,4], rays_rgb_clf, rays_rgb_sds, rays_inp) of fp16 arrays (run.py:613-711)."""
    H, W, focal = int(hwf[0]), int(hwf[1]), hwf[2]

    def records(labels_hw):
        rays = np.stack([get_rays_np(H, W, focal, p) for p in poses[:, :3, :4]], 0)          # [N, ro+rd, H, W, 3]
        labels = np.expand_dims(labels_hw, axis=-1)                                         # [N, H, W, 1]
        labels = np.repeat(labels[:, None], 3, axis=1)                                      # [N, 3, H, W, 1]
        rec = np.concatenate([rays, images[:, None]], 1)                                    # [N, ro+rd+rgb, H, W, 3]
        rec = np.concatenate([rec, labels], -1)                                             # [N, 3, H, W, 4]
        rec = np.transpose(rec, [0, 2, 3, 1, 4])                                            # [N, H, W, 3, 4]
        rec = np.stack([rec[i] for i in i_train], 0)
        return np.reshape(rec, [-1, 3, 4]).astype(np.float16)

    rays_rgb = records(masks)
    rays_inp = records(inpainted_depths)
    label = rays_rgb[:, :, 3]
    out = {'rays_rgb_clf': rays_rgb[label == 0].reshape(-1, 3, 4), 'rays_rgb_sds': rays_rgb.reshape(-1, 3, 4)}
    if inp_pixels == 'unmasked':
        out['rays_inp'] = rays_inp[label == 0].reshape(-1, 3, 4)
    elif inp_pixels == 'masked':
        out['rays_inp'] = rays_inp[label == 1].reshape(-1, 3, 4)
    else:
        out['rays_inp'] = rays_inp
    out['rays_rgb'] = rays_rgb[label == 1].reshape(-1, 3, 4)
    return out


class _EpochSampler:
    """DataLoader(shuffle=True, batch_size=B) on the device: a fresh permutation per epoch, consecutive slices;
    the last (short) batch of an epoch is returned short, as the DataLoader does (drop_last=False)."""

    def __init__(self, n, device, seed):
        self.n, self.device = n, device
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.perm, self.pos = None, 0

    def next(self, batch):
        if self.perm is None or self.pos >= self.n:
            self.perm = torch.randperm(self.n, device=self.device, generator=self.gen)
            self.pos = 0
        idx = self.perm[self.pos:self.pos + batch]
        self.pos += batch
        return idx


class LLFFScene:
    """The fields `SecondStageTrainer` reads, from `load_llff_data`'s return values (or arrays shaped like them)."""

    def __init__(self, images, poses, bds, masks, inpainted_depths, device='cuda', i_train=None, no_ndc=True,
                 inp_pixels='unmasked', seed=0, build_sets=True):
        images = np.asarray(images, np.float32)
        poses = np.asarray(poses, np.float32)
        hwf = poses[0, :3, -1]                                            # run.py:391
        self.H, self.W, self.focal = int(hwf[0]), int(hwf[1]), float(hwf[2])
        if no_ndc:                                                        # run.py:414-417
            self.near, self.far = float(np.ndarray.min(bds) * .9), float(np.ndarray.max(bds) * 1.)
        else:
            self.near, self.far = 0., 1.
        self.i_train = np.arange(images.shape[0]) if i_train is None else np.asarray(i_train)
        self.device = device
        masks = np.asarray(masks, np.float32)
        depths = np.asarray(inpainted_depths, np.float32)
        self.poses = torch.from_numpy(poses[:, :3, :4].copy()).to(device)
        self.images = torch.from_numpy(images).to(device)
        self.depths = torch.from_numpy(depths).to(device)
        self.masks = torch.from_numpy(masks == 1).to(device)              # run.py:876 `mask == 1`
        flat = self.masks.reshape(self.masks.shape[0], -1)
        self._masked_idx = [torch.nonzero(flat[v], as_tuple=False).reshape(-1) for v in range(flat.shape[0])]
        self.sets = None
        if build_sets:
            sets = build_ray_sets(images, poses, masks, depths, (self.H, self.W, self.focal), self.i_train, inp_pixels)
            self.sets = {k: torch.from_numpy(v).to(device) for k, v in sets.items() if k in ('rays_rgb_clf', 'rays_inp')}
            self._samplers = {k: _EpochSampler(v.shape[0], device, seed + 17 * n)
                              for n, (k, v) in enumerate(self.sets.items())}

    # -- what the loop body asks for ------------------------------------------------------------------------------
    def masked_idx_of(self, view):
        """int64 flat pixel indices (raster order) of view's inpainting mask (run.py:875-884)."""
        return self._masked_idx[view]

    def mask_of(self, view):
        return self.masks[view]

    def next_batch(self, which, n_rand, records=None):
        """One batch of the pre-baked fp16 records (run.py:887-908): returns (rays [2, B, 3] fp16, colour [B, 3],
        label [B]) -- the transposes and slices of the reference.  `records` overrides the draw (tests)."""
        if records is None:
            arr = self.sets[which]
            records = arr[self._samplers[which].next(n_rand)]
        batch = torch.transpose(records.to(self.device), 0, 1)
        rays, target = batch[:2], batch[2]
        return rays[:, :, :-1], target[:, :3], target[:, 3]

    # -- constructors ------------------------------------------------------------------------------------------------
    @classmethod
    def from_llff(cls, datadir, factor, device='cuda', **kw):
        from .load_llff import load_llff_data
        images, poses, bds, render_poses, i_test, masks, depths, mask_indices = load_llff_data(
            datadir, factor, recenter=True, bd_factor=.75, spherify=False)
        sc = cls(images, poses, bds, masks, depths, device=device, **kw)
        sc.render_poses, sc.i_test = render_poses, i_test
        return sc

    @classmethod
    def from_fixture(cls, path, size=None, device='cuda', views=None, **kw):
        """tests/golden/scene1_small.npz (every 2nd inpainted view of SPIn-NeRF scene 1 at 1/16 resolution: uint8
        images / depths, bool masks, real poses with the full-resolution hwf column).  `size=(H, W)` resamples the
        rasters (bilinear; masks nearest) and rescales the intrinsics, so the REAL poses, mask shapes and bounds can
        be exercised at the BASELINE configs' resolutions on a box that has no dataset."""
        z = np.load(path, allow_pickle=False)
        images = z['images'].astype(np.float32) / 255.
        depths = z['depths'].astype(np.float32) / 255.
        masks = z['masks'].astype(np.float32)
        poses = z['poses'].astype(np.float32).copy()
        if views is not None:
            images, depths, masks, poses = images[views], depths[views], masks[views], poses[views]
        h0, w0 = images.shape[1:3]
        H, W = (h0, w0) if size is None else size
        if (H, W) != (h0, w0):
            import torch.nn.functional as F
            t = lambda a, mode: F.interpolate(torch.from_numpy(a), size=(H, W), mode=mode,
                                              **({} if mode == 'nearest' else {'align_corners': False})).numpy()
            images = np.ascontiguousarray(t(images.transpose(0, 3, 1, 2), 'bilinear').transpose(0, 2, 3, 1))
            depths = t(depths[:, None], 'bilinear')[:, 0]
            masks = t(masks[:, None], 'nearest')[:, 0]
        full_h, full_w, full_f = poses[0, 0, 4], poses[0, 1, 4], poses[0, 2, 4]
        poses[:, 0, 4], poses[:, 1, 4] = H, W
        poses[:, 2, 4] = full_f * (W / full_w)                           # focal of the requested raster
        return cls(images, poses, z['bds'], masks, depths, device=device, **kw)
