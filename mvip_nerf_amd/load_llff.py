"""LLFF scene loader: mirror of the live part of `DS_NeRF/load_llff.py`.

`load_llff_data` returns the reference's 8-tuple
    (images, poses, bds, render_poses, i_test, masks, inpainted_depths, mask_indices)
from a SPIn-NeRF style directory: `poses_bounds.npy`, `images_<f>/RGB_inpainted/*.png`,
`images_<f>/label/*.png`, `images_<f>/Depth_inpainted/*.png`.  Host-side numpy (it is host-side
numpy in the reference too); images are read with PIL (the reference uses imageio).

Restated arithmetic, in order (DS_NeRF/load_llff.py:308-428): LLFF axis swap [-u, r, -t] -> [r, u, -t];
rescale translations and bounds by 1/(bds.min()*bd_factor); recenter on the average pose; the
`spherify_hack` branch only re-derives `bds` (its render poses are overwritten by the spiral path
that follows); 120-view spiral; holdout = view nearest the average pose; finally the reference's
hard-coded `poses = poses[40:]` (the 60 inpainted training views of SPIn-NeRF scenes).
`_minify` (image resizing through shell tools) is not restated: the resized folders must exist.
"""
import os

import numpy as np


def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """DS_NeRF/load_llff.py:188-194."""
    vec2 = normalize(z)
    vec0 = normalize(np.cross(up, vec2))
    vec1 = normalize(np.cross(vec2, vec0))
    return np.stack([vec0, vec1, vec2, pos], 1)


def poses_avg(poses):
    """DS_NeRF/load_llff.py:204-212."""
    hwf = poses[0, :3, -1:]
    center = poses[:, :3, 3].mean(0)
    vec2 = normalize(poses[:, :3, 2].sum(0))
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([viewmatrix(vec2, up, center), hwf], 1)


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """DS_NeRF/load_llff.py:215-225."""
    out = []
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    for theta in np.linspace(0., 2. * np.pi * rots, N + 1)[:-1]:
        c = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        z = normalize(c - np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.])))
        out.append(np.concatenate([viewmatrix(z, up, c), hwf], 1))
    return out


def recenter_poses(poses):
    """DS_NeRF/load_llff.py:228-240."""
    poses_ = poses + 0
    bottom = np.reshape([0, 0, 0, 1.], [1, 4])
    c2w = np.concatenate([poses_avg(poses)[:3, :4], bottom], -2)
    bottoms = np.tile(np.reshape(bottom, [1, 1, 4]), [poses.shape[0], 1, 1])
    p44 = np.concatenate([poses[:, :3, :4], bottoms], -2)
    p44 = np.linalg.inv(c2w) @ p44
    poses_[:, :3, :4] = p44[:, :3, :4]
    return poses_


def _spherify_bounds_scale(poses):
    """The only effect `spherify_hack` has on the returned values (load_llff.py:246-305, :345-349):
    bds are multiplied and divided by sc = 1/rad, rad = rms camera distance from the point closest
    to all optical axes, in the frame spherify_poses builds."""
    rays_d = poses[:, :3, 2:3]
    rays_o = poses[:, :3, 3:4]
    A_i = np.eye(3) - rays_d * np.transpose(rays_d, [0, 2, 1])
    b_i = -A_i @ rays_o
    center = np.squeeze(-np.linalg.inv((np.transpose(A_i, [0, 2, 1]) @ A_i).mean(0)) @ (b_i).mean(0))
    up = (poses[:, :3, 3] - center).mean(0)
    vec0 = normalize(up)
    vec1 = normalize(np.cross([.1, .2, .3], vec0))
    vec2 = normalize(np.cross(vec0, vec1))
    c2w = np.stack([vec1, vec2, vec0, center], 1)
    p44 = lambda p: np.concatenate([p, np.tile(np.reshape(np.eye(4)[-1, :], [1, 1, 4]), [p.shape[0], 1, 1])], 1)
    reset = np.linalg.inv(p44(c2w[None])) @ p44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    return 1. / rad


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(path))


def _load_data(basedir, factor=None, prepare=False):
    """DS_NeRF/load_llff.py:68-183 for the `factor` form."""
    poses_arr = np.load(os.path.join(basedir, 'poses_bounds.npy'))
    poses = poses_arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = poses_arr[:, -2:].transpose([1, 0])
    sfx = '' if factor is None else '_{}'.format(factor)
    factor = 1 if factor is None else factor
    root = os.path.join(basedir, 'images' + sfx)
    imgdir = root if prepare else os.path.join(root, 'RGB_inpainted')
    mskdir, depthdir = os.path.join(root, 'label'), os.path.join(root, 'Depth_inpainted')
    if not os.path.exists(imgdir):
        raise FileNotFoundError(imgdir)
    isimg = lambda f: f.endswith(('JPG', 'jpg', 'jpeg', 'png'))
    names = [f for f in sorted(os.listdir(imgdir)) if isimg(f)]
    imgfiles = [os.path.join(imgdir, f) for f in names]
    mskfiles = [os.path.join(mskdir, f.split('.')[0] + '.png') for f in names if 'cutout' not in f and 'pseudo' not in f]
    try:
        depthfiles = [os.path.join(depthdir, f.split('.')[0] + '.png') for f in sorted(os.listdir(depthdir)) if isimg(f)]
    except OSError:
        depthfiles = mskfiles
    sh = _imread(imgfiles[0]).shape
    poses[:2, 4, :] = np.array(sh[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    imgs = np.stack([_imread(f)[..., :3] / 255. for f in imgfiles], -1)
    masks, mask_indices = [], []
    for i, f in enumerate(mskfiles):
        try:
            m = _imread(f)
            m = m / m.max()
            if m.ndim > 2:
                m = m[:, :, 0]
            masks.append(m)
            mask_indices.append(i)
        except OSError:
            masks.append(-np.ones((imgs.shape[0], imgs.shape[1])))
    depths = []
    for f in depthfiles:
        try:
            d = _imread(f) / 255.
            if d.ndim > 2:
                d = d[:, :, 0]
            depths.append(d)
        except OSError:
            depths.append(-np.ones((imgs.shape[0], imgs.shape[1])))
    masks = np.stack(masks, -1)
    masks = masks / np.max(masks)
    return poses, bds, imgs, masks, np.stack(depths, -1), mask_indices


def process_poses(poses, bds, bd_factor=.75, recenter=True, spherify_hack=True, path_zflat=False):
    """Everything load_llff_data does to (poses [3,5,N], bds [2,N]) after the image files are read.
    Returns poses [N,3,5] float32 (ALL views), bds [N,2], render_poses [120,3,5], i_test."""
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)
    if spherify_hack:
        s2 = _spherify_bounds_scale(poses)
        bds *= s2            # spherify_poses scales bds in place ...
        bds = bds / s2       # ... and load_llff_data divides the returned array by sc again
    c2w = poses_avg(poses)
    up = normalize(poses[:, :3, 1].sum(0))
    close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
    dt = .75
    focal = 1. / (((1. - dt) / close_depth + dt / inf_depth))
    zdelta = close_depth * .2
    rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
    c2w_path, N_views, N_rots = c2w, 120, 2
    if path_zflat:
        c2w_path[:3, 3] = c2w_path[:3, 3] + (-close_depth * .1) * c2w_path[:3, 2]
        rads[2] = 0.
        N_rots, N_views = 1, N_views // 2
    render_poses = np.array(render_path_spiral(c2w_path, up, rads, focal, zdelta, zrate=.5, rots=N_rots,
                                               N=N_views)).astype(np.float32)
    c2w = poses_avg(poses)
    i_test = np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1))
    return poses.astype(np.float32), bds, render_poses, i_test


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False,
                   spherify_hack=True, prepare=False, refined=False, use_MVSeg=False, args=None):
    """DS_NeRF/load_llff.py:308-428."""
    if spherify:
        raise NotImplementedError('spherify=True (360-degree scenes) is not used by the shipped configs')
    poses, bds, imgs, masks, depths, mask_indices = _load_data(basedir, factor=factor, prepare=prepare)
    poses, bds, render_poses, i_test = process_poses(poses, bds, bd_factor, recenter, spherify_hack, path_zflat)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    masks = np.moveaxis(masks, -1, 0).squeeze().astype(np.float32)
    depths = np.moveaxis(depths, -1, 0).squeeze().astype(np.float32)
    if masks.shape[-1] == 3:
        masks = masks[:, :, :, 0].squeeze()
    if depths.shape[-1] == 3:
        depths = depths[:, :, :, 0].squeeze()
    poses = poses[40:, :, :]         # the reference's hard-coded selection of the 60 inpainted views
    return images, poses, bds, render_poses, i_test, masks, depths, mask_indices
