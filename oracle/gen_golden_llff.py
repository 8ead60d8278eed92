"""Golden vectors for the LLFF loader: runs the REAL `DS_NeRF/load_llff.py::load_llff_data` on
/root/reference/data/1 (factor 4) in the build container.  imageio/cv2 are not installed, so the
module's `imageio` is replaced by a PIL shim and the (missing) full-resolution `images/` folder is
answered with the frame size recorded in poses_bounds.npy.  Also writes a SMALL real-data fixture
(tests/golden/scene1_small.npz: every 2nd training view at 1/16 resolution + masks + depths) for the
end-to-end GPU demo.  Data only; no reference source is copied."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')
DATA = '/root/reference/data/1'


def main():
    from oracle.gen_golden import import_reference
    import_reference()
    import load_llff as L
    from PIL import Image

    class Shim:
        @staticmethod
        def imread(path, *a, **k):
            if os.sep + 'images' + os.sep in path and not os.path.exists(path):
                return np.zeros((2268, 4032, 3), np.uint8)
            return np.asarray(Image.open(path))
    L.imageio = Shim
    real_listdir = os.listdir

    def listdir(p):
        if p.rstrip('/').endswith(os.path.join('1', 'images')) and not os.path.exists(p):
            return ['00000.jpg']
        return real_listdir(p)
    L.os.listdir = listdir
    try:
        images, poses, bds, render_poses, i_test, masks, depths, mask_indices = L.load_llff_data(
            DATA, 4, recenter=True, bd_factor=.75, spherify=False, prepare=False, args=None)
    finally:
        L.os.listdir = real_listdir
    print(images.shape, poses.shape, bds.shape, render_poses.shape, i_test, masks.shape, depths.shape)
    sel = [(0, 10, 20), (30, 500, 700), (59, 566, 1007)]
    np.savez_compressed(os.path.join(OUT, 'llff_scene1_f4.npz'), poses=poses, bds=bds, render_poses=render_poses,
                        i_test=i_test, images_shape=np.array(images.shape), image_mean=images.mean((1, 2, 3)),
                        pixels=np.stack([images[a, b, c] for a, b, c in sel]), pixel_idx=np.array(sel),
                        mask_sum=masks.sum((1, 2)), depth_mean=depths.mean((1, 2)),
                        mask_indices=np.array(mask_indices))
    # small real-data fixture: views 0,2,4,... (30 views), box-filtered 4x further (1/16 of the original)
    views = list(range(0, 60, 2))
    def down(a):
        h, w = a.shape[0] // 4 * 4, a.shape[1] // 4 * 4
        a = a[:h, :w]
        return a.reshape(h // 4, 4, w // 4, 4, *a.shape[2:]).mean((1, 3))
    img_s = np.stack([down(images[v]) for v in views])
    msk_s = np.stack([(down(masks[v]) > 0.5) for v in views])
    dep_s = np.stack([down(depths[v]) for v in views])
    np.savez_compressed(os.path.join(OUT, 'scene1_small.npz'), images=(img_s * 255 + .5).astype(np.uint8),
                        masks=msk_s, depths=(dep_s * 255 + .5).astype(np.uint8), poses=poses[views],
                        bds=bds, views=np.array(views), factor=16)
    # factor-8 fixture (283 x 504: BASELINE configs[0]'s geometry, the size the PSNR clause names): 15 training views (every
    # 4th) + view 30 held out, the factor-4 rasters box-filtered 2x (567 -> 283 rows: the last row dropped)
    views8 = sorted(set(range(0, 60, 4)) | {30})
    def down2(a):
        h, w = a.shape[0] // 2 * 2, a.shape[1] // 2 * 2
        a = a[:h, :w]
        return a.reshape(h // 2, 2, w // 2, 2, *a.shape[2:]).mean((1, 3))
    img8 = np.stack([down2(images[v]) for v in views8])
    msk8 = np.stack([(down2(masks[v]) > 0.5) for v in views8])
    np.savez_compressed(os.path.join(OUT, 'scene1_f8.npz'), images=(img8 * 255 + .5).astype(np.uint8), masks=msk8,
                        poses=poses[views8], bds=bds, views=np.array(views8), factor=8, held_out_view=30)
    for f in ('llff_scene1_f4.npz', 'scene1_small.npz', 'scene1_f8.npz'):
        print(f, os.path.getsize(os.path.join(OUT, f)) / 1024, 'KB')


if __name__ == '__main__':
    main()
