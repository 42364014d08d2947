"""Dataset side of the path: the NeRF++ on-disk layout

    {datadir}/{scene}/{train,test}/{rgb/*.png, pose/*.txt, intrinsics/*.txt, depth/*.png,
                                    depth_<sup_type>/*.png}   +   {datadir}/{scene}/scale

read like nerf-methods/nerfplusplus/data_loader_split.py:27-129 and
nerf_sample_ray_split.py:10-221 (same ray formula, same dict keys, depth = uint16/256 * scale),
with PIL for image I/O (cv2 / imageio are not in this image).  Differences, on purpose: the
reference's `depth_files` typo (data_loader_split.py:89) that crashes without a depth/ directory is
not reproduced; images are not resized (resolution_level is always 1 in the reference's loop).
"""
import glob
import os
from collections import OrderedDict

import numpy as np


def find_files(d, exts):
    if not os.path.isdir(d):
        return []
    out = []
    for e in exts:
        out.extend(glob.glob(os.path.join(d, e)))
    return sorted(out)


def _imread(path):
    from PIL import Image
    return np.array(Image.open(path))


def _imread_resized(path, W, H, nearest):
    from PIL import Image
    im = Image.open(path)
    if im.size != (W, H):
        im = im.resize((W, H), Image.NEAREST if nearest else Image.BILINEAR)
    return np.array(im)


def get_rays_single_image(H, W, intrinsics, c2w):
    """nerf_sample_ray_split.py:10-34 (half-pixel centres, un-normalised directions)."""
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    u = u.reshape(-1).astype(dtype=np.float32) + 0.5
    v = v.reshape(-1).astype(dtype=np.float32) + 0.5
    pixels = np.stack((u, v, np.ones_like(u)), axis=0)
    rays_d = np.dot(np.linalg.inv(intrinsics[:3, :3]), pixels)
    rays_d = np.dot(c2w[:3, :3], rays_d).transpose((1, 0))
    rays_o = np.tile(c2w[:3, 3].reshape((1, 3)), (rays_d.shape[0], 1))
    depth = np.linalg.inv(c2w)[2, 3] * np.ones((rays_o.shape[0],), dtype=np.float32)
    return rays_o, rays_d, depth


class RaySamplerSingleImage(object):
    """One frame: pixels -> rays, random_sample(N_rand) / get_all() dicts of numpy arrays with the
    reference's keys (nerf_sample_ray_split.py:131-221)."""

    def __init__(self, H, W, intrinsics, c2w, img_path=None, depth_gt_path=None, depth_sup_path=None,
                 depth_scale=None, img=None, depth_gt=None, depth_sup=None, mask_path=None, min_depth_path=None,
                 max_depth=None):
        self.H, self.W = H, W
        self.intrinsics, self.c2w_mat = intrinsics, c2w
        self.img_path, self.depth_gt_path, self.depth_sup_path = img_path, depth_gt_path, depth_sup_path
        self.depth_scale = depth_scale
        self.resolution_level = 1
        self.img = img
        self.depth_gt = depth_gt
        self.depth_sup = depth_sup
        if img_path is not None:
            self.img = (_imread(img_path).astype(np.float32) / 255.)[..., :3].reshape((-1, 3))
        if depth_gt_path is not None:
            self.depth_gt = depth_scale * (_imread(depth_gt_path).astype(np.float32) / 256.0).reshape((-1))
        if depth_sup_path is not None:
            self.depth_sup = depth_scale * (_imread(depth_sup_path).astype(np.float32) / 256.0).reshape((-1))
        # optional per-pixel mask / near bound (nerf_sample_ray_split.py:81-92): mask = png / 255 (nearest),
        # min_depth = png / 255 * max_depth + 1e-4 (bilinear; PIL here, cv2 upstream -- only differs when the
        # file is not already H x W)
        self.mask = self.min_depth = None
        if mask_path is not None:
            self.mask = _imread_resized(mask_path, W, H, nearest=True).astype(np.float32).reshape((-1)) / 255.
        if min_depth_path is not None:
            self.min_depth = (_imread_resized(min_depth_path, W, H, nearest=False).astype(np.float32) / 255. *
                              max_depth + 1e-4).reshape((-1)).astype(np.float32)
        self.rays_o, self.rays_d, self.depth = get_rays_single_image(H, W, intrinsics, c2w)

    def get_img(self):
        return None if self.img is None else self.img.reshape((self.H, self.W, 3))

    def get_gt_depth_img(self):
        return None if self.depth_gt is None else self.depth_gt.reshape((self.H, self.W))

    def get_depth_scale(self):
        return self.depth_scale if self.depth_gt is not None else None

    def _select(self, idx):
        ret = OrderedDict([('ray_o', self.rays_o[idx]), ('ray_d', self.rays_d[idx]), ('depth', self.depth[idx]),
                           ('rgb', None if self.img is None else self.img[idx]),
                           ('mask', None if self.mask is None else self.mask[idx]),
                           ('min_depth', self.min_depth[idx] if self.min_depth is not None
                            else 1e-4 * np.ones_like(self.rays_d[idx][..., 0]))])
        if self.depth_gt is not None:
            ret['depth_gt'] = self.depth_gt[idx]
        if self.depth_sup is not None:
            ret['depth_sup'] = self.depth_sup[idx]
        return ret

    def get_all(self):
        return self._select(slice(None))

    def random_sample(self, N_rand, center_crop=False):
        select_inds = np.random.choice(self.H * self.W, size=(N_rand,), replace=False)   # :178
        ret = self._select(select_inds)
        ret['img_name'] = self.img_path
        return ret


def load_data_split(basedir, scene, split, skip=1, try_load_min_depth=True, only_img_files=False,
                    depth_sup_type='gt'):
    """data_loader_split.py:27-129."""
    def parse_txt(filename):
        nums = open(filename).read().split()
        return np.array([float(x) for x in nums]).reshape([4, 4]).astype(np.float32)

    basedir = basedir.rstrip('/')
    split_dir = '{}/{}/{}'.format(basedir, scene, split)
    if only_img_files:
        return find_files('{}/rgb'.format(split_dir), exts=['*.png', '*.jpg'])
    intrinsics_files = find_files('{}/intrinsics'.format(split_dir), exts=['*.txt'])[::skip]
    pose_files = find_files('{}/pose'.format(split_dir), exts=['*.txt'])[::skip]
    cam_cnt = len(pose_files)
    img_files = find_files('{}/rgb'.format(split_dir), exts=['*.png', '*.jpg'])
    img_files = img_files[::skip] if img_files else [None] * cam_cnt
    assert len(img_files) == cam_cnt
    depth_gt_files = find_files('{}/depth'.format(split_dir), exts=['*.png', '*.jpg'])
    depth_scale = None
    if depth_gt_files:
        depth_gt_files = depth_gt_files[::skip]
        assert len(depth_gt_files) == cam_cnt
        depth_scale = float(open(os.path.join(basedir, scene, 'scale')).readlines()[0].strip())
    else:
        depth_gt_files = [None] * cam_cnt
    suffix = '_' + depth_sup_type if depth_sup_type != 'gt' else ''
    depth_sup_files = find_files('{}/depth{}'.format(split_dir, suffix), exts=['*.png', '*.jpg'])
    depth_sup_files = depth_sup_files[::skip] if depth_sup_files else [None] * cam_cnt
    assert len(depth_sup_files) == cam_cnt
    mask_files = find_files('{}/mask'.format(split_dir), exts=['*.png', '*.jpg'])
    mask_files = mask_files[::skip] if mask_files else [None] * cam_cnt
    assert len(mask_files) == cam_cnt
    mindepth_files = find_files('{}/min_depth'.format(split_dir), exts=['*.png', '*.jpg'])
    mindepth_files = mindepth_files[::skip] if (try_load_min_depth and mindepth_files) else [None] * cam_cnt
    assert len(mindepth_files) == cam_cnt
    try:                                                                      # data_loader_split.py:111-114
        max_depth = float(open('{}/max_depth.txt'.format(split_dir)).readline().strip())
    except Exception:
        max_depth = None
    train_imgfile = find_files('{}/{}/train/rgb'.format(basedir, scene), exts=['*.png', '*.jpg'])[0]
    H, W = _imread(train_imgfile).shape[:2]
    return [RaySamplerSingleImage(H=H, W=W, intrinsics=parse_txt(intrinsics_files[i]), c2w=parse_txt(pose_files[i]),
                                  img_path=img_files[i], depth_gt_path=depth_gt_files[i],
                                  depth_sup_path=depth_sup_files[i] if depth_scale is not None else None,
                                  depth_scale=depth_scale, mask_path=mask_files[i], min_depth_path=mindepth_files[i],
                                  max_depth=max_depth) for i in range(cam_cnt)]


def synthetic_ray_samplers(split, skip=1, depth_sup_type='gt', n_frames=295, H=None, W=None):
    """The same RaySamplerSingleImage objects fed from the synthetic KITTI-shaped scene (no disk)."""
    from .synthetic import SyntheticKitti, KITTI_H, KITTI_W
    H, W = H or KITTI_H, W or KITTI_W
    scene = SyntheticKitti(n_frames=n_frames, H=H, W=W, depth_sup_type=depth_sup_type, trainskip=1)
    poses = scene.train_c2w if split == 'train' else scene.test_c2w
    def one(f):
        b = scene.batch(f, np.arange(H * W), split=split)
        return RaySamplerSingleImage(H, W, scene.K, poses[f], depth_scale=float(scene.depth_scale), img=b['rgb'],
                                     depth_gt=b['depth_gt'], depth_sup=b['depth_sup'])
    frames = list(range(0, len(poses), skip))
    if len(frames) * H * W < (1 << 22):
        return [one(f) for f in frames]
    # 266 frames of 375 x 1242 are ~2 minutes of numpy on one core: frames are independent (numpy releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(32, os.cpu_count() or 1))) as ex:
        return list(ex.map(one, frames))
