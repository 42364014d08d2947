"""Synthetic KITTI-shaped scenes.

There is no network for the real KITTI / Argoverse sequences, so benchmarks and tests use a
generator that reproduces the *shape* of the reference's input: 375x1242 pinhole frames on a
gently curving forward trajectory, camera poses normalised exactly like
colmap_runner/normalize_cam_dict.py:8-30 (mean camera centre to the origin, scale =
1/(1.5*max|c|) so every ray origin has |o| <= 2/3), un-normalised ray directions
d = R K^-1 [u+.5, v+.5, 1]^T (nerf_sample_ray_split.py:18-29), and a depth prior in metres *
scale with 0 = invalid (`gt`: ~5 % valid pixels; `*_crop`: dense below a sky crop).
"""
import numpy as np

KITTI_H, KITTI_W = 375, 1242
KITTI_FX = KITTI_FY = 721.5377
KITTI_CX, KITTI_CY = 609.5593, 172.854
DATA_SEED = 20230804


def kitti_intrinsics(H=KITTI_H, W=KITTI_W):
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1] = KITTI_FX * W / KITTI_W, KITTI_FY * H / KITTI_H
    K[0, 2], K[1, 2] = KITTI_CX * W / KITTI_W, KITTI_CY * H / KITTI_H
    return K


def trajectory(n_frames=295, length_m=250.0):
    """OpenCV-convention (x right, y down, z forward) camera-to-world matrices on a gently
    curving road, in metres, before normalisation."""
    t = np.linspace(0.0, 1.0, n_frames)
    yaw = 0.35 * np.sin(1.7 * np.pi * t)                    # heading, rad
    dz = np.cos(yaw)
    dx = np.sin(yaw)
    ds = length_m / max(n_frames - 1, 1)
    x = np.concatenate([[0.0], np.cumsum(dx[:-1] * ds)])
    z = np.concatenate([[0.0], np.cumsum(dz[:-1] * ds)])
    y = 0.15 * np.sin(9.0 * np.pi * t)                       # suspension bounce
    c2w = np.tile(np.eye(4, dtype=np.float64), (n_frames, 1, 1))
    c2w[:, 0, 0], c2w[:, 0, 2] = np.cos(yaw), np.sin(yaw)
    c2w[:, 2, 0], c2w[:, 2, 2] = -np.sin(yaw), np.cos(yaw)
    c2w[:, 0, 3], c2w[:, 1, 3], c2w[:, 2, 3] = x, y, z
    return c2w


def normalize_poses(c2w, target_radius=1.0):
    """colmap_runner/normalize_cam_dict.py:8-30: centre on the mean camera position and scale
    with scale = target_radius / (1.5 * max camera distance).

    A single camera (BASELINE config 1: one frame) has max distance 0, for which the reference's
    `target_radius / radius` (:23-27) is a division by zero.  That case is defined here as the same
    formula applied to a unit distance (1 m): the camera sits at the origin and
    scale = target_radius / 1.5 (finite; metres * scale stays the depth-prior convention)."""
    centres = c2w[:, :3, 3]
    centre = centres.mean(0)
    dist = np.linalg.norm(centres - centre, axis=-1).max()
    if not dist > 1e-9:
        dist = 1.0
    scale = target_radius / (dist * 1.5)
    out = c2w.copy()
    out[:, :3, 3] = (centres - centre) * scale
    return out.astype(np.float32), np.float32(scale)


def get_rays(H, W, K, c2w, pix=None):
    """Same arithmetic as nerf_sample_ray_split.py:get_rays_single_image, optionally restricted
    to the flat pixel indices `pix`."""
    if pix is None:
        pix = np.arange(H * W)
    u = (pix % W).astype(np.float32) + 0.5
    v = (pix // W).astype(np.float32) + 0.5
    pixels = np.stack((u, v, np.ones_like(u)), axis=0)
    rays_d = np.dot(np.linalg.inv(K[:3, :3]), pixels)
    rays_d = np.dot(c2w[:3, :3], rays_d).transpose((1, 0))
    rays_o = np.tile(c2w[:3, 3].reshape((1, 3)), (rays_d.shape[0], 1))
    return rays_o.astype(np.float32), rays_d.astype(np.float32)


class SyntheticKitti(object):
    """A KITTI-seq00-shaped scene held as per-frame cameras; pixels are generated on demand.

    depth_sup_type 'gt' -> sparse (5 % valid) prior, anything ending in '_crop' -> dense prior
    with the top 35 % of rows invalid (sky crop).  RGB is a smooth procedural function of the
    ray (so PSNR moves when a model trains) plus mild per-pixel noise.
    """

    def __init__(self, n_frames=295, H=KITTI_H, W=KITTI_W, depth_sup_type='gt', trainskip=1,
                 seed=DATA_SEED):
        self.H, self.W = H, W
        self.K = kitti_intrinsics(H, W)
        if n_frames < 2:
            # BASELINE config 1 ("1 KITTI-seq00 frame"): ONE frame of a sequence that was normalised as a whole.  A lone
            # camera normalised by itself sits exactly at the origin, where the reference's inverted-sphere
            # parametrisation is 0/0 (ddp_model.py:27-28: rot_axis = cross(ray_o, p_sphere) / |.|) -- NaN in the
            # reference itself -- so the frame is taken from the normalised 295-frame trajectory instead.
            c2w, self.depth_scale = normalize_poses(trajectory(295))
            c2w = c2w[:n_frames]
        else:
            c2w, self.depth_scale = normalize_poses(trajectory(n_frames))
        idx = np.arange(n_frames)
        is_test = (idx % 10) == 9                                # colmap2nerfpp.py:111
        self.train_c2w = c2w[~is_test][::trainskip]
        self.test_c2w = c2w[is_test]
        self.depth_sup_type = depth_sup_type
        self.seed = seed

    def n_train(self):
        return len(self.train_c2w)

    def _pixel_hash(self, frame, pix):
        x = (pix.astype(np.uint64) * np.uint64(2654435761) +
             np.uint64(frame + 1) * np.uint64(40503) + np.uint64(self.seed))
        x ^= x >> np.uint64(13)
        x *= np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(29)
        return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)     # U[0,1)

    def batch(self, frame, pix, split='train'):
        """Ray batch dict with the reference's keys (nerf_sample_ray_split.py:203-221)."""
        c2w = (self.train_c2w if split == 'train' else self.test_c2w)[frame]
        ray_o, ray_d = get_rays(self.H, self.W, self.K, c2w, pix)
        vd = ray_d / np.linalg.norm(ray_d, axis=-1, keepdims=True)
        h = self._pixel_hash(frame, pix)
        rgb = 0.5 + 0.35 * np.stack([np.sin(3.0 * vd[:, 0] + 1.0 + 2.0 * ray_o[:, 2]),
                                     np.sin(4.0 * vd[:, 1] + 2.0),
                                     np.cos(5.0 * vd[:, 0] * vd[:, 1] + ray_o[:, 0])], -1)
        rgb = np.clip(rgb + 0.05 * (h[:, None] - 0.5), 0.0, 1.0).astype(np.float32)
        metres = 2.0 + 78.0 * self._pixel_hash(frame + 7919, pix)            # U(2,80) m
        row = pix // self.W
        if self.depth_sup_type == 'gt':
            valid = self._pixel_hash(frame + 104729, pix) < 0.05
        else:
            valid = row >= int(0.35 * self.H)
        depth_sup = np.where(valid, metres * float(self.depth_scale), 0.0).astype(np.float32)
        return dict(ray_o=ray_o, ray_d=ray_d, rgb=rgb, depth_sup=depth_sup, depth_gt=depth_sup,
                    min_depth=np.full(len(pix), 1e-4, np.float32))

    def random_batch(self, n_rand, rng):
        """One random training frame, n_rand pixels without replacement
        (ddp_train_nerf.py:423-424, nerf_sample_ray_split.py:178)."""
        frame = int(rng.randint(0, self.n_train()))
        pix = rng.choice(self.H * self.W, size=(n_rand,), replace=False)
        out = self.batch(frame, pix)
        out['frame'] = frame
        return out
