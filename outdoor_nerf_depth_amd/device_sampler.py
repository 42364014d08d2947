"""GPU-resident frames and on-device ray-batch sampling.

The reference draws every batch on the host: `np.random.choice(H*W, N_rand, replace=False)` (a full
permutation of 465 750 pixels), numpy fancy indexing of five arrays and five H2D copies per step
(nerf_sample_ray_split.py:155-221, ddp_train_nerf.py:423-427).  At MI355X step times that is the
bottleneck, so here the frames (rgb, depth prior, cameras) live in HBM and a batch is two small kernels:
`nerfpp_sample_pixels` (N_rand distinct pixels, the distribution of np.random.choice(replace=False), drawn from the
counter-based generator without permuting the whole frame -- `torch.randperm(H * W)` cost 0.3 ms per step) and
`nerfpp_gather_rays`, which regenerates the rays from K^-1 / c2w (same formula as get_rays_single_image) and gathers
rgb / depth_sup.

Keys: the ones the training step reads -- ray_o, ray_d, rgb, min_depth, depth_sup (+ frame).  The
reference's sampler dict also carries depth_gt, mask and depth (nerf_sample_ray_split.py:199-221), which
the training loop never reads; `random_sample(..., full_keys=True)` adds depth_gt and mask from their own
device-resident maps (torch indexing, not the kernel).  `depth` (a dummy ones-vector upstream) only exists
with --host_sampling.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


class DeviceRaySamplers(object):
    """All training frames of one split on the device."""

    def __init__(self, ray_samplers, device, seed=777):
        """seed: key of the pixel draw (the CLI passes (rank + 1) * 777 like ddp_train_nerf.py:406); every random_sample()
        advances its own step counter."""
        self.device = torch.device(device)
        self.seed, self.draws = int(seed), 0
        s0 = ray_samplers[0]
        self.H, self.W = s0.H, s0.W
        self.n_frames = len(ray_samplers)
        cams = np.zeros((self.n_frames, 21), np.float32)
        for f, s in enumerate(ray_samplers):
            cams[f, :9] = np.linalg.inv(s.intrinsics[:3, :3]).astype(np.float32).reshape(-1)
            cams[f, 9:] = np.asarray(s.c2w_mat, np.float32)[:3, :4].reshape(-1)
        self.cams = torch.from_numpy(cams).to(self.device)
        self.rgb = None
        if s0.img is not None:
            self.rgb = torch.stack([torch.from_numpy(np.ascontiguousarray(s.img, np.float32)) for s in ray_samplers]
                                   ).to(self.device)                                    # [F, H*W, 3]
        self.depth_sup = None
        if s0.depth_sup is not None:
            self.depth_sup = torch.stack([torch.from_numpy(np.ascontiguousarray(s.depth_sup, np.float32))
                                          for s in ray_samplers]).to(self.device)      # [F, H*W]
        # ground-truth depth and mask maps for full_keys (only uploaded when they are separate data)
        self.depth_gt = None
        if s0.depth_gt is not None:
            if all(s.depth_gt is s.depth_sup or (s.depth_sup is not None and np.array_equal(s.depth_gt, s.depth_sup))
                   for s in ray_samplers):
                self.depth_gt = self.depth_sup                                          # depth_sup_type == 'gt'
            else:
                self.depth_gt = torch.stack([torch.from_numpy(np.ascontiguousarray(s.depth_gt, np.float32))
                                             for s in ray_samplers]).to(self.device)
        self.mask = None
        if getattr(s0, 'mask', None) is not None:
            self.mask = torch.stack([torch.from_numpy(np.ascontiguousarray(s.mask, np.float32))
                                     for s in ray_samplers]).to(self.device)
        self.depth_scale = s0.get_depth_scale()
        # optional per-pixel near bound (min_depth/ pngs): gathered with torch indexing after the kernel
        self.min_depth = None
        if getattr(s0, 'min_depth', None) is not None:
            self.min_depth = torch.stack([torch.from_numpy(np.ascontiguousarray(s.min_depth, np.float32))
                                          for s in ray_samplers]).to(self.device)      # [F, H*W]

    def gather(self, frame, pix, full_keys=False):
        """Ray batch of `frame` at the flat pixel indices `pix` (int64 device tensor)."""
        n = pix.numel()
        dev = self.device
        out = dict(ray_o=torch.empty(n, 3, device=dev), ray_d=torch.empty(n, 3, device=dev),
                   min_depth=torch.empty(n, device=dev))
        rgb_img = self.rgb[frame] if self.rgb is not None else None
        dep_img = self.depth_sup[frame] if self.depth_sup is not None else None
        if rgb_img is not None:
            out['rgb'] = torch.empty(n, 3, device=dev)
        if dep_img is not None:
            out['depth_sup'] = torch.empty(n, device=dev)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        L.check(L.lib().nerfpp_gather_rays(C.c_void_p(torch.cuda.current_stream().cuda_stream), n, self.W,
                                           p(self.cams[frame]), p(pix), p(rgb_img), p(dep_img), p(out['ray_o']),
                                           p(out['ray_d']), p(out.get('rgb')), p(out.get('depth_sup')),
                                           p(out['min_depth'])), 'nerfpp_gather_rays')
        if self.min_depth is not None:
            out['min_depth'] = self.min_depth[frame][pix]
        if full_keys:
            if self.depth_gt is not None:
                out['depth_gt'] = out['depth_sup'] if self.depth_gt is self.depth_sup else self.depth_gt[frame][pix]
            out['mask'] = None if self.mask is None else self.mask[frame][pix]
        return out

    def random_sample(self, N_rand, frame=None, full_keys=False):
        """One random frame (host RNG, like ddp_train_nerf.py:423), N_rand distinct pixels (nerfpp_sample_pixels)."""
        from . import ops
        if frame is None:
            frame = int(np.random.randint(low=0, high=self.n_frames))
        self.draws += 1
        n_pix = self.H * self.W
        if N_rand > n_pix:
            raise ValueError('N_rand = %d exceeds the %d pixels of a frame (np.random.choice(..., replace=False) raises too, '
                             'nerf_sample_ray_split.py:178)' % (N_rand, n_pix))
        if N_rand <= 8192 and 2 * N_rand <= n_pix:
            pix = ops.sample_pixels(n_pix, N_rand, self.seed, self.draws, self.device)
        else:
            # nerfpp_sample_pixels draws up to 8192 pixels in one workgroup and redraws collisions: beyond that size, or when
            # the batch is a large share of the frame (many redraw rounds), a seeded device permutation is the right tool
            g = torch.Generator(device=self.device)
            g.manual_seed((self.seed << 20) + self.draws)
            pix = torch.randperm(n_pix, device=self.device, generator=g)[:N_rand]
        out = self.gather(frame, pix, full_keys)
        out['frame'] = frame
        return out
