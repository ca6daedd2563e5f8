"""Mirror of the training-path methods of GAN2Shape/renderer/renderer.py:13-139,252-264.

Same constructor and method names / argument meaning as the reference `Renderer`; the rasterizer
behind `warp_canon_depth` is the libg2s kernel (through the neural_renderer drop-in).  Differences
that do not change results: tensors are created on `device` directly (the reference hard-codes
.cuda()); the pixel grid, face list and rotation centre are built once; `grid_sample` gets
`align_corners=True` explicitly (the reference targets torch 1.2 where that was the only
behaviour).  The visualisation helpers — render_yaw, render_view, the
grid_sample=False branch of render_given_view (renderer.py:141-277), downscale_K — go through the
texture path of the rasterizer (`render_rgb`); their pose sweeps share one helper (`_sweep`).
"""
import math

import torch
import torch.nn as nn

from .. import fused_geometry as fg
from ..plugins import neural_renderer as nr
from .utils import get_face_idx, get_grid, get_textures_from_im, get_transform_matrices

EPS = 1e-7


class Renderer():
    def __init__(self, cfgs, image_size, min_depth, max_depth, device="cuda"):
        self.device = torch.device(device)
        self.image_size = image_size
        self.min_depth = min_depth
        self.max_depth = max_depth
        self.rot_center_depth = cfgs.get('rot_center_depth', (self.min_depth + self.max_depth) / 2)
        self.fov = cfgs.get('fov', 10)
        self.tex_cube_size = cfgs.get('tex_cube_size', 2)
        self.renderer_min_depth = cfgs.get('renderer_min_depth', 0.1)
        self.renderer_max_depth = cfgs.get('renderer_max_depth', 10.)

        # pinhole intrinsics (renderer.py:35-46):  d * K^-1 (u, v, 1)^T = (x, y, z)^T
        R = torch.eye(3, device=self.device).unsqueeze(0)
        t = torch.zeros(1, 3, dtype=torch.float32, device=self.device)
        fx = (self.image_size - 1) / 2 / (math.tan(self.fov / 2 * math.pi / 180))
        fy = (self.image_size - 1) / 2 / (math.tan(self.fov / 2 * math.pi / 180))
        cx = (self.image_size - 1) / 2
        cy = (self.image_size - 1) / 2
        K = torch.tensor([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]], dtype=torch.float32,
                         device=self.device)
        self.inv_K_origin = torch.inverse(K).unsqueeze(0)
        self.K_origin = K.unsqueeze(0)
        self.inv_K = self.inv_K_origin.clone()
        self.K = self.K_origin.clone()
        self.renderer = nr.Renderer(camera_mode='projection',
                                    light_intensity_ambient=1.0,
                                    light_intensity_directional=0.,
                                    K=self.K, R=R, t=t,
                                    near=self.renderer_min_depth, far=self.renderer_max_depth,
                                    image_size=self.image_size, orig_size=self.image_size,
                                    fill_back=True,
                                    background_color=[1, 1, 1])
        self._centroid = torch.tensor([0., 0., self.rot_center_depth], device=self.device).view(1, 1, 3)
        self._rays = {}
        self._K9 = (fx, 0., cx, 0., fy, cy, 0., 0., 1.)  # host copy of K for the fused kernels
        # fused = True: on CUDA tensors the geometry chains below run as single libg2s kernels
        # (csrc/geometry.hip); the torch code in each method is the specification they are tested
        # against and the path for CPU tensors.
        self.fused = True

    def _use_fused(self, t):
        return self.fused and t.is_cuda and t.dtype == torch.float32

    def _const(self, key, make):
        """Device constants are built once: a host->device copy inside the hot loop would also make
        the iteration impossible to record into a HIP graph."""
        v = self._rays.get(key)
        if v is None:
            v = self._rays[key] = make()
        return v

    def set_transform_matrices(self, view):
        if self._use_fused(view) and view.size(1) == 6:
            self.rot_mat, self.trans_xyz = fg.view_transform(view)
        else:
            self.rot_mat, self.trans_xyz = get_transform_matrices(view)

    def set_view(self, view, rot_scale, txy_scale, tz_scale):
        """set_transform_matrices(get_view_transformation(view)) of GAN2Shape/model.py:119-120,330-335
        in one step: angles = view[:, :3] * rot_scale, t = (view[:, 3:5] * txy, view[:, 5] * tz)."""
        if self._use_fused(view):
            self.rot_mat, self.trans_xyz = fg.view_transform(view, rot_scale, txy_scale, tz_scale)
        else:
            self.set_transform_matrices(torch.cat([view[:, :3] * rot_scale, view[:, 3:5] * txy_scale,
                                                   view[:, 5:] * tz_scale], 1))

    def rotate_pts(self, pts, rot_mat):
        centroid = self._centroid.to(pts.device)
        pts = pts - centroid          # move to centroid
        pts = pts.matmul(rot_mat.transpose(2, 1))  # rotate
        return pts + centroid         # move back

    def translate_pts(self, pts, trans_xyz):
        return pts + trans_xyz

    def _pixel_rays(self, h, w, device):
        """K^-1 (u, v, 1)^T for every pixel, (1, h, w, 3): constant, built once."""
        key = (h, w, str(device))
        r = self._rays.get(key)
        if r is None:
            grid_2d = get_grid(1, h, w, normalize=False, device=device)
            grid_3d = torch.cat((grid_2d, torch.ones(1, h, w, 1, device=device)), dim=3)
            r = grid_3d.matmul(self.inv_K.to(device).transpose(2, 1))
            self._rays[key] = r
        return r

    def depth_to_3d_grid(self, depth):
        b, h, w = depth.shape
        return self._pixel_rays(h, w, depth.device) * depth.unsqueeze(-1)

    def grid_3d_to_2d(self, grid_3d):
        b, h, w, _ = grid_3d.shape
        grid_2d = grid_3d / grid_3d[..., 2:]
        grid_2d = grid_2d.matmul(self.K.to(grid_3d.device).transpose(2, 1))[:, :, :, :2]
        WH = self._const(("wh", h, w, str(grid_3d.device)), lambda: torch.tensor(
            [w - 1, h - 1], dtype=torch.float32, device=grid_3d.device).view(1, 1, 1, 2))
        return grid_2d / WH * 2. - 1.  # normalize to -1~1

    def get_warped_3d_grid(self, depth):
        b, h, w = depth.shape
        if self._use_fused(depth):
            rays = self._pixel_rays(h, w, depth.device).reshape(-1, 3)
            return fg.warp_verts(depth, rays, self.rot_mat, self.trans_xyz,
                                 self.rot_center_depth).reshape(b, h, w, 3)
        grid_3d = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
        grid_3d = self.rotate_pts(grid_3d, self.rot_mat)
        grid_3d = self.translate_pts(grid_3d, self.trans_xyz)
        return grid_3d.reshape(b, h, w, 3)

    def get_inv_warped_3d_grid(self, depth):
        b, h, w = depth.shape
        grid_3d = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
        grid_3d = self.translate_pts(grid_3d, -self.trans_xyz)
        grid_3d = self.rotate_pts(grid_3d, self.rot_mat.transpose(2, 1))
        return grid_3d.reshape(b, h, w, 3)

    def get_warped_2d_grid(self, depth):
        return self.grid_3d_to_2d(self.get_warped_3d_grid(depth))

    def get_inv_warped_2d_grid(self, depth):
        if self._use_fused(depth):
            b, h, w = depth.shape
            rays = self._pixel_rays(h, w, depth.device).reshape(-1, 3)
            return fg.inv_warp_grid(depth, rays, self.rot_mat, self.trans_xyz, self._K9,
                                    self.rot_center_depth)
        return self.grid_3d_to_2d(self.get_inv_warped_3d_grid(depth))

    def warp_canon_depth(self, canon_depth):
        b, h, w = canon_depth.shape
        grid_3d = self.get_warped_3d_grid(canon_depth).reshape(b, -1, 3)
        faces = get_face_idx(b, h, w, device=canon_depth.device)
        warped_depth = self.renderer.render_depth(grid_3d, faces)
        # allow some margin out of valid range
        margin = (self.max_depth - self.min_depth) / 2
        from ..op import clamp
        return clamp(warped_depth, self.min_depth - margin, self.max_depth + margin)

    def get_normal_from_depth(self, depth):
        b, h, w = depth.shape
        if self._use_fused(depth):
            return fg.normal_from_depth(depth, self._pixel_rays(h, w, depth.device).reshape(-1, 3))
        grid_3d = self.depth_to_3d_grid(depth)
        tu = grid_3d[:, 1:-1, 2:] - grid_3d[:, 1:-1, :-2]
        tv = grid_3d[:, 2:, 1:-1] - grid_3d[:, :-2, 1:-1]
        normal = torch.linalg.cross(tu, tv, dim=3)
        # border pixels get (0, 0, 1)
        normal = nn.functional.pad(normal.permute(0, 3, 1, 2), (1, 1, 1, 1)).permute(0, 2, 3, 1)
        def make_border():
            border = torch.zeros(1, h, w, 3, device=depth.device)
            border[..., 2] = 1
            border[:, 1:-1, 1:-1] = 0
            return border
        normal = normal + self._const(("border", h, w, str(depth.device)), make_border)
        return normal / (((normal ** 2).sum(3, keepdim=True)) ** 0.5 + EPS)

    def downscale_K(self, downscale):
        """renderer.py:56-59."""
        if downscale > 1:
            self.K = torch.cat((self.K_origin[:, 0:2] / downscale, self.K_origin[:, 2:]), dim=1)
            self.inv_K = torch.inverse(self.K[0]).unsqueeze(0)
            self._rays.clear()
            self._K9 = tuple(float(v) for v in self.K[0].reshape(9).tolist())
            # like the reference, the nr.Renderer built in __init__ keeps the original intrinsics

    # ------------------------------------------------------------------ texture path (visualisation)
    def _render_textured(self, verts, im):
        """render_rgb of the grid mesh `verts` (b, h*w, 3) coloured by `im` (b, c, h, w), clamped to
        [-1, 1] (renderer.py:196,272)."""
        b, c, h, w = im.shape
        faces = get_face_idx(b, h, w, device=im.device)
        textures = get_textures_from_im(im, tx_size=self.tex_cube_size)
        return self.renderer.render_rgb(verts, faces, textures).clamp(min=-1., max=1.)

    def _warp_by_grid_sample(self, im, depth, view):
        self.set_transform_matrices(view)
        recon_depth = self.warp_canon_depth(depth)
        grid = self.get_inv_warped_2d_grid(recon_depth)
        return nn.functional.grid_sample(im, grid, mode='bilinear', align_corners=True)

    def _canonical_mesh(self, depth, v_before):
        """Vertices of the depth mesh, taken back to the canonical pose when the depth was predicted
        under view `v_before` (renderer.py:166-170,208-212)."""
        b = depth.shape[0]
        verts = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
        if v_before is not None:
            rot_mat, trans_xyz = get_transform_matrices(v_before)
            verts = self.rotate_pts(self.translate_pts(verts, -trans_xyz), rot_mat.transpose(2, 1))
        return verts

    def _sweep(self, im, depth, poses, v_before, v_after, grid_sample, verts):
        """One rendering per pose = (rx, ry, rz) angle triple -> (b, len(poses), c, h, w)."""
        b = im.shape[0]
        frames = []
        for i, angles in enumerate(poses):
            if grid_sample:
                view = torch.tensor([list(angles) + [0., 0., 0.]], dtype=torch.float32, device=im.device)
                if v_before is not None:
                    view = view - v_before
                frames.append(self._warp_by_grid_sample(im, depth, view))
                continue
            rot_i, _ = get_transform_matrices(torch.tensor([list(angles)], dtype=torch.float32, device=im.device))
            posed = self.rotate_pts(verts, rot_i.repeat(b, 1, 1))
            if v_after is not None:
                v_i = v_after[i] if v_after.dim() == 3 else v_after
                rot_mat, trans_xyz = get_transform_matrices(v_i)
                posed = self.translate_pts(self.rotate_pts(posed, rot_mat), trans_xyz)
            frames.append(self._render_textured(posed, im))
        return torch.stack(frames, 1)

    def render_yaw(self, im, depth, v_before=None, v_after=None, rotations=None, maxr=90, nsample=9,
                   grid_sample=False, crop_mesh=None):
        """Yaw sweep of the textured depth mesh (renderer.py:141-198)."""
        grid_3d = self.depth_to_3d_grid(depth)
        if crop_mesh is not None:   # flatten the border band onto its inner neighbour (renderer.py:145-158)
            grid_3d = grid_3d.clone()
            top, bottom, left, right = crop_mesh
            if top > 0:
                grid_3d[:, :top, :, 1:] = grid_3d[:, top:top + 1, :, 1:]
            if bottom > 0:
                grid_3d[:, -bottom:, :, 1:] = grid_3d[:, -bottom - 1:-bottom, :, 1:]
            if left > 0:
                grid_3d[:, :, :left, 0::2] = grid_3d[:, :, left:left + 1, 0::2]
            if right > 0:
                grid_3d[:, :, -right:, 0::2] = grid_3d[:, :, -right - 1:-right, 0::2]
        b = im.shape[0]
        verts = grid_3d.reshape(b, -1, 3)
        if v_before is not None:
            rot_mat, trans_xyz = get_transform_matrices(v_before)
            verts = self.rotate_pts(self.translate_pts(verts, -trans_xyz), rot_mat.transpose(2, 1))
        if rotations is None:
            rotations = torch.linspace(-math.pi / 180 * maxr, math.pi / 180 * maxr, nsample)
        poses = [(0., float(r), 0.) for r in rotations]
        return self._sweep(im, depth, poses, v_before, v_after, grid_sample, verts)

    def render_view(self, im, depth, v_before=None, rotations=None, maxr=[20, 90], nsample=[5, 9],
                    grid_sample=False):
        """Yaw sweep followed by a pitch sweep (renderer.py:200-250)."""
        verts = self._canonical_mesh(depth, v_before)
        pitch = torch.linspace(-math.pi / 180 * maxr[0], math.pi / 180 * maxr[0], nsample[0])
        yaw = torch.linspace(-math.pi / 180 * maxr[1], math.pi / 180 * maxr[1], nsample[1])
        poses = [(0., float(y), 0.) for y in yaw] + [(float(p), 0., 0.) for p in pitch]
        return self._sweep(im, depth, poses, v_before, None, grid_sample, verts)

    def render_given_view(self, im, depth, view, mask=None, grid_sample=True):
        """renderer.py:252-277: warp `im` (and `mask`) to `view` — by inverse-warp sampling
        (grid_sample=True, the training path) or by rendering the textured mesh."""
        if not grid_sample:
            b = im.shape[0]
            rot_mat, trans_xyz = get_transform_matrices(view)
            verts = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
            verts = self.translate_pts(self.rotate_pts(verts, rot_mat), trans_xyz)
            warped_images = self._render_textured(verts, im)
            if mask is not None:
                return warped_images, self._render_textured(verts, mask)
            return warped_images
        self.set_transform_matrices(view)
        recon_depth = self.warp_canon_depth(depth)
        grid_2d_from_canon = self.get_inv_warped_2d_grid(recon_depth)
        warped_images = nn.functional.grid_sample(im, grid_2d_from_canon, mode='bilinear',
                                                  align_corners=True)
        if mask is not None:
            warped_mask = nn.functional.grid_sample(mask, grid_2d_from_canon, mode='nearest',
                                                    align_corners=True)
            return warped_images, warped_mask
        return warped_images
