"""numpy restatement of the reference's renderer geometry — TEST INFRASTRUCTURE ONLY.

Follows GAN2Shape/renderer/utils.py and GAN2Shape/renderer/renderer.py (line numbers per
function below).  Pinned by tests/golden/geometry.npz, generated from the reference's own
importable Python (tests/golden/make_golden.py).
"""
import math

import numpy as np

EPS = 1e-7  # renderer.py:10


def get_grid(b, H, W, normalize=True):
    """utils.py:22-30 — (b, H, W, 2) grid of (x, y)."""
    if normalize:
        h_range = np.linspace(-1, 1, H, dtype=np.float32)
        w_range = np.linspace(-1, 1, W, dtype=np.float32)
    else:
        h_range = np.arange(0, H)
        w_range = np.arange(0, W)
    hh, ww = np.meshgrid(h_range, w_range, indexing="ij")
    grid = np.stack([ww, hh], -1).astype(np.float32)  # flip(3): (h, w) -> (x, y)
    return np.broadcast_to(grid, (b, H, W, 2)).copy()


def get_rotation_matrix(tx, ty, tz):
    """utils.py:33-49 — R = Rz @ Ry @ Rx."""
    n = len(tx)
    m_x = np.zeros((n, 3, 3), np.float32)
    m_y = np.zeros((n, 3, 3), np.float32)
    m_z = np.zeros((n, 3, 3), np.float32)
    m_x[:, 1, 1], m_x[:, 1, 2] = np.cos(tx), -np.sin(tx)
    m_x[:, 2, 1], m_x[:, 2, 2] = np.sin(tx), np.cos(tx)
    m_x[:, 0, 0] = 1
    m_y[:, 0, 0], m_y[:, 0, 2] = np.cos(ty), np.sin(ty)
    m_y[:, 2, 0], m_y[:, 2, 2] = -np.sin(ty), np.cos(ty)
    m_y[:, 1, 1] = 1
    m_z[:, 0, 0], m_z[:, 0, 1] = np.cos(tz), -np.sin(tz)
    m_z[:, 1, 0], m_z[:, 1, 1] = np.sin(tz), np.cos(tz)
    m_z[:, 2, 2] = 1
    return np.matmul(m_z, np.matmul(m_y, m_x))


def get_transform_matrices(view):
    """utils.py:52-73 — view (b, 3|5|6) -> rot (b,3,3), trans (b,1,3)."""
    view = np.asarray(view, np.float32)
    b = view.shape[0]
    if view.shape[1] == 6:
        trans = view[:, 3:].reshape(b, 1, 3)
    elif view.shape[1] == 5:
        trans = np.concatenate([view[:, 3:].reshape(b, 1, 2), np.zeros((b, 1, 1), np.float32)], 2)
    elif view.shape[1] == 3:
        trans = np.zeros((b, 1, 3), np.float32)
    else:
        raise Exception("Unsupported view size. size(1) must be either 3, 5, 6.")
    return get_rotation_matrix(view[:, 0], view[:, 1], view[:, 2]), trans


def get_face_idx(b, h, w):
    """utils.py:76-80 — (b, 2(h-1)(w-1), 3) int32, all faces1 then all faces2."""
    idx = np.arange(h * w).reshape(h, w)
    f1 = np.stack([idx[:h - 1, :w - 1], idx[1:, :w - 1], idx[:h - 1, 1:]], -1).reshape(-1, 3)
    f2 = np.stack([idx[:h - 1, 1:], idx[1:, :w - 1], idx[1:, 1:]], -1).reshape(-1, 3)
    f = np.concatenate([f1, f2], 0).astype(np.int32)
    return np.broadcast_to(f, (b,) + f.shape).copy()


class Geometry:
    """renderer.py:13-139 minus the neural_renderer object."""

    def __init__(self, image_size, min_depth=0.9, max_depth=1.1, rot_center_depth=None, fov=10):
        self.image_size = image_size
        self.min_depth = min_depth
        self.max_depth = max_depth
        self.rot_center_depth = (min_depth + max_depth) / 2 if rot_center_depth is None \
            else rot_center_depth
        fx = (image_size - 1) / 2 / (math.tan(fov / 2 * math.pi / 180))  # renderer.py:35-38
        c = (image_size - 1) / 2
        self.K = np.array([[fx, 0., c], [0., fx, c], [0., 0., 1.]], np.float32)[None]
        self.inv_K = np.linalg.inv(self.K[0].astype(np.float64)).astype(np.float32)[None]
        self.rot_mat = None
        self.trans_xyz = None

    def set_transform_matrices(self, view):  # renderer.py:61-62
        self.rot_mat, self.trans_xyz = get_transform_matrices(view)

    def rotate_pts(self, pts, rot_mat):  # renderer.py:64-69
        c = np.array([0., 0., self.rot_center_depth], np.float32).reshape(1, 1, 3)
        return np.matmul(pts - c, rot_mat.transpose(0, 2, 1)) + c

    def depth_to_3d_grid(self, depth):  # renderer.py:74-80
        b, h, w = depth.shape
        g2 = get_grid(b, h, w, normalize=False)
        g3 = np.concatenate([g2, np.ones((b, h, w, 1), np.float32)], 3)
        g3 = np.matmul(g3, self.inv_K.transpose(0, 2, 1)[:, None]) * depth[..., None]
        return g3.astype(np.float32)

    def grid_3d_to_2d(self, g3):  # renderer.py:82-88
        b, h, w, _ = g3.shape
        g2 = g3 / g3[..., 2:]
        g2 = np.matmul(g2, self.K.transpose(0, 2, 1)[:, None])[..., :2]
        wh = np.array([w - 1, h - 1], np.float32).reshape(1, 1, 1, 2)
        return (g2 / wh * 2. - 1.).astype(np.float32)

    def get_warped_3d_grid(self, depth):  # renderer.py:90-95
        b, h, w = depth.shape
        g = self.depth_to_3d_grid(depth).reshape(b, -1, 3)
        g = self.rotate_pts(g, self.rot_mat) + self.trans_xyz
        return g.reshape(b, h, w, 3).astype(np.float32)

    def get_inv_warped_3d_grid(self, depth):  # renderer.py:97-102
        b, h, w = depth.shape
        g = self.depth_to_3d_grid(depth).reshape(b, -1, 3) - self.trans_xyz
        g = self.rotate_pts(g, self.rot_mat.transpose(0, 2, 1))
        return g.reshape(b, h, w, 3).astype(np.float32)

    def get_inv_warped_2d_grid(self, depth):  # renderer.py:110-114
        return self.grid_3d_to_2d(self.get_inv_warped_3d_grid(depth))

    def get_normal_from_depth(self, depth):  # renderer.py:127-139
        b, h, w = depth.shape
        g = self.depth_to_3d_grid(depth)
        tu = g[:, 1:-1, 2:] - g[:, 1:-1, :-2]
        tv = g[:, 2:, 1:-1] - g[:, :-2, 1:-1]
        n = np.cross(tu, tv, axis=3)
        out = np.zeros((b, h, w, 3), np.float32)
        out[..., 2] = 1.0
        out[:, 1:-1, 1:-1] = n
        return (out / (np.sqrt((out ** 2).sum(3, keepdims=True)) + EPS)).astype(np.float32)

    def warp_canon_depth(self, canon_depth, render_depth_fn):  # renderer.py:116-125
        """render_depth_fn(verts (b,hw,3), faces (F,3)) -> (b,h,w): the rasterizer under test."""
        b, h, w = canon_depth.shape
        g = self.get_warped_3d_grid(canon_depth).reshape(b, -1, 3)
        faces = get_face_idx(1, h, w)[0]
        d = render_depth_fn(g, faces)
        margin = (self.max_depth - self.min_depth) / 2
        return np.clip(d, self.min_depth - margin, self.max_depth + margin)
