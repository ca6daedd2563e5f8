"""Seeded rasterizer test scenes shared by the CPU and GPU parity tests."""
import math

import numpy as np

from oracle import geometry as og


def scene(S, B=2, seed=0, rot=60.0, step=True, noise=0.01):
    """GAN2Shape-like canonical depth in [0.9, 1.1] with a depth discontinuity (object vs
    background plane), viewed from random poses up to +-rot degrees / +-0.1 translation
    (model.py:49-57,330-335).  Returns (geo, verts (B, S*S, 3) f32, faces (F, 3) i32)."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    r2 = ((u - S / 2) ** 2 + (v - S / 2) ** 2) / (0.35 * S) ** 2
    depth = np.where(r2 < 1, 0.92 + 0.08 * r2, 1.08 if step else 1.0)
    depth = depth[None] + noise * rng.standard_normal((B, S, S))
    depth = np.clip(depth, 0.9, 1.1).astype(np.float32)
    view = np.concatenate([rng.uniform(-1, 1, (B, 3)) * rot * math.pi / 180,
                           rng.uniform(-1, 1, (B, 3)) * 0.1], 1).astype(np.float32)
    geo = og.Geometry(S, 0.9, 1.1, rot_center_depth=1.0, fov=10)
    geo.set_transform_matrices(view)
    verts = geo.get_warped_3d_grid(depth).reshape(B, -1, 3).astype(np.float32)
    return geo, verts, og.get_face_idx(1, S, S)[0]


def soup(n_faces=300, n_verts=200, seed=0, B=2):
    """Unstructured triangle soup around z = 1 (explicit-topology path)."""
    rng = np.random.default_rng(seed)
    verts = np.stack([rng.uniform(-0.09, 0.09, (B, n_verts)), rng.uniform(-0.09, 0.09, (B, n_verts)),
                      rng.uniform(0.9, 1.1, (B, n_verts))], -1).astype(np.float32)
    base = rng.integers(0, n_verts, n_faces)
    faces = np.stack([base, (base + rng.integers(1, 6, n_faces)) % n_verts,
                      (base + rng.integers(6, 12, n_faces)) % n_verts], -1).astype(np.int32)
    return verts, faces
