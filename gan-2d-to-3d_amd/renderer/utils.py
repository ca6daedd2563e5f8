"""Mirror of the hot-path helpers of GAN2Shape/renderer/utils.py (get_grid :22-30,
get_rotation_matrix :33-49, get_transform_matrices :52-73, get_face_idx :76-80).  The texture-cube
helpers (:83-109) only feed render_rgb and are out of scope.

Device-agnostic torch code (the -m "not gpu" tests pin it against the golden vectors on CPU);
everything is created directly on the device of its input instead of on the CPU + .to(device).
"""
import torch


def get_grid(b, H, W, normalize=True, device=None):
    """(b, H, W, 2) grid of (x, y) coordinates."""
    if normalize:
        h_range = torch.linspace(-1, 1, H, device=device)
        w_range = torch.linspace(-1, 1, W, device=device)
    else:
        h_range = torch.arange(0, H, device=device)
        w_range = torch.arange(0, W, device=device)
    hh, ww = torch.meshgrid(h_range, w_range, indexing="ij")
    return torch.stack([ww, hh], -1).float().unsqueeze(0).repeat(b, 1, 1, 1)


def get_rotation_matrix(tx, ty, tz):
    """R = Rz @ Ry @ Rx, angles in radians, each (b,)."""
    zero, one = torch.zeros_like(tx), torch.ones_like(tx)
    cx, sx, cy, sy, cz, sz = tx.cos(), tx.sin(), ty.cos(), ty.sin(), tz.cos(), tz.sin()
    m_x = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], 1).view(-1, 3, 3)
    m_y = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], 1).view(-1, 3, 3)
    m_z = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], 1).view(-1, 3, 3)
    return torch.matmul(m_z, torch.matmul(m_y, m_x))


def get_transform_matrices(view):
    """view (b, 3 | 5 | 6) = (rx, ry, rz[, tx, ty[, tz]]) -> rot (b,3,3), trans (b,1,3)."""
    b = view.size(0)
    if view.size(1) == 6:
        trans_xyz = view[:, 3:].reshape(b, 1, 3)
    elif view.size(1) == 5:
        trans_xyz = torch.cat([view[:, 3:].reshape(b, 1, 2), view.new_zeros(b, 1, 1)], 2)
    elif view.size(1) == 3:
        trans_xyz = view.new_zeros(b, 1, 3)
    else:
        raise Exception("Unsupported view size. size(1) must be either 3, 5, 6.")
    return get_rotation_matrix(view[:, 0], view[:, 1], view[:, 2]), trans_xyz


_FACE_CACHE = {}


def get_face_idx(b, h, w, device=None):
    """(b, 2(h-1)(w-1), 3) int32: all faces1 = (i,j),(i+1,j),(i,j+1), then all faces2 =
    (i,j+1),(i+1,j),(i+1,j+1).  Built once per (h, w, device) instead of on the CPU every call
    (renderer.py:119); the returned tensor is tagged so that the rasterizer shim takes the
    implicit-topology path without comparing face lists."""
    key = (h, w, str(device))
    f = _FACE_CACHE.get(key)
    if f is None:
        idx_map = torch.arange(h * w, device=device).reshape(h, w)
        faces1 = torch.stack([idx_map[:h - 1, :w - 1], idx_map[1:, :w - 1], idx_map[:h - 1, 1:]], -1)
        faces2 = torch.stack([idx_map[:h - 1, 1:], idx_map[1:, :w - 1], idx_map[1:, 1:]], -1)
        f = torch.cat([faces1.reshape(-1, 3), faces2.reshape(-1, 3)], 0).int()
        _FACE_CACHE[key] = f
    out = f.unsqueeze(0).expand(b, -1, -1)
    if h == w:
        out._g2s_regular_grid = h
    return out
