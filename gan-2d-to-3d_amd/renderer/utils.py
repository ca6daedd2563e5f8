"""Mirror of the hot-path helpers of GAN2Shape/renderer/utils.py (get_grid :22-30,
get_rotation_matrix :33-49, get_transform_matrices :52-73, get_face_idx :76-80) and of the
texture-cube helpers that feed render_rgb (vcolor_to_texture_cube :83-96, get_textures_from_im
:99-109).

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


def _cube_coefficients(device):
    """(8, 3): weight of vertex colour v at cube corner e = (e0, e1, e2), flattened e0*4 + e1*2 + e2,
    such that trilinear interpolation of the cube at barycentric coordinates (w0, w1, w2), w0 + w1 +
    w2 = 1, gives sum_v w_v c_v.  By the number n of set coordinates: n = 0 -> 1/2 each; n = 1 -> the
    vertex of that axis; n = 2 -> +1/2 for the two set axes, -1/2 for the third; n = 3 -> 0."""
    rows = []
    for corner in range(8):
        e = [(corner >> 2) & 1, (corner >> 1) & 1, corner & 1]
        n = sum(e)
        rows.append({0: [0.5] * 3, 1: [float(x) for x in e], 2: [0.5 if x else -0.5 for x in e], 3: [0.0] * 3}[n])
    return torch.tensor(rows, dtype=torch.float32, device=device)


def vcolor_to_texture_cube(vcolors):
    """vertex colours (b, c, n, 3) -> texture cubes (b, n, 2, 2, 2, c)  (renderer/utils.py:83-96)."""
    b, c, n, _ = vcolors.shape
    cube = torch.matmul(_cube_coefficients(vcolors.device), vcolors.permute(0, 2, 3, 1))   # (b, n, 8, c)
    return cube.reshape(b, n, 2, 2, 2, c)


def get_textures_from_im(im, tx_size=1):
    """Per-face textures of the regular-grid mesh from an image (b, c, h, w)
    (renderer/utils.py:99-109).  tx_size 1: one colour per face — pixel (i, j) for faces1, pixel
    (i+1, j+1) for faces2.  tx_size 2: a 2x2x2 cube per face from the colours at the quad corners, in
    the reference's order ((i,j), (i,j+1), (i+1,j)) / ((i+1,j), (i,j+1), (i+1,j+1))."""
    b, c, h, w = im.shape
    tl, tr, bl, br = im[:, :, :h - 1, :w - 1], im[:, :, :h - 1, 1:], im[:, :, 1:, :w - 1], im[:, :, 1:, 1:]
    if tx_size == 1:
        flat = torch.cat([tl.reshape(b, c, -1), br.reshape(b, c, -1)], 2)
        return flat.transpose(2, 1).reshape(b, -1, 1, 1, 1, c)
    if tx_size == 2:
        first = torch.stack([tl, tr, bl], -1).reshape(b, c, -1, 3)
        second = torch.stack([bl, tr, br], -1).reshape(b, c, -1, 3)
        return vcolor_to_texture_cube(torch.cat([first, second], 2))
    raise NotImplementedError("Currently support texture size of 1 or 2 only.")
