"""Analytic known-answer tests for the rasterizer oracle (oracle/raster_body.inc).

PARITY UNPINNED: the reference's rasterizer is the external, un-vendored CUDA package
neural_renderer (README.md:32-37); the reference holds no fixtures for it.  These tests pin the
restatement of SURVEY.md Appendix A to closed-form answers instead:
  * output pixel (r, c) samples the mesh at grid coordinates (r + {.25,.75}, c + {.25,.75});
    the last row and last column are background (Appendix A item 3);
  * perspective-correct interpolation reproduces a 3-D plane exactly;
  * fill_back renders back-facing triangles through their reversed copy (face id >= F);
  * strict-< z-test: nearest wins, ties go to the lowest face id; near/far rejection;
  * the analytic backward equals finite differences of the fp64 forward (coverage fixed).
"""
import math

import numpy as np
import pytest

from oracle import capi
from oracle import geometry as og

FAR = 100.0


def _canon(S, depth, view=None):
    geo = og.Geometry(S, 0.9, 1.1, rot_center_depth=1.0, fov=10)
    geo.set_transform_matrices(np.zeros((depth.shape[0], 6), np.float32) if view is None else view)
    verts = geo.get_warped_3d_grid(depth.astype(np.float32)).reshape(depth.shape[0], -1, 3)
    return geo, verts, og.get_face_idx(1, S, S)[0]


@pytest.mark.parametrize("S", [8, 16])
def test_flat_plane(S):
    geo, verts, faces = _canon(S, np.full((1, S, S), 1.0))
    out = capi.render_depth(verts, faces, S, geo.K[0], far=FAR)
    d = out["depth"][0]
    np.testing.assert_allclose(d[:S - 1, :S - 1], 1.0, rtol=2e-6)
    np.testing.assert_array_equal(d[S - 1, :], FAR)
    np.testing.assert_array_equal(d[:, S - 1], FAR)
    F = faces.shape[0]
    fi = out["face_idx"][0]
    assert ((fi >= -1) & (fi < 2 * F)).all()
    # canonical orientation: every covered sample is won by an un-reversed copy
    assert (fi[fi >= 0] < F).all()
    w = out["bary"][0][fi >= 0]
    np.testing.assert_allclose(w.sum(-1), 1.0, rtol=1e-5)
    assert (w >= 0).all() and (w <= 1).all()


def test_tilted_plane_perspective_correct():
    S = 16
    geo = og.Geometry(S, fov=10)
    fx, cx = geo.K[0, 0, 0], geo.K[0, 0, 2]
    a, b, c0 = 0.8, -0.5, 1.0
    v, u = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")

    def plane(uu, vv):  # depth of the plane z = a x + b y + c0 along the ray through (u, v)
        return c0 / (1 - a * (uu - cx) / fx - b * (vv - cx) / fx)

    depth = plane(u, v)[None]
    geo, verts, faces = _canon(S, depth)
    out = capi.render_depth(verts.astype(np.float64), faces, S, geo.K[0], far=FAR, dtype=np.float64)
    exp = np.zeros((S, S))
    for dr in (0.25, 0.75):
        for dc in (0.25, 0.75):
            exp += plane(u + dc, v + dr) / 4
    np.testing.assert_allclose(out["depth"][0][:S - 1, :S - 1], exp[:S - 1, :S - 1], rtol=1e-6)
    out32 = capi.render_depth(verts, faces, S, geo.K[0], far=FAR)
    np.testing.assert_allclose(out32["depth"][0][:S - 1, :S - 1], exp[:S - 1, :S - 1], rtol=2e-5)


def test_integer_shift():
    """Translating the flat plane by k/fx moves it k pixels to the right: columns < k background."""
    S, k = 16, 2.0
    geo = og.Geometry(S, fov=10)
    view = np.array([[0, 0, 0, k / geo.K[0, 0, 0], 0, 0]], np.float32)
    geo, verts, faces = _canon(S, np.full((1, S, S), 1.0), view)
    d = capi.render_depth(verts.astype(np.float64), faces, S, geo.K[0], far=FAR,
                          dtype=np.float64)["depth"][0]
    np.testing.assert_array_equal(d[:S - 1, :2], FAR)
    np.testing.assert_allclose(d[:S - 1, 2:], 1.0, rtol=1e-9)  # mesh now extends past the right edge
    np.testing.assert_array_equal(d[S - 1], FAR)


def test_fill_back_two_sided():
    S = 8
    view = np.array([[0, math.pi, 0, 0, 0, 0]], np.float32)  # look at the back of the plane
    geo, verts, faces = _canon(S, np.full((1, S, S), 1.0), view)
    F = faces.shape[0]
    out = capi.render_depth(verts, faces, S, geo.K[0], far=FAR, fill_back=True)
    d, fi = out["depth"][0], out["face_idx"][0]
    assert (fi[fi >= 0] >= F).all() and (fi >= 0).sum() > 0.7 * fi.size
    # mirrored mesh spans u' = S-1-u in [0, S-1]: same coverage as the canonical pose
    np.testing.assert_allclose(d[:S - 1, :S - 1], 1.0, rtol=1e-5)
    out = capi.render_depth(verts, faces, S, geo.K[0], far=FAR, fill_back=False)
    np.testing.assert_array_equal(out["depth"][0], FAR)
    assert (out["face_idx"] == -1).all()


def _quad(z, x0=-0.05, x1=0.05):
    # counter-clockwise seen by the rasterizer after the v flip == front facing (checked below)
    return np.array([[x0, x0, z], [x0, x1, z], [x1, x0, z], [x1, x1, z]], np.float32)


def test_zbuffer_tiebreak_near_far():
    S = 8
    geo = og.Geometry(S, fov=10)
    K = geo.K[0]
    va, vb = _quad(1.0), _quad(1.05)
    verts = np.concatenate([vb, va, va])[None]  # far quad first, then two coincident near quads
    tri = np.array([[0, 1, 2], [2, 1, 3]], np.int32)
    faces = np.concatenate([tri, tri + 4, tri + 8])
    out = capi.render_depth(verts, faces, S, K, far=FAR, fill_back=True)
    fi, dss = out["face_idx"][0], out["depth_ss"][0]
    cov = fi >= 0
    assert cov.sum() > 50
    np.testing.assert_allclose(dss[cov], 1.0, rtol=1e-6)           # nearest wins
    F = faces.shape[0]
    assert set(np.unique(fi[cov] % F)) <= {2, 3}                   # first of the coincident pair
    # near / far rejection: nothing closer than near or farther than far is drawn
    out = capi.render_depth(verts, faces, S, K, near=1.02, far=FAR, fill_back=True)
    np.testing.assert_allclose(out["depth_ss"][0][out["face_idx"][0] >= 0], 1.05, rtol=1e-6)
    out = capi.render_depth(verts, faces, S, K, near=0.1, far=1.02, fill_back=True)
    np.testing.assert_allclose(out["depth_ss"][0][out["face_idx"][0] >= 0], 1.0, rtol=1e-6)
    assert (out["depth_ss"][0][out["face_idx"][0] < 0] == np.float32(1.02)).all()


def test_backward_matches_finite_differences():
    S = 8
    rng = np.random.default_rng(0)
    depth = 1.0 + 0.05 * rng.standard_normal((2, S, S))
    view = np.array([[0.2, -0.3, 0.1, 0.01, -0.02, 0.03], [-0.1, 0.25, 0.0, 0.0, 0.01, -0.02]],
                    np.float32)
    geo, verts, faces = _canon(S, depth, view)
    verts = verts.astype(np.float64)
    K = geo.K[0].astype(np.float64)
    out = capi.render_depth(verts, faces, S, K, far=FAR, dtype=np.float64)
    cov = np.minimum(out["depth"], 5.0) < 5.0           # pixels with no background sample
    g = rng.standard_normal(out["depth"].shape) * cov   # background samples carry no gradient anyway
    gv = capi.render_depth_bwd(verts, faces, g, out["face_idx"], out["bary"], S, K, dtype=np.float64)
    assert np.abs(gv).max() > 0
    eps = 1e-7
    idx = rng.choice(verts.size, 40, replace=False)
    for flat in idx:
        vp, vm = verts.copy(), verts.copy()
        vp.flat[flat] += eps
        vm.flat[flat] -= eps
        op = capi.render_depth(vp, faces, S, K, far=FAR, dtype=np.float64)
        om = capi.render_depth(vm, faces, S, K, far=FAR, dtype=np.float64)
        if not (np.array_equal(op["face_idx"], out["face_idx"]) and
                np.array_equal(om["face_idx"], out["face_idx"])):
            continue  # coverage changed: the renderer's gradient ignores that by design
        fd = ((op["depth"] - om["depth"]) * g).sum() / (2 * eps)
        np.testing.assert_allclose(gv.flat[flat], fd, rtol=2e-4, atol=1e-6)


def test_f32_close_to_f64_on_random_view():
    S = 16
    rng = np.random.default_rng(1)
    depth = 1.0 + 0.03 * rng.standard_normal((1, S, S))
    view = np.array([[0.3, 0.5, -0.2, 0.02, -0.01, 0.04]], np.float32)
    geo, verts, faces = _canon(S, depth, view)
    o32 = capi.render_depth(verts, faces, S, geo.K[0], far=FAR)
    o64 = capi.render_depth(verts.astype(np.float64), faces, S, geo.K[0], far=FAR, dtype=np.float64)
    same = o32["face_idx"] == o64["face_idx"]
    assert same.mean() > 0.995
    np.testing.assert_allclose(o32["depth_ss"][same], o64["depth_ss"][same], rtol=1e-5)


# ----------------------------------------------------------------------------- render_rgb (texture pass)
def _np_textures(im, tx_size):
    import torch
    from gan2shape_amd.renderer import utils as ru
    return ru.get_textures_from_im(torch.as_tensor(im), tx_size).numpy()


def test_render_rgb_constant_texture_and_background():
    """A constant colour comes back unchanged wherever the mesh covers all 4 supersamples, the
    background colour where nothing does, their average on the silhouette (last row / column:
    Appendix A item 3)."""
    S = 8
    geo, verts, faces = _canon(S, np.full((1, S, S), 1.0))
    im = np.broadcast_to(np.array([0.25, -0.5, 0.75], np.float32)[None, :, None, None], (1, 3, S, S)).copy()
    for ts in (1, 2):
        out = capi.render_rgb(verts, faces, _np_textures(im, ts), S, geo.K[0], background=(1, 1, 1))
        np.testing.assert_allclose(out[0, :, :S - 1, :S - 1], im[0, :, :S - 1, :S - 1], atol=2e-6)
        np.testing.assert_array_equal(out[0, :, S - 1, :], 1.0)
        np.testing.assert_array_equal(out[0, :, :, S - 1], 1.0)


def test_render_rgb_texture_cube_is_barycentric_interpolation():
    """tx_size = 2 (renderer/utils.py:83-109): trilinear reading of the 2x2x2 cube at the barycentric
    weights interpolates the face's three vertex colours (up to the package's eps = 1e-3 on the cube
    coordinates).  On a fronto-parallel plane the colours handed to a face in the REFERENCE's order
    make every output pixel a fixed convex combination of its quad's corner colours."""
    S = 12
    geo, verts, faces = _canon(S, np.full((1, S, S), 1.0))
    yy, xx = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    ramp = np.stack([0.05 * xx, 0.03 * yy, 0.02 * xx + 0.01 * yy]).astype(np.float32)[None]
    out = capi.render_rgb(verts, faces, _np_textures(ramp, 2), S, geo.K[0], dtype=np.float64)
    # supersample (r + a, c + b), a, b in {.25, .75}: a + b < 1 lies in faces1 = (i,j),(i+1,j),(i,j+1)
    # with weights (1-a-b, a, b) and the colours tl, tr, bl in that order; a + b > 1 in faces2 =
    # (i,j+1),(i+1,j),(i+1,j+1) with weights (1-a, 1-b, a+b-1) and colours bl, tr, br; a + b = 1: faces1
    tl, tr, bl, br = ramp[0, :, :-1, :-1], ramp[0, :, :-1, 1:], ramp[0, :, 1:, :-1], ramp[0, :, 1:, 1:]
    exp = 0
    for a in (.25, .75):
        for b in (.25, .75):
            if a + b <= 1:
                exp = exp + (1 - a - b) * tl + a * tr + b * bl
            else:
                exp = exp + (1 - a) * bl + (1 - b) * tr + (a + b - 1) * br
    np.testing.assert_allclose(out[0, :, :S - 1, :S - 1], exp / 4, atol=3e-3)   # eps = 1e-3 of the cube edge


def test_render_rgb_fill_back_uses_the_permuted_cube():
    """A mesh seen from behind is drawn through the reversed copies of its faces, whose cubes are
    textures.permute(0,1,4,3,2,5): the picture equals the front view of the mirrored mesh."""
    S = 10
    rng = np.random.default_rng(0)
    im = rng.uniform(-1, 1, (1, 3, S, S)).astype(np.float32)
    geo, verts, faces = _canon(S, np.full((1, S, S), 1.0))
    tex = _np_textures(im, 2)
    front = capi.render_rgb(verts, faces, tex, S, geo.K[0], fill_back=True)
    nofill = capi.render_rgb(verts, faces, tex, S, geo.K[0], fill_back=False)
    np.testing.assert_array_equal(front, nofill)            # front-facing: the copies never win
    flipped = faces[:, ::-1].copy()                         # every face back-facing
    back_only = capi.render_rgb(verts, flipped, tex, S, geo.K[0], fill_back=False)
    np.testing.assert_array_equal(back_only, 1.0)           # nothing drawn without fill_back
    tex_rev = np.ascontiguousarray(tex.transpose(0, 1, 4, 3, 2, 5))
    both = capi.render_rgb(verts, flipped, tex_rev, S, geo.K[0], fill_back=True)
    np.testing.assert_allclose(both, front, atol=1e-6)      # reversed order + permuted cube = the original
