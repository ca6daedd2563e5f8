"""Pin oracle/geometry.py (numpy restatement of GAN2Shape/renderer/{utils,renderer}.py) against
golden vectors generated from the reference's own Python."""
import numpy as np

from oracle import geometry as og


def test_face_idx(golden):
    g = golden("geometry")
    np.testing.assert_array_equal(og.get_face_idx(2, 4, 4), g["face_idx_4x4"])
    np.testing.assert_array_equal(og.get_face_idx(1, 3, 5), g["face_idx_3x5"])
    assert og.get_face_idx(1, 4, 4).dtype == np.int32


def test_grid(golden):
    g = golden("geometry")
    np.testing.assert_allclose(og.get_grid(2, 3, 4, True), g["grid_norm"], atol=1e-7)
    np.testing.assert_array_equal(og.get_grid(1, 3, 4, False), g["grid_px"])


def test_transform_matrices(golden):
    g = golden("geometry")
    for n in (3, 5, 6):
        r, t = og.get_transform_matrices(g["view6"][:, :n])
        np.testing.assert_allclose(r, g[f"rot{n}"], atol=1e-6)
        np.testing.assert_allclose(t, g[f"trans{n}"], atol=0)


def test_renderer_geometry(golden):
    g = golden("geometry")
    S = g["r.depth"].shape[1]
    geo = og.Geometry(S, 0.9, 1.1, rot_center_depth=1.0, fov=10)
    np.testing.assert_allclose(geo.K, g["r.K"], rtol=1e-6)
    np.testing.assert_allclose(geo.inv_K, g["r.inv_K"], rtol=1e-5, atol=1e-7)
    geo.set_transform_matrices(g["r.view"])
    d = g["r.depth"]
    np.testing.assert_allclose(geo.depth_to_3d_grid(d), g["r.grid3d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(geo.get_warped_3d_grid(d), g["r.warped3d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(geo.get_inv_warped_3d_grid(d), g["r.invwarped3d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(geo.get_inv_warped_2d_grid(d), g["r.invwarped2d"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(geo.get_normal_from_depth(d), g["r.normal"], rtol=1e-3, atol=2e-4)
