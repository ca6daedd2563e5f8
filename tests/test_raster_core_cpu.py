"""CPU check of the product's rasterizer arithmetic + tile culling logic.

tests/raster_tile_emulation.cpp runs the structure of gan-2d-to-3d_amd/csrc/raster.hip (chunk
boxes -> face boxes -> candidate list -> covering faces -> lexicographic (depth, id) minimum, flip +
pooling) serially on the host with the SAME header (csrc/raster_core.h) the HIP kernels compile,
and must agree bit for bit with the brute-force oracle.  (The wavefront-specific parts — ballots,
LDS lists — are covered by the -m gpu tests.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import capi
from raster_cases import scene, soup

HERE = os.path.dirname(os.path.abspath(__file__))
FAR = 100.0


@pytest.fixture(scope="module")
def emul():
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    so = os.path.join(HERE, "_build", "libraster_emul.so")
    flags = ["-O2"]
    if os.environ.get("G2S_ORACLE_LIB"):   # the sanitizer pass (tests/test_sanitizers_cpu.py)
        so = os.path.join(HERE, "_build", "libraster_emul_san.so")
        flags = capi.SANITIZE
    subprocess.check_call(["g++"] + flags + ["-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                                             os.path.join(HERE, "raster_tile_emulation.cpp")])
    return C.CDLL(so)


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def run_emul(lib, verts, faces, S, K, ssaa, fill_back, near=0.1, far=FAR, implicit=False):
    B, N, _ = verts.shape
    F = faces.shape[0]
    isz = S * ssaa
    depth = np.empty((B, S, S), np.float32)
    fidx = np.empty((B, isz, isz), np.int32)
    bary = np.empty((B, isz, isz, 3), np.float32)
    stats = (C.c_long * 4)()
    Kf = np.ascontiguousarray(K, np.float32).reshape(9)
    rc = lib.g2s_emul_render_depth(_f(verts), _i(None if implicit else faces), B, N, F, S, _f(Kf),
                                   C.c_float(S), ssaa, int(fill_back), C.c_float(near),
                                   C.c_float(far), _f(depth), _i(fidx), _f(bary), stats)
    assert rc == 0
    return dict(depth=depth, face_idx=fidx, bary=bary, candidates=stats[0], fragments=stats[1],
                max_candidates=stats[2], max_chunks=stats[3])


@pytest.mark.parametrize("S,ssaa,fill_back,implicit", [
    (16, 2, True, True), (16, 2, True, False), (20, 2, True, True), (32, 2, True, True),
    (16, 1, True, True), (16, 2, False, True), (12, 1, False, False)])
def test_tile_algorithm_equals_bruteforce(emul, S, ssaa, fill_back, implicit):
    geo, verts, faces = scene(S, B=2, seed=S + ssaa)
    ref = capi.render_depth(verts, faces, S, geo.K[0], ssaa=ssaa, fill_back=fill_back, far=FAR)
    out = run_emul(emul, verts, faces, S, geo.K[0], ssaa, fill_back, implicit=implicit)
    np.testing.assert_array_equal(out["face_idx"], ref["face_idx"])
    np.testing.assert_array_equal(out["bary"], ref["bary"])
    np.testing.assert_array_equal(out["depth"], ref["depth"])
    isz = S * ssaa
    n_faces = faces.shape[0] * (2 if fill_back else 1)
    brute = 2 * isz * isz * n_faces
    assert out["candidates"] * 64 < brute / 4  # culling actually culls


def test_triangle_soup_explicit_topology(emul):
    verts, faces = soup()
    S = 16
    from oracle import geometry as og
    K = og.Geometry(S).K[0]
    for fill_back in (True, False):
        ref = capi.render_depth(verts, faces, S, K, fill_back=fill_back, far=FAR)
        out = run_emul(emul, verts, faces, S, K, 2, fill_back)
        np.testing.assert_array_equal(out["face_idx"], ref["face_idx"])
        np.testing.assert_array_equal(out["depth"], ref["depth"])
        assert (ref["face_idx"] >= 0).mean() > 0.3


def test_backward_arithmetic(emul):
    S = 16
    geo, verts, faces = scene(S, B=2, seed=3)
    fw = capi.render_depth(verts, faces, S, geo.K[0], far=FAR)
    rng = np.random.default_rng(0)
    g = rng.standard_normal((2, S, S)).astype(np.float32)
    g[fw["depth"] > 1.2] = 0  # what the clamp in warp_canon_depth does (renderer.py:123-124)
    ref = capi.render_depth_bwd(verts, faces, g, fw["face_idx"], fw["bary"], S, geo.K[0])
    ref64 = capi.render_depth_bwd(verts.astype(np.float64), faces, g.astype(np.float64),
                                  fw["face_idx"], fw["bary"].astype(np.float64), S,
                                  geo.K[0].astype(np.float64), dtype=np.float64)
    gv = np.empty_like(verts)
    Kf = np.ascontiguousarray(geo.K[0], np.float32).reshape(9)
    for fptr in (faces, None):
        rc = emul.g2s_emul_render_depth_bwd(_f(verts), _i(fptr), _f(g), _i(fw["face_idx"]),
                                            _f(fw["bary"]), 2, S * S, faces.shape[0], S, _f(Kf),
                                            C.c_float(S), 2, _f(gv))
        assert rc == 0
        scale = np.abs(ref64).max()
        assert scale > 0
        np.testing.assert_allclose(gv, ref64, atol=2e-5 * scale)
        np.testing.assert_allclose(gv, ref, atol=2e-5 * scale)
