"""ctypes bindings of oracle/libg2s_oracle.so — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


SANITIZE = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


def build_sanitized(out_dir):
    """The same sources under AddressSanitizer + UBSan (CPU only; tests/test_sanitizers_cpu.py loads
    it through G2S_ORACLE_LIB in a child process that preloads libasan)."""
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libg2s_oracle_san.so")
    subprocess.check_call(["gcc"] + SANITIZE + ["-fPIC", "-ffp-contract=off", "-fopenmp", "-shared", "-o", so,
                                                os.path.join(_HERE, "g2s_oracle.c"), "-lm"])
    return so


def build(force=False):
    if os.environ.get("G2S_ORACLE_LIB"):     # a sanitizer build made by build_sanitized()
        return os.environ["G2S_ORACLE_LIB"]
    so = os.path.join(_HERE, "libg2s_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("g2s_oracle.c", "raster_body.inc")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libg2s_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """op/fused_bias_act_kernel.cu:19-49. x any shape (N, C, ...), bias (C,) or None."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    step_b = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    r = None if ref is None else np.ascontiguousarray(ref, np.float32)
    lib().g2s_oracle_fused_bias_act_f32(
        _p(x, C.c_float), _p(b, C.c_float), _p(r, C.c_float), _p(y, C.c_float),
        C.c_long(x.size), C.c_long(step_b), C.c_long(0 if b is None else b.size),
        C.c_int(act), C.c_int(grad), C.c_float(alpha), C.c_float(scale))
    return y


def upfirdn2d(x, k, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0)):
    """op/upfirdn2d.py:157-198. x (N, C, H, W); k (kh, kw); pad = (x0, x1, y0, y1)."""
    x = np.ascontiguousarray(x, np.float32)
    k = np.ascontiguousarray(k, np.float32)
    n, c, h, w = x.shape
    kh, kw = k.shape
    oh = (h * up[1] + pad[2] + pad[3] - kh) // down[1] + 1
    ow = (w * up[0] + pad[0] + pad[1] - kw) // down[0] + 1
    y = np.empty((n, c, oh, ow), np.float32)
    lib().g2s_oracle_upfirdn2d_f32(
        _p(x, C.c_float), _p(k, C.c_float), _p(y, C.c_float), C.c_int(n * c), C.c_int(h),
        C.c_int(w), C.c_int(kh), C.c_int(kw), C.c_int(up[0]), C.c_int(up[1]), C.c_int(down[0]),
        C.c_int(down[1]), C.c_int(pad[0]), C.c_int(pad[1]), C.c_int(pad[2]), C.c_int(pad[3]))
    return y


def modconv(x, weight, style, scale, demodulate, mode):
    """stylegan2-pytorch/model.py:250-291. weight (Cout, Cin, k, k); style (B, Cin) post-modulation."""
    x = np.ascontiguousarray(x, np.float32)
    weight = np.ascontiguousarray(weight, np.float32)
    style = np.ascontiguousarray(style, np.float32)
    b, cin, h, w = x.shape
    cout, _, k, _ = weight.shape
    if mode == 0:
        ho, wo = h, w
    elif mode == 1:
        ho, wo = (h - 1) * 2 + k, (w - 1) * 2 + k
    else:
        ho, wo = (h - k) // 2 + 1, (w - k) // 2 + 1
    y = np.empty((b, cout, ho, wo), np.float32)
    lib().g2s_oracle_modconv_f32(
        _p(x, C.c_float), _p(weight, C.c_float), _p(style, C.c_float), C.c_float(scale),
        C.c_int(int(demodulate)), _p(y, C.c_float), C.c_int(b), C.c_int(cin), C.c_int(cout),
        C.c_int(h), C.c_int(w), C.c_int(k), C.c_int(mode))
    return y


def _real(dtype):
    if dtype == np.float32:
        return C.c_float, "_f32"
    return C.c_double, "_f64"


def render_depth(verts, faces, S, K, orig_size=None, ssaa=2, fill_back=True, near=0.1, far=100.0,
                 dtype=np.float32):
    """neural_renderer Renderer.render_depth (external).

    verts (B, N, 3); faces (F, 3) int; K (3, 3).  Returns dict(depth (B,S,S), face_idx, bary,
    depth_ss (unflipped raster))."""
    ct, suf = _real(dtype)
    verts = np.ascontiguousarray(verts, dtype)
    faces = np.ascontiguousarray(faces, np.int32)
    K = np.ascontiguousarray(K, dtype).reshape(9)
    B, N, _ = verts.shape
    F = faces.shape[0]
    isz = S * ssaa
    depth = np.empty((B, S, S), dtype)
    fidx = np.empty((B, isz, isz), np.int32)
    bary = np.empty((B, isz, isz, 3), dtype)
    dss = np.empty((B, isz, isz), dtype)
    fn = getattr(lib(), "g2s_oracle_render_depth" + suf)
    rc = fn(_p(verts, ct), _p(faces, C.c_int), C.c_int(B), C.c_int(N), C.c_int(F), C.c_int(S),
            _p(K, ct), ct(S if orig_size is None else orig_size), C.c_int(ssaa),
            C.c_int(int(fill_back)), ct(near), ct(far), _p(depth, ct), _p(fidx, C.c_int),
            _p(bary, ct), _p(dss, ct))
    assert rc == 0
    return dict(depth=depth, face_idx=fidx, bary=bary, depth_ss=dss)


def render_depth_bwd(verts, faces, grad_depth, face_idx, bary, S, K, orig_size=None, ssaa=2,
                     fill_back=True, dtype=np.float32):
    ct, suf = _real(dtype)
    verts = np.ascontiguousarray(verts, dtype)
    faces = np.ascontiguousarray(faces, np.int32)
    K = np.ascontiguousarray(K, dtype).reshape(9)
    grad_depth = np.ascontiguousarray(grad_depth, dtype)
    face_idx = np.ascontiguousarray(face_idx, np.int32)
    bary = np.ascontiguousarray(bary, dtype)
    B, N, _ = verts.shape
    F = faces.shape[0]
    gv = np.zeros((B, N, 3), dtype)
    fn = getattr(lib(), "g2s_oracle_render_depth_bwd" + suf)
    rc = fn(_p(verts, ct), _p(faces, C.c_int), _p(grad_depth, ct), _p(face_idx, C.c_int),
            _p(bary, ct), C.c_int(B), C.c_int(N), C.c_int(F), C.c_int(S), _p(K, ct),
            ct(S if orig_size is None else orig_size), C.c_int(ssaa), C.c_int(int(fill_back)),
            _p(gv, ct))
    assert rc == 0
    return gv


def render_rgb(verts, faces, textures, S, K, orig_size=None, ssaa=2, fill_back=True, near=0.1, far=10.0,
               background=(1.0, 1.0, 1.0), eps=1e-3, dtype=np.float32):
    """neural_renderer Renderer.render_rgb (external; recalled semantics, PARITY UNPINNED).
    textures (B, F, ts, ts, ts, C).  Returns (B, C, S, S)."""
    ct, suf = _real(dtype)
    d = render_depth(verts, faces, S, K, orig_size, ssaa, fill_back, near, far, dtype)
    verts = np.ascontiguousarray(verts, dtype)
    faces = np.ascontiguousarray(faces, np.int32)
    textures = np.ascontiguousarray(textures, dtype)
    B, N, _ = verts.shape
    F, ts, Cc = faces.shape[0], textures.shape[2], textures.shape[5]
    bg = np.ascontiguousarray(np.asarray(background, dtype)[:Cc])
    out = np.empty((B, Cc, S, S), dtype)
    fn = getattr(lib(), "g2s_oracle_render_rgb" + suf)
    rc = fn(_p(verts, ct), _p(faces, C.c_int), _p(d["face_idx"], C.c_int), _p(d["bary"], ct), _p(textures, ct),
            C.c_int(B), C.c_int(N), C.c_int(F), C.c_int(S), C.c_int(ssaa), C.c_int(ts), C.c_int(Cc), _p(bg, ct),
            ct(eps), _p(out, ct))
    assert rc == 0
    return out
