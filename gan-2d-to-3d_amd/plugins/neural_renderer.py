"""Drop-in for the external CUDA package `neural_renderer` at the boundary GAN2Shape uses.

The reference binds it by name — `import neural_renderer as nr` (GAN2Shape/renderer/renderer.py:6),
builds `nr.Renderer(camera_mode='projection', light_intensity_ambient=1.0,
light_intensity_directional=0., K=K, R=R, t=t, near=.., far=.., image_size=S, orig_size=S,
fill_back=True, background_color=[1,1,1])` (renderer.py:47-54) and calls
`.render_depth(vertices [B,N,3] f32, faces [B,F,3] i32) -> [B,S,S] f32` (renderer.py:120),
differentiable w.r.t. `vertices`.  `.render_rgb(vertices, faces, textures [B,F,T,T,T,C])`
(renderer.py:196,230,248,272,275 — the visualisation helpers) is the forward texture pass
(g2s_raster_rgb_fwd); it returns a tensor without autograd history (the reference never
differentiates through it; the package's silhouette-gradient kernels are not rebuilt).

Semantics follow SURVEY.md Appendix A (the package itself is un-vendored and un-pinned, so parity
is checked against the oracle's restatement, not against the CUDA original):
`render_depth` rasterizes with the package defaults near=0.1 / far=100 (the constructor's near/far
only reach render_rgb/render_silhouettes), anti_aliasing=True (2x supersampling + average pool),
vertical flip, background = far.
"""
import torch
from torch.autograd import Function

from gan2shape_amd import lib as _lib
from gan2shape_amd import zeropool as _zp

DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100.0


def _workspace(device, nbytes):
    """Rasterizer scratch (projected vertices + chunk boxes), allocated PER CALL: under torch's
    caching allocator this is a free-list hit, and during a HIP-graph capture the block comes from
    the graph's private pool and stays reserved for every replay.  (A grown-on-demand shared
    buffer would be freed when a larger batch arrives while captured graphs still hold its
    address.)"""
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _regular_grid_faces(S, device):
    key = (S, device)
    f = _regular_grid_faces.cache.get(key)
    if f is None:
        idx = torch.arange(S * S, device=device).reshape(S, S)
        f1 = torch.stack([idx[:S - 1, :S - 1], idx[1:, :S - 1], idx[:S - 1, 1:]], -1).reshape(-1, 3)
        f2 = torch.stack([idx[:S - 1, 1:], idx[1:, :S - 1], idx[1:, 1:]], -1).reshape(-1, 3)
        f = torch.cat([f1, f2], 0).int()
        _regular_grid_faces.cache[key] = f
    return f


_regular_grid_faces.cache = {}


class RenderDepthFunction(Function):
    """vertices (B,N,3) camera space -> depth (B,S,S).  `faces` is (F,3) int32 or None (implicit
    regular grid of renderer/utils.py:76-80)."""

    @staticmethod
    def forward(ctx, vertices, faces, K, orig_size, image_size, anti_aliasing, fill_back, near, far):
        _lib.require_cuda(vertices)
        if vertices.dtype != torch.float32:
            raise RuntimeError("render_depth: vertices must be float32")
        verts = vertices.contiguous()
        B, N, _ = verts.shape
        S = int(image_size)
        ssaa = 2 if anti_aliasing else 1
        F = 2 * (S - 1) * (S - 1) if faces is None else faces.shape[0]
        need_grad = ctx.needs_input_grad[0]
        L = _lib.load()
        depth = torch.empty((B, S, S), dtype=torch.float32, device=verts.device)
        fidx = bary = None
        if need_grad:
            fidx = torch.empty((B, S * ssaa, S * ssaa), dtype=torch.int32, device=verts.device)
            bary = torch.empty((B, S * ssaa, S * ssaa, 3), dtype=torch.float32, device=verts.device)
        ws_bytes = L.g2s_raster_workspace_bytes(B, N, F, S)
        ws = _workspace(verts.device, ws_bytes)
        Kc = (_lib.C.c_float * 9)(*K)
        _lib.check(L.g2s_raster_depth_fwd(_lib.ptr(verts), _lib.ptr(faces), B, N, F, S, Kc,
                                          float(orig_size), ssaa, int(bool(fill_back)),
                                          float(near), float(far), _lib.ptr(depth), _lib.ptr(fidx),
                                          _lib.ptr(bary), _lib.ptr(ws), ws.numel(), _lib.stream()))
        if need_grad:
            ctx.save_for_backward(verts, faces, fidx, bary)
        ctx.meta = (K, float(orig_size), S, ssaa, F)
        return depth

    @staticmethod
    def backward(ctx, grad_depth):
        verts, faces, fidx, bary = ctx.saved_tensors
        K, orig_size, S, ssaa, F = ctx.meta
        B, N, _ = verts.shape
        g = grad_depth.contiguous().float()
        L = _lib.load()
        Kc = (_lib.C.c_float * 9)(*K)
        # the scatter target must start at zero — default mode: the gradient itself; deterministic mode: the
        # fixed-point scratch buffer (include/g2s.h).  A slice of the step's cleared pool when there is one
        # (the call then runs under lib.precleared), else the call clears it with a memset of its own.
        ws, ws_bytes, pre = None, 0, False
        if L.g2s_get_deterministic():
            ws_bytes = L.g2s_raster_bwd_workspace_bytes(B, N)
            ws = _zp.take(((ws_bytes + 3) // 4,), verts.device)
            pre = ws is not None
            if ws is None:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=verts.device)
            gv = torch.empty_like(verts)
        else:
            gv = _zp.take(tuple(verts.shape), verts.device)
            pre = gv is not None
            if gv is None:
                gv = torch.empty_like(verts)
        with _lib.precleared(pre):
            _lib.check(L.g2s_raster_depth_bwd_ex(_lib.ptr(verts), _lib.ptr(faces), _lib.ptr(g),
                                                 _lib.ptr(fidx), _lib.ptr(bary), B, N, F, S, Kc, orig_size,
                                                 ssaa, _lib.ptr(gv), _lib.ptr(ws), ws_bytes, _lib.stream()))
        return gv, None, None, None, None, None, None, None, None


class Renderer:
    """`nr.Renderer` for camera_mode='projection' (the only mode GAN2Shape uses)."""

    def __init__(self, image_size=256, anti_aliasing=True, background_color=[0, 0, 0],
                 fill_back=True, camera_mode='projection', K=None, R=None, t=None,
                 dist_coeffs=None, orig_size=1024, perspective=True, viewing_angle=30,
                 camera_direction=[0, 0, 1], near=0.1, far=100, light_intensity_ambient=0.5,
                 light_intensity_directional=0.5, light_color_ambient=[1, 1, 1],
                 light_color_directional=[1, 1, 1], light_direction=[0, 1, 0], **unknown):
        if camera_mode != 'projection':
            raise ValueError("only camera_mode='projection' is supported (renderer.py:47)")
        self.image_size = image_size
        self.anti_aliasing = anti_aliasing
        self.background_color = background_color
        self.fill_back = fill_back
        self.camera_mode = camera_mode
        self.K, self.R, self.t = K, R, t
        self.dist_coeffs = dist_coeffs
        self.orig_size = orig_size
        self.near, self.far = near, far
        self.light_intensity_ambient = light_intensity_ambient
        self.light_intensity_directional = light_intensity_directional
        self.rasterizer_eps = 1e-3
        self._K_host = None

    def _host_K(self, K):
        """K lives on the device in the reference; one D2H copy, cached per tensor version."""
        key = (K.data_ptr(), K._version)
        if self._K_host is None or self._K_host[0] != key:
            k = K.detach().float().reshape(-1, 3, 3)
            if k.shape[0] != 1:
                raise NotImplementedError("per-sample intrinsics are not supported (GAN2Shape uses K[1,3,3])")
            self._K_host = (key, tuple(k[0].reshape(9).cpu().tolist()))
        return self._K_host[1]

    def render_depth(self, vertices, faces, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        K = self.K if K is None else K
        R = self.R if R is None else R
        t = self.t if t is None else t
        orig_size = self.orig_size if orig_size is None else orig_size
        dist_coeffs = self.dist_coeffs if dist_coeffs is None else dist_coeffs
        if dist_coeffs is not None and bool((dist_coeffs != 0).any()):
            raise NotImplementedError("lens distortion is not supported (GAN2Shape never sets it)")
        if K is None:
            raise ValueError("camera_mode='projection' needs K")
        # R, t: identity / zero at GAN2Shape's call site (renderer.py:30-34); applied on the host
        # side otherwise so that the kernel keeps the R = I, t = 0 form.
        if self._needs_transform(R, t):
            if R is not None:
                vertices = torch.matmul(vertices, R.reshape(-1, 3, 3).transpose(2, 1))
            if t is not None:
                vertices = vertices + t.reshape(-1, 1, 3)
        S = self.image_size
        f = self._shared_faces(faces, vertices.shape[1], S)
        return RenderDepthFunction.apply(vertices, f, self._host_K(K), orig_size, S,
                                         self.anti_aliasing, self.fill_back, DEFAULT_NEAR,
                                         DEFAULT_FAR)

    def _needs_transform(self, R, t):
        key = tuple(None if x is None else (x.data_ptr(), x._version) for x in (R, t))
        cache = getattr(self, "_rt_cache", None)
        if cache is None or cache[0] != key:
            ident = True
            if R is not None:
                r = R.detach().reshape(-1, 3, 3).float().cpu()
                ident = ident and bool((r == torch.eye(3)).all())
            if t is not None:
                ident = ident and bool((t.detach().cpu() == 0).all())
            self._rt_cache = cache = (key, not ident)
        return cache[1]

    def _shared_faces(self, faces, n_verts, S):
        """Returns None for the regular-grid topology (fast path: the kernel derives the vertex
        ids of renderer/utils.py:76-80 from the face number and reads no face buffer), else one
        (F,3) int32 face list shared by the batch.  `faces=None` or a tensor tagged by
        gan2shape_amd.renderer.utils.get_face_idx skips the comparison."""
        if faces is None or getattr(faces, "_g2s_regular_grid", None) == S:
            if n_verts != S * S:
                raise ValueError("regular-grid faces need S*S vertices")
            return None
        if faces.dim() == 2:
            faces = faces.unsqueeze(0)
        F = faces.shape[1]
        if n_verts == S * S and F == 2 * (S - 1) * (S - 1):
            ref = _regular_grid_faces(S, faces.device)
            if bool((faces.int() == ref.unsqueeze(0)).all()):
                return None
        f0 = faces[0].int().contiguous()
        if faces.shape[0] > 1 and not bool((faces.int() == f0.unsqueeze(0)).all()):
            raise NotImplementedError("per-sample face lists are not supported: call per sample")
        return f0

    def render_rgb(self, vertices, faces, textures, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        """[B, C, S, S] image of the textured mesh: rasterize with the constructor's near / far
        (renderer.py:51), read each winning face's texture cube trilinearly at perspective-corrected
        barycentric coordinates, background colour elsewhere, flip + 2x2 average.  Lighting: ambient
        only (GAN2Shape builds the renderer with light_intensity_ambient=1, directional=0)."""
        K = self.K if K is None else K
        R = self.R if R is None else R
        t = self.t if t is None else t
        orig_size = self.orig_size if orig_size is None else orig_size
        if self.light_intensity_directional != 0:
            raise NotImplementedError("directional lighting is not supported (GAN2Shape uses ambient light only)")
        if K is None:
            raise ValueError("camera_mode='projection' needs K")
        _lib.require_cuda(vertices, textures)
        with torch.no_grad():
            vertices = vertices.detach().float()
            if self._needs_transform(R, t):
                if R is not None:
                    vertices = torch.matmul(vertices, R.reshape(-1, 3, 3).transpose(2, 1))
                if t is not None:
                    vertices = vertices + t.reshape(-1, 1, 3)
            verts = vertices.contiguous()
            B, N, _ = verts.shape
            S = self.image_size
            f = self._shared_faces(faces, N, S)
            F = 2 * (S - 1) * (S - 1) if f is None else f.shape[0]
            tex = textures.detach().float().contiguous()
            if tex.dim() != 6 or tex.shape[0] != B or tex.shape[1] != F or not (tex.shape[2] == tex.shape[3] == tex.shape[4]):
                raise ValueError(f"textures must be [B={B}, F={F}, T, T, T, C], got {tuple(tex.shape)}")
            ts, C = tex.shape[2], tex.shape[5]
            if self.light_intensity_ambient != 1:
                tex = tex * float(self.light_intensity_ambient)
            ssaa = 2 if self.anti_aliasing else 1
            L = _lib.load()
            depth = torch.empty((B, S, S), dtype=torch.float32, device=verts.device)
            fidx = torch.empty((B, S * ssaa, S * ssaa), dtype=torch.int32, device=verts.device)
            bary = torch.empty((B, S * ssaa, S * ssaa, 3), dtype=torch.float32, device=verts.device)
            ws = _workspace(verts.device, L.g2s_raster_workspace_bytes(B, N, F, S))
            Kc = (_lib.C.c_float * 9)(*self._host_K(K))
            _lib.check(L.g2s_raster_depth_fwd(_lib.ptr(verts), _lib.ptr(f), B, N, F, S, Kc, float(orig_size), ssaa,
                                              int(bool(self.fill_back)), float(self.near), float(self.far),
                                              _lib.ptr(depth), _lib.ptr(fidx), _lib.ptr(bary), _lib.ptr(ws),
                                              ws.numel(), _lib.stream()))
            bg = [float(v) for v in self.background_color][:C]
            bg += [bg[-1]] * (C - len(bg))
            rgb = torch.empty((B, C, S, S), dtype=torch.float32, device=verts.device)
            _lib.check(L.g2s_raster_rgb_fwd(_lib.ptr(verts), _lib.ptr(f), _lib.ptr(fidx), _lib.ptr(bary), _lib.ptr(tex),
                                            B, N, F, S, ssaa, ts, C, (_lib.C.c_float * C)(*bg),
                                            float(self.rasterizer_eps), _lib.ptr(rgb), _lib.stream()))
        return rgb

    def render(self, *a, **k):
        raise NotImplementedError("render / render_silhouettes (alpha channel, silhouette gradients) are not part "
                                  "of the boundary GAN2Shape uses (renderer.py:120,196)")

    render_silhouettes = render
