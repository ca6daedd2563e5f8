"""Rasterizer micro-benchmark at the BASELINE size (S = 128, ssaa 2): forward (no grad), forward
with saved maps, backward; B = 1 and 8; GAN2Shape-like scenes (tests/raster_cases.scene)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gan2shape_amd
from gan2shape_amd.plugins import neural_renderer as nr
from raster_cases import scene
from tools.bench_modconv import timeit
S = 128
for B in (1, 8):
    geo, verts, faces = scene(S, B=B, seed=1)
    K = tuple(np.asarray(geo.K[0], np.float32).reshape(9).tolist())
    v = torch.tensor(verts, device="cuda")
    vg = v.clone().requires_grad_(True)
    f = lambda x: nr.RenderDepthFunction.apply(x, None, K, float(S), S, True, True, 0.1, 100.0)
    t_f = timeit(lambda: f(v), 20)
    d = f(vg); g = torch.randn_like(d) * (d < 1.2)
    t_fg = timeit(lambda: f(vg), 20)
    t_b = timeit(lambda: torch.autograd.grad(d, vg, g, retain_graph=True), 20)
    tests = B * (2 * S) ** 2 * 4 * (S - 1) ** 2
    cov = float((d < 50).float().mean())
    print(f"B={B}: fwd {t_f:7.1f} us  fwd+maps {t_fg:7.1f} us  bwd {t_b:7.1f} us | coverage {cov:.2f} | "
          f"brute-force equivalent {tests/t_f/1e6:.1f} T tests/s | algorithmic bytes fwd {B*262144/t_f/1e3:.2f} GB/s")
