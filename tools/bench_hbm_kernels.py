"""Achieved HBM rate of the memory-bound kernels on their largest call of the face128_n8 workload
(algorithmic bytes / HIP-event time; 8 TB/s peak, ~6.3 TB/s achievable)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd  # noqa
from gan2shape_amd.plugins import fused, upfirdn2d_op
from gan2shape_amd.op.fused_act import fused_noise_bias_act
from gan2shape_amd.op.groupnorm import groupnorm_act
from gan2shape_amd.modconv import rows_dot_scale
from gan2shape_amd.stylegan2 import make_kernel
from tools.bench_modconv import timeit


def report(name, nbytes, fn):
    t = timeit(fn, 30)
    print(f"{name:58s} {nbytes / 1e6:8.1f} MB {t:7.1f} us {nbytes / t / 1e6:5.2f} TB/s ({nbytes / t / 1e6 / 8 * 100:4.1f} % of 8 TB/s)", flush=True)


B = 8
k = make_kernel([1, 3, 3, 1]).cuda()
x = torch.randn(B * 128, 129, 129, 1, device="cuda")
report("upfirdn2d blur (8,128,129,129)->(8,128,128,128)", (x.numel() + B * 128 * 128 * 128) * 4,
       lambda: upfirdn2d_op.upfirdn2d(x, k, 1, 1, 1, 1, 1, 1, 1, 1))
x = torch.randn(B * 128, 128, 128, 1, device="cuda")
report("upfirdn2d D blur pad 2 (8,128,128,128)->(8,128,129,129)", (x.numel() + B * 128 * 129 * 129) * 4,
       lambda: upfirdn2d_op.upfirdn2d(x, k, 1, 1, 1, 1, 2, 2, 2, 2))
report("upfirdn2d down 2 (8,128,128,128)->(8,128,64,64)", (x.numel() + B * 128 * 64 * 64) * 4,
       lambda: upfirdn2d_op.upfirdn2d(x, k, 1, 1, 2, 2, 1, 1, 1, 1))
a = torch.randn(B, 128, 128, 128, device="cuda")
bias = torch.randn(128, device="cuda")
e = a.new_empty(0)
report("fused_bias_act fwd (8,128,128,128)", 2 * a.numel() * 4, lambda: fused.fused_bias_act(a, bias, e, 3, 0, 0.2, 2 ** 0.5))
out = fused.fused_bias_act(a, bias, e, 3, 0, 0.2, 2 ** 0.5)
report("fused_bias_act bwd (8,128,128,128)", 3 * a.numel() * 4, lambda: fused.fused_bias_act(a, e, out, 3, 1, 0.2, 2 ** 0.5))
noise = torch.randn(1, 1, 128, 128, device="cuda")
nw = torch.ones(1, device="cuda")
report("noise_bias_act (8,128,128,128)", 2 * a.numel() * 4, lambda: fused_noise_bias_act(a, noise, nw, bias))
b2 = torch.randn_like(a)
s = torch.randn(B, 128, device="cuda")
report("rows_dot_scale (8,128,128,128): dot + scaled copy", 3 * a.numel() * 4, lambda: rows_dot_scale(a, b2, s, None))
g = torch.randn(1, 32, 128, 128, device="cuda", requires_grad=True)
ga, be = torch.ones(32, device="cuda", requires_grad=True), torch.zeros(32, device="cuda", requires_grad=True)
report("groupnorm_act fwd (1,32,128,128) G=8 (2 launches)", 3 * g.numel() * 4, lambda: groupnorm_act(g, ga, be, 8, 1e-5, True, 0.0))
y = groupnorm_act(g, ga, be, 8, 1e-5, True, 0.0)
gy = torch.randn_like(y)
report("groupnorm_act bwd (1,32,128,128) G=8 (2 launches)", 7 * g.numel() * 4,
       lambda: torch.autograd.grad(y, g, gy, retain_graph=True))
