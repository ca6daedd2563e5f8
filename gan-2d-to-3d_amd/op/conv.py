"""nn.Conv2d / nn.ConvTranspose2d of the trained nets (GAN2Shape/networks.py:23-244) on libg2s.so:
forward and data-gradient on the fp32-MFMA implicit-GEMM kernel (g2s_conv2d), weight gradient on
the pixel-reduction GEMM (g2s_conv2d_wgrad), optional fused (leaky-)ReLU.  One launch per
direction instead of MIOpen's solver + layout-transpose + workspace-fill sequences at batch 1.
No native fallback."""
import torch
from torch.autograd import Function

from gan2shape_amd import lib as _lib
from gan2shape_amd import zeropool


class PassArena:
    """Zero-filled scratch for the convolution outputs of ONE pass through a small net (forward
    outputs; data- and weight-gradients of its backward).  The split-K paths of g2s_conv2d /
    g2s_conv2d_wgrad add partial sums into a cleared output: instead of one clear per launch
    (~100 tiny fill kernels per training iteration of the depth / albedo / viewpoint / lighting
    nets) the pass clears one arena and hands out slices.  Slices are ordinary views of a freshly
    allocated tensor (owned by the autograd graph of this pass), so nothing outlives its data; a
    request that does not fit gets its own tensor and the library's own clear."""

    LIMIT = 1 << 20  # elements: larger outputs keep their own allocation

    def __init__(self, device, fwd_elems):
        self.device = device
        # inside a training step the arena is a slice of the step's cleared pool (zeropool.py): no fill of its own
        self.fwd = zeropool.zeros(fwd_elems, device) if fwd_elems else None
        self.fwd_off = 0
        self.bwd_need = 0
        self.bwd = None
        self.bwd_off = 0

    @staticmethod
    def _pad(n):
        return (n + 63) & ~63

    def take_fwd(self, shape):
        n = 1
        for d in shape:
            n *= d
        if self.fwd is None or n > self.LIMIT or self.fwd_off + n > self.fwd.numel():
            return None
        v = self.fwd[self.fwd_off:self.fwd_off + n].view(shape)
        self.fwd_off += self._pad(n)
        return v

    def reserve_bwd(self, n):
        if n <= self.LIMIT:
            self.bwd_need += self._pad(n)

    def take_bwd(self, shape):
        n = 1
        for d in shape:
            n *= d
        if n > self.LIMIT or self.bwd_need == 0:
            return None
        if self.bwd is None:  # first gradient of this pass: one clear for all of them
            self.bwd = zeropool.zeros(self.bwd_need, self.device)
        if self.bwd_off + n > self.bwd.numel():
            return None       # e.g. a second backward through the same graph
        v = self.bwd[self.bwd_off:self.bwd_off + n].view(shape)
        self.bwd_off += self._pad(n)
        return v


def conv_out_hw(mod, H, W):
    k, s, p = mod.kernel_size[0], mod.stride[0], mod.padding[0]
    if isinstance(mod, torch.nn.ConvTranspose2d):
        return (H - 1) * s - 2 * p + k, (W - 1) * s - 2 * p + k
    return (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1


def _conv2d_raw(x, w, bias, Cr, M, k, stride, pad, adjoint, m_major, out_hw, act, slope, out=None, groups=1):
    """Cr -> M channels PER GROUP; x has groups * Cr channels, y groups * M (g2s_conv2d_grouped)."""
    B, _, H, W = x.shape
    if adjoint:
        oh, ow = out_hw if out_hw else ((H - 1) * stride - 2 * pad + k, (W - 1) * stride - 2 * pad + k)
    else:
        oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    zeroed = out is not None
    y = out if zeroed else torch.empty((B, groups * M, oh, ow), dtype=torch.float32, device=x.device)
    assert tuple(y.shape) == (B, groups * M, oh, ow) and x.shape[1] == groups * Cr
    L = _lib.load()
    from gan2shape_amd.modconv import profiled
    sp = H * W if adjoint else oh * ow  # every (input pixel, tap) pair of the strided side once
    with profiled(2.0 * B * groups * Cr * M * k * k * sp, 4.0 * (x.numel() + w.numel() + y.numel())):
        if groups == 1:
            _lib.check(L.g2s_conv2d(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), B, Cr, M, H, W, k,
                                    stride, pad, int(adjoint), int(m_major), oh if adjoint else 0,
                                    ow if adjoint else 0, 1 if act else 0, float(slope), 1.0, int(zeroed),
                                    _lib.stream()))
        else:
            _lib.check(L.g2s_conv2d_grouped(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), B, Cr, M, H, W,
                                            k, stride, pad, int(adjoint), int(m_major), oh if adjoint else 0,
                                            ow if adjoint else 0, 1 if act else 0, float(slope), 1.0,
                                            int(zeroed), groups, _lib.stream()))
    return y


def _wgrad(A, G, k, stride, pad, out=None, groups=1):
    """dw [groups * Ca, Cg, k, k] from A [B, groups * Ca, ...] and G [B, groups * Cg, ...]."""
    B, Ca, PH, PW = A.shape
    _, Cg, GH, GW = G.shape
    Ca, Cg = Ca // groups, Cg // groups
    zeroed = out is not None
    dw = out if zeroed else torch.empty((groups * Ca, Cg, k, k), dtype=torch.float32, device=A.device)
    L = _lib.load()
    if groups == 1:
        _lib.check(L.g2s_conv2d_wgrad(_lib.ptr(A), _lib.ptr(G), _lib.ptr(dw), B, Ca, Cg, PH, PW, GH, GW, k,
                                      stride, pad, int(zeroed), _lib.stream()))
    else:
        _lib.check(L.g2s_conv2d_wgrad_grouped(_lib.ptr(A), _lib.ptr(G), _lib.ptr(dw), B, Ca, Cg, PH, PW, GH, GW,
                                              k, stride, pad, int(zeroed), groups, _lib.stream()))
    return dw


def _conv_bwd_raw(gy, w, x, cin, cout, k, stride, pad, transposed, gx_out=None, gw_out=None, groups=1):
    """Data-gradient AND weight-gradient of one layer as one launch (g2s_conv2d_bwd): what
    _conv2d_raw(gy, w, ...) and _wgrad(...) compute, their workgroups dealt to the CUs together."""
    B, _, H, W = gy.shape
    oh, ow = x.shape[2], x.shape[3]
    adjoint = not transposed
    gx = gx_out if gx_out is not None else torch.empty((B, groups * cin, oh, ow), dtype=torch.float32, device=gy.device)
    A, G = (x, gy) if transposed else (gy, x)
    Ca, Cg = A.shape[1] // groups, G.shape[1] // groups
    dw = gw_out if gw_out is not None else torch.empty((groups * Ca, Cg, k, k), dtype=torch.float32, device=gy.device)
    L = _lib.load()
    from gan2shape_amd.modconv import profiled
    sp = H * W if adjoint else oh * ow
    flop = 2.0 * B * groups * cout * cin * k * k * sp + 2.0 * B * groups * Ca * Cg * k * k * A.shape[2] * A.shape[3]
    with profiled(flop, 4.0 * (2 * gy.numel() + w.numel() + gx.numel() + x.numel() + dw.numel())):
        _lib.check(L.g2s_conv2d_bwd(_lib.ptr(gy), _lib.ptr(w), _lib.ptr(gx), B, cout, cin, H, W, k, stride, pad,
                                    int(adjoint), int(transposed), oh if adjoint else 0, ow if adjoint else 0,
                                    int(gx_out is not None), _lib.ptr(A), _lib.ptr(G), _lib.ptr(dw), Ca, Cg,
                                    A.shape[2], A.shape[3], G.shape[2], G.shape[3], int(gw_out is not None),
                                    groups, _lib.stream()))
    return gx, dw


class ConvFunction(Function):
    """transposed = False: F.conv2d(x, w [Cout,Cin,k,k], bias, stride, pad); True:
    F.conv_transpose2d(x, w [Cin,Cout,k,k], bias, stride, pad); then leaky-ReLU(slope) if
    slope is not None (0 = ReLU).
    groups = 2: two such convolutions side by side (channels of x, rows of w and channels of y hold
    the two nets one after the other) in one launch per direction."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, transposed, slope, arena=None, groups=1):
        _lib.require_cuda(x, w, bias)
        if x.dtype != torch.float32 or w.dtype != torch.float32:
            raise RuntimeError("conv: float32 only")
        x, w = x.contiguous(), w.contiguous()
        k = w.shape[2]
        cin, cout = (w.shape[0] // groups, w.shape[1]) if transposed else (w.shape[1], w.shape[0] // groups)
        if x.shape[1] != groups * cin:
            raise RuntimeError(f"conv: input has {x.shape[1]} channels, weight expects {groups * cin}")
        if groups != 1 and bias is not None:
            raise RuntimeError("grouped conv: no bias")
        b = None if bias is None else bias.contiguous()
        out = None
        if arena is not None:
            H, W = x.shape[2], x.shape[3]
            if transposed:
                oh, ow = (H - 1) * stride - 2 * pad + k, (W - 1) * stride - 2 * pad + k
            else:
                oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
            out = arena.take_fwd((x.shape[0], groups * cout, oh, ow))
            if ctx.needs_input_grad[0]:
                arena.reserve_bwd(x.numel())
            if ctx.needs_input_grad[1]:
                arena.reserve_bwd(w.numel())
        y = _conv2d_raw(x, w, b, cin, cout, k, stride, pad, transposed, not transposed, None,
                        slope is not None, slope or 0.0, out=out, groups=groups)
        ctx.save_for_backward(x, w, y if slope is not None else None)
        ctx.cfg = (stride, pad, transposed, slope, bias is not None, groups)
        ctx.arena = arena
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, transposed, slope, has_bias, groups = ctx.cfg
        k = w.shape[2]
        gy = gy.contiguous()
        if slope is not None:  # gradient through the fused activation: slope taken from sign(y)
            from gan2shape_amd.plugins import fused
            gy = fused.fused_bias_act(gy, gy.new_empty(0), y, 3, 1, float(slope), 1.0)
        cin, cout = (w.shape[0] // groups, w.shape[1]) if transposed else (w.shape[1], w.shape[0] // groups)
        gx = gw = gb = None
        arena = ctx.arena
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            gx_out = None if arena is None else arena.take_bwd(tuple(x.shape))
            gw_out = None if arena is None else arena.take_bwd(tuple(w.shape))
            gx, gw = _conv_bwd_raw(gy, w, x, cin, cout, k, stride, pad, transposed, gx_out, gw_out, groups)
        elif ctx.needs_input_grad[0]:
            out = None if arena is None else arena.take_bwd(tuple(x.shape))
            gx = _conv2d_raw(gy, w, None, cout, cin, k, stride, pad, not transposed, transposed,
                             (x.shape[2], x.shape[3]), False, 0.0, out=out, groups=groups)
        elif ctx.needs_input_grad[1]:
            out = None if arena is None else arena.take_bwd(tuple(w.shape))
            gw = (_wgrad(x, gy, k, stride, pad, out, groups) if transposed
                  else _wgrad(gy, x, k, stride, pad, out, groups))
        if has_bias and ctx.needs_input_grad[2]:
            gb = torch.empty(gy.shape[1], dtype=torch.float32, device=gy.device)     # gy.sum((0, 2, 3)), one small launch
            _lib.check(_lib.load().g2s_channel_sum(_lib.ptr(gy), _lib.ptr(gb), gy.shape[0], gy.shape[1],
                                                   gy.shape[2] * gy.shape[3], _lib.stream()))
        return gx, gw, gb, None, None, None, None, None, None


def supported(mod, x):
    """True if `mod` (nn.Conv2d / nn.ConvTranspose2d) on input x can run on g2s_conv2d."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    k = mod.kernel_size
    return (k[0] == k[1] and 1 <= k[0] <= 5 and mod.stride[0] == mod.stride[1] and mod.stride[0] in (1, 2)
            and mod.padding[0] == mod.padding[1] and isinstance(mod.padding[0], int) and 0 <= mod.padding[0] < k[0]
            and mod.dilation == (1, 1) and mod.groups == 1 and mod.padding_mode == 'zeros'
            and getattr(mod, 'output_padding', (0, 0)) == (0, 0)
            and x.shape[2] + 2 * mod.padding[0] >= k[0] and x.shape[3] + 2 * mod.padding[0] >= k[0])


def conv_module(mod, x, slope=None, arena=None):
    """Run nn.Conv2d / nn.ConvTranspose2d `mod` (its parameters, stride, padding) on libg2s.so."""
    transposed = isinstance(mod, torch.nn.ConvTranspose2d)
    return ConvFunction.apply(x, mod.weight, mod.bias, mod.stride[0], mod.padding[0], transposed, slope,
                              arena)


# ------------------------------------------------------------------------------------- paired nets
class PairView(Function):
    """Two parameters of equal shape that live back to back in ONE storage (networks.pair_parameters)
    seen as a single tensor [2 * n0, ...]: what a grouped launch reads.  Gradients split back."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.n = a.numel()
        ctx.shape = a.shape
        return torch.as_strided(a.detach(), (2 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        n0 = ctx.shape[0]
        return g[:n0], g[n0:]


def adjacent(a, b):
    """True if parameter b starts where parameter a ends (same storage, same shape, contiguous)."""
    return (a.shape == b.shape and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel())


def pair_view(a, b):
    return PairView.apply(a, b)
