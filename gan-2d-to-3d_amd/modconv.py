"""Autograd wrapper of g2s_modconv (libg2s.so) — the grouped convolution of
ModulatedConv2d.forward (GAN2Shape/stylegan2/stylegan2-pytorch/model.py:250-291) in the
input-scaling formulation

    y[b,o] = demod[b,o] * sum_{i,t} w[o,i,t] * (s[b,i] * x[b,i,.])

Gradients (all on the GPU, no CPU path):
    gx  = s      * conv^T(w, demod * gy)            one more MFMA GEMM (transpose=1, same weights)
    gs  = sum_hw x * conv^T(w, demod * gy)          reduction of the same GEMM output
    gd  = sum_hw gy * y / demod
    gw  (only if the weight requires grad — never on GAN2Shape's path, where G is frozen:
         GAN2Shape/model.py:26-37 keeps G in eval mode and no optimiser owns it) via torch's conv
         weight-gradient.
"""
import os

import torch
from torch.autograd import Function

from . import lib as _lib
from . import zeropool as _zp

PLAIN, UP2, DOWN2 = _lib.CONV_PLAIN, _lib.CONV_UP2, _lib.CONV_DOWN2

# bench.py sets this to a list to collect (algorithmic FLOP, start event, end event, algorithmic
# bytes) per launch of the MFMA convolution kernel, recorded on the launch stream; None = no
# instrumentation.
PROFILE = None


class profiled:
    """with profiled(flop, nbytes[, mfma_flop]): <one launch of the MFMA convolution kernels>.
    `flop` is the ALGORITHMIC count of the direct convolution; `mfma_flop` what the matrix cores
    execute (smaller for the Winograd kernels: 16 multiplications per 4 outputs (F(2x2)) or 36 per 16 (F(4x4))
    instead of 36 per 4 / 144 per 16)."""

    def __init__(self, flop, nbytes, mfma_flop=None, kernel=None):
        self.rec = PROFILE
        self.flop, self.nbytes = flop, nbytes
        self.kernel = kernel or ("direct" if mfma_flop is None else "winograd")
        self.mfma_flop = flop if mfma_flop is None else mfma_flop

    def __enter__(self):
        if self.rec is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.rec is not None:
            self.e1.record()
            self.rec.append((self.flop, self.e0, self.e1, self.nbytes, self.mfma_flop, self.kernel))
        return False


def out_size(h, k, mode):
    if mode == PLAIN:
        return h
    if mode == UP2:
        return (h - 1) * 2 + k
    return (h - k) // 2 + 1


# Operand precision of the FROZEN networks' GEMMs (generator, discriminator, VGG):
#   "f32"  exact fp32 MFMA (v_mfma_f32_32x32x2_f32) — the reference's arithmetic, the headline workload;
#   "f16"  fp16 operands, fp32 accumulation (g2s_modconv_f16) — BASELINE config 5 only; tensors stay
#          fp32 in memory, results differ at the 1e-3 level.  Set by GAN2Shape(config['mfma_operands']).
OPERANDS = "f32"


def _f16_launch(x, w, si, so, bias, mode, transpose, act, alpha, gain, y):
    B, _, H, W = x.shape
    Cout, Cin, k, _ = w.shape
    L = _lib.load()
    sp = y.shape[2] * y.shape[3] if mode == PLAIN else min(H * W, y.shape[2] * y.shape[3])
    with profiled(2.0 * B * Cout * Cin * k * k * sp, 4.0 * (x.numel() + w.numel() + y.numel())):
        _lib.check(L.g2s_modconv_f16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(si), _lib.ptr(so), _lib.ptr(bias),
                                     _lib.ptr(y), B, Cin, Cout, H, W, k, mode, int(transpose), int(act),
                                     float(alpha), float(gain), _lib.stream()))
    return y


# Winograd F(2x2, 3x3) for the stride-1 3x3 layers (csrc/winograd.hip).  WINOGRAD = False forces the
# direct implicit GEMM everywhere; layers with fewer than WINO_MIN_TILES 2x2 output tiles (too few
# workgroups for the 64-channel x 64-tile blocks) stay on the direct kernel.
WINOGRAD = True
WINO_MIN_TILES = 512
WINO_FORCE = None   # tests / tuning tools: "direct", or a splitk value (0 = library choice) for every 3x3 stride-1 call
_WINO_U = {}  # (data_ptr, version, shape, transpose) -> (w, U): transformed weights of constant tensors

try:  # measured choices for the workload's own call signatures (tools/tune_wino.py)
    from .wino_tuned import TABLE as _WINO_TABLE
except ImportError:
    _WINO_TABLE = {}


def wino_choice(x, w, mode, transpose, fused):
    """None: direct implicit GEMM; an int: the `splitk` argument of g2s_conv3x3_wino (F(2x2)); "w4:k": the F(4x4)
    kernel with splitk = k."""
    B, Cr, H, W = x.shape
    if not (WINOGRAD and mode == PLAIN and w.shape[2] == 3 and w.shape[3] == 3 and not w.requires_grad
            and H >= 2 and W >= 2):
        return None
    M = w.shape[1] if transpose else w.shape[0]

    def checked(choice):   # "w4:k" only where the F(4x4) kernel takes the shape; else F(2x2), library partition
        if isinstance(choice, str) and choice.startswith("w4"):
            return choice if wino4_supported(B, Cr, M, H, W) else 0
        return None if choice == "direct" else int(choice)

    if WINO_FORCE is not None:
        return checked(WINO_FORCE)
    hit = _WINO_TABLE.get((B, Cr, M, H, W, int(transpose), int(fused)))
    if hit is None:     # the row measured without / with the epilogue
        hit = _WINO_TABLE.get((B, Cr, M, H, W, int(transpose), 1 - int(fused)))
    if hit is not None:
        return checked(hit)
    # unmeasured signature: Winograd when there is enough work for the 64-channel x 64-tile
    # workgroups (>= 2 rounds of whole tiles, or stream-K runs of >= 32 K tiles)
    tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
    blocks = -(-tiles // 64) * -(-M // 64)
    if tiles < WINO_MIN_TILES or blocks * -(-Cr // 4) < 256 * 32 or M <= 32:   # M <= 32: the direct 32x128 tile
        return None
    return 0   # F(4x4) only where the table measured it ahead: on unmeasured signatures it lost as often as it won


def wino_eligible(x, w, mode, H, W):
    return wino_choice(x, w, mode, 0, 0) is not None


def wino_weights(w, transpose):
    """U = G g G^T of a constant weight tensor [Cout, Cin, 3, 3] in the library's tiled layout,
    computed once per (tensor, version, direction)."""
    key = (w.data_ptr(), w._version, tuple(w.shape), int(transpose))
    hit = _WINO_U.get(key)
    if hit is None:
        if len(_WINO_U) > 256:
            _WINO_U.clear()
        Cout, Cin = w.shape[:2]
        L = _lib.load()
        M, Cr = (Cin, Cout) if transpose else (Cout, Cin)
        U = torch.empty(L.g2s_wino_weights_floats(M, Cr), dtype=torch.float32, device=w.device)
        _lib.check(L.g2s_wino_weights(_lib.ptr(w), _lib.ptr(U), Cout, Cin, int(transpose), _lib.stream()))
        hit = _WINO_U[key] = (w, U)   # holds `w`: its data_ptr stays unique while cached
    return hit[1]


# Winograd F(4x4, 3x3) (csrc/winograd4.hip) for the large maps: 1.78x fewer MFMA operations than F(2x2).  A choice
# "w4:k" (WINO_FORCE / the tuned table: wino_choice) sends the launch there with splitk = k (0: library choice);
# unmeasured signatures stay on F(2x2) (car128_joint, whose batch sizes are not in the table, ran 8 % slower with
# a fill-based default).  WINO4 = False keeps every Winograd layer on F(2x2).
WINO4 = os.environ.get("G2S_WINO4", "1") != "0"   # G2S_WINO4=0: measurement runs without the F(4x4) kernel
_WINO4_U = {}


def wino4_supported(B, Cr, M, H, W):
    return WINO4 and _lib.load().g2s_wino4_supported(B, Cr, M, H, W) == 1


def wino4_weights(w, transpose):
    """U = G g G^T (6x6 per channel pair) of a constant weight tensor in the F(4x4) kernel's tiled layout."""
    key = (w.data_ptr(), w._version, tuple(w.shape), int(transpose))
    hit = _WINO4_U.get(key)
    if hit is None:
        if len(_WINO4_U) > 256:
            _WINO4_U.clear()
        Cout, Cin = w.shape[:2]
        L = _lib.load()
        M, Cr = (Cin, Cout) if transpose else (Cout, Cin)
        U = torch.empty(L.g2s_wino4_weights_floats(M, Cr), dtype=torch.float32, device=w.device)
        _lib.check(L.g2s_wino4_weights(_lib.ptr(w), _lib.ptr(U), Cout, Cin, int(transpose), _lib.stream()))
        hit = _WINO4_U[key] = (w, U)
    return hit[1]


def _wino_launch(x, w, in_scale, out_scale, bias, transpose, act, alpha, gain, y, splitk=0, noise=None, noise_w=None):
    """One stride-1 3x3 launch on a Winograd kernel: `splitk` is wino_choice's answer — "w4:k" = F(4x4), an int =
    F(2x2) with that partition.  noise / noise_w: the StyledConv tail (then bias and act = 1 are required)."""
    B, Cr, H, W = x.shape
    M = y.shape[1]
    L = _lib.load()
    flop, nbytes = 2.0 * B * M * Cr * 9 * H * W, 4.0 * (x.numel() + w.numel() + y.numel())
    if isinstance(splitk, str):   # "w4:k"
        U = wino4_weights(w, transpose)
        with profiled(flop, nbytes, 2.0 * 36 * B * (H // 4) * (W // 4) * M * Cr, "winograd4"):
            _lib.check(L.g2s_conv3x3_wino4(_lib.ptr(x), _lib.ptr(U), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                           _lib.ptr(bias), _lib.ptr(noise), _lib.ptr(noise_w), _lib.ptr(y), B, Cr, M, H, W,
                                           int(act), float(alpha), float(gain), int(splitk[3:] or 0), *_lib.split_ws(),
                                           _lib.stream()))
        return y
    U = wino_weights(w, transpose)
    tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
    with profiled(flop, nbytes, 2.0 * 16 * tiles * M * Cr):
        if noise is not None:
            _lib.check(L.g2s_conv3x3_wino_nba(_lib.ptr(x), _lib.ptr(U), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                              _lib.ptr(bias), _lib.ptr(noise), _lib.ptr(noise_w), _lib.ptr(y), B, Cr, M,
                                              H, W, float(alpha), float(gain), int(splitk), *_lib.split_ws(),
                                              _lib.stream()))
        else:
            _lib.check(L.g2s_conv3x3_wino(_lib.ptr(x), _lib.ptr(U), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                          _lib.ptr(bias), _lib.ptr(y), B, Cr, M, H, W, int(act), float(alpha),
                                          float(gain), int(splitk), *_lib.split_ws(), _lib.stream()))
    return y


def modconv_nba_raw(x, w, in_scale, out_scale, bias, noise, noise_w, slope, gain):
    """StyledConv (plain 3x3 or 1x1) with its whole tail in the convolution's epilogue (g2s_modconv_nba /
    g2s_conv3x3_wino_nba): gain * leaky_relu(out_scale * conv(in_scale * x) + noise_w * noise + bias, slope).
    fp32 kernels only."""
    _lib.require_cuda(x, w, in_scale, out_scale, bias, noise, noise_w)
    if OPERANDS != "f32" or x.dtype != torch.float32:
        raise RuntimeError("modconv_nba: fp32 kernels only")
    x, w = x.contiguous(), w.contiguous()
    B, C, H, W = x.shape
    Cout, Cin, k, _ = w.shape
    if C != Cin or tuple(noise.shape[-2:]) != (H, W) or noise.numel() != H * W:
        raise RuntimeError("modconv_nba: shape mismatch")
    si, so = in_scale.contiguous(), out_scale.contiguous()
    bias, noise, noise_w = bias.contiguous(), noise.contiguous(), noise_w.contiguous()
    L = _lib.load()
    choice = wino_choice(x, w, PLAIN, 0, 1)
    if choice is not None:
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        return _wino_launch(x, w, si, so, bias, 0, 1, slope, gain, y, choice, noise, noise_w)
    y = None
    if L.g2s_modconv_needs_zero(B, Cin, Cout, H, W, k, PLAIN, 0, 1, 1) == 1:
        y = _zp.take((B, Cout, H, W), x.device)
    zeroed = y is not None
    if y is None:
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    with profiled(2.0 * B * Cout * Cin * k * k * H * W, 4.0 * (x.numel() + w.numel() + y.numel())):
        _lib.check(L.g2s_modconv_nba(_lib.ptr(x), _lib.ptr(w), _lib.ptr(si), _lib.ptr(so), _lib.ptr(bias), _lib.ptr(noise),
                                     _lib.ptr(noise_w), _lib.ptr(y), B, Cin, Cout, H, W, k, PLAIN, 0, float(slope),
                                     float(gain), int(zeroed), _lib.stream()))
    return y


def modconv_raw(x, w, in_scale, out_scale, mode, transpose):
    """Direct call of g2s_modconv.  w is always [Cout, Cin, k, k]."""
    _lib.require_cuda(x, w, in_scale, out_scale)
    if x.dtype != torch.float32 or w.dtype != torch.float32:
        raise RuntimeError("modconv: float32 only")
    x = x.contiguous()
    w = w.contiguous()
    B, C, H, W = x.shape
    Cout, Cin, k, _ = w.shape
    if C != (Cout if transpose else Cin):
        raise RuntimeError(f"modconv: x has {C} channels, expected {Cout if transpose else Cin}")
    if transpose:
        # adjoint geometry: UP2^T gathers with stride 2, DOWN2^T scatters
        oh = {PLAIN: H, UP2: (H - k) // 2 + 1, DOWN2: (H - 1) * 2 + k}[mode]
        ow = {PLAIN: W, UP2: (W - k) // 2 + 1, DOWN2: (W - 1) * 2 + k}[mode]
        cy = Cin
    else:
        oh, ow = out_size(H, k, mode), out_size(W, k, mode)
        cy = Cout
    for name, t, c in (("in_scale", in_scale, C), ("out_scale", out_scale, cy)):
        if t is not None and (t.shape != (B, c) or t.dtype != torch.float32):
            raise RuntimeError(f"modconv: {name} must be float32 [{B}, {c}]")
    si = None if in_scale is None else in_scale.contiguous()
    so = None if out_scale is None else out_scale.contiguous()
    f16 = OPERANDS == "f16" and not w.requires_grad
    choice = None if f16 else wino_choice(x, w, mode, transpose, 0)
    if f16 or choice is not None:
        y = torch.empty((B, cy, oh, ow), dtype=torch.float32, device=x.device)
        if f16:
            return _f16_launch(x, w, si, so, None, mode, transpose, 0, 0.0, 1.0, y)
        return _wino_launch(x, w, si, so, None, transpose, 0, 0.0, 1.0, y, choice)
    return _direct_launch(x, w, si, so, None, mode, transpose, 0, 0.0, 1.0, (B, cy, oh, ow))


def _direct_launch(x, w, si, so, bias, mode, transpose, act, alpha, gain, out_shape):
    """The direct implicit-GEMM kernel (g2s_modconv_ex).  Outputs of launches that add partial sums
    (split-K slices, polyphase holes) come from the step's cleared pool when one is active."""
    B, _, H, W = x.shape
    Cout, Cin, k, _ = w.shape
    L = _lib.load()
    fused = bias is not None or act != 0
    y = None
    if L.g2s_modconv_needs_zero(B, Cin, Cout, H, W, k, mode, int(transpose), int(si is not None or so is not None),
                                int(fused)) == 1:
        y = _zp.take(out_shape, x.device)
    zeroed = y is not None
    if y is None:
        y = torch.empty(out_shape, dtype=torch.float32, device=x.device)
    # algorithmic FLOP: 2 * B * Cout * Cin * k^2 * (spatial positions of the un-strided side);
    # algorithmic bytes: each operand once
    oh, ow = out_shape[2], out_shape[3]
    sp = min(H * W, oh * ow) if mode != PLAIN else oh * ow
    with profiled(2.0 * B * Cout * Cin * k * k * sp, 4.0 * (x.numel() + w.numel() + y.numel())):
        _lib.check(L.g2s_modconv_ex(_lib.ptr(x), _lib.ptr(w), _lib.ptr(si), _lib.ptr(so), _lib.ptr(bias), _lib.ptr(y),
                                    B, Cin, Cout, H, W, k, mode, int(transpose), int(act), float(alpha), float(gain),
                                    int(zeroed), _lib.stream()))
    return y


def rows_dot_scale(a, b, s=None, inv=None, want_out=True, want_dot=True):
    """Fused row pass (g2s_rows_dot_scale) over [B, C, H, W] tensors viewed as [B*C, H*W]:
    returns (b * s[r], (sum_hw a * b) / inv[r]) — either may be skipped."""
    b = b.contiguous()
    B, C = b.shape[:2]
    n = b[0, 0].numel()
    a_ = None if a is None or not want_dot else a.contiguous()
    out = torch.empty_like(b) if want_out else None
    dot = torch.empty((B, C), dtype=torch.float32, device=b.device) if want_dot else None
    s_ = None if s is None else s.contiguous()
    inv_ = None if inv is None else inv.contiguous()
    L = _lib.load()
    _lib.check(L.g2s_rows_dot_scale(_lib.ptr(a_), _lib.ptr(b), _lib.ptr(s_), _lib.ptr(inv_),
                                    _lib.ptr(out), _lib.ptr(dot), B * C, n, _lib.stream()))
    return out, dot


class DemodFunction(Function):
    """demod[b,o] = rsqrt(sum_i wsq[o,i] s[b,i]^2 + eps) (model.py:254-258), gradient to s only
    (wsq belongs to the frozen generator)."""

    @staticmethod
    def forward(ctx, s, wsq, eps):
        s = s.contiguous()
        wsq = wsq.contiguous()
        B, Cin = s.shape
        Cout = wsq.shape[0]
        demod = torch.empty((B, Cout), dtype=torch.float32, device=s.device)
        L = _lib.load()
        _lib.check(L.g2s_demod_fwd(_lib.ptr(wsq), _lib.ptr(s), _lib.ptr(demod), B, Cin, Cout,
                                   float(eps), _lib.stream()))
        ctx.save_for_backward(s, wsq, demod)
        return demod

    @staticmethod
    def backward(ctx, gd):
        s, wsq, demod = ctx.saved_tensors
        B, Cin = s.shape
        gs = torch.empty_like(s)
        L = _lib.load()
        _lib.check(L.g2s_demod_bwd(_lib.ptr(wsq), _lib.ptr(s), _lib.ptr(demod),
                                   _lib.ptr(gd.contiguous()), _lib.ptr(gs), B, Cin, wsq.shape[0],
                                   _lib.stream()))
        return gs, None, None


def demodulation(s, wsq, eps=1e-8):
    """Fused kernel when wsq is a constant (frozen generator), torch ops otherwise."""
    if wsq.requires_grad:
        return torch.rsqrt(torch.nn.functional.linear(s * s, wsq) + eps)
    _lib.require_cuda(s, wsq)
    return DemodFunction.apply(s, wsq, eps)


class ModConvFunction(Function):
    @staticmethod
    def forward(ctx, x, w, s, demod, mode):
        y = modconv_raw(x, w, s, demod, mode, 0)
        ctx.mode = mode
        ctx.has_s, ctx.has_d = s is not None, demod is not None
        need_y = demod is not None and ctx.needs_input_grad[3]
        ctx.save_for_backward(x, w, s, demod, y if need_y else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, s, demod, y = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gs = gd = None
        if ctx.needs_input_grad[0] or (ctx.has_s and ctx.needs_input_grad[2]):
            gxs = modconv_raw(gy, w, demod, None, ctx.mode, 1)  # gradient w.r.t. (s * x)
            if ctx.has_s:
                # one pass over (x, gxs): gs = sum_hw x * gxs and gx = gxs * s
                gx, gs = rows_dot_scale(x, gxs, s, None, want_out=ctx.needs_input_grad[0],
                                        want_dot=ctx.needs_input_grad[2])
            else:
                gx = gxs
        if ctx.has_d and ctx.needs_input_grad[3]:
            _, gd = rows_dot_scale(gy, y, None, demod, want_out=False, want_dot=True)
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(x, w, s, demod, gy, ctx.mode)
        return gx, gw, gs, gd, None


class ModConvDemodFunction(Function):
    """demodulation + modulated convolution as ONE autograd node (the style then receives one gradient:
    the demodulation path's, with the convolution's in_scale path added inside g2s_demod_bwd_add —
    as two nodes autograd spends an accumulation launch per layer).  wsq: constant (frozen generator)."""

    @staticmethod
    def forward(ctx, x, w, s, wsq, eps, mode):
        s, wsq = s.contiguous(), wsq.contiguous()
        B, Cin = s.shape
        demod = torch.empty((B, wsq.shape[0]), dtype=torch.float32, device=s.device)
        L = _lib.load()
        _lib.check(L.g2s_demod_fwd(_lib.ptr(wsq), _lib.ptr(s), _lib.ptr(demod), B, Cin, wsq.shape[0], float(eps),
                                   _lib.stream()))
        y = modconv_raw(x, w, s, demod, mode, 0)
        ctx.mode = mode
        ctx.save_for_backward(x, w, s, wsq, demod, y if ctx.needs_input_grad[2] else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, s, wsq, demod, y = ctx.saved_tensors
        gy = gy.contiguous()
        need_x, need_w, need_s = ctx.needs_input_grad[:3]
        gx = gw = gs = None
        if need_x or need_s:
            gxs = modconv_raw(gy, w, demod, None, ctx.mode, 1)  # gradient w.r.t. (s * x)
            # one pass over (x, gxs): gs = sum_hw x * gxs and gx = gxs * s
            gx, gs = rows_dot_scale(x, gxs, s, None, want_out=need_x, want_dot=need_s)
        if need_s:
            _, gd = rows_dot_scale(gy, y, None, demod, want_out=False, want_dot=True)
            B, Cin = s.shape
            _lib.check(_lib.load().g2s_demod_bwd_add(_lib.ptr(wsq), _lib.ptr(s), _lib.ptr(demod), _lib.ptr(gd),
                                                     _lib.ptr(gs), _lib.ptr(gs), B, Cin, wsq.shape[0], _lib.stream()))
        if need_w:
            gw = _weight_grad(x, w, s, demod, gy, ctx.mode)
        return gx, gw, gs, None, None, None


def modconv_demod(x, w, s, wsq, eps=1e-8, mode=PLAIN):
    """modconv(x, w, s, demodulation(s, wsq, eps), mode); one node when wsq is a constant."""
    if wsq.requires_grad:
        return modconv(x, w, s, demodulation(s, wsq, eps), mode)
    _lib.require_cuda(s, wsq)
    return ModConvDemodFunction.apply(x, w, s, wsq, eps, mode)


def _weight_grad(x, w, s, demod, gy, mode):
    """Off the hot path (G is frozen in GAN2Shape): weight gradient through torch's conv."""
    import torch.nn.functional as F
    xs = x if s is None else x * s[:, :, None, None]
    g = gy if demod is None else gy * demod[:, :, None, None]
    with torch.enable_grad():
        wv = w.detach().requires_grad_(True)
        k = w.shape[2]
        if mode == PLAIN:
            out = F.conv2d(xs, wv, padding=k // 2)
        elif mode == UP2:
            out = F.conv_transpose2d(xs, wv.transpose(0, 1), stride=2)
        else:
            out = F.conv2d(xs, wv, stride=2)
        (gw,) = torch.autograd.grad(out, wv, g)
    return gw


def modconv(x, w, s=None, demod=None, mode=PLAIN):
    """x [B,Cin,H,W]; w [Cout,Cin,k,k] (scaled); s [B,Cin] or None; demod [B,Cout] or None."""
    return ModConvFunction.apply(x, w, s, demod, mode)


class PlainConvFunction(Function):
    """Un-modulated convolution on the same MFMA kernel (in_scale = out_scale = NULL): the
    discriminator's EqualConv2d layers (stylegan2-pytorch/model.py:94-123,630-697) — 3x3 / 1x1,
    stride 1 with padding k//2, or stride 2 without padding.  Data-gradient on the GPU kernel;
    weight-gradient (never needed for the frozen D) through torch."""

    @staticmethod
    def forward(ctx, x, w, mode):
        ctx.mode = mode
        ctx.save_for_backward(x if ctx.needs_input_grad[1] else None, w)
        return modconv_raw(x, w, None, None, mode, 0)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = modconv_raw(gy.contiguous(), w, None, None, ctx.mode, 1)
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(x, w, None, None, gy, ctx.mode)
        return gx, gw, None


def conv2d_supported(x, w, stride, padding):
    k = w.shape[2]
    if not (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.shape[2] == w.shape[3]
            and k in (1, 3)):
        return None
    if stride == 1 and padding == k // 2:
        return PLAIN
    if stride == 2 and padding == 0 and (x.shape[2] - k) % 2 == 0 and (x.shape[3] - k) % 2 == 0:
        return DOWN2
    return None


def conv2d(x, w, mode):
    return PlainConvFunction.apply(x, w, mode)


def conv_bias_act_raw(x, w, bias, mode=PLAIN, alpha=0.0, gain=1.0):
    """gain * leaky_relu(conv(x, w) + bias, alpha) as ONE launch, outside autograd (Winograd / direct /
    fp16-operand kernel by the same choice as everywhere)."""
    x, w, bias = x.contiguous(), w.contiguous(), bias.contiguous()
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    oh, ow = out_size(H, k, mode), out_size(W, k, mode)
    choice = None if OPERANDS == "f16" else wino_choice(x, w, mode, 0, 1)
    if OPERANDS == "f16" or choice is not None:
        y = torch.empty((B, Cout, oh, ow), dtype=torch.float32, device=x.device)
        if OPERANDS == "f16":
            _f16_launch(x, w, None, None, bias, mode, 0, 1, alpha, gain, y)
        else:
            _wino_launch(x, w, None, None, bias, 0, 1, alpha, gain, y, choice)
        return y
    return _direct_launch(x, w, None, None, bias, mode, 0, 1, alpha, gain, (B, Cout, oh, ow))


def relu_gate(g, y, alpha=0.0, gain=1.0):
    """Backward of y = gain * leaky_relu(., alpha) from the saved OUTPUT: g * (y > 0 ? gain : alpha * gain)."""
    from .plugins import fused
    return fused.fused_bias_act(g.contiguous(), g.new_empty(0), y, 3, 1, alpha, gain)


class ConvBiasActFunction(Function):
    """gain * leaky_relu(conv(x, w) + bias, alpha) for a FROZEN network (no weight / bias gradient):
    the VGG16 trunk of LPIPS (conv3x3 + bias + ReLU, lpips/pretrained_networks.py:97-135) and the
    discriminator's ConvLayer (EqualConv2d + FusedLeakyReLU, stylegan2-pytorch/model.py:630-676).
    Forward is one launch (g2s_conv_bias_act); backward = activation slope from the saved output
    (g2s_fused_bias_act, act 3 / grad 1) + the data-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, w, bias, mode, alpha, gain):
        y = conv_bias_act_raw(x, w, bias, mode, alpha, gain)
        ctx.save_for_backward(w, y)
        ctx.cfg = (mode, float(alpha), float(gain))
        return y

    @staticmethod
    def backward(ctx, gy):
        w, y = ctx.saved_tensors
        mode, alpha, gain = ctx.cfg
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None
        from .plugins import fused
        g = fused.fused_bias_act(gy.contiguous(), gy.new_empty(0), y, 3, 1, alpha, gain)
        return modconv_raw(g, w, None, None, mode, 1), None, None, None, None, None


def conv_bias_act(x, w, bias, mode=PLAIN, alpha=0.0, gain=1.0):
    return ConvBiasActFunction.apply(x, w, bias, mode, alpha, gain)


def conv_bias_relu(x, w, bias):
    return ConvBiasActFunction.apply(x, w, bias, PLAIN, 0.0, 1.0)
