"""The frozen StyleGAN2 synthesis network as ONE autograd node (Generator.forward's layer loop,
GAN2Shape/stylegan2/stylegan2-pytorch/model.py:545-627 over StyledConv :321-355, ToRGB :358-377,
ModulatedConv2d :195-291), for GAN2Shape's step 2 (GAN2Shape/model.py:207-210: G.invert with the
offset encoder's latent) where the generator is frozen and only the per-layer modulated styles carry a
gradient.

What the reference — and this package's op-by-op path — runs per StyledConv is conv, `image + weight *
noise`, fused_bias_act, and in backward the gate, the data-gradient GEMM and three reductions, every one
a full pass over the activation in HBM.  Here:

  forward    plain layers: convolution with noise + bias + leaky-ReLU in its epilogue (g2s_modconv_nba /
             g2s_conv3x3_wino_nba) — the pre-activation is never stored; up-sampling layers: the
             transposed convolution, then the Blur with the same tail in its store (g2s_upfirdn2d_nba);
  backward   per activation x between two layers ONE row pass (g2s_synth_bwd_rows) reads x and the
             data-gradients of its consumers (next convolution, ToRGB) and produces the consumers' style
             gradients (sum x * g), the gated gradient for the producer (FusedLeakyReLU's backward with
             ref = x, op/fused_act.py:33-38) and the producer's demodulation gradient (its convolution
             output recovered from x: the leaky ReLU is invertible) — instead of rows_dot_scale twice, the
             accumulation of the two consumers' gradients, and the gate.

Same arithmetic as the op-by-op path up to fp32 summation order and the 1-ulp inversion of the activation
in the demodulation gradient; tests/test_gpu_round4.py holds image and style gradients to that path."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib as _lib
from .modconv import PLAIN, UP2, modconv_nba_raw, modconv_raw, rows_dot_scale
from .op.fused_act import add_bias_scale
from .op.upfirdn2d import upfirdn2d, upfirdn2d_adjoint


def _ptr_array(tensors):
    return (_lib.C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(values):
    return (_lib.C.c_int * len(values))(*[int(v) for v in values])


def _demod_all(wsqs, styles, eps):
    """demod[b,o] = rsqrt(sum_i wsq[o,i] s[b,i]^2 + eps) (model.py:254-258) of every styled layer: ONE launch."""
    B = styles[0].shape[0]
    demods = [torch.empty((B, w.shape[0]), dtype=torch.float32, device=w.device) for w in wsqs]
    _lib.check(_lib.load().g2s_demod_fwd_multi(_ptr_array(wsqs), _ptr_array(styles), _ptr_array(demods),
                                               _int_array([w.shape[1] for w in wsqs]), _int_array([w.shape[0] for w in wsqs]),
                                               len(wsqs), B, float(eps), _lib.stream()))
    return demods


def _demod_bwd_all(entries):
    """gs += (demodulation path) for every styled layer (entries: wsq, s, demod, gd, gs): ONE launch, in place."""
    wsqs = [e[0] for e in entries]
    B = entries[0][1].shape[0]
    _lib.check(_lib.load().g2s_demod_bwd_multi(_ptr_array(wsqs), _ptr_array([e[1] for e in entries]),
                                               _ptr_array([e[2] for e in entries]), _ptr_array([e[3] for e in entries]),
                                               _ptr_array([e[4] for e in entries]), _int_array([w.shape[1] for w in wsqs]),
                                               _int_array([w.shape[0] for w in wsqs]), len(entries), B, _lib.stream()))


def _blur_nba(yc, blur, act, noise, nw):
    """Blur (model.py:75-91, pad of ModulatedConv2d's up-sampling branch) + NoiseInjection + FusedLeakyReLU."""
    B, C, H, W = yc.shape
    k = blur.kernel
    p0, p1 = blur.pad
    oh, ow = H + p0 + p1 - k.shape[0] + 1, W + p0 + p1 - k.shape[1] + 1
    y = torch.empty((B, C, oh, ow), dtype=torch.float32, device=yc.device)
    _lib.check(_lib.load().g2s_upfirdn2d_nba(_lib.ptr(yc), _lib.ptr(k), _lib.ptr(y), B * C, C, H, W, k.shape[0], k.shape[1],
                                             1, 1, p0, p1, p0, p1, _lib.ptr(act.bias), _lib.ptr(noise), _lib.ptr(nw),
                                             float(act.negative_slope), float(act.scale), _lib.stream()))
    return y


def _rows(x, g1, s1, g2=None, s2=None, tail=None, demod=None, want_out=True):
    """g2s_synth_bwd_rows -> (out, dot1, dot2, gdot); `tail` = (noise, noise_w, bias, slope, gain) of the producer."""
    B, C, H, W = x.shape
    noise, nw, bias, slope, gain = tail
    out = torch.empty_like(x) if want_out else None
    dot1 = torch.empty((B, C), dtype=torch.float32, device=x.device)
    dot2 = torch.empty_like(dot1) if g2 is not None else None
    gdot = torch.empty_like(dot1) if demod is not None else None
    _lib.check(_lib.load().g2s_synth_bwd_rows(
        _lib.ptr(x), _lib.ptr(g1), _lib.ptr(s1), _lib.ptr(g2), _lib.ptr(s2),
        _lib.ptr(noise if demod is not None else None), _lib.ptr(nw if demod is not None else None),
        _lib.ptr(bias if demod is not None else None), _lib.ptr(demod), _lib.ptr(out), _lib.ptr(dot1), _lib.ptr(dot2),
        _lib.ptr(gdot), B * C, C, H * W, float(slope), float(gain), _lib.stream()))
    return out, dot1, dot2, gdot


def _tail(sc, noise):
    a = sc.activate
    return (noise.contiguous(), sc.noise.weight.detach(), a.bias.detach(), a.negative_slope, a.scale)


class _Synthesis(Function):
    """image = synthesis(x0; s_0 .. s_{L-1}); gradients to the modulated styles only (frozen generator)."""

    @staticmethod
    def forward(ctx, G, noise, x0, *styles):
        styles = [s.contiguous() for s in styles]
        layers = []     # per StyledConv: dict(kind, sc, w, wsq, s, demod, x, yc, tail)
        rgbs = []       # per ToRGB: dict(tr, w, s, x)
        n = 0
        # every styled layer's demodulation before the first convolution (the styles are all known): one launch
        mods = [m for m, _ in G._style_layers()]
        styled_pos = [i for i, m in enumerate(mods) if m.demodulate]
        weights = {i: mods[i]._weights() for i in styled_pos}
        eps = mods[styled_pos[0]].eps
        assert all(mods[i].eps == eps for i in styled_pos)
        demods = dict(zip(styled_pos, _demod_all([weights[i][1].contiguous() for i in styled_pos],
                                                 [styles[i] for i in styled_pos], eps)))

        def styled(sc, x, nz, up):
            nonlocal n
            s = styles[n]
            (w, wsq), demod = weights[n], demods[n]
            assert mods[n] is sc.conv
            n += 1
            tail = _tail(sc, nz)
            if up:
                yc = modconv_raw(x, w, s, demod, UP2, 0)
                y = _blur_nba(yc, sc.conv.blur, sc.activate, tail[0], tail[1])
            else:
                yc = None
                y = modconv_nba_raw(x, w, s, demod, tail[2], tail[0], tail[1], tail[3], tail[4])
            layers.append(dict(up=up, sc=sc, w=w, wsq=wsq.contiguous(), s=s, demod=demod, x=x, yc=yc, y=y, tail=tail))
            return y

        def to_rgb(tr, x, skip):
            nonlocal n
            s = styles[n]
            n += 1
            w, _ = tr.conv._weights()
            rgb = modconv_raw(x, w, s, None, PLAIN, 0)
            rgbs.append(dict(tr=tr, w=w, s=s, up=skip is not None, hw=None if skip is None else tuple(skip.shape[2:])))
            return add_bias_scale(rgb, None if skip is None else tr.upsample(skip), tr.bias.detach())

        out = styled(G.conv1, x0.contiguous(), noise[0], False)
        skip = to_rgb(G.to_rgb1, out, None)
        for conv1, conv2, nz1, nz2, tr in zip(G.convs[::2], G.convs[1::2], noise[1::2], noise[2::2], G.to_rgbs):
            out = styled(conv1, out, nz1, True)
            out = styled(conv2, out, nz2, False)
            skip = to_rgb(tr, out, skip)
        assert n == len(styles)
        ctx.layers, ctx.rgbs = layers, rgbs
        return skip

    @staticmethod
    @once_differentiable        # raw kernels on saved activations: no double backward
    def backward(ctx, g_img):
        layers, rgbs = ctx.layers, ctx.rgbs
        g_rgb = g_img.contiguous()
        nxt = None                       # (gxs, s, layer) of the up-sampling convolution above the current level
        order = []                       # (position in the style list, gradient)
        pending = []                     # (wsq, s, demod, gd, gs): the demodulation paths, added in ONE launch at the end
        # positions in the style list: conv1 0, to_rgb1 1, then (up, plain, rgb) triples
        li, ri = len(layers) - 1, len(rgbs) - 1
        while ri >= 0:
            T = rgbs[ri]
            plain = layers[li]                                   # the plain layer whose output feeds T (conv1 at level 0)
            a = plain['y']
            gxs_rgb = modconv_raw(g_rgb, T['w'], None, None, PLAIN, 1)
            pos_T = 1 if ri == 0 else 1 + 3 * ri
            if nxt is None:                                      # top level: ToRGB is the only consumer
                g_pre, dot_T, _, gdot = _rows(a, gxs_rgb, T['s'], None, None, plain['tail'], plain['demod'])
            else:
                g_pre, dot_U, dot_T, gdot = _rows(a, nxt[0], nxt[1], gxs_rgb, T['s'], plain['tail'], plain['demod'])
                U = nxt[2]
                order.append((nxt[3], dot_U))
                pending.append((U['wsq'], U['s'], U['demod'], U['gd'], dot_U))
            order.append((pos_T, dot_T))
            if T['up']:
                tr = T['tr']
                g_rgb = upfirdn2d_adjoint(g_rgb, tr.upsample.kernel, tr.upsample.factor, 1, tr.upsample.pad, T['hw'])
            # the plain layer: data-gradient, then the rows of ITS input
            gxs_P = modconv_raw(g_pre, plain['w'], plain['demod'], None, PLAIN, 1)
            pos_P = 0 if ri == 0 else 3 * ri
            if ri == 0:                                          # conv1 reads the constant input: style gradient only
                _, dot_P = rows_dot_scale(plain['x'], gxs_P, None, None, want_out=False, want_dot=True)
                order.append((pos_P, dot_P))
                pending.append((plain['wsq'], plain['s'], plain['demod'], gdot, dot_P))
                break
            up = layers[li - 1]
            g_pre_U, dot_P, _, _ = _rows(up['y'], gxs_P, plain['s'], None, None, up['tail'], None)
            order.append((pos_P, dot_P))
            pending.append((plain['wsq'], plain['s'], plain['demod'], gdot, dot_P))
            # the up-sampling layer: Blur's adjoint, demodulation gradient from the saved transposed-conv output
            blur = up['sc'].conv.blur
            g_yc = upfirdn2d_adjoint(g_pre_U, blur.kernel, 1, 1, blur.pad, tuple(up['yc'].shape[2:]))
            _, up['gd'] = rows_dot_scale(g_yc, up['yc'], None, up['demod'], want_out=False, want_dot=True)
            gxs_U = modconv_raw(g_yc, up['w'], up['demod'], None, UP2, 1)
            nxt = (gxs_U, up['s'], up, 3 * ri - 1)
            li -= 2
            ri -= 1
        _demod_bwd_all(pending)
        ctx.layers = ctx.rgbs = None
        out = [None] * (max(p for p, _ in order) + 1)
        for p, g in order:
            out[p] = g
        return (None, None, None) + tuple(out)


def eligible(G, x0, styles, noise):
    """The one-node path serves the frozen generator on the fp32 kernels with fixed noise maps."""
    from . import modconv as mc
    if mc.OPERANDS != "f32" or not x0.is_cuda or x0.dtype != torch.float32 or any(nz is None for nz in noise):
        return False
    if any(p.requires_grad for p in G.parameters()):
        return False
    return all(isinstance(s, torch.Tensor) and s.dim() == 2 and s.dtype == torch.float32 for s in styles)


def synthesize(G, noise, x0, styles):
    return _Synthesis.apply(G, list(noise), x0, *styles)
