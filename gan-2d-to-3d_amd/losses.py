"""The three losses of the step (behaviour of GAN2Shape/losses.py:6-79): discriminator-feature L1,
masked photometric L1, second-order smoothness.  On the GPU the smoothness loss is one fused
kernel per direction (fused_geometry.smooth_loss)."""
import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import zeropool


def _weighted_mean(err, weight):
    """sum(err * w) / sum(w) with w broadcast to err; plain mean when w is None."""
    if weight is None:
        return err.mean()
    weight = weight.expand_as(err)
    return (err * weight).sum() / weight.sum()


class _WeightedL1(torch.autograd.Function):
    """sum(|x - y| * w.expand_as(x)) / sum(w.expand_as(x)) for x, y (B, C, H, W) and a mask w
    (B, 1, H, W) — one kernel per direction (g2s_weighted_l1_fwd / _bwd) instead of sub, abs, expand,
    mul and two full-size reductions.  Gradient w.r.t. x only (y is the detached real branch / the
    target image, w a detached mask)."""

    @staticmethod
    def forward(ctx, x, y, w):
        from . import lib as _lib
        x, y = x.contiguous(), y.contiguous()
        B, C, H, W = x.shape
        numden = zeropool.zeros(2, x.device)   # numerator, mask.expand_as(err).sum()
        wc = None if w is None else w.contiguous()
        _lib.check(_lib.load().g2s_weighted_l1_fwd2(_lib.ptr(x), _lib.ptr(y), _lib.ptr(wc), _lib.ptr(numden), B, C,
                                                    H * W, _lib.stream()))
        ctx.save_for_backward(x, y, wc, numden)
        return numden[0] / numden[1]

    @staticmethod
    def backward(ctx, g):
        from . import lib as _lib
        x, y, wc, numden = ctx.saved_tensors
        gx = torch.empty_like(x)
        B, C, H, W = x.shape
        _lib.check(_lib.load().g2s_weighted_l1_bwd2(_lib.ptr(x), _lib.ptr(y), _lib.ptr(wc), _lib.ptr(g.contiguous()),
                                                    _lib.ptr(numden[1:]), None, _lib.ptr(gx), B, C, H * W, _lib.stream()))
        return gx, None, None


def _masked_l1(a, b, weight):
    """(|a - b| * w).sum() / w.sum() with w (B,1,H,W) or (B,C,H,W)-expandable; fused on the GPU."""
    if (a.is_cuda and a.dtype == torch.float32 and a.dim() == 4 and a.shape == b.shape and not b.requires_grad
            and (a.shape[2] * a.shape[3]) % 4 == 0
            and (weight is None or (weight.dim() == 4 and weight.shape[1] == 1 and weight.shape[0] == a.shape[0]
                                    and weight.shape[2:] == a.shape[2:] and not weight.requires_grad))):
        return _WeightedL1.apply(a, b, weight)
    return _weighted_mean((a - b).abs(), weight)


class _DFeatureL1(torch.autograd.Function):
    """The whole discriminator-feature loss as ONE autograd node (frozen D, gradient-free real branch):

    forward   fake and real images go through the first `count` + 1 blocks of D TOGETHER as one batch of
              2N (13 convolution launches instead of 26 at size 128, 8 blurs instead of 16, and twice the
              work per launch for the few-tile 16^2 / 8^2 layers), then the masked feature L1 of every
              level on the two halves (g2s_weighted_l1_fwd);
    backward  hand-written over the FAKE half only (the first N samples of every saved activation are a
              contiguous view): per ResBlock (stylegan2-pytorch/model.py:679-697) the leaky-ReLU gates
              from the saved outputs, the stride-2 / stride-1 data-gradient GEMMs, the blurs' adjoints,
              and the residual join (a + b) / sqrt(2) as one launch.
    Same arithmetic as DiscriminatorLoss.__call__'s op-by-op form below."""

    FUSED_LEVELS = True        # join + masked-L1 gradient + conv2's gate as one launch per level (g2s_weighted_l1_bwd3)

    @staticmethod
    def forward(ctx, fake, real, D, count, weights):
        from . import lib as _lib
        from .op import add_bias_scale
        N = fake.shape[0]
        x = torch.cat([fake, real], 0)
        L = _lib.load()
        first = D.convs[0]
        y0 = first(x)
        blocks, out = [], y0
        numden = zeropool.zeros((count, 2), x.device)   # per level: numerator, sum of weights
        for i in range(1, count + 1):
            blk = D.convs[i]
            in_hw = out.shape[2:]
            y1 = blk.conv1(out)
            y2 = blk.conv2(y1)
            sk = blk.skip(out)
            out = add_bias_scale(y2, sk, None, 2 ** -0.5)
            B2, C, H, W = out.shape
            w = weights[i - 1]
            wc = None if w is None else w.contiguous()
            _lib.check(L.g2s_weighted_l1_fwd2(_lib.ptr(out[:N]), _lib.ptr(out[N:]), _lib.ptr(wc), _lib.ptr(numden[i - 1]),
                                              N, C, H * W, _lib.stream()))
            blocks.append((blk, in_hw, y1, y2, out, wc))
        ctx.D, ctx.N, ctx.first, ctx.y0 = D, N, first, y0
        ctx.blocks, ctx.numden = blocks, numden
        return (numden[:, 0] / numden[:, 1]).sum()

    @staticmethod
    @once_differentiable        # the backward runs raw kernels on saved activations: no double backward
    def backward(ctx, g_total):
        from . import lib as _lib
        from .modconv import DOWN2, PLAIN, modconv_raw, relu_gate
        from .op import add_bias_scale
        from .op.upfirdn2d import upfirdn2d_adjoint
        L = _lib.load()
        N = ctx.N
        slope, gain = 0.2, 2 ** 0.5
        g_total = g_total.contiguous()

        def level_grad(feat, wc, den, add, ref):
            """(gx, gx * gate(ref)) of one launch (g2s_weighted_l1_bwd3): this level's masked-L1 gradient w.r.t. the fake
            half + the residual join (a + b) / sqrt(2) of the block above; the gated copy feeds the block's conv2."""
            shape = ref.shape
            gx = torch.empty(shape, dtype=torch.float32, device=ref.device) if feat is not None else None
            gg = torch.empty(shape, dtype=torch.float32, device=ref.device)
            a0, a1 = (None, None) if add is None else add
            _lib.check(L.g2s_weighted_l1_bwd3(
                _lib.ptr(None if feat is None else feat[:N]), _lib.ptr(None if feat is None else feat[N:]), _lib.ptr(wc),
                _lib.ptr(g_total if feat is not None else None), _lib.ptr(den), _lib.ptr(a0), _lib.ptr(a1), 2 ** -0.5,
                _lib.ptr(gx), _lib.ptr(ref), slope, gain, _lib.ptr(gg), shape[0], shape[1], shape[2] * shape[3], _lib.stream()))
            return gx, gg

        add = None          # (gx_main, gx_skip) of the block above: joined inside the next level's launch
        for level in range(len(ctx.blocks) - 1, -1, -1):
            blk, in_hw, y1, y2, feat, wc = ctx.blocks[level]
            if _DFeatureL1.FUSED_LEVELS:
                g, g2 = level_grad(feat, wc, ctx.numden[level, 1:], add, y2[:N])
            else:
                joined = None if add is None else add_bias_scale(add[0], add[1], None, 2 ** -0.5)
                g = torch.empty(y2[:N].shape, dtype=torch.float32, device=feat.device)
                _lib.check(L.g2s_weighted_l1_bwd2(_lib.ptr(feat[:N]), _lib.ptr(feat[N:]), _lib.ptr(wc), _lib.ptr(g_total),
                                                  _lib.ptr(ctx.numden[level, 1:]), _lib.ptr(joined), _lib.ptr(g), N,
                                                  feat.shape[1], feat.shape[2] * feat.shape[3], _lib.stream()))
                g2 = relu_gate(g, y2[:N], slope, gain)
            # out = (conv2(conv1(x)) + skip(x)) / sqrt(2): both branches keep the common factor, the join applies it
            conv2, conv1, skip = blk.conv2, blk.conv1, blk.skip
            w2 = conv2[-2]._w.get(conv2[-2].weight, conv2[-2].scale)
            w1 = conv1[-2]._w.get(conv1[-2].weight, conv1[-2].scale)
            ws = skip[-1]._w.get(skip[-1].weight, skip[-1].scale)
            gb = modconv_raw(g2, w2, None, None, DOWN2, 1)                       # w.r.t. the blurred conv1 output
            blur2 = conv2[0]
            # (conv1's gate in the store of the Blur's adjoint — a g2s_upfirdn2d_gate — was built and measured 0.03 ms
            # SLOWER per step 2 than the separate gate launch: the per-output reference load stalls the blur's store
            # phase; removed again)
            g1 = relu_gate(upfirdn2d_adjoint(gb, blur2.kernel, 1, blur2.down, blur2.pad, y1.shape[2:]), y1[:N], slope, gain)
            gx_main = modconv_raw(g1, w1, None, None, PLAIN, 1)
            gs = modconv_raw(g, ws, None, None, PLAIN, 1)                        # 1x1 on the blur-downsampled input
            blurs = skip[0]
            gx_skip = upfirdn2d_adjoint(gs, blurs.kernel, 1, blurs.down, blurs.pad, in_hw)
            add = (gx_main, gx_skip)
        first = ctx.first
        w0 = first[-2]._w.get(first[-2].weight, first[-2].scale)
        if _DFeatureL1.FUSED_LEVELS:
            _, g0 = level_grad(None, None, None, add, ctx.y0[:N].contiguous())      # the last join + the first layer's gate
        else:
            g0 = relu_gate(add_bias_scale(add[0], add[1], None, 2 ** -0.5), ctx.y0[:N], slope, gain)
        gx = modconv_raw(g0, w0, None, None, PLAIN, 1)
        return gx, None, None, None, None


def _mask_pyramid(mask, levels):
    """[avg_pool2d(mask, 2, 2), avg_pool2d of that, ...]: one launch (g2s_avg_pyramid) for up to 4 levels."""
    if mask is None:
        return [None] * levels
    if not (1 <= levels <= 4 and mask.is_cuda and mask.dtype == torch.float32):
        out, w = [], mask
        for _ in range(levels):
            w = F.avg_pool2d(w, 2, 2)
            out.append(w)
        return out
    from . import lib as _lib
    m = mask.contiguous()
    B, C, H, W = m.shape
    out = [torch.empty((B, C, H >> (l + 1), W >> (l + 1)), dtype=torch.float32, device=m.device) for l in range(levels)]
    ptrs = (_lib.C.c_void_p * levels)(*[t.data_ptr() for t in out])
    _lib.check(_lib.load().g2s_avg_pyramid(_lib.ptr(m), ptrs, levels, B * C, H, W, _lib.stream()))
    return out


def _d_one_node_ok(D, fake, real, count):
    from .stylegan2 import ConvLayer, ResBlock
    from .op import FusedLeakyReLU
    if not (DiscriminatorLoss.ONE_NODE and isinstance(D, torch.nn.Module) and hasattr(D, "convs")
            and fake.is_cuda and fake.dtype == torch.float32 and fake.dim() == 4 and fake.shape == real.shape
            and not real.requires_grad and 1 <= count < len(D.convs)
            and not any(p.requires_grad for p in D.parameters())):
        return False
    first = D.convs[0]
    if not (isinstance(first, ConvLayer) and len(first) == 2 and isinstance(first[-1], FusedLeakyReLU)):
        return False
    for blk in D.convs[1:count + 1]:
        if not (isinstance(blk, ResBlock) and len(blk.conv1) == 2 and len(blk.conv2) == 3 and len(blk.skip) == 2
                and isinstance(blk.conv1[-1], FusedLeakyReLU) and isinstance(blk.conv2[-1], FusedLeakyReLU)
                and blk.conv1[-2].weight.shape[2] == 3 and blk.conv2[-2].weight.shape[2] == 3
                and blk.skip[-1].weight.shape[2] == 1):
            return False
    return True


class DiscriminatorLoss():
    """Sum over the first `ftr_num` discriminator feature maps of the L1 distance between the
    features of `fake_img` and `real_img`; a mask is box-averaged down to each feature resolution
    and used as the weight (losses.py:6-36).  The real branch runs without autograd."""

    ONE_NODE = True   # fake and real through D as one batch, hand-written backward (_DFeatureL1)

    def __init__(self, ftr_num=4, data_parallel=False):
        if data_parallel:
            raise NotImplementedError("data_parallel is disabled in the reference (model.py:82)")
        self.ftr_num = ftr_num

    def _features(self, D, image):
        return D(image, self.ftr_num)[1]

    def __call__(self, D, fake_img, real_img, mask=None):
        if self.ftr_num is not None and _d_one_node_ok(D, fake_img, real_img.detach(), self.ftr_num) and (
                mask is None or (mask.dim() == 4 and mask.shape[1] == 1 and mask.shape[0] == fake_img.shape[0]
                                 and mask.shape[2:] == fake_img.shape[2:] and not mask.requires_grad
                                 and mask.shape[2] % (2 ** self.ftr_num) == 0 and mask.shape[3] % (2 ** self.ftr_num) == 0)):
            weights = _mask_pyramid(mask, self.ftr_num)   # level l lives at 1 / 2^l of the image: 2x2 box averages
            return _DFeatureL1.apply(fake_img, real_img.detach(), D, self.ftr_num, weights)
        with torch.no_grad():
            real = self._features(D, real_img.detach())
        fake = self._features(D, fake_img)
        count = len(fake) if self.ftr_num is None else self.ftr_num
        total = 0
        pooled = {}     # mask averaged down to a feature resolution, each level from the previous one
        for f_fake, f_real in zip(fake[:count], real[:count]):
            weight = None
            if mask is not None:
                step = (mask.shape[2] // f_fake.shape[2], mask.shape[3] // f_fake.shape[3])
                weight = pooled.get(step)
                if weight is None:
                    # box average over step x step (losses.py:27-31); a power-of-two step is reached by
                    # 2x2 averages of the previous level (same value, sums re-associated)
                    prev = pooled.get((step[0] // 2, step[1] // 2)) if step[0] % 2 == 0 and step[1] % 2 == 0 else None
                    weight = F.avg_pool2d(prev, 2, 2) if prev is not None else \
                        F.avg_pool2d(mask, kernel_size=step, stride=step)
                    pooled[step] = weight
            total = total + _masked_l1(f_fake, f_real, weight)
        return total


class PhotometricLoss():
    """L1 between two images, optionally confidence-weighted (sigma) and mask-averaged
    (losses.py:39-51)."""
    EPS = 1e-7

    def __call__(self, image1, image2, mask=None, conf_sigma=None):
        if conf_sigma is None:
            return _masked_l1(image1, image2, mask)
        err = (image1 - image2).abs()
        sigma = conf_sigma + self.EPS
        err = err * 2 ** 0.5 / sigma + sigma.log()
        return _weighted_mean(err, mask)


class SmoothLoss():
    """mean|d2/dx2| + mean|d2/dxdy| + mean|d2/dydx| + mean|d2/dy2| of a (N,H,W) / (N,C,H,W) map;
    a list of maps is a pyramid weighted 1, 1/2.3, 1/2.3^2, ... (losses.py:54-79)."""

    @staticmethod
    def gradient(pred):
        """Forward differences along x and y: (D_dx, D_dy)."""
        planes = pred.reshape(-1, pred.shape[-2], pred.shape[-1]) if pred.dim() == 4 else pred
        return torch.diff(planes, dim=2), torch.diff(planes, dim=1)

    def _one(self, pred):
        first = self.gradient(pred)
        second = [d for g in first for d in self.gradient(g)]     # dx2, dxdy, dydx, dy2
        return sum(d.abs().mean() for d in second)

    def __call__(self, pred_map):
        if torch.is_tensor(pred_map) and pred_map.is_cuda and pred_map.dtype == torch.float32 \
                and pred_map.dim() in (3, 4):
            from .fused_geometry import smooth_loss  # one kernel forward, one backward
            return smooth_loss(pred_map)
        pyramid = pred_map if isinstance(pred_map, (tuple, list)) else [pred_map]
        loss, weight = 0, 1
        for level in pyramid:
            loss = loss + self._one(level) * weight
            weight /= 2.3
        return loss
