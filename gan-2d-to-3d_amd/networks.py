"""D / A / V / L / E networks of GAN2Shape (arxiv 2011.00844, tables 5-8), state-dict compatible
with GAN2Shape/networks.py:23-244 (`network.<idx>.weight` keys).  On the GPU the convolutions
(forward, data- and weight-gradient) and the GroupNorm + activation pairs run on libg2s.so
(run_fused); on the CPU the same module lists run as plain torch.nn.

PSPNet / BiSeNet / ResNet (networks.py:247-586) only feed the one-shot masking model and are out
of scope.  The debug gradient alerts (debug_grad_updates.py) are dropped; `debug` is accepted and
ignored.
"""
import torch
import torch.nn as nn


def run_fused(mods, x):
    """Run a list of modules on the GPU with the pairs libg2s.so fuses: Conv2d / ConvTranspose2d
    (+ ReLU / LeakyReLU) on the fp32-MFMA convolution (op/conv.py), GroupNorm + ReLU / LeakyReLU
    as one op (op/groupnorm.py).  Module list and parameters — and so the state dict — are those of
    the reference nets; results equal the op-by-op evaluation up to fp32 summation order."""
    from .op import conv as gconv
    from .op.groupnorm import groupnorm_act, supported as gn_supported
    mods = list(mods)
    arena = _pass_arena(mods, x, 1)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        slope = None
        if isinstance(nxt, nn.LeakyReLU):
            slope = nxt.negative_slope
        elif isinstance(nxt, nn.ReLU):
            slope = 0.0
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and gconv.supported(m, x):
            x = gconv.conv_module(m, x, slope, arena)
            i += 2 if slope is not None else 1
        elif isinstance(m, nn.GroupNorm) and slope is not None and gn_supported(x):
            x = groupnorm_act(x, m.weight, m.bias, m.num_groups, m.eps, True, slope)
            i += 2
        elif isinstance(m, nn.Sequential):
            x = run_fused(m, x)
            i += 1
        else:
            x = m(x)
            i += 1
    return x


def _pass_arena(mods, x, width):
    """One cleared arena for the convolution outputs of a pass (see PassArena), sized by walking the
    spatial sizes through the module list; `width` = 2 for a paired pass (twice the channels)."""
    from .op import conv as gconv
    total, (H, W) = 0, (x.shape[2], x.shape[3])
    for m in mods:
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            H, W = gconv.conv_out_hw(m, H, W)
            n = x.shape[0] * m.out_channels * width * H * W
            if n <= gconv.PassArena.LIMIT:
                total += gconv.PassArena._pad(n)
        elif isinstance(m, nn.Upsample):
            H, W = int(H * m.scale_factor), int(W * m.scale_factor)
        elif isinstance(m, (nn.AvgPool2d, nn.MaxPool2d)):
            H, W = H // 2, W // 2
        elif not isinstance(m, (nn.GroupNorm, nn.ReLU, nn.LeakyReLU, nn.Tanh, nn.Sigmoid)):
            break  # unknown shape rule: later outputs simply keep their own allocation
    return gconv.PassArena(x.device, total)


def pair_parameters(net_a, net_b):
    """Give every pair of equally shaped parameters of two structurally identical nets ONE storage
    (a's tensor, then b's): the two nets can then run as a single network with twice the channels
    and grouped convolutions (run_fused_pair), one launch per layer and direction instead of two.
    Parameters stay the nets' own (same names, shapes, values; the optimisers update them in place);
    call after the nets are on their device.  Returns the number of paired parameters."""
    import torch
    n = 0
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        if pa.shape != pb.shape or pa.dtype != pb.dtype or pa.device != pb.device:
            continue
        both = torch.cat([pa.data.reshape(-1), pb.data.reshape(-1)])
        pa.data = both[:pa.numel()].view(pa.shape)
        pb.data = both[pa.numel():].view(pb.shape)
        n += 1
    return n


class _SplitHalves(torch.autograd.Function):
    """x [B, 2c, H, W] -> (x[:, :c], x[:, c:]) as contiguous tensors; backward = ONE concatenation (two
    `.contiguous()` slices cost a zero-fill + copy each and an add in backward)."""

    @staticmethod
    def forward(ctx, x):
        c = x.shape[1] // 2
        return x[:, :c].contiguous(), x[:, c:].contiguous()

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None
        if ga is None:
            ga = torch.zeros_like(gb)
        if gb is None:
            gb = torch.zeros_like(ga)
        return torch.cat([ga, gb], 1)


def run_fused_pair(mods_a, mods_b, x, train_a=True, train_b=True):
    """Two structurally identical module lists on the SAME input as one pass: activations carry the
    channels of net a followed by those of net b, convolutions run as grouped launches
    (g2s_conv2d_grouped, groups = 2; the first layer is one plain convolution with both nets' filters),
    GroupNorm sees twice the groups.  From the first layer whose parameters differ in shape (the nets'
    heads: GAN2Shape/networks.py:53-76,144-167) the two halves continue on their own.  Per net the
    arithmetic is that of run_fused.  train_x = False: that net's parameters get no gradient (the
    reference's no_grad branches of step 1, model.py:99-131).  Returns (y_a, y_b)."""
    import torch
    from .op import conv as gconv
    from .op.groupnorm import groupnorm_act, supported as gn_supported
    mods_a, mods_b = list(mods_a), list(mods_b)

    def pv(pa, pb):
        return gconv.pair_view(pa if train_a else pa.detach(), pb if train_b else pb.detach())

    groups = 1      # 1 until the first convolution has produced the side-by-side activations
    arena = _pass_arena(mods_a, x, 2)
    i = 0
    while i < len(mods_a):
        ma, mb = mods_a[i], mods_b[i]
        nxt = mods_a[i + 1] if i + 1 < len(mods_a) else None
        slope = nxt.negative_slope if isinstance(nxt, nn.LeakyReLU) else 0.0 if isinstance(nxt, nn.ReLU) else None
        if isinstance(ma, (nn.Conv2d, nn.ConvTranspose2d)):
            if not (gconv.adjacent(ma.weight, mb.weight) and ma.bias is None and mb.bias is None
                    and gconv.supported(ma, x[:, :ma.in_channels])
                    and not (groups == 1 and isinstance(ma, nn.ConvTranspose2d))):
                break
            x = gconv.ConvFunction.apply(x, pv(ma.weight, mb.weight), None, ma.stride[0], ma.padding[0],
                                         isinstance(ma, nn.ConvTranspose2d), slope, arena, groups)
            groups = 2
            i += 2 if slope is not None else 1
        elif isinstance(ma, nn.GroupNorm) and slope is not None and groups == 2 and gn_supported(x) \
                and gconv.adjacent(ma.weight, mb.weight) and gconv.adjacent(ma.bias, mb.bias):
            x = groupnorm_act(x, pv(ma.weight, mb.weight), pv(ma.bias, mb.bias), 2 * ma.num_groups, ma.eps, True, slope)
            i += 2
        elif isinstance(ma, (nn.ReLU, nn.LeakyReLU, nn.Upsample, nn.Tanh)) and groups == 2:
            x = ma(x)
            i += 1
        else:
            break
    if groups == 1:
        xa = xb = x
    else:
        xa, xb = _SplitHalves.apply(x)
    def tail(mods, xin, train):
        if i >= len(mods):
            return xin if train else xin.detach()
        if train:
            return run_fused(mods[i:], xin)
        with torch.no_grad():      # a frozen net: no graph through its own head either
            return run_fused(mods[i:], xin)
    return tail(mods_a, xa, train_a), tail(mods_b, xb, train_b)


def forward_pair(net_a, net_b, x, train_a=True, train_b=True):
    """net_a(x), net_b(x) for two nets of the same class whose parameters were paired
    (pair_parameters): one pass with twice the channels on the GPU, the plain two calls otherwise."""
    if x.is_cuda and type(net_a).__mro__[1] is type(net_b).__mro__[1] and hasattr(net_a, "_finish"):
        ya, yb = run_fused_pair(net_a.network, net_b.network, x, train_a, train_b)
        return net_a._finish(ya, x), net_b._finish(yb, x)
    return net_a(x), net_b(x)


class Encoder(nn.Module):
    """ViewpointNet / LightingNet trunk (networks.py:23-50): stride-2 convolutions down to 4x4, a
    4x4 convolution to 1x1, a 1x1 head.  The reference hard-codes FIVE stride-2 layers, i.e.
    image_size 128 (any other size leaves no 1x1 map); here their number follows the size — log2(size)
    - 2, widths nf * 2^i capped at nf * 16 — which is the reference's network at 128 (same state
    dict) and extends it to the 64x64 / 256x256 configs (BASELINE configs 1 and 5)."""

    def __init__(self, cin, cout, size):
        super().__init__()
        if size < 16 or size & (size - 1):
            raise ValueError("image_size must be a power of two >= 16")
        nf = max(4096 // size, 16)
        downs = size.bit_length() - 3          # log2(size) - 2
        chans = [cin] + [min(nf * 2 ** i, nf * 16) for i in range(downs)]
        layers = []
        for a, b in zip(chans[:-1], chans[1:]):
            layers += [nn.Conv2d(a, b, kernel_size=4, stride=2, padding=1, bias=False),
                       nn.ReLU(inplace=True)]
        top = chans[-1]
        layers += [nn.Conv2d(top, top, kernel_size=4, stride=1, padding=0, bias=False),
                   nn.ReLU(inplace=True),
                   nn.Conv2d(top, cout, kernel_size=1, stride=1, padding=0, bias=False),
                   nn.Tanh()]
        self.network = nn.Sequential(*layers)

    @staticmethod
    def _finish(out, input):
        return out.reshape(input.size(0), -1)

    def forward(self, input):
        out = run_fused(self.network, input) if input.is_cuda else self.network(input)
        return self._finish(out, input)


class ViewpointNet(Encoder):
    def __init__(self, image_size, debug=False):
        super().__init__(cin=3, cout=6, size=image_size)


class LightingNet(Encoder):
    def __init__(self, image_size, debug=False):
        super().__init__(cin=3, cout=4, size=image_size)


class EncoderDecoder(nn.Module):
    """DepthNet / AlbedoNet trunk (networks.py:79-141)."""

    def __init__(self, cin, cout, size, activation, zdim=256):
        super().__init__()
        nf = max(4096 // size, 16)
        g = 8 if size >= 128 else 16

        def down(a, b, groups):
            layers = [nn.Conv2d(a, b, kernel_size=4, stride=2, padding=1, bias=False)]
            if groups:
                layers.append(nn.GroupNorm(groups, b))
            return layers + [nn.LeakyReLU(0.2, inplace=True)]

        def up(a, b, groups):
            return [nn.ConvTranspose2d(a, b, kernel_size=4, stride=2, padding=1, bias=False),
                    nn.GroupNorm(groups, b), nn.ReLU(inplace=True),
                    nn.Conv2d(b, b, kernel_size=3, stride=1, padding=1, bias=False),
                    nn.GroupNorm(groups, b), nn.ReLU(inplace=True)]

        network = down(cin, nf, g) + down(nf, nf * 2, g * 2) + down(nf * 2, nf * 4, g * 4) + \
            down(nf * 4, nf * 8, 0)
        network += [nn.Conv2d(nf * 8, zdim, kernel_size=4, stride=1, padding=0, bias=False),
                    nn.ReLU(inplace=True),
                    nn.ConvTranspose2d(zdim, nf * 8, kernel_size=4, stride=1, padding=0, bias=False),
                    nn.ReLU(inplace=True),
                    nn.Conv2d(nf * 8, nf * 8, kernel_size=3, stride=1, padding=1, bias=False),
                    nn.ReLU(inplace=True)]
        network += up(nf * 8, nf * 4, g * 4) + up(nf * 4, nf * 2, g * 2) + up(nf * 2, nf, g)
        network += [nn.Upsample(scale_factor=2, mode='nearest'),
                    nn.Conv2d(nf, nf, kernel_size=3, stride=1, padding=1, bias=False),
                    nn.GroupNorm(g, nf), nn.ReLU(inplace=True),
                    nn.Conv2d(nf, nf, kernel_size=5, stride=1, padding=2, bias=False),
                    nn.GroupNorm(g, nf), nn.ReLU(inplace=True),
                    nn.Conv2d(nf, cout, kernel_size=5, stride=1, padding=2, bias=False)]
        if activation is not None:
            network += [activation()]
        self.network = nn.Sequential(*network)

    @staticmethod
    def _finish(out, input):
        return out

    def forward(self, input):
        return run_fused(self.network, input) if input.is_cuda else self.network(input)


class DepthNet(EncoderDecoder):
    def __init__(self, image_size, debug=False):
        super().__init__(cin=3, cout=1, size=image_size, activation=None)


class AlbedoNet(EncoderDecoder):
    def __init__(self, image_size, debug=False):
        super().__init__(cin=3, cout=3, size=image_size, activation=nn.Tanh)


class _ResSplit(torch.autograd.Function):
    """x -> (relu(x), avg_pool2d(x, 2, 2)): the two branch entries of ResBlock in one launch; backward joins
    the two gradients in one launch (g2s_res_split_*)."""

    @staticmethod
    def forward(ctx, x):
        from . import lib as _lib
        x = x.contiguous()
        B, C, H, W = x.shape
        r = torch.empty_like(x)
        p = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        _lib.check(_lib.load().g2s_res_split_fwd(_lib.ptr(x), _lib.ptr(r), _lib.ptr(p), B * C, H, W, _lib.stream()))
        ctx.save_for_backward(x)
        return r, p

    @staticmethod
    def backward(ctx, gr, gp):
        from . import lib as _lib
        (x,) = ctx.saved_tensors
        if gr is None and gp is None:
            return None
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        _lib.check(_lib.load().g2s_res_split_bwd(_lib.ptr(x), _lib.ptr(None if gr is None else gr.contiguous()),
                                                 _lib.ptr(None if gp is None else gp.contiguous()), _lib.ptr(gx), B * C, H, W,
                                                 _lib.stream()))
        return gx


class ResBlock(nn.Module):
    """Residual block of the offset encoder (networks.py:170-194)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.res_path = nn.Sequential(
            nn.ReLU(), nn.Conv2d(cin, cout, kernel_size=3, stride=2, padding=1),
            nn.ReLU(), nn.Conv2d(cout, cout, kernel_size=3, stride=1, padding=1))
        self.identity_path = nn.Sequential(
            nn.AvgPool2d(stride=2, kernel_size=2),
            nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0))

    def forward(self, x):
        if x.is_cuda:
            if x.dtype == torch.float32 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 \
                    and isinstance(self.res_path[0], nn.ReLU) and isinstance(self.identity_path[0], nn.AvgPool2d):
                r, p = _ResSplit.apply(x)          # the leading ReLU and the 2x2 average in one pass over x
                return run_fused(self.identity_path[1:], p) + run_fused(self.res_path[1:], r)
            return run_fused(self.identity_path, x) + run_fused(self.res_path, x)
        return self.identity_path(x) + self.res_path(x)


class OffsetEncoder(nn.Module):
    """GAN latent-offset encoder E (networks.py:197-243).  image_size 64: the reference passes
    `cout/2` (a float) to nn.Conv2d and cannot be constructed (networks.py:231); here the 64 branch
    emits `cout` channels (documented deviation, plumbing config only)."""

    def __init__(self, image_size=128, cin=3, cout=512, activation=None, debug=False):
        super().__init__()
        # the reference asserts image_size in [64, 128] (networks.py:205-206); 256 is this package's
        # extension for BASELINE config 5: one more ResBlock (8x8 -> 4x4) at constant width
        assert image_size in [64, 128, 256]
        nf = 16
        network = [nn.Conv2d(cin, 2 * nf, kernel_size=4, stride=2, padding=1), nn.ReLU(),
                   ResBlock(2 * nf, 4 * nf), ResBlock(4 * nf, 8 * nf), ResBlock(8 * nf, 16 * nf)]
        if image_size >= 128:
            network += [ResBlock(16 * nf, 32 * nf)]
            if image_size == 256:
                network += [ResBlock(32 * nf, 32 * nf)]
            network += [nn.Conv2d(32 * nf, 64 * nf, kernel_size=4, stride=1, padding=0), nn.ReLU(),
                        nn.Conv2d(64 * nf, cout, kernel_size=1, stride=1, padding=0)]
        else:
            network += [nn.Conv2d(16 * nf, 32 * nf, kernel_size=4, stride=1, padding=0), nn.ReLU(),
                        nn.Conv2d(32 * nf, cout, kernel_size=1, stride=1, padding=0)]
        if activation is not None:
            network += [activation()]
        self.network = nn.Sequential(*network)

    def forward(self, x):
        out = run_fused(self.network, x) if x.is_cuda else self.network(x)
        return out.reshape(x.size(0), -1)
