"""-m gpu: every tile instance of g2s::modconv_kernel (128x128, 128x64, 64x64, 32x128, 64x128) and its split-K
paths on the north-star workload's OWN call signatures, value by value against the C oracle.

The heuristic picks the 128-wide tiles only for M > 64 with >= 512 workgroups and the measured
table (csrc/modconv_tuned.inc) only for B = 8 / 9 signatures, so small shapes never reach
modconv_kernel<128,128,18> / <128,64,18> — the instances that carry a third of the benchmarked
iteration (profiles/r01_c_bench_timed_region.txt).  Here each signature runs with
    * the built-in choice (tuned table / heuristic),
    * every tile in {0: 128x128, 1: 128x64, 2: 64x64, 3: 32x128, 4: 64x128} x split-K in {1, 4, 7} forced through
      g2s_modconv_tune,
forward (in-scale = style, out-scale = demodulation) and transposed (data-gradient: in-scale =
demodulation), against oracle.capi.modconv (stylegan2-pytorch/model.py:250-291 restated in C,
double accumulation).  Tolerance: rtol 2e-4, atol 2e-5 * max(1, sqrt(K / 1152)) (fp32 MFMA sums of
K = Cin * k^2 terms; the summation order is all that differs); where the dispatcher picks the F(4x4,3x3) Winograd
kernel: atol 1e-4 * max(1, sqrt(K / 1152)) (_check4: its transform constants amplify the rounding)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conv_cases import DOWN2, PLAIN, UP2, expected_modconv  # noqa: E402

TILES = (0, 1, 2, 3, 4)
SPLITS = (1, 4, 7)


@pytest.fixture(scope="module")
def L():
    import gan2shape_amd  # noqa: F401
    from gan2shape_amd import lib
    lib_ = lib.load()
    assert torch.cuda.is_available()
    yield lib_
    lib_.g2s_modconv_tune(-1, -1)


class direct_kernel:
    """Route the stride-1 3x3 layers to the direct implicit GEMM (they default to Winograd)."""

    def __enter__(self):
        from gan2shape_amd import modconv
        self.mc, self.saved = modconv, modconv.WINOGRAD
        modconv.WINOGRAD = False

    def __exit__(self, *exc):
        self.mc.WINOGRAD = self.saved


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def _check(y, exp, K, what):
    atol = 2e-5 * max(1.0, math.sqrt(K / 1152.0))
    got = y.cpu().numpy()
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    np.testing.assert_allclose(got, exp, rtol=2e-4, atol=atol, err_msg=what)


def _check_dispatched(y, exp, K, what, xd, wd, mode, transpose, fused):
    """The dispatcher's own choice for this signature: the F(4x4) Winograd kernel carries its own tolerance
    (_check4 below), every other kernel the one of the direct kernel."""
    from gan2shape_amd import modconv as mc
    choice = mc.wino_choice(xd, wd, mode, transpose, fused)
    if isinstance(choice, str):
        _check4(y, exp, K, f"{what} [{choice}]")
    else:
        _check(y, exp, K, f"{what} [{choice}]")


# (B, Cin, Cout, H, k, mode): the generator's / discriminator's own layers at B = 8 (SURVEY §8a
# layer table; D: stylegan2-pytorch/model.py:630-697 at size 128, cm = 1)
SIGNATURES = [
    (8, 128, 128, 128, 3, PLAIN),   # G convs[9]: the heaviest <128,128,18> launch
    (8, 512, 512, 32, 3, PLAIN),    # G convs[5]
    (8, 512, 256, 32, 3, UP2),      # G convs[6] (transposed stride 2: 4 residue classes)
    (8, 256, 512, 33, 3, DOWN2),    # D ResBlock conv2 after the blur (stride 2)
    (8, 128, 256, 129, 3, DOWN2),   # D first ResBlock conv2 (129 = 128 + blur padding)
    (8, 128, 256, 64, 1, PLAIN),    # D skip 1x1 after the stride-2 blur
    (8, 512, 512, 16, 3, PLAIN),    # tuned entry: tile 1, split-K 4
    (8, 512, 512, 17, 3, UP2),      # ragged class sizes (17 -> 35)
]


@pytest.mark.parametrize("transpose", [0, 1])
@pytest.mark.parametrize("B,cin,cout,H,k,mode", SIGNATURES)
def test_modconv_every_tile_vs_oracle(L, B, cin, cout, H, k, mode, transpose):
    from gan2shape_amd.modconv import modconv_raw
    rng = np.random.default_rng(B * 1000 + cin + cout + H + 7 * mode + transpose)
    w = (rng.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k)).astype(np.float32)
    if transpose:
        # the data-gradient call: x is the gradient w.r.t. the layer output
        oh = {PLAIN: H, UP2: (H - 1) * 2 + k, DOWN2: (H - k) // 2 + 1}[mode]
        x = rng.standard_normal((B, cout, oh, oh)).astype(np.float32)
        s_in = (1 + 0.3 * rng.standard_normal((B, cout))).astype(np.float32)   # demodulation
        s_out = None
    else:
        x = rng.standard_normal((B, cin, H, H)).astype(np.float32)
        s_in = (1 + 0.3 * rng.standard_normal((B, cin))).astype(np.float32)    # style
        s_out = (1 + 0.3 * rng.standard_normal((B, cout))).astype(np.float32)  # demodulation
    exp = expected_modconv(x, w, s_in, s_out, mode, transpose)
    xd, wd, sid = dev(x), dev(w), dev(s_in)
    sod = None if s_out is None else dev(s_out)
    K = (cout if transpose else cin) * k * k
    try:
        L.g2s_modconv_tune(-1, -1)
        _check_dispatched(modconv_raw(xd, wd, sid, sod, mode, transpose), exp, K,
                          "built-in choice (measured table: direct / Winograd)", xd, wd, mode, transpose, 0)
        with direct_kernel():
            _check(modconv_raw(xd, wd, sid, sod, mode, transpose), exp, K, "built-in direct choice")
            for tile in TILES:
                for sk in SPLITS:
                    assert L.g2s_modconv_tune(tile, sk) == 0
                    _check(modconv_raw(xd, wd, sid, sod, mode, transpose), exp, K, f"tile {tile} split-K {sk}")
    finally:
        L.g2s_modconv_tune(-1, -1)


# VGG16 trunk of LPIPS at B = 9 (image + 8 projected samples; lpips/pretrained_networks.py:97-135)
# and the discriminator's ConvLayer: conv + bias + (leaky-)ReLU in one launch, deferred to a second
# launch when split-K is on.
@pytest.mark.parametrize("B,cin,cout,H,alpha,gain", [
    (9, 512, 512, 16, 0.0, 1.0),            # tuned: tile 1, split-K 7 (deferred epilogue)
    (9, 64, 64, 128, 0.0, 1.0),             # conv1_2: N = 147456 pixels
    (9, 3, 64, 128, 0.0, 1.0),              # conv1_1: 3 input channels (partial K tile)
    (8, 128, 128, 128, 0.2, 2 ** 0.5),      # D ResBlock conv1 (FusedLeakyReLU epilogue)
])
def test_conv_bias_act_every_tile_vs_oracle(L, B, cin, cout, H, alpha, gain):
    from gan2shape_amd.modconv import conv_bias_act
    rng = np.random.default_rng(B + cin + cout + H)
    x = rng.standard_normal((B, cin, H, H)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    pre = expected_modconv(x, w, None, None, PLAIN, 0) + b[None, :, None, None]
    exp = (np.where(pre > 0, pre, pre * alpha) * gain).astype(np.float32)
    # an activation input within the tolerance of zero may take the other slope: compare there
    # with the absolute tolerance only (|slope difference| * |pre| <= atol)
    xd, wd, bd = dev(x), dev(w), dev(b)
    try:
        _check_dispatched(conv_bias_act(xd, wd, bd, PLAIN, alpha, gain), exp, cin * 9, "built-in choice (measured table)",
                          xd, wd, PLAIN, 0, 1)
        from gan2shape_amd import modconv as mc
        for force in (0, 1, 4, -256):   # Winograd with the fused / reduce-pass / deferred epilogue: library choice, whole tiles, split-K, stream-K
            mc.WINO_FORCE = force
            try:
                _check(conv_bias_act(xd, wd, bd, PLAIN, alpha, gain), exp, cin * 9, f"winograd partition {force}")
            finally:
                mc.WINO_FORCE = None
        with direct_kernel():
            for tile, sk in [(-1, -1)] + [(t, s) for t in TILES for s in SPLITS]:
                assert L.g2s_modconv_tune(tile, sk) == 0
                _check(conv_bias_act(xd, wd, bd, PLAIN, alpha, gain), exp, cin * 9, f"tile {tile} split-K {sk}")
    finally:
        L.g2s_modconv_tune(-1, -1)


def test_forced_tile_is_honoured_and_restored(L):
    """g2s_modconv_tune rejects bad arguments and (-1, -1) restores the built-in choice: same bits
    as before the override for a signature whose built-in choice has no split-K."""
    from gan2shape_amd.modconv import modconv_raw
    assert L.g2s_modconv_tune(5, 1) != 0 and L.g2s_modconv_tune(0, 0) != 0
    torch.manual_seed(0)
    x = torch.randn(8, 128, 64, 64, device="cuda")
    w = torch.randn(128, 128, 3, 3, device="cuda") / 34
    with direct_kernel():
        L.g2s_modconv_tune(-1, -1)
        y0 = modconv_raw(x, w, None, None, PLAIN, 0)
        L.g2s_modconv_tune(2, 1)
        y1 = modconv_raw(x, w, None, None, PLAIN, 0)
        L.g2s_modconv_tune(-1, -1)
        y2 = modconv_raw(x, w, None, None, PLAIN, 0)
    assert torch.equal(y0, y2)
    torch.testing.assert_close(y0, y1, rtol=1e-4, atol=1e-4)


# ----------------------------------------------------------------------------- Winograd F(2x2, 3x3)
WINO_CASES = [  # B, Cin, Cout, H, W
    (8, 128, 128, 128, 128),   # G convs[9]
    (8, 512, 512, 32, 32),     # G convs[5]
    (8, 512, 512, 16, 16),     # G convs[3]: 512 tiles, split-K by the library
    (9, 256, 512, 16, 16),     # VGG conv4_1 at B = 9
    (2, 64, 64, 128, 128),     # VGG conv1_2 of step 1 (B = 2)
    (2, 3, 64, 64, 64),        # 3 input channels: partial K tile
    (3, 70, 130, 23, 37),      # odd sizes, ragged channel counts, tiles spanning batch entries
    (1, 6, 5, 45, 31),         # fewer channels than a tile on both sides
]


@pytest.mark.parametrize("transpose", [0, 1])
@pytest.mark.parametrize("B,cin,cout,H,W", WINO_CASES)
def test_winograd_vs_oracle(L, B, cin, cout, H, W, transpose):
    """g2s_conv3x3_wino (transformed weights from g2s_wino_weights) against the direct-convolution
    oracle, forward and data-gradient form, every partition (library choice, whole tiles, split-K,
    stream-K), with and without the style / demodulation scales; same tolerance as the direct kernel."""
    from gan2shape_amd import modconv as mc
    rng = np.random.default_rng(B + cin + cout + H + W + transpose)
    w = (rng.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    cx, cy = (cout, cin) if transpose else (cin, cout)
    x = rng.standard_normal((B, cx, H, W)).astype(np.float32)
    s_in = (1 + 0.3 * rng.standard_normal((B, cx))).astype(np.float32)
    s_out = None if transpose else (1 + 0.3 * rng.standard_normal((B, cy))).astype(np.float32)
    exp = expected_modconv(x, w, s_in, s_out, PLAIN, transpose)
    exp_plain = expected_modconv(x, w, None, None, PLAIN, transpose)
    xd, wd, sid = dev(x), dev(w), dev(s_in)
    sod = None if s_out is None else dev(s_out)
    saved = (mc.WINOGRAD, mc.WINO_FORCE)
    try:
        mc.WINOGRAD = True
        for sk in (0, 1, 3, -7, -256):     # library choice, whole tiles, split-K 3, stream-K over 7 / 256 WGs
            mc.WINO_FORCE = sk
            assert mc.wino_choice(xd, wd, PLAIN, transpose, 0) == sk
            _check(mc.modconv_raw(xd, wd, sid, sod, PLAIN, transpose), exp, cx * 9, f"winograd partition {sk}")
        mc.WINO_FORCE = 0
        _check(mc.modconv_raw(xd, wd, None, None, PLAIN, transpose), exp_plain, cx * 9, "winograd, no scales")
        # split-K without a workspace (a caller that passes ws = NULL): cleared y + float atomics
        # instead of stored slices + reduce pass
        from gan2shape_amd import lib
        real = lib.split_ws
        lib.split_ws = lambda: (None, 0)
        try:
            for sk in (3, 8):
                mc.WINO_FORCE = sk
                _check(mc.modconv_raw(xd, wd, sid, sod, PLAIN, transpose), exp, cx * 9, f"split-K {sk}, atomics")
        finally:
            lib.split_ws = real
        mc.WINO_FORCE = 8
        _check(mc.modconv_raw(xd, wd, sid, sod, PLAIN, transpose), exp, cx * 9, "split-K 8, stored slices")
    finally:
        mc.WINOGRAD, mc.WINO_FORCE = saved


# (B, Cin, Cout, H, W) for the F(4x4,3x3) kernel: whole 4x4 tiles, W / 4 divides 32, an image holds a multiple of 32 tiles
WINO4_CASES = [
    (8, 128, 128, 128, 128),   # G convs[9] / D conv1 at 128^2: 32 tiles = one tile row
    (8, 256, 256, 64, 64),     # two tile rows per block
    (8, 512, 512, 32, 32),     # four tile rows per block; library choice = 2 K slices
    (9, 64, 128, 64, 64),      # VGG conv2_1 at B = 9 (data-gradient form: 128 -> 64)
    (3, 68, 64, 32, 64),       # Cr a multiple of 4 only, non-square map
    (2, 64, 192, 16, 32),      # the smallest image that holds 32 tiles; three channel blocks
]


def _check4(y, exp, K, what):
    """F(4x4,3x3) in fp32: transform constants up to 8 and 1/24 amplify the rounding of the 36 products;
    measured <= 1.5e-5 of max|y| on the workload's layers (profiles/r04_wino4_microbench.txt) against 7e-7 for
    F(2x2): five times the direct kernel's atol."""
    atol = 1e-4 * max(1.0, math.sqrt(K / 1152.0))
    got = y.cpu().numpy()
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    np.testing.assert_allclose(got, exp, rtol=2e-4, atol=atol, err_msg=what)


@pytest.mark.parametrize("transpose", [0, 1])
@pytest.mark.parametrize("B,cin,cout,H,W", WINO4_CASES)
def test_winograd4_vs_oracle(L, B, cin, cout, H, W, transpose):
    """g2s_conv3x3_wino4 (transformed weights from g2s_wino4_weights) against the direct-convolution oracle:
    forward and data-gradient form, whole tiles and 2 / 3 K slices through the workspace, with and without the
    style / demodulation scales, without a workspace (falls back to whole tiles); shapes the kernel does not
    take go to F(2x2)."""
    from gan2shape_amd import lib, modconv as mc
    rng = np.random.default_rng(B + cin + cout + H + W + transpose + 44)
    w = (rng.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    cx, cy = (cout, cin) if transpose else (cin, cout)
    xd_shape = (B, cx, H, W)
    saved = (mc.WINOGRAD, mc.WINO_FORCE, mc.WINO4)
    try:
        mc.WINOGRAD, mc.WINO4, mc.WINO_FORCE = True, True, "w4:0"
        if not mc.wino4_supported(B, cx, cy, H, W):
            x0 = torch.zeros(xd_shape, device="cuda")
            assert mc.wino_choice(x0, dev(w), PLAIN, transpose, 0) == 0    # F(2x2), library partition
            return
        x = rng.standard_normal(xd_shape).astype(np.float32)
        s_in = (1 + 0.3 * rng.standard_normal((B, cx))).astype(np.float32)
        s_out = None if transpose else (1 + 0.3 * rng.standard_normal((B, cy))).astype(np.float32)
        exp = expected_modconv(x, w, s_in, s_out, PLAIN, transpose)
        exp_plain = expected_modconv(x, w, None, None, PLAIN, transpose)
        xd, wd, sid = dev(x), dev(w), dev(s_in)
        sod = None if s_out is None else dev(s_out)
        for sk in (0, 1, 2, 3):
            mc.WINO_FORCE = f"w4:{sk}"
            assert mc.wino_choice(xd, wd, PLAIN, transpose, 0) == f"w4:{sk}"
            _check4(mc.modconv_raw(xd, wd, sid, sod, PLAIN, transpose), exp, cx * 9, f"F(4x4) splitk {sk}")
        mc.WINO_FORCE = "w4:0"
        _check4(mc.modconv_raw(xd, wd, None, None, PLAIN, transpose), exp_plain, cx * 9, "F(4x4), no scales")
        real = lib.split_ws
        lib.split_ws = lambda: (None, 0)       # no workspace: the K split is dropped, whole tiles
        try:
            mc.WINO_FORCE = "w4:3"
            _check4(mc.modconv_raw(xd, wd, sid, sod, PLAIN, transpose), exp, cx * 9, "F(4x4) splitk 3, no workspace")
        finally:
            lib.split_ws = real
        mc.WINO4 = False
        assert mc.wino_choice(xd, wd, PLAIN, transpose, 0) == 0
    finally:
        mc.WINOGRAD, mc.WINO_FORCE, mc.WINO4 = saved


@pytest.mark.parametrize("B,cin,cout,H", [(8, 128, 128, 128), (8, 512, 512, 32), (2, 64, 192, 32)])
def test_winograd4_epilogues_vs_oracle(L, B, cin, cout, H):
    """The two epilogues of g2s_conv3x3_wino4 on the workload's call paths: the StyledConv tail (noise + bias +
    leaky-ReLU: modconv.modconv_nba_raw) and bias + leaky-ReLU without scales (conv_bias_act: D's conv1), whole
    tiles and K slices (the reduce pass applies the tail there)."""
    from gan2shape_amd import modconv as mc
    from gan2shape_amd.modconv import conv_bias_act
    rng = np.random.default_rng(B + cin + cout + H + 45)
    x = rng.standard_normal((B, cin, H, H)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    s_in = (1 + 0.3 * rng.standard_normal((B, cin))).astype(np.float32)
    s_out = (1 + 0.3 * rng.standard_normal((B, cout))).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    noise = rng.standard_normal((H, H)).astype(np.float32)
    nw, alpha, gain = 0.37, 0.2, 2 ** 0.5
    exp = _nba_expected(expected_modconv(x, w, s_in, s_out, PLAIN, 0), bias, noise, nw, alpha, gain)
    pre = expected_modconv(x, w, None, None, PLAIN, 0) + bias[None, :, None, None]
    exp_ba = (np.where(pre > 0, pre, pre * alpha) * gain).astype(np.float32)
    args = (dev(x), dev(w), dev(s_in), dev(s_out), dev(bias), dev(noise).view(1, 1, H, H), dev(np.array([nw], np.float32)), alpha, gain)
    saved = (mc.WINOGRAD, mc.WINO_FORCE, mc.WINO4)
    try:
        mc.WINOGRAD, mc.WINO4 = True, True
        assert mc.wino4_supported(B, cin, cout, H, H)
        for sk in (1, 2):
            mc.WINO_FORCE = f"w4:{sk}"
            _check4(mc.modconv_nba_raw(*args), exp, cin * 9, f"F(4x4) StyledConv tail, splitk {sk}")
            _check4(conv_bias_act(dev(x), dev(w), dev(bias), PLAIN, alpha, gain), exp_ba, cin * 9, f"F(4x4) bias + act, splitk {sk}")
    finally:
        mc.WINOGRAD, mc.WINO_FORCE, mc.WINO4 = saved


def test_winograd_weights_follow_the_tensor_version(L):
    """The transformed weights are cached per (tensor, version): an in-place update of the weights
    (a different checkpoint loaded into the same module) must be seen."""
    from gan2shape_amd import modconv as mc
    torch.manual_seed(0)
    x = torch.randn(8, 64, 32, 32, device="cuda")
    w = torch.randn(64, 64, 3, 3, device="cuda") / 24
    saved = (mc.WINOGRAD, mc.WINO_FORCE)
    try:
        mc.WINOGRAD, mc.WINO_FORCE = True, 0
        y0 = mc.modconv_raw(x, w, None, None, PLAIN, 0)
        w.mul_(2.0)
        y1 = mc.modconv_raw(x, w, None, None, PLAIN, 0)
        torch.testing.assert_close(y1, 2 * y0, rtol=1e-5, atol=1e-5)
        mc.WINO_FORCE = "direct"
        torch.testing.assert_close(mc.modconv_raw(x, w, None, None, PLAIN, 0), y1, rtol=1e-4, atol=1e-4)
    finally:
        mc.WINOGRAD, mc.WINO_FORCE = saved


# ----------------------------------------------------------------------------- fp16 operands (config 5)
@pytest.mark.parametrize("transpose", [0, 1])
@pytest.mark.parametrize("B,cin,cout,H,k,mode", [
    (4, 128, 128, 64, 3, PLAIN), (2, 512, 512, 16, 3, PLAIN), (2, 256, 128, 32, 3, UP2), (2, 128, 256, 33, 3, DOWN2),
    (2, 128, 3, 64, 1, PLAIN), (3, 70, 50, 19, 3, PLAIN), (2, 40, 24, 9, 3, UP2), (2, 3, 64, 32, 3, PLAIN),
    (1, 64, 64, 256, 3, PLAIN),     # config 5's own 256-wide maps: waves without a border lane (quad form)
    (2, 16, 16, 3, 3, PLAIN)])      # maps narrower than 4: the un-pipelined 3x3 body
def test_modconv_f16_operands_vs_oracle(L, B, cin, cout, H, k, mode, transpose):
    """g2s_modconv_f16 (fp16 operands, fp32 accumulation — BASELINE config 5) against the fp32
    oracle: the operand rounding (2^-11 relative, random sign) leaves ~1e-3 of the output scale after
    K = Cin k^2 terms; bound: 4e-3 of the rms output elementwise."""
    from gan2shape_amd import modconv as mc
    rng = np.random.default_rng(B + cin + cout + H + mode + transpose)
    w = (rng.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k)).astype(np.float32)
    if transpose:
        oh = {PLAIN: H, UP2: (H - 1) * 2 + k, DOWN2: (H - k) // 2 + 1}[mode]
        x = rng.standard_normal((B, cout, oh, oh)).astype(np.float32)
        s_in, s_out = (1 + 0.3 * rng.standard_normal((B, cout))).astype(np.float32), None
    else:
        x = rng.standard_normal((B, cin, H, H)).astype(np.float32)
        s_in = (1 + 0.3 * rng.standard_normal((B, cin))).astype(np.float32)
        s_out = (1 + 0.3 * rng.standard_normal((B, cout))).astype(np.float32)
    exp = expected_modconv(x, w, s_in, s_out, mode, transpose)
    saved = mc.OPERANDS
    try:
        mc.OPERANDS = "f16"
        y = mc.modconv_raw(dev(x), dev(w), dev(s_in), None if s_out is None else dev(s_out), mode, transpose)
    finally:
        mc.OPERANDS = saved
    got = y.cpu().numpy()
    assert got.shape == exp.shape
    rms = float(np.sqrt((exp.astype(np.float64) ** 2).mean()))
    assert np.abs(got - exp).max() <= 4e-3 * rms, (np.abs(got - exp).max(), rms)
    assert np.linalg.norm(got - exp) <= 1e-3 * np.linalg.norm(exp)


# ----------------------------------------------------------------------------- fromRGB: the thin 1x1 path
@pytest.mark.parametrize("B,cin,cout,H,W", [(8, 3, 128, 128, 128), (5, 4, 24, 120, 132), (3, 1, 40, 160, 140)])
def test_thin_1x1_path_vs_oracle_and_mfma(L, B, cin, cout, H, W):
    """ConvLayer(3, C, 1) + bias + leaky-ReLU (the discriminator's fromRGB, stylegan2-pytorch/model.py:709)
    runs as a streaming kernel (csrc/thinconv.hip) when it covers >= 64 K pixels: against the C oracle at
    the tile tests' tolerance and against the MFMA kernel the same call takes when a tile is forced."""
    from gan2shape_amd.modconv import conv_bias_act
    rng = np.random.default_rng(B + cin + cout + H)
    w = (rng.standard_normal((cout, cin, 1, 1)) / math.sqrt(cin)).astype(np.float32)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    pre = expected_modconv(x, w, None, None, PLAIN, 0) + b[None, :, None, None]
    exp = (np.where(pre > 0, pre, pre * 0.2) * 2 ** 0.5).astype(np.float32)
    try:
        L.g2s_modconv_tune(-1, -1)
        _check(conv_bias_act(dev(x), dev(w), dev(b), PLAIN, 0.2, 2 ** 0.5), exp, cin, "thin path")
        L.g2s_modconv_tune(2, 1)
        _check(conv_bias_act(dev(x), dev(w), dev(b), PLAIN, 0.2, 2 ** 0.5), exp, cin, "MFMA path (forced tile)")
    finally:
        L.g2s_modconv_tune(-1, -1)


# ----------------------------------------------------------------------------- StyledConv tail in the epilogue (round 4)
def _nba_expected(pre, bias, noise, nw, alpha, gain):
    v = pre + bias[None, :, None, None] + nw * noise[None, None]
    return (np.where(v > 0, v, v * alpha) * gain).astype(np.float32)


@pytest.mark.parametrize("B,cin,cout,H", [
    (8, 128, 128, 128),    # G convs[9] (Winograd whole tiles; direct <128,128,18>)
    (8, 512, 512, 16),     # G convs[3]: split-K / stream-K partitions
    (8, 512, 512, 8),      # direct kernel by the measured table, split-K (deferred tail: g2s_noise_bias_act in place)
    (8, 512, 512, 4),      # conv1 on the constant input
    (3, 70, 130, 22),      # ragged channels, tiles spanning batch entries
])
def test_styledconv_tail_in_the_epilogue_vs_oracle(L, B, cin, cout, H):
    """g2s_modconv_nba / g2s_conv3x3_wino_nba (modconv.modconv_nba_raw): the modulated 3x3 convolution with
    NoiseInjection + FusedLeakyReLU in its epilogue (stylegan2-pytorch/model.py:321-355) against the C oracle's
    convolution followed by the tail in numpy — every tile x split-K of the direct kernel, every partition of the
    Winograd kernel (whole tiles, split-K with the reduce pass, stream-K with its reduce pass, atomics without a
    workspace).  An activation input within the tolerance of zero may take the other slope: atol covers it."""
    from gan2shape_amd import lib, modconv as mc
    rng = np.random.default_rng(B + cin + cout + H + 4)
    x = rng.standard_normal((B, cin, H, H)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    s_in = (1 + 0.3 * rng.standard_normal((B, cin))).astype(np.float32)
    s_out = (1 + 0.3 * rng.standard_normal((B, cout))).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    noise = rng.standard_normal((H, H)).astype(np.float32)
    nw, alpha, gain = 0.37, 0.2, 2 ** 0.5
    exp = _nba_expected(expected_modconv(x, w, s_in, s_out, PLAIN, 0), bias, noise, nw, alpha, gain)
    args = (dev(x), dev(w), dev(s_in), dev(s_out), dev(bias), dev(noise).view(1, 1, H, H), dev(np.array([nw], np.float32)), alpha, gain)
    K = cin * 9
    saved = (mc.WINOGRAD, mc.WINO_FORCE)
    try:
        L.g2s_modconv_tune(-1, -1)
        _check_dispatched(mc.modconv_nba_raw(*args), exp, K, "built-in choice", args[0], args[1], PLAIN, 0, 1)
        if H >= 8:
            mc.WINOGRAD = True
            for sk in (0, 1, 3, -7, -256):
                mc.WINO_FORCE = sk
                _check(mc.modconv_nba_raw(*args), exp, K, f"winograd partition {sk}")
            real = lib.split_ws
            lib.split_ws = lambda: (None, 0)       # no workspace: float atomics + the deferred tail
            try:
                mc.WINO_FORCE = 3
                _check(mc.modconv_nba_raw(*args), exp, K, "winograd split-K 3, atomics")
            finally:
                lib.split_ws = real
        mc.WINO_FORCE = "direct"
        for tile, sk in [(-1, -1)] + [(t, s) for t in TILES for s in SPLITS]:
            assert L.g2s_modconv_tune(tile, sk) == 0
            _check(mc.modconv_nba_raw(*args), exp, K, f"direct tile {tile} split-K {sk}")
    finally:
        L.g2s_modconv_tune(-1, -1)
        mc.WINOGRAD, mc.WINO_FORCE = saved


@pytest.mark.parametrize("B,C,H", [(8, 128, 129), (2, 512, 9), (3, 20, 35), (2, 8, 67)])
def test_blur_with_the_styledconv_tail_vs_oracle(L, B, C, H):
    """g2s_upfirdn2d_nba: the Blur behind an up-sampling StyledConv's transposed convolution (model.py:264-275;
    kernel [1,3,3,1] x 4, pad (1, 1)) with noise + bias + leaky ReLU in its store, against the C oracle's
    upfirdn2d and the tail in numpy."""
    from oracle import capi
    from gan2shape_amd import lib, stylegan2 as sg2
    rng = np.random.default_rng(B + C + H)
    x = rng.standard_normal((B, C, H, H)).astype(np.float32)
    k = (sg2.make_kernel([1, 3, 3, 1]) * 4).numpy().astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32)
    nw, alpha, gain = -0.8, 0.2, 2 ** 0.5
    pre = capi.upfirdn2d(x, k, (1, 1), (1, 1), (1, 1, 1, 1))
    oh = pre.shape[2]
    noise = rng.standard_normal((oh, oh)).astype(np.float32)
    exp = _nba_expected(pre, bias, noise, nw, alpha, gain)
    y = torch.empty((B, C, oh, oh), device="cuda")
    xd, kd, bd, nd, nwd = dev(x), dev(k), dev(bias), dev(noise), dev(np.array([nw], np.float32))
    lib.check(L.g2s_upfirdn2d_nba(lib.ptr(xd), lib.ptr(kd), lib.ptr(y), B * C, C, H, H, 4, 4, 1, 1, 1, 1, 1, 1,
                                  lib.ptr(bd), lib.ptr(nd), lib.ptr(nwd), alpha, gain, lib.stream()))
    np.testing.assert_allclose(y.cpu().numpy(), exp, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("B,C,H,two,with_gdot", [(8, 128, 128, True, True), (3, 20, 4, False, True), (2, 7, 9, True, False),
                                                 (2, 512, 16, False, False)])
def test_synth_bwd_rows_vs_float64(L, B, C, H, two, with_gdot):
    """g2s_synth_bwd_rows against the chain it replaces, in float64 torch: the consumers' style gradients
    sum_hw x * g, the gated joined gradient (g1 s1 + g2 s2) * gain * (x > 0 ? 1 : slope), and the producer's
    demodulation gradient sum_hw out * yconv / demod with yconv recovered from x."""
    from gan2shape_amd import lib
    torch.manual_seed(B + C + H)
    slope, gain, nw = 0.2, 2 ** 0.5, 0.6
    yconv = torch.randn(B, C, H, H, dtype=torch.float64)
    noise, bias = torch.randn(H, H, dtype=torch.float64), torch.randn(C, dtype=torch.float64)
    pre = yconv + nw * noise + bias.view(1, C, 1, 1)
    x = torch.where(pre > 0, pre, pre * slope) * gain
    g1, g2 = torch.randn_like(x), torch.randn_like(x)
    s1, s2, demod = torch.randn(B, C, dtype=torch.float64), torch.randn(B, C, dtype=torch.float64), 0.5 + torch.rand(B, C, dtype=torch.float64)
    joined = g1 * s1[:, :, None, None] + (g2 * s2[:, :, None, None] if two else 0)
    out_ref = joined * torch.where(x > 0, gain, gain * slope)
    f = lambda t: None if t is None else t.float().cuda().contiguous()      # noqa: E731
    xd, g1d, g2d = f(x), f(g1), f(g2) if two else None
    s1d, s2d, nzd, nwd, bd, dmd = f(s1), f(s2) if two else None, f(noise), f(torch.tensor([nw])), f(bias), f(demod)   # kept alive
    out = torch.empty_like(xd)
    dot1, dot2, gdot = (torch.empty(B, C, device="cuda") for _ in range(3))
    lib.check(L.g2s_synth_bwd_rows(lib.ptr(xd), lib.ptr(g1d), lib.ptr(s1d), lib.ptr(g2d), lib.ptr(s2d),
                                   lib.ptr(nzd), lib.ptr(nwd), lib.ptr(bd), lib.ptr(dmd),
                                   lib.ptr(out), lib.ptr(dot1), lib.ptr(dot2 if two else None),
                                   lib.ptr(gdot if with_gdot else None), B * C, C, H * H, slope, gain, lib.stream()))
    scale = float(out_ref.abs().max())
    assert float((out.double().cpu() - out_ref).abs().max()) <= 2e-6 * scale
    n = H * H
    np.testing.assert_allclose(dot1.double().cpu().numpy(), (x * g1).sum((2, 3)).numpy(), rtol=1e-4, atol=2e-5 * n ** 0.5)
    if two:
        np.testing.assert_allclose(dot2.double().cpu().numpy(), (x * g2).sum((2, 3)).numpy(), rtol=1e-4, atol=2e-5 * n ** 0.5)
    if with_gdot:
        ref = (out_ref * yconv).sum((2, 3)) / demod
        np.testing.assert_allclose(gdot.double().cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4 * n ** 0.5)
