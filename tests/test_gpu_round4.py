"""Round-4 GPU tests: whole-step gradients against central differences of the loss, and the descent
assertions that round 3 had relaxed (ADVICE tests/test_gpu_round3.py:182)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_PROJ = 4


@pytest.fixture(scope="module")
def g2s():
    import gan2shape_amd  # noqa: F401
    from gan2shape_amd import lib
    lib.load()
    return gan2shape_amd


@pytest.fixture(scope="module")
def scenario(g2s):
    """The set-up of round 3's optimisation test: fresh model (seed 0), 60 iterations of prior pre-training,
    25 iterations of step 1 at lr 1e-3, then the hand-offs of one (untrained) step-2 forward."""
    import bench
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import Trainer
    torch.manual_seed(0)
    cfg = bench.face_config(n_proj=N_PROJ)
    cfg["n_epochs_prior"] = 60
    cfg["learning_rate"] = 1e-3          # step 1 only (see below): the reference's 1e-4 moves the albedo net too little in 25 iterations
    t = Trainer(GAN2Shape, cfg, device="cuda")
    image, latent = bench.synthetic_sample(t.model, 4321, torch.device("cuda"))
    prior_losses = t.pretrain_on_prior(image, 0)
    m = t.model
    torch.manual_seed(101)
    losses1, out = [], None
    for _ in range(25):
        t.optim_step1.zero_grad()
        loss, out = m.forward_step1(image, latent, None, n_proj_samples=N_PROJ)
        loss.backward()
        t.optim_step1.step()
        losses1.append(float(loss.detach()))
    collected = {2: out}
    torch.manual_seed(7)
    with torch.no_grad():
        _, collected[3] = m.forward_step2(image, latent, out, n_proj_samples=N_PROJ)
    return dict(t=t, m=m, image=image, latent=latent, collected=collected, prior=prior_losses, step1=losses1)


def _params(t, step):
    return [p for group in getattr(t, f"optim_step{step}").param_groups for p in group["params"]]


def _flat(ts):
    return torch.cat([x.reshape(-1) for x in ts])


def _set(ps, vec):
    off = 0
    with torch.no_grad():
        for p in ps:
            p.copy_(vec[off:off + p.numel()].view_as(p))
            off += p.numel()


def _evaluate(sc, step, grad):
    """Loss (and flat float64 gradient) of the FIXED objective of a step: the random draws of step 2 are
    re-seeded before every evaluation."""
    m, ps = sc["m"], _params(sc["t"], step)
    torch.manual_seed(7)
    for p in ps:
        p.grad = None
    with torch.enable_grad() if grad else torch.no_grad():
        loss, _ = getattr(m, f"forward_step{step}")(sc["image"], sc["latent"], sc["collected"][step], n_proj_samples=N_PROJ)
        if not grad:
            return float(loss)
        loss.backward()
    g = _flat([p.grad if p.grad is not None else torch.zeros_like(p) for p in ps]).double()
    for p in ps:
        p.grad = None
    return float(loss.detach()), g


@pytest.mark.parametrize("step", [2, 3])
def test_step_gradient_is_the_directional_derivative_of_the_loss(scenario, step):
    """(L(theta + eps d) - L(theta - eps d)) / (2 eps) against <g, d> through forward_step2 / forward_step3 as the
    trainer runs them — the product path: discriminator-feature loss and LPIPS as single autograd nodes with
    hand-written backward passes (losses._DFeatureL1, lpips._VggLpips), demodulated convolutions as one node
    (modconv.ModConvDemodFunction), every kernel's analytic backward.  Directions: d = g / |g| (the whole gradient:
    the difference quotient must return |g|) and g restricted to each parameter group — step 3: each of the four
    nets; step 2: thirds of the encoder's tensor list —, so a wrong gradient in ONE net or layer range cannot hide
    in the norm of the rest.  fp32 window measured with tools/dirderiv.py (gpurun_out/r4_dirderiv.log): at eps =
    3e-4 the quotient along g / |g| is 0.2 - 0.3 % below <g, d> (curvature) and the loss repeats to 1.4e-6
    (2.4e-3 on the quotient).  Also: the gradient of the op-by-op autograd path (ONE_NODE off) equals the
    product's (measured 4.6e-4 / 3.6e-4 of its norm, cosine 0.9999999)."""
    from gan2shape_amd import lib, losses, lpips
    sc = scenario
    ps = _params(sc["t"], step)
    theta0 = _flat([p.detach() for p in ps]).clone()
    # step 3 in deterministic mode: in the default mode its gradient scatters by 6e-3 from run to run (float-atomic
    # noise in the nets' forward decides supersamples on fold edges) — the loss itself does not
    prev = lib.set_deterministic(step == 3)
    try:
        loss0, g = _evaluate(sc, step, True)
        try:
            losses.DiscriminatorLoss.ONE_NODE = lpips.PNetLin.ONE_NODE = False
            _, g_ops = _evaluate(sc, step, True)
        finally:
            losses.DiscriminatorLoss.ONE_NODE = lpips.PNetLin.ONE_NODE = True
        cos = float(g @ g_ops / (g.norm() * g_ops.norm()))
        rel = float((g - g_ops).norm() / g_ops.norm())
        print(f"[step {step}] loss {loss0:.6f} |g| {float(g.norm()):.5e}; product vs op-by-op: cosine {cos:.8f}, rel diff {rel:.2e}")
        assert cos >= 0.999999 and rel <= 2e-3, (cos, rel)

        sizes = [p.numel() for p in ps]
        if step == 3:
            m = sc["m"]
            groups = {}
            for name in ("lighting", "viewpoint", "depth", "albedo"):
                ids = {id(p) for p in getattr(m, f"{name}_net").parameters()}
                groups[name] = [id(p) in ids for p in ps]
        else:
            third = (len(ps) + 2) // 3
            groups = {f"tensors {i * third}..{min(len(ps), (i + 1) * third) - 1}": [i * third <= j < (i + 1) * third for j in range(len(ps))]
                      for i in range(3)}
        dirs = {"g / |g|": g / g.norm()}
        for name, sel in groups.items():
            mask = torch.cat([torch.full((n,), float(s_), dtype=torch.float64, device=g.device) for n, s_ in zip(sizes, sel)])
            d = g * mask
            assert float(d.norm()) > 0, name
            dirs[name] = d / d.norm()
        # step 2: one central difference of half-width 3e-4 (the loss repeats to 1.4e-6: 2.4e-3 on the quotient).
        # Step 3's loss repeats to 1e-7 (deterministic mode) but is only PIECEWISE smooth in the pose and the depth —
        # L1 kinks, bilinear-cell borders of grid_sample, supersamples changing face: along the viewpoint net's
        # gradient the slope measured over [-3e-4, -1e-4] and [1e-4, 3e-4] is 15 - 20 % below the slope at theta in every
        # run, and one run of three had a jump inside +-1e-4 (+8.6 %).  The derivative AT theta is what autograd returns,
        # so the quotient is taken at three small half-widths (2e-5, 5e-5, 1e-4: rounding noise 2e-3 .. 5e-4 on the
        # quotient) and their median is held to 2 % (10 % along the viewpoint net's gradient, see below)
        halves = (3e-4,) if step == 2 else (2e-5, 5e-5, 1e-4)
        for name, d in dirs.items():
            want = float(g @ d)
            quotients = []
            for half in halves:
                _set(ps, (theta0.double() + half * d).float())
                lp_ = _evaluate(sc, step, False)
                _set(ps, (theta0.double() - half * d).float())
                lm_ = _evaluate(sc, step, False)
                quotients.append((lp_ - lm_) / (2 * half))
            got = float(np.median(quotients))
            print(f"[step {step}] {name:18s} <g, d> = {want:.5e}   central differences " + " ".join(f"{q:.5e}" for q in quotients)
                  + f"   (median {got / want - 1:+.2%})")
            # the viewpoint direction moves the silhouette: supersamples at the mesh border change between surface and
            # background as the pose moves, a term the rasterizer's gradient (the reference's algorithm: no gradient
            # through coverage) does not contain and a difference quotient does — measured +3.9 .. +6.9 % over five runs,
            # always the same sign; depth / albedo / lighting directions and the whole gradient: 0.0 .. 0.3 %
            rtol = 2e-2 if name != "viewpoint" else 1e-1
            assert abs(got - want) <= rtol * abs(want) + 6e-3, (step, name, want, quotients)
    finally:
        _set(ps, theta0)
        lib.set_deterministic(prev)


def test_every_step_kind_lowers_its_own_loss(scenario):
    """The descent assertion round 3 had relaxed to `tail < 1.25 * head`, restored.

    CAUSE of the red run (gpurun_out/r3_gpu_tests_11_1.log: step 2's loss 3.31 -> 3.51 over 25 updates on a fixed
    objective), established with tools/dirderiv.py and tools/descent_curves.py (gpurun_out/r4_dirderiv*.log,
    r4_descent_{old,new}.log): not the gradient — the test above holds it to central differences, and
    the hand-written backward equals autograd's op-by-op chain to 5e-4 — but the test's step size.  It ran the
    offset encoder at lr 1e-3, 10x the reference's (trainer.py:163-171): Adam's first updates are sign-like steps
    of |d theta| ~ lr * sqrt(P) = 3.7 on a parameter vector of norm 38.6 (P = 13.8 M), 700x the radius (~5e-3) in
    which a line search finds the loss linear.  The trajectory is chaotic: float-atomic noise of 1e-7 separates
    two runs within three iterations, and spikes to 4.0 .. 6.1 appear at random iterations in 4 of 12 runs — in
    the one-node and the op-by-op form alike, on the tree of commit 2519979 and on today's.  All 12 runs still
    descend over 25 iterations (tails 1.8 .. 2.9 from 3.3) unless a spike falls into the last three.  At the
    reference's lr 1e-4 both torch.optim.Adam and g2s_adam_step descend smoothly (3.375 -> 1.77 after 25, 1.02
    after 60 iterations; step 3: 1.379 -> 0.81 after 25).

    So: step 1 at lr 1e-3 as before (the albedo net against a fixed target: smooth), steps 2 and 3 at the
    reference's 1e-4 on their fixed objectives with real descent bounds (measured: step 2 3.31 -> 1.60 in 30 iterations, ratio
    0.48 - 0.50; step 3, started from that trained encoder's samples, 0.86 -> 0.72 / 0.88 -> 0.78 in 25 iterations, ratios
    0.86 / 0.89 over runs, monotone; 40 iterations here)."""
    sc = scenario
    t, m = sc["t"], sc["m"]
    assert len(sc["prior"]) == 60 and sc["prior"][-1] < 0.2 * sc["prior"][0], (sc["prior"][0], sc["prior"][-1])
    l1 = sc["step1"]
    assert np.mean(l1[-3:]) < 0.97 * np.mean(l1[:3]), l1
    collected = sc["collected"][2]
    for step, n_it, bound in ((2, 30, 0.70), (3, 40, 0.93)):
        optim = getattr(t, f"optim_step{step}")
        for group in optim.param_groups:
            group["lr"] = 1e-4
        forward = getattr(m, f"forward_step{step}")
        curve, out = [], None
        for _ in range(n_it):
            torch.manual_seed(7)             # step 2: the same pseudo views / lights every iteration: one objective
            optim.zero_grad()
            loss, out = forward(sc["image"], sc["latent"], collected, n_proj_samples=N_PROJ)
            loss.backward()
            optim.step()
            curve.append(float(loss.detach()))
        collected = out
        assert all(np.isfinite(curve)), (step, curve)
        head, tail = np.mean(curve[:3]), np.mean(curve[-3:])
        print(f"[descent] step {step}: head {head:.4f} tail {tail:.4f} ratio {tail / head:.3f} | " + " ".join(f"{v:.3f}" for v in curve))
        assert tail < bound * head, (step, head, tail, curve)
        # and no update overshoots: at the reference's step size the curve never rises above its start, and
        # (measured: step 2 falls with 3 - 5 small upticks in 30 iterations, step 3 monotonically) most updates lower it
        assert max(curve) <= 1.02 * curve[0], (step, curve)
        assert np.mean(np.diff(curve) < 0) >= 0.7, (step, curve)


# ----------------------------------------------------------------------------- joint mode: flat bucket + graph segments
def _joint_graph_worker(rank, world, port, q, graphs):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from gan2shape_amd import sharding
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import GeneralizingTrainer2
    if world > 1:
        sharding.init_distributed("gloo")      # both ranks share the one GPU of the test box
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    cfg = bench.face_config(n_proj=2)
    cfg.update(n_epochs_prior=1, n_epochs_generalized=1)
    torch.manual_seed(0)
    t = GeneralizingTrainer2(GAN2Shape, cfg, device=dev, capturable=True)
    data = []
    for i in range(2 * world):
        image, latent = bench.synthetic_sample(t.model, 100 + i, dev)
        data.append((image[0].cpu(), latent[0].cpu(), i))
    stages = [{'step1': 1, 'step2': 5, 'step3': 4}]
    n = t.fit(data, stages=stages, batch_size=world, rank=rank, world_size=world, graphs=graphs)
    nets = ('depth_net', 'albedo_net', 'viewpoint_net', 'lighting_net', 'offset_encoder_net')
    flat = torch.cat([p.detach().reshape(-1).cpu() for nme, p in t.model.named_parameters() if nme.split('.')[0] in nets])
    # the optimisers' persistent flat buckets exist (one per optimiser that packed): 17 / 55 / 90 MB
    bound = (len(sharding._BUCKETS) >= (3 if world > 1 else 2)) if (world > 1 or graphs) else True
    q.put((rank, n, float(flat.double().sum()), float(flat.double().abs().sum()), bool(torch.isfinite(flat).all()),
           [h[3] for h in t.history], bound))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _spawn_joint(world, graphs):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_joint_graph_worker, args=(r, world, port, q, graphs)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_joint_trainer_graph_segments_one_rank():
    """GeneralizingTrainer2.fit(graphs=True): steps 2 / 3 of every image replay as two captured segments
    (forward + backward + GradBucket.pack | optimiser step from the bucket's views); same iteration count and
    history shape as the eager loop, finite losses, and losses of the same size (the random draws of a
    replayed graph come from the graph-registered generator, so the two runs are not bit-equal)."""
    (_, n_e, s_e, a_e, ok_e, hist_e, _), = _spawn_joint(1, False)
    (_, n_g, s_g, a_g, ok_g, hist_g, bound), = _spawn_joint(1, True)
    assert ok_e and ok_g and bound
    assert n_e == n_g == 2 * (1 + 5 + 4)
    assert len(hist_e) == len(hist_g) and all(np.isfinite(v) for v in hist_g)
    # step 1 draws nothing at random: its loss on the first batch is the same run (summation-order noise only)
    assert abs(hist_e[0] - hist_g[0]) <= 1e-4 * abs(hist_e[0]), (hist_e[0], hist_g[0])
    for a, b in zip(hist_e, hist_g):
        assert 0.5 < a / b < 2.0, (hist_e, hist_g)
    assert abs(a_e - a_g) <= 1e-2 * a_e


def test_joint_trainer_graph_segments_two_ranks_share_parameters():
    """The same with world_size 2 (gloo between two processes on the one GPU): each rank replays segment A on
    ITS image, the flat bucket is all-reduced between the segments, segment B steps from the averaged
    bucket — both ranks must end on the same parameters (they would drift apart within one iteration if
    the collective missed a replay, since the ranks train different images)."""
    res = _spawn_joint(2, True)
    (_, n0, s0, a0, ok0, h0, b0), (_, n1, s1, a1, ok1, h1, b1) = res
    assert ok0 and ok1 and b0 and b1 and n0 == n1 == 2 * (1 + 5 + 4)
    assert abs(s0 - s1) <= 1e-6 * a0 and abs(a0 - a1) <= 1e-6 * a0
    assert h0 != h1          # different images: different losses, same parameters


# ----------------------------------------------------------------------------- frozen generator as one autograd node
@pytest.mark.parametrize("size,batch", [(128, 8), (64, 3)])
def test_generator_one_node_equals_op_by_op(g2s, size, batch):
    """synthesis._Synthesis (convolutions with the noise + bias + leaky-ReLU tail in their epilogue, the Blur with
    the same tail in its store, one fused row pass per activation in backward) against the op-by-op autograd path
    of Generator.forward (stylegan2-pytorch/model.py:545-627): the image to fp32 summation order, the latent
    gradient to fp32 noise (every layer's two style paths — modulation and demodulation — and the activation
    gates are in it; the demodulation gradient recovers the convolution output by inverting the leaky ReLU)."""
    from gan2shape_amd import stylegan2 as sg2
    from test_gpu_model import fill_deterministic
    from model_cases import prepare_generator
    torch.manual_seed(0)
    G = sg2.Generator(size, 512, 8, channel_multiplier=1)
    prepare_generator(G, 77, fill_deterministic)
    G = G.cuda().eval().requires_grad_(False)
    w0 = (0.5 * torch.randn(batch, 512)).cuda()
    gy = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(5)).cuda()
    out = {}
    try:
        for one in (True, False):
            sg2.Generator.ONE_NODE = one
            w = w0.clone().requires_grad_(True)
            img, _ = G([w], input_is_w=True, randomize_noise=False)
            (gw,) = torch.autograd.grad(img, w, gy)
            out[one] = (img.detach(), gw)
    finally:
        sg2.Generator.ONE_NODE = True
    (img1, g1), (img0, g0) = out[True], out[False]
    e_img = float((img1 - img0).abs().max() / img0.abs().max())
    e_g = float((g1.double() - g0.double()).norm() / g0.double().norm())
    cos = float((g1.double() * g0.double()).sum() / (g1.double().norm() * g0.double().norm()))
    print(f"[G one node, size {size} B {batch}] image max |d| / max {e_img:.2e}; latent gradient rel {e_g:.2e}, cosine {cos:.9f}")
    assert e_img <= 2e-6
    # two fp32 evaluations of a gradient that sums 131 072 pixels x 17 layers: the reference's own fp32 run is 5.4e-4
    # (L2) from its float64 run on this generator (gan128.npz g128.ref_fp32_err), and
    # test_generator_128_batch8_vs_reference_golden holds THIS path to the float64 gradient; measured here 6.6e-4
    assert e_g <= 1.5e-3 and cos >= 0.999999
    with torch.no_grad():          # the no-grad forward (sample generation) takes the same node
        img2, _ = G([w0], input_is_w=True, randomize_noise=False)
    assert float((img2 - img1).abs().max()) <= 2e-6 * float(img1.abs().max())


def test_channel_sum_is_the_bias_gradient(g2s):
    """g2s_channel_sum = gy.sum((0, 2, 3)) (the bias gradient of the offset encoder's biased convolutions)."""
    from gan2shape_amd import lib
    L = lib.load()
    torch.manual_seed(0)
    for shape in [(8, 512, 4, 4), (8, 64, 32, 32), (3, 7, 5, 9), (1, 1, 1, 1), (8, 1024, 1, 1)]:
        gy = torch.randn(*shape, device="cuda")
        out = torch.empty(shape[1], device="cuda")
        lib.check(L.g2s_channel_sum(lib.ptr(gy), lib.ptr(out), shape[0], shape[1], shape[2] * shape[3], lib.stream()))
        ref = gy.double().sum((0, 2, 3))
        assert float((out.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), shape


def test_precleared_skips_the_memset_and_the_pool_path_equals_the_plain_one(g2s):
    """g2s_set_precleared (include/g2s.h): while the flag is on, g2s_shading_bwd adds into the caller's glight instead
    of clearing it first (so a buffer prefilled with 1 comes back 1 too high — the memset is really skipped — and a
    cleared one gives the plain result); the flag is per call (restored by lib.precleared); and the op-level wrappers
    give bit-identical gradients with a step's zero pool active (accumulators carved from it) and without one."""
    from gan2shape_amd import fused_geometry as fg, lib, zeropool
    L = lib.load()
    torch.manual_seed(3)
    B, H, W = 4, 16, 16
    normal = torch.nn.functional.normalize(torch.randn(B, H, W, 3, device="cuda"), dim=-1).contiguous()
    light = torch.randn(B, 4, device="cuda")
    albedo = torch.randn(B, 3, H, W, device="cuda")
    gd, gt = torch.randn(B, 1, H, W, device="cuda"), torch.randn(B, 3, H, W, device="cuda")
    gn, ga = torch.empty(B, H, W, 3, device="cuda"), torch.empty(B, 3, H, W, device="cuda")

    def call(glight, flag):
        with lib.precleared(flag):
            lib.check(L.g2s_shading_bwd(lib.ptr(normal), lib.ptr(light), lib.ptr(albedo), lib.ptr(gd), lib.ptr(gt),
                                        lib.ptr(gn), lib.ptr(ga), lib.ptr(glight), B, B, B, H * W, lib.stream()))
        return glight.clone()

    prev = lib.set_deterministic(True)      # one workgroup per image: the sums have a fixed order
    try:
        plain = call(torch.full((B, 4), 7.0, device="cuda"), False)              # cleared by the library
        assert torch.equal(call(torch.zeros(B, 4, device="cuda"), True), plain)  # cleared by the caller
        dirty = call(torch.ones(B, 4, device="cuda"), True)                      # NOT cleared by anyone
        assert float((dirty - plain - 1.0).abs().max()) <= 1e-5 * max(1.0, float(plain.abs().max()))
        assert L.g2s_set_precleared(0) == 0                                      # restored after every call
        # wrapper level: gradients with and without an active pool
        def grads():
            n, l, a = normal.clone().requires_grad_(True), light.clone().requires_grad_(True), albedo.clone().requires_grad_(True)
            diffuse, texture = fg.ShadingFunction.apply(n, l, a)
            (diffuse * gd).sum().add((texture * gt).sum()).backward()
            return n.grad, l.grad, a.grad
        zeropool.end()
        ref = grads()
        for _ in range(2):          # first pass sizes the pool, second is served from it
            zeropool.begin(97, torch.device("cuda", torch.cuda.current_device()))
            got = grads()
            served = zeropool._state["off"] > 0
            zeropool.end()
        assert served, "the second step did not carve its accumulators from the pool"
        for r, g_ in zip(ref, got):
            assert torch.equal(r, g_)
    finally:
        zeropool.end()
        lib.set_deterministic(prev)
