"""GAN2Shape training-step throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1 without a launcher: starts
                                                                 N ranks itself, see self_launch)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload = "face128_n8"): configs/face.yml restated for the offline box — image
128x128, StyleGAN2 size 128 / channel_multiplier 1 / z_dim 512, n_proj_samples 8, ellipsoid-prior
depth, random-init weights, synthetic image + latent (seed 1234 + rank), synthetic view/light
distributions.  One bench "step" = ONE training iteration (zero_grad, forward_stepK, backward,
Adam.step) exactly as GAN2Shape/trainer.py:99-109 runs it; the step kinds cycle in the stage-0
ratio of main.py:148 (700:700:600 = 7:7:6): seven step-1, seven step-2, six step-3 iterations per
20 steps, each kind handing its `collected` to the next as the trainer does.  The reported value is
therefore the stage-0-weighted aggregate  iterations / second.

Multi-GPU: the per-image optimisation is independent, so each rank trains its own image with its
own model replica and optimisers; no data-path collective (weak scaling).  value = N * K / T with T
the max over ranks of the barrier-bracketed wall time.

Also reported on the same JSON line: `roofline` of the dominant custom kernels (the fp32-MFMA
convolution kernels — direct implicit GEMM and Winograd F(2x2,3x3): algorithmic direct-convolution
FLOP / HIP-event time on the launch stream against the 157.3 TFLOP/s fp32 matrix peak, with the
MFMA-executed rate beside it), `roofline_other` (rasterizer, upfirdn2d, fused_bias_act against the
8 TB/s HBM peak) and `cpu_baseline` (the CPU oracle's restatement of the native hot ops plus
torch-CPU timings of the networks on this box's host cores, bounded sample, scaled to
iterations/second).
"""
import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PATTERN = [1] * 7 + [2] * 7 + [3] * 6   # main.py:148 stage 0 = {700, 700, 600}
F32_MFMA_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32


WORKLOADS = {
    # BASELINE.json configs[1]: the headline — configs/face.yml at 128x128, fp32 (the reference's arithmetic)
    "face128_n8": dict(image_size=128, gan_size=128, prior_name='ellipsoid', mfma_operands='f32', dtype="f32"),
    # BASELINE.json configs[4]: 256x256, confidence-map prior, fp16 operands / fp32 accumulation for the
    # frozen G / D / VGG GEMMs (a non-reference extension: SURVEY.md §8d config 5) — its own JSON line,
    # never the headline
    "face256_fp16": dict(image_size=256, gan_size=256, prior_name='confidence', mfma_operands='f16', dtype="f16"),
    # the same workload on the fp32 kernels (direct + Winograd): the line the fp16 path has to beat
    "face256_f32": dict(image_size=256, gan_size=256, prior_name='confidence', mfma_operands='f32', dtype="f32"),
    # BASELINE.json configs[3]: configs/car.yml (gan_size 512, channel_multiplier 2, 8 projected samples) at 128x128
    # under JOINT training (GeneralizingTrainer2, trainer.py:338-479; main.py:141 step mix 13:22:18): one shared
    # model, one image per rank and iteration, ONE flat-bucket gradient all-reduce (mean) per optimiser step
    "car128_joint": dict(image_size=128, gan_size=512, channel_multiplier=2, prior_name='ellipsoid', mfma_operands='f32',
                         dtype="f32", joint=True),
}
JOINT_PATTERN = [1] * 13 + [2] * 22 + [3] * 18      # main.py:141: stages = [{'step1': 13, 'step2': 22, 'step3': 18}]


def face_config(n_proj=8, workload="face128_n8"):
    wl = WORKLOADS[workload]
    cfg = _face_config(n_proj)
    cfg.update(image_size=wl["image_size"], gan_size=wl["gan_size"], prior_name=wl["prior_name"],
               mfma_operands=wl["mfma_operands"], channel_multiplier=wl.get("channel_multiplier", 1))
    if wl.get("joint"):
        cfg["category"] = "car"
    return cfg


def _face_config(n_proj=8):
    return {
        'image_size': 128, 'z_dim': 512, 'gan_size': 128, 'channel_multiplier': 1,
        'gan_ckpt_path': None, 'n_proj_samples': n_proj, 'category': 'face',
        'prior_name': 'ellipsoid', 'n_epochs_prior': 0, 'learning_rate': 1e-4, 'view_scale': 1,
        'rot_center_depth': 1.0, 'fov': 10, 'tex_cube_size': 2,
        'view_mvn_path': None, 'light_mvn_path': None,
        # SURVEY.md §8d synthetic distributions
        'view_mvn': {'mean': [0.] * 6,
                     'cov': [[v * v if i == j else 0. for j in range(6)]
                             for i, v in enumerate([.05, .15, .03, .05, .05, .05])]},
        'light_mvn': {'mean': [0.] * 4,
                      'cov': [[0.01 if i == j else 0. for j in range(4)] for i in range(4)]},
    }


def synthetic_sample(model, seed, device):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(1, 3, model.image_size // 4, model.image_size // 4, generator=g)
    img = torch.tanh(torch.nn.functional.interpolate(img, scale_factor=4, mode='bilinear')).to(device)
    with torch.no_grad():
        w = model.generator.style_forward(torch.randn(1, 512, generator=g).to(device))
    return img, w


class StepRunner:
    """Runs training iterations in the trainer's order, keeping `collected` between kinds."""

    def __init__(self, trainer, image, latent):
        self.t, self.image, self.latent = trainer, image, latent
        self.collected = {1: None, 2: None, 3: None}
        self.kind_ms = {1: [], 2: [], 3: []}
        self.last_loss = {}

    def run(self, kind, timed=False):
        t = self.t
        optim = getattr(t, f'optim_step{kind}')
        src = {1: None, 2: self.collected[1], 3: self.collected[2]}[kind]
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        optim.zero_grad()
        loss, collected = getattr(t.model, f'forward_step{kind}')(
            self.image, self.latent, src, n_proj_samples=t.n_proj_samples)
        loss.backward()
        optim.step()
        sys.modules["gan2shape_amd.zeropool"].end()      # as the trainer does after every iteration
        if timed:
            e1.record()
            self.kind_ms[kind].append((e0, e1))
        self.collected[kind] = collected
        self.last_loss[kind] = loss.detach()
        return loss


class JointStepRunner(StepRunner):
    """Joint (data-parallel) iterations as GeneralizingTrainer2.fit runs them (trainer.py:338-479 semantics):
    every optimiser step is preceded by ONE all-reduce (mean) of the optimiser's persistent flat gradient
    bucket (sharding.GradBucket) — a single process packs and binds the same bucket without the collective,
    so the N = 1 line measures the path the N > 1 ranks run; step 1 centres the depth over all ranks' images."""

    def __init__(self, trainer, image, latent, world):
        super().__init__(trainer, image, latent)
        from gan2shape_amd import sharding
        self.world, self.sh = world, sharding

    def run(self, kind, timed=False):
        t = self.t
        optim = getattr(t, f'optim_step{kind}')
        src = {1: None, 2: self.collected[1], 3: self.collected[2]}[kind]
        t.model.batch_mean = self.sh.global_mean if (kind == 1 and self.world > 1) else None
        optim.zero_grad()
        loss, collected = getattr(t.model, f'forward_step{kind}')(
            self.image, self.latent, src, n_proj_samples=t.n_proj_samples)
        loss.backward()
        t.model.batch_mean = None
        bucket = self.sh.bucket_of([p for g in optim.param_groups for p in g['params']])
        bucket.pack()
        bucket.all_reduce_mean()
        bucket.bind()
        optim.step()
        sys.modules["gan2shape_amd.zeropool"].end()
        self.collected[kind] = collected
        self.last_loss[kind] = loss.detach()
        return loss


class GraphedJointRunner:
    """Steps 2 / 3 (and step 1 of a single process) as two replayed graph segments around the collective
    (graphs.GraphedJointSteps); step 1 of W > 1 ranks holds a collective in its forward and runs eagerly."""

    def __init__(self, graphed, eager, world):
        self.g, self.eager, self.world = graphed, eager, world
        self.kind_ms = {1: [], 2: [], 3: []}
        self.last_loss = graphed.loss

    def run(self, kind, timed=False):
        if kind == 1 and self.world > 1:
            loss = self.eager.run(1)
            self.g.loss[1] = loss.detach()
            return loss
        return self.g.run(kind)


class GraphedRunner:
    """Same interface as StepRunner, one hipGraph replay per iteration."""

    def __init__(self, graphed):
        self.g = graphed
        self.kind_ms = {1: [], 2: [], 3: []}
        self.last_loss = graphed.loss

    def run(self, kind, timed=False):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        loss = self.g.run(kind)
        if timed:
            e1.record()
            self.kind_ms[kind].append((e0, e1))
        return loss


def _large(prof, gflop=1.0):
    """The same figure over the launches of at least `gflop` GFLOP (the small nets' layers are
    launch-latency-bound, not MFMA-bound)."""
    big = [p for p in prof if p[0] >= gflop * 1e9]
    ms = sum(p[1].elapsed_time(p[2]) for p in big)
    if not big or ms <= 0:
        return None
    fl = sum(p[0] for p in big)
    return {"min_gflop": gflop, "launches": len(big), "achieved": fl / (ms * 1e-3) / 1e12,
            "frac": fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
            "mfma_executed": sum(p[4] for p in big) / (ms * 1e-3) / 1e12,
            "share_of_flop": fl / max(sum(p[0] for p in prof), 1.0),
            "share_of_time": ms / max(sum(p[1].elapsed_time(p[2]) for p in prof), 1e-9)}


def _repeat(fn, min_seconds):
    """Run fn until `min_seconds` of wall time have been spent (at least once; a first call that
    already takes longer than that is the only one); returns (seconds per call, calls)."""
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            return dt / n, n


def _host_threads():
    """Threads for the CPU baseline: the cores this process may run on, capped at 16 (the CPU share
    of a one-GPU box on this pool: more threads than that only thrash); G2S_CPU_THREADS overrides."""
    if os.environ.get("G2S_CPU_THREADS"):
        return max(1, int(os.environ["G2S_CPU_THREADS"]))
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    return max(1, min(allowed, 16))


def _cpu_conv_stack(layers, batch, backward, seconds):
    """torch CPU convolutions of a frozen network's layer list [(cin, cout, k, stride, H)], forward
    (+ data-gradient): seconds per pass."""
    import torch.nn.functional as F
    xs = [torch.randn(batch, cin, H, H, requires_grad=backward) for cin, _, _, _, H in layers]
    ws = [torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5 for cin, cout, k, _, _ in layers]

    def run():
        for x, w, (_, _, k, stride, _) in zip(xs, ws, layers):
            y = F.conv2d(x, w, stride=stride, padding=k // 2 if stride == 1 else 0)
            if backward:
                torch.autograd.grad(y, x, torch.ones_like(y))
    return _repeat(run, seconds)[0]


def cpu_baseline(n_proj):
    """CPU timings of one whole iteration's arithmetic on this box's host cores ("port": the oracle's
    C restatement for the native hot ops — modulated convolution, brute-force rasterizer — and
    torch's CPU kernels for the networks the reference also runs through torch: the trained D / A /
    V / L / E nets of this package evaluated on CPU tensors, the VGG16 and discriminator convolution
    stacks as F.conv2d of their layer shapes).  ~25 s bounded sample: each piece is timed once or a
    few times and scaled by calls per iteration; elementwise glue is not included, so the figure
    still flatters the CPU."""
    import numpy as np
    from oracle import capi
    from oracle import geometry as og
    from gan2shape_amd import networks
    cores = _host_threads()
    os.environ["OMP_NUM_THREADS"] = str(cores)   # read when the oracle's OpenMP runtime starts
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    n = n_proj
    # (a) generator: convs[5], 512 -> 512 channels at 32x32, B = 1: 4.83 GFLOP, scaled by GFLOP
    x = rng.standard_normal((1, 512, 32, 32)).astype(np.float32)
    w = rng.standard_normal((512, 512, 3, 3)).astype(np.float32)
    s = np.ones((1, 512), np.float32)
    capi.modconv(x[:, :, :4, :4], w, s, 1.0, True, 0)
    t_conv, n_conv = _repeat(lambda: capi.modconv(x, w, s, 1.0, True, 0), 5.0)
    gflop_sample = 2 * 512 * 512 * 9 * 32 * 32 / 1e9
    t_gen_oracle = t_conv * (22.52 * n * 2) / gflop_sample   # 22.52 GFLOP / image forward, + data-gradient
    # the same generator arithmetic through torch's CPU convolution (what the reference's fallback path
    # runs, op/ + model.py:250-291 on CPU tensors): its five plain 3x3 layers, forward + data-gradient,
    # scaled by GFLOP to the whole generator
    gen = [(512, 512, 3, 1, 8), (512, 512, 3, 1, 16), (512, 512, 3, 1, 32), (256, 256, 3, 1, 64), (128, 128, 3, 1, 128)]
    gen_gflop = sum(2 * ci * co * 9 * h * h for ci, co, _, _, h in gen) / 1e9
    t_gen_torch = _cpu_conv_stack(gen, 1, True, 2.0) * (22.52 * n) / gen_gflop
    t_gen = min(t_gen_oracle, t_gen_torch)
    # (b) brute-force rasterizer (the reference algorithm), one 128x128 image per call
    S = 128
    geo = og.Geometry(S)
    geo.set_transform_matrices(np.array([[0.2, -0.3, 0.05, 0.01, 0.02, -0.03]], np.float32))
    depth = (1.0 + 0.05 * np.sin(np.arange(S)[None, :, None] / 9.0) * np.ones((1, S, S))).astype(np.float32)
    verts = geo.get_warped_3d_grid(depth).reshape(1, -1, 3)
    faces = og.get_face_idx(1, S, S)[0]
    t_raster, n_raster = _repeat(lambda: capi.render_depth(verts, faces, S, geo.K[0]), 5.0)
    # (c) trained nets on CPU tensors (torch kernels): forward + backward
    def net_time(name, batch):
        net = getattr(networks, name)(128)
        xin = torch.randn(batch, 3, 128, 128)

        def run():
            out = net(xin)
            torch.autograd.grad(out.sum(), list(net.parameters()))
        return _repeat(run, 0.5)[0]
    t_da = net_time("DepthNet", 1) + net_time("AlbedoNet", 1)
    t_vl1 = net_time("ViewpointNet", 1) + net_time("LightingNet", 1)
    t_vl9 = net_time("ViewpointNet", 1 + n) + net_time("LightingNet", 1 + n)
    t_e = net_time("OffsetEncoder", n)
    # (d) frozen convolution stacks: VGG16 to relu5_3 (lpips/pretrained_networks.py:97-135) and the
    # discriminator up to ftr_num = 4 (stylegan2-pytorch/model.py:700-750) at 128x128
    vgg = [(3, 64, 3, 1, 128), (64, 64, 3, 1, 128), (64, 128, 3, 1, 64), (128, 128, 3, 1, 64), (128, 256, 3, 1, 32),
           (256, 256, 3, 1, 32), (256, 256, 3, 1, 32), (256, 512, 3, 1, 16), (512, 512, 3, 1, 16), (512, 512, 3, 1, 16),
           (512, 512, 3, 1, 8), (512, 512, 3, 1, 8), (512, 512, 3, 1, 8)]
    disc = [(3, 128, 1, 1, 128), (128, 128, 3, 1, 128), (128, 256, 3, 2, 129), (128, 256, 1, 1, 64),
            (256, 256, 3, 1, 64), (256, 512, 3, 2, 65), (256, 512, 1, 1, 32), (512, 512, 3, 1, 32),
            (512, 512, 3, 2, 33), (512, 512, 1, 1, 16), (512, 512, 3, 1, 16), (512, 512, 3, 2, 17), (512, 512, 1, 1, 8)]
    t_vgg_f = _cpu_conv_stack(vgg, 2, False, 1.0) / 2          # per image, forward
    t_vgg_fb = _cpu_conv_stack(vgg, 2, True, 1.0) / 2          # per image, forward + data-gradient
    t_d_f = _cpu_conv_stack(disc, n, False, 1.0)
    t_d_fb = _cpu_conv_stack(disc, n, True, 1.0)
    # one iteration of each kind (trainer.py:99-109; model.py:95-280)
    t1 = t_raster + t_da + t_vl1 + t_vgg_f + t_vgg_fb
    t2 = n * t_raster + t_gen + t_d_f + t_d_fb + t_e
    t3 = (1 + n) * t_raster + t_da + t_vl9 + (1 + n) * (t_vgg_f + t_vgg_fb)
    its = 20.0 / (7 * t1 + 7 * t2 + 6 * t3)
    cpu = "?"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": its, "unit": "iters/s", "cores": cores, "kind": "port", "cpu": cpu,
            "seconds_per_iteration": {"step1": t1, "step2": t2, "step3": t3},
            "generator_fwd_bwd_seconds": {"oracle_c": t_gen_oracle, "torch_cpu": t_gen_torch},
            "sample": f"{n_conv} x oracle modconv 512->512@32x32 B=1 ({t_conv:.3f} s, scaled by GFLOP to the "
                      f"generator fwd+bwd at B={n}: {t_gen_oracle:.2f} s; torch-CPU convolutions of the same layers: "
                      f"{t_gen_torch:.2f} s; the faster one counts) + {n_raster} x brute-force raster of one 128x128 image "
                      f"({t_raster:.3f} s, x images/iteration) + torch-CPU fwd+bwd of the trained nets (D+A {t_da:.3f} s, "
                      f"V+L B=1 {t_vl1:.3f} s / B={1 + n} {t_vl9:.3f} s, E B={n} {t_e:.3f} s) + torch-CPU conv stacks of VGG16 "
                      f"({t_vgg_f:.3f} s fwd, {t_vgg_fb:.3f} s fwd+bwd per image) and the discriminator at B={n} "
                      f"({t_d_f:.3f} s fwd, {t_d_fb:.3f} s fwd+bwd); elementwise glue not included"}


def roofline_other(device):
    """The memory-bound custom kernels at their largest call of the workload: algorithmic bytes
     / HIP-event time on the launch stream, against 8 TB/s HBM3E."""
    import ctypes as C
    from gan2shape_amd import lib
    from gan2shape_amd.plugins import fused, upfirdn2d_op
    from gan2shape_amd.renderer import Renderer
    from gan2shape_amd.stylegan2 import make_kernel
    L = lib.load()
    out = []

    def timed(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    def add(kernel, nbytes, fn, note=None):
        t = timed(fn)
        row = {"kernel": kernel, "bound": "hbm", "achieved": nbytes / t / 1e9, "peak": 8000.0, "unit": "GB/s",
               "frac": nbytes / t / 8e12, "algorithmic_bytes": nbytes, "us": t * 1e6}
        if note:
            row["note"] = note
        out.append(row)

    # rasterizer, S = 128, B = 8: random views of a folded depth map (the step-2 / step-3 call)
    S, B = 128, 8
    R = Renderer({"rot_center_depth": 1.0, "fov": 10}, S, 0.9, 1.1, device=device)
    g = torch.Generator().manual_seed(0)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
    depth = (1.0 - 0.08 * torch.exp(-(xx ** 2 + yy ** 2) * 3) + 0.004 * torch.randn(S, S, generator=g))
    depth = depth.to(device)[None].expand(B, S, S).contiguous()
    view = (torch.randn(B, 6, generator=g) * torch.tensor([.3, .5, .1, .03, .03, .03])).to(device)
    R.set_transform_matrices(view)
    verts = R.get_warped_3d_grid(depth).reshape(B, -1, 3).contiguous()
    N, F = S * S, 2 * (S - 1) ** 2
    K = (C.c_float * 9)(*R._K9)
    ws = torch.empty(L.g2s_raster_workspace_bytes(B, N, F, S), dtype=torch.uint8, device=device)
    d = torch.empty(B, S, S, device=device)
    fi = torch.empty(B, 2 * S, 2 * S, dtype=torch.int32, device=device)
    ba = torch.empty(B, 2 * S, 2 * S, 3, device=device)
    gv = torch.empty(B, N, 3, device=device)

    def fwd(maps):
        lib.check(L.g2s_raster_depth_fwd(lib.ptr(verts), None, B, N, F, S, K, float(S), 2, 1, 0.1, 100.0, lib.ptr(d),
                                         lib.ptr(fi) if maps else None, lib.ptr(ba) if maps else None, lib.ptr(ws),
                                         ws.numel(), lib.stream()))
    latency = "latency / VALU bound (serial binning + list walk of the busiest tile), not HBM bound: DESIGN.md §4.1"
    add("g2s raster forward, no saved maps (B=8)", B * (S * S * 12 + S * S * 4), lambda: fwd(False), latency)
    add("g2s raster forward + saved maps (B=8)", B * (S * S * 16 + (2 * S) ** 2 * 16), lambda: fwd(True), latency)
    gd = (torch.randn(B, S, S, device=device) * (d < 50)).contiguous()
    add("g2s raster backward (B=8)", B * (S * S * 4 + (2 * S) ** 2 * 16 + S * S * 24),
        lambda: lib.check(L.g2s_raster_depth_bwd(lib.ptr(verts), None, lib.ptr(gd), lib.ptr(fi), lib.ptr(ba), B, N, F, S,
                                                 K, float(S), 2, lib.ptr(gv), lib.stream())), latency)
    # upfirdn2d: the generator's blur after the last up-convolution
    k = make_kernel([1, 3, 3, 1]).to(device)
    x = torch.randn(B * 128, 129, 129, 1, device=device)
    add("g2s upfirdn2d blur (8,128,129,129)->(8,128,128,128)", (x.numel() + B * 128 * 128 * 128) * 4,
        lambda: upfirdn2d_op.upfirdn2d(x, k, 1, 1, 1, 1, 1, 1, 1, 1))
    # fused bias + leaky-ReLU, forward and backward form
    a = torch.randn(B, 128, 128, 128, device=device)
    bias = torch.randn(128, device=device)
    e = a.new_empty(0)
    add("g2s fused_bias_act forward (8,128,128,128)", 2 * a.numel() * 4, lambda: fused.fused_bias_act(a, bias, e, 3, 0, 0.2, 2 ** 0.5))
    y = fused.fused_bias_act(a, bias, e, 3, 0, 0.2, 2 ** 0.5)
    add("g2s fused_bias_act backward (8,128,128,128)", 3 * a.numel() * 4, lambda: fused.fused_bias_act(a, e, y, 3, 1, 0.2, 2 ** 0.5))
    # Adam over the step-3 parameter set (lighting + viewpoint + depth + albedo nets: 22.6 M parameters in
    # one launch): read p, g, m, v, write p, m, v = 28 bytes per parameter
    from gan2shape_amd import networks
    from gan2shape_amd.optim import Adam
    nets = [networks.LightingNet(128), networks.ViewpointNet(128), networks.DepthNet(128), networks.AlbedoNet(128)]
    params = [p for n in nets for p in n.to(device).parameters()]
    for p in params:
        p.grad = torch.randn_like(p) * 1e-3
    opt = Adam(params, lr=1e-4, weight_decay=5e-4)
    add("g2s adam step (step-3 parameter set, %.1f M parameters)" % (sum(p.numel() for p in params) / 1e6),
        28 * sum(p.numel() for p in params), opt.step)
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks of this script through
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) as CHILD processes — this
    parent never initialises the GPU — and relay rank 0's JSON line.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this driver
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out_line in proc.stdout.splitlines():
        obj = None
        try:
            obj = json.loads(out_line)
        except ValueError:
            pass
        # a library may print to stdout too (gloo / RCCL banners; a bare number parses as JSON):
        # only a JSON OBJECT with the contract's keys is the result line
        if isinstance(obj, dict) and "metric" in obj and "value" in obj:
            line = out_line
        else:
            sys.stderr.write(out_line + "\n")
    if proc.returncode == 0 and line is None:
        sys.stderr.write("[bench] the ranks exited without printing a result line\n")
        return 1
    if line is not None:
        print(line, flush=True)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 40 = two cycles of the 7:7:6 mix; car128_joint: 53 = one cycle of 13:22:18)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n-proj", type=int, default=8)
    ap.add_argument("--workload", default="face128_n8", choices=sorted(WORKLOADS),
                    help="face128_n8 = the BASELINE metric's configuration (default); face256_fp16 = BASELINE "
                         "config 5 (256x256, fp16-operand MFMA), reported on its own line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python (no HIP graphs)")
    ap.add_argument("--deterministic", action="store_true",
                    help="analysis only: g2s_set_deterministic(1) — bit-reproducible iterations (fixed-order "
                         "partitions, fixed-point scatters); the default line is measured without it")
    ap.add_argument("--only", type=int, default=0, choices=[0, 1, 2, 3],
                    help="analysis only: time a single step kind instead of the 7:7:6 mix")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))   # before anything in this process touches the GPU
    global PATTERN
    joint = bool(WORKLOADS[args.workload].get("joint"))
    if joint:
        PATTERN = list(JOINT_PATTERN)
        if args.steps is None:
            args.steps = len(JOINT_PATTERN)   # a whole cycle: the timed region starts at step 1 of the mix
    if args.only:
        PATTERN = [args.only]
    if args.steps is None:
        args.steps = 40
    import gan2shape_amd  # noqa: F401
    from gan2shape_amd import sharding
    # RCCL ("nccl") on a multi-GPU node; G2S_DIST_BACKEND=gloo rehearses the N>1 path on one GPU
    rank, world, local_rank = sharding.init_distributed(os.environ.get("G2S_DIST_BACKEND", "nccl"))
    local_rank %= max(torch.cuda.device_count(), 1)
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # All work runs on one non-default stream: an eager backward on the legacy default stream makes
    # a later HIP-graph capture of the same autograd nodes fail (cross-stream AccumulateGrad sync).
    torch.cuda.set_stream(torch.cuda.Stream(device))

    from gan2shape_amd import lib, modconv as mc
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import GeneralizingTrainer2, Trainer
    lib.load()

    torch.manual_seed(0)  # identical random-init weights on every rank
    cfg = face_config(args.n_proj, args.workload)
    if args.deterministic:
        cfg["deterministic"] = True
    if os.environ.get("G2S_PARALLEL_NETS"):      # experiment switch: the two paired net passes
        cfg["parallel_nets"] = True              # as two branches of the captured graph — measured 3 % slower
    trainer = (GeneralizingTrainer2 if joint else Trainer)(GAN2Shape, cfg, device=device, capturable=not args.eager)
    image, latent = synthetic_sample(trainer.model, 1234 + rank, device)
    torch.manual_seed(1234 + rank)
    runner = JointStepRunner(trainer, image, latent, world) if joint else StepRunner(trainer, image, latent)

    # setup pass (not a benchmark step): one iteration of each kind creates the `collected`
    # hand-offs the trainer would have at this point, and triggers lazy library initialisation
    for kind in (1, 2, 3):
        runner.run(kind)
    eager_runner = runner
    if not args.eager:
        # record each step kind into a HIP graph; a replay is one training iteration
        from gan2shape_amd.graphs import GraphedJointSteps, GraphedSteps
        graphed = (GraphedJointSteps if joint else GraphedSteps)(trainer, image, latent)
        graphed.collected = dict(runner.collected)
        for kind in (1, 2, 3):
            if joint and kind == 1 and world > 1:
                graphed.loss[1] = runner.last_loss[1]       # eager (collective in the forward): GraphedJointRunner
                continue
            graphed.capture(kind)
            graphed.run(kind)   # a capture only records: one replay fills the hand-off buffers
        runner = GraphedJointRunner(graphed, eager_runner, world) if joint else GraphedRunner(graphed)
    for i in range(args.warmup):
        runner.run(PATTERN[i % len(PATTERN)])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # (flops, start event, end event) per g2s_modconv launch, recorded on its launch stream.
    # Eager mode: recorded inside the timed region.  Graph mode: HIP events cannot be recorded
    # between the nodes of a replayed graph, so the same launches are timed in an eager pass of
    # one 20-step cycle right after the timed region (same process, same shapes, same kernels).
    mc.PROFILE = [] if args.eager else None
    # G2S_BENCH_MARK=1 (profiling runs only): a kernel no training step uses brackets the timed
    # region in a rocprofv3 kernel trace (tools/window_trace.py keeps what lies between the two)
    mark = torch.ones(3, dtype=torch.int32, device=device) if os.environ.get("G2S_BENCH_MARK") else None
    barrier()
    if mark is not None:
        mark.bitwise_not()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        runner.run(PATTERN[i % len(PATTERN)])
    barrier()
    elapsed = time.perf_counter() - t0
    if mark is not None:
        mark.bitwise_not()
        torch.cuda.synchronize()
    # per-kind times (report only): a few iterations of each kind, outside the timed region —
    # recording events between graph launches costs ~1 ms per iteration
    for kind in sorted(set(PATTERN)):
        torch.cuda.synchronize()
        tk = time.perf_counter()
        for _ in range(10):
            runner.run(kind)
        torch.cuda.synchronize()
        runner.kind_ms[kind] = (time.perf_counter() - tk) / 10 * 1e3
    if not args.eager:
        mc.PROFILE = []
        # The eager loop is host-bound: an event pair around a small launch would also time the
        # host's gap before the launch.  Each iteration is therefore queued behind a device-side
        # spin (~25 ms) so that the host runs ahead and the events bracket back-to-back kernels.
        for kind in PATTERN:
            torch.cuda._sleep(50_000_000)
            eager_runner.run(kind)
        torch.cuda.synchronize()
    prof, mc.PROFILE = mc.PROFILE, None
    # what an event pair costs by itself: back-to-back pairs with nothing in between, queued behind a
    # spin like the launches above (a LOWER bound of what the pair adds around a kernel)
    pair_us = None
    if prof and rank == 0:
        torch.cuda._sleep(20_000_000)
        pairs = []
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        pair_us = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2] * 1e3

    # whole-job rate: iterations of all ranks / the slowest rank's barrier-bracketed time
    rate, elapsed = sharding.job_throughput(args.steps, elapsed, device)

    final_loss = {f"step{k}": float(v) for k, v in sorted(runner.last_loss.items())}
    if not all(math.isfinite(v) for v in final_loss.values()):
        raise RuntimeError(f"non-finite training loss after the timed region: {final_loss}")
    if rank == 0:
        kind_ms = {k: (v if not isinstance(v, list) else None) for k, v in runner.kind_ms.items()}
        times = [p[1].elapsed_time(p[2]) for p in prof]
        flops = sum(p[0] for p in prof)            # algorithmic: direct-convolution FLOP
        mfma = sum(p[4] for p in prof)             # what the matrix cores execute (Winograd F(2x2): 16 / 36 of it, F(4x4): 36 / 144)
        ms = sum(times)
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        executed = mfma / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        per_kernel = {}
        for name in ("direct", "winograd", "winograd4"):
            sel = [(p, t) for p, t in zip(prof, times) if p[5] == name]
            tk = sum(t for _, t in sel)
            if sel and tk > 0:
                per_kernel[name] = {"launches": len(sel), "share_of_time": tk / ms,
                                    "achieved": sum(p[0] for p, _ in sel) / (tk * 1e-3) / 1e12,
                                    "mfma_executed": sum(p[4] for p, _ in sel) / (tk * 1e-3) / 1e12,
                                    "avg_launch_us": tk * 1e3 / len(sel)}
        # executed / algorithmic FLOP of the convolution launches per training iteration (the profiled cycle is one
        # pass over PATTERN): what rocprofv3's per-iteration kernel time of the same launches is priced against
        cycle = max(len(PATTERN), 1)
        exec_gflop_it, algo_gflop_it = mfma / 1e9 / cycle, flops / 1e9 / cycle
        rocprof = None
        import glob
        rp = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_conv_rocprof.json")))
        if rp and prof and args.workload == "face128_n8" and not args.only and not args.deterministic:
            with open(rp[-1]) as f:
                conv_ms = json.load(f)["conv_ms_per_iteration"]
            rocprof = {"achieved": exec_gflop_it / conv_ms["total"], "frac": exec_gflop_it / conv_ms["total"] / F32_MFMA_PEAK_TFLOPS,
                       "conv_ms_per_iteration": conv_ms, "source": os.path.relpath(rp[-1], ROOT),
                       "is": "this run's executed FLOP per iteration / the convolution kernels' time per iteration by rocprofv3 "
                             "--kernel-trace inside the graph-replayed timed region of the SAME command, read from the committed "
                             "profiles file (tools/profile_round.sh) — not measured inside this run; the HIP-event figure above "
                             "carries ~5 us of event-pair cost per launch"}
        traffic, traffic_src = None, None
        pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_modconv_pmc.json")))
        if pmc:  # HBM bytes per launch from separate rocprofv3 --pmc passes of this command
            with open(pmc[-1]) as f:
                traffic = json.load(f)["traffic_bytes_per_launch"]
            traffic_src = os.path.relpath(pmc[-1], ROOT)
        # `frac` is the fraction of the fp32-MFMA peak the matrix cores actually EXECUTE (a Winograd F(2x2)
        # launch does 16 multiplications per 4 outputs where the direct convolution does 36, an F(4x4) launch
        # 36 per 16 instead of 144): the only figure that cannot exceed 1.  The direct-convolution
        # (algorithmic) rate is kept beside it.  Moving a layer from F(2x2) to F(4x4) LOWERS `frac` (1.78x
        # fewer FLOP for the same outputs, 1.2x less time) while the algorithmic rate and the step rate rise.
        roofline = {"bound": "mfma", "achieved": executed, "peak": F32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": executed / F32_MFMA_PEAK_TFLOPS,
                    "achieved_is": "FLOP the matrix cores execute / HIP-event time over every launch of the fp32-MFMA "
                                   "convolution kernels in one 20-step cycle (direct kernel: the algorithmic FLOP of the "
                                   "convolution; Winograd F(2x2): 16/36 of it; Winograd F(4x4), round 4: 36/144 of it — which is why this "
                                   "fraction fell from round 3 while the algorithmic rate and iters/s rose)",
                    "algorithmic": {"achieved": achieved, "frac": achieved / F32_MFMA_PEAK_TFLOPS,
                                    "is": "direct-convolution FLOP of the same launches / the same time (Winograd "
                                          "launches alone exceed 1.0 on this scale: not a roofline, a speed-up)"},
                    "mfma_executed": {"achieved": executed, "frac": executed / F32_MFMA_PEAK_TFLOPS},   # = achieved / frac (kept: round-2 name)
                    "executed_gflop_per_iteration": exec_gflop_it, "algorithmic_gflop_per_iteration": algo_gflop_it,
                    "rocprof": rocprof,
                    # the same with the cost of an EMPTY event pair (measured in this run) taken off every
                    # launch: still conservative — rocprofv3's own durations (profiles/) are shorter again
                    "event_pair_overhead_us": pair_us,
                    "net_of_event_pair": None if not pair_us or ms * 1e3 <= pair_us * len(prof) else {
                        "achieved": mfma / ((ms * 1e-3) - pair_us * 1e-6 * len(prof)) / 1e12,
                        "frac": mfma / ((ms * 1e-3) - pair_us * 1e-6 * len(prof)) / 1e12 / F32_MFMA_PEAK_TFLOPS},
                    "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_is": "HBM bytes per launch from SEPARATE rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  "tools/pmc_iter.py (same 20-step cycle), read from the committed profiles file — "
                                  "not measured inside this run",
                    "algorithmic_bytes_per_launch": (sum(p[3] for p in prof) / len(prof)) if prof else None,
                    "kernel": "the fp32-MFMA convolution kernels g2s::modconv_kernel (direct implicit GEMM; with "
                              "g2s::conv_bwd_kernel, the same body next to the weight-gradient GEMM of a layer) + "
                              "g2s::wino_kernel (Winograd F(2x2,3x3)) + g2s::wino4_kernel (Winograd F(4x4,3x3)): every launch of "
                              "the 20-step cycle — "
                              "generator, discriminator, VGG and the small trained nets",
                    "per_kernel": per_kernel,
                    "large_launches": _large(prof),
                    "launches": len(prof), "avg_launch_us": (ms * 1e3 / len(prof)) if prof else None,
                    "gflop_per_launch": (flops / 1e9 / len(prof)) if prof else None,
                    "timed_by": "hip events around each launch, " + (
                        "inside the timed region" if args.eager else
                        "eager pass of one 20-step cycle after the graph-replayed timed region")}
        out = {
            "metric": "GAN2Shape step iters/sec, faces 128\u00d7128 b=8, 1/2/4/8 MI355X",  # BASELINE.json's string
            "value": rate, "unit": "iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": WORKLOADS[args.workload]["dtype"], "data": "synthetic",
            "config": {"workload": args.workload if args.n_proj == 8 or not args.workload.startswith("face128")
                       else f"{args.workload.rsplit('_', 1)[0]}_n{args.n_proj}",
                       "image_size": cfg['image_size'], "gan_size": cfg['gan_size'], "n_proj_samples": args.n_proj,
                       "mfma_operands": cfg['mfma_operands'], "prior": cfg['prior_name'],
                       "step_mix": (f"step{args.only} only (analysis run, not the BASELINE metric)" if args.only else
                                    "13:22:18 (step1:step2:step3, main.py:141 joint training)" if joint else
                                    "7:7:6 (step1:step2:step3, main.py:148 stage 0)"),
                       "images_per_rank": 1,
                       "sharding": ("joint training: one image per rank and iteration, one shared model, ONE all-reduce (mean) of "
                                    "the optimiser's flat gradient bucket per step (17 / 55 / 90 MB), depth centre of step 1 over "
                                    "all ranks" if joint else "one image per rank, no collective"),
                       **({"deterministic": "g2s_set_deterministic(1): bit-reproducible iterations (analysis run, "
                                            "not the BASELINE metric)"} if args.deterministic else {})},
            "ms_per_step_kind": {f"step{k}": v for k, v in kind_ms.items()},
            "launch_mode": "eager" if args.eager else (
                "hipGraph replay: two segments per step kind (forward + backward + bucket pack | optimiser step) with the "
                "gradient all-reduce between them" if joint else "hipGraph replay (one graph per step kind)"),
            "final_loss": final_loss,
            "roofline": roofline,
            "roofline_other": roofline_other(device) if world == 1 and args.workload == "face128_n8"
            and not args.deterministic else None,
        }
        if cfg['mfma_operands'] == 'f16':
            # priced against the dense fp16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s); every launch of this workload's
            # frozen networks runs fp16 operands / fp32 accumulation, the trained nets' launches stay fp32
            f16_peak = 2500.0
            r = out["roofline"]
            r["fp32_peak_figures"] = {"peak": r["peak"], "frac": r["frac"]}
            r["peak"], r["frac"] = f16_peak, r["achieved"] / f16_peak
            r["peak_note"] = ("dense fp16 MFMA peak; the kernels are bound by the fp32 -> fp16 operand conversion (activations stay "
                              "fp32 in HBM, DESIGN.md section 4.4), not by the matrix pipe")
        if world == 1 and not args.no_cpu_baseline and args.workload == "face128_n8":
            out["cpu_baseline"] = cpu_baseline(args.n_proj)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
