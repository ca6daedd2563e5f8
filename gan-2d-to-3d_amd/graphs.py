"""HIP-graph execution of the training iterations.

One GAN2Shape iteration (zero_grad, forward_stepK, backward, Adam.step — trainer.py:99-109) is
1000-1500 small kernel launches; on MI355X the eager loop is host-bound (the GPU is busy ~2/3 of
the wall time at batch 1).  Each step kind is therefore captured ONCE into a hipGraph
(torch.cuda.CUDAGraph: kernels of libg2s.so launch on torch's capture stream, so they are recorded
like any other op) and replayed per iteration.  A replay is exactly one iteration of the reference
loop: same kernels, same order, same arithmetic; random draws come from the graph-registered
device generator, so successive replays draw fresh samples.

Constraint (PyTorch/HIP): every eager backward of these networks that precedes a capture must have
run on a non-default stream (torch.cuda.set_stream(torch.cuda.Stream()) at program start);
otherwise autograd's AccumulateGrad nodes stay bound to the legacy default stream and the capture
aborts.

Static buffers: `image`, `latent` (copy a new image in to reuse the graphs), and the `collected`
hand-offs — graph k+1 reads the output tensors of graph k in place.
"""
import torch

from . import zeropool


class GraphedSteps:
    def __init__(self, trainer, image, latent, warmup=3):
        self.t = trainer
        self.model = trainer.model
        self.image = image.clone()
        self.latent = latent.clone()
        self.graphs = {}
        self.loss = {}
        self.collected = {1: None, 2: None, 3: None}
        self._static_src = {}
        self.warmup = warmup
        for opt in (trainer.optim_step1, trainer.optim_step2, trainer.optim_step3):
            for group in opt.param_groups:
                if not group.get('capturable', False):
                    raise RuntimeError("GraphedSteps needs Adam(capturable=True): build the trainer "
                                       "with Trainer(..., capturable=True)")

    def _iteration(self, kind, src):
        optim = getattr(self.t, f'optim_step{kind}')
        loss, collected = getattr(self.model, f'forward_step{kind}')(
            self.image, self.latent, src, n_proj_samples=self.t.n_proj_samples)
        loss.backward()
        optim.step()
        zeropool.end()
        return loss, collected

    def _source(self, kind):
        return {1: None, 2: self.collected[1], 3: self.collected[2]}[kind]

    def capture(self, kind, warmup=None):
        """Warm up `warmup` (default: the constructor's) eager iterations on a side stream (library
        handles, lazily built caches), then capture one iteration (recorded, not executed).
        Warm-up iterations are real training iterations; callers that count iterations must count
        them (return value)."""
        warmup = self.warmup if warmup is None else int(warmup)
        if warmup < 1:
            raise ValueError("capture needs at least one eager warm-up iteration")
        optim = getattr(self.t, f'optim_step{kind}')
        src = self._source(kind)
        if kind > 1 and src is None:
            raise RuntimeError(f"capture step {kind - 1} first")
        if torch.cuda.current_stream() == torch.cuda.default_stream():
            # an eager backward on the legacy default stream binds the parameters' AccumulateGrad
            # nodes to it; capturing the same autograd graph afterwards aborts inside
            # hipStreamEndCapture (a host segfault, not an exception)
            raise RuntimeError("GraphedSteps.capture must run with a non-default current stream: call "
                               "torch.cuda.set_stream(torch.cuda.Stream()) before the first backward "
                               "of the networks (see graphs.py)")
        # warm-up (side stream) and capture (capture stream) run backward on different non-default
        # streams than the parameters' AccumulateGrad nodes were created on: expected here
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                optim.zero_grad(set_to_none=True)
                warm_loss, warm_collected = self._iteration(kind, src)
        torch.cuda.current_stream().wait_stream(s)
        optim.zero_grad(set_to_none=True)
        self.graphs[kind], loss, collected = self._record(kind, src)
        self.loss[kind] = loss.detach()
        self.collected[kind] = collected
        # A capture records, it does not run: the graph's static outputs (the loss, the hand-off
        # to the next step kind) are filled with the results of the last warm-up iteration, so they
        # are valid even if no replay follows (a block that asks for no more than `warmup`
        # iterations).
        with torch.no_grad():
            self.loss[kind].copy_(warm_loss.detach())
            for dst, srct in zip(collected or (), warm_collected or ()):
                if torch.is_tensor(dst):
                    dst.copy_(srct.detach())
        return warmup  # iterations actually executed

    def _record(self, kind, src):
        """Capture one iteration -> (what run() replays, loss, collected)."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss, collected = self._iteration(kind, src)
        return g, loss, collected

    def run(self, kind):
        """One training iteration of kind `kind` (graph replay)."""
        self.graphs[kind].replay()
        return self.loss[kind]

    def set_sample(self, image, latent):
        self.image.copy_(image)
        self.latent.copy_(latent)


class GraphedJointSteps(GraphedSteps):
    """Joint (data-parallel) training iterations as TWO captured segments with the collective between them:

        segment A   forward_stepK + backward + GradBucket.pack      (the step's gradients -> the flat bucket)
        eager       ONE all-reduce (mean) of the bucket over RCCL   (skipped by a single process)
        segment B   optimiser step reading the bucket's views       (GradBucket.bind: nothing is copied back)

    Every address is fixed (the bucket is persistent, the gradients of segment A live in the graph's pool), so
    a replayed iteration is  A.replay(); all_reduce(bucket); B.replay().  The forward itself must hold no
    collective: steps 2 / 3 always qualify (one image at a time, centred by its own mean at any W —
    GeneralizingTrainer2); step 1 qualifies when the depth centre is local (`model.batch_mean is None`: a
    single process, or one image per rank WITHOUT the whole-batch centre) and is refused otherwise — the
    trainer then runs it eagerly (sharding.global_mean is a differentiable all-reduce inside the forward)."""

    def __init__(self, trainer, image, latent, warmup=3):
        super().__init__(trainer, image, latent, warmup)
        from .sharding import bucket_of
        self.buckets = {k: bucket_of([p for g in getattr(trainer, f'optim_step{k}').param_groups for p in g['params']])
                        for k in (1, 2, 3)}

    def _forward_backward(self, kind, src):
        loss, collected = getattr(self.model, f'forward_step{kind}')(
            self.image, self.latent, src, n_proj_samples=self.t.n_proj_samples)
        loss.backward()
        self.buckets[kind].pack()
        return loss, collected

    def _update(self, kind):
        self.buckets[kind].bind()
        getattr(self.t, f'optim_step{kind}').step()
        zeropool.end()

    def _iteration(self, kind, src):
        if kind == 1 and self.model.batch_mean is not None:
            raise RuntimeError("GraphedJointSteps: step 1 with a whole-batch depth centre holds a collective in its forward; run it eagerly")
        loss, collected = self._forward_backward(kind, src)
        self.buckets[kind].all_reduce_mean()
        self._update(kind)
        return loss, collected

    def _record(self, kind, src):
        if kind == 1 and self.model.batch_mean is not None:
            raise RuntimeError("GraphedJointSteps: step 1 with a whole-batch depth centre holds a collective in its forward; run it eagerly")
        a, b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(a):
            loss, collected = self._forward_backward(kind, src)
        with torch.cuda.graph(b, pool=a.pool()):
            self._update(kind)
        return (a, b), loss, collected

    def run(self, kind):
        a, b = self.graphs[kind]
        a.replay()
        self.buckets[kind].all_reduce_mean()
        b.replay()
        return self.loss[kind]

    def set_source(self, kind, collected):
        """Copy a hand-off into the static tensors the captured graph of step `kind` reads (the joint trainer
        hands every image its own slice of the batched step 1)."""
        cur = self._static_src.get(kind)
        if cur is None:
            self._static_src[kind] = cur = tuple(t.detach().clone() if torch.is_tensor(t) else t for t in collected)
            self.collected[kind - 1] = cur
            return
        with torch.no_grad():
            for dst, srct in zip(cur, collected):
                if torch.is_tensor(dst):
                    dst.copy_(srct)
