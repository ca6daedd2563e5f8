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
        g = torch.cuda.CUDAGraph()
        optim.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            loss, collected = self._iteration(kind, src)
        self.graphs[kind] = g
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

    def run(self, kind):
        """One training iteration of kind `kind` (graph replay)."""
        self.graphs[kind].replay()
        return self.loss[kind]

    def set_sample(self, image, latent):
        self.image.copy_(image)
        self.latent.copy_(latent)
