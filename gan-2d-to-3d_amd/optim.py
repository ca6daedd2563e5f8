"""torch.optim.Adam(params, lr, betas, eps, weight_decay) as the reference uses it
(GAN2Shape/trainer.py:163-171: lr 1e-4, betas (0.9, 0.999), classic L2 weight decay 5e-4, amsgrad
off) with the whole step as ONE launch of libg2s (g2s_adam_step): every parameter tensor of the
optimiser in one grid, the step counters on the device (a captured HIP graph replays the right bias
correction).  CUDA float32 parameters only — CPU models take torch's own Adam (trainer.default_optimizer)."""
import ctypes as C

import numpy as np
import torch

from gan2shape_amd import lib as _lib

MAX_TENSORS = 128   # G2S_ADAM_MAX_TENSORS: gradient pointers of one launch travel as kernel arguments


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        # capturable: the step counters live on the device (graphs.GraphedSteps checks the flag)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True))
        self._tables = {}      # ids of the parameters of one launch -> (device table, device chunk prefix, n_chunks)

    def _table(self, params):
        """Device copy of the g2s_adam_tensor array + chunk prefix of these parameters: everything
        that persists (parameter, moments, counters).  Built once per distinct parameter set, by a
        synchronous upload — so the first step of a set must run eagerly, not inside a graph capture
        (GraphedSteps warms every step kind up before it records)."""
        key = tuple(id(p) for p in params)
        ent = self._tables.get(key)
        if ent is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("g2s Adam: the first step over a set of parameters builds its device table "
                                   "and cannot be captured: run one eager step first")
            chunk = int(_lib.load().g2s_adam_chunk())
            rows = [(p.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr(),
                     p.numel(), self.state[p]['step'].data_ptr(), self.state[p]['ticket'].data_ptr()) for p in params]
            prefix = np.zeros(len(rows) + 1, dtype=np.int32)
            prefix[1:] = np.cumsum([(r[3] + chunk - 1) // chunk for r in rows])
            dev = params[0].device
            ent = (torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev), torch.from_numpy(prefix).to(dev),
                   int(prefix[-1]))
            self._tables[key] = ent
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.load()
        for group in self.param_groups:
            params = [p for p in group['params'] if p.grad is not None]
            for p in params:
                _lib.require_cuda(p, p.grad)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("g2s Adam: contiguous float32 parameters only")
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                if not self.state[p]:
                    self.state[p]['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    self.state[p]['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    # device-side step count (one per parameter, as torch keeps it) + arrival ticket
                    self.state[p]['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
                    self.state[p]['ticket'] = torch.zeros((), dtype=torch.int32, device=p.device)
            b1, b2 = group['betas']
            for i in range(0, len(params), MAX_TENSORS):
                part = params[i:i + MAX_TENSORS]
                table, prefix, n_chunks = self._table(part)
                grads = (C.c_void_p * len(part))(*[p.grad.data_ptr() for p in part])
                _lib.check(L.g2s_adam_step(_lib.ptr(table), _lib.ptr(prefix), grads, len(part), n_chunks,
                                           float(group['lr']), float(b1), float(b2), float(group['eps']),
                                           float(group['weight_decay']), _lib.stream()))
        return loss
