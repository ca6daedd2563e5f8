"""Forward-pass companion of tools/census_glue.py: which source line of this package issues each small
aten op of a step?  One eager iteration under a TorchDispatchMode; every aten op that is not a pure view
is attributed to the innermost frame inside gan-2d-to-3d_amd/ (backward ops run on the autograd thread
and are listed by census_glue.py instead).   python tools/census_frames.py <kind> [top]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
import gan2shape_amd  # noqa
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer

VIEWS = {"view", "reshape", "_unsafe_view", "expand", "slice", "select", "t", "transpose", "permute", "unsqueeze",
         "squeeze", "detach", "alias", "as_strided", "empty", "empty_like", "empty_strided", "unbind", "split",
         "split_with_sizes", "_reshape_alias", "new_empty", "new_empty_strided", "lift_fresh", "is_same_size",
         "unfold", "narrow", "chunk", "size", "stride", "sym_size", "_local_scalar_dense", "resize_", "set_",
         "record_stream", "is_pinned", "_to_copy", "zeros", "ones", "zeros_like", "ones_like", "new_zeros", "new_ones"}
kind = int(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 80
PKG = os.path.realpath(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gan-2d-to-3d_amd"))
agg = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            where = "<outside the package>"
            for fr in reversed(traceback.extract_stack()):
                if os.path.realpath(fr.filename).startswith(PKG):
                    where = f"{os.path.relpath(fr.filename, PKG)}:{fr.lineno} {fr.name}"
                    break
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:3]
            agg[(name, str(shapes), where)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
torch.manual_seed(0)
tr = Trainer(GAN2Shape, bench.face_config(8), device=dev)
image, latent = bench.synthetic_sample(tr.model, 1234, dev)
r = bench.StepRunner(tr, image, latent)
for k in (1, 2, 3):
    r.run(k)
r.run(kind)
torch.cuda.synchronize()
with Census():
    r.run(kind)
torch.cuda.synchronize()
by_line = collections.Counter()
for (name, shapes, where), n in agg.items():
    by_line[where] += n
print(f"step {kind}: {sum(agg.values())} non-view aten ops on the calling thread (forward + optimiser)")
print("-- by source line")
for where, n in by_line.most_common(top):
    ops = collections.Counter()
    for (name, shapes, w), c in agg.items():
        if w == where:
            ops[name] += c
    print(f"{n:4d}  {where:60s} " + ", ".join(f"{k} x{v}" for k, v in ops.most_common(8)))
