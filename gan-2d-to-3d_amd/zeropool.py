"""One cleared pool per training step for the outputs of launches that ADD partial sums.

The split-K slices of the direct convolution kernel (small layers: too few tiles for 256 CUs) and
the polyphase classes that leave holes add into an output that must start at zero.  The library
clears such an output itself (hipMemsetAsync) — one more graph node per launch, 36 of step 1's 319.
Here each step kind clears ONE buffer at its start and hands out slices (g2s_modconv_ex,
y_is_zero = 1).  The buffer is a fresh tensor of the caching allocator per step (sized by what the
previous step of that kind asked for), slices are tensors on its storage (they keep it alive; each has its
own autograd version counter), nothing is ever handed out twice: no lifetime hazards, and inside a captured HIP graph it lives in the graph's
private pool like every other tensor of the step.  A request that does not fit falls back to the
library's own clear."""
import torch

_state = {"buf": None, "off": 0, "key": None, "capturing": False}
_demand = {}          # step kind -> floats requested during the last step of that kind
_want = {}            # running count of the current step


def _pad(n):
    return (n + 63) & ~63


def begin(key, device):
    """Start of a training step of kind `key`: clear a pool as large as that kind's last demand."""
    cur = _state["key"]
    if cur is not None:
        _demand[cur] = _want.get(cur, 0)
    _want[key] = 0
    n = _demand.get(key, 0)
    _state["key"] = key
    _state["off"] = 0
    _state["capturing"] = _capturing(device)
    _state["buf"] = torch.zeros(n, dtype=torch.float32, device=device) if n else None


def _capturing(device):
    return torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()


def end():
    """End of the step (the trainers call it after backward + optimiser step, forward_step1(eval=True) before
    it returns): nothing outside a step is ever served from a step's pool — in particular not from a pool that
    lives in a captured graph's private memory, which every replay clears again."""
    cur = _state["key"]
    if cur is not None:
        _demand[cur] = _want.get(cur, 0)
    _state.update(buf=None, off=0, key=None, capturing=False)


def take(shape, device):
    """A zero-filled tensor of `shape` carved from the step's pool, or None (no step active / first
    step of its kind / does not fit)."""
    key = _state["key"]
    if key is None:
        return None
    n = 1
    for d in shape:
        n *= int(d)
    _want[key] = _want.get(key, 0) + _pad(n)
    buf, off = _state["buf"], _state["off"]
    if buf is None or buf.device != device or off + n > buf.numel():
        return None
    if _capturing(device) != _state["capturing"]:
        return None     # a pool begun under capture is graph memory (and vice versa): never hand it across
    _state["off"] = off + _pad(n)
    # a tensor of its own on the pool's storage, NOT a view of `buf`: views share one version counter, so an
    # autograd-visible in-place op on one slice would invalidate every saved slice of the step
    return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(buf.untyped_storage(), buf.storage_offset() + off,
                                                                   tuple(int(d) for d in shape))


def zeros(shape, device):
    """torch.zeros(shape) for a step-local accumulator: a slice of the step's pool when one is active (no
    fill launch of its own), else a fresh cleared tensor."""
    if isinstance(shape, int):
        shape = (shape,)
    t = take(tuple(shape), device)
    return t if t is not None else torch.zeros(shape, dtype=torch.float32, device=device)
