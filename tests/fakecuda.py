"""Run the `-m gpu` test bodies against the simulator build (test infrastructure, opt-in: CACO_GPU_ON_SIM=1).

The GPU suite is the parity gate, and a case that was never executed can fail for reasons that have nothing to do with
a kernel (a typo, a stale signature).  With no GPU to run it on, this shim lets the same test files run here:

* `CACO_LIB_PATH` points at tools/wavesim's libcaco_sim.so (device memory = host memory, streams are ignored);
* every `device='cuda…'` in a torch call is rewritten to the CPU by a TorchFunctionMode, `Tensor.cuda()` is the identity,
  `Tensor.is_cuda` / `Tensor.device` answer as a GPU tensor would;
* the handful of `torch.cuda.*` calls the package and the tests make (streams, events, synchronize, device context)
  become no-ops.

Nothing here is imported by `cacophony_amd/`; the product path still refuses to run without a GPU.  What a pass here says:
the test's Python and the kernels' arithmetic are right at that size under tools/wavesim's model.  It says nothing about
the hardware; the `-m gpu` run on MI355X remains the gate.  Full-size cases take hours at ≈5 GFLOP/s: use --timeout.
"""
import contextlib
import os

import torch
from torch.overrides import TorchFunctionMode

FAKE = torch.device("cuda", 0)


def _is_cuda_dev(x):
    if isinstance(x, torch.device):
        return x.type == "cuda"
    if isinstance(x, str):
        return x == "cuda" or x.startswith("cuda:")
    return False


def _fix(x):
    if _is_cuda_dev(x):
        return torch.device("cpu")
    if isinstance(x, (list, tuple)) and any(_is_cuda_dev(y) for y in x):
        return type(x)(_fix(y) for y in x)
    return x


class _Mode(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        qual = getattr(func, "__qualname__", "")
        if func is torch.Tensor.cuda:
            return args[0]
        if func is torch.Tensor.is_cuda.__get__:
            return True
        if func is torch.Tensor.device.__get__:
            return FAKE
        if func is torch.Tensor.pin_memory or func is torch.Tensor.is_pinned:
            return args[0] if func is torch.Tensor.pin_memory else False
        if name == "record_stream":
            return None
        if func is torch.device:                      # constructing a device object needs no GPU; keep what was asked for
            return func(*args, **kwargs)
        args = tuple(_fix(a) for a in args)
        kwargs = {k: (False if k in ("pin_memory", "non_blocking") else _fix(v)) for k, v in kwargs.items()}
        return func(*args, **kwargs)


class _Stream:
    cuda_stream = 0
    device = FAKE

    def __init__(self, *a, **k):
        pass

    def wait_stream(self, s):
        pass

    def wait_event(self, e):
        pass

    def record_event(self, e=None):
        return e or _Event()

    def synchronize(self):
        pass

    def query(self):
        return True

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Event:
    def __init__(self, *a, **k):
        import time
        self._t = time.perf_counter()

    def record(self, stream=None):
        import time
        self._t = time.perf_counter()

    def wait(self, stream=None):
        pass

    def synchronize(self):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return (other._t - self._t) * 1e3


_mode = None


def install():
    """Idempotent.  Call before `cacophony_amd` is imported (CACO_LIB_PATH is read at import)."""
    global _mode
    if _mode is not None:
        return
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "tools", "wavesim"))
    import build_sim
    # CACO_SIM_LIB: a variant build of the simulator library (e.g. tools/wavesim/libcaco_sim_skew.so) under the GPU suite's bodies
    path = os.environ.get("CACO_SIM_LIB") or build_sim.build(verbose=False)      # rebuilds tools/wavesim/libcaco_sim.so if it is stale
    os.environ["CACO_LIB_PATH"] = path
    os.environ["CACO_ALLOW_VARIANT_LIB"] = "1"          # _lib.load() refuses any library but the product's without it
    if "cacophony_amd._lib" in sys.modules:
        sys.modules["cacophony_amd._lib"].LIB_PATH = path
    c = torch.cuda
    one = _Stream()
    c.is_available = lambda: True
    c.device_count = lambda: 1
    c.current_device = lambda: 0
    c.set_device = lambda d: None
    c.synchronize = lambda d=None: None
    c.current_stream = lambda d=None: one
    c.default_stream = lambda d=None: one
    c.Stream = _Stream
    c.Event = _Event
    c.stream = lambda s: contextlib.nullcontext()
    c.device = lambda d: contextlib.nullcontext()
    c.empty_cache = lambda: None
    c.get_device_name = lambda d=None: "wavesim (CPU)"
    c.mem_get_info = lambda d=None: (1 << 34, 1 << 34)
    c.memory_allocated = lambda d=None: 0
    c.max_memory_allocated = lambda d=None: 0
    c.reset_peak_memory_stats = lambda d=None: None
    # the C getters of plain tensors do not pass through the mode: shadow them on the Python class
    torch.Tensor.device = property(lambda self: FAKE)
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.Tensor.get_device = lambda self: 0
    real_generator = torch.Generator
    torch.Generator = lambda device="cpu": real_generator("cpu")
    _mode = _Mode()
    _mode.__enter__()
