"""
Execution-mode plumbing for the ``hip`` mode.

Counterpart of the reference's ``sailfish/kernel/system.py:99-183``
(``get_array_module``, ``execution_context``, ``num_devices``, ``measure_time``)
for a mode that sits beside ``cpu`` / ``omp`` / ``gpu``: arrays are torch tensors
in MI355X HBM (torch-ROCm is used for memory, streams and ``torch.distributed``
only; all arithmetic on the hot path is in libsfhip.so).

Modes understood here:
  ``hip``   device arrays on the current ROCm device, kernels from libsfhip.so
  ``host``  torch CPU arrays; NO kernels exist for it in this package -- it is
            there so that the decomposition / halo / reduction logic can be
            exercised by multi-process gloo tests that inject their own kernel
            backend.
"""
import contextlib
import time

MODES = ("hip", "host")


class ArrayModule:
    """
    The handful of ``xp.*`` calls the reference's solvers make
    (``xp.zeros``, ``xp.array``, ``xp.linspace``, ``xp.zeros_like``;
    cbdiso_2d.py:89-94,118), returning float64 torch tensors on `device`.
    """

    float64 = None  # filled in __init__ (torch import is deferred)

    def __init__(self, device):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.float64 = torch.float64
        self.ndarray = torch.Tensor

    def _dev(self):
        if self.device.type == "cuda" and self.device.index is None:
            return self.torch.device("cuda", self.torch.cuda.current_device())
        return self.device

    def zeros(self, shape, dtype=None):
        if isinstance(shape, int):
            shape = (shape,)
        return self.torch.zeros(tuple(shape), dtype=dtype or self.float64, device=self._dev())

    def zeros_like(self, a):
        return self.torch.zeros_like(a)

    def empty(self, shape):
        return self.torch.empty(tuple(shape), dtype=self.float64, device=self._dev())

    def array(self, a):
        """Copy `a` (numpy array, nested list or tensor) onto the device as float64."""
        import numpy as np

        if isinstance(a, self.torch.Tensor):
            return a.to(device=self._dev(), dtype=self.float64, copy=True).contiguous()
        host = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
        return self.torch.from_numpy(host).to(self._dev(), copy=True)

    def asarray(self, a):
        if isinstance(a, self.torch.Tensor) and a.device == self._dev() and a.dtype == self.float64:
            return a
        return self.array(a)

    def linspace(self, start, stop, num):
        # numpy's linspace on the host, then copied: bit-identical to the
        # reference's xp.linspace for cpu/omp modes (cbdiso_2d.py:89-90)
        import numpy as np

        return self.array(np.linspace(start, stop, num))


def get_array_module(mode):
    """``xp`` for `mode` (reference: system.py:99-119)."""
    if mode == "hip":
        return ArrayModule("cuda")
    if mode == "host":
        return ArrayModule("cpu")
    raise ValueError(f"unknown execution mode {mode}, must be one of {MODES}")


class _HipDevice:
    """Re-enterable device guard (a patch keeps one and enters it for every launch)."""

    def __init__(self, device_id):
        self.device_id = device_id
        self._stack = []

    def __enter__(self):
        import torch

        if torch.cuda.current_device() == self.device_id:  # the usual case (one process per GPU): nothing to switch
            self._stack.append(None)
            return self
        guard = torch.cuda.device(self.device_id)
        guard.__enter__()
        self._stack.append(guard)
        return self

    def __exit__(self, *exc):
        guard = self._stack.pop()
        return None if guard is None else guard.__exit__(*exc)


def execution_context(mode, device_id=None):
    """
    Context manager that makes `device_id` current (reference: system.py:122-135).

    torch's device guard calls hipSetDevice on the shared HIP runtime, which is
    also what libsfhip.so launches on.
    """
    if mode == "host":
        return contextlib.nullcontext()
    if mode == "hip":
        return _HipDevice(device_id or 0)
    raise ValueError(f"unknown execution mode {mode}")


def num_devices(mode):
    """Reference: system.py:138-145."""
    if mode == "host":
        return 1
    if mode == "hip":
        import torch

        return torch.cuda.device_count()
    raise ValueError(f"unknown execution mode {mode}")


def synchronize(mode, all_devices=True):
    """
    Block until device work is done.  The reference only synchronises the
    current device at the end of a fold (system.py:179-183); we synchronise
    every device that holds a patch so the Mzps clock is honest.
    """
    if mode != "hip":
        return
    import torch

    if all_devices:
        for d in range(torch.cuda.device_count()):
            torch.cuda.synchronize(d)
    else:
        torch.cuda.synchronize()


@contextlib.contextmanager
def measure_time(mode="host"):
    """
    ``with measure_time(mode) as t: ...; seconds = t()`` (reference: system.py:163-183).
    Device work is drained before the clock starts and before it is read.
    """
    synchronize(mode)
    start = time.perf_counter()
    end = []

    def elapsed():
        return (end[0] if end else time.perf_counter()) - start

    try:
        yield elapsed
    finally:
        synchronize(mode)
        end.append(time.perf_counter())


def to_host(a):
    """Device array -> numpy (reference: subdivide.py:6-10 uses cupy's ``.get()``)."""
    try:
        return a.detach().cpu().numpy()
    except AttributeError:
        return a
