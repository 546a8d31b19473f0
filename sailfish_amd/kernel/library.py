"""
``Library``: the kernel-library object of the ``hip`` execution mode.

Same surface as the reference's ``sailfish/kernel/library.py:90-249``:

    lib = Library(mode="hip")
    lib.cbdiso_2d_advance_rk[(ni, nj)](xl, xr, yl, yr, conserved0, primitive1, ...)

``lib.<symbol>`` is a ``Kernel``; ``kernel[shape]`` checks the shape's length
against the kernel's rank (``ValueError`` otherwise, library.py:148-157) and
returns a ``KernelInvocation`` whose ``__call__`` prepends the shape integers to
the positional arguments (library.py:110).  What differs:

  * nothing is compiled at run time: the kernels are hand-written HIP in
    ``sailfish_amd/csrc`` built ahead of time into ``libsfhip.so``; a missing
    library is a hard ``KernelError`` -- there is no CPU fallback;
  * array arguments are float64 torch tensors in device memory and are passed
    as raw device pointers; the launch goes to torch's current stream on the
    tensor's device;
  * the C entry points return a status, raised here as ``KernelError``
    (the reference's kernels are ``void`` and cannot report anything).
"""
import ctypes
import os
from logging import getLogger

from .parse_api import parse_api

logger = getLogger(__name__)

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))
HEADER_PATH = os.path.join(_REPO, "include", "sfhip.h")
LIBRARY_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libsfhip.so")


class KernelError(RuntimeError):
    pass


class KernelInvocation:
    """A kernel whose execution shape is fixed and is ready to be launched."""

    def __init__(self, kernel, shape):
        self.kernel = kernel
        self.shape = tuple(int(n) for n in shape)

    def __call__(self, *args):
        self.bind(*args)()

    def bind(self, *args):
        """
        Validate and marshal now, launch later: returns a zero-argument callable that performs
        only the C call.  (Lets a caller put its timing events right around the launch.)
        """
        lib = self.kernel.lib
        sym = self.kernel.symbol
        args = list(self.shape) + list(args)
        if lib.debug:
            validate_types(args, sym.args, sym.name, lib)
            validate_constraints(args, sym.args, sym.name)
        elif len(args) != len(sym.args):
            raise arglen_error(sym.name, args, sym.args)
        cargs = to_ctypes(args, sym.args)
        cfunc = self.kernel.cfunc

        def fire():
            lib.bind_stream(args)
            status = cfunc(*cargs)
            if status != 0:
                raise KernelError(f"{sym.name} failed ({status}): {lib.last_error()}")

        return fire


class Kernel:
    """``kernel[shape]`` -> :class:`KernelInvocation` (reference: library.py:138-157)."""

    def __init__(self, lib, symbol, cfunc):
        self.lib = lib
        self.symbol = symbol
        self.cfunc = cfunc

    def __getitem__(self, shape):
        if type(shape) == int:
            return self[(shape,)]
        if len(shape) != self.symbol.rank:
            raise ValueError(
                f"incompatible shape {shape} for kernel "
                f"{self.symbol.name} with rank {self.symbol.rank}"
            )
        return KernelInvocation(self, shape)


_CTYPES = {
    "int": ctypes.c_int,
    "double": ctypes.c_double,
    "double*": ctypes.c_void_p,
}


class Library:
    """
    Loads ``libsfhip.so`` and exposes its kernels with the reference's calling
    convention (``sailfish.kernel.library.Library``, library.py:165-185).  Nothing is
    compiled in this mode, but `code` -- the kernel source text a reference solver reads
    from the ``.c`` file beside it and hands over -- still says what the caller is about
    to call: every ``PUBLIC void`` kernel it declares must exist in the library with the
    same argument types and names in the same order, or construction fails with
    ``KernelError`` (a binding gets that check at run time, without the reference's
    sources having to be anywhere near the library's tests).  `define_macros`: the
    srhd_2d kernels' ``PLM_THETA`` / ``MACH_CEILING`` are settings of the prebuilt
    library (``sfhip_srhd_2d_set_macros``) and are applied; any other macro cannot be
    honoured by prebuilt code and is refused.  `name` is accepted and unused.
    """

    def __init__(self, code=None, mode="hip", name="sfhip", debug=True, define_macros=None, path=None):
        if mode != "hip":
            raise ValueError(f"sailfish_amd.kernel.Library only implements mode 'hip', got {mode!r}")
        # torch must own the process's HIP runtime before libsfhip.so is mapped,
        # so that both resolve to the same libamdhip64 (same SONAME) and device
        # pointers / streams can be shared.
        import torch  # noqa: F401

        self.mode = mode
        self.debug = debug
        self.cpu_mode = False
        self.path = path or os.environ.get("SFHIP_LIBRARY", LIBRARY_PATH)
        if not os.path.exists(self.path):
            raise KernelError(
                f"HIP kernel library not found at {self.path}; build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)"
            )
        with open(HEADER_PATH) as f:
            self.api = parse_api(f.read())
        self.module = ctypes.CDLL(self.path)
        self._kernels = {}
        for sym in self.api.values():
            try:
                cfunc = getattr(self.module, sym.name)
            except AttributeError:
                raise KernelError(f"{self.path} does not export {sym.name} declared in {HEADER_PATH}")
            cfunc.restype = ctypes.c_int
            cfunc.argtypes = [_CTYPES[a.dtype] for a in sym.args]
            self._kernels[sym.name] = Kernel(self, sym, cfunc)
        self.module.sfhip_last_error.restype = ctypes.c_char_p
        self.module.sfhip_set_stream.argtypes = [ctypes.c_void_p]
        self.module.sfhip_selftest_lane_shift.argtypes = [ctypes.c_void_p] * 3
        self._torch = torch
        logger.info(f"loaded {self.path} with {len(self.api)} kernels")
        if code is not None:
            self.check_code(code)
        if define_macros:
            self.apply_macros(dict(define_macros))

    def check_code(self, code):
        """The ``PUBLIC void`` kernels of `code` against the library's: names, argument types, names and order."""
        from .parse_api import parse_reference_api, signature_mismatch

        try:
            declared = parse_reference_api(code)
        except ValueError as exc:
            raise KernelError(f"cannot read the kernel declarations of the code handed to Library: {exc}")
        for name, sym in declared.items():
            if name not in self.api:
                raise KernelError(f"the code declares kernel {name}, which {self.path} does not provide")
            problem = signature_mismatch(sym, self.api[name])
            if problem:
                raise KernelError(f"the code handed to Library does not match {self.path}: {problem}")
        self.declared = sorted(declared)

    def apply_macros(self, macros):
        """PLM_THETA / MACH_CEILING of the srhd_2d kernels (srhd_2d.py:232-240); nothing else can change prebuilt code."""
        unknown = sorted(set(macros) - {"PLM_THETA", "MACH_CEILING"})
        if unknown:
            raise KernelError(f"libsfhip.so is prebuilt: the macros {unknown} cannot be applied to it")
        kernel = self._kernels.get("sfhip_srhd_2d_set_macros")
        if kernel is None:
            raise KernelError("this library has no srhd_2d kernels to apply PLM_THETA / MACH_CEILING to")
        # (what the macros default to in the reference's source, srhd_2d.c, when only one of them is given)
        status = kernel.cfunc(1, float(macros.get("PLM_THETA", 1.5)), float(macros.get("MACH_CEILING", 1e6)))
        if status != 0:
            raise KernelError(f"sfhip_srhd_2d_set_macros failed ({status}): {self.last_error()}")

    @property
    def xp(self):
        from .system import get_array_module

        return get_array_module("hip")

    def last_error(self):
        return (self.module.sfhip_last_error() or b"").decode()

    def reductions_workspace_len(self, ni, nj):
        """Doubles of scratch the sfhip_*_reductions entry points need (a size query: returns a value, not a status)."""
        return int(self.module.sfhip_reductions_workspace_len(int(ni), int(nj)))

    def bind_stream(self, args):
        """
        Point the library at torch's current stream on the device that holds the
        first array argument (what cupy's current-stream state does for the
        reference's gpu mode, library.py:118-135).
        """
        torch = self._torch
        for a in args:
            if isinstance(a, torch.Tensor):
                if not a.is_cuda:
                    raise KernelError("hip kernels need device arrays; got a CPU tensor")
                if torch.cuda.current_device() != a.device.index:
                    raise KernelError(
                        f"array lives on cuda:{a.device.index} but the current device is "
                        f"cuda:{torch.cuda.current_device()}; wrap the call in the patch's execution_context"
                    )
                stream = torch.cuda.current_stream(a.device)
                self.module.sfhip_set_stream(ctypes.c_void_p(stream.cuda_stream))
                return
        self.module.sfhip_set_stream(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def __getattr__(self, symbol):
        try:
            return self.__dict__["_kernels"][symbol]
        except KeyError:
            raise AttributeError(f"kernel library has no symbol {symbol}")


def to_ctypes(args, spec):
    """Coerce positional arguments to C values (reference: library.py:252-264)."""
    out = []
    for arg, (typename, _, _) in zip(args, spec):
        if typename == "int":
            out.append(int(arg))
        elif typename == "double":
            out.append(float(arg))
        else:
            out.append(None if arg is None else arg.data_ptr())
    return out


def type_error(sym, n, a, b):
    return TypeError(f"argument {n} to {sym} has type {type(a).__name__}, expected {b}")


def dtype_error(sym, n, a, b):
    return TypeError(f"argument {n} to {sym} has dtype {a.dtype}, expected {b}")


def arglen_error(sym, a, b):
    return TypeError(f"{sym} takes exactly {len(b)} arguments ({len(a)} given)")


def layout_error(sym, n):
    return TypeError(f"arg {n} to {sym} is not c-contiguous")


def validate_types(args, spec, symbol, lib):
    """Same checks and messages as the reference's validate_types (library.py:283-300)."""
    import numpy as np

    torch = lib._torch
    if len(args) != len(spec):
        raise arglen_error(symbol, args, spec)
    for n, (arg, (typename, argname, constraint)) in enumerate(zip(args, spec)):
        if typename == "int":
            if type(arg) not in (int, bool, np.int32, np.int64):
                raise type_error(symbol, n, arg, "int")
        elif typename == "double":
            if type(arg) not in (float, np.float64):
                raise type_error(symbol, n, arg, "float64")
        elif typename == "double*":
            if arg is None:
                continue
            if not isinstance(arg, torch.Tensor):
                raise type_error(symbol, n, arg, "ndarray")
            if arg.dtype != torch.float64:
                raise dtype_error(symbol, n, arg, "float64")
            if not arg.is_contiguous():
                raise layout_error(symbol, n)


def validate_constraints(args, spec, symbol):
    """Evaluate the header's ``// :: <constraint>`` comments (reference: library.py:303-315)."""
    scope = dict(zip([a[1] for a in spec], args))
    for arg, (_, name, constraint) in zip(args, spec):
        if constraint and arg is not None:
            c = constraint.replace("$", name)
            if not eval(c, None, scope):
                raise ValueError(f"argument constraint for {symbol} not satisfied: {c}")
