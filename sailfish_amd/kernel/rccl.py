"""
RCCL called directly (ctypes on librccl.so) for the two exchanges on the solver's critical path:
the halo slabs between neighbouring ranks -- grouped ``ncclSend`` / ``ncclRecv`` on the solver's own
communication stream -- and the MAX all-reduce of the device-side wavespeed slot.

Why not torch.distributed for these: its ``batch_isend_irecv`` of one exchange (2 sends + 2 receives) costs the
host about 0.15 ms (work objects, stream bookkeeping; tools/p2p_cost.py), twice per RK2 step, against stage
kernels of well under 0.1 ms on the slabs of a strong-scaling run.  The same four calls inside
``ncclGroupStart`` / ``ncclGroupEnd`` cost about 10 us.  Everything is stream-ordered: nothing here blocks the
host, and completion is an event on the stream.

The communicator is created collectively (``RcclComm.create``): rank 0 makes the unique id, it travels to the
other ranks through the existing torch.distributed group, every rank calls ``ncclCommInitRank``.  It is the
default transport of device runs whose torch.distributed backend is RCCL (``SFHIP_HALO_TRANSPORT=torch`` selects
torch.distributed's own calls instead); if any rank fails to join, every rank falls back to torch.distributed
together (``SolverBase._create_rccl``).  The same communicator is handed to libsfhip.so's step groups
(``sfhip_step_group_*``), which then issue these calls from C (``library_path`` names the file to bind).

The MAX all-reduce of the wavespeed slot reduces the doubles' BIT PATTERNS as ``ncclUint64``: for non-negative
doubles the patterns order like the values, and a NaN's pattern ranks above every finite value and infinity, so
a rank whose state has blown up gives every rank a NaN dt -- as the one-rank fold (``max_nan``, the integer
``atomicMax``) does.  ``ncclMax`` on ``ncclDouble`` is free to drop the NaN.
"""
import ctypes
import os

NCCL_UNIQUE_ID_BYTES = 128
NCCL_FLOAT64 = 8   # ncclDataType_t: ncclDouble
NCCL_UINT64 = 5    # ncclDataType_t: ncclUint64
NCCL_MAX = 2       # ncclRedOp_t


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


class RcclError(RuntimeError):
    pass


def _load():
    """librccl.so: the copy torch has already mapped (same SONAME), else the ROCm installation's."""
    candidates = []
    try:
        import torch

        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    except ImportError:
        pass
    candidates += ["librccl.so", "/opt/rocm/lib/librccl.so"]
    last = None
    for path in candidates:
        try:
            lib = ctypes.CDLL(path)
            break
        except OSError as exc:
            last = exc
    else:
        raise RcclError(f"librccl.so not found: {last}")
    vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), i, _UniqueId, i]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclGroupStart.argtypes = []
    lib.ncclGroupEnd.argtypes = []
    lib.ncclSend.argtypes = [vp, sz, i, i, vp, vp]
    lib.ncclRecv.argtypes = [vp, sz, i, i, vp, vp]
    lib.ncclAllReduce.argtypes = [vp, vp, sz, i, i, vp, vp]
    lib.ncclGetErrorString.argtypes = [i]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclCommCount.argtypes = [vp, ctypes.POINTER(i)]
    lib.ncclCommUserRank.argtypes = [vp, ctypes.POINTER(i)]
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd",
                 "ncclSend", "ncclRecv", "ncclAllReduce", "ncclCommCount", "ncclCommUserRank"):
        getattr(lib, name).restype = i
    lib.sfhip_path = path  # the file this handle came from: libsfhip.so binds the same one (sfhip_rccl_bind)
    return lib


class RcclComm:
    """One RCCL communicator over the ranks of a torch.distributed group, driven on caller-given streams."""

    def __init__(self, lib, comm, rank, size):
        self.lib, self.comm, self.rank, self.size = lib, comm, rank, size

    @classmethod
    def create(cls, dist=None, rank=0, size=1):
        """Collective over the ranks of `dist` (torch.distributed, initialised); a one-rank communicator needs none."""
        lib = _load()
        uid = _UniqueId()
        if rank == 0:
            cls._check(lib, lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        if size > 1:
            box = [ctypes.string_at(ctypes.addressof(uid), NCCL_UNIQUE_ID_BYTES) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctypes.memmove(ctypes.addressof(uid), box[0], NCCL_UNIQUE_ID_BYTES)
        comm = ctypes.c_void_p()
        cls._check(lib, lib.ncclCommInitRank(ctypes.byref(comm), size, uid, rank), "ncclCommInitRank")
        return cls(lib, comm, rank, size)

    @staticmethod
    def _check(lib, status, what):
        if status != 0:
            raise RcclError(f"{what} failed: {lib.ncclGetErrorString(status).decode()}")

    def exchange(self, ops, stream):
        """
        `ops`: [(kind, tensor, peer)], kind "send" | "recv", contiguous float64 device tensors; all of them go out
        as ONE group on `stream` (a raw hipStream_t), so sends and receives between any pair of ranks cannot
        deadlock whatever their order; between two ranks messages are matched in posting order.
        """
        lib = self.lib
        self._check(lib, lib.ncclGroupStart(), "ncclGroupStart")
        for kind, tensor, peer in ops:
            call = lib.ncclSend if kind == "send" else lib.ncclRecv
            self._check(lib, call(tensor.data_ptr(), tensor.numel(), NCCL_FLOAT64, peer, self.comm, stream),
                        "ncclSend" if kind == "send" else "ncclRecv")
        self._check(lib, lib.ncclGroupEnd(), "ncclGroupEnd")

    @property
    def library_path(self):
        return self.lib.sfhip_path

    def count(self):
        """Number of ranks as RCCL itself reports it (ncclCommCount) and this process's rank in it."""
        n, r = ctypes.c_int(-1), ctypes.c_int(-1)
        self._check(self.lib, self.lib.ncclCommCount(self.comm, ctypes.byref(n)), "ncclCommCount")
        self._check(self.lib, self.lib.ncclCommUserRank(self.comm, ctypes.byref(r)), "ncclCommUserRank")
        return n.value, r.value

    def max_scalar(self, tensor, stream):
        """In-place MAX all-reduce on `stream` of a float64 device tensor of NON-NEGATIVE values (or NaN), exact and
        NaN-keeping: reduced as the unsigned integers their bit patterns are (see the module docstring)."""
        ptr = tensor.data_ptr()
        self._check(self.lib, self.lib.ncclAllReduce(ptr, ptr, tensor.numel(), NCCL_UINT64, NCCL_MAX, self.comm, stream),
                    "ncclAllReduce")

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None
