"""
Machinery shared by the cbdiso_2d and cbdgam_2d solvers in ``hip`` mode.

It reproduces the observable behaviour of the reference's ``Patch`` / ``Solver``
pairs (``sailfish/solvers/cbdiso_2d.py:49-590``, ``cbdgam_2d.py:46-520``) --
``Solver.advance(dt)``, ``maximum_wavespeed()``, ``solution``, ``time``, the RK
schedules, the stage-time formula, slab patches from ``subdivide`` -- with a
device-resident, fused data flow:

reference, per RK2 step and patch          here
-----------------------------------------  ------------------------------------
primitive_to_conserved pass (48 B/zone)    u^n recomputed in-kernel from the
                                           retained p^n buffer (un_mode 2)
set_bc: ~12 strided slice copies           outflow edges are index clamps inside
                                           the kernel; only the 2 contiguous
                                           halo slabs between patches move
2 x advance_rk                             2 x fused stage kernel
wavespeed kernel + xp.max() (40 B/zone)    folded into the last stage's epilogue
3 state arrays                             2 (RK1/RK2) or 3 (RK3) state arrays

With ``fused=False`` the solver instead drives the reference-named entry points
in the reference's own order (p2c, set_bc slice copies, advance_rk with
conserved0, wavespeed array + max): the drop-in path, kept for parity tests and
for measuring what the fusions buy.

Multi-GPU: one process per GPU under ``torch.distributed`` (RCCL on device,
gloo in the CPU tests).  Global patch k of ``subdivide(ni, num_patches)`` lives
on rank ``k * world // num_patches``; halo slabs between ranks travel as
``isend``/``irecv`` pairs on a side stream while the rows that do not depend on
them are updated, and dt is a MAX all-reduce of one double.
"""
import contextlib
import os
from logging import getLogger

import numpy as np

from ..kernel.library import Library
from ..kernel.system import execution_context, get_array_module, num_devices, to_host
from ..mesh import PlanarCartesian2DMesh
from ..subdivide import concat_on_host, subdivide

logger = getLogger(__name__)

NG = 2  # guard zones per side

# quantity codes of the reductions kernels' spec array (csrc/sf_reduce.hpp, include/sfhip.h)
QUANTITY_CODES = {
    "mdot": 0, "fx": 1, "fy": 2, "torque": 10, "power": 11, "sigma_m1": 12, "eccentricity_vector": 13,
    "angular_momentum": 14, "mass": 15,
}
COMPLEX_QUANTITIES = ("sigma_m1", "eccentricity_vector")
RATE_QUANTITIES = ("mdot", "fx", "fy", "torque", "power")
DIAGNOSTICS_PER_LAUNCH = 8


# The fused wavespeed epilogue evaluates exactly (IEEE, bit-identical to the reference's wavespeed kernel) only
# the zones whose estimated speed reaches this fraction of the previous step's global maximum; a step whose new
# maximum falls below that bar is redone from an exact recomputation, so the choice affects speed only.
# Measured (MI355X, circumbinary disk from its analytic start): 0.5 -> 0.8 takes 1 % off a 4096^2 step and 4 % off
# a 512^2 step; at 0.9 steps already repeat during the initial transient (0.555 -> 0.606 ms per step at 4096^2).
WAVESPEED_FLOOR_FRACTION = float(os.environ.get("SFHIP_WS_FLOOR_FRACTION", "0.75"))


class HipBackend:
    """The product backend: libsfhip.so kernels on torch device arrays."""

    mode = "hip"

    def __init__(self, debug=False):
        self.lib = Library(mode="hip", debug=debug)
        self.xp = get_array_module("hip")

    def context(self, device_id):
        return execution_context("hip", device_id)

    def num_devices(self):
        return num_devices("hip")


_UNSET = object()


class Comm:
    """Thin view of torch.distributed: who am I, and the few collectives the solver needs."""

    def __init__(self, enabled):
        self.enabled = enabled
        self._host_group = _UNSET
        self._staged = None
        if enabled:
            import torch.distributed as dist

            self.dist = dist
            self.rank = dist.get_rank()
            self.size = dist.get_world_size()
        else:
            self.dist = None
            self.rank = 0
            self.size = 1

    @classmethod
    def auto(cls, distributed=None):
        if distributed is None:
            try:
                import torch.distributed as dist

                distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            except ImportError:
                distributed = False
        return cls(bool(distributed))

    @property
    def host_staged(self):
        """
        True when the default group is gloo.  Its point-to-point calls take a tensor's data pointer as host memory
        and know nothing of streams: a device array handed to them is read or written by the host, through the
        BAR, whenever the transport thread gets to it -- not ordered with the kernels that produce or consume it.
        Device data therefore crosses a gloo group through host buffers, with the stream ordering made explicit
        (max_scalar here, SolverBase.start_halo_exchange).  RCCL groups are stream-ordered by themselves.
        """
        if self._staged is None:
            self._staged = bool(self.enabled and self.dist.get_backend() == "gloo")
        return self._staged

    def max_scalar(self, tensor):
        """
        In-place MAX all-reduce of a one-element float64 tensor of NON-NEGATIVE values (wavespeeds), exact and
        NaN-keeping.  A backend's floating-point MAX is free to drop a NaN (fmax semantics, or whatever the argument
        order gives), and then a rank whose state has blown up would not stop the others: the reference, and the
        one-rank fold here (max_nan, the integer atomicMax), hand every caller a NaN dt.  So the BIT PATTERNS are
        reduced, as integers: for non-negative doubles they order like the values, and a NaN's pattern (sign bit
        cleared) ranks above every finite value and infinity.
        """
        if self.enabled:
            if self.host_staged and getattr(tensor, "is_cuda", False):
                host = tensor.cpu()  # waits for the current stream: the value is final
                self._max_of_bit_patterns(host)
                tensor.copy_(host)
            else:
                self._max_of_bit_patterns(tensor)
        return tensor

    def _max_of_bit_patterns(self, tensor):
        import torch

        tensor.abs_()  # a sign-bit operation: x86 makes NaNs with the sign bit set, which would rank lowest as int64
        self.dist.all_reduce(tensor.view(torch.int64), op=self.dist.ReduceOp.MAX)

    def host_group(self):
        """
        The process group host arrays travel on: the default group when it is gloo, else a gloo group beside
        the RCCL one (created on first use; every rank reaches that point together, inside a collective
        gather).  Checkpoint gathers and the diagnostics' per-patch sums are host data: moving them through
        RCCL would stage them in device memory next to the solver state.
        """
        if self._host_group is _UNSET:
            self._host_group = None if self.dist.get_backend() == "gloo" else self.dist.new_group(backend="gloo")
        return self._host_group

    def gather_arrays(self, local, root=0):
        """
        Rank `root` receives the list of every rank's float64 array (in rank order) and the other ranks, which
        only send theirs, get None: one copy of the mesh on one host, as the reference's concat_on_host makes
        (subdivide.py:63-122).  Arrays travel as raw doubles (no pickling); shapes may differ between ranks.
        """
        if not self.enabled:
            return [local]
        import torch

        dist, group = self.dist, self.host_group()
        local = np.ascontiguousarray(local, dtype=np.float64)
        meta = torch.zeros(8, dtype=torch.int64)
        meta[0] = local.ndim
        meta[1:1 + local.ndim] = torch.tensor(local.shape, dtype=torch.int64)
        metas = [torch.zeros(8, dtype=torch.int64) for _ in range(self.size)]
        dist.all_gather(metas, meta, group=group)
        if self.rank != root:
            dist.send(torch.from_numpy(local), dst=root, group=group)
            return None
        parts, requests = [], []
        for r in range(self.size):
            if r == root:
                parts.append(local)
                continue
            shape = tuple(int(v) for v in metas[r][1:1 + int(metas[r][0])])
            buf = torch.empty(shape, dtype=torch.float64)
            requests.append(dist.irecv(buf, src=r, group=group))
            parts.append(buf.numpy())
        for req in requests:
            req.wait()
        return parts

    def all_gather_arrays(self, local):
        """Every rank receives the list of every rank's array; for SMALL arrays of one shape (per-patch sums)."""
        if not self.enabled:
            return [local]
        import torch

        t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
        out = [torch.zeros_like(t) for _ in range(self.size)]
        self.dist.all_gather(out, t, group=self.host_group())
        return [o.numpy() for o in out]


class PatchBase:
    """
    One slab of the mesh with its device arrays.

    Mirrors the attributes of the reference's Patch (cbdiso_2d.py:55-94):
    ``primitive1`` is the current state, ``primitive2`` the other ping-pong
    buffer, ``time`` / ``time0`` the stage and step-start times, ``xl..yr`` the
    patch extents from ``mesh.vertex_coordinates``.
    """

    num_fields = None

    def __init__(self, time, primitive, mesh, index_range, physics, options, backend, context, fused,
                 first, last, rk_order):
        i0, i1 = index_range
        self.index_range = index_range
        self.mesh = mesh
        self.lib = backend.lib
        self.xp = xp = backend.xp
        self.execution_context = context
        self.time = self.time0 = time
        self.shape = (i1 - i0, mesh.shape[1])  # not including guard zones
        self.physics = physics
        self.options = options
        self.fused = fused
        self.first = first  # touches the lower domain edge in i
        self.last = last    # touches the upper domain edge in i
        self.xl, self.yl = mesh.vertex_coordinates(i0, 0)
        self.xr, self.yr = mesh.vertex_coordinates(i1, mesh.shape[1])
        self.wavespeed_valid = False
        with self.execution_context:
            if callable(primitive):
                # initial data made on the device: `primitive(array)` fills the interior of a zeroed array
                self.primitive1 = xp.zeros([i1 - i0 + 2 * NG, mesh.shape[1] + 2 * NG, self.num_fields])
                primitive(self.primitive1)
            else:
                self.primitive1 = xp.array(primitive)
            self.primitive2 = self.primitive1.clone()
            self.primitive3 = self.primitive1.clone() if (fused and rk_order == 3) else None
            self.wavespeed_max = xp.zeros(1)
            if not fused:
                self.conserved0 = xp.zeros(self.primitive1.shape)
                self.wavespeeds = xp.zeros(self.primitive1.shape[:2])
        self._coords = None
        self._reduce_workspace = None
        self.plan = None  # launch plan of the fused stage (device runs; see kernel/plan.py)

    @property
    def primitive(self):
        return self.primitive1

    @property
    def cell_center_coordinate_arrays(self):
        if self._coords is None:
            ni, nj = self.shape
            with self.execution_context:
                x0 = self.xl + 0.5 * self.mesh.dx
                x1 = self.xr - 0.5 * self.mesh.dx
                y0 = self.yl + 0.5 * self.mesh.dy
                y1 = self.yr - 0.5 * self.mesh.dy
                self._coords = (self.xp.linspace(x0, x1, ni)[:, None], self.xp.linspace(y0, y1, nj)[None, :])
        return self._coords

    def reduce_diagnostics(self, masses, spec, num_diagnostics, sums):
        """Launch the fused diagnostics reduction of this patch (`spec`, `sums` on its device)."""
        ni, nj = self.shape
        if self._reduce_workspace is None:
            n = self.lib.reductions_workspace_len(ni, nj)
            self._reduce_workspace = self.xp.zeros(n)
        x, y = self.cell_center_coordinate_arrays
        m1, m2 = masses
        self.lib_reductions[self.shape](
            *self._extent(), *m1.kernel_args(), *m2.kernel_args(), *self._reductions_args(),
            self.primitive1, x, y, num_diagnostics, spec, self._reduce_workspace.numel(), self._reduce_workspace,
            sums,
        )

    def _reductions_args(self):
        return ()

    def stage_time(self, rk_param, dt):
        """Time after a stage; the reference's expression, literally (cbdiso_2d.py:239)."""
        return self.time0 * rk_param + (self.time + dt) * (1.0 - rk_param)

    # --- reference-style (unfused) operations -------------------------------
    def new_iteration(self):
        self.time0 = self.time
        if not self.fused:
            self.recompute_conserved()

    # --- fused stage -----------------------------------------------------------
    def stage_buffers(self, stage, nstages):
        """
        (read, u^n source, write) for stage `stage` of `nstages`.

        After stage 0 the buffers are swapped, so from then on ``primitive1`` is
        the latest stage result and ``primitive2`` holds p^n.  p^n stays intact
        until the last stage, whose output overwrites it in place: each zone's
        u^n is read by the very lane that then writes that zone and by nothing
        else.  Only RK3's middle stage needs a third buffer.
        """
        if stage == 0:
            return self.primitive1, None, self.primitive2
        if nstages == 3 and stage == 1:
            return self.primitive1, self.primitive2, self.primitive3
        return self.primitive1, self.primitive2, self.primitive2

    def finish_stage(self, rk_param, dt, wrote=None):
        self.time = self.stage_time(rk_param, dt)
        if wrote is None or wrote is self.primitive2:
            self.primitive1, self.primitive2 = self.primitive2, self.primitive1
        else:
            self.primitive1, self.primitive3 = self.primitive3, self.primitive1


class SolverBase:
    """Common part of the two Solver classes; see the module docstring."""

    num_cons = None
    patch_class = None
    default_rk_order = 2
    _staging = None   # pinned host buffers of the exchange over a gloo group (_start_host_staged_exchange)
    wavespeed_redos = 0   # steps whose stage 0 was repeated because the filtered wavespeed fell below its floor
    _rccl = None      # direct RCCL transport (kernel/rccl.py): solvers set up by _setup_patches
    _self_exchange = False
    _group = None

    def _setup_patches(self, setup, mesh, time, solution, num_patches, mode, backend, fused, distributed,
                       initial_condition):
        if type(mesh) is not PlanarCartesian2DMesh:
            raise ValueError("solver only supports 2D cartesian mesh")
        if setup.boundary_condition != "outflow":
            raise ValueError("solver only supports outflow boundary condition")
        if backend is None:
            if mode != "hip":
                raise ValueError(
                    f"sailfish_amd solvers implement execution mode 'hip' only (got {mode!r}); "
                    "use the reference for cpu/omp"
                )
            backend = HipBackend()
        self.backend = backend
        self.mode = backend.mode
        self.xp = backend.xp
        self.mesh = mesh
        self.setup = setup
        self.num_guard = NG
        self.fused = fused
        self.comm = Comm.auto(distributed)
        ni, nj = mesh.shape
        nq = self.num_cons

        world = self.comm.size
        if num_patches is None or num_patches < world:
            num_patches = world
        if num_patches % world:
            raise ValueError(f"num_patches ({num_patches}) must be a multiple of the number of ranks ({world})")
        per_rank = num_patches // world
        ranges = subdivide(ni, num_patches)
        self.num_patches = num_patches
        self.patch_ranges = ranges
        self.local_ids = list(range(self.comm.rank * per_rank, (self.comm.rank + 1) * per_rank))
        self.owner = lambda k: k // per_rank  # (one rank: 0, which is also what a self-exchange sends to)

        logger.info(f"initiate with time={time:0.4f}")
        logger.info(f"subdivide grid over {num_patches} patches on {world} rank(s)")
        logger.info(f"mesh is {mesh}")

        ndev = backend.num_devices()
        if self.comm.enabled and backend.mode == "hip":
            base_device = int(os.environ.get("LOCAL_RANK", self.comm.rank % max(ndev, 1)))
        else:
            base_device = None

        self.patches = []
        rk_order = getattr(self._options, "rk_order", self.default_rk_order)
        # initial data are made on the device where the setup can make them there bit for bit (SFHIP_HOST_IC=1:
        # always on the host, as the reference does)
        device_ic = (solution is None and backend.mode == "hip" and getattr(setup, "has_device_primitive", False)
                     and hasattr(backend.lib, "sfhip_circumbinary_disk_primitive")
                     and not os.environ.get("SFHIP_HOST_IC"))
        for n, k in enumerate(self.local_ids):
            a, b = ranges[k]
            if device_ic:
                def prim(out, a=a, b=b):
                    setup.device_primitive(backend.lib, time, mesh, a, b, out)
            else:
                prim = np.zeros([b - a + 2 * NG, nj + 2 * NG, nq])
                if solution is None:
                    prim[NG:-NG, NG:-NG] = initial_condition(setup, mesh, time, a, b)
                else:
                    prim[NG:-NG, NG:-NG] = solution[a:b]
            device = base_device if base_device is not None else (n % max(ndev, 1))
            patch = self.patch_class(
                time, prim, mesh, (a, b), self._physics, self._options, backend,
                backend.context(device), fused, first=(k == 0), last=(k == num_patches - 1),
                rk_order=rk_order, **self._patch_kwargs(),
            )
            self.patches.append(patch)
        self._comm_stream = None
        self._rccl = None
        # Halo slabs and the wavespeed all-reduce between ranks: "rccl" = RCCL called directly on the solver's
        # streams (kernel/rccl.py; from C inside a step group), the default where the torch.distributed backend is
        # RCCL; "torch" = torch.distributed's own calls (always over a gloo group).  If any rank cannot join the
        # direct communicator, all ranks fall back to "torch" together (_create_rccl).
        # SFHIP_SELF_EXCHANGE=1 (tests; no torch.distributed): the halos between the patches of ONE process travel
        # through a one-rank RCCL communicator, sends and receives to the own rank -- all of the multi-rank data
        # path that a one-GPU box can run.
        self._self_exchange = bool(os.environ.get("SFHIP_SELF_EXCHANGE")) and not self.comm.enabled and backend.mode == "hip"
        default_transport = "rccl" if (self.comm.enabled and backend.mode == "hip"
                                       and self.comm.dist.get_backend() == "nccl") else "torch"
        self.halo_transport = os.environ.get("SFHIP_HALO_TRANSPORT", default_transport)
        if self.comm.enabled and backend.mode == "hip" and self.halo_transport == "rccl":
            self._rccl = self._create_rccl()
            if self._rccl is None:
                self.halo_transport = "torch"
        elif self._self_exchange:
            from ..kernel.rccl import RcclComm

            with self.patches[0].execution_context:
                self._rccl = RcclComm.create(None, 0, 1)
            self.halo_transport = "rccl"
        if backend.mode == "hip" and fused and not os.environ.get("SFHIP_NO_PLANS"):
            from ..kernel.plan import StagePlan

            for p in self.patches:
                with p.execution_context:
                    p.plan = StagePlan(p.lib, p, type(self).__module__.rsplit(".", 1)[-1])
        # A whole RK2 step from one C call where nothing needs Python between the stages: point masses that are
        # the setup's own Kepler binary, every patch on one device, and either no other ranks or the direct RCCL
        # transport.  One patch without neighbours: sfhip_plan_step_rk2; otherwise a step group
        # (sfhip_step_group_rk2: slab copies / RCCL exchange, interior and boundary launches, all-reduce).
        self._c_step = False
        self._group = None
        fn = getattr(self._physics, "point_mass_function", None)
        kepler = (hasattr(setup, "kepler_binary") and getattr(fn, "__self__", None) is setup
                  and getattr(fn, "__func__", None) is getattr(type(setup), "point_masses", None))
        plans = bool(self.patches) and all(p.plan is not None for p in self.patches)
        if plans and kepler and not os.environ.get("SFHIP_NO_C_STEP"):
            if len(self.patches) == 1 and not self.comm.enabled:
                self.patches[0].plan.set_kepler_binary(*setup.kepler_binary())
                self._c_step = True
            elif (len({str(p.primitive1.device) for p in self.patches}) == 1
                  and (self._rccl is not None or not self.comm.enabled)
                  and all(p.shape[0] >= 2 * NG for p in self.patches)):
                try:
                    self._group = self._create_step_group(setup)
                except Exception as exc:
                    # Not fatal: the Python stages issue the same exchanges in the same order over the same
                    # communicator (a rank stepping from C and a rank stepping from Python stay in step), only slower.
                    logger.warning(f"no C step group ({exc!r}); the stages are driven from Python")
                    self._group = None
        self._mass_cache = (None, None)
        self._wavespeed_floor = 0.0
        self._dt_dev = None
        self._slots = None        # two device doubles: the reduced wavespeed of the current / next state (step())
        self._slot_views = None
        self._slot_cur = 0
        self._slot_valid = False  # slot[_slot_cur] holds the global maximum wavespeed of the current state
        self._dt_dx = self._dt_cfl = 0.0
        self._host_seq = 0.0
        self._fetch_stream = None
        self._remote_neighbors = None
        self.event_log = None  # bench.py sets this to a list to time every stage launch with HIP events

    def _patch_kwargs(self):
        return {}

    def _create_step_group(self, setup):
        """The C-side step group of this process's patches (kernel/plan.py: StepGroup)."""
        from ..kernel.plan import PEER_EDGE, PEER_LOCAL, StepGroup

        local = set(self.local_ids)
        lower, upper = [], []
        for k in self.local_ids:
            codes = []
            for n in self._neighbors(k):
                if n is None:
                    codes.append(PEER_EDGE)
                elif n in local and not self._self_exchange:
                    codes.append(PEER_LOCAL)
                else:
                    codes.append(self.owner(n))
            lower.append(codes[0])
            upper.append(codes[1])
        remote = any(c >= 0 for c in lower + upper)
        torch = self._torch()
        with self.patches[0].execution_context:
            if remote and self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
                for p in self.patches:
                    p.plan.set_side_stream(self._comm_stream)
            return StepGroup(self.patches[0].lib, self.patches, self.local_ids, lower, upper,
                             self._rccl if (remote or self.comm.enabled) else None, self.comm.enabled,
                             self._comm_stream if remote else None, *setup.kepler_binary())

    def _create_rccl(self):
        """Collective: every rank tries to join a direct RCCL communicator; all fall back together if any failed."""
        from ..kernel.rccl import RcclComm

        comm, error = None, None
        try:
            comm = RcclComm.create(self.comm.dist, self.comm.rank, self.comm.size)
        except Exception as exc:  # a missing library, a failed bootstrap: not fatal, the default transport remains
            error = exc
        torch = self._torch()
        # (a tensor where the process group moves data: device memory on RCCL, host memory on gloo)
        where = "cuda" if self.comm.dist.get_backend() == "nccl" else "cpu"
        ok = torch.tensor([0.0 if comm is None else 1.0], dtype=torch.float64, device=where)
        self.comm.dist.all_reduce(ok, op=self.comm.dist.ReduceOp.MIN)
        if float(ok.item()) != 1.0:
            if comm is not None:
                comm.close()
            logger.warning(f"direct RCCL transport unavailable ({error!r}); using torch.distributed")
            return None
        return comm

    # ------------------------------------------------------------------ state
    @property
    def solution(self):
        """
        (ni, nj, NCONS) float64 host array without guards (reference: cbdiso_2d.py:357-361).  In a multi-rank
        run this is a collective call: rank 0 returns the array, the other ranks return None.
        """
        local = concat_on_host([p.primitive for p in self.patches], (NG, NG))
        parts = self.comm.gather_arrays(local)
        return None if parts is None else np.concatenate(parts, axis=0)

    @property
    def primitive(self):
        return None  # as the reference: the solution array is primitive already

    @property
    def time(self):
        return self.patches[0].time

    @property
    def options(self):
        return self._options._asdict()

    @property
    def physics(self):
        return self._physics._asdict()

    @property
    def maximum_cfl(self):
        return 0.4

    def point_masses(self, time):
        """Kepler solve once per distinct time, shared by all local patches."""
        if self._mass_cache[0] != time:
            self._mass_cache = (time, self._physics.point_masses(time))
        return self._mass_cache[1]

    # ----------------------------------------------------------- diagnostics
    def reductions(self):
        """
        Time-series reductions of the solution (reference: cbdiso_2d.py:370-503,
        cbdgam_2d.py:341-443): one entry per ``physics.diagnostics`` item, a float
        (``time`` in units of the setup's reference time scale, or a real sum times
        dx*dy) or a complex number (``sigma_m1``, ``eccentricity_vector``).

        The reference builds four full-size source-term arrays per patch and one
        temporary per diagnostic; here each patch makes one fused pass per 8
        diagnostics on the device (``sfhip_cbd*_2d_reductions``) and returns 16
        doubles.  Patches are added in global patch order, then scaled by dx*dy,
        as the reference does (:494-499).
        """
        diagnostics = self._physics.diagnostics
        work = [(k, d) for k, d in enumerate(diagnostics) if d.quantity != "time"]
        rows = [self._diagnostic_spec(d) for _, d in work]
        per_patch = np.zeros([len(self.patches), len(work), 2])
        if work:
            torch = self._torch()
            pending = []
            for p in self.patches:
                masses = self.point_masses(p.time)
                with p.execution_context:
                    spec = p.xp.array(np.array(rows, dtype=float))
                    sums = p.xp.zeros(2 * len(work))
                    for lo in range(0, len(work), DIAGNOSTICS_PER_LAUNCH):
                        n = min(DIAGNOSTICS_PER_LAUNCH, len(work) - lo)
                        p.reduce_diagnostics(masses, spec[lo:], n, sums[2 * lo:])
                    pending.append(sums)
            for n, sums in enumerate(pending):  # all patches launched before the first blocking read
                per_patch[n] = to_host(sums).reshape(len(work), 2)
            del torch
        gathered = np.concatenate(self.comm.all_gather_arrays(per_patch), axis=0)  # global patch order
        da = self.mesh.dx * self.mesh.dy
        result = [None] * len(diagnostics)
        for k, d in enumerate(diagnostics):
            if d.quantity == "time":
                result[k] = self.time / self.setup.reference_time_scale
        for n, (k, d) in enumerate(work):
            re = sum(float(v) for v in gathered[:, n, 0]) * da
            if d.quantity in COMPLEX_QUANTITIES:
                result[k] = complex(re, sum(float(v) for v in gathered[:, n, 1]) * da)
            else:
                result[k] = re
        return result

    def _diagnostic_spec(self, d):
        """One row of the kernels' spec array for Diagnostic `d`; rejects what the reference cannot evaluate."""
        q = d.quantity
        if isinstance(q, str):
            if q not in QUANTITY_CODES:
                raise ValueError(f"unknown diagnostic quantity {q!r}")
            code = QUANTITY_CODES[q]
        else:
            code = int(q)
            if not 0 <= code < self.num_cons:
                raise ValueError(f"conserved field index {q} out of range for {self.num_cons} fields")
        is_rate = not isinstance(q, str) or q in RATE_QUANTITIES
        which = 0
        if is_rate:
            if not (d.accretion or d.gravity):
                raise ValueError(f"diagnostic {q!r} needs gravity=True or accretion=True")
            if d.which_mass not in (1, 2, "both") or (q == "power" and d.which_mass == "both"):
                raise ValueError(f"bad which_mass {d.which_mass!r} for diagnostic {q!r}")
            which = 3 if d.which_mass == "both" else int(d.which_mass)
        cut = d.radial_cut
        return [code, which, float(bool(d.gravity)), float(bool(d.accretion)), 0.0 if cut is None else 1.0,
                0.0 if cut is None else float(cut[0]), 0.0 if cut is None else float(cut[1]), 0.0]

    # -------------------------------------------------------------------- dt
    def maximum_wavespeed(self):
        """
        Global max wavespeed (reference: cbdiso_2d.py:525-534 via lazy_reduce).
        Launch on every patch first, then one device->host read, then the
        cross-rank MAX.  After a fused step the per-patch maxima are already on
        the device (epilogue of the last stage) and no kernel runs at all.

        The fused epilogue only folds zones at least WAVESPEED_FLOOR_FRACTION as fast as the
        previous maximum (``wavespeed_floor``); its result is exact whenever it
        is not below that floor, which is checked here on the global value.
        """
        floor = self._wavespeed_floor
        if self._slot_valid:  # left by step(): already reduced over patches and ranks
            value = float(self._slot_views[self._slot_cur].item())
            if not value < floor:
                self._wavespeed_floor = WAVESPEED_FLOOR_FRACTION * value
                return value
        have_fused = self.fused and all(p.wavespeed_valid for p in self.patches)
        value = self._reduce_wavespeed(recompute=not have_fused)
        if have_fused and value < floor:
            value = self._reduce_wavespeed(recompute=True)
        self._wavespeed_floor = WAVESPEED_FLOOR_FRACTION * value
        return value

    def _reduce_wavespeed(self, recompute):
        same_device = len({str(p.wavespeed_max.device) for p in self.patches}) == 1
        if same_device:
            return float(self._reduce_wavespeed_device(recompute).item())
        # patches spread over several devices of one process: reduce on the host, as the reference does
        if recompute:
            for p in self.patches:
                with p.execution_context:
                    p.launch_maximum_wavespeed(self.point_masses(p.time))
                p.wavespeed_valid = True
        local = self._torch().tensor([max(float(p.wavespeed_max.item()) for p in self.patches)],
                                     dtype=self._torch().float64, device=self.patches[0].wavespeed_max.device)
        if self.comm.enabled:
            self.comm.max_scalar(local)
        return float(local.item())

    def _torch(self):
        import torch

        return torch

    # --------------------------------------------------------------- advance
    def rk_parameters(self):
        order = getattr(self._options, "rk_order", self.default_rk_order)
        return {1: [0.0], 2: [0.0, 0.5], 3: [0.0, 0.75, 1.0 / 3.0]}[order]

    def advance(self, dt):
        """One full Runge-Kutta step (reference: cbdiso_2d.py:536-546, cbdgam_2d.py:473-476)."""
        self.new_iteration()
        params = self.rk_parameters()
        for stage, a in enumerate(params):
            self.advance_rk(a, dt, stage, len(params))

    def new_iteration(self):
        for p in self.patches:
            with p.execution_context:
                p.new_iteration()

    def advance_rk(self, rk_param, dt, stage=0, nstages=1, slots=None, host_seq=0.0):
        """
        One RK stage on every local patch.  `slots` (Solver.step only) = (current, next) one-element views
        of the device-side wavespeed slots: stage 0 then forms dt in the kernel from `current` and zeroes
        `next`, and the last stage folds the new maximum into `next` (see step()).
        """
        self._slot_valid = False
        if not self.fused:
            self.set_bc("primitive1")
            for p in self.patches:
                with p.execution_context:
                    p.advance_rk(rk_param, dt, self.point_masses(p.time))
            return

        last_stage = stage == nstages - 1
        # With halos from other ranks (device runs): the rows that need them are updated on the comm
        # stream, right behind the receives, concurrently with the interior launch on the compute
        # stream -- neither the halo latency nor the thin boundary launches sit on the critical path.
        side = self.backend.mode == "hip" and self._has_remote_neighbors()
        if side and last_stage and slots is None:
            # the wavespeed fold is then shared by launches on two streams: reset it before either starts
            for p in self.patches:
                with p.execution_context:
                    p.wavespeed_max.zero_()
        plans = []
        # what the launches of this stage get beyond the state: where dt comes from, what to zero, where to fold
        extra = {}
        if slots is not None:
            if stage == 0:
                extra = dict(amax_device=slots[0], dt_dx=self._dt_dx, dt_cfl=self._dt_cfl, zero_slot=slots[1])
            if last_stage:
                extra["fold_into"] = slots[1]
        accumulate = side or slots is not None  # the fold target has been zeroed already: never memset it

        def launch_interiors():
            for p in self.patches:
                ni = p.shape[0]
                lo = 0 if p.first else min(NG, ni)       # rows [0, lo) need the lower halo
                hi = ni if p.last else max(ni - NG, lo)  # rows [hi, ni) need the upper halo
                rd, un, wr = p.stage_buffers(stage, nstages)
                masses = self.point_masses(p.time)
                next_masses = self.point_masses(p.stage_time(rk_param, dt)) if last_stage else masses
                plans.append((p, lo, hi, rd, un, wr, masses, next_masses))
                ex = extra
                if host_seq and p is self.patches[0]:
                    ex = dict(extra, host_seq=host_seq)  # this launch hands the reduced wavespeed to the host
                with p.execution_context:
                    self._launch(p, rk_param, dt, rd, un, wr, masses, next_masses, lo, hi,
                                 (1 if accumulate else 2) if last_stage else 0, ex)

        if side:
            # The transport calls cost the host ~0.15 ms per exchange on RCCL (tools/p2p_cost.py), half a stage
            # kernel: the interior launch goes to the device FIRST, and the comm stream is ordered behind an
            # event recorded before it, so that the exchange still depends on the previous stage only.
            pending = self.start_halo_exchange(between=launch_interiors)
        else:
            pending = self.start_halo_exchange()
            launch_interiors()
        self.finish_halo_exchange(pending, join=not side)
        torch = self._torch() if side else None
        for p, lo, hi, rd, un, wr, masses, next_masses in plans:
            ni = p.shape[0]
            with p.execution_context:
                stream = torch.cuda.stream(self._comm_stream) if side else contextlib.nullcontext()
                with stream:
                    if lo > 0:
                        self._launch(p, rk_param, dt, rd, un, wr, masses, next_masses, 0, lo,
                                     1 if last_stage else 0, extra, side)
                    if hi < ni:
                        self._launch(p, rk_param, dt, rd, un, wr, masses, next_masses, hi, ni,
                                     1 if last_stage else 0, extra, side)
                if side:
                    torch.cuda.current_stream().wait_stream(self._comm_stream)  # the stage is complete on both
            if "amax_device" not in extra:
                p.finish_stage(rk_param, dt, wr)
            p.wavespeed_valid = last_stage and slots is None
        return [plan[5] for plan in plans]

    def _has_remote_neighbors(self):
        if self._remote_neighbors is None:
            local = set() if self._self_exchange else set(self.local_ids)
            self._remote_neighbors = any(
                n is not None and n not in local for k in self.local_ids for n in self._neighbors(k))
        return self._remote_neighbors

    def step(self, cfl_number, dx=None):
        """
        One driver iteration -- ``dt = dx / maximum_wavespeed() * cfl; advance(dt)`` (driver.py:508-511)
        -- with nothing but the stage kernels on the device and no host round trip on the critical path.

        The reduced wavespeed lives in one of two device slots.  Stage 0 forms dt in the kernel from the
        current slot (``dx / amax * cfl``, the driver's IEEE operations: no timestep launch) and zeroes the
        other slot, into which the last stage folds the wavespeed of the state it produces (no memset, no
        wavespeed pass); the slots then swap.  Meanwhile the host reads the current slot back on a side
        stream -- it has been final since the previous step ended, so the read does not wait for stage 0 --
        and forms the identical dt for its own bookkeeping (time, point-mass positions of the later stages).
        Same dt, same state, bit for bit, as the two blocking calls.  Returns dt.
        """
        dx = self.mesh.min_spacing() if dx is None else dx
        params = self.rk_parameters()
        same_device = len({str(p.wavespeed_max.device) for p in self.patches}) == 1
        if not (self.fused and len(params) > 1 and same_device and hasattr(self.patches[0].lib, "sfhip_timestep")):
            dt = dx / self.maximum_wavespeed() * cfl_number
            self.advance(dt)
            return dt

        p0 = self.patches[0]
        if self._slots is None:
            with p0.execution_context:
                self._slots = p0.xp.zeros(2)
                self._slot_views = (self._slots[0:1], self._slots[1:2])
        floor = self._wavespeed_floor
        filtered = self._slot_valid or all(p.wavespeed_valid for p in self.patches)
        self._dt_dx, self._dt_cfl = float(dx), float(cfl_number)
        while True:
            cur, nxt = self._slot_views[self._slot_cur], self._slot_views[1 - self._slot_cur]
            if not self._slot_valid:
                amax_dev = self._reduce_wavespeed_device(recompute=not filtered)
                with p0.execution_context:
                    cur.copy_(amax_dev)
            plan = getattr(p0, "plan", None)
            if (self._c_step or self._group is not None) and self.event_log is None and params == [0.0, 0.5]:
                # everything between here and the end of the step in one C call (include/sfhip.h): one patch
                # without neighbours through its plan, several patches and / or other ranks through the step group
                self._host_seq += 1.0
                if self._host_seq % 256.0 == 1.0:
                    plan.check_stream()  # the plans launch on the stream they were made on
                if self._c_step:
                    redo, amax, dt, t_after = plan.step_rk2(
                        p0.primitive1, p0.primitive2, cur, nxt, float(dx), float(cfl_number), p0.time,
                        floor if filtered else 0.0, WAVESPEED_FLOOR_FRACTION, self._host_seq)
                else:
                    redo, amax, dt, t_after = self._group.step_rk2(
                        [p.primitive1 for p in self.patches], [p.primitive2 for p in self.patches], cur, nxt,
                        float(dx), float(cfl_number), p0.time, floor if filtered else 0.0, WAVESPEED_FLOOR_FRACTION,
                        self._host_seq)
                if redo:
                    filtered = False
                    self._slot_valid = False
                    self.wavespeed_redos += 1
                    continue
                if dt != dt:
                    # the run has blown up (NaN wavespeed): fail where and how the Python stages and the reference
                    # do, in the Kepler solve of the next stage time (kepler.py:356-366: floor of a NaN)
                    self.point_masses(p0.stage_time(0.0, dt))
                self._wavespeed_floor = WAVESPEED_FLOOR_FRACTION * amax
                for p in self.patches:
                    p.time0 = p.time
                    p.time = p.stage_time(0.0, dt)
                    p.time = p.stage_time(0.5, dt)  # (the buffers swap twice: the state is where it was)
                    p.wavespeed_valid = False
                if p0.time != t_after and not (p0.time != p0.time and t_after != t_after):  # (a NaN dt gives NaN twice)
                    raise RuntimeError(f"the C step and Patch.stage_time disagree: {t_after!r} != {p0.time!r}")
                self._slot_cur = 1 - self._slot_cur
                self._slot_valid = True
                return dt
            if plan is not None:
                # stage 0 itself hands the value over through pinned memory as soon as it starts (no copy, no event)
                self._host_seq += 1.0
                seq = self._host_seq
                fetch = lambda: plan.host_wait(seq)  # noqa: E731
            else:
                seq = 0.0
                with p0.execution_context:
                    fetch = self._start_scalar_fetch(cur)  # read-back on a side stream, not behind stage 0
            self.new_iteration()
            # dt is not known on the host yet: nothing below may depend on it (stage 0 never does)
            wrote = self.advance_rk(params[0], float("nan"), 0, len(params), slots=(cur, nxt), host_seq=seq)
            amax = fetch()  # the one device->host sync of the step; stage 0 is still running on the device
            if filtered and amax < floor:
                # the filtered maximum is only a lower bound: recompute exactly and redo stage 0
                filtered = False
                self._slot_valid = False
                self.wavespeed_redos += 1
                continue
            break
        self._wavespeed_floor = WAVESPEED_FLOOR_FRACTION * amax
        dt = dx / amax * cfl_number
        for p, wr in zip(self.patches, wrote):
            p.finish_stage(params[0], dt, wr)
        for stage in range(1, len(params)):
            self.advance_rk(params[stage], dt, stage, len(params), slots=(cur, nxt))
        if self.comm.enabled:
            with p0.execution_context:
                # on the compute stream, behind the last stage; the result stays on the device
                if self._rccl is not None:
                    self._rccl.max_scalar(nxt, self._torch().cuda.current_stream().cuda_stream)
                else:
                    self.comm.max_scalar(nxt)
        self._slot_cur = 1 - self._slot_cur
        self._slot_valid = True
        return dt

    def _start_scalar_fetch(self, scalar):
        """
        Begin copying a one-element device array to the host on a side stream, ordered after the
        work already queued on the current stream but NOT after what is queued later; returns a
        function that blocks until the value has arrived and returns it as a float.
        """
        if self.backend.mode != "hip":
            return lambda: float(scalar.item())
        torch = self._torch()
        if self._fetch_stream is None:
            self._fetch_stream = torch.cuda.Stream()
            self._fetch_host = torch.empty(1, dtype=torch.float64).pin_memory()
            self._fetch_done = torch.cuda.Event()
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self._fetch_stream):
            self._fetch_stream.wait_event(ready)
            self._fetch_host.copy_(scalar, non_blocking=True)
            self._fetch_done.record()
        scalar.record_stream(self._fetch_stream)

        def wait():
            self._fetch_done.synchronize()
            return float(self._fetch_host[0])

        return wait

    def _reduce_wavespeed_device(self, recompute):
        """Max wavespeed over local patches and ranks as a device scalar; nothing here blocks the host."""
        if recompute:
            for p in self.patches:
                with p.execution_context:
                    p.launch_maximum_wavespeed(self.point_masses(p.time))
                p.wavespeed_valid = True
        local = self.patches[0].wavespeed_max
        if len(self.patches) > 1 or self.comm.enabled:
            local = local.clone()
            for p in self.patches[1:]:
                local = self._torch().maximum(local, p.wavespeed_max)
        if self.comm.enabled:
            self.comm.max_scalar(local)
        return local

    def _launch(self, p, rk_param, dt, rd, un, wr, masses, next_masses, i_begin, i_end, want_wavespeed,
                extra=None, on_side_stream=False):
        """One stage-kernel launch on patch `p`, optionally bracketed by HIP events on its stream."""
        floor = self._wavespeed_floor
        extra = extra or {}
        if self.event_log is None:
            p.launch_stage(rk_param, dt, rd, un, wr, masses, next_masses, i_begin, i_end, want_wavespeed, floor,
                           on_side_stream=on_side_stream, **extra)
            return
        torch = self._torch()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        fire = p.launch_stage(rk_param, dt, rd, un, wr, masses, next_masses, i_begin, i_end, want_wavespeed, floor,
                              defer=True, on_side_stream=on_side_stream, **extra)
        e0.record()
        fire()
        e1.record()
        self.event_log.append(dict(zones=(i_end - i_begin) * p.shape[1], reads_un=un is not None, start=e0, stop=e1))

    # ----------------------------------------------------------- halo exchange
    def _neighbors(self, k):
        return (k - 1 if k > 0 else None), (k + 1 if k < self.num_patches - 1 else None)

    def start_halo_exchange(self, array="primitive1", between=None):
        """
        Fill the guard rows between patches (reference: set_bc step 1,
        cbdiso_2d.py:569-571): each patch's first/last two interior rows go to
        its lower/upper neighbour's guard rows.  Rows are contiguous, so a halo
        is a single slab of 2*(nj+4)*NCONS doubles.  Same-process neighbours are
        a device copy; other ranks get isend/irecv on a side stream so the
        caller can update halo-independent rows meanwhile.

        `between` (device runs): called after the same-process copies have been queued and an event has
        been recorded on the compute stream, before the transport calls; the comm stream then waits for
        that event only, so what `between` queues on the compute stream (the interior launch) does not
        hold the exchange back and is not held back by the host time of the transport calls.
        """
        mine = {k: p for k, p in zip(self.local_ids, self.patches)}
        local = {} if self._self_exchange else mine  # (a self-exchange sends every slab through RCCL, to the own rank)
        ng = self.num_guard
        ops = []
        for k, p in mine.items():
            lower, upper = self._neighbors(k)
            pc = getattr(p, array)
            if lower is not None and lower in local:
                with p.execution_context:
                    pc[:ng].copy_(getattr(local[lower], array)[-2 * ng:-ng], non_blocking=True)
            if upper is not None and upper in local:
                with p.execution_context:
                    pc[-ng:].copy_(getattr(local[upper], array)[ng:2 * ng], non_blocking=True)
            # A message is named by the guard slab it fills: (destination patch, 0 lower | 1 upper).
            if upper is not None and upper not in local:
                ops.append(((upper, 0), "send", pc[-2 * ng:-ng], self.owner(upper)))
                ops.append(((k, 1), "recv", pc[-ng:], self.owner(upper)))
            if lower is not None and lower not in local:
                ops.append(((lower, 1), "send", pc[ng:2 * ng], self.owner(lower)))
                ops.append(((k, 0), "recv", pc[:ng], self.owner(lower)))
        # Between two ranks, messages are matched in posting order (RCCL has no tags).  Posting in
        # the order of the message names on both sides keeps the pairs aligned also when a rank
        # exchanges several slabs with the same peer (two ranks, several patches, wrap-around halos).
        ops.sort(key=lambda op: op[0])
        ops = [op[1:] for op in ops]
        if not ops:
            if between is not None:
                between()
            return None
        dist = self.comm.dist
        torch = self._torch()
        if self._rccl is not None:
            # one ncclGroup of sends and receives on the comm stream: stream-ordered, nothing to wait for on the host
            with self.patches[0].execution_context:
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream()
                    for p in self.patches:
                        if getattr(p, "plan", None) is not None:
                            p.plan.set_side_stream(self._comm_stream)
                ready = torch.cuda.Event()
                ready.record()
                if between is not None:
                    between()
                self._comm_stream.wait_event(ready)
                self._rccl.exchange(ops, self._comm_stream.cuda_stream)
            return "stream-ordered"
        if self.backend.mode == "hip" and self.comm.host_staged:
            return self._start_host_staged_exchange(ops, between)
        p2p = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer) for kind, t, peer in ops]
        if self.backend.mode == "hip":
            with self.patches[0].execution_context:
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream()
                    for p in self.patches:
                        if getattr(p, "plan", None) is not None:
                            p.plan.set_side_stream(self._comm_stream)
                if between is not None:
                    ready = torch.cuda.Event()
                    ready.record()
                    between()
                    self._comm_stream.wait_event(ready)
                else:
                    self._comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._comm_stream):
                    reqs = dist.batch_isend_irecv(p2p)
            return reqs
        if between is not None:
            between()
        return dist.batch_isend_irecv(p2p)

    def _start_host_staged_exchange(self, ops, between):
        """
        The exchange over a gloo group (Comm.host_staged): outgoing slabs are copied to pinned host buffers on
        the comm stream, behind the event that marks their producer; the host waits for those copies, then posts
        the sends and receives on host memory.  finish_halo_exchange copies what arrived into the guard slabs on
        the comm stream.  The host blocks here, which a rehearsal of the N > 1 flow on one GPU can afford; runs
        on several GPUs use RCCL.
        """
        torch = self._torch()
        dist = self.comm.dist
        if self._staging is None:
            self._staging = {}
        with self.patches[0].execution_context:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
                for p in self.patches:
                    if getattr(p, "plan", None) is not None:
                        p.plan.set_side_stream(self._comm_stream)
            ready = torch.cuda.Event()
            ready.record()
            if between is not None:
                between()
            self._comm_stream.wait_event(ready)
            staged = []
            with torch.cuda.stream(self._comm_stream):
                for n, (kind, t, peer) in enumerate(ops):
                    key = (n, kind, tuple(t.shape))
                    host = self._staging.get(key)
                    if host is None:
                        host = self._staging[key] = torch.empty(tuple(t.shape), dtype=t.dtype).pin_memory()
                    if kind == "send":
                        host.copy_(t, non_blocking=True)
                    staged.append((kind, host, t, peer))
                copied = torch.cuda.Event()
                copied.record()
            # (the comm stream is in order: the previous exchange's copies out of these buffers are complete too)
            copied.synchronize()
            reqs = [dist.isend(host, peer) if kind == "send" else dist.irecv(host, peer) for kind, host, _, peer in staged]
        return ("host-staged", reqs, staged)

    def finish_halo_exchange(self, reqs, join=True):
        """Wait for the halos on the comm stream; with `join` the compute stream then waits for it too."""
        if not reqs:
            return
        if isinstance(reqs, tuple) and reqs[0] == "host-staged":
            torch = self._torch()
            for r in reqs[1]:
                r.wait()
            with self.patches[0].execution_context:
                with torch.cuda.stream(self._comm_stream):
                    for kind, host, t, _ in reqs[2]:
                        if kind == "recv":
                            t.copy_(host, non_blocking=True)
                if join:
                    torch.cuda.current_stream().wait_stream(self._comm_stream)
            return
        if self.backend.mode == "hip":
            torch = self._torch()
            with self.patches[0].execution_context:
                if reqs != "stream-ordered":
                    with torch.cuda.stream(self._comm_stream):
                        for r in reqs:
                            r.wait()
                if join:
                    torch.cuda.current_stream().wait_stream(self._comm_stream)
        else:
            for r in reqs:
                r.wait()

    def set_bc(self, array):
        """
        Reference-style boundary fill for the unfused path (cbdiso_2d.py:553-586):
        halos between patches, then outflow copies on the domain edges in i and
        on both edges in j (which also fills the corners).
        """
        self.finish_halo_exchange(self.start_halo_exchange(array))
        ng = self.num_guard
        for p in self.patches:
            pc = getattr(p, array)
            with p.execution_context:
                if p.first:
                    for i in range(ng):
                        pc[i] = pc[ng]
                if p.last:
                    for i in range(pc.shape[0] - ng, pc.shape[0]):
                        pc[i] = pc[-ng - 1]
                for j in range(ng):
                    pc[:, j] = pc[:, ng]
                for j in range(pc.shape[1] - ng, pc.shape[1]):
                    pc[:, j] = pc[:, -ng - 1]

    def synchronize(self):
        if self.backend.mode == "hip":
            torch = self._torch()
            for p in self.patches:
                with p.execution_context:
                    torch.cuda.synchronize()
