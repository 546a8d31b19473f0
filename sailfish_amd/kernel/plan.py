"""
Launch plans: the fused stage entry points with their constant arguments marshalled once.

``Library``'s ``kernel[shape](*args)`` converts every one of the 60-odd arguments of
``sfhip_cbd{iso,gam}_2d_stage`` on every call and looks torch's current stream up (about 20 us of
Python per launch).  That is the reference's calling convention and stays the drop-in path; the
device-resident solvers launch through a ``StagePlan`` instead (include/sfhip.h, "launch plans"):
constants go to the C side when the patch is created, a launch passes the arrays, the two masses
and a dozen scalars.  The same C function performs the launch either way.
"""
import ctypes

_c_double_p = ctypes.POINTER(ctypes.c_double)


def _bind(module):
    """ctypes signatures of the plan entry points (declared as plain C in include/sfhip.h)."""
    if getattr(module, "_sfhip_plan_bound", False):
        return
    i, d, p = ctypes.c_int, ctypes.c_double, ctypes.c_void_p
    module.sfhip_plan_create_cbdiso_2d.argtypes = [i, i, d, d, d, d, d, d, d, d, i, d, d, i, d, d, d, i, i, p]
    module.sfhip_plan_create_cbdgam_2d.argtypes = [i, i, d, d, d, d, d, d, d, d, d, d, i, d, d, d, d, d, d, i, i, i, p]
    module.sfhip_plan_set_side_stream.argtypes = [i, p]
    module.sfhip_plan_destroy.argtypes = [i]
    module.sfhip_plan_stage.argtypes = [i, _c_double_p, _c_double_p, p, p, p, d, d, i, i, i, d, p, p, d, d, p, i, d]
    module.sfhip_plan_host_wait.argtypes = [i, d, _c_double_p]
    module.sfhip_plan_step_rk2.argtypes = [i, p, p, p, p, d, d, d, _c_double_p, _c_double_p, d, d, d, _c_double_p]
    module.sfhip_kepler_orbital_state.argtypes = [d, d, d, d, d, _c_double_p]
    module.sfhip_plan_step_rk2.restype = i
    module.sfhip_plan_step_timing.argtypes = [i, _c_double_p]
    module.sfhip_plan_step_timing.restype = i
    module.sfhip_kepler_orbital_state.restype = i
    for name in ("create_cbdiso_2d", "create_cbdgam_2d", "set_side_stream", "destroy", "stage", "host_wait"):
        getattr(module, "sfhip_plan_" + name).restype = i
    _c_int_p = ctypes.POINTER(ctypes.c_int)
    module.sfhip_rccl_bind.argtypes = [ctypes.c_char_p]
    module.sfhip_step_group_create.argtypes = [i, _c_int_p, _c_int_p, _c_int_p, _c_int_p, p, i, p]
    module.sfhip_step_group_destroy.argtypes = [i]
    module.sfhip_step_group_rk2.argtypes = [i, ctypes.POINTER(p), ctypes.POINTER(p), p, p, d, d, d, _c_double_p,
                                            _c_double_p, d, d, d, _c_double_p]
    module.sfhip_step_group_timing.argtypes = [i, _c_double_p]
    for name in ("sfhip_rccl_bind", "sfhip_step_group_create", "sfhip_step_group_destroy", "sfhip_step_group_rk2",
                 "sfhip_step_group_timing"):
        getattr(module, name).restype = i
    module._sfhip_plan_bound = True


def _ptr(t):
    return None if t is None else t.data_ptr()


class StagePlan:
    """One patch's plan.  Created with the patch's device current; launches go to the stream that was
    torch's current stream at that moment (``on_side_stream``: to the stream given to ``set_side_stream``)."""

    def __init__(self, lib, patch, kind):
        import torch

        from .library import KernelError

        self._error = KernelError
        self.lib = lib
        self.module = module = lib.module
        _bind(module)
        self.stream = stream = torch.cuda.current_stream().cuda_stream
        create = module.sfhip_plan_create_cbdiso_2d if kind == "cbdiso_2d" else module.sfhip_plan_create_cbdgam_2d
        self.id = create(*patch.plan_constants(), ctypes.c_void_p(stream))
        if self.id < 0:
            raise KernelError(f"sfhip_plan_create_{kind} failed: {lib.last_error()}")
        self._stage = module.sfhip_plan_stage
        self._masses = (ctypes.c_double * 18)()
        self._next = (ctypes.c_double * 8)()
        self._packed = (None, None)
        self._packed_next = (None, None)
        self._host_value = ctypes.c_double(0.0)
        self._host_ref = ctypes.byref(self._host_value)

    def close(self):
        if self.id >= 0:
            self.module.sfhip_plan_destroy(self.id)
            self.id = -1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check_stream(self):
        """A plan launches on the stream that was torch's current stream when it was made; the torch-side work of
        a step (slot copies, all-reduces) follows torch's current stream.  They must be the same stream."""
        import torch

        now = torch.cuda.current_stream().cuda_stream
        if now != self.stream:
            raise self._error(f"launch plan made on stream {self.stream:#x}, but torch's current stream is {now:#x}: "
                              "step the solver on the stream it was constructed on")

    def set_side_stream(self, stream):
        self.module.sfhip_plan_set_side_stream(self.id, ctypes.c_void_p(stream.cuda_stream))

    def stage(self, patch, m1, m2, w1, w2, un, rd, wr, rk_param, dt, i_begin, i_end, want_wavespeed, floor, wmax,
              amax_device, dt_dx, dt_cfl, zero_slot, on_side_stream, host_seq=0.0):
        """Marshal now, launch later: returns the zero-argument callable that performs the C call."""
        if self._packed[0] is not m1 or self._packed[1] is not m2:  # same tuple objects for a given time
            self._masses[:] = [float(v) for v in (*m1.kernel_args(), *m2.kernel_args())]
            self._packed = (m1, m2)
        nxt = None
        if want_wavespeed and w1 is not None:
            if self._packed_next[0] is not w1 or self._packed_next[1] is not w2:
                self._next[:] = [w1.position_x, w1.position_y, w1.mass, w1.softening_length,
                                 w2.position_x, w2.position_y, w2.mass, w2.softening_length]
                self._packed_next = (w1, w2)
            nxt = self._next
        args = (self.id, self._masses, nxt, _ptr(un), rd.data_ptr(), wr.data_ptr(), rk_param, dt, i_begin, i_end,
                want_wavespeed, floor, _ptr(wmax), _ptr(amax_device), dt_dx, dt_cfl, _ptr(zero_slot),
                1 if on_side_stream else 0, host_seq)

        def fire():
            status = self._stage(*args)
            if status != 0:
                raise self._error(f"sfhip_plan_stage failed ({status}): {self.lib.last_error()}")

        return fire

    def set_kepler_binary(self, elements4, shared4):
        """Orbital elements and shared mass constants for `step_rk2` (sailfish_amd.setups: `kepler_binary`)."""
        self._elements = (ctypes.c_double * 4)(*elements4)
        self._shared = (ctypes.c_double * 4)(*shared4)
        self._result = (ctypes.c_double * 3)()
        self._step = self.module.sfhip_plan_step_rk2

    def step_rk2(self, state, other, slot_current, slot_next, dx, cfl, time, wavespeed_floor, floor_fraction, host_seq):
        """One whole RK2 step (sfhip_plan_step_rk2).  Returns (redo, amax, dt, time_after)."""
        status = self._step(self.id, state.data_ptr(), other.data_ptr(), slot_current.data_ptr(), slot_next.data_ptr(),
                            dx, cfl, time, self._elements, self._shared, wavespeed_floor, floor_fraction, host_seq, self._result)
        if status not in (0, 1, 2):
            raise self._error(f"sfhip_plan_step_rk2 failed ({status}): {self.lib.last_error()}")
        r = self._result
        if status == 2:  # NaN wavespeed: dt and the time after the step are NaN, stage 1 was not launched
            return False, r[0], float("nan"), float("nan")
        return status == 1, r[0], r[1], r[2]

    def step_timing(self):
        """Mean host microseconds per step_rk2 call by phase since the last call of this method:
        (stage-0 launch, wait for the hand-over, dt + orbit, stage-1 launch, steps)."""
        out = (ctypes.c_double * 5)()
        self.module.sfhip_plan_step_timing(self.id, out)
        return tuple(out)

    def host_wait(self, host_seq):
        """The reduced wavespeed handed over by the stage that was launched with this `host_seq`."""
        if self.module.sfhip_plan_host_wait(self.id, host_seq, self._host_ref) != 0:
            raise self._error(f"sfhip_plan_host_wait failed: {self.lib.last_error()}")
        return self._host_value.value



PEER_EDGE, PEER_LOCAL = -1, -2  # neighbour codes of sfhip_step_group_create (a rank number otherwise)


class StepGroup:
    """
    All the patches a process owns, stepped from one C call (include/sfhip.h, "step groups"): per stage the slab
    copies between the group's patches, the interior launches, one RCCL group of sends and receives on the side
    stream, the rows behind the halos, and after the last stage the MAX all-reduce of the wavespeed slot.

    `patches`: in global patch order, each with a StagePlan; `global_ids`: their patch numbers; `lower` / `upper`:
    per patch PEER_EDGE, PEER_LOCAL or the rank that owns the neighbour; `rccl`: an RcclComm (kernel/rccl.py) when
    any neighbour is on another rank or the slot is to be reduced over ranks.
    """

    def __init__(self, lib, patches, global_ids, lower, upper, rccl, reduce_over_ranks, side_stream, elements4, shared4):
        from .library import KernelError

        self._error = KernelError
        self.lib = lib
        self.module = module = lib.module
        _bind(module)
        n = len(patches)
        if rccl is not None and module.sfhip_rccl_bind(rccl.library_path.encode()) != 0:
            raise KernelError(f"sfhip_rccl_bind failed: {lib.last_error()}")
        ints = ctypes.c_int * n
        self.id = module.sfhip_step_group_create(
            n, ints(*[p.plan.id for p in patches]), ints(*global_ids), ints(*lower), ints(*upper),
            rccl.comm if rccl is not None else None, 1 if reduce_over_ranks else 0,
            ctypes.c_void_p(side_stream.cuda_stream) if side_stream is not None else None)
        if self.id < 0:
            raise KernelError(f"sfhip_step_group_create failed: {lib.last_error()}")
        self._keep = (rccl, side_stream)
        self._n = n
        self._state = (ctypes.c_void_p * n)()
        self._other = (ctypes.c_void_p * n)()
        self._elements = (ctypes.c_double * 4)(*elements4)
        self._shared = (ctypes.c_double * 4)(*shared4)
        self._result = (ctypes.c_double * 3)()
        self._step = module.sfhip_step_group_rk2

    def close(self):
        if self.id >= 0:
            self.module.sfhip_step_group_destroy(self.id)
            self.id = -1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step_rk2(self, states, others, slot_current, slot_next, dx, cfl, time, wavespeed_floor, floor_fraction, host_seq):
        """One RK2 step of every patch (sfhip_step_group_rk2).  Returns (redo, amax, dt, time_after)."""
        for k in range(self._n):
            self._state[k] = states[k].data_ptr()
            self._other[k] = others[k].data_ptr()
        status = self._step(self.id, self._state, self._other, slot_current.data_ptr(), slot_next.data_ptr(), dx, cfl, time,
                            self._elements, self._shared, wavespeed_floor, floor_fraction, host_seq, self._result)
        if status not in (0, 1, 2):
            raise self._error(f"sfhip_step_group_rk2 failed ({status}): {self.lib.last_error()}")
        r = self._result
        if status == 2:
            return False, r[0], float("nan"), float("nan")
        return status == 1, r[0], r[1], r[2]

    def step_timing(self):
        out = (ctypes.c_double * 5)()
        self.module.sfhip_step_group_timing(self.id, out)
        return tuple(out)
