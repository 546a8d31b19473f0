/*
 * sfhip.h -- C ABI of libsfhip.so, the MI355X (gfx950) implementation of
 * sailfish's cbdiso_2d / cbdgam_2d finite-volume update.
 *
 * Drop-in boundary.  The reference loads a shared object with ctypes.CDLL and
 * calls `void <kernel>(int ni, int nj, <args in source order>)`
 * (sailfish/kernel/library.py:99-117, 252-264).  The entry points below that
 * carry a reference name have EXACTLY the reference's argument list (types,
 * order, meaning), so `lib.<kernel>[shape](*args)` in sailfish's Patch classes
 * works unchanged; the only differences are
 *   - `double *` arguments are DEVICE pointers (HBM of the current device);
 *   - the launch is asynchronous on the calling thread's current stream
 *     (sfhip_set_stream), like the reference's cupy `gpu` mode;
 *   - functions return int (0 = ok) instead of void; a caller that ignores the
 *     value, as a `void` binding does, is still ABI-correct.  The reference has
 *     no error channel at all; ours is sfhip_last_error().
 *
 * This header is also the single source of truth for the Python loader:
 * sailfish_amd/kernel/parse_api.py reads the declarations (one argument per
 * line, optional `// :: <constraint>` comments -- the convention of
 * sailfish/kernel/parse_api.py:29-84 and doc/source/kernels.rst:138-157) and
 * derives ctypes signatures and the kernel rank (number of leading ints).
 *
 * Array layout everywhere: C-contiguous (ni + 4, nj + 4, NCONS) fp64, two guard
 * zones per side (cbdiso_2d.c:443-445).  Kernels write interior zones only.
 */
#ifndef SFHIP_H
#define SFHIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SFHIP_API

/* ---------------------------------------------------------------- runtime --
 * No counterpart in the reference's C kernels; they replace what cupy's
 * current-device / current-stream state does for its `gpu` mode
 * (sailfish/kernel/system.py:122-135, library.py:118-135).
 */
int sfhip_set_device(int device);
int sfhip_device_count(void);
int sfhip_set_stream(void *stream); /* hipStream_t; thread-local; NULL = default stream */
int sfhip_synchronize(void);        /* hipStreamSynchronize on the current stream */
const char *sfhip_last_error(void); /* thread-local message of the last failing call */
int sfhip_selftest_lane_shift(const double *in64, double *out_below64, double *out_above64);

/* Device self test of the fast reciprocal / rsqrt / sqrt used by the stage
 * kernels: out3n = [1/x | 1/sqrt(x) | sqrt(x)] for n positive inputs. */
SFHIP_API int sfhip_selftest_fastmath(
    int n,
    double *x, // :: $.numel() >= n
    double *out3n) // :: $.numel() >= 3 * n
;

/* The driver's `dt = dx / amax * cfl` (driver.py:509-510) evaluated on the device
 * with the same IEEE operations, from the reduced wavespeed; n must be 1. */
SFHIP_API int sfhip_timestep(
    int n,
    double dx,
    double cfl,
    double *wavespeed_max, // :: $.numel() >= 1
    double *dt_out) // :: $.numel() >= 1
;

/* Replaces `self.wavespeeds.max()` (cbdiso_2d.py:174, cbdgam_2d.py:152): max of
 * n non-negative doubles into *out (device scalar). */
SFHIP_API int sfhip_max_nonneg(
    int n,
    double *x, // :: $.numel() >= n
    double *out) // :: $.numel() >= 1
;

/* Replaces initial_condition() for the isothermal circumbinary-disk setup: the Python double
 * loop over CircumbinaryDisk.primitive (sailfish/solvers/cbdiso_2d.py:33-46,
 * sailfish/setups/circumbinary_disk.py:82-93) plus the host-to-device copy (cbdiso_2d.py:92-94).
 * Fills the interior of a patch's primitive array -- zones [i0, i0 + ni) x [0, nj) of the mesh
 * with origin (x0, y0) and spacing (dx, dy) -- in device memory; guards are left alone.
 * softening2 = softening_length^2 as the host forms it.  Only + * / sqrt max, correctly rounded:
 * bit-identical to the reference's loop. */
SFHIP_API int sfhip_circumbinary_disk_primitive(
    int ni,
    int nj,
    double x0,
    double y0,
    double dx,
    double dy,
    int i0, // :: $ >= 0
    double softening2,
    double sigma,
    double *primitive) // :: $.shape == (ni + 4, nj + 4, 3)
;

/* ---------------------------------------------------------------- cbdiso_2d */

/* Replaces cbdiso_2d_advance_rk, sailfish/solvers/cbdiso_2d.c:385-604
 * (caller: Patch.advance_rk, cbdiso_2d.py:186-240). */
SFHIP_API int cbdiso_2d_advance_rk(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double *conserved_rk, // :: $.shape == (ni + 4, nj + 4, 3)
    double *primitive_rd, // :: $.shape == (ni + 4, nj + 4, 3)
    double *primitive_wr, // :: $.shape == (ni + 4, nj + 4, 3)
    double buffer_surface_density,
    double buffer_central_mass,
    double buffer_driving_rate,
    double buffer_outer_radius,
    double buffer_onset_width,
    int buffer_is_enabled,
    double x1, // point mass 1
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2, // point mass 2
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double cs2, // equation of state
    double mach_squared,
    int eos_type,
    double nu, // kinematic viscosity coefficient
    double a, // RK parameter
    double dt, // timestep
    double velocity_ceiling,
    double density_floor)
;

/* Replaces cbdiso_2d_primitive_to_conserved, cbdiso_2d.c:606-624
 * (caller: Patch.recompute_conserved, cbdiso_2d.py:176-184). */
SFHIP_API int cbdiso_2d_primitive_to_conserved(
    int ni,
    int nj,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 3)
    double *conserved) // :: $.shape == (ni + 4, nj + 4, 3)
;

/* Replaces cbdiso_2d_point_mass_source_term, cbdiso_2d.c:626-664
 * (caller: Patch.point_mass_source_term, cbdiso_2d.py:105-137). */
SFHIP_API int cbdiso_2d_point_mass_source_term(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double x1, // particle
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 3)
    double *cons_rate) // :: $.shape == (ni + 4, nj + 4, 3)
;

/* Replaces cbdiso_2d_wavespeed, cbdiso_2d.c:666-722
 * (caller: Patch.maximum_wavespeed, cbdiso_2d.py:139-174).  Bit-exact. */
SFHIP_API int cbdiso_2d_wavespeed(
    int ni, // mesh
    int nj,
    double patch_xl,
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double soundspeed2, // equation of state
    double mach_squared,
    int eos_type,
    double x1, // point mass 1
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2, // point mass 2
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 3)
    double *wavespeed) // :: $.shape == (ni + 4, nj + 4)
;

/* Fused stage.  cbdiso_2d_advance_rk plus the passes the reference runs around
 * it: Solver.set_bc outflow copies (cbdiso_2d.py:553-586) become index clamps,
 * Patch.recompute_conserved (cbdiso_2d.py:176-184) becomes un_mode 2, and
 * Patch.maximum_wavespeed of the OUTPUT state (cbdiso_2d.py:139-174) becomes
 * want_wavespeed (0 off, 1 accumulate into *wavespeed_max, 2 zero it first).
 * Only zone rows [i_begin, i_end) are updated (interior / boundary split while
 * halos are in flight).  The w* arguments are the point masses at the time the
 * output state belongs to.  wavespeed_floor is a filter, not a result: zones whose
 * speed is below it are not folded in, so the reduced value is the exact global
 * maximum whenever it comes out >= wavespeed_floor (after the cross-rank MAX); a
 * caller that sees a smaller value recomputes with *_max_wavespeed.  Pass 0 to
 * fold every zone.  dt_device, when not NULL, overrides `dt` with a timestep that
 * sfhip_timestep left in device memory, so the first stage of a step can be
 * launched before the host has read the wavespeed back.  amax_device, when not NULL
 * (then dt_device must be NULL), goes one step further: the kernel forms the
 * driver's `dt = dx / amax * cfl` (driver.py:509-510) itself from the reduced
 * wavespeed in device memory -- dt_dx / *amax_device * dt_cfl, IEEE operations in
 * that order, bit-identical to the host's -- so no separate timestep launch exists.
 * zero_slot, when not NULL, is set to 0.0 by the launch: the slot the step's LAST
 * stage will fold the next wavespeed into (want_wavespeed 1), which saves that
 * stage's memset; it must not be the slot this launch itself folds into. */
SFHIP_API int sfhip_cbdiso_2d_stage(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double *un_source, // :: $.shape == (ni + 4, nj + 4, 3)
    double *primitive_rd, // :: $.shape == (ni + 4, nj + 4, 3)
    double *primitive_wr, // :: $.shape == (ni + 4, nj + 4, 3)
    double buffer_surface_density,
    double buffer_central_mass,
    double buffer_driving_rate,
    double buffer_outer_radius,
    double buffer_onset_width,
    int buffer_is_enabled,
    double x1, // point mass 1
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2, // point mass 2
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double cs2, // equation of state
    double mach_squared,
    int eos_type,
    double nu, // kinematic viscosity coefficient
    double a, // RK parameter
    double dt, // timestep
    double velocity_ceiling,
    double density_floor,
    int un_mode, // :: $ in [0, 1, 2]
    int clamp_ilo,
    int clamp_ihi,
    int i_begin, // :: 0 <= $ <= ni
    int i_end, // :: i_begin <= $ <= ni
    int want_wavespeed, // :: $ in [0, 1, 2]
    double wavespeed_floor, // :: $ >= 0.0
    double wx1, // point masses at the output time
    double wy1,
    double wmass1,
    double wsoftening_length1,
    double wx2,
    double wy2,
    double wmass2,
    double wsoftening_length2,
    double *wavespeed_max, // :: $.numel() >= 1
    double *dt_device, // :: $.numel() >= 1
    double *amax_device, // :: $.numel() >= 1
    double dt_dx,
    double dt_cfl,
    double *zero_slot) // :: $.numel() >= 1
;

/* Wavespeed kernel + max in one pass (no per-zone array): replaces the pair
 * cbdiso_2d_wavespeed + xp.max(), cbdiso_2d.py:139-174.  Bit-exact. */
SFHIP_API int sfhip_cbdiso_2d_max_wavespeed(
    int ni,
    int nj,
    double patch_xl,
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double soundspeed2,
    double mach_squared,
    int eos_type,
    double x1,
    double y1,
    double mass1,
    double softening_length1,
    double x2,
    double y2,
    double mass2,
    double softening_length2,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 3)
    double *wavespeed_max) // :: $.numel() >= 1
;

/* ---------------------------------------------------------------- cbdgam_2d */

/* Replaces cbdgam_2d_advance_rk, sailfish/solvers/cbdgam_2d.c:439-683
 * (caller: Patch.advance_rk, cbdgam_2d.py:162-215). */
SFHIP_API int cbdgam_2d_advance_rk(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double *conserved_rk, // :: $.shape == (ni + 4, nj + 4, 4)
    double *primitive_rd, // :: $.shape == (ni + 4, nj + 4, 4)
    double *primitive_wr, // :: $.shape == (ni + 4, nj + 4, 4)
    double gamma_law_index,
    double buffer_surface_density,
    double buffer_surface_pressure,
    double buffer_central_mass,
    double buffer_driving_rate,
    double buffer_outer_radius,
    double buffer_onset_width,
    int buffer_is_enabled,
    double x1, // point mass 1
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2, // point mass 2
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double alpha, // other
    double a,
    double dt,
    double velocity_ceiling,
    double cooling_coefficient,
    double mach_ceiling,
    double density_floor,
    double pressure_floor,
    int constant_softening)
;

/* Replaces cbdgam_2d_wavespeed, cbdgam_2d.c:685-708 (caller cbdgam_2d.py:145-152). */
SFHIP_API int cbdgam_2d_wavespeed(
    int ni,
    int nj,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 4)
    double *wavespeed, // :: $.shape == (ni + 4, nj + 4)
    double gamma_law_index)
;

/* Replaces cbdgam_2d_primitive_to_conserved, cbdgam_2d.c:710-729 (caller cbdgam_2d.py:154-160). */
SFHIP_API int cbdgam_2d_primitive_to_conserved(
    int ni,
    int nj,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 4)
    double *conserved, // :: $.shape == (ni + 4, nj + 4, 4)
    double gamma_law_index)
;

/* Replaces cbdgam_2d_point_mass_source_term, cbdgam_2d.c:731-784 (caller cbdgam_2d.py:104-143). */
SFHIP_API int cbdgam_2d_point_mass_source_term(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double x1, // point mass 1
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2, // point mass 2
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    int which_mass, // :: $ in [1, 2]
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 4)
    double *cons_rate, // :: $.shape == (ni + 4, nj + 4, 4)
    int constant_softening,
    double gamma_law_index)
;

/* Fused stage for cbdgam_2d; flags as for sfhip_cbdiso_2d_stage.  The
 * gamma-law wavespeed needs no point masses (cbdgam_2d.c:685-708). */
SFHIP_API int sfhip_cbdgam_2d_stage(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double *un_source, // :: $.shape == (ni + 4, nj + 4, 4)
    double *primitive_rd, // :: $.shape == (ni + 4, nj + 4, 4)
    double *primitive_wr, // :: $.shape == (ni + 4, nj + 4, 4)
    double gamma_law_index,
    double buffer_surface_density,
    double buffer_surface_pressure,
    double buffer_central_mass,
    double buffer_driving_rate,
    double buffer_outer_radius,
    double buffer_onset_width,
    int buffer_is_enabled,
    double x1, // point mass 1
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2, // point mass 2
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double alpha, // other
    double a,
    double dt,
    double velocity_ceiling,
    double cooling_coefficient,
    double mach_ceiling,
    double density_floor,
    double pressure_floor,
    int constant_softening,
    int un_mode, // :: $ in [0, 1, 2]
    int clamp_ilo,
    int clamp_ihi,
    int i_begin, // :: 0 <= $ <= ni
    int i_end, // :: i_begin <= $ <= ni
    int want_wavespeed, // :: $ in [0, 1, 2]
    double wavespeed_floor, // :: $ >= 0.0
    double *wavespeed_max, // :: $.numel() >= 1
    double *dt_device, // :: $.numel() >= 1
    double *amax_device, // :: $.numel() >= 1
    double dt_dx,
    double dt_cfl,
    double *zero_slot) // :: $.numel() >= 1
;

SFHIP_API int sfhip_cbdgam_2d_max_wavespeed(
    int ni,
    int nj,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 4)
    double gamma_law_index,
    double *wavespeed_max) // :: $.numel() >= 1
;

/* Replaces the arithmetic of Solver.reductions() for one patch (cbdiso_2d.py:370-503):
 * the four Patch.point_mass_source_term launches (mass 1 / 2 x gravity-only /
 * accretion-only, cbdiso_2d.py:376-379), the per-diagnostic temporaries of get_field
 * (:383-478) and `f.sum()` (:486), fused into one pass over the primitive array.
 * Masses are passed unscaled; the kernel forms the gravity-only (sink_rate = 0) and
 * accretion-only (mass = 0) variants as Patch.point_mass_source_term does (:129-131).
 * xcoord / ycoord are Patch.cell_center_coordinate_arrays (cbdiso_2d.py:96-103), the
 * moment arms of torque, sigma_m1, eccentricity_vector, angular_momentum.
 * spec holds 8 doubles per diagnostic: quantity code (0..2 = rate of conserved field
 * q: "mdot", "fx", "fy"; 10 torque, 11 power, 12 sigma_m1, 13 eccentricity_vector,
 * 14 angular_momentum, 15 mass), which_mass (1, 2, 3 = "both"), gravity, accretion,
 * has_radial_cut, r0, r1, unused.  sums receives (re, im) per diagnostic, NOT yet
 * multiplied by dx*dy (the caller does, after adding the patches, :499).
 * workspace_len must be >= sfhip_reductions_workspace_len(ni, nj). Deterministic. */
SFHIP_API int sfhip_cbdiso_2d_reductions(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double x1,
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2,
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 3)
    double *xcoord, // :: $.numel() >= ni
    double *ycoord, // :: $.numel() >= nj
    int num_diagnostics, // :: 1 <= $ <= 8
    double *spec, // :: $.numel() >= 8 * num_diagnostics
    int workspace_len,
    double *workspace, // :: $.numel() >= workspace_len
    double *sums) // :: $.numel() >= 2 * num_diagnostics
;

/* No counterpart in the reference.  For callers that cannot allocate (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc) and so cannot bring the
 * workspace either: sfhip_cbdiso_2d_reductions with its block partials in a device array
 * built into the library (128 KiB per GPU, held until the process ends) instead of a
 * caller-provided workspace.  Same kernels, same launch geometry, bit-identical sums.
 * Calls no hipMalloc / hipFree or other allocating routine.  Limit: the launch must fit
 * 1024 blocks of partials, i.e. nj <= 262 144 (any ni); larger patches are refused with
 * -1 and a message.  All calls on one GPU share that GPU's array: calls that overlap on
 * one GPU (different streams or threads) race; calls on one stream are ordered. */
SFHIP_API int sfhip_cbdiso_2d_reductions_noalloc(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double x1,
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2,
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 3)
    double *xcoord, // :: $.numel() >= ni
    double *ycoord, // :: $.numel() >= nj
    int num_diagnostics, // :: 1 <= $ <= 8
    double *spec, // :: $.numel() >= 8 * num_diagnostics
    double *sums) // :: $.numel() >= 2 * num_diagnostics
;

/* As above for cbdgam_2d (cbdgam_2d.py:341-443); quantity codes 0..3 are the rates
 * of the four conserved fields. */
SFHIP_API int sfhip_cbdgam_2d_reductions(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double x1,
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2,
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    int constant_softening,
    double gamma_law_index,
    double *primitive, // :: $.shape == (ni + 4, nj + 4, 4)
    double *xcoord, // :: $.numel() >= ni
    double *ycoord, // :: $.numel() >= nj
    int num_diagnostics, // :: 1 <= $ <= 8
    double *spec, // :: $.numel() >= 8 * num_diagnostics
    int workspace_len,
    double *workspace, // :: $.numel() >= workspace_len
    double *sums) // :: $.numel() >= 2 * num_diagnostics
;

/* Number of doubles the reductions entry points need in `workspace` for an (ni, nj) patch. */
SFHIP_API int sfhip_reductions_workspace_len(
    int ni,
    int nj)
;

/* ---------------------------------------------------------------- launch plans --
 * No counterpart in the reference (its Patch marshals all 41/44 arguments per call,
 * cbdiso_2d.py:198-238).  sfhip_cbd{iso,gam}_2d_stage take 60-odd scalars; through
 * ctypes that costs the Python host ~20 us per launch, which bounds the step rate of
 * small patches.  A plan stores what does not change between launches (mesh, physics
 * constants, boundary clamps, stream) and sfhip_plan_stage passes only the arrays,
 * the two masses, the RK parameter, dt, the row range and the wavespeed arguments;
 * the launch is the same as the full entry point's (un_mode 2 whenever un_source is
 * given).  Host-side arrays here are HOST pointers:
 *   masses      18 doubles: x, y, vx, vy, mass, softening_length, sink_rate,
 *               sink_radius, sink_model of mass 1, then of mass 2, at the stage time;
 *   next_masses  8 doubles: x, y, mass, softening_length of the two masses at the time
 *               the output state belongs to (cbdiso_2d with want_wavespeed; else NULL).
 * A plan also owns 64 bytes of pinned host memory through which a stage hands the reduced
 * wavespeed to the host (sfhip_plan_host_wait); it is allocated with the plan -- the
 * per-step calls allocate nothing.
 * create returns a plan id >= 0, or -1 (sfhip_last_error); the others 0 or an error.
 * compute_stream / side_stream: hipStream_t the launches go to (on_side_stream 0 / 1).
 */
int sfhip_plan_create_cbdiso_2d(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double buffer_surface_density, double buffer_driving_rate, double buffer_outer_radius,
    double buffer_onset_width, int buffer_is_enabled, double cs2, double mach_squared, int eos_type, double nu,
    double velocity_ceiling, double density_floor, int clamp_ilo, int clamp_ihi, void *compute_stream);
int sfhip_plan_create_cbdgam_2d(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr, double gamma_law_index,
    double buffer_surface_density, double buffer_surface_pressure, double buffer_driving_rate,
    double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled, double alpha,
    double velocity_ceiling, double cooling_coefficient, double mach_ceiling, double density_floor,
    double pressure_floor, int constant_softening, int clamp_ilo, int clamp_ihi, void *compute_stream);
int sfhip_plan_set_side_stream(int plan, void *side_stream);
int sfhip_plan_destroy(int plan);
int sfhip_plan_stage(int plan, const double *masses, const double *next_masses,
                     double *un_source, double *primitive_rd, double *primitive_wr, double a, double dt,
                     int i_begin, int i_end, int want_wavespeed, double wavespeed_floor, double *wavespeed_max,
                     double *amax_device, double dt_dx, double dt_cfl, double *zero_slot, int on_side_stream,
                     double host_seq);
/* host_seq != 0 (needs amax_device): as soon as the launch starts on the device one lane writes
 * *amax_device and then host_seq to the plan's pinned memory, and sfhip_plan_host_wait(plan, host_seq, &v)
 * returns that value: the host learns the reduced wavespeed without a copy or an event in the stream. */
int sfhip_plan_host_wait(int plan, double host_seq, double *host_value);
/* A whole RK2 step of a one-patch solver from one call, for point masses on a Kepler orbit (the
 * circumbinary setups): stage 0 with dt formed in the kernel from *slot_current (and *slot_next zeroed, the
 * reduced wavespeed handed to the host through the plan's pinned memory), `dt = dx / amax * cfl`
 * (driver.py:509-510), the stage times of Patch.advance_rk (cbdiso_2d.py:239), the orbit at those times --
 * OrbitalElements.orbital_state (physics/kepler.py:329-373, 414-421) restated on the host in C with the same
 * libm, bit-identical -- and stage 1 folding the new wavespeed into *slot_next.  `state` holds p^n and
 * receives the new state, `other` is the second buffer (device); elements4 = semimajor axis, total mass,
 * mass ratio, eccentricity and shared4 = softening length, sink rate, sink radius, sink model (host);
 * result3 (host) receives amax, dt and the time after the step.  Returns 1 when amax is below
 * wavespeed_floor (only stage 0 was launched; the caller recomputes the wavespeed exactly and repeats), 2 when
 * the wavespeed and with it dt is NaN (only stage 0 was launched; the reference fails at this point, in the Kepler
 * solve of the next stage time).  Failures are negative (-1 for bad arguments, minus the hipError_t of a failed
 * launch), so that they cannot be taken for one of these outcomes.
 * Stage 1's epilogue evaluates exactly the zones that reach floor_fraction * amax (the caller's next
 * wavespeed_floor): any value in [0, 1) gives the same bits, higher is faster until steps start to repeat. */
int sfhip_plan_step_rk2(int plan, double *state, double *other, double *slot_current, double *slot_next,
                        double dx, double cfl, double time, const double *elements4, const double *shared4,
                        double wavespeed_floor, double floor_fraction, double host_seq, double *result3);
/* Diagnostic: mean host microseconds per sfhip_plan_step_rk2 call by phase since the last call of this function
 * (stage-0 launch, wait for the hand-over, dt + orbit, stage-1 launch) and the number of steps; out5 on the host. */
int sfhip_plan_step_timing(int plan, double *out5);
/* ----------------------------------------------------------------- step groups --
 * No counterpart in the reference, whose single Python process walks over its patches and devices per stage
 * (solvers/cbdiso_2d.py:536-586: set_bc's slice copies between patches, then one kernel call per patch;
 * driver.py:506-512 around it).  A step group is ALL the patches one process owns (in global patch order, on
 * one device and one compute stream) with, for either side of each patch, what lies there:
 *     -1  the domain edge (outflow: index clamp inside the kernel),
 *     -2  the previous / next patch of the group (a device-to-device copy of the 2-row slab),
 *    >=0  the rank of the process that owns the neighbouring patch: the slab travels as ncclSend / ncclRecv on
 *         `nccl_comm`, all of a stage's messages in one ncclGroup on `side_stream`, posted in the order of their
 *         names (destination patch, lower | upper guard slab) on every rank.
 * sfhip_step_group_rk2 is sfhip_plan_step_rk2 for the group: per stage slab copies, the rows that need no halo
 * (compute stream), the exchange and the <= 2 + 2 rows per patch behind it (side stream, ordered by two events),
 * and after the last stage -- with reduce_over_ranks -- ncclAllReduce(MAX) of *slot_next over the communicator,
 * on the doubles' bit patterns as ncclUint64 (exact for non-negative values, and a NaN survives).  `state` and
 * `other` are HOST arrays of the patches' device pointers.  Outcomes 0 / 1 / 2 and negative failures as there;
 * *slot_current must hold the maximum over all ranks, so that every rank takes the same decisions.
 * RCCL is bound at run time: sfhip_rccl_bind(path) dlopens the library file the communicator was made with
 * (libsfhip.so does not link against RCCL). */
int sfhip_rccl_bind(const char *library_path);
int sfhip_step_group_create(int npatches, const int *plans, const int *global_ids, const int *lower, const int *upper,
                            void *nccl_comm, int reduce_over_ranks, void *side_stream);
int sfhip_step_group_destroy(int group);
int sfhip_step_group_rk2(int group, double *const *state, double *const *other, double *slot_current, double *slot_next,
                         double dx, double cfl, double time, const double *elements4, const double *shared4,
                         double wavespeed_floor, double floor_fraction, double host_seq, double *result3);
int sfhip_step_group_timing(int group, double *out5);

/* Host only, no GPU involved: initial data of the GAMMA-LAW circumbinary disk (sailfish/setups/circumbinary_disk.py:
 * 94-108 through initial_condition(), sailfish/solvers/cbdgam_2d.py:30-43 -- the Python double loop over every zone),
 * zones [i0, i0 + ni) x [0, nj) of the mesh with origin (x0, y0) and spacing (dx, dy), into the HOST array
 * out_host[ni][nj][4] (no guards).  The formulas call pow and exp, which no device library reproduces bit for bit, so
 * this one stays on the host: the same libm calls the interpreter makes per point (x ** y is pow(x, y), math.exp is
 * exp), in the same order, spread over `threads` host threads (<= 0: as many as the process may use, at most 64) --
 * bit-identical to the reference's loop on the same machine, at 67 M zones (8192^2) in under a second instead of
 * minutes.  exponent_sigma / exponent_pressure are -3/5 and -3/2 as the caller's own divisions give them. */
int sfhip_host_gamma_law_disk_primitive(int ni, int nj, double x0, double y0, double dx, double dy, int i0, double softening2,
                                        double sigma, double pressure, double exponent_sigma, double exponent_pressure,
                                        int threads, double *out_host);

/* Host only, no GPU involved: the orbital state (mass, x, y, vx, vy of body 1, then body 2) into 10 host doubles. */
int sfhip_kepler_orbital_state(double semimajor_axis, double total_mass, double mass_ratio, double eccentricity,
                               double time_since_periapse, double *out10_host);

/* ---------------------------------------------------------------------------
 * srhd_2d -- 2-D special-relativistic hydrodynamics on a log-spherical, homologously
 * expanding mesh (BASELINE.json configs[4]).
 * State arrays are (ni + 4, nj, 4): two guard zones in i only, fields fastest.
 * Allocation / synchronisation: the first call on a device, and the first call for a new
 * (nj, polar_extent), builds a table of sin/cos of the polar face angles on the host (libm,
 * for bit-identical zone volumes) and uploads it with a blocking copy; tables live until the
 * process ends (at most 32 meshes per device).  Every later call is an asynchronous launch.
 * --------------------------------------------------------------------------- */

/* The reference compiles PLM_THETA and MACH_CEILING into the kernel library
 * (Library(..., define_macros=...), srhd_2d.py:225-233; srhd_2d.c:17-27).  Here they
 * are library state, set once per Solver; n must be 1. */
SFHIP_API int sfhip_srhd_2d_set_macros(
    int n,
    double plm_theta,
    double mach_ceiling)
;

/* Replaces srhd_2d_primitive_to_conserved, srhd_2d.c:511-538 (caller: Patch.__init__,
 * srhd_2d.py:127-133).  No guard zones on either array. */
SFHIP_API int srhd_2d_primitive_to_conserved(
    int ni,
    int nj,
    double *face_positions, // :: $.shape == (ni + 1,)
    double *primitive, // :: $.shape == (ni, nj, 4)
    double *conserved, // :: $.shape == (ni, nj, 4)
    double polar_extent,
    double scale_factor) // :: $ >= 0.0
;

/* Replaces srhd_2d_conserved_to_primitive, srhd_2d.c:544-575 (caller:
 * Patch.recompute_primitive, srhd_2d.py:144-154).  Newton iteration on the pressure
 * from the previous primitive[..., 3]; where the reference printf()s and exit(1)s
 * (srhd_2d.c:164-184) this entry point counts the zone in the library's failure
 * counter (sfhip_srhd_2d_failures) and carries on. */
SFHIP_API int srhd_2d_conserved_to_primitive(
    int ni,
    int nj,
    double *face_positions, // :: $.shape == (ni + 1,)
    double *conserved1, // :: $.shape == (ni + 4, nj, 4)
    double *conserved2, // :: $.shape == (ni + 4, nj, 4)
    double *primitive, // :: $.shape == (ni + 4, nj, 4)
    double polar_extent,
    double scale_factor) // :: $ >= 0.0
;

/* Replaces srhd_2d_max_wavespeeds, srhd_2d.c:581-609 (caller: Patch.maximum_wavespeed,
 * srhd_2d.py:179-188). */
SFHIP_API int srhd_2d_max_wavespeeds(
    int ni,
    int nj,
    double *face_positions, // :: $.shape == (ni + 1,)
    double *primitive, // :: $.shape == (ni + 4, nj, 4)
    double *wavespeed, // :: $.shape == (ni, nj)
    double adot) // :: $ >= 0.0
;

/* Replaces srhd_2d_advance_rk, srhd_2d.c:616-749 (caller: Patch.advance_rk,
 * srhd_2d.py:156-177). */
SFHIP_API int srhd_2d_advance_rk(
    int ni,
    int nj,
    double *face_positions, // :: $.shape == (ni + 1,)
    double *conserved_rk, // :: $.shape == (ni + 4, nj, 4)
    double *primitive_rd, // :: $.shape == (ni + 4, nj, 4)
    double *conserved_rd, // :: $.shape == (ni + 4, nj, 4)
    double *conserved_wr, // :: $.shape == (ni + 4, nj, 4)
    double polar_extent,
    double a0,
    double adot,
    double time,
    double rk_param,
    double dt,
    double jet_mdot,
    double jet_gamma_beta,
    double jet_theta,
    double jet_duration,
    int num_first_order_zones)
;

/* Fused srhd_2d stage: srhd_2d_advance_rk (srhd_2d.c:616-749) followed, zone by zone while the
 * updated state is still in registers, by the srhd_2d_conserved_to_primitive call (srhd_2d.c:544-575)
 * with which the reference opens the NEXT stage (Solver.advance_rk, srhd_2d.py:339-341):
 * conserved_wr receives the updated state as advance_rk leaves it, conserved_floored and
 * primitive_wr what conserved_to_primitive would then write into conserved2 / primitive
 * (guess: primitive_rd[..., 3] of the zone; volume at scale_factor_next).  Removes the
 * recovery pass (128 B per zone, HBM-bound).  Not for patches with the jet boundary
 * (jet_mdot > 0 is refused: zones the reference leaves unwritten keep stale data there). */
SFHIP_API int sfhip_srhd_2d_stage(
    int ni,
    int nj,
    double *face_positions, // :: $.shape == (ni + 1,)
    double *conserved_rk, // :: $.shape == (ni + 4, nj, 4)
    double *primitive_rd, // :: $.shape == (ni + 4, nj, 4)
    double *conserved_rd, // :: $.shape == (ni + 4, nj, 4)
    double *conserved_wr, // :: $.shape == (ni + 4, nj, 4)
    double polar_extent,
    double a0,
    double adot,
    double time,
    double rk_param,
    double dt,
    int num_first_order_zones,
    double scale_factor_next, // :: $ >= 0.0
    double *primitive_wr, // :: $.shape == (ni + 4, nj, 4)
    double *conserved_floored) // :: $.shape == (ni + 4, nj, 4)
;

/* Number of zones for which srhd_2d_conserved_to_primitive hit one of the reference's
 * fatal conditions since the last call (device counter read back: synchronises).
 * out[0] receives the count; n must be 1. */
SFHIP_API int sfhip_srhd_2d_failures(
    int n,
    double *out) // :: $.numel() >= 1
;

/* ---------------------------------------------------------------------------
 * cbdisodg_2d -- third-order modal discontinuous-Galerkin variant of the isothermal
 * binary-accretion solver.  State arrays are
 * (ni + 2, nj + 2, 3, 3, 3): ONE guard zone per side, then conserved field q and the
 * Legendre orders m (x), n (y); the scheme uses the six slots with m + n < 3.
 * --------------------------------------------------------------------------- */

/* Replaces cbdisodg_2d_slope_limit, cbdisodg_2d.c:498-605 (caller: Patch.slope_limit,
 * cbdisodg_2d.py:211-230).  Copies every zone's 27 weights and, in troubled zones,
 * zeroes the higher moments and replaces the two slopes by the reference's minmod
 * value -- which its `int` return type (cbdisodg_2d.c:455) truncates to an integer;
 * reproduced.  The mesh and point-mass arguments are unused there and here.  Bit-exact. */
SFHIP_API int cbdisodg_2d_slope_limit(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double point_mass1_x, // point mass 1
    double point_mass1_y,
    double point_mass2_x, // point mass 2
    double point_mass2_y,
    double *weights1, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double *weights2) // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
;

/* Replaces cbdisodg_2d_advance_rk, cbdisodg_2d.c:607-983 (caller: Patch.advance_rk,
 * cbdisodg_2d.py:232-285).  Writes the 18 used slots of each interior zone of weights2. */
SFHIP_API int cbdisodg_2d_advance_rk(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double *weights0, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double *weights1, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double *weights2, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double buffer_surface_density,
    double buffer_central_mass,
    double buffer_driving_rate,
    double buffer_outer_radius,
    double buffer_onset_width,
    int buffer_is_enabled,
    double x1, // point mass 1
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2, // point mass 2
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double soundspeed2, // equation of state
    double mach_squared,
    int eos_type,
    double nu, // kinematic viscosity coefficient
    double rk_param, // RK parameter
    double dt, // timestep
    double velocity_ceiling)
;

/* Replaces cbdisodg_2d_wavespeed, cbdisodg_2d.c:1108-1176 (caller: Patch.maximum_wavespeed,
 * cbdisodg_2d.py:169-205).  Bit-exact. */
SFHIP_API int cbdisodg_2d_wavespeed(
    int ni, // mesh
    int nj,
    double patch_xl,
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double soundspeed2, // equation of state
    double mach_squared,
    int eos_type,
    double x1, // point mass 1
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2, // point mass 2
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double velocity_ceiling,
    double *weights, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double *wavespeed) // :: $.shape == (ni + 2, nj + 2)
;

/* cbdisodg_2d_wavespeed fused with the `self.wavespeeds.max()` that follows it
 * (cbdisodg_2d.py:205): the maximum goes straight to *wavespeed_max (device scalar,
 * overwritten).  Bit-exact. */
SFHIP_API int sfhip_cbdisodg_2d_max_wavespeed(
    int ni, // mesh
    int nj,
    double patch_xl,
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double soundspeed2, // equation of state
    double mach_squared,
    int eos_type,
    double x1, // point mass 1
    double y1,
    double mass1,
    double softening_length1,
    double x2, // point mass 2
    double y2,
    double mass2,
    double softening_length2,
    double velocity_ceiling,
    double *weights, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double *wavespeed_max) // :: $.numel() >= 1
;

/* cbdisodg_2d_slope_limit on rows [i_begin, i_end) with the outflow guard fill that
 * precedes it in the reference (Solver.set_bc, cbdisodg_2d.py:511-544) folded in as
 * index clamps: with clamp_j the j-neighbours beyond the patch are the edge zone itself,
 * with clamp_lower_i / clamp_upper_i likewise for the first / last row (set these on the
 * patches that touch the domain edge; other patches read their guard rows, filled by the
 * halo exchange). */
SFHIP_API int sfhip_cbdisodg_2d_limit(
    int ni,
    int nj,
    double *weights1, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double *weights2, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    int clamp_lower_i,
    int clamp_upper_i,
    int clamp_j,
    int i_begin, // :: 0 <= $ <= ni
    int i_end) // :: i_begin <= $ <= ni
;

/* cbdisodg_2d_advance_rk on rows [i_begin, i_end) with the same index clamps, the time
 * step optionally taken from device memory (dt_device != NULL overrides dt: the step is
 * launched before the host knows dt, sfhip_timestep) and, with want_wavespeed, the
 * cbdisodg_2d_wavespeed + max of the UPDATED state (point masses at their next positions
 * wx1..wrsoft2) folded into *wavespeed_max by an exact atomic max: 2 = zero it first,
 * 1 = accumulate into it, 0 = off. */
SFHIP_API int sfhip_cbdisodg_2d_stage(
    int ni,
    int nj,
    double patch_xl, // mesh
    double patch_xr,
    double patch_yl,
    double patch_yr,
    double *weights0, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double *weights1, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double *weights2, // :: $.shape == (ni + 2, nj + 2, 3, 3, 3)
    double buffer_surface_density,
    double buffer_central_mass,
    double buffer_driving_rate,
    double buffer_outer_radius,
    double buffer_onset_width,
    int buffer_is_enabled,
    double x1, // point mass 1
    double y1,
    double vx1,
    double vy1,
    double mass1,
    double softening_length1,
    double sink_rate1,
    double sink_radius1,
    int sink_model1,
    double x2, // point mass 2
    double y2,
    double vx2,
    double vy2,
    double mass2,
    double softening_length2,
    double sink_rate2,
    double sink_radius2,
    int sink_model2,
    double soundspeed2, // equation of state
    double mach_squared,
    int eos_type,
    double nu, // kinematic viscosity coefficient
    double rk_param, // RK parameter
    double dt, // timestep
    double velocity_ceiling,
    int clamp_lower_i,
    int clamp_upper_i,
    int clamp_j,
    int i_begin, // :: 0 <= $ <= ni
    int i_end, // :: i_begin <= $ <= ni
    int want_wavespeed, // :: 0 <= $ <= 2
    double wx1,
    double wy1,
    double wmass1,
    double wsoftening_length1,
    double wx2,
    double wy2,
    double wmass2,
    double wsoftening_length2,
    double *wavespeed_max, // :: $.numel() >= 1
    double *dt_device) // :: $.numel() >= 1
;

#ifdef __cplusplus
}
#endif
#endif /* SFHIP_H */
