// sf_iso.hpp -- cbdiso_2d (isothermal circumbinary disk) kernels for gfx950.
//
// Design:
//   * One wave owns a strip of 64 columns (60 outputs + 2 halo lanes per side)
//     and marches down a chunk of rows.  The x-direction stencil (reach 2) lives
//     in a register window that slides with the march; the y-direction stencil
//     is the neighbouring lanes, reached with DPP wave shifts.  No LDS tiles, no
//     barriers; every primitive row is loaded from HBM once per strip.  (LDS holds
//     one small per-wave table of the rows' lane-independent quantities, round 3.)
//   * Slopes are evaluated once per zone and HLLE fluxes once per face (the
//     reference evaluates every face twice, cbdiso_2d.c:525-563); both views
//     perform identical IEEE operations, so results agree bit-for-bit wherever
//     libm's pow/exp are not involved.
//   * Optional fusions that remove whole passes over HBM: outflow boundary
//     handled by index clamping (no set_bc launches), u^n recomputed from the
//     retained p^n (no primitive_to_conserved pass), and the CFL wavespeed
//     max-reduction folded into the epilogue of the last RK stage.
#pragma once
#include "sf_device.hpp"

namespace sf {

struct IsoParams {
    int ni, nj;                 // interior zones of the patch
    double xl, yl, dx, dy;      // patch origin and spacing (dx = (xr-xl)/ni on the host, cbdiso_2d.c:440)
    double inv_dx, inv_dy;      // 1/dx, 1/dy (host, correctly rounded)
    Mass m[2];                  // point masses at the stage time
    // Keplerian buffer (cbdiso_2d.c:213-246); everything that depends only on
    // scalars is evaluated once on the host with the libm calls the reference
    // makes per zone.
    double buf_sd, buf_mc, buf_rate, buf_rout, buf_ron, buf_omega;
    double buf_ron2, buf_rate_scale; // ron^2 and rate*omega/(rout-ron)
    int buf_on;
    int eos;                    // 1 globally, 2 locally isothermal (cbdiso_2d.c:194-211)
    int eos_local;              // eos == 2 and at least one mass > 0: cs2 varies in space
    double cs2, mach2;
    double cs2_uniform, cs_uniform, inv_mach2; // used when !eos_local / eos_local
    double nu, a, dt, vmax, dfloor;
    // uniform values the kernel would otherwise form with vector instructions (there is no scalar fp64
    // unit) and keep in vector registers: evaluated on the host, they arrive in SGPRs
    double half_nu, half_inv_dx, half_inv_dy, one_minus_a, sink_reach2[2];
    // fusion / boundary flags
    int clamp_ilo, clamp_ihi;   // patch touches the domain edge in i: clamp instead of reading guards
    int clamp_j;                // outflow in j by clamping (always true for slab decomposition)
    int un_mode;                // 0: a == 0, no u^n; 1: u^n from a conserved array; 2: u^n = p2c(p^n array)
    int rows_per_chunk;         // rows marched by one wave
    int reverse;                // chunks are handed out from the last rows to the first (see launch_iso_stage)
    int i_begin, i_end;         // zone-row range [i_begin, i_end) to update (interior/boundary split for halo overlap)
    int want_wavespeed;         // fold max wavespeed of the OUTPUT state into *wmax
    double ws_floor;            // zones slower than this are not folded in (see include/sfhip.h)
    const double *dt_ptr;       // when set, the timestep is read from device memory instead of `dt`
    // when amax_ptr is set, the timestep is FORMED here from the reduced wavespeed in device memory:
    // dt = dt_dx / *amax_ptr * dt_cfl, the driver's expression (driver.py:509-510) in IEEE operations
    const double *amax_ptr;
    double dt_dx, dt_cfl;
    double *zero_ptr;           // when set, one lane stores 0.0 there (the slot the NEXT fold accumulates into)
    // when set (with amax_ptr): one lane writes *amax_ptr to host_out[0] and then host_seq to host_out[1] --
    // pinned host memory the host polls, so it learns the reduced wavespeed without a copy or an event
    double *host_out;
    double host_seq;
    Mass mw[2];                 // masses at the time the output state belongs to (for want_wavespeed)
};

// ---- exact (IEEE op-for-op) pieces: wavespeed / dt path ----------------------

// softened potential, cbdiso_2d.c:70-95
__device__ __forceinline__ double iso_potential(const Mass *m, double x, double y)
{
    double phi = 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (m[p].mass > 0.0) {
            double ddx = x - m[p].x;
            double ddy = y - m[p].y;
            double r2 = ddx * ddx + ddy * ddy;
            phi -= m[p].mass / sqrt(r2 + m[p].rsoft * m[p].rsoft);
        }
    }
    return phi;
}

// cbdiso_2d.c:194-211
__device__ __forceinline__ double iso_cs2(const IsoParams &P, const Mass *m, double x, double y)
{
    if (P.eos == 1) return P.cs2;
    if (P.eos == 2) return -iso_potential(m, x, y) / P.mach2;
    return 1.0;
}

// The exact wavespeed of one zone as an out-of-line call: the fused epilogue reaches it for a
// handful of zones per launch, and inlined its IEEE sqrt/divide expansions would raise the register
// allocation of the whole stage kernel (3 -> 2 waves per SIMD).
__device__ __attribute__((noinline)) double iso_exact_speed(const IsoParams *Pk, double x, double y, double vx, double vy)
{
#pragma clang fp contract(off)
    // few arguments on purpose (each one is a register pair the call site has to free); the rest is
    // read from the parameter block here
    return max_speed(vx, vy, iso_cs2(*Pk, Pk->mw, x, y));
}

// ---- fast pieces: the stage kernel (within ~1 ulp of the above) -------------

// cs2 and cs at a point for the locally isothermal EOS: cs2 = sum m/sqrt(r^2+rs^2) / M^2.
// `c0`, `c1` are the lane-invariant parts (y - y_p)^2 + rs_p^2 of the two masses.
// FULL: the caller has established (once, on the host) that the EOS is locally isothermal and both
// masses are positive, so the tests below are compile-time true; the arithmetic is the same.
// `d0`, `d1`: x - x_p of the two masses (wave-uniform along a row: the march reads them from its row table).
template <bool FULL = false>
__device__ __forceinline__ void iso_sound_fast(const IsoParams &P, const Mass *m, double d0, double d1, double c0, double c1,
                                               double &cs2, double &cs)
{
#pragma clang fp contract(fast)
    if (!FULL && !P.eos_local) { cs2 = P.cs2_uniform; cs = P.cs_uniform; return; }
    double q = 0.0;
    if (FULL || m[0].mass > 0.0) q += m[0].mass * fast_rsqrt(d0 * d0 + c0);
    if (FULL || m[1].mass > 0.0) q += m[1].mass * fast_rsqrt(d1 * d1 + c1);
    cs2 = q * P.inv_mach2;
    cs = fast_sqrt(cs2);
}

// HLLE flux through one face, cbdiso_2d.c:285-380.  DIR 0: x-face, 1: y-face.
template <int DIR>
__device__ __forceinline__ void iso_hlle(const double *pl, const double *pr, double cs2, double cs, double *f)
{
#pragma clang fp contract(fast)
    // F = (Fl ap - Fr am - (Ul - Ur) ap am) / (ap - am), regrouped by side:
    //   F = wl (Fl - am Ul) - wr (Fr - ap Ur),  wl = ap / (ap - am), wr = am / (ap - am),
    // and Fl - am Ul = (vl - am) Ul + p e: per side one factor A = wl (vl - am), B = wr (vr - ap) for the advected
    // part of all three components, and one pressure term cs2 (wl rho_l - wr rho_r) on the normal momentum.
    // Same value as the textbook form up to rounding; 11 instructions per face fewer.
    const double ul1 = pl[1] * pl[0], ul2 = pl[2] * pl[0];
    const double ur1 = pr[1] * pr[0], ur2 = pr[2] * pr[0];
    const double vl = pl[1 + DIR], vr = pr[1 + DIR];
    // min(vl - cs, vr - cs) = min(vl, vr) - cs exactly (rounding is monotone): one subtraction instead of two
    const double am = min0_f64(min_f64(vl, vr) - cs);
    const double ap = max0_f64(max_f64(vl, vr) + cs);
    const double rden = fast_rcp(ap - am);
    const double wl = ap * rden, wr = am * rden;
    const double A = wl * (vl - am), B = wr * (vr - ap);
    const double pt = cs2 * (wl * pl[0] - wr * pr[0]);
    f[0] = A * pl[0] - B * pr[0];
    f[1] = A * ul1 - B * ur1;
    f[2] = A * ul2 - B * ur2;
    f[1 + DIR] += pt;
}

// Primitive recovery with floors, cbdiso_2d.c:268-283 (the division through a ~1-ulp reciprocal: fast_rcp).
__device__ __forceinline__ void iso_c2p(const double *u, double vmax, double dfloor, double *p)
{
#pragma clang fp contract(fast)
    double rho = max_s_f64(u[0], dfloor);
    double rinv = fast_rcp(rho);
    p[0] = rho;
    p[1] = copysign(min_abs_s_f64(u[1] * rinv, vmax), u[1]);
    p[2] = copysign(min_abs_s_f64(u[2] * rinv, vmax), u[2]);
}

// One point mass: gravity + sink, cbdiso_2d.c:97-175.  Accumulates into dc.
// `ddx` = x - x_p (uniform along a row); `ddy` = y - y_p and `cy` = ddy^2 + rs^2 are lane-invariant and precomputed; `sink_reach2` =
// (4 r_sink)^2 + rs^2.  `cold` points at the same mass in memory: what only the sink branch needs
// (a handful of waves per launch) is loaded there, on demand, instead of occupying SGPRs throughout.
// `ds`: where the sink's contributions go (the diagnostics keep gravity and accretion apart); `dc` when null.
__device__ __forceinline__ double iso_mass_source(double ddx, double mmass, double sink_reach2, const Mass *cold,
                                                  double ddy, double cy, double dt,
                                                  const double *prim, double *dc, double *ds = nullptr)
{
#pragma clang fp contract(fast)
    if (!ds) ds = dc;
    double sigma = prim[0];
    double r2s = ddx * ddx + cy;
    const double y = fast_rsqrt(r2s); // 1/sqrt(r^2 + rs^2): returned, the wavespeed epilogue reuses it
    double fnum = sigma * mmass * (y * y * y) * dt; // pow(r^2 + rs^2, -1.5) of the reference
    dc[1] -= fnum * ddx;
    dc[2] -= fnum * ddy;
    if (r2s < sink_reach2) { // superset of dr < 4 r_sink; exact test inside
        const double sink_radius = cold->sink_radius;
        double rs4 = 4.0 * sink_radius;
        double r2 = ddx * ddx + ddy * ddy;
        double dr = sqrt(r2);
        if (dr < rs4) { // lazily evaluated like the C ternary: r_sink == 0 never divides
            const double mvx = cold->vx, mvy = cold->vy;
            const int sink_model = cold->sink_model;
            double q = dr / sink_radius;
            double q2 = q * q;
            double rate = cold->sink_rate * exp_cold(-(q2 * q2));
            double mdot = 0.0;
            if (rate > 0.0) mdot = -rate * sigma;
            else if (rate < 0.0) mdot = -rate;
            if (sink_model == 1) {
                ds[0] += dt * mdot;
                ds[1] += dt * mdot * prim[1];
                ds[2] += dt * mdot * prim[2];
            } else if (sink_model == 2) {
                double rinv = 1.0 / (dr + 1e-12);
                double rhx = ddx * rinv;
                double rhy = ddy * rinv;
                double dv = (prim[1] - mvx) * rhx + (prim[2] - mvy) * rhy;
                double vxs = dv * rhx + mvx;
                double vys = dv * rhy + mvy;
                ds[0] += dt * mdot;
                ds[1] += dt * mdot * vxs;
                ds[2] += dt * mdot * vys;
            } else if (sink_model == 3) {
                ds[0] += dt * mdot;
            }
        }
    }
    return y;
}

// convenience form for the small kernels (everything from the by-value struct)
__device__ __forceinline__ void iso_mass_source(const Mass &m, double x1, double ddy, double cy, double dt,
                                                const double *prim, double *dc, double *ds = nullptr)
{
    double rs4 = 4.0 * m.sink_radius;
    iso_mass_source(x1 - m.x, m.mass, rs4 * rs4 + m.rsoft * m.rsoft, &m, ddy, cy, dt, prim, dc, ds);
}

// Keplerian buffer, cbdiso_2d.c:213-246
template <bool FULL = false>
__device__ __forceinline__ void iso_buffer(const IsoParams &P, double xc, double yc, double yc2, const double *u, double *dc)
{
#pragma clang fp contract(fast)
    if (!FULL && !P.buf_on) return;
    double rc2 = xc * xc + yc2;
    if (rc2 > P.buf_ron2) {
        double rinv = fast_rsqrt(rc2);
        double rc = sqrt_from_rsqrt(rc2, rinv);
        double vk2 = P.buf_mc * rinv;                           // central_mass / rc
        double vk = fast_sqrt(vk2);
        double pk = P.buf_sd * vk * rinv;                       // sigma v_kep / rc
        double br = P.buf_rate_scale * (rc - P.buf_ron) * P.dt; // driving_rate omega (rc-ron)/(rout-ron) dt
        dc[0] -= (u[0] - P.buf_sd) * br;
        dc[1] -= (u[1] + pk * yc) * br;
        dc[2] -= (u[2] - pk * xc) * br;
    }
}

// ---------------------------------------------------------------------------
// One Runge-Kutta stage (cbdiso_2d_advance_rk, cbdiso_2d.c:385-604).
// grid: 1-D, 256 threads = 4 waves; wave w of the grid owns column strip
// (w % nstrips) and row chunk (w / nstrips).
// ---------------------------------------------------------------------------
// WS: 0 no wavespeed epilogue; 1 epilogue with its own masses P.mw (output time != stage time: RK1, RK3);
//     2 epilogue when P.mw == P.m bit for bit (last stage of RK2): the softened distances of the
//     source terms are reused for the sound-speed estimate.
// What a row contributes to the faces around it, from the window (rows below, itself, above) and its y-slopes:
// twice its x-slopes and its strain (cbdiso_2d.c:248-263, slopes carried doubled).  And what it leaves behind for
// the x-face above it: the lower-side state of that face and its viscous stresses.  The bring-up before the march
// and the march itself both call these, and a zone's bits must not depend on which of the two evaluated its
// neighbour: explicit operations only, nothing the compiler could contract one way here and another way there.
__device__ __forceinline__ void iso_row_slopes(const double *pa, const double *pb, const double *pc, const double *gyb,
                                               bool visc, double hidx, double hidy,
                                               double *gxb, double &sxx, double &syy, double &sxy)
{
#pragma unroll
    for (int q = 0; q < 3; ++q) gxb[q] = plm2<18>(pa[q], pb[q], pc[q]); // 2 x slope
    sxx = 0.0; syy = 0.0; sxy = 0.0;
    if (visc) {
        const double c43 = 4.0 / 3.0, c23 = 2.0 / 3.0;
        const double ax = gxb[1] * hidx, by = gyb[2] * hidy;
        sxx = __builtin_fma(c43, ax, -(c23 * by));
        syy = __builtin_fma(c43, by, -(c23 * ax));
        sxy = __builtin_fma(gxb[2], hidx, gyb[1] * hidy);
    }
}

__device__ __forceinline__ void iso_row_leaves(const double *pb, const double *gxb, double sxx, double sxy, double quarter,
                                               double *pmx, double &txx, double &txy)
{
#pragma unroll
    for (int q = 0; q < 3; ++q) pmx[q] = __builtin_fma(quarter, gxb[q], pb[q]);
    txx = pb[0] * sxx;
    txy = pb[0] * sxy;
}

// FULL: the benchmark's physics, established on the host -- viscosity on, locally isothermal EOS with both
//       masses positive, buffer on.  The uniform tests on those flags (and the selects and zero defaults the
//       untaken sides need) drop out of the march.  The formulas are unchanged; the compiler contracts a few
//       of them into FMAs differently, so FULL and generic runs agree to the last places, not bit for bit
//       (tests/test_hip_solver.py::test_full_physics_instantiation).  One run uses one instantiation
//       throughout, on every patch and rank.
template <int WS, bool UN, bool FULL>
#if !defined(SFHIP_ISO_A0_WAVES)
#define SFHIP_ISO_A0_WAVES 3
#endif
#if defined(SFHIP_DIAG_ISO_TWO_WAVES)  // diagnosis builds only: what the third resident wave per SIMD is worth
__attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
__global__ __launch_bounds__(256, (WS == 0 && !UN) ? SFHIP_ISO_A0_WAVES : 3) void iso_stage_kernel(
    IsoParams P, const double *__restrict__ prd, const double *unsrc,
    double *pwr, double *__restrict__ wmax, int nstrips, int nchunks)
{
#pragma clang fp contract(fast)
    const int lane = threadIdx.x & 63;
    // wave-uniform by construction: tell the compiler, so rows, branches and addresses stay scalar
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int strip = gw % nstrips;
    if (gw / nstrips >= nchunks) return; // whole wave leaves together
    const int chunk = P.reverse ? nchunks - 1 - gw / nstrips : gw / nstrips;
    if (P.dt_ptr) P.dt = *P.dt_ptr; // dt computed on the device by sfhip_timestep (uniform scalar load)
    if (P.amax_ptr) P.dt = device_timestep(P.dt_dx, *P.amax_ptr, P.dt_cfl);
    if (gw == 0 && lane == 0) step_bookkeeping(P.zero_ptr, P.host_out, P.host_seq, P.amax_ptr);
    // The same parameter block, addressed through a pointer whose origin the optimiser cannot see:
    // loads through it stay inside the rarely taken branches that need them (sink interior, exact
    // wavespeed) instead of being hoisted into SGPRs that the whole march would then have to spill.
    const IsoParams *Pk = (const IsoParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(Pk));

    const int ni = P.ni, nj = P.nj;
    const long sj = 3, si = 3L * (nj + 2 * NG);

    // column owned by this lane (guard-inclusive index) and where it loads from
    const int J = strip * STRIP + lane;
    int Jld = J;
    if (P.clamp_j) Jld = Jld < NG ? NG : (Jld > nj + NG - 1 ? nj + NG - 1 : Jld);
    else           Jld = Jld > nj + 2 * NG - 1 ? nj + 2 * NG - 1 : Jld;
    const long col = Jld * sj;
    const bool store_lane = lane >= HALO && lane < WAVE - HALO && J < nj + NG;
    const int jz = J - NG; // zone index in j (halo lanes may be out of range; values unused)
    const double yc = mesh_coord(P.yl, jz + 0.5, P.dy);
    const double yf = mesh_coord(P.yl, (double)jz, P.dy); // lower y-face of this lane's zone

    // rows (guard-inclusive index) updated by this wave
    const int I0 = NG + P.i_begin + chunk * P.rows_per_chunk;
    int I1 = I0 + P.rows_per_chunk;
    if (I1 > NG + P.i_end) I1 = NG + P.i_end;
    const int rlo = P.clamp_ilo ? NG : 0;
    const int rhi = P.clamp_ihi ? ni + NG - 1 : ni + 2 * NG - 1;
    const bool visc = FULL || P.nu > 0.0;
    const double hnu = P.half_nu;
    // lane-invariant pieces of the distances to the two masses
    const double dyc0 = yc - P.m[0].y, dyc1 = yc - P.m[1].y;
    const double cyc0 = dyc0 * dyc0 + P.m[0].rsoft * P.m[0].rsoft;
    const double cyc1 = dyc1 * dyc1 + P.m[1].rsoft * P.m[1].rsoft;
    const double cyf0 = (yf - P.m[0].y) * (yf - P.m[0].y) + P.m[0].rsoft * P.m[0].rsoft;
    const double cyf1 = (yf - P.m[1].y) * (yf - P.m[1].y) + P.m[1].rsoft * P.m[1].rsoft;
    const double yc2 = yc * yc;
    const double reach0 = P.sink_reach2[0], reach1 = P.sink_reach2[1]; // (4 r_sink)^2 + rs^2
    double cyw0 = 0, cyw1 = 0;
    if (WS == 1) {
        cyw0 = (yc - P.mw[0].y) * (yc - P.mw[0].y) + P.mw[0].rsoft * P.mw[0].rsoft;
        cyw1 = (yc - P.mw[1].y) * (yc - P.mw[1].y) + P.mw[1].rsoft * P.mw[1].rsoft;
    }
    const double hidx = P.half_inv_dx, hidy = P.half_inv_dy; // slopes are carried doubled
    // 0.25 is not an inline constant on gfx9: as a literal it forces v_mov + v_fmac (VOP2) for every
    // p +- 0.25 g.  Held opaquely in an SGPR pair it is a plain operand of a 3-address v_fma_f64.
    double quarter = 0.25;
    asm volatile("" : "+s"(quarter));

    // dt/dx and dt/dy, wave-uniform: formed once (dt may have come from device memory) and held in SGPRs
    const double kx = uniform_value(P.inv_dx * P.dt), ky = uniform_value(P.inv_dy * P.dt);

    double pa[3], pb[3], pc[3];
    double gyb[3], gyc[3];
    double pmx[3] = {1, 0, 0}, txx = 0, txy = 0;
    // what the row in flight has collected so far of -(dF/dx + dG/dy) dt: its lower x-face flux and its
    // y-flux difference.  The upper x-face flux is subtracted when the next iteration has evaluated it.
    // (Every row goes through the same operations, so a zone's bits do not depend on chunk boundaries.)
    double acc[3] = {0, 0, 0};
    double amax = 0.0;            // exact running maximum of this lane
    double athr = P.ws_floor;     // what an estimate has to reach to be worth the exact evaluation

    // The window is filled before the march: rows I0-2, I0-1, I0 with the y-slopes of the middle one (no
    // operation here can contract, so these slopes have the bits the loop would have given them).
    // Software prefetch: a row is requested one iteration before it enters the window.
    double pn[3], un[3] = {0, 0, 0};
    auto row = [&](int r) { return prd + (r < rlo ? rlo : (r > rhi ? rhi : r)) * si + col; };
    {
        const double *s0 = row(I0 - 2), *s1 = row(I0 - 1), *s2 = row(I0);
#pragma unroll
        for (int q = 0; q < 3; ++q) { pa[q] = s0[q]; pb[q] = s1[q]; pc[q] = s2[q]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            { const double d = pb[q] - from_lane_below(pb[q]); gyb[q] = plm2_diffs<18>(d, from_lane_above(d)); }
            { const double d = pc[q] - from_lane_below(pc[q]); gyc[q] = plm2_diffs<18>(d, from_lane_above(d)); }
        }
        // All three rows have to have arrived before the march starts: a load still pending on the loop's
        // entry edge makes the compiler wait for it inside the loop, in every iteration, behind the
        // requests just issued there.
#pragma unroll
        for (int q = 0; q < 3; ++q) asm volatile("" : "+v"(pa[q]));
    }
    // Row table.  What a row contributes through its x-coordinate alone -- the face and centre positions and their
    // distances to the two masses -- is the same for all 64 lanes, but there is no scalar fp64 unit: formed in the
    // march it costs every lane 16 vector instructions per row.  Instead lane l forms the entry of row T - 2 + l,
    // once per 60 rows, into this wave's 64 entries of LDS, and the march reads the two entries an iteration needs
    // (same address in every lane: a broadcast, no VALU).  The operations are the ones the march performed itself.
    __shared__ double row_table[4][WAVE][ROW_ENTRY];
    double (*const tab)[ROW_ENTRY] = row_table[threadIdx.x >> 6];
    auto fill_rows = [&](int T) { // entries of rows T-2 ... T+61, slot = row & 63
        const int rho = T - 2 + lane;
        const double xfr = mesh_coord(P.xl, (double)(rho - NG), P.dx);  // lower x-face of row rho
        const double xcr = mesh_coord(P.xl, (rho - NG) + 0.5, P.dx);    // its centre
        double *e = tab[rho & (WAVE - 1)];
        e[0] = xfr - P.m[0].x; e[1] = xfr - P.m[1].x;
        e[2] = xcr - P.m[0].x; e[3] = xcr - P.m[1].x;
        e[4] = xcr;
    };
    constexpr bool want_un = UN;
    const long ucol = (long)(J < nj + 2 * NG ? J : nj + 2 * NG - 1) * sj;
    auto enter = [&](const double *row_in) { // a row enters the window: its y-slopes from the neighbouring lanes
        // (2 x slope; the upper one-sided difference is the next lane's lower one), then the window slides
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            gyb[q] = gyc[q];
            { const double d = row_in[q] - from_lane_below(row_in[q]); gyc[q] = plm2_diffs<18>(d, from_lane_above(d)); }
            pa[q] = pb[q]; pb[q] = pc[q]; pc[q] = row_in[q];
        }
    };
    // Bring-up: all that the faces of row I0-1 have to provide is the lower side of the x-face between rows I0-1
    // and I0.  (Round 1 ran a whole iteration of the march for it, both fluxes included: one wasted iteration per
    // chunk, which is a fifth of the work of the 3-row chunks of a small patch.)
    {
        const double *src = row(I0 + 1);
        pn[0] = src[0]; pn[1] = src[1]; pn[2] = src[2];
        double gx[3], sxx, syy, sxy;
        iso_row_slopes(pa, pb, pc, gyb, visc, hidx, hidy, gx, sxx, syy, sxy);
        iso_row_leaves(pb, gx, sxx, sxy, quarter, pmx, txx, txy);
        enter(pn);
    }

    // One iteration of the march; on entry `pc` holds row r.  The first two iterations only bring the
    // pipeline up (faces of rows I0-1 and I0; nothing to update yet: `upd` false).  There is ONE instance of
    // the face arithmetic, so a zone's bits cannot depend on where chunk or patch boundaries fall (two
    // instances -- bring-up and steady state -- were tried: the compiler contracts them differently).
    //
    // Order of the vector-memory traffic.  Loads and stores share one in-order counter (vmcnt) on gfx9, and
    // hipcc, which assumes that a load and a pending store may complete in either order, waits for
    // vmcnt(0) at the first use of a loaded value whenever a store is outstanding.  With the stores issued
    // in the update block, a few instructions before the next row's first use, every iteration stalled for
    // the stores' round trip.  Here the row just updated is stored as the LAST thing of an iteration, after
    // the requested row has been put to its first use (its y-slopes, which is what makes the compiler
    // wait for it there), and the next requests follow as the first thing of the next iteration: whatever a
    // wait drains is then most of an iteration old.
    int refill = I0 + 1; // the iteration that next renews the row table
#pragma unroll 1
    for (int r = I0 + 1; r <= I1 + 1; ++r) {
        if (r == refill) { fill_rows(r); refill += ROW_TABLE_SPAN; }
        // (read here, with the global requests, a whole face block before their first use: the march has no other
        // wave's work to hide an LDS round trip behind)
        const double *ebp = tab[(r - 1) & (WAVE - 1)]; // row r-1: faces
        const double *eap = tab[(r - 2) & (WAVE - 1)]; // row r-2: the update
        const double eb[4] = {ebp[0], ebp[1], ebp[2], ebp[3]};
        const double ea[5] = {0.0, 0.0, eap[2], eap[3], eap[4]};
        const bool upd = r - 2 >= I0;
#if defined(SFHIP_DIAG_PRIO)  // diagnosis builds only: the wave's issue priority raised while it issues its memory requests
        __builtin_amdgcn_s_setprio(SFHIP_DIAG_PRIO);
#endif
        if (upd && want_un) {
            // u^n of the row updated in this iteration: in flight during both face blocks
            const double *us = unsrc + (long)(r - 2) * si + ucol;
#if defined(SFHIP_DIAG_COALESCED)
            {
                const long e0 = (long)(J - lane) * 3 + lane, last = si - 1;
                us -= ucol;
                un[0] = us[e0 < last ? e0 : last]; un[1] = us[e0 + 64 < last ? e0 + 64 : last];
                un[2] = us[e0 + 128 < last ? e0 + 128 : last];
            }
#else
            un[0] = us[0]; un[1] = us[1]; un[2] = us[2];
#endif
        }
        { // (the last iteration's request is never consumed: a clamped, valid row; saves a branch)
#if defined(SFHIP_DIAG_SAME_ROW)  // diagnosis builds only: every request hits the cache (what the loads' latency costs)
            const double *src = row(I0 + (r & 1));
#else
            const double *src = row(r + 1);
#endif
#if defined(SFHIP_DIAG_COALESCED)  // diagnosis builds only (WRONG data): the same bytes, lane-contiguous
            {   // (element index within the row, clamped to the row: the last row of the array has nothing behind it)
                const long e0 = (long)(J - lane) * 3 + lane, last = si - 1;
                src -= col;
                pn[0] = src[e0 < last ? e0 : last]; pn[1] = src[e0 + 64 < last ? e0 + 64 : last];
                pn[2] = src[e0 + 128 < last ? e0 + 128 : last];
            }
#else
            pn[0] = src[0]; pn[1] = src[1]; pn[2] = src[2];
#endif
        }
        __builtin_amdgcn_sched_barrier(0); // keep the requests first (the scheduler would sink them by ~150 instructions)
#if defined(SFHIP_DIAG_PRIO)
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#endif
#if defined(SFHIP_DIAG_NO_COMPUTE) && defined(SFHIP_DIAG_TWO_ROWS)  // the same traffic, two rows' requests in flight per wave
        {
            if ((r - I0) & 1) continue;
            const double *s2 = row(r + 2);
            double pq[3] = {s2[0], s2[1], s2[2]}, uq[3] = {0, 0, 0};
            if (want_un) { const double *us = unsrc + (long)(r - 1) * si + ucol; uq[0] = us[0]; uq[1] = us[1]; uq[2] = us[2]; }
            __builtin_amdgcn_sched_barrier(0);
            double o[3], o2[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                o[q] = pa[q] + un[q]; o2[q] = pb[q] + uq[q];
                pa[q] = pc[q]; pb[q] = pn[q]; pc[q] = pq[q];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (upd && store_lane) {
                const long c = (long)(r - 2) * si + (long)J * sj;
                pwr[c] = o[0]; pwr[c + 1] = o[1]; pwr[c + 2] = o[2];
                pwr[c + si] = o2[0]; pwr[c + si + 1] = o2[1]; pwr[c + si + 2] = o2[2];
            }
            continue;
        }
#elif defined(SFHIP_DIAG_NO_COMPUTE)  // diagnosis builds only (WRONG results): the march's memory traffic without its arithmetic
        {
            double o[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                o[q] = pa[q] + (want_un ? un[q] : 0.0);
                pa[q] = pb[q]; pb[q] = pc[q]; pc[q] = pn[q];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (upd && store_lane) {
#if defined(SFHIP_DIAG_COALESCED)
                const long e0 = (long)(J - lane) * 3 + lane, last = si - 1; // (clamped to the row, as the loads)
                double *w = pwr + (long)(r - 2) * si;
                w[e0 < last ? e0 : last] = o[0]; w[e0 + 64 < last ? e0 + 64 : last] = o[1];
                w[e0 + 128 < last ? e0 + 128 : last] = o[2];
#else
                const long c = (long)(r - 2) * si + (long)J * sj;
                pwr[c] = o[0]; pwr[c + 1] = o[1]; pwr[c + 2] = o[2];
#endif
            }
            continue;
        }
#endif
        // row r-1 now has both neighbours: x-slopes and strain (cbdiso_2d.c:248-263)
        double gxb[3], sxx, syy, sxy;
        iso_row_slopes(pa, pb, pc, gyb, visc, hidx, hidy, gxb, sxx, syy, sxy);


        // Both faces are evaluated in every iteration, so that the fluxes are plain locals and the
        // loop-carried `acc` is written unconditionally.  The first iteration's x-face (row I0-1, whose
        // lower-side states are not formed yet) and the last one's y-face (row I1) go unused: two face
        // evaluations per chunk.
        // x-face between rows r-2 and r-1 (cbdiso_2d.c:536-541, 560-561, 579-582)
        double F[3], tot[3], out[3];
        {
            double ppx[3], c2, cs;
#pragma unroll
            for (int q = 0; q < 3; ++q) ppx[q] = pb[q] - quarter * gxb[q];
            iso_sound_fast<FULL>(P, P.m, eb[0], eb[1], cyc0, cyc1, c2, cs); // at the lower x-face of row r-1
            iso_hlle<0>(pmx, ppx, c2, cs, F);
            if (visc) {
                F[1] -= hnu * (txx + pb[0] * sxx);
                F[2] -= hnu * (txy + pb[0] * sxy);
            }
        }
        // row r-2 is complete with its upper face: -(dF/dx + dG/dy) dt (cbdiso_2d.c:593-596)
#pragma unroll
        for (int q = 0; q < 3; ++q) tot[q] = __builtin_fma(-F[q], kx, acc[q]);
        if (r <= I1) { // (the row above the chunk's last one is somebody else's: its y-fluxes are not needed here)
            // y-face below this lane's zone in row r-1 (cbdiso_2d.c:543-546, 562-563, 583-586)
            double pmy[3], ppy[3], G[3], c2, cs;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                pmy[q] = from_lane_below(pb[q] + quarter * gyb[q]);
                ppy[q] = pb[q] - quarter * gyb[q];
            }
            iso_sound_fast<FULL>(P, P.m, eb[2], eb[3], cyf0, cyf1, c2, cs); // at the centre of row r-1 in x
            iso_hlle<1>(pmy, ppy, c2, cs, G);
            if (visc) {
                double tyx = pb[0] * sxy, tyy = pb[0] * syy;
                G[1] -= hnu * (from_lane_below(tyx) + tyx);
                G[2] -= hnu * (from_lane_below(tyy) + tyy);
            }
            // row r-1 so far: its lower x-face flux and its y-flux difference
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = __builtin_fma(-(from_lane_above(G[q]) - G[q]), ky, F[q] * kx);
        }
        if (upd) {
            // update row r-2 (cbdiso_2d.c:588-602)
            const double xc = ea[4];
            double u[3] = {pa[0], pa[1] * pa[0], pa[2] * pa[0]};
            double dc[3] = {0.0, 0.0, 0.0};
            iso_buffer<FULL>(P, xc, yc, yc2, u, dc);
            const double y0 = iso_mass_source(ea[2], P.m[0].mass, reach0, &Pk->m[0], dyc0, cyc0, P.dt, pa, dc);
            const double y1 = iso_mass_source(ea[3], P.m[1].mass, reach1, &Pk->m[1], dyc1, cyc1, P.dt, pa, dc);
#pragma unroll
            for (int q = 0; q < 3; ++q) u[q] += dc[q] + tot[q];
            if (want_un) {
                double un0 = un[0], un1 = un[1], un2 = un[2];
                if (P.un_mode == 2) { un1 *= un0; un2 *= un0; } // u^n = p2c(p^n), cbdiso_2d.c:285-298
                const double b = P.one_minus_a;
                u[0] = b * u[0] + P.a * un0;
                u[1] = b * u[1] + P.a * un1;
                u[2] = b * u[2] + P.a * un2;
            }
            iso_c2p(u, P.vmax, P.dfloor, out);
            const double rho = out[0], vx = out[1], vy = out[2];
            if (store_lane) {
                if (WS) {
                    // max(|v-cs|,|v+cs|) == fl(|v| + cs) for cs >= 0, so the zone's speed is
                    // fl(max(|vx|,|vy|) + cs).  A ~1-ulp estimate of cs filters the zones that
                    // cannot raise this lane's running maximum; the rest take the IEEE path,
                    // so the reduced value is bit-identical to cbdiso_2d_wavespeed + max.
                    // (`?:` as the reference's max2: the zone's speed is NaN when vy or cs is, not when only vx is)
                    double vm = max2(fabs(vx), fabs(vy));
                    double c2e, cse;
                    if (WS == 2) {
                        if (!FULL && !P.eos_local) cse = P.cs_uniform;
                        else {
                            c2e = 0.0;
                            if (FULL || P.m[0].mass > 0.0) c2e += P.m[0].mass * y0;
                            if (FULL || P.m[1].mass > 0.0) c2e += P.m[1].mass * y1;
                            c2e *= P.inv_mach2;
                            cse = fast_sqrt(c2e);
                        }
                    } else {
                        iso_sound_fast(P, P.mw, xc - P.mw[0].x, xc - P.mw[1].x, cyw0, cyw1, c2e, cse);
                    }
                    if (!((vm + cse) * (1.0 + 1e-9) < athr)) { // unordered passes: a NaN must reach the fold
                        double sp = iso_exact_speed(Pk, xc, yc, vx, vy);
                        amax = max_nan(amax, sp);
                        athr = max2(athr, amax);
                    }
                }
            }
        }
        // slide the window: row r-1 becomes the lower side of the next x-face, row r+1 enters
        iso_row_leaves(pb, gxb, sxx, sxy, quarter, pmx, txx, txy);
        enter(pn);
        __builtin_amdgcn_sched_barrier(0); // the requested row is in use above this line, stores go below
#if defined(SFHIP_DIAG_PRIO) && defined(SFHIP_DIAG_PRIO_STORES)
        __builtin_amdgcn_s_setprio(SFHIP_DIAG_PRIO);
#endif
#if defined(SFHIP_DIAG_STORE_EVERY)  // diagnosis builds only (WRONG results): the same bytes, stored every Nth iteration
        if (upd && store_lane && (r % SFHIP_DIAG_STORE_EVERY) == 0) {
            for (int k = 0; k < SFHIP_DIAG_STORE_EVERY; ++k) {
                const int rr = r - 2 - k < NG ? NG : r - 2 - k; // (never below the first interior row)
                const long c = (long)rr * si + (long)J * sj;
                pwr[c] = out[0]; pwr[c + 1] = out[1]; pwr[c + 2] = out[2];
            }
        }
        if (false) {
#elif defined(SFHIP_DIAG_NO_STORE)  // diagnosis builds only: what the stores cost
        if (upd && store_lane && out[0] < -1e300) {
#else
        if (upd && store_lane) {
#endif
#if defined(SFHIP_DIAG_PLANAR_STORE)  // diagnosis builds only (WRONG results): the same bytes, 512 contiguous bytes per instruction
            const long c = (long)(r - 2) * si + (long)(J - lane) * sj + lane;
            pwr[c] = out[0]; pwr[c + 64] = out[1]; pwr[c + 128] = out[2];
#else
#if defined(SFHIP_DIAG_SAME_ROW_STORE)  // diagnosis builds only (WRONG results): every store hits one of two cached rows
            const long c = (long)(I0 + (r & 1)) * si + (long)J * sj;
#else
            const long c = (long)(r - 2) * si + (long)J * sj;
#endif
            // (non-temporal stores were measured: -1.5 % on the a = 0 stage alone, nothing on the RK2 pair)
            pwr[c] = out[0]; pwr[c + 1] = out[1]; pwr[c + 2] = out[2];
#endif
        }
    }
    if (WS) {
#if defined(SFHIP_DIAG_NO_WS_ATOMIC)       // diagnosis builds only (tools/build_variant.sh): what the fold's pieces cost
        if (amax < 0.0) wmax[0] = amax;
#elif defined(SFHIP_DIAG_NO_WS_WAVEMAX)
        if (lane == 0) atomic_max_nonneg(wmax, amax);
#else
        amax = wave_max_nonneg(amax);
        if (lane == 0) atomic_max_nonneg(wmax, amax);
#endif
    }
}

// cbdiso_2d_primitive_to_conserved, cbdiso_2d.c:606-624 (interior zones only)
__global__ void iso_p2c_kernel(int ni, int nj, const double *__restrict__ prim, double *__restrict__ cons)
{
    long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    long i = blockIdx.y;
    if (j >= nj) return;
    long n = ((i + NG) * (nj + 2 * NG) + (j + NG)) * 3;
    double rho = prim[n], vx = prim[n + 1], vy = prim[n + 2];
    cons[n] = rho; cons[n + 1] = vx * rho; cons[n + 2] = vy * rho;
}

// cbdiso_2d_wavespeed, cbdiso_2d.c:666-722.  Writes the per-zone array when
// `out` is given (reference behaviour) and/or folds the max into *wmax.
__global__ void iso_wavespeed_kernel(IsoParams P, const double *__restrict__ prim,
                                     double *__restrict__ out, double *__restrict__ wmax)
{
    long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    double a = 0.0;
    if (j < P.nj) {
        const long NJ = P.nj + 2 * NG;
        const double y = P.yl + (j + 0.5) * P.dy;
        // four rows per trip, their loads issued together: with one row per trip the kernel was bound by the
        // latency of its single outstanding load per lane (0.39 ms at 4096^2 for 0.54 GB)
        constexpr int U = 4;
        for (long i0 = (long)blockIdx.y * U; i0 < P.ni; i0 += (long)gridDim.y * U) {
            double vx[U], vy[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const long i = i0 + k < P.ni ? i0 + k : P.ni - 1;
                const long z = (i + NG) * NJ + (j + NG);
                vx[k] = prim[z * 3 + 1]; vy[k] = prim[z * 3 + 2];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const long i = i0 + k;
                if (i < P.ni) {
                    const long z = (i + NG) * NJ + (j + NG);
                    double x = P.xl + (i + 0.5) * P.dx;
                    double c2 = iso_cs2(P, P.m, x, y);
                    double s = max_speed(vx[k], vy[k], c2);
                    if (out) out[z] = s;
                    a = max_nan(a, s);
                }
            }
        }
    }
    if (wmax) {
        a = wave_max_nonneg(a);
        if ((threadIdx.x & 63) == 0) atomic_max_nonneg(wmax, a);
    }
}

// cbdiso_2d_point_mass_source_term, cbdiso_2d.c:626-664 (dt = 1, one mass, accumulates)
__global__ void iso_mass_rate_kernel(IsoParams P, const double *__restrict__ prim, double *__restrict__ rate)
{
    long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    long i = blockIdx.y;
    if (j >= P.nj) return;
    long n = ((i + NG) * (P.nj + 2 * NG) + (j + NG)) * 3;
    double xc = P.xl + (i + 0.5) * P.dx;
    double yc = P.yl + (j + 0.5) * P.dy;
    double p[3] = {prim[n], prim[n + 1], prim[n + 2]};
    double dc[3] = {rate[n], rate[n + 1], rate[n + 2]};
    double ddy = yc - P.m[0].y;
    iso_mass_source(P.m[0], xc, ddy, ddy * ddy + P.m[0].rsoft * P.m[0].rsoft, 1.0, p, dc);
    rate[n] = dc[0]; rate[n + 1] = dc[1]; rate[n + 2] = dc[2];
}

} // namespace sf
