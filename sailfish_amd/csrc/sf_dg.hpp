// sf_dg.hpp -- cbdisodg_2d on gfx950: third-order modal discontinuous-Galerkin update of the
// isothermal binary-accretion problem (reference: /root/reference/sailfish/solvers/cbdisodg_2d.c;
// line numbers below refer to it).
//
// Work decomposition.  One lane owns one zone (lanes run along j, the contiguous direction);
// the zone's 18 live weights, the three 18-entry accumulators of the update (volume integral,
// surface integral, sources) and one neighbour's weights stay in VGPRs.  The loops over the 9
// volume nodes and the 12 face nodes are ROLLED: every per-node quantity that does not depend on
// the zone -- basis values, Gauss weights, node offsets -- sits in a __constant__ table indexed by
// the (wave-uniform) loop counter, so it arrives through scalar loads and is consumed as an SGPR
// operand.  Unrolled, the kernel would be ~100 KB of code against a 64 KB instruction cache.
//
// Arithmetic.  The scheme's slope limiter takes discrete decisions (troubled-cell threshold,
// "did minmod change the slope") on its inputs, so this file keeps the reference's operation
// order with IEEE division and square root and no FMA contraction: except for pow()/exp() in
// the point-mass source (device libm vs glibc, last-bit differences) every operation is the
// reference's, and kernels that do not touch those two functions (limiter, wavespeed, and the
// update wherever the masses are switched off) are bit-identical to it.  Three deliberate
// savings change at most the sign of an exact zero: terms whose basis coefficient is zero for
// EVERY node (x-derivatives of the m = 0 modes, y-derivatives of the n = 0 modes) are skipped,
// and the face-normal "gradient" evaluation, which in the reference equals the face value
// (basis_phi_1d returns the end-point value whatever `deriv` says, :383-390), reuses that value.
#pragma once
#include "sf_device.hpp"

namespace sf {

// No FMA contraction anywhere in this file: the library is compiled with -ffp-contract=off
// (__graft_entry__.HIPCC_FLAGS) and nothing here overrides it; the FMAs of dg_rcp / dg_div are explicit.

constexpr int DG_NW = 27;                         // doubles per zone (3 fields x 3 x 3)
constexpr int DG_K = 6;                           // used slots per field: m + n < 3
constexpr int DG_SLOT[DG_K] = {0, 1, 2, 3, 4, 6}; // m * 3 + n for (0,0) (0,1) (0,2) (1,0) (1,1) (2,0)
constexpr int DG_M[DG_K] = {0, 0, 0, 1, 1, 2};
constexpr int DG_N[DG_K] = {0, 1, 2, 0, 1, 0};

struct DgMass {
    double x, y, vx, vy, mass, rsoft, sink_rate, sink_radius;
    int sink_model;
    int pad_;
};

struct DgParams {
    int ni, nj;
    double xl, yl, dx, dy, cell_volume;
    int buffer_on, eos;
    double buf_sd, buf_mass, buf_onset, buf_rate_prefix, buf_span; // prefix = driving_rate * omega_outer (:214-215)
    DgMass m[2];
    double cs2, mach2;
    double nu, rk, dt, vmax;
    int clamp_lo, clamp_hi, clamp_j, i_begin, i_end, want_ws;
    int rows_per_wave;          // rows a wave of dg_advance_kernel marches up
    double wx[2], wy[2], wm[2], wrs[2]; // point masses of the wavespeed epilogue
    const double *dt_ptr;
};

// ---------------------------------------------------------------------------------------------
// Basis tables (:363-409), built at compile time with the reference's literals; a product of two
// literals is rounded once, at compile time or at run time alike.
// ---------------------------------------------------------------------------------------------
constexpr double dg_basis(int at, int m, int deriv) // at: 0..2 Gauss node, 3 left end, 4 right end
{
    constexpr double node[3][3] = {
        {+1.000000000000000, +1.00000000000000, +1.000000000000000},
        {-1.341640786499873, +0.00000000000000, +1.341640786499873},
        {+0.894427190999914, -1.11803398874990, +0.894427190999914},
    };
    constexpr double der[3][3] = {
        {+0.000000000000000, +0.000000000000000, +0.000000000000000},
        {+1.732050807568877, +1.732050807568877, +1.732050807568877},
        {-5.196152422706629, +0.000000000000000, +5.196152422706629},
    };
    constexpr double left[3] = {+1.000000000000000, -1.732050807568877, +2.23606797749979};
    constexpr double right[3] = {+1.000000000000000, +1.732050807568877, +2.23606797749979};
    if (at == 3) return left[m];
    if (at == 4) return right[m];
    return deriv ? der[m][at] : node[m][at];
}

struct DgTables {
    // volume nodes, index a * 3 + b
    double vol_phi[9][DG_K]; // phi_volume
    double vol_dpx[9][DG_K]; // phi_gradient_x
    double vol_dpy[9][DG_K]; // phi_gradient_y
    double vol_cx[9], vol_cy[9], vol_gw[9];
    // face nodes, index quad * 4 + face (face: 0 xl, 1 xr, 2 yl, 3 yr); "own" is the zone's own
    // trace on that face, "nb" the neighbour's
    double face_own[12][DG_K]; // value, which is also the test function of the surface integral
    double face_nb[12][DG_K];
    double face_own_t[12][DG_K]; // tangential derivative
    double face_nb_t[12][DG_K];
    double face_cx[12], face_cy[12], face_gw[12];
};

constexpr DgTables dg_make_tables()
{
    constexpr double gw[3] = {0.555555555555556, 0.888888888888889, 0.555555555555556};  // :648
    constexpr double gx[3] = {-0.774596669241483, 0.000000000000000, 0.774596669241483}; // :649
    DgTables t = {};
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) {
            const int nd = a * 3 + b;
            for (int k = 0; k < DG_K; ++k) {
                t.vol_phi[nd][k] = dg_basis(a, DG_M[k], 0) * dg_basis(b, DG_N[k], 0);
                t.vol_dpx[nd][k] = dg_basis(a, DG_M[k], 1) * dg_basis(b, DG_N[k], 0);
                t.vol_dpy[nd][k] = dg_basis(a, DG_M[k], 0) * dg_basis(b, DG_N[k], 1);
            }
            t.vol_cx[nd] = 0.5 * gx[a];
            t.vol_cy[nd] = 0.5 * gx[b];
            t.vol_gw[nd] = gw[a] * gw[b];
        }
    }
    for (int q = 0; q < 3; ++q) {
        for (int f = 0; f < 4; ++f) {
            const int fn = q * 4 + f, dir = f >> 1, high = f & 1;
            const int own_end = high ? 4 : 3, nb_end = high ? 3 : 4;
            for (int k = 0; k < DG_K; ++k) {
                const int m = DG_M[k], n = DG_N[k];
                if (dir == 0) {
                    t.face_own[fn][k] = dg_basis(own_end, m, 0) * dg_basis(q, n, 0);
                    t.face_nb[fn][k] = dg_basis(nb_end, m, 0) * dg_basis(q, n, 0);
                    t.face_own_t[fn][k] = dg_basis(own_end, m, 0) * dg_basis(q, n, 1);
                    t.face_nb_t[fn][k] = dg_basis(nb_end, m, 0) * dg_basis(q, n, 1);
                } else {
                    t.face_own[fn][k] = dg_basis(q, m, 0) * dg_basis(own_end, n, 0);
                    t.face_nb[fn][k] = dg_basis(q, m, 0) * dg_basis(nb_end, n, 0);
                    t.face_own_t[fn][k] = dg_basis(q, m, 1) * dg_basis(own_end, n, 0);
                    t.face_nb_t[fn][k] = dg_basis(q, m, 1) * dg_basis(nb_end, n, 0);
                }
            }
            t.face_cx[fn] = dir == 0 ? (high ? 0.5 : -0.5) : 0.5 * gx[q];
            t.face_cy[fn] = dir == 1 ? (high ? 0.5 : -0.5) : 0.5 * gx[q];
            t.face_gw[fn] = gw[q];
        }
    }
    return t;
}

__constant__ DgTables dg_tab = dg_make_tables();

// Division: the ~700 divisions per zone use a handful of denominators per node (the surface density of
// a state, dx, dy, a+ - a- ...), so they go through exact_rcp / exact_div (sf_device.hpp): the reciprocal
// refinement is done once per denominator, each quotient costs three instructions, and the result is
// bit-identical to `a / b`.
using DgRcp = ExactRcp;
__device__ __forceinline__ DgRcp dg_rcp(double den) { return exact_rcp(den); }
__device__ __forceinline__ double dg_div(double num, const DgRcp &d) { return exact_div(num, d); }

// ---------------------------------------------------------------------------------------------
// Physics pieces, each the reference's expression tree
// ---------------------------------------------------------------------------------------------

// gravitational_potential + sound_speed_squared, :56-81, :171-188; masses given as 4 arrays
// (LOCISO: locally isothermal with both masses positive, established on the host)
template <bool LOCISO = false>
__device__ __forceinline__ double dg_cs2(double cs2, const DgRcp &mach2, int eos, double x, double y,
                                         const double (&mx)[2], const double (&my)[2], const double (&mm)[2],
                                         const double (&mrs)[2])
{
    if (!LOCISO && eos == 1) return cs2;
    if (!LOCISO && eos != 2) return 1.0;
    double phi = 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (LOCISO || mm[p] > 0.0) {
            double ddx = x - mx[p];
            double ddy = y - my[p];
            double r2 = ddx * ddx + ddy * ddy;
            phi -= exact_quot(mm[p], exact_sqrt0(r2 + mrs[p] * mrs[p]));
        }
    }
    return dg_div(-phi, mach2);
}

// conserved_to_primitive, :227-241
__device__ __forceinline__ void dg_c2p(const double (&u)[3], const DgRcp &ru0, double (&p)[3], double vmax)
{
    p[0] = u[0];
    p[1] = copysign(1.0, u[1]) * min2(fabs(dg_div(u[1], ru0)), vmax);
    p[2] = copysign(1.0, u[2]) * min2(fabs(dg_div(u[2], ru0)), vmax);
}

// primitive_to_flux, :255-269, with the direction as two 0/1 factors
__device__ __forceinline__ void dg_flux(const double (&p)[3], const double (&u)[3], double cs2, double vn, double e0,
                                        double e1, double (&f)[3])
{
    double pressure = p[0] * cs2;
    f[0] = vn * u[0];
    f[1] = vn * u[1] + pressure * e0;
    f[2] = vn * u[2] + pressure * e1;
}

// The four velocity gradients and three stresses of add_viscous_flux, :326-349 (mu included)
__device__ __forceinline__ void dg_stress(const double (&u)[3], const DgRcp &r0, const double (&ux)[3],
                                          const double (&uy)[3], double nu, const DgRcp &dx, const DgRcp &dy,
                                          double &txx, double &txy, double &tyy)
{
    double dvxdx = dg_div(2.0 * (dg_div(ux[1], r0) - dg_div(dg_div(u[1] * ux[0], r0), r0)), dx);
    double dvydx = dg_div(2.0 * (dg_div(ux[2], r0) - dg_div(dg_div(u[2] * ux[0], r0), r0)), dx);
    double dvxdy = dg_div(2.0 * (dg_div(uy[1], r0) - dg_div(dg_div(u[1] * uy[0], r0), r0)), dy);
    double dvydy = dg_div(2.0 * (dg_div(uy[2], r0) - dg_div(dg_div(u[2] * uy[0], r0), r0)), dy);
    double mu = u[0] * nu;
    double sxx = 4.0 / 3.0 * dvxdx - 2.0 / 3.0 * dvydy;
    double syy = 4.0 / 3.0 * dvydy - 2.0 / 3.0 * dvxdx;
    double sxy = dvxdy + dvydx;
    txx = mu * sxx;
    txy = mu * sxy;
    tyy = mu * syy;
}

// point_mass_source_term, :83-152.  pow(x, -1.5) and exp() are the two operations of this file
// that are not the reference's bit for bit.
__device__ __forceinline__ void dg_mass_source(const DgMass &ms, double x, double y, double dt, const double (&prim)[3],
                                               double (&du)[3])
{
    double sigma = prim[0];
    double ddx = x - ms.x;
    double ddy = y - ms.y;
    double r2 = ddx * ddx + ddy * ddy;
    double dr = exact_sqrt0(r2);
    double rs2 = r2 + ms.rsoft * ms.rsoft;
    double fnum = sigma * ms.mass * exact_quot(1.0, rs2 * exact_sqrt0(rs2));
    double fx = -fnum * ddx;
    double fy = -fnum * ddy;
    double rate = 0.0;
    if (dr < 4.0 * ms.sink_radius) {
        double s = dr / ms.sink_radius;
        double s2 = s * s;
        rate = ms.sink_rate * exp(-(s2 * s2));
    }
    double mdot = sigma * rate * -1.0;
    du[0] += 0.0;
    du[1] += fx * dt;
    du[2] += fy * dt;
    switch (ms.sink_model) {
    case 1:
        du[0] += dt * mdot;
        du[1] += dt * mdot * prim[1];
        du[2] += dt * mdot * prim[2];
        break;
    case 2: {
        const DgRcp rdr = dg_rcp(dr + 1e-12);
        double rhx = dg_div(ddx, rdr);
        double rhy = dg_div(ddy, rdr);
        double dv = (prim[1] - ms.vx) * rhx + (prim[2] - ms.vy) * rhy;
        double vxs = dv * rhx + ms.vx;
        double vys = dv * rhy + ms.vy;
        du[0] += dt * mdot;
        du[1] += dt * mdot * vxs;
        du[2] += dt * mdot * vys;
        break;
    }
    case 3:
        du[0] += dt * mdot;
        du[1] += 0.0;
        du[2] += 0.0;
        break;
    default:
        du[0] += 0.0;
        du[1] += 0.0;
        du[2] += 0.0;
        break;
    }
}

// Both point masses of a node: the same operations in the same order on du as two calls of dg_mass_source, with the
// two gravity terms (a square root, a division and a square root again each) formed side by side ahead of the sink
// tests, which are divergent and end a scheduling region each.
__device__ __forceinline__ void dg_two_mass_sources(const DgMass (&ms)[2], double x, double y, double dt,
                                                    const double (&prim)[3], double (&du)[3])
{
    const double sigma = prim[0];
    double ddx[2], ddy[2], r2[2], fx[2], fy[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ddx[k] = x - ms[k].x;
        ddy[k] = y - ms[k].y;
        r2[k] = ddx[k] * ddx[k] + ddy[k] * ddy[k];
        double rs2 = r2[k] + ms[k].rsoft * ms[k].rsoft;
        double fnum = sigma * ms[k].mass * exact_quot(1.0, rs2 * exact_sqrt0(rs2));
        fx[k] = -fnum * ddx[k];
        fy[k] = -fnum * ddy[k];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        // The distance itself is needed inside the sink (dr < 4 r_sink) and by sink model 2 only: its square root is
        // taken there, behind a test on the squared distance that lets through a little more than the sink (the
        // exact test decides inside), instead of at every node of every zone.
        const double rs4 = 4.0 * ms[k].sink_radius;
        double rate = 0.0, dr_k = 0.0;
        if (ms[k].sink_model == 2 || r2[k] < rs4 * rs4 * 1.000001) {
            dr_k = exact_sqrt0(r2[k]);
            if (dr_k < rs4) {
                double s = dr_k / ms[k].sink_radius;
                double s2 = s * s;
                rate = ms[k].sink_rate * exp(-(s2 * s2));
            }
        }
        double mdot = sigma * rate * -1.0;
        du[0] += 0.0;
        du[1] += fx[k] * dt;
        du[2] += fy[k] * dt;
        switch (ms[k].sink_model) {
        case 1:
            du[0] += dt * mdot;
            du[1] += dt * mdot * prim[1];
            du[2] += dt * mdot * prim[2];
            break;
        case 2: {
            const DgRcp rdr = dg_rcp(dr_k + 1e-12);
            double rhx = dg_div(ddx[k], rdr);
            double rhy = dg_div(ddy[k], rdr);
            double dv = (prim[1] - ms[k].vx) * rhx + (prim[2] - ms[k].vy) * rhy;
            double vxs = dv * rhx + ms[k].vx;
            double vys = dv * rhy + ms[k].vy;
            du[0] += dt * mdot;
            du[1] += dt * mdot * vxs;
            du[2] += dt * mdot * vys;
            break;
        }
        case 3:
            du[0] += dt * mdot;
            du[1] += 0.0;
            du[2] += 0.0;
            break;
        default:
            du[0] += 0.0;
            du[1] += 0.0;
            du[2] += 0.0;
            break;
        }
    }
}

// buffer_source_term, :190-223
__device__ __forceinline__ void dg_buffer_source(const DgParams &P, double x, double y, double dt, const double (&u)[3],
                                                 double (&du)[3])
{
    if (!P.buffer_on) return;
    double rc = exact_sqrt0(x * x + y * y);
    if (rc > P.buf_onset) {
        const DgRcp rrc = dg_rcp(rc);
        double vk = exact_sqrt0(dg_div(P.buf_mass, rrc));
        double u0[3] = {P.buf_sd, P.buf_sd * dg_div(-y, rrc) * vk, P.buf_sd * dg_div(+x, rrc) * vk};
        double rate = exact_quot(P.buf_rate_prefix * (rc - P.buf_onset), P.buf_span);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            du[q] -= (u[q] - u0[q]) * rate * dt;
    }
}

// sum_k w[q][k] * c[k] from 0.0, in slot order (reconstruct_2d / reconstruct_1d, :411-453);
// slots whose bit is clear in MASK have a zero coefficient at every node and are skipped
template <int MASK>
__device__ __forceinline__ void dg_eval(const double (&w)[3][DG_K], const double *c, double (&out)[3])
{
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < DG_K; ++k)
            if (MASK & (1 << k)) s += w[q][k] * c[k];
        out[q] = s;
    }
}

__device__ __forceinline__ void dg_load(const double *zone, double (&w)[3][DG_K])
{
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < DG_K; ++k)
            w[q][k] = zone[q * 9 + DG_SLOT[k]];
}

// ---------------------------------------------------------------------------------------------
// cbdisodg_2d_advance_rk, :607-983 (+ index clamps, device dt and wavespeed epilogue; sfhip.h)
//
// Every face once (almost).  The reference's zone evaluates the Riemann problems of its four faces itself, so each
// face is solved twice, by the zones on either side of it -- with the same operands in the same order: both see
// the low-side zone's trace as `um` and the high-side zone's as `up`, both subtract the high-side zone's viscous
// stress first (:860-869).  The ONLY thing that can differ between the two evaluations is the position the sound
// speed is taken at: the face node as xc + 0.5 dx of the zone below or as xc - 0.5 dx of the zone above, which agree
// to rounding, not always bit for bit.  Here a wave owns 63 columns (lane 63 owns no zone: it provides lane 62's
// high y-face) and marches up `rows_per_wave` rows.  For a row it goes through the twelve face nodes in the
// reference's order -- per Gauss node: xl, xr, yl, yr -- adding to the surface integral as it goes, but SOLVES only
//   xr  between this row and the next: the fluxes as this row sees them, and as the next row will see them as its
//       xl (kept in LDS for the next iteration; the sound speed at the next row's position -- when both positions
//       are the same doubles in every lane, always the case for a globally isothermal gas and for dyadic mesh
//       spacings, the second solve is skipped);
//   yl  between the lane to the left and this lane: as this lane sees them, and as the left lane sees them as its
//       yr, which that lane fetches with one lane shift per component;
//   xl  only in the first row of the march (the row below belongs to another wave, which solves that face again).
// 12 -> 6 + 3 / rows_per_wave face nodes per zone; every wave is its own workgroup, no barrier.
// Memory: the wave reads a row ONCE, with unit-stride sweeps, into 14 KB of LDS one iteration before the row is its
// own (the zone above the three xr nodes and the row's own weights come from there); the left zone's trace at a yl
// node comes from the lane below (lane 0 forms it from the staged copy of the zone left of the wave).  19 KB of LDS
// per wave, 8 waves per CU.  The parameters of a row's two phases (face loop, volume loop + update) are read from
// the kernel-argument block at the start of each phase (DESIGN.md section 9 says why).
// (A tile of 4 rows x 63 columns with one wave per row and the x-faces exchanged through LDS behind a barrier was
// built first: 6.75 face nodes per zone, 24 % fewer instructions, but 10 % less time only -- the barrier and the
// waves that solve the tile's upper edge cost 6 % and 8 %.)
// Results are those of the zone kernel bit for bit (tests/test_hip_dg.py).
// grid: (ceil(nj / 63), ceil((i_end - i_begin) / rows_per_wave)), block 64
// ---------------------------------------------------------------------------------------------

// The trace of a zone's weights at face-table row fn (and, for the viscous stress, its tangential derivative there).
__device__ __forceinline__ void dg_traces(const double (&w)[3][DG_K], const double (*tab)[DG_K], const double (*tan)[DG_K],
                                          int fn, bool viscous, double (&u)[3], double (&t)[3])
{
    dg_eval<63>(w, tab[fn], u);
    if (viscous) dg_eval<63>(w, tan[fn], t);
}

// One face node between two zones: `a` (this wave's row / this lane, its weights) and `b` (the zone on the other side,
// given by its trace ub and tangential derivative tb at the node: the caller forms them from b's weights, or takes
// them from the lane that holds b).  `swap` false: `a` is the zone ABOVE the face (the face is a's low face `fn`, b
// is seen through its high-end basis dg_tab.face_nb); true: `a` is the zone BELOW it (the face is b's low face `fn`,
// b is seen through dg_tab.face_own).  fa: the fluxes with the sound speed at (xa, ya), a's own position of the
// node; fb: with it at (xb, yb), b's -- solved only if `both` and the positions differ.
template <bool LOCISO>
__device__ __forceinline__ void dg_face_node(const DgParams &P, const double (&a)[3][DG_K], const double (&ub)[3],
                                             const double (&tb)[3], int fn,
                                             bool ydir, bool swap, bool both, double xa, double ya, double xb, double yb,
                                             const DgRcp &rmach2, const DgRcp &rdx, const DgRcp &rdy,
                                             const double (&mx)[2], const double (&my)[2], const double (&mm)[2],
                                             const double (&mrs)[2], bool viscous, double (&fa)[3], double (&fb)[3])
{
    const double cs2 = dg_cs2<LOCISO>(P.cs2, rmach2, P.eos, xa, ya, mx, my, mm, mrs);
    // a's traces: the zone above the face through its low-end basis, the zone below through its high-end basis
    double ua[3], ta[3];
    dg_traces(a, swap ? dg_tab.face_nb : dg_tab.face_own, swap ? dg_tab.face_nb_t : dg_tab.face_own_t, fn, viscous, ua, ta);
    const DgRcp ra = dg_rcp(ua[0]), rb = dg_rcp(ub[0]);
    double up[3], um[3]; // high-side and low-side traces
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        up[c] = swap ? ub[c] : ua[c];
        um[c] = swap ? ua[c] : ub[c];
    }
    const DgRcp rp = {up[0], swap ? rb.r : ra.r}, rm = {um[0], swap ? ra.r : rb.r};
    double pl[3], pr[3];
    dg_c2p(um, rm, pl, P.vmax);
    dg_c2p(up, rp, pr, P.vmax);
    const double e0 = ydir ? 0.0 : 1.0, e1 = ydir ? 1.0 : 0.0;
    const double vl = ydir ? pl[2] : pl[1], vr = ydir ? pr[2] : pr[1];
    double h1 = 0.0, h2 = 0.0, l1 = 0.0, l2 = 0.0; // viscous stresses of the high-side and the low-side zone
    if (viscous) {
        // the face-normal "gradient" IS the trace
        double gxa[3], gya[3], gxb[3], gyb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gxa[c] = ydir ? ta[c] : ua[c];
            gya[c] = ydir ? ua[c] : ta[c];
            gxb[c] = ydir ? tb[c] : ub[c];
            gyb[c] = ydir ? ub[c] : tb[c];
        }
        double axx, axy, ayy, bxx, bxy, byy;
        dg_stress(ua, ra, gxa, gya, P.nu, rdx, rdy, axx, axy, ayy);
        dg_stress(ub, rb, gxb, gyb, P.nu, rdx, rdy, bxx, bxy, byy);
        const double a1 = ydir ? axy : axx, a2 = ydir ? ayy : axy;
        const double b1 = ydir ? bxy : bxx, b2 = ydir ? byy : bxy;
        h1 = swap ? b1 : a1; h2 = swap ? b2 : a2;
        l1 = swap ? a1 : b1; l2 = swap ? a2 : b2;
    }
    // riemann_hlle, :295-324, and the viscous fluxes: high side first, then low side (:860-869)
    auto solve = [&](double c2, double (&fhat)[3]) {
        double fl[3], fr[3];
        dg_flux(pl, um, c2, vl, e0, e1, fl);
        dg_flux(pr, up, c2, vr, e0, e1, fr);
        const double cs = exact_sqrt0(c2);
        const double am = min2(0.0, min2(vl - cs, vr - cs));
        const double ap = max2(0.0, max2(vl + cs, vr + cs));
        const DgRcp rspan = dg_rcp(ap - am);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            fhat[c] = dg_div(fl[c] * ap - fr[c] * am - (um[c] - up[c]) * ap * am, rspan);
        if (viscous) {
            fhat[1] -= h1;
            fhat[2] -= h2;
            fhat[1] -= l1;
            fhat[2] -= l2;
        }
    };
    solve(cs2, fa);
    bool differs = false;
#if !defined(SFHIP_DIAG_DG_ONE_VIEW)  // diagnosis builds only (WRONG in the last place): checks that the tests see the second view
    if (both && (LOCISO || P.eos == 2)) differs = xb != xa || yb != ya;
#endif
    if (__builtin_amdgcn_ballot_w64(differs) != 0) { // (wave-uniform)
        const double cs2b = dg_cs2<LOCISO>(P.cs2, rmach2, P.eos, xb, yb, mx, my, mm, mrs);
        solve(cs2b, fb);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) fb[c] = fa[c];
    }
}

template <int F> struct DgFace { static constexpr int value = F; };

// VISC: nu > 0; LOCISO: locally isothermal gas, both masses positive -- established on the host (the uniform tests on
// them, in every node of both loops, drop out).
template <bool VISC, bool LOCISO>
__global__ __launch_bounds__(64, 2) void dg_advance_kernel(DgParams P, const double *__restrict__ weights0,
                                                        const double *__restrict__ weights1,
                                                        double *__restrict__ weights2, double *wavespeed_max)
{
    // the fluxes through the next row's low x-face as that row sees them: [node][field][lane]
    __shared__ double next_xl[3][3][64];
    // the weights of the wave's 64 zones in the NEXT row and of the zone to their left, as they lie in memory (27
    // doubles per zone, 14 KB: 27 unit-stride sweeps of the wave and a part of one).  The three x-faces above a row
    // read the zone above from here, and the next iteration takes it from here as its own.  (Read from memory where
    // they are used, those were four 216-byte-strided reads of the same bytes, 15-50 us apart -- longer than a line
    // survives in L2 at this rate: the launch fetched 3.7 x its algorithmic bytes, profiles/r3_dg2048_pmc.txt.)
    __shared__ double row_up[65 * DG_NW];
    // lane 0's left neighbour in the CURRENT row (another wave's zone), kept back from row_up before it is restaged
    __shared__ double left_zone[DG_NW];
    const int lane0 = threadIdx.x;
    const int j0 = blockIdx.x * 63;
    const int i_first = P.i_begin + blockIdx.y * P.rows_per_wave;
    const int i_stop = i_first + P.rows_per_wave < P.i_end ? i_first + P.rows_per_wave : P.i_end;
    const long si = (long)DG_NW * (P.nj + 2);
    const double dx = P.dx, dy = P.dy;
    const bool viscous = VISC;
    const DgRcp rdx = dg_rcp(dx), rdy = dg_rcp(dy), rmach2 = dg_rcp(P.mach2), rvol = dg_rcp(P.cell_volume);
    const double dt = P.dt_ptr ? *P.dt_ptr : P.dt;
    // the zone left of lane 0's: slot 0 of the staged row (under an outflow clamp at the row's start: its own zone)
    const int left_slot = (P.clamp_j && j0 == 0) ? 1 : 0;
    // the staged segment: memory columns j0 ... j0 + 64 of a row (zones j0 - 1 ... j0 + 63), as far as the row, its
    // high guard column included, goes; addresses past it are clamped to its last double
    const int nstage = (P.nj + 2 - j0 < 65 ? P.nj + 2 - j0 : 65) * DG_NW;
    auto stage_row = [&](int row, int lane_) {
        const double *seg = weights1 + (long)(row + 1) * si + (long)j0 * DG_NW;
        double staged[DG_NW], tail = 0.0;
        if (nstage == 65 * DG_NW) { // (wave-uniform: every wave but the last of a row)
#pragma unroll
            for (int r = 0; r < DG_NW; ++r) staged[r] = (seg + 64 * r)[lane_]; // (scalar base + lane: no address registers)
            if (lane_ < DG_NW) tail = (seg + 64 * DG_NW)[lane_];
        } else {
            // (the bound is made opaque here: as a loop invariant the 27 clamped indices are formed once, ahead of the
            // row loop, and live in scratch memory from then on)
            unsigned last = nstage - 1, first = lane_;
            asm volatile("" : "+v"(last), "+v"(first));
#pragma unroll
            for (int r = 0; r < DG_NW; ++r) {
                const unsigned k = first + 64 * r;
                staged[r] = seg[k < last ? k : last];
            }
        }
#pragma unroll
        for (int r = 0; r < DG_NW; ++r) row_up[lane_ + 64 * r] = staged[r];
        if (lane_ < DG_NW) row_up[lane_ + 64 * DG_NW] = tail; // (zone j0 + 63: lane 63's, which is needed only when full)
    };
    double speed = 0.0;

    if (i_first < i_stop) stage_row(i_first, lane0);
#pragma unroll 1
    for (int i = i_first; i < i_stop; ++i) {
        // What depends on the lane alone is formed again in every row, from a copy of the lane index that the
        // compiler cannot see through: as loop invariants those values (and everything the node loops derive from
        // them) are formed once ahead of the row loop, and at 2 waves per SIMD they then live in scratch memory.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int j = j0 + lane;
        const bool present = j <= P.nj;            // solves its low y-face
        const bool owner = lane < 63 && j < P.nj;  // updates zones
        const double yc = P.yl + (j + 0.5) * dy;
        const double yc_left = P.yl + ((j - 1) + 0.5) * dy; // the left lane's own centre, formed as that lane forms it
        // an outflow edge's guard zone is a copy of the edge zone itself: the lane beyond the edge (it only provides
        // the edge zone's high y-face) then IS the edge zone.  (Lanes beyond the row's last column, in the last wave
        // of a row only, go through the face nodes of that column: nothing of theirs is used.)
        const int ja = (P.clamp_j && j == P.nj) ? j - 1 : (present ? j : P.nj);
        const unsigned zoff = (ja + 1) * DG_NW;                 // this lane's zone within a row of the arrays
        const double *mine_up = row_up + (ja - j0 + 1) * DG_NW; // ... and within the staged row
        double cc[3][DG_K];
        dg_load(mine_up, cc);
        if (lane < DG_NW) left_zone[lane] = row_up[left_slot * DG_NW + lane];
        // the next row (above the domain's last row under an outflow clamp: the row itself).  One wave, and LDS
        // accesses of a wave complete in program order: the reads above come before these writes.
        stage_row((P.clamp_hi && i == P.ni - 1) ? i : i + 1, lane);
        {
            const double xc = P.xl + (i + 0.5) * dx;
            const double xc_up = P.xl + ((i + 1) + 0.5) * dx; // the next row's own centre, formed as that row forms it
            // the row below is read from memory (in the first row of a march only)
            const double *row_below = weights1 + (long)((P.clamp_lo && i == 0) ? i + 1 : i) * si;

            // ---- surface integral, :848-961: per Gauss node the faces xl, xr, yl, yr ----------
            // The parameters of each phase of a row are read from the kernel's argument block at the start of that
            // phase, through a pointer the compiler cannot see through: as values of the whole kernel the ~100 of
            // them would all occupy scalar registers across both node loops, and those that do not fit are kept in
            // lanes of a vector register (one v_readlane per use: 13 % of the loops' instructions).
            const __attribute__((address_space(4))) DgParams *pf =
                (const __attribute__((address_space(4))) DgParams *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(pf));
            const DgParams &Pf = *(const DgParams *)pf;
            double srf[3][DG_K]; // equation_20
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int k = 0; k < DG_K; ++k) srf[q][k] = 0.0;
            double yr[3] = {0.0, 0.0, 0.0};
            {
            const double mx[2] = {Pf.m[0].x, Pf.m[1].x}, my[2] = {Pf.m[0].y, Pf.m[1].y};
            const double mm[2] = {Pf.m[0].mass, Pf.m[1].mass}, mrs[2] = {Pf.m[0].rsoft, Pf.m[1].rsoft};
#pragma unroll 1
            for (int q = 0; q < 3; ++q) {
            // (the four faces of a Gauss node as straight-line code: the face index is a compile-time constant in each)
            auto face_step = [&](auto face_tag) __attribute__((always_inline)) {
                constexpr int face = decltype(face_tag)::value;
                const int fn = q * 4 + face;
                constexpr bool high = face & 1, ydir = face >> 1;
                double fhat[3];
                if (face == 3) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) fhat[c] = yr[c];
                } else if (face == 0 && i != i_first) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) fhat[c] = next_xl[q][c][lane];
                } else {
                    const double x = xc + dg_tab.face_cx[fn] * dx; // the node as this zone places it
                    const double y = yc + dg_tab.face_cy[fn] * dy;
                    double fother[3], nb[3][DG_K], ub[3], tb[3];
                    if (face == 1) {
                        // the zone above, from the staged row (through an offset the compiler cannot see through:
                        // hoisted out of the node loop, its 18 doubles would stay in registers across it)
                        int up_at = (ja - j0 + 1) * DG_NW;
                        asm volatile("" : "+v"(up_at));
                        dg_load(row_up + up_at, nb);
                        // as the next row's low x-face: its table row, its position of the node
                        const int fu = q * 4;
                        const double xu = xc_up + dg_tab.face_cx[fu] * dx;
                        const double yu = yc + dg_tab.face_cy[fu] * dy;
                        dg_traces(nb, dg_tab.face_own, dg_tab.face_own_t, fu, viscous, ub, tb);
                        dg_face_node<LOCISO>(Pf, cc, ub, tb, fu, false, true, i + 1 < i_stop, x, y, xu, yu, rmach2, rdx, rdy,
                                     mx, my, mm, mrs, viscous, fhat, fother);
#pragma unroll
                        for (int c = 0; c < 3; ++c) next_xl[q][c][lane] = fother[c];
                    } else if (face == 2) {
                        // the left lane sees this face as its high y-face fn + 1
                        const double xl_ = xc + dg_tab.face_cx[fn + 1] * dx;
                        const double yl_ = yc_left + dg_tab.face_cy[fn + 1] * dy;
                        // the zone to the left is the lane below's: every lane forms ITS zone's high-end trace (the
                        // same operations on the same weights as forming it here from that zone's weights) and
                        // hands it up one lane; lane 0's left zone is another wave's, read from memory
                        double uh[3], th[3] = {0.0, 0.0, 0.0};
                        dg_traces(cc, dg_tab.face_nb, dg_tab.face_nb_t, fn, viscous, uh, th);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            ub[c] = from_lane_below(uh[c]);
                            tb[c] = from_lane_below(th[c]);
                        }
                        if (lane == 0) {
                            int left_at = 0;
                            asm volatile("" : "+v"(left_at));
                            dg_load(left_zone + left_at, nb);
                            dg_traces(nb, dg_tab.face_nb, dg_tab.face_nb_t, fn, viscous, ub, tb);
                        }
                        dg_face_node<LOCISO>(Pf, cc, ub, tb, fn, true, false, true, x, y, xl_, yl_, rmach2, rdx, rdy, mx, my, mm,
                                     mrs, viscous, fhat, fother);
#pragma unroll
                        for (int c = 0; c < 3; ++c) yr[c] = from_lane_above(fother[c]); // the high y-face of THIS lane's zone
                    } else {
                        dg_load(row_below + zoff, nb);
                        dg_traces(nb, dg_tab.face_nb, dg_tab.face_nb_t, fn, viscous, ub, tb);
                        dg_face_node<LOCISO>(Pf, cc, ub, tb, fn, false, false, false, x, y, x, y, rmach2, rdx, rdy, mx, my, mm,
                                     mrs, viscous, fhat, fother);
                    }
                }
                const double len = ydir ? dx : dy;
                const double gw = dg_tab.face_gw[fn];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int k = 0; k < DG_K; ++k) {
                        double term = len * fhat[c] * dg_tab.face_own[fn][k] * gw;
                        srf[c][k] += high ? term : -term;
                    }
                }
            };
            face_step(DgFace<0>{});
            face_step(DgFace<1>{});
            face_step(DgFace<2>{});
            face_step(DgFace<3>{});
            }
            }
            if (!owner) continue;

            const __attribute__((address_space(4))) DgParams *pv =
                (const __attribute__((address_space(4))) DgParams *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(pv));
            const DgParams &Pv = *(const DgParams *)pv;
            const double mx[2] = {Pv.m[0].x, Pv.m[1].x}, my[2] = {Pv.m[0].y, Pv.m[1].y};
            const double mm[2] = {Pv.m[0].mass, Pv.m[1].mass}, mrs[2] = {Pv.m[0].rsoft, Pv.m[1].rsoft};
            // ---- volume integral and sources, :776-844 ---------------------------------------
            double vol[3][DG_K], src[3][DG_K]; // equation_19, source_weights
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int k = 0; k < DG_K; ++k) vol[q][k] = src[q][k] = 0.0;
#pragma unroll 1
            for (int nd = 0; nd < 9; ++nd) {
                const double x = xc + dg_tab.vol_cx[nd] * dx;
                const double y = yc + dg_tab.vol_cy[nd] * dy;
                const double gw = dg_tab.vol_gw[nd];
                const double cs2 = dg_cs2<LOCISO>(Pv.cs2, rmach2, Pv.eos, x, y, mx, my, mm, mrs);
                double u[3], p[3], fx[3], fy[3];
                dg_eval<63>(cc, dg_tab.vol_phi[nd], u);
                const DgRcp ru = dg_rcp(u[0]);
                dg_c2p(u, ru, p, Pv.vmax);
                dg_flux(p, u, cs2, p[1], 1.0, 0.0, fx);
                dg_flux(p, u, cs2, p[2], 0.0, 1.0, fy);
                if (viscous) {
                    double ux[3], uy[3], txx, txy, tyy;
                    dg_eval<0x38>(cc, dg_tab.vol_dpx[nd], ux); // m >= 1: slots 3, 4, 5
                    dg_eval<0x16>(cc, dg_tab.vol_dpy[nd], uy); // n >= 1: slots 1, 2, 4
                    dg_stress(u, ru, ux, uy, Pv.nu, rdx, rdy, txx, txy, tyy);
                    fx[1] -= txx;
                    fx[2] -= txy;
                    fy[1] -= txy;
                    fy[2] -= tyy;
                }
#pragma unroll
                for (int k = 1; k < DG_K; ++k) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        if (0x38 & (1 << k)) vol[q][k] += dx * fx[q] * dg_tab.vol_dpx[nd][k] * gw;
                        if (0x16 & (1 << k)) vol[q][k] += dy * fy[q] * dg_tab.vol_dpy[nd][k] * gw;
                    }
                }
                double du[3] = {0.0, 0.0, 0.0};
                dg_buffer_source(Pv, x, y, dt, u, du);
#if defined(SFHIP_DIAG_DG_MASSES_APART)  // diagnosis builds only: the form before the gravity terms were formed side by side
                dg_mass_source(Pv.m[0], x, y, dt, p, du);
                dg_mass_source(Pv.m[1], x, y, dt, p, du);
#else
                dg_two_mass_sources(Pv.m, x, y, dt, p, du);
#endif
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int k = 0; k < DG_K; ++k)
                        src[q][k] += 0.25 * du[q] * dg_tab.vol_phi[nd][k] * gw;
            }

            // ---- :963-981 -------------------------------------------------------------------------
            const double *w0 = weights0 + (long)(i + 1) * si + zoff; // (scalar row base + this lane's 32-bit offset)
            double *w2 = weights2 + (long)(i + 1) * si + zoff;
            double avg[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int k = 0; k < DG_K; ++k) {
                    const int s = q * 9 + DG_SLOT[k];
                    double wn = cc[q][k] + dg_div((vol[q][k] - srf[q][k]) * 0.5 * dt, rvol) + src[q][k];
                    wn = (1.0 - Pv.rk) * wn + Pv.rk * w0[s];
                    w2[s] = wn;
                    if (k == 0) avg[q] = wn;
                }
            }

            if (Pv.want_ws) { // cbdisodg_2d_wavespeed (:1152-1175) of the zone just written
                double pa[3];
                dg_c2p(avg, dg_rcp(avg[0]), pa, Pv.vmax);
                double cs2 = dg_cs2(Pv.cs2, rmach2, Pv.eos, xc, yc, Pv.wx, Pv.wy, Pv.wm, Pv.wrs);
                speed = max_nan(speed, max_speed(pa[1], pa[2], cs2));
            }
        }
    }
    if (P.want_ws) {
        speed = wave_max_nonneg(speed);
        // a plain load first skips the same-address atomic for every wave that cannot raise the maximum any more
        if (lane0 == 0 && !(speed <= *(volatile double *)wavespeed_max)) atomic_max_nonneg(wavespeed_max, speed);
    }
}

// ---------------------------------------------------------------------------------------------
// cbdisodg_2d_advance_rk, :607-983, the form of rounds 1-2: one lane per zone, every zone evaluates its four faces
// itself (as the reference does), so every face is evaluated twice.  Kept behind SFHIP_DG_ZONE_KERNEL=1 for A/B
// timing against dg_advance_kernel below, whose results are the same bit for bit.
// grid: (ceil(nj / 64), i_end - i_begin), block 64
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dg_advance_zone_kernel(DgParams P, const double *__restrict__ weights0,
                                                        const double *__restrict__ weights1,
                                                        double *__restrict__ weights2, double *wavespeed_max)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = P.i_begin + blockIdx.y;
    const bool active = j < P.nj;
    double speed = 0.0;
    if (active) {
        const long sj = DG_NW, si = (long)DG_NW * (P.nj + 2);
        const long at = (i + 1) * si + (j + 1) * sj;
        const double dt = P.dt_ptr ? *P.dt_ptr : P.dt;
        const double dx = P.dx, dy = P.dy;
        const double xc = P.xl + (i + 0.5) * dx;
        const double yc = P.yl + (j + 0.5) * dy;
        const double mx[2] = {P.m[0].x, P.m[1].x}, my[2] = {P.m[0].y, P.m[1].y};
        const double mm[2] = {P.m[0].mass, P.m[1].mass}, mrs[2] = {P.m[0].rsoft, P.m[1].rsoft};
        const bool viscous = P.nu > 0.0;
        const DgRcp rdx = dg_rcp(dx), rdy = dg_rcp(dy), rmach2 = dg_rcp(P.mach2);

        double cc[3][DG_K];
        dg_load(weights1 + at, cc);

        // neighbour offsets, by face; an outflow edge's guard zone is a copy of the edge zone itself
        long nb_off[4] = {-si, +si, -sj, +sj};
        if (P.clamp_lo && i == 0) nb_off[0] = 0;
        if (P.clamp_hi && i == P.ni - 1) nb_off[1] = 0;
        if (P.clamp_j && j == 0) nb_off[2] = 0;
        if (P.clamp_j && j == P.nj - 1) nb_off[3] = 0;

        double vol[3][DG_K], srf[3][DG_K], src[3][DG_K]; // equation_19, equation_20, source_weights
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int k = 0; k < DG_K; ++k)
                vol[q][k] = srf[q][k] = src[q][k] = 0.0;

        // ---- surface integral, :848-961: per Gauss node the faces xl, xr, yl, yr ----------
#pragma unroll 1
        for (int fn = 0; fn < 12; ++fn) {
            const int face = fn & 3;
            const bool high = face & 1, ydir = face >> 1;
            // the neighbour's weights are requested first: the sound speed and the own trace below
            // (about 150 instructions) do not need them and cover part of the latency
            double nb[3][DG_K];
            dg_load(weights1 + at + (face == 0 ? nb_off[0] : face == 1 ? nb_off[1] : face == 2 ? nb_off[2] : nb_off[3]), nb);
            const double x = xc + dg_tab.face_cx[fn] * dx;
            const double y = yc + dg_tab.face_cy[fn] * dy;
            const double cs2 = dg_cs2(P.cs2, rmach2, P.eos, x, y, mx, my, mm, mrs);
            double uo[3], un[3]; // own and neighbour traces
            dg_eval<63>(cc, dg_tab.face_own[fn], uo);
            dg_eval<63>(nb, dg_tab.face_nb[fn], un);
            const DgRcp ro = dg_rcp(uo[0]), rn = dg_rcp(un[0]);
            double um[3], up[3]; // low-side and high-side states
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                um[q] = high ? uo[q] : un[q];
                up[q] = high ? un[q] : uo[q];
            }
            const DgRcp rm = {um[0], high ? ro.r : rn.r}, rp = {up[0], high ? rn.r : ro.r};
            // riemann_hlle, :295-324
            double pl[3], pr[3], fl[3], fr[3], fhat[3];
            dg_c2p(um, rm, pl, P.vmax);
            dg_c2p(up, rp, pr, P.vmax);
            const double e0 = ydir ? 0.0 : 1.0, e1 = ydir ? 1.0 : 0.0;
            const double vl = ydir ? pl[2] : pl[1], vr = ydir ? pr[2] : pr[1];
            dg_flux(pl, um, cs2, vl, e0, e1, fl);
            dg_flux(pr, up, cs2, vr, e0, e1, fr);
            const double cs = exact_sqrt0(cs2);
            const double am = min2(0.0, min2(vl - cs, vr - cs));
            const double ap = max2(0.0, max2(vl + cs, vr + cs));
            const DgRcp rspan = dg_rcp(ap - am);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                fhat[q] = dg_div(fl[q] * ap - fr[q] * am - (um[q] - up[q]) * ap * am, rspan);
            if (viscous) {
                // high side first, then low side (:860-869); the face-normal "gradient" IS the trace
                double to[3], tn[3];
                dg_eval<63>(cc, dg_tab.face_own_t[fn], to);
                dg_eval<63>(nb, dg_tab.face_nb_t[fn], tn);
                double oxx, oxy, oyy, nxx, nxy, nyy;
                double gxo[3], gyo[3], gxn[3], gyn[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    gxo[q] = ydir ? to[q] : uo[q];
                    gyo[q] = ydir ? uo[q] : to[q];
                    gxn[q] = ydir ? tn[q] : un[q];
                    gyn[q] = ydir ? un[q] : tn[q];
                }
                dg_stress(uo, ro, gxo, gyo, P.nu, rdx, rdy, oxx, oxy, oyy);
                dg_stress(un, rn, gxn, gyn, P.nu, rdx, rdy, nxx, nxy, nyy);
                const double o1 = ydir ? oxy : oxx, o2 = ydir ? oyy : oxy;
                const double n1 = ydir ? nxy : nxx, n2 = ydir ? nyy : nxy;
                fhat[1] -= high ? n1 : o1;
                fhat[2] -= high ? n2 : o2;
                fhat[1] -= high ? o1 : n1;
                fhat[2] -= high ? o2 : n2;
            }
            const double len = ydir ? dx : dy;
            const double gw = dg_tab.face_gw[fn];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int k = 0; k < DG_K; ++k) {
                    double term = len * fhat[q] * dg_tab.face_own[fn][k] * gw;
                    srf[q][k] += high ? term : -term;
                }
            }
        }

        // ---- volume integral and sources, :776-844 ---------------------------------------
#pragma unroll 1
        for (int nd = 0; nd < 9; ++nd) {
            const double x = xc + dg_tab.vol_cx[nd] * dx;
            const double y = yc + dg_tab.vol_cy[nd] * dy;
            const double gw = dg_tab.vol_gw[nd];
            const double cs2 = dg_cs2(P.cs2, rmach2, P.eos, x, y, mx, my, mm, mrs);
            double u[3], p[3], fx[3], fy[3];
            dg_eval<63>(cc, dg_tab.vol_phi[nd], u);
            const DgRcp ru = dg_rcp(u[0]);
            dg_c2p(u, ru, p, P.vmax);
            dg_flux(p, u, cs2, p[1], 1.0, 0.0, fx);
            dg_flux(p, u, cs2, p[2], 0.0, 1.0, fy);
            if (viscous) {
                double ux[3], uy[3], txx, txy, tyy;
                dg_eval<0x38>(cc, dg_tab.vol_dpx[nd], ux); // m >= 1: slots 3, 4, 5
                dg_eval<0x16>(cc, dg_tab.vol_dpy[nd], uy); // n >= 1: slots 1, 2, 4
                dg_stress(u, ru, ux, uy, P.nu, rdx, rdy, txx, txy, tyy);
                fx[1] -= txx;
                fx[2] -= txy;
                fy[1] -= txy;
                fy[2] -= tyy;
            }
#pragma unroll
            for (int k = 1; k < DG_K; ++k) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (0x38 & (1 << k)) vol[q][k] += dx * fx[q] * dg_tab.vol_dpx[nd][k] * gw;
                    if (0x16 & (1 << k)) vol[q][k] += dy * fy[q] * dg_tab.vol_dpy[nd][k] * gw;
                }
            }
            double du[3] = {0.0, 0.0, 0.0};
            dg_buffer_source(P, x, y, dt, u, du);
            dg_mass_source(P.m[0], x, y, dt, p, du);
            dg_mass_source(P.m[1], x, y, dt, p, du);
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int k = 0; k < DG_K; ++k)
                    src[q][k] += 0.25 * du[q] * dg_tab.vol_phi[nd][k] * gw;
        }

        // ---- :963-981 -------------------------------------------------------------------------
        const double *w0 = weights0 + at;
        double *w2 = weights2 + at;
        double avg[3];
        const DgRcp rvol = dg_rcp(P.cell_volume);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int k = 0; k < DG_K; ++k) {
                const int s = q * 9 + DG_SLOT[k];
                double w = cc[q][k] + dg_div((vol[q][k] - srf[q][k]) * 0.5 * dt, rvol) + src[q][k];
                w = (1.0 - P.rk) * w + P.rk * w0[s];
                w2[s] = w;
                if (k == 0) avg[q] = w;
            }
        }

        if (P.want_ws) { // cbdisodg_2d_wavespeed (:1152-1175) of the zone just written
            double pa[3];
            dg_c2p(avg, dg_rcp(avg[0]), pa, P.vmax);
            double cs2 = dg_cs2(P.cs2, rmach2, P.eos, xc, yc, P.wx, P.wy, P.wm, P.wrs);
            speed = max_speed(pa[1], pa[2], cs2);
        }
    }
    if (P.want_ws) {
        speed = wave_max_nonneg(speed);
        // 65 536 one-wave blocks at 2048^2, finishing over 2 ms: a plain load first skips the same-address atomic for
        // every wave that cannot raise the maximum any more (1-2 % of the stage)
        if (threadIdx.x == 0 && !(speed <= *(volatile double *)wavespeed_max)) atomic_max_nonneg(wavespeed_max, speed);
    }
}

// ---------------------------------------------------------------------------------------------
// cbdisodg_2d_slope_limit, :498-605.  grid: (ceil(nj / 64), i_end - i_begin), block 64
// ---------------------------------------------------------------------------------------------

// minmod_simple, :455-493, including the truncation of its `int` return type; out-of-range and
// NaN picks give INT_MIN as the reference's x86 conversion does
__device__ __forceinline__ double dg_minmod_truncated(double w1, double w0l, double w0, double w0r)
{
    const double s3 = 1.7320508075688772; // sqrt(3.0), correctly rounded
    double a = w1;
    double b = (w0 - w0l) * 1.0 / s3;
    double c = (w0r - w0) * 1.0 / s3;
    double pick = 0.0;
    if (a <= 0.0 && b <= 0.0 && c <= 0.0) {
        if (a >= b && a >= c) pick = a;
        else if (b >= c && b >= a) pick = b;
        else pick = c;
    } else if (a >= 0.0 && b >= 0.0 && c >= 0.0) {
        if (a <= b && a <= c) pick = a;
        else if (b <= c && b <= a) pick = b;
        else pick = c;
    }
    if (!(fabs(pick) < 2147483648.0)) return -2147483648.0;
    return (double)(int)pick;
}

// One wave per 64 consecutive zones of a row, DG_LIMIT_ROWS consecutive rows per workgroup.  The zones' 27 doubles are
// contiguous in memory, so a wave moves its 13.8 KB segment (plus the two neighbouring zones in j) through LDS with
// unit-stride loads and stores; a lane then works on "its" zone in LDS (stride 27 doubles = 54 banks: conflict-free
// for 64-bit accesses).  The five doubles it needs of its two neighbours in i come from the tiles of the waves above
// and below it in the workgroup; only the workgroup's first and last row read them from global memory with the zone
// stride (one request per lane and double: with every row doing that -- 10 strided requests per lane beside 55
// unit-stride sweeps per wave -- the kernel took 0.41 ms at 2048^2, without them 0.31).
// grid: (ceil(nj / 64), ceil((i_end - i_begin) / DG_LIMIT_ROWS)), block (64, DG_LIMIT_ROWS)
#if !defined(SFHIP_DG_LIMIT_ROWS)
#define SFHIP_DG_LIMIT_ROWS 4
#endif
constexpr int DG_LIMIT_ROWS = SFHIP_DG_LIMIT_ROWS;

__global__ __launch_bounds__(64 * DG_LIMIT_ROWS) void dg_limit_kernel(int ni, int nj, const double *__restrict__ weights1,
                                                                      double *__restrict__ weights2, int clamp_lo, int clamp_hi,
                                                                      int clamp_j, int i_begin, int i_end)
{
    constexpr int SWEEPS = (66 * DG_NW + 63) / 64; // 28 unit-stride sweeps of the wave cover 66 zones
    __shared__ double tiles[DG_LIMIT_ROWS][SWEEPS * 64];
    const int t = threadIdx.x, w = threadIdx.y;
    double *tile = tiles[w];
    const int j0 = blockIdx.x * 64;
    const int i = i_begin + blockIdx.y * DG_LIMIT_ROWS + w;
    const bool row_ok = i < i_end; // (a workgroup's last waves may stand beyond the range: they only keep the barriers)
    const int nvalid = nj - j0 < 64 ? nj - j0 : 64;
    const long sj = DG_NW, si = (long)DG_NW * (nj + 2);
    const int nload = (nvalid + 2) * DG_NW;
    if (row_ok) {
        const double *row = weights1 + (i + 1) * si + (long)j0 * sj; // zone j0 - 1 (a guard zone when j0 == 0)
        // all loads are issued before the first wait: addresses past the segment are clamped (the
        // value lands in an unused part of the tile) instead of branching around the load
        double staged[SWEEPS];
#pragma unroll
        for (int r = 0; r < SWEEPS; ++r) {
            const int k = t + 64 * r;
            staged[r] = row[k < nload ? k : nload - 1];
        }
#pragma unroll
        for (int r = 0; r < SWEEPS; ++r)
            tile[t + 64 * r] = staged[r];
    }
    __syncthreads();

    const int j = j0 + t;
    const bool active = row_ok && t < nvalid;
    double wx[3] = {0.0, 0.0, 0.0}, wy[3] = {0.0, 0.0, 0.0};
    bool change[3] = {false, false, false};
    if (active) {
        const double *cc = tile + (t + 1) * DG_NW;
        const double *lj = (clamp_j && j == 0) ? cc : cc - DG_NW;
        const double *rj = (clamp_j && j == nj - 1) ? cc : cc + DG_NW;
        const double *gcc = weights1 + (i + 1) * si + (j + 1) * sj;
        // the same zone one row down / up: in a neighbouring wave's tile (still as loaded: nobody has limited yet),
        // else in global memory
        const bool lo_tile = w > 0, hi_tile = w + 1 < DG_LIMIT_ROWS && i + 1 < i_end;
        const double *lo_cc = tiles[lo_tile ? w - 1 : w] + (t + 1) * DG_NW;
        const double *hi_cc = tiles[hi_tile ? w + 1 : w] + (t + 1) * DG_NW;
        double li[5], ri[5]; // slots [0][0][0], [0][1][0], [0][2][0], [1][0][0], [2][0][0]
        const int need[5] = {0, 3, 6, 9, 18};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            li[k] = (clamp_lo && i == 0) ? cc[need[k]] : lo_tile ? lo_cc[need[k]] : gcc[need[k] - si];
            ri[k] = (clamp_hi && i == ni - 1) ? cc[need[k]] : hi_tile ? hi_cc[need[k]] : gcc[need[k] + si];
        }
        const double s3 = 1.7320508075688772, s5 = 2.23606797749979; // sqrt(3.0), sqrt(5.0) correctly rounded
        const double c00 = cc[0];
        const double big = max2(max2(fabs(c00), fabs(li[0])), max2(fabs(ri[0]), max2(fabs(lj[0]), fabs(rj[0]))));
        const double a = 4.0 * li[0] + 8.0 * s3 * li[1] + 24.0 * s5 * li[2];
        const double b = 4.0 * ri[0] - 8.0 * s3 * ri[1] + 24.0 * s5 * ri[2];
        const double c = 4.0 * lj[0] + 8.0 * s3 * lj[1] + 24.0 * s5 * lj[2];
        const double d = 4.0 * rj[0] - 8.0 * s3 * rj[1] + 24.0 * s5 * rj[2];
        const double tci = (fabs(c00 - a / 4.0) + fabs(c00 - b / 4.0) + fabs(c00 - c / 4.0) + fabs(c00 - d / 4.0)) / big;
        if (tci > 0.1) {
            const double li0[3] = {li[0], li[3], li[4]}, ri0[3] = {ri[0], ri[3], ri[4]};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                wx[q] = dg_minmod_truncated(cc[q * 9 + 3], li0[q], cc[q * 9], ri0[q]);
                wy[q] = dg_minmod_truncated(cc[q * 9 + 1], lj[q * 9], cc[q * 9], rj[q * 9]);
                change[q] = wx[q] != cc[q * 9 + 3] || wy[q] != cc[q * 9 + 1];
            }
        }
    }
    __syncthreads(); // every lane of every wave has read its neighbours' weights before anyone overwrites them
    if (active) {
        double *cc = tile + (t + 1) * DG_NW;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (change[q]) {
#pragma unroll
                for (int k = 1; k < 9; ++k)
                    cc[q * 9 + k] = 0.0;
                cc[q * 9 + 3] = wx[q];
                cc[q * 9 + 1] = wy[q];
            }
        }
    }
    __syncthreads();
    if (row_ok) {
        double *orow = weights2 + (i + 1) * si + (long)(j0 + 1) * sj;
        const int nstore = nvalid * DG_NW;
#pragma unroll
        for (int r = 0; r < DG_NW; ++r) {
            const int k = t + 64 * r;
            if (k < nstore) orow[k] = tile[DG_NW + k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cbdisodg_2d_wavespeed, :1108-1176, as a per-zone array (reference form) and/or folded into a
// device scalar.  grid: (ceil(nj / 64), ni), block 64
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dg_wavespeed_kernel(DgParams P, const double *__restrict__ weights,
                                                          double *__restrict__ wavespeed, double *wavespeed_max)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y;
    double speed = 0.0;
    if (j < P.nj) {
        const double *cc = weights + ((long)(i + 1) * (P.nj + 2) + (j + 1)) * DG_NW;
        const double x = P.xl + (i + 0.5) * P.dx;
        const double y = P.yl + (j + 0.5) * P.dy;
        const double u[3] = {cc[0], cc[9], cc[18]};
        double p[3];
        dg_c2p(u, dg_rcp(u[0]), p, P.vmax);
        const double cs2 = dg_cs2(P.cs2, dg_rcp(P.mach2), P.eos, x, y, P.wx, P.wy, P.wm, P.wrs);
        speed = max_speed(p[1], p[2], cs2);
        if (wavespeed) wavespeed[(long)(i + 1) * (P.nj + 2) + (j + 1)] = speed;
    }
    if (wavespeed_max) {
        speed = wave_max_nonneg(speed);
        if (threadIdx.x == 0) atomic_max_nonneg(wavespeed_max, speed);
    }
}

} // namespace sf
