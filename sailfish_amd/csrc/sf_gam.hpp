// sf_gam.hpp -- cbdgam_2d (gamma-law, energy-conserving circumbinary disk)
// kernels for gfx950.  Same marching-wave structure as sf_iso.hpp.
//
// Differences that shape the kernel (cbdgam_2d.c:439-683):
//   * 4 fields; PLM theta = 1.5.
//   * The sound speed used by HLLE at a face is that of the NEIGHBOURING zone
//     centre, so the two zones sharing a face use different signal speeds
//     (cbdgam_2d.c:613-622).  A face therefore yields two flux vectors that
//     share the reconstructed states and their conserved/flux vectors:
//     `fR` (what the upper zone R subtracts as its lower-face flux, built with
//     cs2 of L) and `fL` (what the lower zone L uses as its upper-face flux,
//     built with cs2 of R).
//   * alpha viscosity nu = alpha h cs with the neighbour's scale height taken
//     at the FACE position but the zone's own at its centre
//     (cbdgam_2d.c:638-648) -- again two variants per face.
//   * Sources act on the conserved state directly, in the order buffer ->
//     point masses -> cooling (cbdgam_2d.c:669-672).
#pragma once
#include "sf_device.hpp"

namespace sf {

struct GamParams {
    int ni, nj;
    double xl, yl, dx, dy, inv_dx, inv_dy;
    Mass m[2];
    double gamma, gm1, inv_gm1, sqrt_gamma;      // gamma-1, 1/(gamma-1), sqrt(gamma) from the host
    double buf_sd, buf_sp, buf_mc, buf_rate, buf_rout, buf_ron, buf_omega;
    double buf_ron2, buf_rate_scale, buf_e0;     // ron^2, rate*omega/(rout-ron), surface_pressure/(gamma-1)
    int buf_on;
    int constant_softening;
    int any_mass;                                // some mass != 0: disk_height is not the constant 1
    double alpha, a, dt, vmax, cooling, mach_ceiling_m2 /* pow(mach_ceiling,-2) */, dfloor, pfloor;
    double mach_floor_scale;                     // 2/gamma/(gamma-1)*mach_ceiling^-2, cbdgam_2d.c:314
    int clamp_ilo, clamp_ihi, clamp_j, un_mode;
    int rows_per_chunk, i_begin, i_end, want_wavespeed;
    int reverse;                // chunks are handed out from the last rows to the first (see launch_gam_stage)
    double ws_floor; // zones slower than this are not folded in (see include/sfhip.h)
    const double *dt_ptr;       // when set, the timestep is read from device memory instead of `dt`
    const double *amax_ptr;     // when set: dt = dt_dx / *amax_ptr * dt_cfl formed in the kernel (see IsoParams)
    double dt_dx, dt_cfl;
    double *zero_ptr;           // when set, one lane stores 0.0 there
    double *host_out;           // when set (with amax_ptr): *amax_ptr, then host_seq, to pinned host memory (see IsoParams)
    double host_seq;
    // uniform values evaluated on the host so that they arrive in SGPRs (no scalar fp64 unit: formed
    // in the kernel they would sit in vector registers for the whole march)
    double half_inv_dx, half_inv_dy, one_minus_a, inv_buf_sd, sink_near2[2]; // sink_near2 = (4 r_sink)^2
    double half_alpha;          // alpha / 2
};

// ---- exact pieces (reference-named small kernels, wavespeed/dt path) -------

// sum_p m_p r_p^-3 with r^2 + 1e-12 (cbdgam_2d.c:85-101); returns < 0 when both masses are zero
__device__ __forceinline__ double gam_omega2(const Mass *m, double x, double y)
{
    if (m[0].mass == 0.0 && m[1].mass == 0.0) return -1.0;
    double om2 = 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (m[p].mass > 0.0) {
            double ddx = x - m[p].x;
            double ddy = y - m[p].y;
            double r2 = ddx * ddx + ddy * ddy + 1e-12;
            double r = sqrt(r2);
            om2 += m[p].mass * (1.0 / (r * r * r)); // pow(r, -3.0)
        }
    }
    return om2;
}

// disk_height, cbdgam_2d.c:75-106, from sqrt(p/sigma) of the zone and omega^2 at the evaluation point
__device__ __forceinline__ double gam_height(double sqrt_p_over_sigma, double om2)
{
    return om2 < 0.0 ? 1.0 : sqrt_p_over_sigma / sqrt(om2);
}

__device__ __forceinline__ void gam_p2c(const double *p, double *u, double gamma)
{
    u[0] = p[0];
    u[1] = p[1] * p[0];
    u[2] = p[2] * p[0];
    u[3] = p[3] / (gamma - 1.0) + 0.5 * p[0] * (p[1] * p[1] + p[2] * p[2]);
}

// One point mass, cbdgam_2d.c:108-203, IEEE arithmetic (diagnostics kernel).  Adds its contribution to u.
// `us`: where the sink's contributions go (the diagnostics keep gravity and accretion apart); `u` when null.
// With the two apart, each part has exactly the value the reference's gravity-only (sink_rate * 0) and
// accretion-only (mass * 0) calls give it: the other part's terms are exact zeros there.
__device__ __forceinline__ void gam_mass_source(const Mass &m, double x1, double y1, double dt, const double *prim,
                                                double h, double *u, int constant_softening, double gamma,
                                                double *us = nullptr)
{
    double sigma = prim[0];
    double eps = prim[3] / sigma / (gamma - 1.0);
    double ddx = x1 - m.x;
    double ddy = y1 - m.y;
    double r2 = ddx * ddx + ddy * ddy;
    double dr = sqrt(r2);
    double r_sink = m.sink_radius;
    double r_soft;
    if (constant_softening) {
        r_soft = m.rsoft;
    } else if (dr > r_sink) {
        r_soft = 0.5 * h;
    } else {
        double q = dr / r_sink;
        double t = 1.0 - q * q;
        double tr = t * t;
        r_soft = tr * r_sink + (1.0 - tr) * 0.5 * h;
    }
    double rate = 0.0;
    if (dr < 4.0 * r_sink) {
        double q = dr / r_sink;
        double q2 = q * q;
        rate = m.sink_rate * exp(-(q2 * q2));
    }
    double r2s = r2 + r_soft * r_soft;
    double fnum = sigma * m.mass * (1.0 / (r2s * sqrt(r2s)));
    double fx = -fnum * ddx;
    double fy = -fnum * ddy;
    double mdot = sigma * rate * -1.0;
    double vx = prim[1], vy = prim[2];
    if (us) {
        if (m.sink_model < 1 || m.sink_model > 3) return;
        u[1] += dt * fx;
        u[2] += dt * fy;
        u[3] += dt * (fx * vx + fy * vy);
        us[0] += dt * mdot;
        if (m.sink_model == 1) {
            us[1] += dt * mdot * vx;
            us[2] += dt * mdot * vy;
            us[3] += dt * (mdot * eps + 0.5 * mdot * (vx * vx + vy * vy));
        } else if (m.sink_model == 2) {
            double rhx = ddx / (dr + 1e-12);
            double rhy = ddy / (dr + 1e-12);
            double dv = (vx - m.vx) * rhx + (vy - m.vy) * rhy;
            double vxs = dv * rhx + m.vx;
            double vys = dv * rhy + m.vy;
            us[1] += dt * mdot * vxs;
            us[2] += dt * mdot * vys;
            us[3] += dt * (mdot * eps + 0.5 * mdot * (vxs * vxs + vys * vys));
        }
        return;
    }
    if (m.sink_model == 1) {
        u[0] += dt * mdot;
        u[1] += dt * mdot * vx + dt * fx;
        u[2] += dt * mdot * vy + dt * fy;
        u[3] += dt * (mdot * eps + 0.5 * mdot * (vx * vx + vy * vy)) + dt * (fx * vx + fy * vy);
    } else if (m.sink_model == 2) {
        double rhx = ddx / (dr + 1e-12);
        double rhy = ddy / (dr + 1e-12);
        double dv = (vx - m.vx) * rhx + (vy - m.vy) * rhy;
        double vxs = dv * rhx + m.vx;
        double vys = dv * rhy + m.vy;
        u[0] += dt * mdot;
        u[1] += dt * mdot * vxs + dt * fx;
        u[2] += dt * mdot * vys + dt * fy;
        u[3] += dt * (mdot * eps + 0.5 * mdot * (vxs * vxs + vys * vys)) + dt * (fx * vx + fy * vy);
    } else if (m.sink_model == 3) {
        u[0] += dt * mdot;
        u[1] += dt * fx;
        u[2] += dt * fy;
        u[3] += dt * (fx * vx + fy * vy);
    }
}

// ---- fast pieces: the stage kernel (fast_rcp / fast_rsqrt, FMA contraction) -

// 1/sqrt(sum_p m_p r_p^-3) at a point; c0, c1 = (y - y_p)^2 + 1e-12 precomputed per lane.
// Multiplying sqrt(p/sigma) by it gives disk_height (cbdgam_2d.c:75-106).
// FULL: both masses are known (on the host) to be positive, see gam_stage_kernel.
// `d0`, `d1`: x - x_p of the two masses (uniform along a row: the march reads them from its row table, sf_iso.hpp).
template <bool FULL = false>
__device__ __forceinline__ double gam_inv_omega_fast(const Mass *m, double d0, double d1, double c0, double c1)
{
#pragma clang fp contract(fast)
    // (explicit operations: the bring-up before the march and the march itself evaluate this for the same zone)
    double om2 = 0.0;
    if (FULL || m[0].mass > 0.0) { double y = fast_rsqrt(__builtin_fma(d0, d0, c0)); om2 = __builtin_fma(m[0].mass, y * y * y, om2); }
    if (FULL || m[1].mass > 0.0) { double y = fast_rsqrt(__builtin_fma(d1, d1, c1)); om2 = __builtin_fma(m[1].mass, y * y * y, om2); }
    return fast_rsqrt(om2);
}

// Primitive recovery with floors, cbdgam_2d.c:319-345 (the divisions through a ~1-ulp reciprocal: fast_rcp).
__device__ __forceinline__ void gam_c2p(const double *u, double gm1, double vmax, double dfloor, double pfloor, double *p)
{
#pragma clang fp contract(fast)
    double rinv = fast_rcp(u[0]);
    // max2(pressure_floor, .) of the reference: a NaN energy stays a NaN pressure (fmax would floor it)
    double pres = (u[3] - 0.5 * (u[1] * u[1] + u[2] * u[2]) * rinv) * gm1;
    pres = pres < pfloor ? pfloor : pres;
    double vx = copysign(min_abs_s_f64(u[1] * rinv, vmax), u[1]);
    double vy = copysign(min_abs_s_f64(u[2] * rinv, vmax), u[2]);
    double rho = u[0];
    if (u[0] < dfloor) { rho = dfloor; vx = 0.0; vy = 0.0; pres = pfloor; }
    p[0] = rho; p[1] = vx; p[2] = vy; p[3] = pres;
}

// Two HLLE fluxes through one face (see header comment), cbdgam_2d.c:374-434;
// csL / csR are the sound speeds of the zones below / above the face.
template <int DIR>
__device__ __forceinline__ void gam_hlle2_fast(const double *pm, const double *pp, double csL, double csR,
                                               double inv_gm1, double *fR, double *fL)
{
#pragma clang fp contract(fast)
    // (Fl ap - Fr am - (Ul - Ur) ap am) / (ap - am) with the three weights formed once per flux.  The regrouping by
    // side that iso_hlle uses (sf_iso.hpp; 12 instructions per face fewer here) was built and withdrawn: equal to
    // rounding on every test, but the free-running 384^2 disk, whose domain corners sit on the pressure floor next to
    // zones of order-one pressure from t ~ 1.3 on (the reference scheme's own weak spot), then blew up at t = 1.42
    // where this form carries on -- and the step rate gained 1 %.
    double ul[4], ur[4];
    ul[0] = pm[0]; ul[1] = pm[1] * pm[0]; ul[2] = pm[2] * pm[0];
    ul[3] = pm[3] * inv_gm1 + 0.5 * pm[0] * (pm[1] * pm[1] + pm[2] * pm[2]);
    ur[0] = pp[0]; ur[1] = pp[1] * pp[0]; ur[2] = pp[2] * pp[0];
    ur[3] = pp[3] * inv_gm1 + 0.5 * pp[0] * (pp[1] * pp[1] + pp[2] * pp[2]);
    const double vl = pm[1 + DIR], vr = pp[1 + DIR];
    double fl[4], fr[4], du[4];
    fl[0] = vl * ul[0]; fl[1] = vl * ul[1]; fl[2] = vl * ul[2]; fl[3] = vl * (ul[3] + pm[3]);
    fr[0] = vr * ur[0]; fr[1] = vr * ur[1]; fr[2] = vr * ur[2]; fr[3] = vr * (ur[3] + pp[3]);
    fl[1 + DIR] += pm[3];
    fr[1 + DIR] += pp[3];
#pragma unroll
    for (int q = 0; q < 4; ++q) du[q] = ul[q] - ur[q];
    // min(vl - cs, vr - cs) = min(vl, vr) - cs exactly (rounding is monotone): the extremes of the two normal
    // velocities are formed once per face, each flux subtracts / adds its own sound speed
    const double vlo = min_f64(vl, vr), vhi = max_f64(vl, vr);
    auto flux = [&](double cs, double *f) {
        const double am = min0_f64(vlo - cs);
        const double ap = max0_f64(vhi + cs);
        const double rden = fast_rcp(ap - am);
        const double wl = ap * rden, wr = am * rden, wd = wl * am;
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = fl[q] * wl - fr[q] * wr - du[q] * wd;
    };
    flux(csL, fR);
    flux(csR, fL);
}

// One point mass, cbdgam_2d.c:108-203.  `ddy`, `cy` = ddy^2 (+ rs^2 when softening is constant) per lane;
// `eps` = specific internal energy, `h` = disk height (non-constant softening only).
// `near2` = (4 r_sink)^2 from the host; `cold` addresses the same mass in memory for what only the
// sink interior needs (loaded there, on demand, instead of occupying SGPRs throughout the march).
template <bool FULL = false>
__device__ __forceinline__ void gam_mass_source_fast(const Mass &m, double near2, const Mass *cold,
                                                     double ddx, double ddy, double dt,
                                                     const double *prim, double eps, double h, double *u,
                                                     int constant_softening)
{
#pragma clang fp contract(fast)
    if (!FULL && (m.sink_model < 1 || m.sink_model > 3)) return; // inactive: no gravity either (cbdgam_2d.c:194-201)
    double sigma = prim[0];
    double r2 = ddx * ddx + ddy * ddy;
    double r_sink = m.sink_radius;
    double rs4 = 4.0 * r_sink;
    double r_soft = m.rsoft;
    double rate = 0.0;
    double dr = 0.0;
    const bool near = r2 < near2;
    if (!constant_softening || near) dr = sqrt(r2);
    if (!constant_softening) {
        if (dr > r_sink) r_soft = 0.5 * h;
        else {
            double q = dr / r_sink;
            double t = 1.0 - q * q;
            double tr = t * t;
            r_soft = tr * r_sink + (1.0 - tr) * 0.5 * h;
        }
    }
    if (near && dr < rs4) {
        double q = dr / r_sink;
        double q2 = q * q;
        rate = cold->sink_rate * exp_cold(-(q2 * q2));
    }
    double y = fast_rsqrt(r2 + r_soft * r_soft);
    double fnum = sigma * m.mass * (y * y * y);
    double fx = -fnum * ddx;
    double fy = -fnum * ddy;
    double vx = prim[1], vy = prim[2];
    double work = dt * (fx * vx + fy * vy);
    u[1] += dt * fx;
    u[2] += dt * fy;
    u[3] += work;
    if (rate != 0.0) {
        double mdot = -(sigma * rate);
        u[0] += dt * mdot;
        if (m.sink_model == 1) {
            u[1] += dt * mdot * vx;
            u[2] += dt * mdot * vy;
            u[3] += dt * (mdot * eps + 0.5 * mdot * (vx * vx + vy * vy));
        } else if (m.sink_model == 2) {
            double rinv = 1.0 / (dr + 1e-12);
            double rhx = ddx * rinv, rhy = ddy * rinv;
            const double mvx = cold->vx, mvy = cold->vy;
            double dv = (vx - mvx) * rhx + (vy - mvy) * rhy;
            double vxs = dv * rhx + mvx;
            double vys = dv * rhy + mvy;
            u[1] += dt * mdot * vxs;
            u[2] += dt * mdot * vys;
            u[3] += dt * (mdot * eps + 0.5 * mdot * (vxs * vxs + vys * vys));
        }
    }
}

// Both point masses with constant softening (the full-physics instantiation): the same operations in the same order
// on u as two calls of gam_mass_source_fast, arranged so that the two gravity terms -- a reciprocal square root and
// its refinement each -- are formed side by side, ahead of the (divergent, rarely taken) sink blocks, of which there
// is one per mass instead of three.
__device__ __forceinline__ void gam_two_masses_fast(const Mass *m, const double *near2, const Mass *cold, double ddx0,
                                                    double ddx1, double ddy0, double ddy1, double dt, const double *prim,
                                                    double eps, double *u)
{
#pragma clang fp contract(fast)
    const double sigma = prim[0], vx = prim[1], vy = prim[2];
    const double ddx[2] = {ddx0, ddx1}, ddy[2] = {ddy0, ddy1};
    double r2[2], fx[2], fy[2], work[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        r2[k] = ddx[k] * ddx[k] + ddy[k] * ddy[k];
        double r_soft = m[k].rsoft;
        double y = fast_rsqrt(r2[k] + r_soft * r_soft);
        double fnum = sigma * m[k].mass * (y * y * y);
        fx[k] = -fnum * ddx[k];
        fy[k] = -fnum * ddy[k];
        work[k] = dt * (fx[k] * vx + fy[k] * vy);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        u[1] += dt * fx[k];
        u[2] += dt * fy[k];
        u[3] += work[k];
        if (r2[k] < near2[k]) {
            double r_sink = m[k].sink_radius;
            double dr = sqrt(r2[k]);
            if (dr < 4.0 * r_sink) {
                double q = dr / r_sink;
                double q2 = q * q;
                double rate = cold[k].sink_rate * exp_cold(-(q2 * q2));
                if (rate != 0.0) {
                    double mdot = -(sigma * rate);
                    u[0] += dt * mdot;
                    if (m[k].sink_model == 1) {
                        u[1] += dt * mdot * vx;
                        u[2] += dt * mdot * vy;
                        u[3] += dt * (mdot * eps + 0.5 * mdot * (vx * vx + vy * vy));
                    } else if (m[k].sink_model == 2) {
                        double rinv = 1.0 / (dr + 1e-12);
                        double rhx = ddx[k] * rinv, rhy = ddy[k] * rinv;
                        const double mvx = cold[k].vx, mvy = cold[k].vy;
                        double dv = (vx - mvx) * rhx + (vy - mvy) * rhy;
                        double vxs = dv * rhx + mvx;
                        double vys = dv * rhy + mvy;
                        u[1] += dt * mdot * vxs;
                        u[2] += dt * mdot * vys;
                        u[3] += dt * (mdot * eps + 0.5 * mdot * (vxs * vxs + vys * vys));
                    }
                }
            }
        }
    }
}

// exact wavespeed of one zone, out of line (see iso_exact_speed in sf_iso.hpp)
__device__ __attribute__((noinline)) double gam_exact_speed(double vx, double vy, double pres, double rho, double gamma)
{
#pragma clang fp contract(off)
    return max_speed(vx, vy, pres / rho * gamma);
}

// cbdgam_2d.c:238-276
template <bool FULL = false>
__device__ __forceinline__ void gam_buffer_fast(const GamParams &P, double xc, double yc, double yc2, double *u)
{
#pragma clang fp contract(fast)
    if (!FULL && !P.buf_on) return;
    double rc2 = xc * xc + yc2;
    if (rc2 > P.buf_ron2) {
        double rinv = fast_rsqrt(rc2);
        double rc = sqrt_from_rsqrt(rc2, rinv);
        double vk2 = P.buf_mc * rinv;
        double pf = P.buf_sd * fast_sqrt(vk2) * rinv; // sigma v_kep / rc
        double px = -pf * yc, py = pf * xc;
        double en = P.buf_e0 + 0.5 * (px * px + py * py) * P.inv_buf_sd;
        double br = P.buf_rate_scale * (rc - P.buf_ron) * P.dt;
        u[0] -= (u[0] - P.buf_sd) * br;
        u[1] -= (u[1] - px) * br;
        u[2] -= (u[2] - py) * br;
        u[3] -= (u[3] - en) * br;
    }
}

// cbdgam_2d.c:298-317
template <bool FULL = false>
__device__ __forceinline__ void gam_cooling_fast(const GamParams &P, const double *prim, double eps, double rinv_sigma,
                                                 double *u)
{
#pragma clang fp contract(fast)
    double sigma = prim[0];
    double ec = eps;
    if (!FULL && P.cooling != 0.0) {
        double arg = 1.0 + 3.0 * P.cooling * (rinv_sigma * rinv_sigma) * (eps * eps * eps) * P.dt;
        ec = eps * rcbrt(arg);
    }
    double vx = prim[1], vy = prim[2];
    double ek = 0.5 * (vx * vx + vy * vy);
    ec = max_f64(ec, ek * P.mach_floor_scale);
    u[3] += sigma * (ec - eps);
}

// What a row contributes to the faces around it, from the window (rows below, itself, above) and its y-slopes: twice
// its x-slopes, its strain (cbdgam_2d.c:278-293, slopes carried doubled), sqrt(p/sigma), its sound speed
// (cbdgam_2d.c:231-236) and HALF its viscosity nu = alpha h cs at the zone centre (cbdgam_2d.c:644; the factor 1/2
// of the face averages, :651-666, is folded into alpha on the host -- a power of two, the same bits).  The
// bring-up before the march and the march itself both call this for the row below a chunk's first, and a zone's
// bits must not depend on which of the two did: explicit operations only, nothing the compiler could contract one
// way here and another way there.
template <bool FULL>
__device__ __forceinline__ void gam_row_state(const GamParams &P, const double *pa, const double *pb, const double *pc,
                                              const double *gyb, bool visc, bool hvar, double hidx, double hidy,
                                              double dxc0, double dxc1, double oyc0, double oyc1, double *gxb, double &sxx,
                                              double &syy, double &sxy, double &qhb, double &csb, double &nub)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) gxb[q] = plm2<15>(pa[q], pb[q], pc[q]); // 2 x slope
    sxx = 0.0; syy = 0.0; sxy = 0.0;
    if (visc) {
        const double c43 = 4.0 / 3.0, c23 = 2.0 / 3.0;
        const double ax = gxb[1] * hidx, by = gyb[2] * hidy;
        sxx = __builtin_fma(c43, ax, -(c23 * by));
        syy = __builtin_fma(c43, by, -(c23 * ax));
        sxy = __builtin_fma(gxb[2], hidx, gyb[1] * hidy);
    }
    const double prb = pb[3] * fast_rcp(pb[0]);
    qhb = fast_sqrt(prb);
    csb = qhb * P.sqrt_gamma;
    nub = 0.0;
    if (visc) nub = P.half_alpha * (hvar ? qhb * gam_inv_omega_fast<FULL>(P.m, dxc0, dxc1, oyc0, oyc1) : 1.0) * csb;
}

// FULL: the benchmark's physics, established on the host -- alpha viscosity on, both masses positive with
//       active sinks, buffer on, constant softening, no cooling: the uniform tests on those flags drop out of the
//       march (a scalar compare and a branch each, but every one ends a scheduling region: -1.1 % for the last two;
//       same formulas,
//       results equal to the generic instantiation's up to FMA contraction; see iso_stage_kernel).
template <bool WS, bool UN, bool FULL>
#ifndef SFHIP_GAM_WAVES
#define SFHIP_GAM_WAVES 2
#endif
#if defined(SFHIP_DIAG_GAM_TWO_WAVES)  // diagnosis builds only: hold a kernel that would fit three waves per SIMD to two
__attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
__global__ __launch_bounds__(256, SFHIP_GAM_WAVES) void gam_stage_kernel(
    GamParams P, const double *__restrict__ prd, const double *unsrc,
    double *pwr, double *__restrict__ wmax, int nstrips, int nchunks)
{
#pragma clang fp contract(fast)
    const int lane = threadIdx.x & 63;
    // wave-uniform by construction: tell the compiler, so rows, branches and addresses stay scalar
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int strip = gw % nstrips;
    if (gw / nstrips >= nchunks) return;
    const int chunk = P.reverse ? nchunks - 1 - gw / nstrips : gw / nstrips;
    if (P.dt_ptr) P.dt = *P.dt_ptr; // dt computed on the device by sfhip_timestep (uniform scalar load)
    if (P.amax_ptr) P.dt = device_timestep(P.dt_dx, *P.amax_ptr, P.dt_cfl);
    if (gw == 0 && lane == 0) step_bookkeeping(P.zero_ptr, P.host_out, P.host_seq, P.amax_ptr);
    // the parameter block again, through a pointer of hidden origin: loads through it stay inside the
    // rarely taken branches that need them (see sf_iso.hpp)
    const GamParams *Pk = (const GamParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(Pk));

    const int ni = P.ni, nj = P.nj;
    const long sj = 4, si = 4L * (nj + 2 * NG);
    const double gamma = P.gamma;
#if defined(SFHIP_DIAG_NO_VISC)  // diagnosis builds only (WRONG results)
    const bool visc = false;
#else
    const bool visc = FULL || P.alpha > 0.0;
#endif
    const bool hvar = FULL || P.any_mass != 0; // disk height varies in space

    const int J = strip * STRIP + lane;
    int Jld = J;
    if (P.clamp_j) Jld = Jld < NG ? NG : (Jld > nj + NG - 1 ? nj + NG - 1 : Jld);
    else           Jld = Jld > nj + 2 * NG - 1 ? nj + 2 * NG - 1 : Jld;
    const long col = Jld * sj;
    const long ucol = (long)(J < nj + 2 * NG ? J : nj + 2 * NG - 1) * sj;
    const bool store_lane = lane >= HALO && lane < WAVE - HALO && J < nj + NG;
    const int jz = J - NG;
    const double yc = mesh_coord(P.yl, jz + 0.5, P.dy);
    const double yf = mesh_coord(P.yl, (double)jz, P.dy);

    const int I0 = NG + P.i_begin + chunk * P.rows_per_chunk;
    int I1 = I0 + P.rows_per_chunk;
    if (I1 > NG + P.i_end) I1 = NG + P.i_end;
    const int rlo = P.clamp_ilo ? NG : 0;
    const int rhi = P.clamp_ihi ? ni + NG - 1 : ni + 2 * NG - 1;

    // lane-invariant pieces of the distances to the masses
    const double dyc0 = yc - P.m[0].y, dyc1 = yc - P.m[1].y;
    const double oyc0 = dyc0 * dyc0 + 1e-12, oyc1 = dyc1 * dyc1 + 1e-12;                       // omega at centres / x-faces
    const double oyf0 = (yf - P.m[0].y) * (yf - P.m[0].y) + 1e-12, oyf1 = (yf - P.m[1].y) * (yf - P.m[1].y) + 1e-12;
    const double yc2 = yc * yc;
    const double hidx = P.half_inv_dx, hidy = P.half_inv_dy; // slopes are carried doubled
    // 0.25 is not an inline constant on gfx9 (see sf_iso.hpp): keep it as an opaque SGPR operand
    double quarter = 0.25;
    asm volatile("" : "+s"(quarter));

    // dt/dx and dt/dy, wave-uniform: formed once (dt may have come from device memory) and held in SGPRs
    const double kx = uniform_value(P.inv_dx * P.dt), ky = uniform_value(P.inv_dy * P.dt);

    double pa[4], pb[4], pc[4], pn[4], un[4] = {0, 0, 0, 0};
    double gyb[4], gyc[4];
    // carried from row r-2 (the lower side of the next x-face)
    double pmx[4] = {1, 0, 0, 1}, sxxa = 0, sxya = 0, csa = 1, qha = 1, nua = 0;
    // what the row in flight has collected so far of -(dF/dx + dG/dy) dt: its lower x-face flux (as that row
    // sees it) and its y-flux difference; the upper x-face flux follows in the next iteration (see sf_iso.hpp)
    double acc[4] = {0, 0, 0, 0};
    double amax = 0.0, athr = P.ws_floor;

    // The window is filled before the march (rows I0-2, I0-1, I0 and the y-slopes of the last two; nothing
    // here can contract), and all of it has arrived before the loop is entered -- see sf_iso.hpp for this
    // and for the order of the vector-memory traffic inside an iteration (requests first, stores last).
    auto row = [&](int r) { return prd + (r < rlo ? rlo : (r > rhi ? rhi : r)) * si + col; };
    {
        const double *s0 = row(I0 - 2), *s1 = row(I0 - 1), *s2 = row(I0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { pa[q] = s0[q]; pb[q] = s1[q]; pc[q] = s2[q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            { const double d = pb[q] - from_lane_below(pb[q]); gyb[q] = plm2_diffs<15>(d, from_lane_above(d)); }
            { const double d = pc[q] - from_lane_below(pc[q]); gyc[q] = plm2_diffs<15>(d, from_lane_above(d)); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(pa[q]));
    }
    auto enter = [&](const double *row_in) { // a row enters the window: its y-slopes from the neighbouring lanes (2 x slope)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gyb[q] = gyc[q];
            { const double d = row_in[q] - from_lane_below(row_in[q]); gyc[q] = plm2_diffs<15>(d, from_lane_above(d)); }
            pa[q] = pb[q]; pb[q] = pc[q]; pc[q] = row_in[q];
        }
    };
    // Row table (see iso_stage_kernel): the x-dependent, lane-independent quantities of 64 rows in LDS, filled by
    // the lanes once per 60 rows, read by broadcast in the march.
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
    // Bring-up: all that row I0-1 has to provide is the lower side of the x-face between it and row I0 (see
    // sf_iso.hpp: a whole iteration of the march used to be spent on it).
    {
        const double *src = row(I0 + 1);
        pn[0] = src[0]; pn[1] = src[1]; pn[2] = src[2]; pn[3] = src[3];
        double gx[4], sxx, syy, sxy;
        const double xc0 = mesh_coord(P.xl, (I0 - 1 - NG) + 0.5, P.dx); // (the row table's operations: same bits)
        gam_row_state<FULL>(P, pa, pb, pc, gyb, visc, hvar, hidx, hidy, xc0 - P.m[0].x, xc0 - P.m[1].x, oyc0, oyc1,
                            gx, sxx, syy, sxy, qha, csa, nua);
#pragma unroll
        for (int q = 0; q < 4; ++q) pmx[q] = __builtin_fma(quarter, gx[q], pb[q]);
        sxxa = sxx; sxya = sxy;
        enter(pn);
    }

    int refill = I0 + 1; // the iteration that next renews the row table
#pragma unroll 1
    for (int r = I0 + 1; r <= I1 + 1; ++r) {
        if (r == refill) { fill_rows(r); refill += ROW_TABLE_SPAN; }
        // (read here, with the global requests, well before their first use)
        const double *ebp = tab[(r - 1) & (WAVE - 1)]; // row r-1: faces
        const double *eap = tab[(r - 2) & (WAVE - 1)]; // row r-2: the update
        const double eb[4] = {ebp[0], ebp[1], ebp[2], ebp[3]};
        const double ea[5] = {0.0, 0.0, eap[2], eap[3], eap[4]};
        const bool upd = r - 2 >= I0; // the first iteration only brings the pipeline up
        if (UN && upd) {
            // u^n of the row updated at the bottom of this iteration: the whole iteration covers its latency
            const double *us = unsrc + (long)(r - 2) * si + ucol;
            un[0] = us[0]; un[1] = us[1]; un[2] = us[2]; un[3] = us[3];
        }
        { // (the last iteration's request is never consumed: a clamped, valid row; saves a branch)
#if defined(SFHIP_DIAG_SAME_ROW)  // diagnosis builds only: every request hits the cache
            const double *src = row(I0 + (r & 1));
#else
            const double *src = row(r + 1);
#endif
            pn[0] = src[0]; pn[1] = src[1]; pn[2] = src[2]; pn[3] = src[3];
        }
        __builtin_amdgcn_sched_barrier(0); // keep the requests first
#if defined(SFHIP_DIAG_NO_COMPUTE)  // diagnosis builds only (WRONG results): the march's memory traffic without its arithmetic
        {
            double o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[q] = pa[q] + (UN ? un[q] : 0.0);
                pa[q] = pb[q]; pb[q] = pc[q]; pc[q] = pn[q];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (upd && store_lane) {
                const long c = (long)(r - 2) * si + (long)J * sj;
                pwr[c] = o[0]; pwr[c + 1] = o[1]; pwr[c + 2] = o[2]; pwr[c + 3] = o[3];
            }
            continue;
        }
#endif

        {
            // row r-1: slopes, strain, sqrt(p/sigma), sound speed, nu / 2 at the zone centre (all viscosities below
            // are HALF the reference's nu, cbdgam_2d.c:639-648: see gam_row_state)
            double gxb[4], sxx, syy, sxy, qhb, csb, nub;
            gam_row_state<FULL>(P, pa, pb, pc, gyb, visc, hvar, hidx, hidy, eb[2], eb[3], oyc0, oyc1, gxb, sxx, syy, sxy, qhb, csb, nub);

            // Both faces are evaluated in every iteration (the first one's x-face and the last one's y-face go
            // unused), so that the fluxes are plain locals and `acc` is written unconditionally.
            double FR[4], FL[4], tot[4], out[4];
            {
                // x-face between rows r-2 (L) and r-1 (R)
                double ppx[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ppx[q] = pb[q] - quarter * gxb[q];
                gam_hlle2_fast<0>(pmx, ppx, csa, csb, P.inv_gm1, FR, FL);
                if (visc) {
                    double iof = hvar ? gam_inv_omega_fast<FULL>(P.m, eb[0], eb[1], oyc0, oyc1) : 1.0;
                    double nufL = P.half_alpha * (hvar ? qha * iof : 1.0) * csa; // cbdgam_2d.c:639,645
                    double nufR = P.half_alpha * (hvar ? qhb * iof : 1.0) * csb; // cbdgam_2d.c:640,646
                    double txa = pa[0] * sxxa, tya = pa[0] * sxya, txb = pb[0] * sxx, tyb = pb[0] * sxy;
                    double exa = txa * pa[1] + tya * pa[2], exb = txb * pb[1] + tyb * pb[2];
                    // (two fused multiply-adds per component; as `F -= a x + b y` it is a product, an FMA and a subtraction)
                    FR[1] = __builtin_fma(-nub, txb, __builtin_fma(-nufL, txa, FR[1]));
                    FR[2] = __builtin_fma(-nub, tyb, __builtin_fma(-nufL, tya, FR[2]));
                    FR[3] = __builtin_fma(-nub, exb, __builtin_fma(-nufL, exa, FR[3]));
                    FL[1] = __builtin_fma(-nufR, txb, __builtin_fma(-nua, txa, FL[1]));
                    FL[2] = __builtin_fma(-nufR, tyb, __builtin_fma(-nua, tya, FL[2]));
                    FL[3] = __builtin_fma(-nufR, exb, __builtin_fma(-nua, exa, FL[3]));
                }
            }
            // row r-2 is complete with its upper face (as that row sees it): -(dF/dx + dG/dy) dt
#pragma unroll
            for (int q = 0; q < 4; ++q) tot[q] = __builtin_fma(-FL[q], kx, acc[q]);
            if (r <= I1) { // (the row above the chunk's last one is somebody else's: its y-fluxes are not needed here)
                // y-face between lane j-1 (L) and this lane (R), row r-1
                double pmy[4], ppy[4], GR[4], GL[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pmy[q] = from_lane_below(pb[q] + quarter * gyb[q]);
                    ppy[q] = pb[q] - quarter * gyb[q];
                }
                const double qhl = from_lane_below(qhb);
                const double csl = qhl * P.sqrt_gamma; // the neighbour's own product, bit for bit (gam_row_state)
                gam_hlle2_fast<1>(pmy, ppy, csl, csb, P.inv_gm1, GR, GL);
                if (visc) {
                    double txb = pb[0] * sxy, tyb = pb[0] * syy;          // rho tau^y_x, rho tau^y_y of this zone
                    double exb = txb * pb[1] + tyb * pb[2];
                    double txl = from_lane_below(txb), tyl = from_lane_below(tyb), exl = from_lane_below(exb);
                    double nul = from_lane_below(nub);
                    double iof = hvar ? gam_inv_omega_fast<FULL>(P.m, eb[2], eb[3], oyf0, oyf1) : 1.0;
                    double nufL = P.half_alpha * (hvar ? qhl * iof : 1.0) * csl; // cbdgam_2d.c:641,647
                    double nufR = P.half_alpha * (hvar ? qhb * iof : 1.0) * csb; // cbdgam_2d.c:642,648
                    GR[1] = __builtin_fma(-nub, txb, __builtin_fma(-nufL, txl, GR[1]));
                    GR[2] = __builtin_fma(-nub, tyb, __builtin_fma(-nufL, tyl, GR[2]));
                    GR[3] = __builtin_fma(-nub, exb, __builtin_fma(-nufL, exl, GR[3]));
                    GL[1] = __builtin_fma(-nufR, txb, __builtin_fma(-nul, txl, GL[1]));
                    GL[2] = __builtin_fma(-nufR, tyb, __builtin_fma(-nul, tyl, GL[2]));
                    GL[3] = __builtin_fma(-nufR, exb, __builtin_fma(-nul, exl, GL[3]));
                }
                // upper face of this zone is the lower face of lane j+1, seen from below
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_fma(-(from_lane_above(GL[q]) - GR[q]), ky, FR[q] * kx);
            }
            if (upd) {
                // update row r-2 (cbdgam_2d.c:669-681)
                const double xc = ea[4];
                const double rinva = fast_rcp(pa[0]);
                const double eps = pa[3] * rinva * P.inv_gm1;
                double u[4];
                u[0] = pa[0]; u[1] = pa[1] * pa[0]; u[2] = pa[2] * pa[0];
                u[3] = pa[3] * P.inv_gm1 + 0.5 * pa[0] * (pa[1] * pa[1] + pa[2] * pa[2]);
                // disk_height at the zone centre is only consumed by non-constant softening
                double hcc = 1.0;
                if (!FULL && !P.constant_softening && hvar) hcc = qha * gam_inv_omega_fast<FULL>(P.m, ea[2], ea[3], oyc0, oyc1);
#if !defined(SFHIP_DIAG_NO_SOURCES)  // diagnosis builds only (WRONG results): a kernel that fits three waves per SIMD
                gam_buffer_fast<FULL>(P, xc, yc, yc2, u);
#if !defined(SFHIP_DIAG_GAM_MASSES_APART)
                if (FULL) gam_two_masses_fast(P.m, P.sink_near2, Pk->m, ea[2], ea[3], dyc0, dyc1, P.dt, pa, eps, u);
                else
#endif
                {
                gam_mass_source_fast<FULL>(P.m[0], P.sink_near2[0], &Pk->m[0], ea[2], dyc0, P.dt, pa, eps, hcc, u, FULL ? 1 : P.constant_softening);
                gam_mass_source_fast<FULL>(P.m[1], P.sink_near2[1], &Pk->m[1], ea[3], dyc1, P.dt, pa, eps, hcc, u, FULL ? 1 : P.constant_softening);
                }
                gam_cooling_fast<FULL>(P, pa, eps, rinva, u);
#endif
#pragma unroll
                for (int q = 0; q < 4; ++q) u[q] += tot[q];
                if (UN) {
                    double w[4] = {un[0], un[1], un[2], un[3]};
                    if (P.un_mode == 2) { // u^n = p2c(p^n), cbdgam_2d.c:347-362
                        w[3] = un[3] * P.inv_gm1 + 0.5 * un[0] * (un[1] * un[1] + un[2] * un[2]);
                        w[1] = un[1] * un[0];
                        w[2] = un[2] * un[0];
                    }
                    const double b = P.one_minus_a;
#pragma unroll
                    for (int q = 0; q < 4; ++q) u[q] = b * u[q] + P.a * w[q];
                }
                gam_c2p(u, P.gm1, P.vmax, P.dfloor, P.pfloor, out);
                const double rho = out[0], vx = out[1], vy = out[2], pres = out[3];
                if (store_lane) {
                    if (WS) {
                        // same filter as the isothermal kernel: a ~1-ulp estimate decides which zones
                        // get the IEEE evaluation, so the reduced maximum is bit-exact
                        double pre = pres * fast_rcp(rho) * gamma;
                        double cse = fast_sqrt(pre);
                        if (!((max2(fabs(vx), fabs(vy)) + cse) * (1.0 + 1e-9) < athr)) { // unordered passes
                            amax = max_nan(amax, gam_exact_speed(vx, vy, pres, rho, gamma));
                            athr = max2(athr, amax);
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) pmx[q] = __builtin_fma(quarter, gxb[q], pb[q]);
            sxxa = sxx; sxya = sxy; csa = csb; qha = qhb; nua = nub;
            enter(pn); // row r+1 enters the window
            __builtin_amdgcn_sched_barrier(0); // the requested row is in use above this line, stores go below
#if defined(SFHIP_DIAG_NO_STORE)  // diagnosis builds only: what the stores cost
            if (upd && store_lane && out[0] < -1e300) {
#else
            if (upd && store_lane) {
#endif
                const long c = (long)(r - 2) * si + (long)J * sj;
                pwr[c] = out[0]; pwr[c + 1] = out[1]; pwr[c + 2] = out[2]; pwr[c + 3] = out[3];
            }
        }
    }
    if (WS) {
        amax = wave_max_nonneg(amax);
        if (lane == 0) atomic_max_nonneg(wmax, amax);
    }
}

// cbdgam_2d_primitive_to_conserved, cbdgam_2d.c:710-729
__global__ void gam_p2c_kernel(int ni, int nj, double gamma, const double *__restrict__ prim, double *__restrict__ cons)
{
    long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    long i = blockIdx.y;
    if (j >= nj) return;
    long n = ((i + NG) * (nj + 2 * NG) + (j + NG)) * 4;
    double p[4] = {prim[n], prim[n + 1], prim[n + 2], prim[n + 3]}, u[4];
    gam_p2c(p, u, gamma);
    cons[n] = u[0]; cons[n + 1] = u[1]; cons[n + 2] = u[2]; cons[n + 3] = u[3];
}

// cbdgam_2d_wavespeed, cbdgam_2d.c:685-708
__global__ void gam_wavespeed_kernel(int ni, int nj, double gamma, const double *__restrict__ prim,
                                     double *__restrict__ out, double *__restrict__ wmax)
{
    long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    double a = 0.0;
    if (j < nj) {
        const long NJ = nj + 2 * NG;
        // four rows per trip, their loads issued together (see iso_wavespeed_kernel)
        constexpr int U = 4;
        for (long i0 = (long)blockIdx.y * U; i0 < ni; i0 += (long)gridDim.y * U) {
            double p[U][4];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const long i = i0 + k < ni ? i0 + k : ni - 1;
                const long z = (i + NG) * NJ + (j + NG);
#pragma unroll
                for (int q = 0; q < 4; ++q) p[k][q] = prim[z * 4 + q];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const long i = i0 + k;
                if (i < ni) {
                    const long z = (i + NG) * NJ + (j + NG);
                    double c2 = p[k][3] / p[k][0] * gamma;
                    double s = max_speed(p[k][1], p[k][2], c2);
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

// cbdgam_2d_point_mass_source_term, cbdgam_2d.c:731-784 (dt = 1; ASSIGNS the rate of mass `which`)
__global__ void gam_mass_rate_kernel(GamParams P, int which, const double *__restrict__ prim, double *__restrict__ rate)
{
    long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    long i = blockIdx.y;
    if (j >= P.nj) return;
    long n = ((i + NG) * (P.nj + 2 * NG) + (j + NG)) * 4;
    double xc = P.xl + (i + 0.5) * P.dx;
    double yc = P.yl + (j + 0.5) * P.dy;
    double p[4] = {prim[n], prim[n + 1], prim[n + 2], prim[n + 3]};
    double h = gam_height(sqrt(p[3] / p[0]), gam_omega2(P.m, xc, yc));
    double dc[4] = {0.0, 0.0, 0.0, 0.0};
    gam_mass_source(P.m[which], xc, yc, 1.0, p, h, dc, P.constant_softening, P.gamma);
    rate[n] = dc[0]; rate[n + 1] = dc[1]; rate[n + 2] = dc[2]; rate[n + 3] = dc[3];
}

} // namespace sf
