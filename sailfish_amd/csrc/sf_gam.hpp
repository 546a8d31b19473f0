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
    double xl, yl, dx, dy;
    Mass m[2];
    double gamma;
    double buf_sd, buf_sp, buf_mc, buf_rate, buf_rout, buf_ron, buf_omega;
    int buf_on;
    int constant_softening;
    double alpha, a, dt, vmax, cooling, mach_ceiling_m2 /* pow(mach_ceiling,-2) */, dfloor, pfloor;
    int clamp_ilo, clamp_ihi, clamp_j, un_mode;
    int rows_per_chunk, i_begin, i_end, want_wavespeed;
    double ws_floor; // accepted for symmetry with cbdiso; the gamma-law speed is cheap, every zone is folded
};

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

// Two HLLE fluxes through one face (see header comment).  cbdgam_2d.c:374-434.
template <int DIR>
__device__ __forceinline__ void gam_hlle2(const double *pm, const double *pp, double cs2L, double cs2R,
                                          double gamma, double *fR, double *fL)
{
    double ul[4], ur[4], fl[4], fr[4];
    gam_p2c(pm, ul, gamma);
    gam_p2c(pp, ur, gamma);
    double vl = pm[1 + DIR], vr = pp[1 + DIR];
    fl[0] = vl * ul[0]; fl[1] = vl * ul[1]; fl[2] = vl * ul[2]; fl[3] = vl * (ul[3] + pm[3]);
    fr[0] = vr * ur[0]; fr[1] = vr * ur[1]; fr[2] = vr * ur[2]; fr[3] = vr * (ur[3] + pp[3]);
    fl[1 + DIR] += pm[3];
    fr[1 + DIR] += pp[3];
    {
        double cs = sqrt(cs2L);
        double am = min2(0.0, min2(vl - cs, vr - cs));
        double ap = max2(0.0, max2(vl + cs, vr + cs));
        double den = ap - am;
#pragma unroll
        for (int q = 0; q < 4; ++q) fR[q] = (fl[q] * ap - fr[q] * am - (ul[q] - ur[q]) * ap * am) / den;
    }
    {
        double cs = sqrt(cs2R);
        double am = min2(0.0, min2(vl - cs, vr - cs));
        double ap = max2(0.0, max2(vl + cs, vr + cs));
        double den = ap - am;
#pragma unroll
        for (int q = 0; q < 4; ++q) fL[q] = (fl[q] * ap - fr[q] * am - (ul[q] - ur[q]) * ap * am) / den;
    }
}

// One point mass, cbdgam_2d.c:108-203.  Adds its contribution to u.
__device__ __forceinline__ void gam_mass_source(const Mass &m, double x1, double y1, double dt, const double *prim,
                                                double h, double *u, int constant_softening, double gamma)
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

// cbdgam_2d.c:238-276
__device__ __forceinline__ void gam_buffer(const GamParams &P, double xc, double yc, double *u)
{
    if (!P.buf_on) return;
    double rc = sqrt(xc * xc + yc * yc);
    if (rc > P.buf_ron) {
        double pf = P.buf_sd * sqrt(P.buf_mc / rc);
        double px = pf * (-yc / rc);
        double py = pf * (xc / rc);
        double ke = 0.5 * (px * px + py * py) / P.buf_sd;
        double en = P.buf_sp / (P.gamma - 1.0) + ke;
        double u0[4] = {P.buf_sd, px, py, en};
        double br = P.buf_rate * P.buf_omega * (rc - P.buf_ron) / (P.buf_rout - P.buf_ron);
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] -= (u[q] - u0[q]) * br * P.dt;
    }
}

// cbdgam_2d.c:298-317
__device__ __forceinline__ void gam_cooling(const GamParams &P, const double *prim, double *u)
{
    double gamma = P.gamma;
    double sigma = prim[0];
    double eps = prim[3] / prim[0] / (gamma - 1.0);
    double ec = eps;
    if (P.cooling != 0.0) {
        double arg = 1.0 + 3.0 * P.cooling * (1.0 / (sigma * sigma)) * (eps * eps * eps) * P.dt;
        ec = eps * pow(arg, -1.0 / 3.0);
    }
    double vx = prim[1], vy = prim[2];
    double ek = 0.5 * (vx * vx + vy * vy);
    ec = max2(ec, 2.0 * ek / gamma / (gamma - 1.0) * P.mach_ceiling_m2);
    u[3] += sigma * (ec - eps);
}

__global__ __launch_bounds__(256) void gam_stage_kernel(
    GamParams P, const double *__restrict__ prd, const double *unsrc,
    double *pwr, double *__restrict__ wmax, int nstrips, int nchunks)
{
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int strip = gw % nstrips;
    const int chunk = gw / nstrips;
    if (chunk >= nchunks) return;

    const int ni = P.ni, nj = P.nj;
    const long sj = 4, si = 4L * (nj + 2 * NG);
    const double gamma = P.gamma;
    const bool visc = P.alpha > 0.0;

    const int J = strip * STRIP + lane;
    int Jld = J;
    if (P.clamp_j) Jld = Jld < NG ? NG : (Jld > nj + NG - 1 ? nj + NG - 1 : Jld);
    else           Jld = Jld > nj + 2 * NG - 1 ? nj + 2 * NG - 1 : Jld;
    const long col = Jld * sj;
    const bool store_lane = lane >= HALO && lane < WAVE - HALO && J < nj + NG;
    const int jz = J - NG;
    const double yc = P.yl + (jz + 0.5) * P.dy;
    const double yf = P.yl + (double)jz * P.dy;

    const int I0 = NG + P.i_begin + chunk * P.rows_per_chunk;
    int I1 = I0 + P.rows_per_chunk;
    if (I1 > NG + P.i_end) I1 = NG + P.i_end;
    const int rlo = P.clamp_ilo ? NG : 0;
    const int rhi = P.clamp_ihi ? ni + NG - 1 : ni + 2 * NG - 1;

    double pa[4] = {1, 0, 0, 1}, pb[4] = {1, 0, 0, 1}, pc[4];
    double gyb[4] = {0, 0, 0, 0};
    // carried from row r-2 (the lower side of the next x-face)
    double pmx[4] = {1, 0, 0, 1}, sxxa = 0, sxya = 0, c2a = 1, qha = 1, nua = 0;
    double Fp[4] = {0, 0, 0, 0};  // lower-face flux of row r-2 as that row sees it
    double dGp[4] = {0, 0, 0, 0}; // (upper - lower) y-flux difference of row r-2
    double amax = 0.0;

    for (int r = I0 - 2; r <= I1 + 1; ++r) {
        int rl = r < rlo ? rlo : (r > rhi ? rhi : r);
        const double *src = prd + rl * si + col;
        pc[0] = src[0]; pc[1] = src[1]; pc[2] = src[2]; pc[3] = src[3];

        double gyc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            gyc[q] = plm<15>(from_lane_below(pc[q]), pc[q], from_lane_above(pc[q]));

        if (r >= I0) {
            double gxb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gxb[q] = plm<15>(pa[q], pb[q], pc[q]);
            double sxx = 4.0 / 3.0 * gxb[1] / P.dx - 2.0 / 3.0 * gyb[2] / P.dy;
            double syy = -2.0 / 3.0 * gxb[1] / P.dx + 4.0 / 3.0 * gyb[2] / P.dy;
            double sxy = gxb[2] / P.dx + gyb[1] / P.dy;

            const double xf = P.xl + (double)(r - 1 - NG) * P.dx;
            const double xcb = P.xl + ((r - 1 - NG) + 0.5) * P.dx;
            const double c2b = pb[3] / pb[0] * gamma;              // cbdgam_2d.c:231-236
            double qhb = 0, nub = 0;
            if (visc) {
                qhb = sqrt(pb[3] / pb[0]);
                nub = P.alpha * gam_height(qhb, gam_omega2(P.m, xcb, yc)) * sqrt(c2b); // cbdgam_2d.c:644
            }

            double FR[4] = {0, 0, 0, 0}, FL[4] = {0, 0, 0, 0}, dG[4] = {0, 0, 0, 0};

            if (r >= I0 + 1) {
                // x-face between rows r-2 (L) and r-1 (R)
                double ppx[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ppx[q] = pb[q] - 0.5 * gxb[q];
                gam_hlle2<0>(pmx, ppx, c2a, c2b, gamma, FR, FL);
                if (visc) {
                    double om2f = gam_omega2(P.m, xf, yc);
                    double nufL = P.alpha * gam_height(qha, om2f) * sqrt(c2a); // cbdgam_2d.c:639,645
                    double nufR = P.alpha * gam_height(qhb, om2f) * sqrt(c2b); // cbdgam_2d.c:640,646
                    FR[1] -= 0.5 * (nufL * pa[0] * sxxa + nub * pb[0] * sxx);
                    FR[2] -= 0.5 * (nufL * pa[0] * sxya + nub * pb[0] * sxy);
                    FR[3] -= 0.5 * (nufL * pa[0] * sxxa * pa[1] + nub * pb[0] * sxx * pb[1]);
                    FR[3] -= 0.5 * (nufL * pa[0] * sxya * pa[2] + nub * pb[0] * sxy * pb[2]);
                    FL[1] -= 0.5 * (nua * pa[0] * sxxa + nufR * pb[0] * sxx);
                    FL[2] -= 0.5 * (nua * pa[0] * sxya + nufR * pb[0] * sxy);
                    FL[3] -= 0.5 * (nua * pa[0] * sxxa * pa[1] + nufR * pb[0] * sxx * pb[1]);
                    FL[3] -= 0.5 * (nua * pa[0] * sxya * pa[2] + nufR * pb[0] * sxy * pb[2]);
                }
            }
            if (r - 1 >= I0 && r - 1 < I1) {
                // y-face between lane j-1 (L) and this lane (R), row r-1
                double pmy[4], ppy[4], GR[4], GL[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pmy[q] = from_lane_below(pb[q] + 0.5 * gyb[q]);
                    ppy[q] = pb[q] - 0.5 * gyb[q];
                }
                double c2l = from_lane_below(c2b);
                gam_hlle2<1>(pmy, ppy, c2l, c2b, gamma, GR, GL);
                if (visc) {
                    double rhol = from_lane_below(pb[0]);
                    double vxl = from_lane_below(pb[1]);
                    double vyl = from_lane_below(pb[2]);
                    double syxl = from_lane_below(sxy);
                    double syyl = from_lane_below(syy);
                    double qhl = from_lane_below(qhb);
                    double nul = from_lane_below(nub);
                    double om2f = gam_omega2(P.m, xcb, yf);
                    double nufL = P.alpha * gam_height(qhl, om2f) * sqrt(c2l); // cbdgam_2d.c:641,647
                    double nufR = P.alpha * gam_height(qhb, om2f) * sqrt(c2b); // cbdgam_2d.c:642,648
                    GR[1] -= 0.5 * (nufL * rhol * syxl + nub * pb[0] * sxy);
                    GR[2] -= 0.5 * (nufL * rhol * syyl + nub * pb[0] * syy);
                    GR[3] -= 0.5 * (nufL * rhol * syxl * vxl + nub * pb[0] * sxy * pb[1]);
                    GR[3] -= 0.5 * (nufL * rhol * syyl * vyl + nub * pb[0] * syy * pb[2]);
                    GL[1] -= 0.5 * (nul * rhol * syxl + nufR * pb[0] * sxy);
                    GL[2] -= 0.5 * (nul * rhol * syyl + nufR * pb[0] * syy);
                    GL[3] -= 0.5 * (nul * rhol * syxl * vxl + nufR * pb[0] * sxy * pb[1]);
                    GL[3] -= 0.5 * (nul * rhol * syyl * vyl + nufR * pb[0] * syy * pb[2]);
                }
                // upper face of this zone is the lower face of lane j+1, seen from below
#pragma unroll
                for (int q = 0; q < 4; ++q) dG[q] = from_lane_above(GL[q]) - GR[q];
            }
            if (r - 2 >= I0) {
                // update row r-2 (cbdgam_2d.c:669-681)
                const double xc = P.xl + ((r - 2 - NG) + 0.5) * P.dx;
                double u[4];
                gam_p2c(pa, u, gamma);
                // disk_height at the zone centre is only consumed by non-constant softening
                double hcc = 1.0;
                if (!P.constant_softening) hcc = gam_height(sqrt(pa[3] / pa[0]), gam_omega2(P.m, xc, yc));
                gam_buffer(P, xc, yc, u);
                gam_mass_source(P.m[0], xc, yc, P.dt, pa, hcc, u, P.constant_softening, gamma);
                gam_mass_source(P.m[1], xc, yc, P.dt, pa, hcc, u, P.constant_softening, gamma);
                gam_cooling(P, pa, u);
#pragma unroll
                for (int q = 0; q < 4; ++q) u[q] -= ((FL[q] - Fp[q]) / P.dx + dGp[q] / P.dy) * P.dt;
                const long c = (long)(r - 2) * si + (long)J * sj;
                if (P.un_mode && store_lane) {
                    double un[4] = {unsrc[c], unsrc[c + 1], unsrc[c + 2], unsrc[c + 3]};
                    if (P.un_mode == 2) { double t[4]; gam_p2c(un, t, gamma); un[0] = t[0]; un[1] = t[1]; un[2] = t[2]; un[3] = t[3]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) u[q] = (1.0 - P.a) * u[q] + P.a * un[q];
                }
                // cbdgam_2d.c:319-345
                double pres = max2(P.pfloor, (u[3] - 0.5 * (u[1] * u[1] + u[2] * u[2]) / u[0]) * (gamma - 1.0));
                double vx = copysign(1.0, u[1]) * min2(fabs(u[1] / u[0]), P.vmax);
                double vy = copysign(1.0, u[2]) * min2(fabs(u[2] / u[0]), P.vmax);
                double rho = u[0];
                if (u[0] < P.dfloor) { rho = P.dfloor; vx = 0.0; vy = 0.0; pres = P.pfloor; }
                if (store_lane) {
                    pwr[c] = rho; pwr[c + 1] = vx; pwr[c + 2] = vy; pwr[c + 3] = pres;
                    if (P.want_wavespeed) amax = max2(amax, max_speed(vx, vy, pres / rho * gamma));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pmx[q] = pb[q] + 0.5 * gxb[q];
                Fp[q] = FR[q];
                dGp[q] = dG[q];
            }
            sxxa = sxx; sxya = sxy; c2a = c2b; qha = qhb; nua = nub;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { gyb[q] = gyc[q]; pa[q] = pb[q]; pb[q] = pc[q]; }
    }
    if (P.want_wavespeed) {
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
        for (long i = blockIdx.y; i < ni; i += gridDim.y) {
            long z = (i + NG) * NJ + (j + NG);
            double c2 = prim[z * 4 + 3] / prim[z * 4] * gamma;
            double s = max_speed(prim[z * 4 + 1], prim[z * 4 + 2], c2);
            if (out) out[z] = s;
            a = max2(a, s);
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
