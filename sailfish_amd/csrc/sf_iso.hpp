// sf_iso.hpp -- cbdiso_2d (isothermal circumbinary disk) kernels for gfx950.
//
// Design:
//   * One wave owns a strip of 64 columns (60 outputs + 2 halo lanes per side)
//     and marches down a chunk of rows.  The x-direction stencil (reach 2) lives
//     in a register window that slides with the march; the y-direction stencil
//     is the neighbouring lanes, reached with DPP wave shifts.  No LDS, no
//     barriers; every primitive row is loaded from HBM once per strip.
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
    Mass m[2];                  // point masses at the stage time
    // Keplerian buffer (cbdiso_2d.c:213-246); onset radius and outer orbital
    // frequency depend only on scalars and are evaluated once on the host with
    // the same libm calls the reference makes per zone.
    double buf_sd, buf_mc, buf_rate, buf_rout, buf_ron, buf_omega;
    int buf_on;
    int eos;                    // 1 globally, 2 locally isothermal (cbdiso_2d.c:194-211)
    double cs2, mach2;
    double nu, a, dt, vmax, dfloor;
    // fusion / boundary flags
    int clamp_ilo, clamp_ihi;   // patch touches the domain edge in i: clamp instead of reading guards
    int clamp_j;                // outflow in j by clamping (always true for slab decomposition)
    int un_mode;                // 0: a == 0, no u^n; 1: u^n from a conserved array; 2: u^n = p2c(p^n array)
    int rows_per_chunk;         // rows marched by one wave
    int i_begin, i_end;         // zone-row range [i_begin, i_end) to update (interior/boundary split for halo overlap)
    int want_wavespeed;         // fold max wavespeed of the OUTPUT state into *wmax
    Mass mw[2];                 // masses at the time the output state belongs to (for want_wavespeed)
};

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

// HLLE flux through one face, cbdiso_2d.c:285-380.  DIR 0: x-face, 1: y-face.
template <int DIR>
__device__ __forceinline__ void iso_hlle(const double *pl, const double *pr, double cs2, double *f)
{
    double ul0 = pl[0], ul1 = pl[1] * pl[0], ul2 = pl[2] * pl[0];
    double ur0 = pr[0], ur1 = pr[1] * pr[0], ur2 = pr[2] * pr[0];
    double vl = pl[1 + DIR], vr = pr[1 + DIR];
    double prl = pl[0] * cs2, prr = pr[0] * cs2;
    double fl0 = vl * ul0, fr0 = vr * ur0;
    double fl1 = vl * ul1, fr1 = vr * ur1;
    double fl2 = vl * ul2, fr2 = vr * ur2;
    if (DIR == 0) { fl1 += prl; fr1 += prr; }
    else          { fl2 += prl; fr2 += prr; }
    double cs = sqrt(cs2);
    double am = min2(0.0, min2(vl - cs, vr - cs));
    double ap = max2(0.0, max2(vl + cs, vr + cs));
    double den = ap - am;
    f[0] = (fl0 * ap - fr0 * am - (ul0 - ur0) * ap * am) / den;
    f[1] = (fl1 * ap - fr1 * am - (ul1 - ur1) * ap * am) / den;
    f[2] = (fl2 * ap - fr2 * am - (ul2 - ur2) * ap * am) / den;
}

// One point mass: gravity + sink, cbdiso_2d.c:97-175.  Accumulates into dc.
__device__ __forceinline__ void iso_mass_source(const Mass &m, double x1, double y1, double dt,
                                                const double *prim, double *dc)
{
    double sigma = prim[0];
    double ddx = x1 - m.x;
    double ddy = y1 - m.y;
    double r2 = ddx * ddx + ddy * ddy;
    double dr = sqrt(r2);
    double r2s = r2 + m.rsoft * m.rsoft;
    // pow(r2s, -1.5) of the reference; 1/(r2s*sqrt(r2s)) agrees to ~1 ulp
    double fnum = sigma * m.mass * (1.0 / (r2s * sqrt(r2s)));
    dc[1] += (-fnum * ddx) * dt;
    dc[2] += (-fnum * ddy) * dt;
    if (dr < 4.0 * m.sink_radius) { // lazily evaluated like the C ternary: r_sink == 0 never divides
        double q = dr / m.sink_radius;
        double q2 = q * q;
        double rate = m.sink_rate * exp(-(q2 * q2));
        double mdot = 0.0;
        if (rate > 0.0) mdot = -rate * sigma;
        else if (rate < 0.0) mdot = -rate;
        if (m.sink_model == 1) {
            dc[0] += dt * mdot;
            dc[1] += dt * mdot * prim[1];
            dc[2] += dt * mdot * prim[2];
        } else if (m.sink_model == 2) {
            double rhx = ddx / (dr + 1e-12);
            double rhy = ddy / (dr + 1e-12);
            double dv = (prim[1] - m.vx) * rhx + (prim[2] - m.vy) * rhy;
            double vxs = dv * rhx + m.vx;
            double vys = dv * rhy + m.vy;
            dc[0] += dt * mdot;
            dc[1] += dt * mdot * vxs;
            dc[2] += dt * mdot * vys;
        } else if (m.sink_model == 3) {
            dc[0] += dt * mdot;
        }
    }
}

// Keplerian buffer, cbdiso_2d.c:213-246
__device__ __forceinline__ void iso_buffer(const IsoParams &P, double xc, double yc, const double *u, double *dc)
{
    if (!P.buf_on) return;
    double rc = sqrt(xc * xc + yc * yc);
    if (rc > P.buf_ron) {
        double vk = sqrt(P.buf_mc / rc);
        double u0[3] = {P.buf_sd, P.buf_sd * (-yc / rc) * vk, P.buf_sd * (+xc / rc) * vk};
        double br = P.buf_rate * P.buf_omega * (rc - P.buf_ron) / (P.buf_rout - P.buf_ron);
#pragma unroll
        for (int q = 0; q < 3; ++q) dc[q] -= (u[q] - u0[q]) * br * P.dt;
    }
}

// ---------------------------------------------------------------------------
// One Runge-Kutta stage (cbdiso_2d_advance_rk, cbdiso_2d.c:385-604).
// grid: 1-D, 256 threads = 4 waves; wave w of the grid owns column strip
// (w % nstrips) and row chunk (w / nstrips).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iso_stage_kernel(
    IsoParams P, const double *__restrict__ prd, const double *unsrc,
    double *pwr, double *__restrict__ wmax, int nstrips, int nchunks)
{
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int strip = gw % nstrips;
    const int chunk = gw / nstrips;
    if (chunk >= nchunks) return; // whole wave leaves together

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
    const double yc = P.yl + (jz + 0.5) * P.dy;
    const double yf = P.yl + (double)jz * P.dy; // lower y-face of this lane's zone

    // rows (guard-inclusive index) updated by this wave
    const int I0 = NG + P.i_begin + chunk * P.rows_per_chunk;
    int I1 = I0 + P.rows_per_chunk;
    if (I1 > NG + P.i_end) I1 = NG + P.i_end;
    const int rlo = P.clamp_ilo ? NG : 0;
    const int rhi = P.clamp_ihi ? ni + NG - 1 : ni + 2 * NG - 1;

    double pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0}, pc[3];
    double gyb[3] = {0, 0, 0};
    double pmx[3] = {0, 0, 0}, txx = 0, txy = 0;
    double Fp[3] = {0, 0, 0}, dGp[3] = {0, 0, 0};
    double amax = 0.0;

    for (int r = I0 - 2; r <= I1 + 1; ++r) {
        int rl = r < rlo ? rlo : (r > rhi ? rhi : r);
        const double *src = prd + rl * si + col;
        pc[0] = src[0]; pc[1] = src[1]; pc[2] = src[2];

        // y-slopes of row r from the neighbouring lanes
        double gyc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            gyc[q] = plm<18>(from_lane_below(pc[q]), pc[q], from_lane_above(pc[q]));

        if (r >= I0) {
            // row r-1 now has both neighbours: x-slopes and strain (cbdiso_2d.c:248-263)
            double gxb[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) gxb[q] = plm<18>(pa[q], pb[q], pc[q]);
            double sxx = 4.0 / 3.0 * gxb[1] / P.dx - 2.0 / 3.0 * gyb[2] / P.dy;
            double syy = -2.0 / 3.0 * gxb[1] / P.dx + 4.0 / 3.0 * gyb[2] / P.dy;
            double sxy = gxb[2] / P.dx + gyb[1] / P.dy;

            double F[3] = {0, 0, 0}, dG[3] = {0, 0, 0};
            const double xf = P.xl + (double)(r - 1 - NG) * P.dx;       // lower x-face of row r-1
            const double xcb = P.xl + ((r - 1 - NG) + 0.5) * P.dx;      // centre of row r-1

            if (r >= I0 + 1) {
                // x-face between rows r-2 and r-1 (cbdiso_2d.c:536-541, 560-561, 579-582)
                double ppx[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) ppx[q] = pb[q] - 0.5 * gxb[q];
                iso_hlle<0>(pmx, ppx, iso_cs2(P, P.m, xf, yc), F);
                if (P.nu > 0.0) {
                    F[1] -= 0.5 * P.nu * (txx + pb[0] * sxx);
                    F[2] -= 0.5 * P.nu * (txy + pb[0] * sxy);
                }
            }
            if (r - 1 >= I0 && r - 1 < I1) {
                // y-face below this lane's zone in row r-1 (cbdiso_2d.c:543-546, 562-563, 583-586)
                double pmy[3], ppy[3], G[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    pmy[q] = from_lane_below(pb[q] + 0.5 * gyb[q]);
                    ppy[q] = pb[q] - 0.5 * gyb[q];
                }
                iso_hlle<1>(pmy, ppy, iso_cs2(P, P.m, xcb, yf), G);
                if (P.nu > 0.0) {
                    double tyx = pb[0] * sxy, tyy = pb[0] * syy;
                    G[1] -= 0.5 * P.nu * (from_lane_below(tyx) + tyx);
                    G[2] -= 0.5 * P.nu * (from_lane_below(tyy) + tyy);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) dG[q] = from_lane_above(G[q]) - G[q];
            }
            if (r - 2 >= I0) {
                // update row r-2 (cbdiso_2d.c:588-602)
                const double xc = P.xl + ((r - 2 - NG) + 0.5) * P.dx;
                double u[3] = {pa[0], pa[1] * pa[0], pa[2] * pa[0]};
                double dc[3] = {0.0, 0.0, 0.0};
                iso_buffer(P, xc, yc, u, dc);
                iso_mass_source(P.m[0], xc, yc, P.dt, pa, dc);
                iso_mass_source(P.m[1], xc, yc, P.dt, pa, dc);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    dc[q] -= ((F[q] - Fp[q]) / P.dx + dGp[q] / P.dy) * P.dt;
                    u[q] += dc[q];
                }
                const long c = (long)(r - 2) * si + (long)J * sj;
                if (P.un_mode && store_lane) {
                    double un0 = unsrc[c], un1 = unsrc[c + 1], un2 = unsrc[c + 2];
                    if (P.un_mode == 2) { un1 *= un0; un2 *= un0; } // u^n = p2c(p^n), cbdiso_2d.c:285-298
                    u[0] = (1.0 - P.a) * u[0] + P.a * un0;
                    u[1] = (1.0 - P.a) * u[1] + P.a * un1;
                    u[2] = (1.0 - P.a) * u[2] + P.a * un2;
                }
                // primitive recovery with floors (cbdiso_2d.c:268-283)
                double rho = max2(u[0], P.dfloor);
                double vx = copysign(1.0, u[1]) * min2(fabs(u[1] / rho), P.vmax);
                double vy = copysign(1.0, u[2]) * min2(fabs(u[2] / rho), P.vmax);
                if (store_lane) {
                    pwr[c] = rho; pwr[c + 1] = vx; pwr[c + 2] = vy;
                    if (P.want_wavespeed) {
                        double c2 = iso_cs2(P, P.mw, xc, yc);
                        amax = max2(amax, max_speed(vx, vy, c2));
                    }
                }
            }
            // slide the window: row r-1 becomes the lower side of the next x-face
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                pmx[q] = pb[q] + 0.5 * gxb[q];
                Fp[q] = F[q];
                dGp[q] = dG[q];
            }
            txx = pb[0] * sxx;
            txy = pb[0] * sxy;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { gyb[q] = gyc[q]; pa[q] = pb[q]; pb[q] = pc[q]; }
    }
    if (P.want_wavespeed) {
        amax = wave_max_nonneg(amax);
        if (lane == 0) atomic_max_nonneg(wmax, amax);
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
        for (long i = blockIdx.y; i < P.ni; i += gridDim.y) {
            long z = (i + NG) * NJ + (j + NG);
            double x = P.xl + (i + 0.5) * P.dx;
            double c2 = iso_cs2(P, P.m, x, y);
            double s = max_speed(prim[z * 3 + 1], prim[z * 3 + 2], c2);
            if (out) out[z] = s;
            a = max2(a, s);
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
    iso_mass_source(P.m[0], xc, yc, 1.0, p, dc);
    rate[n] = dc[0]; rate[n + 1] = dc[1]; rate[n + 2] = dc[2];
}

} // namespace sf
