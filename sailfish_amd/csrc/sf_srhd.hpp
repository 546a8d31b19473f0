// sf_srhd.hpp -- srhd_2d kernels for gfx950: 2-D special-relativistic hydrodynamics on a
// log-spherical, homologously expanding mesh (reference: sailfish/solvers/srhd_2d.c).
//
// The four reference-named kernels.  p2c, c2p and the wavespeed are one
// thread per zone with threadIdx.x along j (the contiguous axis; the reference's gpu mode puts
// threadIdx.x on i, library.py:75-78, i.e. fully uncoalesced), AoS rows read as 32-byte per-lane
// loads; advance_rk is a marching-wave kernel (srhd_march_kernel below: slopes once per zone, one
// HLLC per face, optional fused primitive recovery), its thread-per-zone form kept for A/B timing.
// All arithmetic is IEEE fp64 in the reference's operation order
// (contraction off, correctly rounded divide and sqrt), so results are bit-identical to the C
// path wherever no transcendental is involved.  The only transcendentals on the path are
//   * sin/cos of the polar face angles: taken from a table the host fills with libm (one
//     entry per polar face), because a last-place difference in cos(q1) - cos(q0) near the
//     pole is a 1e-10 relative difference in the zone volume;
//   * exp/pow in the jet inflow profile (innermost row only) -- device libm, within 1e-12.
// State arrays are (ni + 4, nj, 4): guard zones in i only; j is clamped (srhd_2d.c:666-669).
#pragma once
#include "sf_device.hpp"

namespace sf {

constexpr double SR_GAMMA = 4.0 / 3.0;       // srhd_2d.c:14
constexpr double SR_PI = 3.141592653589793;  // srhd_2d.c:15

struct SrhdParams {
    int ni, nj;
    double polar_extent, dq;
    double theta;          // PLM_THETA (srhd_2d.c:25-27)
    double mach_ceiling2;  // pow(MACH_CEILING, 2.0), evaluated on the host with libm (srhd_2d.c:156)
    double scale_factor;   // p2c / c2p
    double a0, adot, time, rk_param, dt;
    double jet_mdot, jet_gamma_beta, jet_theta, jet_duration;
    int num_first_order_zones;
    // fused stage (sfhip_srhd_2d_stage): the primitive recovery that the reference runs at the start of the
    // NEXT stage is done here, on the zone just updated, while it is still in registers
    int fuse_c2p;
    double scale_factor_next;      // scale factor at the time after this stage
    double *prim_wr;               // (ni + 4, nj, 4) primitives of the updated state (interior rows)
    double *cons_floored;          // (ni + 4, nj, 4) updated state after the Mach-ceiling floor
    unsigned int *failures;
    const double *faces;   // (ni + 1) comoving radial face positions
    const double *sinq;    // (nj + 1) sin of the polar face angles dq * j, from the host's libm
    const double *cosq;    // (nj + 1)
};

__device__ __forceinline__ double sr_slope(double yl, double y0, double yr, double theta) // srhd_2d.c:40-46
{
#pragma clang fp contract(off)
    double a = (y0 - yl) * theta;
    double b = (yr - yl) * 0.5;
    double c = (yr - y0) * theta;
    double sa = copysign(1.0, a), sb = copysign(1.0, b), sc = copysign(1.0, c);
    double m = min2(fabs(a), min2(fabs(b), fabs(c)));
    return 0.25 * fabs(sa + sb) * (sa + sc) * m;
}

__device__ __forceinline__ double sr_uu(const double *p) { return p[1] * p[1] + p[2] * p[2]; }
// the square roots and single divisions below go through exact_sqrt_* / exact_quot (sf_device.hpp): the
// compiler's own iterations without its range handling, bit-identical for the magnitudes of a valid state
__device__ __forceinline__ double sr_lorentz(const double *p) { return exact_sqrt_pos(1.0 + sr_uu(p)); }
__device__ __forceinline__ double sr_rho_h(const double *p) { return p[0] + p[3] * (1.0 + 1.0 / (SR_GAMMA - 1.0)); }

// srhd_2d.c:90-107
__device__ __forceinline__ void sr_p2c(const double *p, double *u, double dv)
{
#pragma clang fp contract(off)
    double w = sr_lorentz(p);
    double h = exact_quot(sr_rho_h(p), p[0]);
    double m = p[0] * w;
    u[0] = dv * m;
    u[1] = dv * m * h * p[1];
    u[2] = dv * m * h * p[2];
    u[3] = dv * m * (h * w - 1.0) - dv * p[3];
}

// srhd_2d.c:109-185; returns nonzero where the reference would print and exit
__device__ __forceinline__ int sr_c2p(const double *u1, double *u2, double *p, double dv, double mach_ceiling2)
{
#pragma clang fp contract(off)
    const int iter_max = 500;
    const ExactRcp rdv = exact_rcp(dv); // shared denominators: same bits as `/`, see sf_device.hpp
    const double tol = exact_div(1e-12 * (u1[0] + u1[3]), rdv);
    const double m = exact_div(u1[0], rdv);
    const double tau = exact_div(u1[3], rdv);
    const double s1 = exact_div(u1[1], rdv);
    const double s2 = exact_div(u1[2], rdv);
    const double ss = s1 * s1 + s2 * s2;
    double pr = p[3];
    double w0, f;
    int it = 0;
    for (;;) {
        double et = tau + pr + m;
        const ExactRcp ret = exact_rcp(et);
        double b2 = min2(exact_div(exact_div(ss, ret), ret), 1.0 - 1e-10);
        double w2 = exact_quot(1.0, 1.0 - b2);
        double w = exact_sqrt_pos(w2);
        double e = exact_quot(tau + m * (1.0 - w) + pr * (1.0 - w2), m * w);
        double d = exact_quot(m, w);
        double h = 1.0 + e + exact_quot(pr, d);
        double a2 = exact_quot(SR_GAMMA * pr, d * h);
        double g = b2 * a2 - 1.0;
        f = d * e * (SR_GAMMA - 1.0) - pr;
        pr -= exact_quot(f, g);
        if (fabs(f) < tol || it == iter_max) {
            w0 = w;
            break;
        }
        it += 1;
    }
    p[0] = exact_quot(m, w0);
    const ExactRcp ren = exact_rcp(tau + m + pr);
    p[1] = exact_div(w0 * s1, ren);
    p[2] = exact_div(w0 * s2, ren);
    p[3] = pr;
    double uu = p[1] * p[1] + p[2] * p[2];
    double e = exact_quot(p[3], p[0]) * 3.0;
    double emin = exact_quot(exact_quot(uu, 1.0 + uu), mach_ceiling2);
    if (e < emin) {
        p[3] = p[0] * emin * (SR_GAMMA - 1.0);
        sr_p2c(p, u2, dv);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) u2[q] = u1[q];
    }
    if (it == iter_max) return 1;
    if (u1[3] <= 0.0) return 2;
    if (p[3] <= 0.0 || p[3] != p[3]) return 3;
    return 0;
}

// srhd_2d.c:201-220; DIR 1 radial, 2 polar
template <int DIR>
__device__ __forceinline__ void sr_outer_speeds(const double *p, double &lm, double &lp)
{
#pragma clang fp contract(off)
    double a2 = exact_quot(SR_GAMMA * p[3], sr_rho_h(p));
    double uu = sr_uu(p);
    double vn = exact_quot(p[DIR], sr_lorentz(p));
    double vv = exact_quot(uu, 1.0 + uu);
    double v2 = vn * vn;
    double k0 = exact_sqrt0(a2 * (1.0 - vv) * (1.0 - vv * a2 - v2 * (1.0 - a2)));
    const ExactRcp rden = exact_rcp(1.0 - vv * a2);
    lm = exact_div(vn * (1.0 - a2) - k0, rden);
    lp = exact_div(vn * (1.0 - a2) + k0, rden);
}

// srhd_2d.c:299-451
template <int DIR>
__device__ __forceinline__ void sr_hllc(const double *pl, const double *pr, double vface, double *flux)
{
#pragma clang fp contract(off)
    double ul[4], ur[4], fl[4], fr[4];
    sr_p2c(pl, ul, 1.0);
    sr_p2c(pr, ur, 1.0);
    const double vl = exact_quot(pl[DIR], sr_lorentz(pl)), vr = exact_quot(pr[DIR], sr_lorentz(pr));
    fl[0] = vl * ul[0]; fr[0] = vr * ur[0];
    fl[1] = vl * ul[1] + pl[3] * (DIR == 1); fr[1] = vr * ur[1] + pr[3] * (DIR == 1);
    fl[2] = vl * ul[2] + pl[3] * (DIR == 2); fr[2] = vr * ur[2] + pr[3] * (DIR == 2);
    fl[3] = vl * ul[3] + pl[3] * vl; fr[3] = vr * ur[3] + pr[3] * vr;
    double alm, alp, arm, arp;
    sr_outer_speeds<DIR>(pl, alm, alp);
    sr_outer_speeds<DIR>(pr, arm, arp);
    const double am = min2(alm, arm);
    const double ap = max2(alp, arp);
    if (vface < am) {
#pragma unroll
        for (int q = 0; q < 4; ++q) flux[q] = fl[q] - vface * ul[q];
        return;
    }
    if (vface > ap) {
#pragma unroll
        for (int q = 0; q < 4; ++q) flux[q] = fr[q] - vface * ur[q];
        return;
    }
    double uh[4], fh[4];
    const ExactRcp rspan = exact_rcp(ap - am);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uh[q] = exact_div(ur[q] * ap - ul[q] * am + (fl[q] - fr[q]), rspan);
        fh[q] = exact_div(fl[q] * ap - fr[q] * am - (ul[q] - ur[q]) * ap * am, rspan);
    }
    const double un = uh[DIR], fn = fh[DIR];
    const double a = fh[3] + fh[0];
    const double b = -(uh[3] + uh[0] + fn);
    const double c = un;
    double vs;
    if (fabs(a) < 1e-10) vs = exact_quot(-c, b);
    else vs = exact_quot(-b - exact_sqrt0(b * b - 4.0 * a * c), 2.0 * a);
    const double ps = -a * vs + fn;
    const bool left = vface < vs;
    constexpr int TR = DIR == 1 ? 2 : 1;
    const double vk = left ? vl : vr;
    const double uk0 = left ? ul[0] : ur[0], ukn = left ? ul[DIR] : ur[DIR], ukt = left ? ul[TR] : ur[TR],
                 uk3 = left ? ul[3] : ur[3];
    const ExactRcp rstar = exact_rcp(am - vs);
    const double D = exact_div(uk0 * (am - vk), rstar);
    const double E = exact_div(am * (uk3 + uk0) - ukn + ps * vs, rstar);
    const double Sn = (E + ps) * vs;
    const double St = exact_div(ukt * (am - vk), rstar);
    const double S1 = DIR == 1 ? Sn : St, S2 = DIR == 1 ? St : Sn;
    const double taus = E - D;
    flux[0] = D * vs - vface * D;
    flux[1] = S1 * vs + ps * (DIR == 1) - vface * S1;
    flux[2] = S2 * vs + ps * (DIR == 2) - vface * S2;
    flux[3] = Sn - D * vs - vface * taus;
}

// face_area with the trigonometry from the table (srhd_2d.c:457-466)
__device__ __forceinline__ double sr_area(double r0, double r1, double s0, double c0, double s1, double c1)
{
#pragma clang fp contract(off)
    double R0 = r0 * s0, R1 = r1 * s1;
    double z0 = r0 * c0, z1 = r1 * c1;
    double dR = R1 - R0, dz = z1 - z0;
    return SR_PI * (R0 + R1) * exact_sqrt0(dR * dR + dz * dz);
}

__device__ __forceinline__ double sr_volume(double r0, double r1, double c0, double c1) // srhd_2d.c:468-471
{
#pragma clang fp contract(off)
    return exact_quot(-(r1 * r1 * r1 - r0 * r0 * r0) * (c1 - c0) * 2.0 * SR_PI, 3.0);
}

// grid (ceil(nj/256), ni)
__global__ void srhd_p2c_kernel(SrhdParams P, const double *__restrict__ prim, double *__restrict__ cons)
{
#pragma clang fp contract(off)
    const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long i = blockIdx.y;
    if (j >= P.nj) return;
    const long n = (i * P.nj + j) * 4;
    const double r0 = P.faces[i] * P.scale_factor, r1 = P.faces[i + 1] * P.scale_factor;
    double p[4] = {prim[n], prim[n + 1], prim[n + 2], prim[n + 3]}, u[4];
    sr_p2c(p, u, sr_volume(r0, r1, P.cosq[j], P.cosq[j + 1]));
    cons[n] = u[0]; cons[n + 1] = u[1]; cons[n + 2] = u[2]; cons[n + 3] = u[3];
}

__global__ void srhd_c2p_kernel(SrhdParams P, const double *__restrict__ cons1, double *__restrict__ cons2,
                                double *__restrict__ prim, unsigned int *__restrict__ failures)
{
#pragma clang fp contract(off)
    const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long i = blockIdx.y;
    if (j >= P.nj) return;
    const long n = ((i + NG) * P.nj + j) * 4;
    const double r0 = P.faces[i] * P.scale_factor, r1 = P.faces[i + 1] * P.scale_factor;
    double u1[4] = {cons1[n], cons1[n + 1], cons1[n + 2], cons1[n + 3]}, u2[4];
    double p[4] = {0.0, 0.0, 0.0, prim[n + 3]};
    if (sr_c2p(u1, u2, p, sr_volume(r0, r1, P.cosq[j], P.cosq[j + 1]), P.mach_ceiling2)) atomicAdd(failures, 1u);
    prim[n] = p[0]; prim[n + 1] = p[1]; prim[n + 2] = p[2]; prim[n + 3] = p[3];
    cons2[n] = u2[0]; cons2[n + 1] = u2[1]; cons2[n + 2] = u2[2]; cons2[n + 3] = u2[3];
}

__global__ void srhd_wavespeed_kernel(SrhdParams P, const double *__restrict__ prim, double *__restrict__ wavespeed)
{
#pragma clang fp contract(off)
    const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long i = blockIdx.y;
    if (j >= P.nj) return;
    const long n = ((i + NG) * P.nj + j) * 4;
    const double p[4] = {prim[n], prim[n + 1], prim[n + 2], prim[n + 3]};
    const double beta = 0.5 * (P.faces[i] + P.faces[i + 1]) * P.adot;
    const double gw = 1.0 / sqrt(1.0 - beta * beta); // primitive_with_radial_boost, srhd_2d.c:222-233
    const double u0 = sr_lorentz(p);
    const double pb[4] = {p[0], p[1] * gw - u0 * gw * beta, p[2], p[3]};
    double a0, a1, b0, b1;
    sr_outer_speeds<1>(pb, a0, a1);
    sr_outer_speeds<2>(pb, b0, b1);
    wavespeed[i * P.nj + j] = max2(max2(fabs(a0), fabs(a1)), max2(fabs(b0), fabs(b1)));
}

__global__ __launch_bounds__(256) void srhd_advance_kernel(SrhdParams P, const double *__restrict__ cons_rk,
                                                           const double *__restrict__ prim_rd,
                                                           const double *__restrict__ cons_rd,
                                                           double *__restrict__ cons_wr)
{
#pragma clang fp contract(off)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    const int nj = P.nj;
    if (j >= nj) return;
    const long si = 4L * nj;
    const double x0 = P.faces[i], x1 = P.faces[i + 1];
    const double r0 = x0 * (P.a0 + P.adot * P.time), r1 = x1 * (P.a0 + P.adot * P.time);
    const double q0 = P.dq * (j + 0), q1 = P.dq * (j + 1);
    const double qc = 0.5 * (q0 + q1);
    const double s0 = P.sinq[j], s1 = P.sinq[j + 1], c0 = P.cosq[j], c1 = P.cosq[j + 1];
    const long c = (long)(i + NG) * si + (long)j * 4;

    if (i == 0 && P.jet_mdot > 0.0 && qc < P.jet_theta * 2.0 && P.time < 1.0 + P.jet_duration) {
        // jet inflow zone, srhd_2d.c:642-651
        double ju = P.jet_gamma_beta;
        double jrho = P.jet_mdot / (4.0 * SR_PI * r0 * r0 * ju);
        double prof = exp(-pow(qc / P.jet_theta, 2.0));
        double pj[4] = {jrho, ju * prof, 0.0, 1e-6 * jrho}, u[4];
        sr_p2c(pj, u, sr_volume(r0, r1, c0, c1));
        cons_wr[c] = u[0]; cons_wr[c + 1] = u[1]; cons_wr[c + 2] = u[2]; cons_wr[c + 3] = u[3];
        return;
    }
    if (P.jet_mdot > 0.0 && i == 0) return; // innermost zone held fixed, srhd_2d.c:652-655

    const int jl = j - 1 > 0 ? j - 1 : 0, jr = j + 1 < nj - 1 ? j + 1 : nj - 1;
    const int jk = j - 2 > 0 ? j - 2 : 0, jt = j + 2 < nj - 1 ? j + 2 : nj - 1;
    const double *row = prim_rd + (long)(i + NG) * si;
    double pcc[4], pli[4], pri[4], pki[4], pti[4], plj[4], prj[4], pkj[4], ptj[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        pcc[q] = row[(long)j * 4 + q];
        pli[q] = row[(long)j * 4 + q - si];
        pri[q] = row[(long)j * 4 + q + si];
        pki[q] = row[(long)j * 4 + q - 2 * si];
        pti[q] = row[(long)j * 4 + q + 2 * si];
        plj[q] = row[(long)jl * 4 + q];
        prj[q] = row[(long)jr * 4 + q];
        pkj[q] = row[(long)jk * 4 + q];
        ptj[q] = row[(long)jt * 4 + q];
    }
    const bool second_order = i >= P.num_first_order_zones;
    double lim[4], lip[4], rim[4], rip[4], ljm[4], ljp[4], rjm[4], rjp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double gli = 0.0, gci = 0.0, gri = 0.0, glj = 0.0, gcj = 0.0, grj = 0.0;
        if (second_order) {
            gli = sr_slope(pki[q], pli[q], pcc[q], P.theta);
            gci = sr_slope(pli[q], pcc[q], pri[q], P.theta);
            gri = sr_slope(pcc[q], pri[q], pti[q], P.theta);
            glj = sr_slope(pkj[q], plj[q], pcc[q], P.theta);
            gcj = sr_slope(plj[q], pcc[q], prj[q], P.theta);
            grj = sr_slope(pcc[q], prj[q], ptj[q], P.theta);
        }
        lim[q] = pli[q] + 0.5 * gli;
        lip[q] = pcc[q] - 0.5 * gci;
        rim[q] = pcc[q] + 0.5 * gci;
        rip[q] = pri[q] - 0.5 * gri;
        ljm[q] = plj[q] + 0.5 * glj;
        ljp[q] = pcc[q] - 0.5 * gcj;
        rjm[q] = pcc[q] + 0.5 * gcj;
        rjp[q] = prj[q] - 0.5 * grj;
    }
    const double a_r0 = sr_area(r0, r0, s0, c0, s1, c1), a_r1 = sr_area(r1, r1, s0, c0, s1, c1);
    const double a_q0 = sr_area(r0, r1, s0, c0, s0, c0), a_q1 = sr_area(r0, r1, s1, c1, s1, c1);
    double fli[4], fri[4], flj[4], frj[4];
    sr_hllc<1>(lim, lip, x0 * P.adot, fli);
    sr_hllc<1>(rim, rip, x1 * P.adot, fri);
    sr_hllc<2>(ljm, ljp, 0.0, flj);
    sr_hllc<2>(rjm, rjp, 0.0, frj);
    // geometric_source_terms, srhd_2d.c:473-499 (u_phi = 0)
    const double ur = pcc[1], uq = pcc[2], up = 0.0, pg = pcc[3];
    const double rhoh = sr_rho_h(pcc);
    const double dcosq = c1 - c0, dsinq = s1 - s0, dr2 = r1 * r1 - r0 * r0;
    double src[4];
    src[0] = 0.0;
    src[1] = -SR_PI * dr2 * dcosq * (rhoh * (uq * uq + up * up) + 2 * pg);
    src[2] = +SR_PI * dr2 * (dcosq * rhoh * ur * uq + dsinq * (pg + rhoh * up * up));
    src[3] = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = cons_rd[c + q] + (fli[q] * a_r0 - fri[q] * a_r1 + flj[q] * a_q0 - frj[q] * a_q1 + src[q]) * P.dt;
        cons_wr[c + q] = (1.0 - P.rk_param) * v + P.rk_param * cons_rk[c + q];
    }
}


// ---------------------------------------------------------------------------
// srhd_2d_advance_rk as marching waves (the design of the cbd stage kernels, sf_iso.hpp): one wave
// owns 64 columns in j (60 outputs + 2 halo lanes per side, columns clamped to [0, nj) exactly as
// the reference clamps its j-neighbours, srhd_2d.c:666-669) and marches down a chunk of rows.  Slopes
// are evaluated once per zone and each HLLC flux once per face -- the thread-per-zone kernel above, like
// the reference, evaluates every face twice -- with the same IEEE operations, so the result is still
// bit-identical.  One subtlety: `num_first_order_zones` zeroes the slopes by the index of the zone that
// LOOKS at a face (srhd_2d.c:694-702), so the face between rows nfo-1 and nfo is seen differently from
// its two sides; that one row of faces is evaluated twice here as well.
// grid: 1-D, 256 threads = 4 waves; wave w owns strip (w % nstrips) and row chunk (w / nstrips).
// ---------------------------------------------------------------------------
#if !defined(SFHIP_SRHD_WAVES)
#define SFHIP_SRHD_WAVES 2
#endif
template <bool V> struct SrTag { static constexpr bool value = V; };

// FO: the run has first-order zones (num_first_order_zones > 0).  Without them the tests on the viewing zone's index
//     (uniform per row, but selects on every slope and state all the same) and the second view of one row of faces
//     drop out of the march: the same operations on the same operands, so the same bits.
template <bool FO>
__global__ __launch_bounds__(256, SFHIP_SRHD_WAVES) void srhd_march_kernel(SrhdParams P, const double *__restrict__ cons_rk,
                                                            const double *__restrict__ prim_rd,
                                                            const double *__restrict__ cons_rd,
                                                            double *__restrict__ cons_wr,
                                                            int rows_per_chunk, int nstrips, int nchunks)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int strip = gw % nstrips;
    const int chunk = gw / nstrips;
    if (chunk >= nchunks) return;

    const int ni = P.ni, nj = P.nj;
    const long si = 4L * nj;
    const int Jraw = strip * STRIP + lane - HALO;                   // column this lane stands for
    const int J = Jraw < 0 ? 0 : (Jraw > nj - 1 ? nj - 1 : Jraw);   // where it loads from (polar clamp)
    const bool store_lane = lane >= HALO && lane < WAVE - HALO && Jraw < nj;
    const long col = (long)J * 4;
    // Lanes standing for columns outside [0, nj) hold copies of the edge zone's state (the reference's
    // clamped neighbours), but their faces are where they stand: the lane for column nj has its lower
    // face at the polar angle of face nj -- the edge zone's upper face, whose flux it supplies.
    const double q0 = P.dq * (J + 0), q1 = P.dq * (J + 1);
    const double qc = 0.5 * (q0 + q1);
    const int f0 = Jraw < 0 ? 0 : (Jraw > nj ? nj : Jraw), f1 = Jraw + 1 < 0 ? 0 : (Jraw + 1 > nj ? nj : Jraw + 1);
    const double s0 = P.sinq[f0], s1 = P.sinq[f1], c0 = P.cosq[f0], c1 = P.cosq[f1];
    const double dcosq = c1 - c0, dsinq = s1 - s0;
    const double sfac = P.a0 + P.adot * P.time;
    const bool jet_on = P.jet_mdot > 0.0;
    const int nfo = FO ? P.num_first_order_zones : 0;

    // zone rows [i0, i1) of this chunk; array rows are zone rows + NG
    const int i0 = chunk * rows_per_chunk;
    int i1 = i0 + rows_per_chunk;
    if (i1 > ni) i1 = ni;

    double pa[4] = {1, 0, 0, 1}, pb[4] = {1, 0, 0, 1}, pc[4], pn[4];
    double gyb[4] = {0, 0, 0, 0};       // y-slopes of row r-1 (already zeroed where first order)
    double gxa[4] = {0, 0, 0, 0};       // x-slopes of row r-2, NOT zeroed (who looks decides, see above)
    double Fp[4] = {0, 0, 0, 0};        // flux through the lower face of row r-2, as that row sees it
    double GLp[4] = {0, 0, 0, 0};       // flj * da_q0 of row r-2
    double GUp[4] = {0, 0, 0, 0};       // frj * da_q1 of row r-2
    {
        const double *src = prim_rd + (long)(i0 - 2 + NG) * si + col;
        pn[0] = src[0]; pn[1] = src[1]; pn[2] = src[2]; pn[3] = src[3];
    }
    // r = zone row whose primitives arrive this iteration; row r-1 gets its x-slopes and faces,
    // row r-2 is updated
    // One body for every row; STEADY (a tag type) marks the iterations in which all of its row tests are known to
    // be true -- rows i0 + 2 ... i1 of the chunk -- so that they drop out there: the two Riemann problems of an
    // iteration (x-face and polar face, independent of each other, each a long chain of IEEE divisions and square
    // roots) and the update then lie in one scheduling region and interleave.  Contraction is off and every
    // operation is IEEE: the two instantiations of the body give the same bits.
    auto body = [&](const int r, auto steady_tag) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_tag)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) pc[q] = pn[q];
        if (STEADY || r < i1 + 1) {
            const double *src = prim_rd + (long)(r + 1 + NG) * si + col;
            pn[0] = src[0]; pn[1] = src[1]; pn[2] = src[2]; pn[3] = src[3];
        }
        // the conserved state (and u^n) of the row updated in this iteration: requested here, in flight during the
        // Riemann problems
        // (at the point of use they cost their latency in every row: 0.945 -> 0.930 ms per launch at 4800 x 3274)
        double ur_[4] = {0, 0, 0, 0}, uk_[4] = {0, 0, 0, 0};
        if (r - 2 >= i0 && store_lane) {
            const long c = (long)(r - 2 + NG) * si + (long)J * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) { ur_[q] = cons_rd[c + q]; uk_[q] = cons_rk[c + q]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // y-slopes of row r (neighbouring lanes hold the clamped neighbours)
        double gyc[4];
        const bool so_c = !FO || r >= nfo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double g = sr_slope(from_lane_below(pc[q]), pc[q], from_lane_above(pc[q]), P.theta);
            gyc[q] = so_c ? g : 0.0;
        }
        if (STEADY || r >= i0) {
            const int ib = r - 1;       // zone row of pb
            const bool so_b = !FO || ib >= nfo;
            double gxb[4];              // x-slopes of row r-1, not zeroed
#pragma unroll
            for (int q = 0; q < 4; ++q) gxb[q] = sr_slope(pa[q], pb[q], pc[q], P.theta);

            const double xlo = P.faces[ib < 0 ? 0 : (ib > ni ? ni : ib)];   // lower face of row r-1 (clamped: unused when out of range)
            const double rlo = xlo * sfac;
            double Fup[4] = {0, 0, 0, 0};   // face between rows r-2 and r-1 as row r-2 sees it (its upper face)
            double Flo[4] = {0, 0, 0, 0};   // ... as row r-1 sees it (its lower face)
            if (STEADY || r >= i0 + 1) {
                // The left state uses row r-2's slopes, the right state row r-1's; whether they count is
                // decided by the viewing zone (see header).  The two views differ only when ib == nfo.
                double pl[4], pr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pl[q] = pa[q] + 0.5 * (so_b ? gxa[q] : 0.0);
                    pr[q] = pb[q] - 0.5 * (so_b ? gxb[q] : 0.0);
                }
                sr_hllc<1>(pl, pr, xlo * P.adot, Flo);
                if (FO && ib == nfo && ib > 0) {
                    // row r-2 < nfo looks at this face without slopes (p +- 0.5 * 0, literally)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        pl[q] = pa[q] + 0.5 * 0.0;
                        pr[q] = pb[q] - 0.5 * 0.0;
                    }
                    sr_hllc<1>(pl, pr, xlo * P.adot, Fup);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) Fup[q] = Flo[q];
                }
            }
            double GL[4] = {0, 0, 0, 0}, GU[4] = {0, 0, 0, 0};
            if (STEADY || (ib >= i0 && ib < i1)) {
                // polar face below this lane's zone in row r-1; radii of the row
                const double rhi = P.faces[ib + 1] * sfac;
                double pl[4], pr[4], G[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pl[q] = from_lane_below(pb[q] + 0.5 * gyb[q]);
                    pr[q] = pb[q] - 0.5 * gyb[q];
                }
                sr_hllc<2>(pl, pr, 0.0, G);
                const double a_q0 = sr_area(rlo, rhi, s0, c0, s0, c0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    GL[q] = G[q] * a_q0;              // flj * da_q0
                    GU[q] = from_lane_above(GL[q]);   // frj * da_q1: the upper face is the lower face of lane j+1
                }
            }
            const int iu = r - 2;       // zone row updated now (pa)
            if (STEADY || iu >= i0) {
                const double x0 = P.faces[iu], x1 = P.faces[iu + 1];
                const double r0 = x0 * sfac, r1 = x1 * sfac;
                const long c = (long)(iu + NG) * si + (long)J * 4;
                if (iu == 0 && jet_on) {
                    // srhd_2d.c:642-655: jet inflow zone, or the innermost zone held fixed
                    if (qc < P.jet_theta * 2.0 && P.time < 1.0 + P.jet_duration && store_lane) {
                        double ju = P.jet_gamma_beta;
                        double jrho = P.jet_mdot / (4.0 * SR_PI * r0 * r0 * ju);
                        double prof = exp(-pow(qc / P.jet_theta, 2.0));
                        double pj[4] = {jrho, ju * prof, 0.0, 1e-6 * jrho}, u[4];
                        sr_p2c(pj, u, sr_volume(r0, r1, c0, c1));
                        cons_wr[c] = u[0]; cons_wr[c + 1] = u[1]; cons_wr[c + 2] = u[2]; cons_wr[c + 3] = u[3];
                    }
                } else {
                    const double a_r0 = sr_area(r0, r0, s0, c0, s1, c1), a_r1 = sr_area(r1, r1, s0, c0, s1, c1);
                    // geometric_source_terms, srhd_2d.c:473-499 (u_phi = 0)
                    const double ur = pa[1], uq = pa[2], up = 0.0, pg = pa[3];
                    const double rhoh = sr_rho_h(pa);
                    const double dr2 = r1 * r1 - r0 * r0;
                    double src[4];
                    src[0] = 0.0;
                    src[1] = -SR_PI * dr2 * dcosq * (rhoh * (uq * uq + up * up) + 2 * pg);
                    src[2] = +SR_PI * dr2 * (dcosq * rhoh * ur * uq + dsinq * (pg + rhoh * up * up));
                    src[3] = 0.0;
                    if (store_lane) {
                        double un[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            // fli*da_r0 - fri*da_r1 + flj*da_q0 - frj*da_q1 + sources, in the reference's order
                            double v = ur_[q] + (Fp[q] * a_r0 - Fup[q] * a_r1 + GLp[q] - GUp[q] + src[q]) * P.dt;
                            un[q] = (1.0 - P.rk_param) * v + P.rk_param * uk_[q];
                            cons_wr[c + q] = un[q];
                        }
                        if (P.fuse_c2p) {
                            // srhd_2d_conserved_to_primitive of the next stage (srhd_2d.c:544-575): guess = this
                            // zone's current pressure, volume at the next stage's scale factor
                            double uf[4], pw[4] = {0.0, 0.0, 0.0, pa[3]};
                            const double dvn = sr_volume(x0 * P.scale_factor_next, x1 * P.scale_factor_next, c0, c1);
                            if (sr_c2p(un, uf, pw, dvn, P.mach_ceiling2)) atomicAdd(P.failures, 1u);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                P.prim_wr[c + q] = pw[q];
                                P.cons_floored[c + q] = uf[q];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gxa[q] = gxb[q];
                Fp[q] = Flo[q];
                GLp[q] = GL[q];
                GUp[q] = GU[q];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { gyb[q] = gyc[q]; pa[q] = pb[q]; pb[q] = pc[q]; }
    };
    int r = i0 - 2;
    for (; r <= i1 + 1 && r < i0 + 2; ++r) body(r, SrTag<false>{});   // the window fills
#if !defined(SFHIP_SRHD_ONE_BODY)
    for (; r <= i1; ++r) body(r, SrTag<true>{});
#endif
    for (; r <= i1 + 1; ++r) body(r, SrTag<false>{});                   // the last row's update
}

} // namespace sf
