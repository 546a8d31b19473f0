// sf_reduce.hpp -- time-series diagnostics of the circumbinary solvers as one
// fused on-device sum reduction per patch.
//
// Reference: Solver.reductions(), sailfish/solvers/cbdiso_2d.py:370-503 and
// cbdgam_2d.py:341-443.  There, every call launches the point-mass source-term
// kernel four times per patch (mass 1 / 2, gravity-only / accretion-only), each
// into a full-size temporary array, then builds one more full-size temporary
// per diagnostic (radial cut masks, moment arms, complex weights) and sums it.
// Here a single pass over the primitive array evaluates the four source-term
// vectors of a zone in registers (the same device functions the
// *_point_mass_source_term kernels use), forms up to RED_MAXD diagnostic
// integrands from them and accumulates; no temporary array exists.  The sums
// are deterministic: per-thread partial sums over rows, a fixed-shape tree in
// the block, then a second kernel that adds the block partials in index order.
#pragma once
#include "sf_gam.hpp"
#include "sf_iso.hpp"

namespace sf {

constexpr int RED_MAXD = 8;   // diagnostics evaluated per launch
constexpr int RED_SPEC = 8;   // doubles per diagnostic in the spec array
constexpr int RED_SLOTS = 2 * RED_MAXD; // (re, im) per diagnostic

// quantity codes of the spec array (host side: sailfish_amd/solvers/cbd_common.py)
enum {
    Q_FIELD = 0,        // 0..3: rate of conserved field q ("mdot" = 0, "fx" = 1, "fy" = 2)
    Q_TORQUE = 10,
    Q_POWER = 11,
    Q_SIGMA_M1 = 12,
    Q_ECCENTRICITY = 13,
    Q_ANGULAR_MOMENTUM = 14,
    Q_MASS = 15,
};

struct ReduceParams {
    int ni, nj;
    double xl, yl, dx, dy;
    // The reference evaluates the source-term kernel four times: per mass with sink_rate * 0 (gravity = True,
    // accretion = False) and with mass * 0 (accretion = True, gravity = False).  In either call the other
    // part's terms are exact zeros, so one evaluation per mass that keeps the two parts in separate
    // accumulators gives both calls' values (up to the sign of a zero) for half the arithmetic.
    Mass m[2];
    double v[2][2]; // velocities of the two masses (for "power")
    double gamma;
    int constant_softening;
};

// gravity (rg) and accretion (ra) rates of the conserved fields from the two masses
template <int NQ>
__device__ __forceinline__ void zone_rates(const ReduceParams &P, const Mass *m, double xc, double yc,
                                           const double *p, double (*rg)[NQ], double (*ra)[NQ])
{
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int q = 0; q < NQ; ++q) { rg[w][q] = 0.0; ra[w][q] = 0.0; }
    if (NQ == 3) {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            double ddy = yc - m[w].y;
            iso_mass_source(m[w], xc, ddy, ddy * ddy + m[w].rsoft * m[w].rsoft, 1.0, p, rg[w], ra[w]);
        }
    } else {
        double h = gam_height(sqrt(p[3] / p[0]), gam_omega2(m, xc, yc));
#pragma unroll
        for (int w = 0; w < 2; ++w)
            gam_mass_source(m[w], xc, yc, 1.0, p, h, rg[w], P.constant_softening, P.gamma, ra[w]);
    }
}

// grid: (ceil(nj / 256), rows_y); block 256.  partials: [gridDim.y * gridDim.x][RED_SLOTS]
template <int NQ>
__global__ __launch_bounds__(256) void reduce_kernel(ReduceParams P, const double *__restrict__ prim,
                                                     const double *__restrict__ xcoord, const double *__restrict__ ycoord,
                                                     const double *__restrict__ spec, int nd, double *__restrict__ partials)
{
#pragma clang fp contract(off)
    const long j = blockIdx.x * 256L + threadIdx.x;
    double acc[RED_SLOTS];
#pragma unroll
    for (int k = 0; k < RED_SLOTS; ++k) acc[k] = 0.0;

    if (j < P.nj) {
        const long NJ = P.nj + 2 * NG;
        const double y = ycoord[j];                      // moment arm: the Patch's linspace coordinate
        const double yc = P.yl + (j + 0.5) * P.dy;       // source terms: the kernels' own zone centre
        for (long i = blockIdx.y; i < P.ni; i += gridDim.y) {
            const double x = xcoord[i];
            const double xc = P.xl + (i + 0.5) * P.dx;
            const long n = ((i + NG) * NJ + (j + NG)) * NQ;
            double p[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) p[q] = prim[n + q];
            double rg[2][NQ], ra[2][NQ];
            zone_rates<NQ>(P, P.m, xc, yc, p, rg, ra);
            const double r = sqrt(x * x + y * y);
#pragma unroll
            for (int d = 0; d < RED_MAXD; ++d) {
                if (d < nd) {
                    const double *s = spec + d * RED_SPEC;
                    const int quantity = (int)s[0];
                    const int which = (int)s[1];      // 1, 2, or 3 = both
                    const bool accretion = s[3] != 0.0;
                    const bool in_cut = s[4] == 0.0 || (s[5] < r && r < s[6]);
                    // rate of field q from the selected mass(es) and term, radial cut applied (cbdiso_2d.py:462-478)
                    auto field = [&](int q) -> double {
                        double f1 = 0.0, f2 = 0.0;
#pragma unroll
                        for (int k = 0; k < NQ; ++k) {
                            if (k == q) {
                                f1 = accretion ? ra[0][k] : rg[0][k];
                                f2 = accretion ? ra[1][k] : rg[1][k];
                            }
                        }
                        double f = which == 3 ? f1 + f2 : (which == 1 ? f1 : f2);
                        return in_cut ? f : 0.0;
                    };
                    double re = 0.0, im = 0.0;
                    if (quantity < NQ) {
                        re = field(quantity);
                    } else if (quantity == Q_TORQUE) {
                        re = x * field(2) - y * field(1);
                    } else if (quantity == Q_POWER) {
                        const double vx = which == 1 ? P.v[0][0] : P.v[1][0];
                        const double vy = which == 1 ? P.v[0][1] : P.v[1][1];
                        re = vx * field(1) + vy * field(2);
                    } else {
                        const double sigma = in_cut ? p[0] : 0.0;
                        const double vx = in_cut ? p[1] : 0.0;
                        const double vy = in_cut ? p[2] : 0.0;
                        if (quantity == Q_SIGMA_M1) {
                            re = sigma * (x / r);
                            im = sigma * (y / r);
                        } else if (quantity == Q_ECCENTRICITY) {
                            const double vv = vx * vx + vy * vy;
                            const double vr = vx * x + vy * y;
                            const double ex = (vv * x - vr * vx) / 1.0 - x / r;
                            const double ey = (vv * y - vr * vy) / 1.0 - y / r;
                            re = sigma * ex;
                            im = sigma * ey;
                        } else if (quantity == Q_ANGULAR_MOMENTUM) {
                            re = sigma * (x * vy - y * vx);
                        } else if (quantity == Q_MASS) {
                            re = sigma;
                        }
                    }
                    acc[2 * d] += re;
                    acc[2 * d + 1] += im;
                }
            }
        }
    }

    // block tree: lanes (xor shuffle), then the four waves through LDS
    __shared__ double wave_sum[4][RED_SLOTS];
#pragma unroll
    for (int k = 0; k < RED_SLOTS; ++k) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < RED_SLOTS) {
        const int k = threadIdx.x;
        const long block = (long)blockIdx.y * gridDim.x + blockIdx.x;
        partials[block * RED_SLOTS + k] = (wave_sum[0][k] + wave_sum[1][k]) + (wave_sum[2][k] + wave_sum[3][k]);
    }
}

// one block of 256 threads: out[k] = sum over blocks of partials[b][k], fixed order
__global__ __launch_bounds__(256) void reduce_final_kernel(const double *__restrict__ partials, int nblocks, int nslots,
                                                           double *__restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ double part[16][RED_SLOTS];
    const int k = threadIdx.x % RED_SLOTS;
    const int g = threadIdx.x / RED_SLOTS; // 16 groups
    double v = 0.0;
    for (int b = g; b < nblocks; b += 16) v += partials[(long)b * RED_SLOTS + k];
    part[g][k] = v;
    __syncthreads();
    if (threadIdx.x < nslots) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += part[q][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

} // namespace sf
