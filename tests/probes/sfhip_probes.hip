// sfhip_probes.hip -- TEST INFRASTRUCTURE: the stage kernels' device functions, one call per array element.
//
// The marching kernels (sailfish_amd/csrc/sf_iso.hpp, sf_gam.hpp) are checked as a whole against the oracle; this
// library exposes the pieces SURVEY.md section 4 asks known-answer tests for -- the PLM slope, the HLLE flux and
// the primitive recovery with its floors -- by including the SAME headers and calling the SAME inline functions
// from one-thread-per-element kernels (tests/test_function_probes.py).  Not part of the product: nothing under
// sailfish_amd/ loads it, include/sfhip.h does not declare it.  Built in-tree by __graft_entry__.build().
//
// All pointers are device pointers to fp64; every call runs on the null stream and returns after it has finished.
#include <hip/hip_runtime.h>

#include "sf_gam.hpp"
#include "sf_iso.hpp"

namespace {

constexpr int BLOCK = 256;

int finish(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        fprintf(stderr, "sfhip_probes: %s: %s\n", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

template <int THETA10>
__global__ __launch_bounds__(BLOCK) void plm_kernel(int n, const double *yl, const double *y0, const double *yr,
                                                    double *slope, double *twice, double *twice_from_diffs)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    slope[i] = sf::plm<THETA10>(yl[i], y0[i], yr[i]);
    twice[i] = sf::plm2<THETA10>(yl[i], y0[i], yr[i]);
    twice_from_diffs[i] = sf::plm2_diffs<THETA10>(y0[i] - yl[i], yr[i] - y0[i]);
}

// the kernels' own square root of cs2 (iso_sound_fast, gam march): reciprocal square root seed + one correction
__device__ double kernel_sqrt(double x) { return sf::sqrt_from_rsqrt(x, sf::fast_rsqrt(x)); }

template <int DIR>
__global__ __launch_bounds__(BLOCK) void iso_hlle_kernel(int n, const double *pl, const double *pr, const double *cs2, double *flux)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double l[3], r[3], f[3];
    for (int q = 0; q < 3; ++q) { l[q] = pl[3 * i + q]; r[q] = pr[3 * i + q]; }
    sf::iso_hlle<DIR>(l, r, cs2[i], kernel_sqrt(cs2[i]), f);
    for (int q = 0; q < 3; ++q) flux[3 * i + q] = f[q];
}

template <int DIR>
__global__ __launch_bounds__(BLOCK) void gam_hlle2_kernel(int n, const double *pm, const double *pp, const double *cs2_below,
                                                          const double *cs2_above, double inv_gm1, double *flux_for_above,
                                                          double *flux_for_below)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double l[4], r[4], fa[4], fb[4];
    for (int q = 0; q < 4; ++q) { l[q] = pm[4 * i + q]; r[q] = pp[4 * i + q]; }
    sf::gam_hlle2_fast<DIR>(l, r, kernel_sqrt(cs2_below[i]), kernel_sqrt(cs2_above[i]), inv_gm1, fa, fb);
    for (int q = 0; q < 4; ++q) { flux_for_above[4 * i + q] = fa[q]; flux_for_below[4 * i + q] = fb[q]; }
}

__global__ __launch_bounds__(BLOCK) void iso_c2p_kernel(int n, const double *cons, double vmax, double dfloor, double *prim)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double u[3] = {cons[3 * i], cons[3 * i + 1], cons[3 * i + 2]}, p[3];
    sf::iso_c2p(u, vmax, dfloor, p);
    for (int q = 0; q < 3; ++q) prim[3 * i + q] = p[q];
}

__global__ __launch_bounds__(BLOCK) void gam_c2p_kernel(int n, const double *cons, double gm1, double vmax, double dfloor,
                                                        double pfloor, double *prim)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double u[4] = {cons[4 * i], cons[4 * i + 1], cons[4 * i + 2], cons[4 * i + 3]}, p[4];
    sf::gam_c2p(u, gm1, vmax, dfloor, pfloor, p);
    for (int q = 0; q < 4; ++q) prim[4 * i + q] = p[q];
}

int blocks(int n) { return (n + BLOCK - 1) / BLOCK; }

} // namespace

extern "C" {

// theta10: 18 (cbdiso_2d.c:11) or 15 (cbdgam_2d.c:15)
int sfprobe_plm(int n, int theta10, const double *yl, const double *y0, const double *yr, double *slope, double *twice,
                double *twice_from_diffs)
{
    if (n < 1) return 0;
    if (theta10 == 18) plm_kernel<18><<<blocks(n), BLOCK>>>(n, yl, y0, yr, slope, twice, twice_from_diffs);
    else if (theta10 == 15) plm_kernel<15><<<blocks(n), BLOCK>>>(n, yl, y0, yr, slope, twice, twice_from_diffs);
    else return -1;
    return finish("plm");
}

int sfprobe_iso_hlle(int n, int direction, const double *pl, const double *pr, const double *cs2, double *flux)
{
    if (n < 1) return 0;
    if (direction == 0) iso_hlle_kernel<0><<<blocks(n), BLOCK>>>(n, pl, pr, cs2, flux);
    else if (direction == 1) iso_hlle_kernel<1><<<blocks(n), BLOCK>>>(n, pl, pr, cs2, flux);
    else return -1;
    return finish("iso_hlle");
}

// inv_gm1 = 1 / (gamma - 1), formed by the caller as the solver's host code forms it
int sfprobe_gam_hlle2(int n, int direction, const double *pm, const double *pp, const double *cs2_below, const double *cs2_above,
                      double inv_gm1, double *flux_for_above, double *flux_for_below)
{
    if (n < 1) return 0;
    if (direction == 0) gam_hlle2_kernel<0><<<blocks(n), BLOCK>>>(n, pm, pp, cs2_below, cs2_above, inv_gm1, flux_for_above, flux_for_below);
    else if (direction == 1) gam_hlle2_kernel<1><<<blocks(n), BLOCK>>>(n, pm, pp, cs2_below, cs2_above, inv_gm1, flux_for_above, flux_for_below);
    else return -1;
    return finish("gam_hlle2");
}

int sfprobe_iso_c2p(int n, const double *cons, double vmax, double dfloor, double *prim)
{
    if (n < 1) return 0;
    iso_c2p_kernel<<<blocks(n), BLOCK>>>(n, cons, vmax, dfloor, prim);
    return finish("iso_c2p");
}

int sfprobe_gam_c2p(int n, const double *cons, double gm1, double vmax, double dfloor, double pfloor, double *prim)
{
    if (n < 1) return 0;
    gam_c2p_kernel<<<blocks(n), BLOCK>>>(n, cons, gm1, vmax, dfloor, pfloor, prim);
    return finish("gam_c2p");
}

} // extern "C"
