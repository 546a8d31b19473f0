// sfhip.hip -- C-ABI of libsfhip.so: hand-written gfx950 kernels for sailfish's
// cbdiso_2d / cbdgam_2d finite-volume update.  Declarations, with the reference
// interface each entry point replaces, are in include/sfhip.h.
//
// Conventions
//   * every pointer is a DEVICE pointer on the current device;
//   * launches go to the calling thread's current stream (sfhip_set_stream); nothing here
//     synchronises or allocates, with two exceptions: the first srhd_2d call on a device or
//     for a new (nj, polar_extent) builds a small table (see srhd_state), and
//     sfhip_srhd_2d_failures reads a counter back;
//   * entry points return 0 on success, a hipError_t value (or -1 for argument
//     errors) otherwise; sfhip_last_error() describes the last failure.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <sched.h>
#include <thread>
#include <vector>

#include "sf_dg.hpp"
#include "sf_gam.hpp"
#include "sf_iso.hpp"
#include "sf_reduce.hpp"
#include "sf_setup.hpp"
#include "sf_srhd.hpp"
#include "sfhip.h"

namespace {

thread_local hipStream_t t_stream = nullptr;
thread_local char t_error[512] = "";

int fail(int code, const char *what)
{
    snprintf(t_error, sizeof(t_error), "%s", what);
    return code;
}

// Launch errors surface here; a fault INSIDE a kernel is asynchronous and would show up at some later, unrelated
// synchronisation without a name.  SFHIP_SYNC_LAUNCHES=1 (debugging) waits for the device after every launch, so
// that such a fault is reported by the entry point that caused it, with the kernel's name.
int check_launch(const char *what)
{
    static const bool sync_launches = [] { const char *s = getenv("SFHIP_SYNC_LAUNCHES"); return s && *s && *s != '0'; }();
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && sync_launches) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        snprintf(t_error, sizeof(t_error), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

int env_int(const char *name, int fallback)
{
    const char *s = getenv(name);
    return s && *s ? atoi(s) : fallback;
}

// Split `nrows` rows into chunks, one wave per (chunk, strip).
// Large patches: about 8192 waves (8 workgroups of 4 waves per CU; cbdgam and srhd, at two waves per SIMD: 9216, i.e.
// 32-row chunks at 4096^2, 1 % faster there and a wash for cbdiso) and at least 32 rows per chunk, so that the
// 4-row pipeline warm-up stays small.  On the device the stage time is flat (+-2 %) between 28 and 47 rows per
// chunk at 4096^2, as for the 1448 x 11584 patch of an 8-rank run (tools/sweep_chunks.sh).
// Patches that cannot reach that many waves with 32-row chunks are latency-bound, and what a launch takes is
// what its slowest SIMD takes: (rows per chunk + 2 rows' worth of start-up) x a factor for the number of waves
// that share a SIMD -- 1, 1.28 and 1.97 for one, two and three (measured; the third adds no throughput), with
// another such round for what does not fit three resident waves.  The chunk length is the one that minimises this estimate; it
// reproduces the measured optima of the sustained RK2 pair (tools/kbench.py): 3 rows at 512^2 (0.024 ms against
// 0.038 ms with round 1's 8-row minimum), 10 at 1024^2 (0.050 against 0.060), ~20 at 2048^2, and the flat 6..20
// of the 512 x 4096 slab of an 8-rank strong-scaling run.
void plan_rows(int nrows, int nstrips, int *rows_per_chunk, int *nchunks, int default_target_waves = 8192)
{
    int rpc = env_int("SFHIP_ROWS_PER_CHUNK", 0);
    if (rpc <= 0) {
        const int target_waves = env_int("SFHIP_TARGET_WAVES", default_target_waves);
        const int max_nc = (nrows + 31) / 32;
        if ((long)max_nc * nstrips >= target_waves) {
            int nc = (target_waves + nstrips - 1) / nstrips;
            if (nc > max_nc) nc = max_nc;
            if (nc < 1) nc = 1;
            rpc = (nrows + nc - 1) / nc;
        } else {
            double best = 0.0;
            for (int r = 3; r <= 32; ++r) {
                const double waves_per_simd = (double)((nrows + r - 1) / r) * nstrips / 1024.0;
                const int full_rounds = (int)(waves_per_simd / 3.0);      // three waves are resident per SIMD
                const double rest = waves_per_simd - 3.0 * full_rounds;   // what the busiest SIMD still has to run then
                const double sharing = 1.97 * full_rounds + (rest <= 0.0 ? 0.0 : rest <= 1.0 ? 1.0 : rest <= 2.0 ? 1.28 : 1.97);
                const double cost = sharing * (r + 2);
                if (rpc <= 0 || cost < best) { best = cost; rpc = r; }
            }
            const int nc = (nrows + rpc - 1) / rpc;
            rpc = (nrows + nc - 1) / nc;   // equal chunks
        }
    }
    if (rpc < 1) rpc = 1;
    *rows_per_chunk = rpc;
    *nchunks = (nrows + rpc - 1) / rpc;
}

// Stages with a != 0 hand their chunks out from the last rows to the first.  A stage streams 0.8-1.2 GB through the
// 256 MB memory-side cache, which therefore ends a stage holding the rows it touched LAST; the stage that follows
// reads the same arrays, and by starting where its predecessor stopped it finds those rows still cached (and its
// successor, marching up again, finds what it left).  Same chunks, same bits.  Measured at 4096^2 (bench.py, A/B on
// one box): cbdiso_2d 0.544 -> 0.533 ms per step, cbdgam_2d 0.937 -> 0.920; no effect at 2048^2 (everything is
// cached either way) and within noise at 8192^2 (the cache holds 8 % of an array).  SFHIP_REVERSE=0 turns it off.
bool reverse_later_stages()
{
    static const bool on = env_int("SFHIP_REVERSE", 1) != 0;
    return on;
}

// waves per workgroup of the marching kernels (their launch bounds allow up to 4)
int waves_per_block()
{
    const int w = env_int("SFHIP_WAVES_PER_BLOCK", 4);
    return w == 1 || w == 2 ? w : 4;
}

sf::Mass make_mass(double x, double y, double vx, double vy, double mass, double rsoft, double rate, double rsink, int model)
{
    sf::Mass m;
    m.x = x; m.y = y; m.vx = vx; m.vy = vy; m.mass = mass; m.rsoft = rsoft;
    m.sink_rate = rate; m.sink_radius = rsink; m.sink_model = model; m.pad_ = 0;
    return m;
}

int launch_iso_stage(sf::IsoParams &P, const double *prd, const double *unsrc, double *pwr, double *wmax)
{
    if (P.ni < 1 || P.nj < 1) return fail(-1, "cbdiso_2d stage: ni, nj must be >= 1");
    if (!prd || !pwr) return fail(-1, "cbdiso_2d stage: null primitive array");
    if (prd == pwr) return fail(-1, "cbdiso_2d stage: primitive_rd and primitive_wr must not alias");
    if (P.un_mode && !unsrc) return fail(-1, "cbdiso_2d stage: a != 0 needs the u^n source array");
    if (P.want_wavespeed && !wmax) return fail(-1, "cbdiso_2d stage: want_wavespeed needs wmax");
    if (P.i_begin < 0 || P.i_end > P.ni || P.i_begin > P.i_end) return fail(-1, "cbdiso_2d stage: bad row range");
    if (P.want_wavespeed == 2) {
        hipError_t e = hipMemsetAsync(wmax, 0, sizeof(double), t_stream);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    const int nrows = P.i_end - P.i_begin;
    if (P.host_out && !P.amax_ptr) return fail(-1, "stage: the host hand-over needs amax_device");
    if (nrows == 0) {
        if (P.zero_ptr || P.host_out) { // the step protocol's bookkeeping must happen even when no row is updated
            hipLaunchKernelGGL(sf::step_bookkeeping_kernel, dim3(1), dim3(64), 0, t_stream, P.zero_ptr, P.host_out,
                               P.host_seq, P.amax_ptr);
            return check_launch("step_bookkeeping_kernel");
        }
        return 0;
    }
    const int nstrips = (P.nj + sf::STRIP - 1) / sf::STRIP;
    int nchunks;
    plan_rows(nrows, nstrips, &P.rows_per_chunk, &nchunks);
    const long waves = (long)nstrips * nchunks;
    const int wpb = waves_per_block();
    dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(64 * wpb);
    P.reverse = P.un_mode != 0 && reverse_later_stages();
    // epilogue masses identical to the stage's (RK2's last stage): the kernel reuses the source terms' distances
    bool same = P.want_wavespeed != 0;
    for (int k = 0; k < 2 && same; ++k)
        same = memcmp(&P.m[k].x, &P.mw[k].x, sizeof(double)) == 0 && memcmp(&P.m[k].y, &P.mw[k].y, sizeof(double)) == 0 &&
               memcmp(&P.m[k].mass, &P.mw[k].mass, sizeof(double)) == 0 &&
               memcmp(&P.m[k].rsoft, &P.mw[k].rsoft, sizeof(double)) == 0;
    // the full-physics instantiations (see iso_stage_kernel): same formulas, flag tests resolved here
    const bool full = P.nu > 0.0 && P.eos_local && P.m[0].mass > 0.0 && P.m[1].mass > 0.0 && P.buf_on &&
                      !env_int("SFHIP_GENERIC_STAGE", 0);
    const int ws = !P.want_wavespeed ? 0 : (same ? 2 : 1);
    using Kernel = void (*)(sf::IsoParams, const double *, const double *, double *, double *, int, int);
    static const Kernel table[3][2][2] = {
        {{sf::iso_stage_kernel<0, false, false>, sf::iso_stage_kernel<0, false, true>},
         {sf::iso_stage_kernel<0, true, false>, sf::iso_stage_kernel<0, true, true>}},
        {{sf::iso_stage_kernel<1, false, false>, sf::iso_stage_kernel<1, false, true>},
         {sf::iso_stage_kernel<1, true, false>, sf::iso_stage_kernel<1, true, true>}},
        {{sf::iso_stage_kernel<2, false, false>, sf::iso_stage_kernel<2, false, true>},
         {sf::iso_stage_kernel<2, true, false>, sf::iso_stage_kernel<2, true, true>}},
    };
    Kernel kernel = table[ws][P.un_mode ? 1 : 0][full ? 1 : 0];
    hipLaunchKernelGGL(kernel, grid, block, 0, t_stream, P, prd, unsrc, pwr, wmax, nstrips, nchunks);
    return check_launch("cbdiso_2d stage launch");
}

int launch_gam_stage(sf::GamParams &P, const double *prd, const double *unsrc, double *pwr, double *wmax)
{
    if (P.ni < 1 || P.nj < 1) return fail(-1, "cbdgam_2d stage: ni, nj must be >= 1");
    if (!prd || !pwr) return fail(-1, "cbdgam_2d stage: null primitive array");
    if (prd == pwr) return fail(-1, "cbdgam_2d stage: primitive_rd and primitive_wr must not alias");
    if (P.un_mode && !unsrc) return fail(-1, "cbdgam_2d stage: a != 0 needs the u^n source array");
    if (P.want_wavespeed && !wmax) return fail(-1, "cbdgam_2d stage: want_wavespeed needs wmax");
    if (P.i_begin < 0 || P.i_end > P.ni || P.i_begin > P.i_end) return fail(-1, "cbdgam_2d stage: bad row range");
    if (P.want_wavespeed == 2) {
        hipError_t e = hipMemsetAsync(wmax, 0, sizeof(double), t_stream);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    const int nrows = P.i_end - P.i_begin;
    if (P.host_out && !P.amax_ptr) return fail(-1, "stage: the host hand-over needs amax_device");
    if (nrows == 0) {
        if (P.zero_ptr || P.host_out) { // the step protocol's bookkeeping must happen even when no row is updated
            hipLaunchKernelGGL(sf::step_bookkeeping_kernel, dim3(1), dim3(64), 0, t_stream, P.zero_ptr, P.host_out,
                               P.host_seq, P.amax_ptr);
            return check_launch("step_bookkeeping_kernel");
        }
        return 0;
    }
    const int nstrips = (P.nj + sf::STRIP - 1) / sf::STRIP;
    int nchunks;
    plan_rows(nrows, nstrips, &P.rows_per_chunk, &nchunks, 9216);
    const long waves = (long)nstrips * nchunks;
    const int wpb = waves_per_block();
    dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(64 * wpb);
    P.reverse = P.un_mode != 0 && reverse_later_stages();
    // the full-physics instantiations (see gam_stage_kernel): same formulas, flag tests resolved here
    auto active = [](const sf::Mass &m) { return m.mass > 0.0 && m.sink_model >= 1 && m.sink_model <= 3; };
    const bool full = P.alpha > 0.0 && active(P.m[0]) && active(P.m[1]) && P.buf_on && P.constant_softening &&
                      P.cooling == 0.0 && !env_int("SFHIP_GENERIC_STAGE", 0);
    using Kernel = void (*)(sf::GamParams, const double *, const double *, double *, double *, int, int);
    static const Kernel table[2][2][2] = {
        {{sf::gam_stage_kernel<false, false, false>, sf::gam_stage_kernel<false, false, true>},
         {sf::gam_stage_kernel<false, true, false>, sf::gam_stage_kernel<false, true, true>}},
        {{sf::gam_stage_kernel<true, false, false>, sf::gam_stage_kernel<true, false, true>},
         {sf::gam_stage_kernel<true, true, false>, sf::gam_stage_kernel<true, true, true>}},
    };
    Kernel kernel = table[P.want_wavespeed ? 1 : 0][P.un_mode ? 1 : 0][full ? 1 : 0];
    hipLaunchKernelGGL(kernel, grid, block, 0, t_stream, P, prd, unsrc, pwr, wmax, nstrips, nchunks);
    return check_launch("cbdgam_2d stage launch");
}

void fill_iso(sf::IsoParams &P, int ni, int nj, double pxl, double pxr, double pyl, double pyr,
              double bsd, double bmc, double brate, double brout, double bw, int bon,
              const sf::Mass &m1, const sf::Mass &m2, double cs2, double mach2, int eos,
              double nu, double a, double dt, double vmax, double dfloor)
{
    memset(&P, 0, sizeof(P));
    P.ni = ni; P.nj = nj;
    P.xl = pxl; P.yl = pyl;
    P.dx = (pxr - pxl) / ni; // cbdiso_2d.c:440-441
    P.dy = (pyr - pyl) / nj;
    P.inv_dx = 1.0 / P.dx;
    P.inv_dy = 1.0 / P.dy;
    P.m[0] = m1; P.m[1] = m2;
    P.buf_sd = bsd; P.buf_mc = bmc; P.buf_rate = brate; P.buf_rout = brout;
    P.buf_ron = brout - bw;                               // cbdiso_2d.c:229
    P.buf_omega = sqrt(bmc * pow(P.buf_ron, -3.0));       // cbdiso_2d.c:237 (host libm, as the reference)
    P.buf_ron2 = P.buf_ron * P.buf_ron;
    P.buf_rate_scale = brate * P.buf_omega / (brout - P.buf_ron); // cbdiso_2d.c:238
    P.buf_on = bon;
    P.eos = eos; P.cs2 = cs2; P.mach2 = mach2;
    P.eos_local = eos == 2 && (m1.mass > 0.0 || m2.mass > 0.0);
    // cbdiso_2d.c:194-211 for a spatially uniform sound speed (eos 2 without masses gives -0/M^2)
    P.cs2_uniform = eos == 1 ? cs2 : (eos == 2 ? -0.0 / mach2 : 1.0);
    P.cs_uniform = sqrt(P.cs2_uniform);
    P.inv_mach2 = 1.0 / mach2;
    P.nu = nu; P.a = a; P.dt = dt; P.vmax = vmax; P.dfloor = dfloor;
    P.half_nu = 0.5 * nu;
    P.half_inv_dx = 0.5 * P.inv_dx;
    P.half_inv_dy = 0.5 * P.inv_dy;
    P.one_minus_a = 1.0 - a;
    P.sink_reach2[0] = 16.0 * m1.sink_radius * m1.sink_radius + m1.rsoft * m1.rsoft;
    P.sink_reach2[1] = 16.0 * m2.sink_radius * m2.sink_radius + m2.rsoft * m2.rsoft;
    P.clamp_ilo = P.clamp_ihi = P.clamp_j = 0;
    P.un_mode = a != 0.0 ? 1 : 0;
    P.i_begin = 0; P.i_end = ni;
    P.want_wavespeed = 0;
    P.mw[0] = m1; P.mw[1] = m2;
}

void fill_gam(sf::GamParams &P, int ni, int nj, double pxl, double pxr, double pyl, double pyr, double gamma,
              double bsd, double bsp, double bmc, double brate, double brout, double bw, int bon,
              const sf::Mass &m1, const sf::Mass &m2, double alpha, double a, double dt, double vmax,
              double cooling, double mach_ceiling, double dfloor, double pfloor, int constant_softening)
{
    memset(&P, 0, sizeof(P));
    P.ni = ni; P.nj = nj;
    P.xl = pxl; P.yl = pyl;
    P.dx = (pxr - pxl) / ni; // cbdgam_2d.c:498-499
    P.dy = (pyr - pyl) / nj;
    P.inv_dx = 1.0 / P.dx;
    P.inv_dy = 1.0 / P.dy;
    P.m[0] = m1; P.m[1] = m2;
    P.any_mass = !(m1.mass == 0.0 && m2.mass == 0.0); // cbdgam_2d.c:81
    P.gamma = gamma;
    P.gm1 = gamma - 1.0;
    P.inv_gm1 = 1.0 / (gamma - 1.0);
    P.sqrt_gamma = sqrt(gamma);
    P.buf_sd = bsd; P.buf_sp = bsp; P.buf_mc = bmc; P.buf_rate = brate; P.buf_rout = brout;
    P.buf_ron = brout - bw;                               // cbdgam_2d.c:255
    P.buf_omega = sqrt(bmc * pow(P.buf_ron, -3.0));       // cbdgam_2d.c:266
    P.buf_ron2 = P.buf_ron * P.buf_ron;
    P.buf_rate_scale = brate * P.buf_omega / (brout - P.buf_ron); // cbdgam_2d.c:268
    P.buf_e0 = bsp / (gamma - 1.0);                       // cbdgam_2d.c:263
    P.buf_on = bon;
    P.constant_softening = constant_softening;
    P.alpha = alpha; P.a = a; P.dt = dt; P.vmax = vmax; P.cooling = cooling;
    P.mach_ceiling_m2 = pow(mach_ceiling, -2.0);          // cbdgam_2d.c:314
    P.mach_floor_scale = 2.0 / gamma / (gamma - 1.0) * P.mach_ceiling_m2;
    P.dfloor = dfloor; P.pfloor = pfloor;
    P.half_inv_dx = 0.5 * P.inv_dx;
    P.half_inv_dy = 0.5 * P.inv_dy;
    P.one_minus_a = 1.0 - a;
    P.half_alpha = 0.5 * alpha;
    P.inv_buf_sd = bsd != 0.0 ? 1.0 / bsd : 0.0;
    P.sink_near2[0] = 16.0 * m1.sink_radius * m1.sink_radius;
    P.sink_near2[1] = 16.0 * m2.sink_radius * m2.sink_radius;
    P.un_mode = a != 0.0 ? 1 : 0;
    P.i_begin = 0; P.i_end = ni;
}

__global__ void max_nonneg_kernel(const double *__restrict__ x, long n, double *__restrict__ out)
{
    double a = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    // eight independent loads per trip (one per trip left the kernel latency-bound at 1.2 TB/s)
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < n; k += 8 * stride) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k + u * stride < n ? x[k + u * stride] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) a = sf::max_nan(a, v[u]);
    }
    // one atomic per block: thousands of same-address atomics at the end of so short a kernel serialise
    __shared__ double wave_max[4];
    a = sf::wave_max_nonneg(a);
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = wave_max[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = sf::max_nan(m, wave_max[w]);
        sf::atomic_max_nonneg(out, m);
    }
}

// lane-shift self test: out_below[t] = value from lane t-1, out_above[t] = value from lane t+1
__global__ void lane_shift_kernel(const double *in, double *out_below, double *out_above)
{
    int t = threadIdx.x;
    double x = in[t];
    out_below[t] = sf::from_lane_below(x);
    out_above[t] = sf::from_lane_above(x);
}

// dt = dx / amax * cfl in exactly the reference driver's operation order (driver.py:509-510), IEEE ops
__global__ void timestep_kernel(double dx, double cfl, const double *amax, double *dt)
{
#pragma clang fp contract(off)
    if (threadIdx.x == 0 && blockIdx.x == 0) dt[0] = dx / amax[0] * cfl;
}

// fast-math self test: out[0..n) = fast_rcp(x), out[n..2n) = fast_rsqrt(x), out[2n..3n) = sqrt via rsqrt
__global__ void fastmath_kernel(const double *x, double *out, int n)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double v = x[k];
    double y = sf::fast_rsqrt(v);
    out[k] = sf::fast_rcp(v);
    out[n + k] = y;
    out[2 * n + k] = sf::sqrt_from_rsqrt(v, y);
}

} // namespace

extern "C" {

int sfhip_set_device(int device)
{
    hipError_t e = hipSetDevice(device);
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

int sfhip_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int sfhip_set_stream(void *stream)
{
    t_stream = (hipStream_t)stream;
    return 0;
}

int sfhip_synchronize(void)
{
    hipError_t e = hipStreamSynchronize(t_stream);
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

const char *sfhip_last_error(void) { return t_error; }

int sfhip_selftest_lane_shift(const double *in64, double *out_below64, double *out_above64)
{
    hipLaunchKernelGGL(lane_shift_kernel, dim3(1), dim3(64), 0, t_stream, in64, out_below64, out_above64);
    return check_launch("lane_shift_kernel");
}

int sfhip_timestep(int n, double dx, double cfl, double *wavespeed_max, double *dt_out)
{
    if (n != 1 || !wavespeed_max || !dt_out) return fail(-1, "sfhip_timestep: bad arguments");
    hipLaunchKernelGGL(timestep_kernel, dim3(1), dim3(64), 0, t_stream, dx, cfl, wavespeed_max, dt_out);
    return check_launch("timestep_kernel");
}

int sfhip_selftest_fastmath(int n, double *x, double *out3n)
{
    if (n < 1 || !x || !out3n) return fail(-1, "sfhip_selftest_fastmath: bad arguments");
    hipLaunchKernelGGL(fastmath_kernel, dim3((n + 255) / 256), dim3(256), 0, t_stream, x, out3n, n);
    return check_launch("fastmath_kernel");
}

// max over an array of non-negative doubles into *out (device); zeroes *out first
int sfhip_max_nonneg(int n, double *x, double *out)
{
    if (!x || !out || n < 0) return fail(-1, "sfhip_max_nonneg: bad arguments");
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), t_stream);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    if (n == 0) return 0;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(max_nonneg_kernel, dim3((unsigned)blocks), dim3(256), 0, t_stream, x, n, out);
    return check_launch("max_nonneg_kernel");
}

// Initial data of the isothermal circumbinary disk, made in place in device memory (sf_setup.hpp)
int sfhip_circumbinary_disk_primitive(int ni, int nj, double x0, double y0, double dx, double dy, int i0,
                                      double softening2, double sigma, double *primitive)
{
    if (ni < 1 || nj < 1 || i0 < 0 || !primitive) return fail(-1, "sfhip_circumbinary_disk_primitive: bad arguments");
    dim3 block(256), grid((nj + 255) / 256, ni < 4096 ? ni : 4096);
    hipLaunchKernelGGL(sf::disk_primitive_kernel, grid, block, 0, t_stream, ni, nj, i0, x0, y0, dx, dy, softening2,
                       sigma, primitive);
    return check_launch("sfhip_circumbinary_disk_primitive");
}

// Initial data of the gamma-law circumbinary disk on the HOST (include/sfhip.h): CircumbinaryDisk.primitive's
// operations, circumbinary_disk.py:82-108, with the libm calls the interpreter makes (host code is compiled with
// contraction off as well), rows spread over threads.
int sfhip_host_gamma_law_disk_primitive(int ni, int nj, double x0, double y0, double dx, double dy, int i0, double softening2,
                                        double sigma, double pressure, double exponent_sigma, double exponent_pressure,
                                        int threads, double *out)
{
    if (ni < 1 || nj < 1 || i0 < 0 || !out) return fail(-1, "sfhip_host_gamma_law_disk_primitive: bad arguments");
    if (threads <= 0) {
        cpu_set_t set;
        threads = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : 1;
        const int cap = env_int("OMP_NUM_THREADS", 64);
        if (threads > cap) threads = cap;
        if (threads > 64) threads = 64;
    }
    if (threads > ni) threads = ni;
    if (threads < 1) threads = 1;
    auto rows = [=](int a, int b) {
        for (int i = a; i < b; ++i) {
            const double x = x0 + ((double)(i0 + i) + 0.5) * dx; // mesh.cell_coordinates
            for (int j = 0; j < nj; ++j) {
                const double y = y0 + ((double)j + 0.5) * dy;
                const double r = sqrt(x * x + y * y);
                const double rs = sqrt(x * x + y * y + softening2);
                const double rc = r > 1e-12 ? r : 1e-12;                       // max(r, 1e-12)
                const double phx = -y / rc, phy = x / rc;
                const double cavity = 0.0001 + 0.9999 * exp(-pow(1.0 / rs, 30.0));
                double *p = out + ((long)i * nj + j) * 4;
                p[0] = sigma * pow(rs, exponent_sigma) * cavity;
                p[1] = sqrt(1.0 / rs) * phx;
                p[2] = sqrt(1.0 / rs) * phy;
                p[3] = pressure * pow(rs, exponent_pressure) * cavity;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t)
        pool.emplace_back(rows, (int)((long)ni * t / threads), (int)((long)ni * (t + 1) / threads));
    rows(0, (int)((long)ni / threads));
    for (auto &th : pool) th.join();
    return 0;
}

// ---------------------------------------------------------------------------
// cbdiso_2d -- reference-named entry points
// ---------------------------------------------------------------------------
int cbdiso_2d_advance_rk(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double *conserved_rk, double *primitive_rd, double *primitive_wr,
    double buffer_surface_density, double buffer_central_mass, double buffer_driving_rate,
    double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double cs2, double mach_squared, int eos_type, double nu, double a, double dt,
    double velocity_ceiling, double density_floor)
{
    sf::IsoParams P;
    fill_iso(P, ni, nj, patch_xl, patch_xr, patch_yl, patch_yr,
             buffer_surface_density, buffer_central_mass, buffer_driving_rate, buffer_outer_radius,
             buffer_onset_width, buffer_is_enabled,
             make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
             make_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
             cs2, mach_squared, eos_type, nu, a, dt, velocity_ceiling, density_floor);
    return launch_iso_stage(P, primitive_rd, conserved_rk, primitive_wr, nullptr);
}

// Fused stage: cbdiso_2d_advance_rk plus the passes the reference runs around it.
int sfhip_cbdiso_2d_stage(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double *un_source, double *primitive_rd, double *primitive_wr,
    double buffer_surface_density, double buffer_central_mass, double buffer_driving_rate,
    double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double cs2, double mach_squared, int eos_type, double nu, double a, double dt,
    double velocity_ceiling, double density_floor,
    int un_mode, int clamp_ilo, int clamp_ihi, int i_begin, int i_end, int want_wavespeed,
    double wavespeed_floor,
    double wx1, double wy1, double wmass1, double wsoftening_length1,
    double wx2, double wy2, double wmass2, double wsoftening_length2,
    double *wavespeed_max, double *dt_device,
    double *amax_device, double dt_dx, double dt_cfl, double *zero_slot)
{
    if (zero_slot && want_wavespeed && zero_slot == wavespeed_max)
        return fail(-1, "sfhip_cbdiso_2d_stage: zero_slot must not be the slot this launch folds into");
    if (amax_device && dt_device) return fail(-1, "sfhip_cbdiso_2d_stage: give dt_device or amax_device, not both");
    sf::IsoParams P;
    fill_iso(P, ni, nj, patch_xl, patch_xr, patch_yl, patch_yr,
             buffer_surface_density, buffer_central_mass, buffer_driving_rate, buffer_outer_radius,
             buffer_onset_width, buffer_is_enabled,
             make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
             make_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
             cs2, mach_squared, eos_type, nu, a, dt, velocity_ceiling, density_floor);
    P.un_mode = a != 0.0 ? un_mode : 0;
    P.clamp_ilo = clamp_ilo; P.clamp_ihi = clamp_ihi; P.clamp_j = 1;
    P.i_begin = i_begin; P.i_end = i_end;
    P.want_wavespeed = want_wavespeed;
    P.ws_floor = wavespeed_floor;
    P.mw[0] = make_mass(wx1, wy1, 0, 0, wmass1, wsoftening_length1, 0, 0, 0);
    P.mw[1] = make_mass(wx2, wy2, 0, 0, wmass2, wsoftening_length2, 0, 0, 0);
    P.dt_ptr = dt_device;
    P.amax_ptr = amax_device; P.dt_dx = dt_dx; P.dt_cfl = dt_cfl; P.zero_ptr = zero_slot;
    return launch_iso_stage(P, primitive_rd, un_source, primitive_wr, wavespeed_max);
}

int cbdiso_2d_primitive_to_conserved(int ni, int nj, double *primitive, double *conserved)
{
    if (ni < 1 || nj < 1 || !primitive || !conserved) return fail(-1, "cbdiso_2d_primitive_to_conserved: bad arguments");
    dim3 block(256), grid((nj + 255) / 256, ni);
    hipLaunchKernelGGL(sf::iso_p2c_kernel, grid, block, 0, t_stream, ni, nj, primitive, conserved);
    return check_launch("cbdiso_2d_primitive_to_conserved");
}

static int iso_wavespeed_impl(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double soundspeed2, double mach_squared, int eos_type, const sf::Mass &m1, const sf::Mass &m2,
    double *primitive, double *wavespeed, double *wmax)
{
    if (ni < 1 || nj < 1 || !primitive) return fail(-1, "cbdiso_2d_wavespeed: bad arguments");
    sf::IsoParams P;
    fill_iso(P, ni, nj, patch_xl, patch_xr, patch_yl, patch_yr, 0, 0, 0, 0, 0, 0, m1, m2,
             soundspeed2, mach_squared, eos_type, 0, 0, 0, 0, 0);
    if (wmax) {
        hipError_t e = hipMemsetAsync(wmax, 0, sizeof(double), t_stream);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    int gy = (ni + 3) / 4 < 512 ? (ni + 3) / 4 : 512; // four rows per trip of a thread's loop
    dim3 block(256), grid((nj + 255) / 256, gy);
    hipLaunchKernelGGL(sf::iso_wavespeed_kernel, grid, block, 0, t_stream, P, primitive, wavespeed, wmax);
    return check_launch("cbdiso_2d_wavespeed");
}

int cbdiso_2d_wavespeed(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double soundspeed2, double mach_squared, int eos_type,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double *primitive, double *wavespeed)
{
    if (!wavespeed) return fail(-1, "cbdiso_2d_wavespeed: null output");
    return iso_wavespeed_impl(ni, nj, patch_xl, patch_xr, patch_yl, patch_yr, soundspeed2, mach_squared, eos_type,
        make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
        make_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
        primitive, wavespeed, nullptr);
}

// wavespeed + max in one pass, no per-zone array (replaces kernel + xp.max(), cbdiso_2d.py:139-174)
int sfhip_cbdiso_2d_max_wavespeed(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double soundspeed2, double mach_squared, int eos_type,
    double x1, double y1, double mass1, double softening_length1,
    double x2, double y2, double mass2, double softening_length2,
    double *primitive, double *wavespeed_max)
{
    if (!wavespeed_max) return fail(-1, "sfhip_cbdiso_2d_max_wavespeed: null output");
    return iso_wavespeed_impl(ni, nj, patch_xl, patch_xr, patch_yl, patch_yr, soundspeed2, mach_squared, eos_type,
        make_mass(x1, y1, 0, 0, mass1, softening_length1, 0, 0, 0),
        make_mass(x2, y2, 0, 0, mass2, softening_length2, 0, 0, 0),
        primitive, nullptr, wavespeed_max);
}

int cbdiso_2d_point_mass_source_term(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double *primitive, double *cons_rate)
{
    if (ni < 1 || nj < 1 || !primitive || !cons_rate) return fail(-1, "cbdiso_2d_point_mass_source_term: bad arguments");
    sf::IsoParams P;
    sf::Mass m1 = make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1);
    fill_iso(P, ni, nj, patch_xl, patch_xr, patch_yl, patch_yr, 0, 0, 0, 0, 0, 0, m1, m1, 0, 0, 0, 0, 0, 0, 0, 0);
    dim3 block(256), grid((nj + 255) / 256, ni);
    hipLaunchKernelGGL(sf::iso_mass_rate_kernel, grid, block, 0, t_stream, P, primitive, cons_rate);
    return check_launch("cbdiso_2d_point_mass_source_term");
}

// ---------------------------------------------------------------------------
// cbdgam_2d -- reference-named entry points
// ---------------------------------------------------------------------------
int cbdgam_2d_advance_rk(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double *conserved_rk, double *primitive_rd, double *primitive_wr,
    double gamma_law_index,
    double buffer_surface_density, double buffer_surface_pressure, double buffer_central_mass,
    double buffer_driving_rate, double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double alpha, double a, double dt, double velocity_ceiling, double cooling_coefficient,
    double mach_ceiling, double density_floor, double pressure_floor, int constant_softening)
{
    sf::GamParams P;
    fill_gam(P, ni, nj, patch_xl, patch_xr, patch_yl, patch_yr, gamma_law_index,
             buffer_surface_density, buffer_surface_pressure, buffer_central_mass, buffer_driving_rate,
             buffer_outer_radius, buffer_onset_width, buffer_is_enabled,
             make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
             make_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
             alpha, a, dt, velocity_ceiling, cooling_coefficient, mach_ceiling, density_floor, pressure_floor,
             constant_softening);
    return launch_gam_stage(P, primitive_rd, conserved_rk, primitive_wr, nullptr);
}

int sfhip_cbdgam_2d_stage(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double *un_source, double *primitive_rd, double *primitive_wr,
    double gamma_law_index,
    double buffer_surface_density, double buffer_surface_pressure, double buffer_central_mass,
    double buffer_driving_rate, double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double alpha, double a, double dt, double velocity_ceiling, double cooling_coefficient,
    double mach_ceiling, double density_floor, double pressure_floor, int constant_softening,
    int un_mode, int clamp_ilo, int clamp_ihi, int i_begin, int i_end, int want_wavespeed,
    double wavespeed_floor, double *wavespeed_max, double *dt_device,
    double *amax_device, double dt_dx, double dt_cfl, double *zero_slot)
{
    if (zero_slot && want_wavespeed && zero_slot == wavespeed_max)
        return fail(-1, "sfhip_cbdgam_2d_stage: zero_slot must not be the slot this launch folds into");
    if (amax_device && dt_device) return fail(-1, "sfhip_cbdgam_2d_stage: give dt_device or amax_device, not both");
    sf::GamParams P;
    fill_gam(P, ni, nj, patch_xl, patch_xr, patch_yl, patch_yr, gamma_law_index,
             buffer_surface_density, buffer_surface_pressure, buffer_central_mass, buffer_driving_rate,
             buffer_outer_radius, buffer_onset_width, buffer_is_enabled,
             make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
             make_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
             alpha, a, dt, velocity_ceiling, cooling_coefficient, mach_ceiling, density_floor, pressure_floor,
             constant_softening);
    P.un_mode = a != 0.0 ? un_mode : 0;
    P.clamp_ilo = clamp_ilo; P.clamp_ihi = clamp_ihi; P.clamp_j = 1;
    P.i_begin = i_begin; P.i_end = i_end;
    P.want_wavespeed = want_wavespeed;
    P.ws_floor = wavespeed_floor;
    P.dt_ptr = dt_device;
    P.amax_ptr = amax_device; P.dt_dx = dt_dx; P.dt_cfl = dt_cfl; P.zero_ptr = zero_slot;
    return launch_gam_stage(P, primitive_rd, un_source, primitive_wr, wavespeed_max);
}

int cbdgam_2d_wavespeed(int ni, int nj, double *primitive, double *wavespeed, double gamma_law_index)
{
    if (ni < 1 || nj < 1 || !primitive || !wavespeed) return fail(-1, "cbdgam_2d_wavespeed: bad arguments");
    int gy = (ni + 3) / 4 < 512 ? (ni + 3) / 4 : 512; // four rows per trip of a thread's loop
    dim3 block(256), grid((nj + 255) / 256, gy);
    hipLaunchKernelGGL(sf::gam_wavespeed_kernel, grid, block, 0, t_stream, ni, nj, gamma_law_index, primitive, wavespeed, (double *)nullptr);
    return check_launch("cbdgam_2d_wavespeed");
}

int sfhip_cbdgam_2d_max_wavespeed(int ni, int nj, double *primitive, double gamma_law_index, double *wavespeed_max)
{
    if (ni < 1 || nj < 1 || !primitive || !wavespeed_max) return fail(-1, "sfhip_cbdgam_2d_max_wavespeed: bad arguments");
    hipError_t e = hipMemsetAsync(wavespeed_max, 0, sizeof(double), t_stream);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    int gy = (ni + 3) / 4 < 512 ? (ni + 3) / 4 : 512; // four rows per trip of a thread's loop
    dim3 block(256), grid((nj + 255) / 256, gy);
    hipLaunchKernelGGL(sf::gam_wavespeed_kernel, grid, block, 0, t_stream, ni, nj, gamma_law_index, primitive, (double *)nullptr, wavespeed_max);
    return check_launch("sfhip_cbdgam_2d_max_wavespeed");
}

int cbdgam_2d_primitive_to_conserved(int ni, int nj, double *primitive, double *conserved, double gamma_law_index)
{
    if (ni < 1 || nj < 1 || !primitive || !conserved) return fail(-1, "cbdgam_2d_primitive_to_conserved: bad arguments");
    dim3 block(256), grid((nj + 255) / 256, ni);
    hipLaunchKernelGGL(sf::gam_p2c_kernel, grid, block, 0, t_stream, ni, nj, gamma_law_index, primitive, conserved);
    return check_launch("cbdgam_2d_primitive_to_conserved");
}

int cbdgam_2d_point_mass_source_term(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    int which_mass, double *primitive, double *cons_rate, int constant_softening, double gamma_law_index)
{
    if (ni < 1 || nj < 1 || !primitive || !cons_rate) return fail(-1, "cbdgam_2d_point_mass_source_term: bad arguments");
    if (which_mass != 1 && which_mass != 2) return fail(-1, "cbdgam_2d_point_mass_source_term: which_mass must be 1 or 2");
    sf::GamParams P;
    fill_gam(P, ni, nj, patch_xl, patch_xr, patch_yl, patch_yr, gamma_law_index, 0, 0, 0, 0, 0, 0, 0,
             make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
             make_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
             0, 0, 0, 0, 0, 1.0, 0, 0, constant_softening);
    dim3 block(256), grid((nj + 255) / 256, ni);
    hipLaunchKernelGGL(sf::gam_mass_rate_kernel, grid, block, 0, t_stream, P, which_mass - 1, primitive, cons_rate);
    return check_launch("cbdgam_2d_point_mass_source_term");
}


} // extern "C"

// ---------------------------------------------------------------------------
// Launch plans: the fused stage with its constant arguments stored once.
//
// sfhip_cbd{iso,gam}_2d_stage take 60-odd scalars per call; marshalling them through ctypes costs the Python
// host ~20 us per launch, which is what bounds the step rate of small patches (512^2 on one GPU; the slabs of a
// strong-scaling run).  A plan keeps what does not change between launches -- mesh, physics constants, boundary
// flags, streams -- and a launch passes only the arrays, the two masses, the RK parameter, dt, the row range and
// the wavespeed arguments.  The launch itself is the very same launch_{iso,gam}_stage as the full entry point's.
// A plan also owns what the host needs to read the reduced wavespeed back without torch's stream/event
// bookkeeping: 8 bytes of pinned host memory, a stream and two events, created with the plan (the only
// allocations; the per-step calls allocate nothing).
// ---------------------------------------------------------------------------
namespace {

struct StagePlan {
    int kind = 0; // 0 free, 1 cbdiso_2d, 2 cbdgam_2d
    int device = 0;
    int ni = 0, nj = 0;
    double xl = 0, xr = 0, yl = 0, yr = 0;
    double bsd = 0, bsp = 0, brate = 0, brout = 0, bw = 0;
    int bon = 0;
    double cs2 = 0, mach2 = 0;
    int eos = 0;
    double nu = 0, vmax = 0, dfloor = 0;
    double gamma = 0, alpha = 0, cooling = 0, mach_ceiling = 0, pfloor = 0;
    int constant_softening = 0;
    int clamp_ilo = 0, clamp_ihi = 0;
    hipStream_t compute = nullptr, side = nullptr;
    double *pinned = nullptr;     // host view: [2] value and [3] sequence number written by a stage
    double *pinned_dev = nullptr; // device view of the same memory
    // sfhip_plan_step_rk2's host time by phase, accumulated (ns): stage-0 launch, wait for the hand-over,
    // dt + orbit, stage-1 launch; and the number of steps (sfhip_plan_step_timing)
    double phase_ns[4] = {0, 0, 0, 0};
    long steps_timed = 0;
};

constexpr int MAX_PLANS = 1024;
StagePlan g_plans[MAX_PLANS];
std::mutex g_plan_mutex;

StagePlan *plan_of(int id)
{
    if (id < 0 || id >= MAX_PLANS || g_plans[id].kind == 0) {
        fail(-1, "sfhip_plan: no such plan");
        return nullptr;
    }
    return &g_plans[id];
}

int plan_alloc(StagePlan **out)
{
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    for (int id = 0; id < MAX_PLANS; ++id) {
        if (g_plans[id].kind == 0) {
            StagePlan &pl = g_plans[id];
            pl = StagePlan();
            hipError_t e = hipGetDevice(&pl.device);
            if (e == hipSuccess) e = hipHostMalloc((void **)&pl.pinned, 8 * sizeof(double), hipHostMallocDefault);
            if (e == hipSuccess) {
                memset(pl.pinned, 0, 8 * sizeof(double));
                e = hipHostGetDevicePointer((void **)&pl.pinned_dev, pl.pinned, 0);
            }
            if (e != hipSuccess) {
                if (pl.pinned) (void)hipHostFree(pl.pinned);
                pl = StagePlan();
                fail((int)e, hipGetErrorString(e));
                return -1;
            }
            pl.kind = -1; // taken; the caller sets the real kind
            *out = &pl;
            return id;
        }
    }
    fail(-1, "sfhip_plan: too many plans");
    return -1;
}

sf::Mass mass_from(const double *m)
{
    return make_mass(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], (int)m[8]);
}

} // namespace

extern "C" {

int sfhip_plan_create_cbdiso_2d(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double buffer_surface_density, double buffer_driving_rate, double buffer_outer_radius,
    double buffer_onset_width, int buffer_is_enabled, double cs2, double mach_squared, int eos_type, double nu,
    double velocity_ceiling, double density_floor, int clamp_ilo, int clamp_ihi, void *compute_stream)
{
    if (ni < 1 || nj < 1) return fail(-1, "sfhip_plan_create_cbdiso_2d: ni, nj must be >= 1");
    StagePlan *pl = nullptr;
    const int id = plan_alloc(&pl);
    if (id < 0) return id;
    pl->ni = ni; pl->nj = nj; pl->xl = patch_xl; pl->xr = patch_xr; pl->yl = patch_yl; pl->yr = patch_yr;
    pl->bsd = buffer_surface_density; pl->brate = buffer_driving_rate; pl->brout = buffer_outer_radius;
    pl->bw = buffer_onset_width; pl->bon = buffer_is_enabled;
    pl->cs2 = cs2; pl->mach2 = mach_squared; pl->eos = eos_type; pl->nu = nu;
    pl->vmax = velocity_ceiling; pl->dfloor = density_floor;
    pl->clamp_ilo = clamp_ilo; pl->clamp_ihi = clamp_ihi;
    pl->compute = (hipStream_t)compute_stream;
    pl->side = (hipStream_t)compute_stream;
    pl->kind = 1;
    return id;
}

int sfhip_plan_create_cbdgam_2d(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr, double gamma_law_index,
    double buffer_surface_density, double buffer_surface_pressure, double buffer_driving_rate,
    double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled, double alpha,
    double velocity_ceiling, double cooling_coefficient, double mach_ceiling, double density_floor,
    double pressure_floor, int constant_softening, int clamp_ilo, int clamp_ihi, void *compute_stream)
{
    if (ni < 1 || nj < 1) return fail(-1, "sfhip_plan_create_cbdgam_2d: ni, nj must be >= 1");
    StagePlan *pl = nullptr;
    const int id = plan_alloc(&pl);
    if (id < 0) return id;
    pl->ni = ni; pl->nj = nj; pl->xl = patch_xl; pl->xr = patch_xr; pl->yl = patch_yl; pl->yr = patch_yr;
    pl->gamma = gamma_law_index;
    pl->bsd = buffer_surface_density; pl->bsp = buffer_surface_pressure; pl->brate = buffer_driving_rate;
    pl->brout = buffer_outer_radius; pl->bw = buffer_onset_width; pl->bon = buffer_is_enabled;
    pl->alpha = alpha; pl->vmax = velocity_ceiling; pl->cooling = cooling_coefficient;
    pl->mach_ceiling = mach_ceiling; pl->dfloor = density_floor; pl->pfloor = pressure_floor;
    pl->constant_softening = constant_softening;
    pl->clamp_ilo = clamp_ilo; pl->clamp_ihi = clamp_ihi;
    pl->compute = (hipStream_t)compute_stream;
    pl->side = (hipStream_t)compute_stream;
    pl->kind = 2;
    return id;
}

int sfhip_plan_set_side_stream(int plan, void *side_stream)
{
    StagePlan *pl = plan_of(plan);
    if (!pl) return -1;
    pl->side = (hipStream_t)side_stream;
    return 0;
}

int sfhip_plan_destroy(int plan)
{
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    StagePlan *pl = plan_of(plan);
    if (!pl) return -1;
    (void)hipHostFree(pl->pinned);
    *pl = StagePlan();
    return 0;
}

// masses: 18 host doubles, (x, y, vx, vy, mass, softening_length, sink_rate, sink_radius, sink_model) of mass 1,
// then of mass 2, at the stage time; next_masses: 8 host doubles, (x, y, mass, softening_length) of the two masses
// at the time the output state belongs to (cbdiso_2d with want_wavespeed only; may be NULL otherwise).
// Everything else as in sfhip_cbd{iso,gam}_2d_stage; un_mode is 2 when un_source is given.
int sfhip_plan_stage(int plan, const double *masses, const double *next_masses,
                     double *un_source, double *primitive_rd, double *primitive_wr, double a, double dt,
                     int i_begin, int i_end, int want_wavespeed, double wavespeed_floor, double *wavespeed_max,
                     double *amax_device, double dt_dx, double dt_cfl, double *zero_slot, int on_side_stream,
                     double host_seq)
{
    StagePlan *pl = plan_of(plan);
    if (!pl) return -1;
    if (!masses) return fail(-1, "sfhip_plan_stage: null masses");
    if (host_seq != 0.0 && !amax_device) return fail(-1, "sfhip_plan_stage: host_seq needs amax_device");
    if (zero_slot && want_wavespeed && zero_slot == wavespeed_max)
        return fail(-1, "sfhip_plan_stage: zero_slot must not be the slot this launch folds into");
    const sf::Mass m1 = mass_from(masses), m2 = mass_from(masses + 9);
    const hipStream_t saved = t_stream;
    t_stream = on_side_stream ? pl->side : pl->compute;
    int rc;
    if (pl->kind == 1) {
        sf::IsoParams P;
        fill_iso(P, pl->ni, pl->nj, pl->xl, pl->xr, pl->yl, pl->yr, pl->bsd, m1.mass + m2.mass /* cbdiso_2d.py:194 */,
                 pl->brate, pl->brout, pl->bw, pl->bon, m1, m2, pl->cs2, pl->mach2, pl->eos, pl->nu, a, dt, pl->vmax,
                 pl->dfloor);
        P.un_mode = a != 0.0 ? 2 : 0; // (a null un_source then fails the launch's own check)
        P.clamp_ilo = pl->clamp_ilo; P.clamp_ihi = pl->clamp_ihi; P.clamp_j = 1;
        P.i_begin = i_begin; P.i_end = i_end;
        P.want_wavespeed = want_wavespeed;
        P.ws_floor = wavespeed_floor;
        if (want_wavespeed) {
            if (!next_masses) { t_stream = saved; return fail(-1, "sfhip_plan_stage: want_wavespeed needs next_masses"); }
            P.mw[0] = make_mass(next_masses[0], next_masses[1], 0, 0, next_masses[2], next_masses[3], 0, 0, 0);
            P.mw[1] = make_mass(next_masses[4], next_masses[5], 0, 0, next_masses[6], next_masses[7], 0, 0, 0);
        }
        P.amax_ptr = amax_device; P.dt_dx = dt_dx; P.dt_cfl = dt_cfl; P.zero_ptr = zero_slot;
        if (host_seq != 0.0) { P.host_out = pl->pinned_dev + 2; P.host_seq = host_seq; }
        rc = launch_iso_stage(P, primitive_rd, un_source, primitive_wr, wavespeed_max);
    } else {
        sf::GamParams P;
        fill_gam(P, pl->ni, pl->nj, pl->xl, pl->xr, pl->yl, pl->yr, pl->gamma, pl->bsd, pl->bsp,
                 m1.mass + m2.mass /* cbdgam_2d.py:164 */, pl->brate, pl->brout, pl->bw, pl->bon, m1, m2, pl->alpha, a,
                 dt, pl->vmax, pl->cooling, pl->mach_ceiling, pl->dfloor, pl->pfloor, pl->constant_softening);
        P.un_mode = a != 0.0 ? 2 : 0; // (a null un_source then fails the launch's own check)
        P.clamp_ilo = pl->clamp_ilo; P.clamp_ihi = pl->clamp_ihi; P.clamp_j = 1;
        P.i_begin = i_begin; P.i_end = i_end;
        P.want_wavespeed = want_wavespeed;
        P.ws_floor = wavespeed_floor;
        P.amax_ptr = amax_device; P.dt_dx = dt_dx; P.dt_cfl = dt_cfl; P.zero_ptr = zero_slot;
        if (host_seq != 0.0) { P.host_out = pl->pinned_dev + 2; P.host_seq = host_seq; }
        rc = launch_gam_stage(P, primitive_rd, un_source, primitive_wr, wavespeed_max);
    }
    t_stream = saved;
    return rc;
}

// ---------------------------------------------------------------------------
// A whole RK2 step from one call: the Python between the two stage launches (dt, the stage time, the Kepler solve
// for the point masses at t + dt, argument packing) is a few tens of microseconds, which is what bounds the step
// rate of small patches once the kernels take 20 us.  Here the orbit is solved on the host in C, with the same
// libm the Python reference calls and the reference's operations in the reference's order (kepler.py:329-373,
// 414-421; contraction is off for host code too), so the ten numbers are bit-identical to
// OrbitalElements.orbital_state (tests/golden/kepler.npz through sfhip_kepler_orbital_state).
// ---------------------------------------------------------------------------
namespace {

// out: mass, x, y, vx, vy of body 1, then of body 2.  Returns 0, or -1 when Newton's method does not converge.
int kepler_orbital_state(double a, double m, double q, double e, double time_since_periapse, double *out)
{
    const double w = sqrt(1.0 * m / a / a / a);                    // omega, NEWTON_G = 1 (kepler.py:296-302)
    const double period = 2.0 * M_PI / w;
    const double t = time_since_periapse - period * floor(time_since_periapse / period);
    const double n = w * t;                                        // mean anomaly
    double k = n;                                                  // eccentric anomaly, Newton from the mean anomaly
    int corrections = 0;
    while (fabs(k - e * sin(k) - n) > 1e-15) {
        k -= (k - e * sin(k) - n) / (1.0 - e * cos(k));
        if (++corrections > 10) return -1;
    }
    const double m1 = m / (1.0 + q);
    const double m2 = m - m1;
    const double ck = cos(k), sk = sin(k);
    const double x1 = -a * q / (1.0 + q) * (e - ck);
    const double y1 = +a * q / (1.0 + q) * (sk) * sqrt(1.0 - e * e);
    const double vx1 = -a * q / (1.0 + q) * w / (1.0 - e * ck) * sk;
    const double vy1 = +a * q / (1.0 + q) * w / (1.0 - e * ck) * ck * sqrt(1.0 - e * e);
    out[0] = m1; out[1] = x1; out[2] = y1; out[3] = vx1; out[4] = vy1;
    out[5] = m2; out[6] = -x1 / q; out[7] = -y1 / q; out[8] = -vx1 / q; out[9] = -vy1 / q;
    return 0;
}

// the 18 doubles sfhip_plan_stage takes, from an orbital state and the four constants every mass shares
void pack_masses(const double *body10, const double *shared4, double *masses18)
{
    for (int k = 0; k < 2; ++k) {
        const double *b = body10 + 5 * k;
        double *m = masses18 + 9 * k;
        m[0] = b[1]; m[1] = b[2]; m[2] = b[3]; m[3] = b[4]; m[4] = b[0];
        m[5] = shared4[0]; m[6] = shared4[1]; m[7] = shared4[2]; m[8] = shared4[3];
    }
}

} // namespace

// Host-only: the orbital state at `time_since_periapse` into 10 HOST doubles (test hook for the C restatement).
int sfhip_kepler_orbital_state(double semimajor_axis, double total_mass, double mass_ratio, double eccentricity,
                               double time_since_periapse, double *out10_host)
{
    if (!out10_host) return fail(-1, "sfhip_kepler_orbital_state: null output");
    if (kepler_orbital_state(semimajor_axis, total_mass, mass_ratio, eccentricity, time_since_periapse, out10_host))
        return fail(-1, "sfhip_kepler_orbital_state: Newton's method did not converge");
    return 0;
}

// sfhip_plan_step_rk2 returns 1 and 2 as outcomes (repeat the step, NaN wavespeed): a failure underneath -- a
// hipError_t, which is positive and can be 1 or 2 -- must not be mistaken for one.
static int step_failure(int rc) { return rc > 0 ? -rc : rc; }

// One RK2 step of a one-patch solver whose point masses are a Kepler binary: stage 0 (dt formed in the kernel from
// *slot_current, *slot_next zeroed, reduced wavespeed handed to the host), dt = dx / amax * cfl on the host, the
// orbit at the stage time, stage 1 with the wavespeed fold into *slot_next.  `state` holds p^n and receives the
// new state; `other` is the second buffer.  elements4: semimajor axis, total mass, mass ratio, eccentricity;
// shared4: softening length, sink rate, sink radius, sink model.  result3 (host): amax, dt, the time after the
// step.  Returns 0; 1 when amax came out below wavespeed_floor (only stage 0 has been launched: the caller
// recomputes the wavespeed exactly and redoes the step); 2 when amax is NaN (likewise only stage 0: the caller
// fails as the reference does); negative on failure.
int sfhip_plan_step_rk2(int plan, double *state, double *other, double *slot_current, double *slot_next,
                        double dx, double cfl, double time, const double *elements4, const double *shared4,
                        double wavespeed_floor, double floor_fraction, double host_seq, double *result3)
{
    StagePlan *pl = plan_of(plan);
    if (!pl) return -1;
    if (!state || !other || !slot_current || !slot_next || !elements4 || !shared4 || !result3 || host_seq == 0.0)
        return fail(-1, "sfhip_plan_step_rk2: bad arguments");
    using clock = std::chrono::steady_clock;
    const auto c0 = clock::now();
    double body[10], masses[18], next8[8];
    if (kepler_orbital_state(elements4[0], elements4[1], elements4[2], elements4[3], time, body))
        return fail(-1, "sfhip_plan_step_rk2: Kepler's equation did not converge");
    pack_masses(body, shared4, masses);
    // stage 0 (a = 0): p^n -> other; dt from the device slot
    int rc = sfhip_plan_stage(plan, masses, nullptr, nullptr, state, other, 0.0, NAN, 0, pl->ni, 0, 0.0, nullptr,
                              slot_current, dx, cfl, slot_next, 0, host_seq);
    if (rc) return step_failure(rc);
    const auto c1 = clock::now();
    double amax = 0.0;
    rc = sfhip_plan_host_wait(plan, host_seq, &amax);
    if (rc) return step_failure(rc);
    const auto c2 = clock::now();
    result3[0] = amax;
    if (amax < wavespeed_floor) return 1;
    const double dt = dx / amax * cfl;                                  // driver.py:509-510
    // A blown-up state (NaN wavespeed): the reference fails in the Kepler solve of the next stage time (floor of a
    // NaN, kepler.py:356-366).  Nothing more is launched; the caller raises that error.
    if (!(dt == dt)) return 2;
    const double time0 = time;
    const double t1 = time0 * 0.0 + (time + dt) * (1.0 - 0.0);          // cbdiso_2d.py:239 after stage 0
    const double t2 = time0 * 0.5 + (t1 + dt) * (1.0 - 0.5);            // ... after stage 1
    if (kepler_orbital_state(elements4[0], elements4[1], elements4[2], elements4[3], t1, body))
        return fail(-1, "sfhip_plan_step_rk2: Kepler's equation did not converge");
    pack_masses(body, shared4, masses);
    if (kepler_orbital_state(elements4[0], elements4[1], elements4[2], elements4[3], t2, body))
        return fail(-1, "sfhip_plan_step_rk2: Kepler's equation did not converge");
    for (int k = 0; k < 2; ++k) {
        next8[4 * k + 0] = body[5 * k + 1]; next8[4 * k + 1] = body[5 * k + 2];
        next8[4 * k + 2] = body[5 * k + 0]; next8[4 * k + 3] = shared4[0];
    }
    const auto c3 = clock::now();
    // stage 1 (a = 1/2): other -> state in place over p^n (u^n recomputed from it), wavespeed folded into the next slot
    rc = sfhip_plan_stage(plan, masses, next8, state, other, state, 0.5, dt, 0, pl->ni, 1, floor_fraction * amax, slot_next,
                          nullptr, 0.0, 0.0, nullptr, 0, 0.0);
    if (rc) return step_failure(rc);
    const auto c4 = clock::now();
    pl->phase_ns[0] += std::chrono::duration<double, std::nano>(c1 - c0).count();
    pl->phase_ns[1] += std::chrono::duration<double, std::nano>(c2 - c1).count();
    pl->phase_ns[2] += std::chrono::duration<double, std::nano>(c3 - c2).count();
    pl->phase_ns[3] += std::chrono::duration<double, std::nano>(c4 - c3).count();
    pl->steps_timed += 1;
    result3[1] = dt;
    result3[2] = t2;
    return 0;
}

// Mean host time per sfhip_plan_step_rk2 call by phase, in microseconds, since the plan was made or this was last
// called: out5 = stage-0 launch (with the orbit at t), wait for the hand-over, dt + orbits at the stage times,
// stage-1 launch, number of steps.  Resets the counters.
int sfhip_plan_step_timing(int plan, double *out5)
{
    StagePlan *pl = plan_of(plan);
    if (!pl || !out5) return -1;
    const double n = pl->steps_timed > 0 ? (double)pl->steps_timed : 1.0;
    for (int k = 0; k < 4; ++k) { out5[k] = pl->phase_ns[k] / n * 1e-3; pl->phase_ns[k] = 0.0; }
    out5[4] = (double)pl->steps_timed;
    pl->steps_timed = 0;
    return 0;
}

// Block until a stage launched with this `host_seq` has started on the device and handed the reduced wavespeed
// over through the plan's pinned memory (no copy, no event: the host polls two words).  Gives up after 30 s.
int sfhip_plan_host_wait(int plan, double host_seq, double *host_value)
{
    StagePlan *pl = plan_of(plan);
    if (!pl || !host_value) return pl ? fail(-1, "sfhip_plan_host_wait: null pointer") : -1;
    volatile double *word = pl->pinned + 2;
    const auto start = std::chrono::steady_clock::now();
    for (unsigned long spins = 0; word[1] != host_seq; ++spins) {
        if ((spins & 0xfffff) == 0xfffff &&
            std::chrono::steady_clock::now() - start > std::chrono::seconds(30))
            return fail(-1, "sfhip_plan_host_wait: no stage reported within 30 s (was one launched with this host_seq?)");
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    *host_value = word[0];
    return 0;
}


} // extern "C"

// ---------------------------------------------------------------------------
// Step groups: one RK2 step of ALL the patches a process owns from one call, halo exchange included.
//
// sfhip_plan_step_rk2 serves a solver with one patch and no neighbours.  With neighbours every stage used to go
// through Python: two slab copies or a torch.distributed batch of sends and receives, three launches per patch,
// the events that order the communication stream -- 0.1-0.3 ms of host time per step against 0.08 ms of kernels
// on the 512 x 4096 slabs of BASELINE's own metric line (4096^2 over 8 GPUs).  A step group holds the patches'
// plans in global patch order with, for either side of each, what its neighbour is: the domain edge, the next
// patch of the group (a device-to-device slab copy) or a patch of another rank (RCCL), and performs per stage
//     slab copies between the group's patches, event `ready`             (compute stream)
//     the rows that need no halo, every patch                           (compute stream)
//     wait for `ready`; one ncclGroup of all sends and receives          (side stream)
//     the <= 2 + 2 rows per patch that need the halos                   (side stream; compute stream if no RCCL)
//     event `joined`; the compute stream waits for it
// which is the data flow of Solver.advance_rk (solvers/cbd_common.py), launch for launch, with the messages of a
// group posted in the order of their names (destination patch, lower | upper guard slab) as the Python transport
// posts them.  After the last stage the wavespeed slot is MAX-all-reduced over the ranks as ncclUint64: for
// non-negative doubles the order of the bit patterns is the order of the values, and a NaN's pattern ranks above
// all of them, so a blown-up rank gives every rank a NaN dt (ncclMax on ncclDouble may drop it).
//
// RCCL is bound at run time (sfhip_rccl_bind, dlopen of the very library file that made the communicator):
// libsfhip.so itself does not link against it.
// ---------------------------------------------------------------------------
namespace {

struct RcclApi {
    void *handle = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
} g_rccl;

constexpr int NCCL_UINT64 = 5, NCCL_FLOAT64 = 8, NCCL_MAX_OP = 2; // ncclDataType_t / ncclRedOp_t (rccl.h)

int rccl_fail(int status, const char *what)
{
    snprintf(t_error, sizeof(t_error), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(status) : "RCCL error");
    return -1;
}

constexpr int MAX_GROUPS = 64;
constexpr int MAX_GROUP_PATCHES = 64;
constexpr int PEER_EDGE = -1, PEER_LOCAL = -2;

struct StepGroup {
    int used = 0;
    int n = 0;
    int plans[MAX_GROUP_PATCHES];
    int global_id[MAX_GROUP_PATCHES];
    int lower[MAX_GROUP_PATCHES], upper[MAX_GROUP_PATCHES]; // PEER_EDGE, PEER_LOCAL or the owning rank
    void *comm = nullptr;   // ncclComm_t, or null when no patch has a neighbour on another rank
    int reduce_over_ranks = 0;
    int remote = 0;         // some neighbour is reached through RCCL
    hipStream_t compute = nullptr, side = nullptr;
    hipEvent_t ready = nullptr, joined = nullptr;
    double phase_ns[4] = {0, 0, 0, 0};
    long steps_timed = 0;
};

StepGroup g_groups[MAX_GROUPS];

StepGroup *group_of(int id)
{
    if (id < 0 || id >= MAX_GROUPS || !g_groups[id].used) {
        fail(-1, "sfhip_step_group: no such group");
        return nullptr;
    }
    return &g_groups[id];
}

struct HaloOp {
    long name;      // 2 * destination patch + (0 lower | 1 upper guard slab)
    int send;
    double *ptr;
    size_t count;
    int peer;
};

// One RK stage of every patch of the group (see above).  rd / un / wr: per patch; un may be null (a == 0).
int group_stage(StepGroup &G, double *const *rd, double *const *un, double *const *wr, const double *masses,
                const double *next_masses, double a, double dt, int want_wavespeed, double wavespeed_floor,
                double *wavespeed_max, double *amax_device, double dt_dx, double dt_cfl, double *zero_slot, double host_seq)
{
    StagePlan *pl[MAX_GROUP_PATCHES];
    for (int k = 0; k < G.n; ++k)
        if (!(pl[k] = plan_of(G.plans[k]))) return -1;
    // halo slabs between the group's own patches (Solver.start_halo_exchange: the same-process copies)
    for (int k = 0; k < G.n; ++k) {
        const int nq = pl[k]->kind == 1 ? 3 : 4;
        const size_t row = (size_t)(pl[k]->nj + 2 * sf::NG) * nq, slab = sf::NG * row * sizeof(double);
        hipError_t e = hipSuccess;
        if (G.lower[k] == PEER_LOCAL)
            e = hipMemcpyAsync(rd[k], rd[k - 1] + (size_t)pl[k - 1]->ni * row, slab, hipMemcpyDeviceToDevice, G.compute);
        if (e == hipSuccess && G.upper[k] == PEER_LOCAL)
            e = hipMemcpyAsync(rd[k] + (size_t)(pl[k]->ni + sf::NG) * row, rd[k + 1] + (size_t)sf::NG * row, slab,
                               hipMemcpyDeviceToDevice, G.compute);
        if (e != hipSuccess) return fail(-(int)e, hipGetErrorString(e));
    }
    if (G.remote) {
        hipError_t e = hipEventRecord(G.ready, G.compute); // everything the exchange depends on has been queued
        if (e != hipSuccess) return fail(-(int)e, hipGetErrorString(e));
    }
    int lo[MAX_GROUP_PATCHES], hi[MAX_GROUP_PATCHES];
    for (int k = 0; k < G.n; ++k) { // the rows that need no halo (queued before the transport calls: neither waits for the other)
        const int ni = pl[k]->ni;
        lo[k] = G.lower[k] == PEER_EDGE ? 0 : (sf::NG < ni ? sf::NG : ni);
        hi[k] = G.upper[k] == PEER_EDGE ? ni : (ni - sf::NG > lo[k] ? ni - sf::NG : lo[k]);
        int rc = sfhip_plan_stage(G.plans[k], masses, next_masses, un ? un[k] : nullptr, rd[k], wr[k], a, dt, lo[k], hi[k],
                                  want_wavespeed, wavespeed_floor, wavespeed_max, amax_device, dt_dx, dt_cfl, zero_slot, 0,
                                  k == 0 ? host_seq : 0.0);
        if (rc) return rc > 0 ? -rc : rc;
    }
    if (G.remote) {
        hipError_t e = hipStreamWaitEvent(G.side, G.ready, 0);
        if (e != hipSuccess) return fail(-(int)e, hipGetErrorString(e));
        HaloOp ops[4 * MAX_GROUP_PATCHES];
        int nops = 0;
        for (int k = 0; k < G.n; ++k) {
            const int nq = pl[k]->kind == 1 ? 3 : 4;
            const size_t row = (size_t)(pl[k]->nj + 2 * sf::NG) * nq, count = sf::NG * row;
            const long id = G.global_id[k];
            if (G.upper[k] >= 0) {
                ops[nops++] = HaloOp{2 * (id + 1) + 0, 1, rd[k] + (size_t)pl[k]->ni * row, count, G.upper[k]};
                ops[nops++] = HaloOp{2 * id + 1, 0, rd[k] + (size_t)(pl[k]->ni + sf::NG) * row, count, G.upper[k]};
            }
            if (G.lower[k] >= 0) {
                ops[nops++] = HaloOp{2 * (id - 1) + 1, 1, rd[k] + (size_t)sf::NG * row, count, G.lower[k]};
                ops[nops++] = HaloOp{2 * id + 0, 0, rd[k], count, G.lower[k]};
            }
        }
        // between two ranks RCCL matches messages in posting order: post in the order of the names on both sides
        for (int i = 1; i < nops; ++i)
            for (int j = i; j > 0 && ops[j].name < ops[j - 1].name; --j) { HaloOp t = ops[j]; ops[j] = ops[j - 1]; ops[j - 1] = t; }
        int st = g_rccl.GroupStart();
        if (st) return rccl_fail(st, "ncclGroupStart");
        for (int i = 0; i < nops; ++i) {
            st = ops[i].send ? g_rccl.Send(ops[i].ptr, ops[i].count, NCCL_FLOAT64, ops[i].peer, G.comm, G.side)
                             : g_rccl.Recv(ops[i].ptr, ops[i].count, NCCL_FLOAT64, ops[i].peer, G.comm, G.side);
            if (st) { (void)g_rccl.GroupEnd(); return rccl_fail(st, ops[i].send ? "ncclSend" : "ncclRecv"); }
        }
        st = g_rccl.GroupEnd();
        if (st) return rccl_fail(st, "ncclGroupEnd");
    }
    for (int k = 0; k < G.n; ++k) { // the rows behind the halos
        const int ni = pl[k]->ni;
        const int ranges[2][2] = {{0, lo[k]}, {hi[k], ni}};
        for (int s = 0; s < 2; ++s) {
            if (ranges[s][0] >= ranges[s][1]) continue;
            int rc = sfhip_plan_stage(G.plans[k], masses, next_masses, un ? un[k] : nullptr, rd[k], wr[k], a, dt, ranges[s][0],
                                      ranges[s][1], want_wavespeed, wavespeed_floor, wavespeed_max, amax_device, dt_dx, dt_cfl,
                                      zero_slot, G.remote, 0.0);
            if (rc) return rc > 0 ? -rc : rc;
        }
    }
    if (G.remote) {
        hipError_t e = hipEventRecord(G.joined, G.side);
        if (e == hipSuccess) e = hipStreamWaitEvent(G.compute, G.joined, 0); // the stage is complete on both streams
        if (e != hipSuccess) return fail(-(int)e, hipGetErrorString(e));
    }
    return 0;
}

} // namespace

extern "C" {

// Bind the RCCL entry points the step groups call from `library_path` -- the file the caller's communicator was
// made with (a second copy of the library would not know it).  Returns 0, or -1 (sfhip_last_error).
int sfhip_rccl_bind(const char *library_path)
{
    if (!library_path) return fail(-1, "sfhip_rccl_bind: null path");
    void *h = dlopen(library_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        snprintf(t_error, sizeof(t_error), "sfhip_rccl_bind: %s", dlerror());
        return -1;
    }
    RcclApi api;
    api.handle = h;
    api.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    api.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    api.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclSend");
    api.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclRecv");
    api.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
    api.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv || !api.AllReduce || !api.GetErrorString)
        return fail(-1, "sfhip_rccl_bind: the library lacks ncclGroupStart/End, ncclSend/Recv, ncclAllReduce or ncclGetErrorString");
    g_rccl = api;
    return 0;
}

// npatches plans (in global patch order, all on the current device and with the same compute stream), their
// global patch numbers, and per patch what lies below / above it: -1 the domain edge, -2 the previous / next patch
// of this group, >= 0 the rank that owns the neighbouring patch (needs nccl_comm and sfhip_rccl_bind).
// side_stream: where the exchange and the rows behind it run.  reduce_over_ranks: MAX-all-reduce the wavespeed
// slot over nccl_comm after the last stage.  Returns a group id >= 0, or -1.
int sfhip_step_group_create(int npatches, const int *plans, const int *global_ids, const int *lower, const int *upper,
                            void *nccl_comm, int reduce_over_ranks, void *side_stream)
{
    if (npatches < 1 || npatches > MAX_GROUP_PATCHES || !plans || !global_ids || !lower || !upper)
        return fail(-1, "sfhip_step_group_create: bad arguments");
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    int id = 0;
    while (id < MAX_GROUPS && g_groups[id].used) ++id;
    if (id == MAX_GROUPS) return fail(-1, "sfhip_step_group_create: too many groups");
    StepGroup G;
    G.n = npatches;
    for (int k = 0; k < npatches; ++k) {
        StagePlan *pl = plan_of(plans[k]);
        if (!pl) return -1;
        if (k == 0) G.compute = pl->compute;
        else if (pl->compute != G.compute || pl->kind != g_plans[plans[0]].kind || pl->nj != g_plans[plans[0]].nj)
            return fail(-1, "sfhip_step_group_create: the plans must share stream, solver and nj");
        G.plans[k] = plans[k]; G.global_id[k] = global_ids[k]; G.lower[k] = lower[k]; G.upper[k] = upper[k];
        if (lower[k] < PEER_LOCAL || upper[k] < PEER_LOCAL) return fail(-1, "sfhip_step_group_create: bad neighbour code");
        if ((lower[k] == PEER_LOCAL && k == 0) || (upper[k] == PEER_LOCAL && k == npatches - 1))
            return fail(-1, "sfhip_step_group_create: a local neighbour must be a patch of the group");
        if (lower[k] >= 0 || upper[k] >= 0) G.remote = 1;
        if (pl->ni < 2 * sf::NG && (lower[k] != PEER_EDGE || upper[k] != PEER_EDGE))
            return fail(-1, "sfhip_step_group_create: a patch with neighbours needs at least 4 rows");
    }
    if ((G.remote || reduce_over_ranks) && (!nccl_comm || !g_rccl.handle))
        return fail(-1, "sfhip_step_group_create: neighbours on other ranks need a communicator and sfhip_rccl_bind");
    G.comm = nccl_comm;
    G.reduce_over_ranks = reduce_over_ranks;
    G.side = (hipStream_t)side_stream;
    hipError_t e = hipEventCreateWithFlags(&G.ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&G.joined, hipEventDisableTiming);
    if (e != hipSuccess) return fail(-1, hipGetErrorString(e));
    for (int k = 0; k < npatches; ++k) g_plans[plans[k]].side = G.side;
    G.used = 1;
    g_groups[id] = G;
    return id;
}

int sfhip_step_group_destroy(int group)
{
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    StepGroup *G = group_of(group);
    if (!G) return -1;
    (void)hipEventDestroy(G->ready);
    (void)hipEventDestroy(G->joined);
    *G = StepGroup();
    return 0;
}

// sfhip_plan_step_rk2 for a group: `state` / `other` are HOST arrays of the patches' device pointers; everything
// else, the outcomes 1 and 2 included, as there.  With reduce_over_ranks *slot_current must hold the maximum over
// all ranks (as this call leaves *slot_next), so that every rank takes the same decisions.
int sfhip_step_group_rk2(int group, double *const *state, double *const *other, double *slot_current, double *slot_next,
                         double dx, double cfl, double time, const double *elements4, const double *shared4,
                         double wavespeed_floor, double floor_fraction, double host_seq, double *result3)
{
    StepGroup *G = group_of(group);
    if (!G) return -1;
    if (!state || !other || !slot_current || !slot_next || !elements4 || !shared4 || !result3 || host_seq == 0.0)
        return fail(-1, "sfhip_step_group_rk2: bad arguments");
    using clock = std::chrono::steady_clock;
    const auto c0 = clock::now();
    double body[10], masses[18], next8[8];
    if (kepler_orbital_state(elements4[0], elements4[1], elements4[2], elements4[3], time, body))
        return fail(-1, "sfhip_step_group_rk2: Kepler's equation did not converge");
    pack_masses(body, shared4, masses);
    int rc = group_stage(*G, state, nullptr, other, masses, nullptr, 0.0, NAN, 0, 0.0, nullptr, slot_current, dx, cfl,
                         slot_next, host_seq);
    if (rc) return rc;
    const auto c1 = clock::now();
    double amax = 0.0;
    rc = sfhip_plan_host_wait(G->plans[0], host_seq, &amax);
    if (rc) return step_failure(rc);
    const auto c2 = clock::now();
    result3[0] = amax;
    if (amax < wavespeed_floor) return 1;
    const double dt = dx / amax * cfl;                                  // driver.py:509-510
    if (!(dt == dt)) return 2;
    const double time0 = time;
    const double t1 = time0 * 0.0 + (time + dt) * (1.0 - 0.0);          // cbdiso_2d.py:239 after stage 0
    const double t2 = time0 * 0.5 + (t1 + dt) * (1.0 - 0.5);            // ... after stage 1
    if (kepler_orbital_state(elements4[0], elements4[1], elements4[2], elements4[3], t1, body))
        return fail(-1, "sfhip_step_group_rk2: Kepler's equation did not converge");
    pack_masses(body, shared4, masses);
    if (kepler_orbital_state(elements4[0], elements4[1], elements4[2], elements4[3], t2, body))
        return fail(-1, "sfhip_step_group_rk2: Kepler's equation did not converge");
    for (int k = 0; k < 2; ++k) {
        next8[4 * k + 0] = body[5 * k + 1]; next8[4 * k + 1] = body[5 * k + 2];
        next8[4 * k + 2] = body[5 * k + 0]; next8[4 * k + 3] = shared4[0];
    }
    const auto c3 = clock::now();
    // stage 1: other -> state in place over p^n; the new wavespeed accumulates in *slot_next (zeroed by stage 0)
    rc = group_stage(*G, other, state, state, masses, next8, 0.5, dt, 1, floor_fraction * amax, slot_next, nullptr, 0.0, 0.0,
                     nullptr, 0.0);
    if (rc) return rc;
    if (G->reduce_over_ranks) {
        int st = g_rccl.AllReduce(slot_next, slot_next, 1, NCCL_UINT64, NCCL_MAX_OP, G->comm, G->compute);
        if (st) return rccl_fail(st, "ncclAllReduce");
    }
    const auto c4 = clock::now();
    G->phase_ns[0] += std::chrono::duration<double, std::nano>(c1 - c0).count();
    G->phase_ns[1] += std::chrono::duration<double, std::nano>(c2 - c1).count();
    G->phase_ns[2] += std::chrono::duration<double, std::nano>(c3 - c2).count();
    G->phase_ns[3] += std::chrono::duration<double, std::nano>(c4 - c3).count();
    G->steps_timed += 1;
    result3[1] = dt;
    result3[2] = t2;
    return 0;
}

// as sfhip_plan_step_timing, for a group
int sfhip_step_group_timing(int group, double *out5)
{
    StepGroup *G = group_of(group);
    if (!G || !out5) return -1;
    const double n = G->steps_timed > 0 ? (double)G->steps_timed : 1.0;
    for (int k = 0; k < 4; ++k) { out5[k] = G->phase_ns[k] / n * 1e-3; G->phase_ns[k] = 0.0; }
    out5[4] = (double)G->steps_timed;
    G->steps_timed = 0;
    return 0;
}

} // extern "C"

// ---------------------------------------------------------------------------
// time-series diagnostics (Solver.reductions)
// ---------------------------------------------------------------------------
static void reductions_grid(int ni, int nj, dim3 *grid)
{
    const int bx = (nj + 255) / 256;
    int by = ni < 1024 / bx ? ni : 1024 / bx; // about 1024 blocks: 4 per CU
    if (by < 1) by = 1;
    *grid = dim3(bx, by);
}

static int reductions_workspace_len(int ni, int nj)
{
    if (ni < 1 || nj < 1) return 0;
    dim3 grid;
    reductions_grid(ni, nj, &grid);
    return (int)(grid.x * grid.y) * sf::RED_SLOTS;
}

template <int NQ>
static int reductions_impl(const char *name, int ni, int nj, double pxl, double pxr, double pyl, double pyr,
                           const sf::Mass &m1, const sf::Mass &m2, int constant_softening, double gamma,
                           const double *primitive, const double *xcoord, const double *ycoord,
                           int nd, const double *spec, int workspace_len, double *workspace, double *sums)
{
    if (ni < 1 || nj < 1 || !primitive || !xcoord || !ycoord || !spec || !workspace || !sums) return fail(-1, name);
    if (nd < 1 || nd > sf::RED_MAXD) return fail(-1, "reductions: num_diagnostics must be in 1..8 per call");
    if (workspace_len < reductions_workspace_len(ni, nj)) return fail(-1, "reductions: workspace too small");
    sf::ReduceParams P;
    memset(&P, 0, sizeof(P));
    P.ni = ni; P.nj = nj;
    P.xl = pxl; P.yl = pyl;
    P.dx = (pxr - pxl) / ni;
    P.dy = (pyr - pyl) / nj;
    P.m[0] = m1; P.m[1] = m2;
    P.v[0][0] = m1.vx; P.v[0][1] = m1.vy;
    P.v[1][0] = m2.vx; P.v[1][1] = m2.vy;
    P.gamma = gamma;
    P.constant_softening = constant_softening;
    dim3 grid, block(256);
    reductions_grid(ni, nj, &grid);
    hipLaunchKernelGGL(sf::reduce_kernel<NQ>, grid, block, 0, t_stream, P, primitive, xcoord, ycoord, spec, nd, workspace);
    int rc = check_launch(name);
    if (rc) return rc;
    hipLaunchKernelGGL(sf::reduce_final_kernel, dim3(1), dim3(256), 0, t_stream,
                       (const double *)workspace, (int)(grid.x * grid.y), 2 * nd, sums);
    return check_launch(name);
}

extern "C" {

int sfhip_reductions_workspace_len(int ni, int nj) { return reductions_workspace_len(ni, nj); }

int sfhip_cbdiso_2d_reductions(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double *primitive, double *xcoord, double *ycoord, int num_diagnostics, double *spec,
    int workspace_len, double *workspace, double *sums)
{
    return reductions_impl<3>("sfhip_cbdiso_2d_reductions: bad arguments", ni, nj, patch_xl, patch_xr, patch_yl, patch_yr,
        make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
        make_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
        1, 0.0, primitive, xcoord, ycoord, num_diagnostics, spec, workspace_len, workspace, sums);
}

int sfhip_cbdgam_2d_reductions(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    int constant_softening, double gamma_law_index,
    double *primitive, double *xcoord, double *ycoord, int num_diagnostics, double *spec,
    int workspace_len, double *workspace, double *sums)
{
    return reductions_impl<4>("sfhip_cbdgam_2d_reductions: bad arguments", ni, nj, patch_xl, patch_xr, patch_yl, patch_yr,
        make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
        make_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
        constant_softening, gamma_law_index, primitive, xcoord, ycoord, num_diagnostics, spec,
        workspace_len, workspace, sums);
}

} // extern "C"

// ---------------------------------------------------------------------------
// The entry point that allocates nothing (no counterpart in the reference; for callers that cannot call
// hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid it -- nor bring a
// workspace).  sfhip_cbdiso_2d_reductions with its block partials in a file-scope device array instead of a
// caller-provided workspace.  Limit: the launch may have at most NOALLOC_MAX_BLOCKS blocks (any ni; nj up to
// 262 144), 128 KiB of device memory per GPU, held until the process ends.
//
// `noalloc_scratch` is a host-side stand-in in host code, not a device address: the address of each GPU's own
// copy is looked up once per device with hipGetSymbolAddress (with that device current) and kept; kernels,
// and anything else on the host that needs the memory, get that pointer.  Every call on a GPU shares that
// GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls on one stream do not.
// Nothing has to start at zero -- every block writes all of its slots before the second kernel reads them --
// so the array keeps the previous call's contents and is not cleared.
// ---------------------------------------------------------------------------
constexpr int NOALLOC_MAX_BLOCKS = 1024;
static __device__ __align__(256) double noalloc_scratch[NOALLOC_MAX_BLOCKS * sf::RED_SLOTS];

namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
double *g_noalloc_ptr[NOALLOC_MAX_DEVICES]; // per device: that GPU's noalloc_scratch
std::mutex g_noalloc_mutex;

int noalloc_scratch_ptr(double **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return fail(-1, "sfhip_cbdiso_2d_reductions_noalloc: device index out of range");
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_ptr[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(noalloc_scratch)); // the current device's copy
        if (e != hipSuccess || !ptr) {
            snprintf(t_error, sizeof(t_error), "hipGetSymbolAddress(noalloc_scratch) on device %d: %s", dev,
                     e != hipSuccess ? hipGetErrorString(e) : "null address");
            return e != hipSuccess ? (int)e : -1;
        }
        g_noalloc_ptr[dev] = (double *)ptr;
    }
    *out = g_noalloc_ptr[dev];
    return 0;
}
} // namespace

extern "C" int sfhip_cbdiso_2d_reductions_noalloc(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double *primitive, double *xcoord, double *ycoord, int num_diagnostics, double *spec, double *sums)
{
    const int need = reductions_workspace_len(ni, nj);
    if (need < 1) return fail(-1, "sfhip_cbdiso_2d_reductions_noalloc: ni, nj must be >= 1");
    if (need > NOALLOC_MAX_BLOCKS * sf::RED_SLOTS) {
        snprintf(t_error, sizeof(t_error),
                 "sfhip_cbdiso_2d_reductions_noalloc: a %d x %d patch needs %d doubles of block partials, the built-in "
                 "scratch array holds %d (nj <= 262144); use sfhip_cbdiso_2d_reductions with a workspace",
                 ni, nj, need, NOALLOC_MAX_BLOCKS * sf::RED_SLOTS);
        return -1;
    }
    double *scratch = nullptr;
    int rc = noalloc_scratch_ptr(&scratch);
    if (rc) return rc;
    return reductions_impl<3>("sfhip_cbdiso_2d_reductions_noalloc: bad arguments", ni, nj, patch_xl, patch_xr, patch_yl,
        patch_yr,
        make_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
        make_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
        1, 0.0, primitive, xcoord, ycoord, num_diagnostics, spec, NOALLOC_MAX_BLOCKS * sf::RED_SLOTS, scratch, sums);
}


// ---------------------------------------------------------------------------
// srhd_2d -- reference-named entry points
// ---------------------------------------------------------------------------
namespace {

constexpr int SR_MAX_DEVICES = 64;
constexpr int SR_MAX_TABLES = 32;
thread_local double t_sr_theta = 1.5;         // srhd_2d.py:66 (Options.plm_theta)
thread_local double t_sr_mach_ceiling = 1e6;  // srhd_2d.py:67

// Per-device state of the srhd entry points.  The first call for a (device, nj, polar_extent) allocates
// and fills a polar trigonometry table -- on the host with libm, so that zone volumes and face areas carry
// exactly the reference's bits -- and the first call on a device allocates the conserved_to_primitive failure
// counter; those first calls block (hipMalloc, synchronous copy).  Tables are kept until the process ends
// (a solver per mesh, a handful of meshes per process), so steady-state calls neither allocate nor
// synchronise, and two solvers with different meshes do not evict each other.  The cache is guarded by a
// mutex; the stream and error buffer are per thread.
struct SrhdTable {
    int nj;
    double polar_extent;
    double *trig; // sin(dq j), j = 0..nj, then cos(dq j), j = 0..nj
};
struct SrhdDeviceState {
    unsigned int *failures = nullptr;
    int ntables = 0;
    SrhdTable tables[SR_MAX_TABLES];
};
SrhdDeviceState g_sr[SR_MAX_DEVICES];
std::mutex g_sr_mutex;

int srhd_state(int nj, double polar_extent, SrhdDeviceState **out, const double **trig)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    if (dev < 0 || dev >= SR_MAX_DEVICES) return fail(-1, "srhd_2d: device index out of range");
    std::lock_guard<std::mutex> lock(g_sr_mutex);
    SrhdDeviceState &st = g_sr[dev];
    if (!st.failures) {
        e = hipMalloc((void **)&st.failures, sizeof(unsigned int));
        if (e == hipSuccess) e = hipMemset(st.failures, 0, sizeof(unsigned int));
        if (e != hipSuccess) { st.failures = nullptr; return fail((int)e, hipGetErrorString(e)); }
    }
    *out = &st;
    for (int k = 0; k < st.ntables; ++k) {
        if (st.tables[k].nj == nj && memcmp(&st.tables[k].polar_extent, &polar_extent, sizeof(double)) == 0) {
            *trig = st.tables[k].trig;
            return 0;
        }
    }
    if (st.ntables == SR_MAX_TABLES) return fail(-1, "srhd_2d: too many distinct (nj, polar_extent) meshes in one process");
    const size_t n = (size_t)nj + 1;
    double *host = (double *)malloc(2 * n * sizeof(double));
    if (!host) return fail(-1, "srhd_2d: out of host memory");
    const double dq = polar_extent / nj; // srhd_2d.c:524
    for (size_t j = 0; j < n; ++j) {
        const double q = dq * (double)j; // dq * (j + 0) and dq * (j + 1) of the reference
        host[j] = sin(q);
        host[n + j] = cos(q);
    }
    double *table = nullptr;
    e = hipMalloc((void **)&table, 2 * n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(table, host, 2 * n * sizeof(double), hipMemcpyHostToDevice);
    free(host);
    if (e != hipSuccess) {
        if (table) (void)hipFree(table);
        return fail((int)e, hipGetErrorString(e));
    }
    st.tables[st.ntables++] = SrhdTable{nj, polar_extent, table};
    *trig = table;
    return 0;
}

int srhd_fill(sf::SrhdParams &P, int ni, int nj, const double *faces, double polar_extent, SrhdDeviceState **st)
{
    if (ni < 1 || nj < 1 || !faces) return fail(-1, "srhd_2d: bad shape or null face_positions");
    const double *trig = nullptr;
    int rc = srhd_state(nj, polar_extent, st, &trig);
    if (rc) return rc;
    memset(&P, 0, sizeof(P));
    P.ni = ni; P.nj = nj;
    P.polar_extent = polar_extent;
    P.dq = polar_extent / nj;
    P.theta = t_sr_theta;
    P.mach_ceiling2 = pow(t_sr_mach_ceiling, 2.0); // srhd_2d.c:156, host libm as in the reference
    P.faces = faces;
    P.sinq = trig;
    P.cosq = trig + (nj + 1);
    return 0;
}

} // namespace

extern "C" {

int sfhip_srhd_2d_set_macros(int n, double plm_theta, double mach_ceiling)
{
    if (n != 1 || !(plm_theta >= 1.0 && plm_theta <= 2.0) || !(mach_ceiling > 0.0))
        return fail(-1, "sfhip_srhd_2d_set_macros: need n == 1, 1 <= plm_theta <= 2, mach_ceiling > 0");
    t_sr_theta = plm_theta;
    t_sr_mach_ceiling = mach_ceiling;
    return 0;
}

int srhd_2d_primitive_to_conserved(int ni, int nj, double *face_positions, double *primitive, double *conserved,
                                   double polar_extent, double scale_factor)
{
    if (!primitive || !conserved) return fail(-1, "srhd_2d_primitive_to_conserved: null array");
    sf::SrhdParams P;
    SrhdDeviceState *st;
    int rc = srhd_fill(P, ni, nj, face_positions, polar_extent, &st);
    if (rc) return rc;
    P.scale_factor = scale_factor;
    dim3 block(256), grid((nj + 255) / 256, ni);
    hipLaunchKernelGGL(sf::srhd_p2c_kernel, grid, block, 0, t_stream, P, primitive, conserved);
    return check_launch("srhd_2d_primitive_to_conserved");
}

int srhd_2d_conserved_to_primitive(int ni, int nj, double *face_positions, double *conserved1, double *conserved2,
                                   double *primitive, double polar_extent, double scale_factor)
{
    if (!conserved1 || !conserved2 || !primitive) return fail(-1, "srhd_2d_conserved_to_primitive: null array");
    sf::SrhdParams P;
    SrhdDeviceState *st;
    int rc = srhd_fill(P, ni, nj, face_positions, polar_extent, &st);
    if (rc) return rc;
    P.scale_factor = scale_factor;
    dim3 block(256), grid((nj + 255) / 256, ni);
    hipLaunchKernelGGL(sf::srhd_c2p_kernel, grid, block, 0, t_stream, P, conserved1, conserved2, primitive, st->failures);
    return check_launch("srhd_2d_conserved_to_primitive");
}

int srhd_2d_max_wavespeeds(int ni, int nj, double *face_positions, double *primitive, double *wavespeed, double adot)
{
    if (ni < 1 || nj < 1 || !face_positions || !primitive || !wavespeed) return fail(-1, "srhd_2d_max_wavespeeds: bad arguments");
    sf::SrhdParams P;
    memset(&P, 0, sizeof(P));
    P.ni = ni; P.nj = nj; P.faces = face_positions; P.adot = adot;
    dim3 block(256), grid((nj + 255) / 256, ni);
    hipLaunchKernelGGL(sf::srhd_wavespeed_kernel, grid, block, 0, t_stream, P, primitive, wavespeed);
    return check_launch("srhd_2d_max_wavespeeds");
}

static int srhd_advance_impl(int ni, int nj, double *face_positions, double *conserved_rk, double *primitive_rd,
                             double *conserved_rd, double *conserved_wr, double polar_extent, double a0, double adot,
                             double time, double rk_param, double dt, double jet_mdot, double jet_gamma_beta,
                             double jet_theta, double jet_duration, int num_first_order_zones,
                             bool fuse, double scale_factor_next, double *primitive_wr, double *conserved_floored)
{
    if (!conserved_rk || !primitive_rd || !conserved_rd || !conserved_wr) return fail(-1, "srhd_2d_advance_rk: null array");
    if (conserved_wr == conserved_rd || conserved_wr == conserved_rk)
        return fail(-1, "srhd_2d_advance_rk: conserved_wr must not alias an input");
    if (fuse) {
        if (!primitive_wr || !conserved_floored) return fail(-1, "sfhip_srhd_2d_stage: null output array");
        if (primitive_wr == primitive_rd) return fail(-1, "sfhip_srhd_2d_stage: primitive_wr must not alias primitive_rd");
        if (conserved_floored == conserved_rd || conserved_floored == conserved_rk || conserved_floored == conserved_wr)
            return fail(-1, "sfhip_srhd_2d_stage: conserved_floored must not alias another conserved array");
    }
    sf::SrhdParams P;
    SrhdDeviceState *st;
    int rc = srhd_fill(P, ni, nj, face_positions, polar_extent, &st);
    if (rc) return rc;
    P.a0 = a0; P.adot = adot; P.time = time; P.rk_param = rk_param; P.dt = dt;
    P.jet_mdot = jet_mdot; P.jet_gamma_beta = jet_gamma_beta; P.jet_theta = jet_theta; P.jet_duration = jet_duration;
    P.num_first_order_zones = num_first_order_zones;
    P.fuse_c2p = fuse ? 1 : 0;
    P.scale_factor_next = scale_factor_next;
    P.prim_wr = primitive_wr;
    P.cons_floored = conserved_floored;
    P.failures = st->failures;
    if (!fuse && env_int("SFHIP_SRHD_ZONE_KERNEL", 0)) { // thread-per-zone form (every face twice), kept for A/B timing
        dim3 block(256), grid((nj + 255) / 256, ni);
        hipLaunchKernelGGL(sf::srhd_advance_kernel, grid, block, 0, t_stream, P, conserved_rk, primitive_rd,
                           conserved_rd, conserved_wr);
        return check_launch("srhd_2d_advance_rk");
    }
    const int nstrips = (nj + sf::STRIP - 1) / sf::STRIP;
    int rows_per_chunk, nchunks;
    plan_rows(ni, nstrips, &rows_per_chunk, &nchunks, 9216);
    const long waves = (long)nstrips * nchunks;
    const int wpb = waves_per_block();
    dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(64 * wpb);
    if (P.num_first_order_zones > 0)
        hipLaunchKernelGGL((sf::srhd_march_kernel<true>), grid, block, 0, t_stream, P, conserved_rk, primitive_rd, conserved_rd,
                       conserved_wr, rows_per_chunk, nstrips, nchunks);
    else
        hipLaunchKernelGGL((sf::srhd_march_kernel<false>), grid, block, 0, t_stream, P, conserved_rk, primitive_rd, conserved_rd,
                       conserved_wr, rows_per_chunk, nstrips, nchunks);
    return check_launch("srhd_2d_advance_rk");
}

int srhd_2d_advance_rk(int ni, int nj, double *face_positions, double *conserved_rk, double *primitive_rd,
                       double *conserved_rd, double *conserved_wr, double polar_extent, double a0, double adot,
                       double time, double rk_param, double dt, double jet_mdot, double jet_gamma_beta,
                       double jet_theta, double jet_duration, int num_first_order_zones)
{
    return srhd_advance_impl(ni, nj, face_positions, conserved_rk, primitive_rd, conserved_rd, conserved_wr,
                             polar_extent, a0, adot, time, rk_param, dt, jet_mdot, jet_gamma_beta, jet_theta,
                             jet_duration, num_first_order_zones, false, 0.0, nullptr, nullptr);
}

int sfhip_srhd_2d_stage(int ni, int nj, double *face_positions, double *conserved_rk, double *primitive_rd,
                        double *conserved_rd, double *conserved_wr, double polar_extent, double a0, double adot,
                        double time, double rk_param, double dt, int num_first_order_zones,
                        double scale_factor_next, double *primitive_wr, double *conserved_floored)
{
    return srhd_advance_impl(ni, nj, face_positions, conserved_rk, primitive_rd, conserved_rd, conserved_wr,
                             polar_extent, a0, adot, time, rk_param, dt, 0.0, 0.0, 0.0, 0.0,
                             num_first_order_zones, true, scale_factor_next, primitive_wr, conserved_floored);
}

int sfhip_srhd_2d_failures(int n, double *out)
{
    if (n != 1 || !out) return fail(-1, "sfhip_srhd_2d_failures: bad arguments");
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    unsigned int count = 0;
    if (dev >= 0 && dev < SR_MAX_DEVICES && g_sr[dev].failures) {
        e = hipStreamSynchronize(t_stream);
        if (e == hipSuccess) e = hipMemcpy(&count, g_sr[dev].failures, sizeof(count), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemset(g_sr[dev].failures, 0, sizeof(count));
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    const double value = (double)count;
    e = hipMemcpy(out, &value, sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

} // extern "C"

// ---------------------------------------------------------------------------
// cbdisodg_2d -- reference-named entry points and their clamped / fused forms
// ---------------------------------------------------------------------------
namespace {

sf::DgMass dg_mass(double x, double y, double vx, double vy, double mass, double rsoft, double rate, double rsink, int model)
{
    sf::DgMass m;
    m.x = x; m.y = y; m.vx = vx; m.vy = vy; m.mass = mass; m.rsoft = rsoft;
    m.sink_rate = rate; m.sink_radius = rsink; m.sink_model = model; m.pad_ = 0;
    return m;
}

int dg_mesh(sf::DgParams &P, const char *who, int ni, int nj, double xl, double xr, double yl, double yr)
{
    if (ni < 1 || nj < 1) return fail(-1, who);
    memset(&P, 0, sizeof(P));
    P.ni = ni; P.nj = nj;
    P.xl = xl; P.yl = yl;
    P.dx = (xr - xl) / ni; // cbdisodg_2d.c:650-652
    P.dy = (yr - yl) / nj;
    P.cell_volume = P.dx * P.dy;
    return 0;
}

int dg_stage_impl(
    const char *who, int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double *weights0, double *weights1, double *weights2,
    double buffer_surface_density, double buffer_central_mass, double buffer_driving_rate,
    double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled,
    const sf::DgMass &m1, const sf::DgMass &m2, double soundspeed2, double mach_squared, int eos_type,
    double nu, double rk_param, double dt, double velocity_ceiling,
    int clamp_lower_i, int clamp_upper_i, int clamp_j, int i_begin, int i_end, int want_wavespeed,
    const double *wmass8, double *wavespeed_max, double *dt_device)
{
    sf::DgParams P;
    int rc = dg_mesh(P, who, ni, nj, patch_xl, patch_xr, patch_yl, patch_yr);
    if (rc) return rc;
    if (!weights0 || !weights1 || !weights2) return fail(-1, who);
    if (weights2 == weights1 || weights2 == weights0) return fail(-1, "cbdisodg_2d: weights2 must not alias an input");
    if (i_begin < 0 || i_end > ni || i_begin > i_end) return fail(-1, "cbdisodg_2d: bad row range");
    if (want_wavespeed < 0 || want_wavespeed > 2 || (want_wavespeed && !wavespeed_max)) return fail(-1, who);
    P.buffer_on = buffer_is_enabled;
    P.eos = eos_type;
    P.buf_sd = buffer_surface_density;
    P.buf_mass = buffer_central_mass;
    P.buf_onset = buffer_outer_radius - buffer_onset_width;                                 // cbdisodg_2d.c:206
    P.buf_rate_prefix = buffer_driving_rate * sqrt(buffer_central_mass * pow(P.buf_onset, -3.0)); // :214-215, host libm
    P.buf_span = buffer_outer_radius - P.buf_onset;
    P.m[0] = m1; P.m[1] = m2;
    P.cs2 = soundspeed2; P.mach2 = mach_squared;
    P.nu = nu; P.rk = rk_param; P.dt = dt; P.vmax = velocity_ceiling;
    P.clamp_lo = clamp_lower_i; P.clamp_hi = clamp_upper_i; P.clamp_j = clamp_j;
    P.i_begin = i_begin; P.i_end = i_end; P.want_ws = want_wavespeed ? 1 : 0;
    if (wmass8) {
        for (int p = 0; p < 2; ++p) {
            P.wx[p] = wmass8[4 * p + 0]; P.wy[p] = wmass8[4 * p + 1];
            P.wm[p] = wmass8[4 * p + 2]; P.wrs[p] = wmass8[4 * p + 3];
        }
    }
    P.dt_ptr = dt_device;
    if (want_wavespeed == 2) {
        hipError_t e = hipMemsetAsync(wavespeed_max, 0, sizeof(double), t_stream);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    if (i_end == i_begin) return 0;
    const bool zone_kernel = env_int("SFHIP_DG_ZONE_KERNEL", 0) != 0; // rounds 1-2: every zone solves its four faces (read per call: tests switch)
    if (zone_kernel) {
        dim3 block(64), grid((nj + 63) / 64, i_end - i_begin);
        hipLaunchKernelGGL(sf::dg_advance_zone_kernel, grid, block, 0, t_stream, P, weights0, weights1, weights2, wavespeed_max);
        return check_launch(who);
    }
    // one wave per 63 columns x rows_per_wave rows (the 64th lane only solves faces).  A long march solves fewer faces
    // twice (6 + 3 / rows face nodes per zone); a patch with few rows keeps enough waves to fill the device.
    const int nstrips = (nj + 62) / 63, nrows = i_end - i_begin;
    int rpw = env_int("SFHIP_DG_ROWS_PER_WAVE", 0);
    if (rpw <= 0) {
        rpw = (int)((long)nrows * nstrips / 16384); // (2048^2: 2, 4 and 8 rows run alike, 16 leave too few waves)
        rpw = rpw < 1 ? 1 : rpw > 8 ? 8 : rpw;
    }
    P.rows_per_wave = rpw;
    dim3 block(64), grid(nstrips, (nrows + rpw - 1) / rpw);
    // instantiations by what the host knows of the physics: viscosity on; locally isothermal gas with both masses positive
    const bool visc = P.nu > 0.0, lociso = P.eos == 2 && P.m[0].mass > 0.0 && P.m[1].mass > 0.0;
    auto kernel = visc ? (lociso ? sf::dg_advance_kernel<true, true> : sf::dg_advance_kernel<true, false>)
                       : (lociso ? sf::dg_advance_kernel<false, true> : sf::dg_advance_kernel<false, false>);
    hipLaunchKernelGGL(kernel, grid, block, 0, t_stream, P, weights0, weights1, weights2, wavespeed_max);
    return check_launch(who);
}

int dg_wavespeed_impl(const char *who, int ni, int nj, double xl, double xr, double yl, double yr, double soundspeed2,
                      double mach_squared, int eos_type, const double *wmass8, double velocity_ceiling,
                      double *weights, double *wavespeed, double *wavespeed_max)
{
    sf::DgParams P;
    int rc = dg_mesh(P, who, ni, nj, xl, xr, yl, yr);
    if (rc) return rc;
    if (!weights || (!wavespeed && !wavespeed_max)) return fail(-1, who);
    P.cs2 = soundspeed2; P.mach2 = mach_squared; P.eos = eos_type; P.vmax = velocity_ceiling;
    for (int p = 0; p < 2; ++p) {
        P.wx[p] = wmass8[4 * p + 0]; P.wy[p] = wmass8[4 * p + 1];
        P.wm[p] = wmass8[4 * p + 2]; P.wrs[p] = wmass8[4 * p + 3];
    }
    if (wavespeed_max) {
        hipError_t e = hipMemsetAsync(wavespeed_max, 0, sizeof(double), t_stream);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    dim3 block(64), grid((nj + 63) / 64, ni);
    hipLaunchKernelGGL(sf::dg_wavespeed_kernel, grid, block, 0, t_stream, P, weights, wavespeed, wavespeed_max);
    return check_launch(who);
}

int dg_limit_impl(const char *who, int ni, int nj, double *weights1, double *weights2, int clamp_lower_i,
                  int clamp_upper_i, int clamp_j, int i_begin, int i_end)
{
    if (ni < 1 || nj < 1 || !weights1 || !weights2 || weights1 == weights2) return fail(-1, who);
    if (i_begin < 0 || i_end > ni || i_begin > i_end) return fail(-1, "cbdisodg_2d: bad row range");
    if (i_end == i_begin) return 0;
    dim3 block(64, sf::DG_LIMIT_ROWS), grid((nj + 63) / 64, (i_end - i_begin + sf::DG_LIMIT_ROWS - 1) / sf::DG_LIMIT_ROWS);
    hipLaunchKernelGGL(sf::dg_limit_kernel, grid, block, 0, t_stream, ni, nj, weights1, weights2, clamp_lower_i,
                       clamp_upper_i, clamp_j, i_begin, i_end);
    return check_launch(who);
}

} // namespace

extern "C" {

int cbdisodg_2d_slope_limit(int ni, int nj, double, double, double, double, double, double, double, double,
                            double *weights1, double *weights2)
{
    return dg_limit_impl("cbdisodg_2d_slope_limit: bad arguments", ni, nj, weights1, weights2, 0, 0, 0, 0, ni);
}

int sfhip_cbdisodg_2d_limit(int ni, int nj, double *weights1, double *weights2, int clamp_lower_i, int clamp_upper_i,
                            int clamp_j, int i_begin, int i_end)
{
    return dg_limit_impl("sfhip_cbdisodg_2d_limit: bad arguments", ni, nj, weights1, weights2, clamp_lower_i,
                         clamp_upper_i, clamp_j, i_begin, i_end);
}

int cbdisodg_2d_advance_rk(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double *weights0, double *weights1, double *weights2,
    double buffer_surface_density, double buffer_central_mass, double buffer_driving_rate,
    double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double soundspeed2, double mach_squared, int eos_type, double nu, double rk_param, double dt,
    double velocity_ceiling)
{
    return dg_stage_impl("cbdisodg_2d_advance_rk: bad arguments", ni, nj, patch_xl, patch_xr, patch_yl, patch_yr,
        weights0, weights1, weights2, buffer_surface_density, buffer_central_mass, buffer_driving_rate,
        buffer_outer_radius, buffer_onset_width, buffer_is_enabled,
        dg_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
        dg_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
        soundspeed2, mach_squared, eos_type, nu, rk_param, dt, velocity_ceiling,
        0, 0, 0, 0, ni, 0, nullptr, nullptr, nullptr);
}

int sfhip_cbdisodg_2d_stage(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double *weights0, double *weights1, double *weights2,
    double buffer_surface_density, double buffer_central_mass, double buffer_driving_rate,
    double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double soundspeed2, double mach_squared, int eos_type, double nu, double rk_param, double dt,
    double velocity_ceiling,
    int clamp_lower_i, int clamp_upper_i, int clamp_j, int i_begin, int i_end, int want_wavespeed,
    double wx1, double wy1, double wmass1, double wsoftening_length1,
    double wx2, double wy2, double wmass2, double wsoftening_length2,
    double *wavespeed_max, double *dt_device)
{
    const double w[8] = {wx1, wy1, wmass1, wsoftening_length1, wx2, wy2, wmass2, wsoftening_length2};
    return dg_stage_impl("sfhip_cbdisodg_2d_stage: bad arguments", ni, nj, patch_xl, patch_xr, patch_yl, patch_yr,
        weights0, weights1, weights2, buffer_surface_density, buffer_central_mass, buffer_driving_rate,
        buffer_outer_radius, buffer_onset_width, buffer_is_enabled,
        dg_mass(x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1),
        dg_mass(x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2),
        soundspeed2, mach_squared, eos_type, nu, rk_param, dt, velocity_ceiling,
        clamp_lower_i, clamp_upper_i, clamp_j, i_begin, i_end, want_wavespeed, w, wavespeed_max, dt_device);
}

int cbdisodg_2d_wavespeed(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double soundspeed2, double mach_squared, int eos_type,
    double x1, double y1, double, double, double mass1, double softening_length1, double, double, int,
    double x2, double y2, double, double, double mass2, double softening_length2, double, double, int,
    double velocity_ceiling, double *weights, double *wavespeed)
{
    const double w[8] = {x1, y1, mass1, softening_length1, x2, y2, mass2, softening_length2};
    return dg_wavespeed_impl("cbdisodg_2d_wavespeed: bad arguments", ni, nj, patch_xl, patch_xr, patch_yl, patch_yr,
                             soundspeed2, mach_squared, eos_type, w, velocity_ceiling, weights, wavespeed, nullptr);
}

int sfhip_cbdisodg_2d_max_wavespeed(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double soundspeed2, double mach_squared, int eos_type,
    double x1, double y1, double mass1, double softening_length1,
    double x2, double y2, double mass2, double softening_length2,
    double velocity_ceiling, double *weights, double *wavespeed_max)
{
    const double w[8] = {x1, y1, mass1, softening_length1, x2, y2, mass2, softening_length2};
    if (!wavespeed_max) return fail(-1, "sfhip_cbdisodg_2d_max_wavespeed: null wavespeed_max");
    return dg_wavespeed_impl("sfhip_cbdisodg_2d_max_wavespeed: bad arguments", ni, nj, patch_xl, patch_xr, patch_yl,
                             patch_yr, soundspeed2, mach_squared, eos_type, w, velocity_ceiling, weights, nullptr,
                             wavespeed_max);
}

} // extern "C"
