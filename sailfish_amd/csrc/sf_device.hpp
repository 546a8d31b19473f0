// sf_device.hpp -- device-side building blocks shared by the cbdiso_2d and
// cbdgam_2d kernels (gfx950 / CDNA4, wave64, fp64 throughout).
//
// Written from the algorithm description in SURVEY.md §8(a); reference line
// numbers are given so each piece can be checked against
// /root/reference/sailfish/solvers/cbd{iso,gam}_2d.c.
#pragma once
#include <hip/hip_runtime.h>

namespace sf {

constexpr int NG = 2;          // guard zones per side (cbdiso_2d.c:443)
constexpr int WAVE = 64;       // CDNA wavefront
constexpr int HALO = 2;        // lanes lost on each side of a wave's column strip
constexpr int STRIP = WAVE - 2 * HALO; // output columns per wave (60)

// Point mass as the reference passes it: 8 doubles + sink model (cbdiso_2d.c:42-52).
struct Mass {
    double x, y, vx, vy, mass, rsoft, sink_rate, sink_radius;
    int sink_model;
    int pad_;
};

// ---------------------------------------------------------------------------
// Cross-lane moves.  A wave marches down the rows of a 64-column strip with one
// column per lane, so "the zone at j-1 / j+1" is the neighbouring lane.  gfx9
// DPP wave shifts move a dword one lane up or down the whole wave in a single
// VALU op (no LDS traffic); lanes shifted in at the wave ends keep their own
// value and are never used (HALO lanes).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double from_lane_below(double x) // value held by lane-1
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_lane_above(double x) // value held by lane+1
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// C `?:` min/max of the reference (cbdiso_2d.c:16-17); NOT fmin/fmax.
__device__ __forceinline__ double min2(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double max2(double a, double b) { return a > b ? a : b; }

// Generalised minmod slope, cbdiso_2d.c:23-29 / cbdgam_2d.c:27-33.
// The reference's 0.25*|sgn a + sgn b|*(sgn a + sgn c)*min|.| is +-min when the
// three sign bits agree and a signed zero otherwise; all factors are powers of
// two, so selecting on the sign bits gives the identical value.
template <int THETA10> // theta * 10 as an integer template constant
__device__ __forceinline__ double plm(double yl, double y0, double yr)
{
    constexpr double theta = THETA10 / 10.0;
    double a = (y0 - yl) * theta;
    double b = (yr - yl) * 0.5;
    double c = (yr - y0) * theta;
    double m = min2(fabs(a), min2(fabs(b), fabs(c)));
    int sa = __double2hiint(a), sb = __double2hiint(b), sc = __double2hiint(c);
    bool same = ((sa ^ sb) | (sa ^ sc)) >= 0; // sign bit clear <=> all three signs equal
    return same ? copysign(m, a) : 0.0;
}

// max of a non-negative double over the wave, result valid in every lane
__device__ __forceinline__ double wave_max_nonneg(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// Non-negative doubles order like their bit patterns, so a 64-bit integer
// atomic max implements the floating-point max exactly (bit-exact dt).
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v)
{
    atomicMax(reinterpret_cast<unsigned long long *>(addr),
              static_cast<unsigned long long>(__double_as_longlong(v)));
}

// primitive_max_wavespeed, cbdiso_2d.c:340-350 == cbdgam_2d.c:401-409
__device__ __forceinline__ double max_speed(double vx, double vy, double cs2)
{
    double cs = sqrt(cs2);
    double ax = max2(fabs(vx - cs), fabs(vx + cs));
    double ay = max2(fabs(vy - cs), fabs(vy + cs));
    return max2(ax, ay);
}

} // namespace sf
