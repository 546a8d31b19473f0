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

constexpr int ROW_ENTRY = 6;        // doubles per entry of a marching wave's row table (5 used; 48-byte entries)
constexpr int ROW_TABLE_SPAN = 60;  // rows between two refills of the 64-entry table (an iteration looks back 2 rows)

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
// VALU op (no LDS traffic); the lane at the wave end that has no source lane reads
// zero (bound_ctrl, so the move needs no prior copy of the destination) and its
// value is never used (HALO lanes).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double from_lane_below(double x) // value held by lane-1
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_lane_above(double x) // value held by lane+1
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// (The same moves through the LDS crossbar -- ds_bpermute_b32, not a VALU instruction -- were measured: 2-3 %
// slower; the stage kernels do not have the lgkm latency to spare at three waves per SIMD.)

// A mesh coordinate x0 + idx * dx in the reference's two roundings (cbdiso_2d.c:446-454: plain C, no fused
// multiply-add), also where the caller's other arithmetic is allowed to contract.
__device__ __forceinline__ double mesh_coord(double x0, double idx, double dx)
{
#pragma clang fp contract(off)
    const double t = idx * dx;
    return x0 + t;
}

// A wave-uniform double moved to scalar registers: a plain operand of vector instructions from then on.
__device__ __forceinline__ double uniform_value(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)),
                            __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// dt = dx / amax * cfl in exactly the reference driver's operation order (driver.py:509-510), IEEE division,
// nothing to contract; every lane computes the same value, which then lives in scalar registers.
__device__ __forceinline__ double device_timestep(double dx, double amax, double cfl)
{
#pragma clang fp contract(off)
    const double q = dx / amax;
    return uniform_value(q * cfl);
}

// What one lane of a launch does besides the update when the step protocol asks for it (see IsoParams): zero
// the slot the next wavespeed fold accumulates into, and hand the current reduced wavespeed to the host
// through pinned memory: value first, then, after a system-scope fence, the sequence number the host waits for.
__device__ __forceinline__ void step_bookkeeping(double *zero_ptr, double *host_out, double host_seq, const double *amax_ptr)
{
    if (zero_ptr) *zero_ptr = 0.0;
    if (host_out) {
        volatile double *out = host_out;
        out[0] = *amax_ptr;
        __threadfence_system();
        out[1] = host_seq;
    }
}

// the same as a launch of its own, for a stage launch that has no rows to update
__global__ void step_bookkeeping_kernel(double *zero_ptr, double *host_out, double host_seq, const double *amax_ptr)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) step_bookkeeping(zero_ptr, host_out, host_seq, amax_ptr);
}

// C `?:` min/max of the reference (cbdiso_2d.c:16-17); NOT fmin/fmax.
__device__ __forceinline__ double min2(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double max2(double a, double b) { return a > b ? a : b; }

// v_min_f64 / v_max_f64 as single instructions.  __builtin_fmin / fmax are llvm.minnum / maxnum, whose operands the
// compiler first "canonicalises" (v_max_f64 x, x, x: quiets a signalling NaN) whenever it cannot prove them to be
// results of arithmetic -- values that arrived through a lane move, a loop-carried register or a kernel argument:
// ten extra instructions per row of the cbdiso_2d march.  No operand here is ever a signalling NaN, and the
// instruction treats quiet NaNs exactly as the builtin does (the other operand is returned).
__device__ __forceinline__ double min_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double max_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double min_abs_f64(double a, double b) { double r; asm("v_min_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; } // min(|a|, |b|)
__device__ __forceinline__ double min_abs2_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; } // min(a, |b|)
__device__ __forceinline__ double min0_f64(double a) { double r; asm("v_min_f64 %0, %1, 0" : "=v"(r) : "v"(a)); return r; }           // min(a, 0)
__device__ __forceinline__ double max0_f64(double a) { double r; asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(a)); return r; }           // max(a, 0)
// the second operand wave-uniform (scalar registers): no vector copy of it
__device__ __forceinline__ double max_s_f64(double a, double s) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s)); return r; }
__device__ __forceinline__ double min_abs_s_f64(double a, double s) { double r; asm("v_min_f64 %0, |%1|, %2" : "=v"(r) : "v"(a), "s"(s)); return r; } // min(|a|, s)

// Generalised minmod slope, cbdiso_2d.c:23-29 / cbdgam_2d.c:27-33.
// The reference's 0.25*|sgn a + sgn b|*(sgn a + sgn c)*min|.| is +-min when the
// three sign bits agree and a signed zero otherwise; all factors are powers of
// two, so selecting on the sign bits gives the identical value.  theta > 0 and
// rounding is monotone, so min(|theta d1|, |theta d2|) == theta min(|d1|, |d2|).
template <int THETA10> // theta * 10 as an integer template constant
__device__ __forceinline__ double plm(double yl, double y0, double yr)
{
    constexpr double theta = THETA10 / 10.0;
    double d1 = y0 - yl;
    double d2 = yr - y0;
    double b = (yr - yl) * 0.5;
    double m = __builtin_fmin(__builtin_fmin(fabs(d1), fabs(d2)) * theta, fabs(b));
    int sa = __double2hiint(d1), sb = __double2hiint(b), sc = __double2hiint(d2);
    bool same = ((sa ^ sb) | (sa ^ sc)) >= 0; // sign bit clear <=> all three signs equal
    return same ? copysign(m, d1) : 0.0;
}

// TWICE the slope above, exactly (all scalings are powers of two): saves the
// multiply by 0.5; callers reconstruct with p +- 0.25 * plm2.  When d1 and d2
// agree in sign, yr - yl has that sign too (yl <= y0 <= yr or the reverse, and
// comparisons are exact), so only two sign bits need testing.
template <int THETA10>
__device__ __forceinline__ double plm2(double yl, double y0, double yr)
{
    constexpr double theta2 = 2.0 * (THETA10 / 10.0);
    double d1 = y0 - yl;
    double d2 = yr - y0;
    double m = min_abs2_f64(min_abs_f64(d1, d2) * theta2, yr - yl);
    int sx = __double2hiint(d1) ^ __double2hiint(d2);
    asm("" : "+v"(sx)); // keep the sign test on the high dwords (the optimiser widens it to a 64-bit xor + compare)
    bool same = sx >= 0;
    return same ? copysign(m, d1) : 0.0;
}

// The same slope from the two one-sided differences d1 = y0 - yl, d2 = yr - y0 (the central difference is
// their sum, rounded once more than yr - yl: last-place differences only).  For the y-direction, where the
// neighbours are other lanes: d2 is the next lane's d1, one lane move instead of a move and a subtraction.
template <int THETA10>
__device__ __forceinline__ double plm2_diffs(double d1, double d2)
{
    constexpr double theta2 = 2.0 * (THETA10 / 10.0);
    double m = min_abs2_f64(min_abs_f64(d1, d2) * theta2, d1 + d2);
    int sx = __double2hiint(d1) ^ __double2hiint(d2);
    asm("" : "+v"(sx));
    bool same = sx >= 0;
    return same ? copysign(m, d1) : 0.0;
}

// ---------------------------------------------------------------------------
// Fast fp64 reciprocal / reciprocal square root.  IEEE fp64 division and sqrt
// expand to ~11-12 VALU instructions each on CDNA; the stage kernels are
// VALU-bound and the reference evaluates ~40 of them per zone, so the hot path
// uses the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-23 relative) plus Newton
// steps in FMA form instead.  Results are within ~1 ulp (checked on the device
// by tests/test_hip_kernels.py::test_fast_math_accuracy), far inside the 1e-12
// parity budget.  The wavespeed / dt path does NOT use these: it stays on
// correctly rounded division and sqrt so that dt is bit-exact.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double fast_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
#if defined(SFHIP_RSQ_RAW)
    return r;
#endif
#if defined(SFHIP_RCP_TWO_NEWTON)  // rounds 1 and 2: two quadratic steps, one instruction more
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    // one cubically convergent step, r (1 + e + e^2), e = 1 - d r: the seed's 2^-24 becomes 2^-72 before rounding
    double e = __builtin_fma(-d, r, 1.0);
    double t = __builtin_fma(e, e, e);
    return __builtin_fma(r, t, r);
#endif
}

// 1/sqrt(x), x > 0: one cubically convergent step, y (1 + e/2 + 3 e^2/8), e = 1 - x y^2
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
#if defined(SFHIP_RSQ_RAW) // tools/seedacc.py: accuracy of the hardware seeds themselves (2^-24.3, measured)
    return y;
#elif defined(SFHIP_RSQ_QUADRATIC)
    // measured and not adopted: one Newton step leaves 2^-47.8 (18 ulp) for one instruction less, 2 % of a stage
    double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * e, 0.5, y);
#else
    double e = __builtin_fma(-(x * y), y, 1.0);
    double t = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y * e, t, y);
#endif
}

// sqrt(x) from y ~ 1/sqrt(x): s = x y corrected once with its own residual
__device__ __forceinline__ double sqrt_from_rsqrt(double x, double y)
{
    double s = x * y;
    double r = __builtin_fma(-s, s, x);
    return __builtin_fma(r, 0.5 * y, s);
}

// sqrt(x), x > 0, where 1/sqrt(x) itself is not needed: the coupled (Goldschmidt) iteration from the hardware seed
// y0 ~ 1/sqrt(x) -- g = x y0 -> sqrt(x), h = y0/2 -> 1/(2 sqrt(x)), one joint step (2^-24 -> 2^-47), then g corrected
// with its own residual -- 8 instructions against the 10 of fast_rsqrt + sqrt_from_rsqrt; within 1 ulp.
__device__ __forceinline__ double fast_sqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// ---------------------------------------------------------------------------
// Correctly rounded fp64 division with a shared denominator (used by the kernels that must
// reproduce the reference bit for bit: srhd_2d, cbdisodg_2d).
//
// The compiler expands `a / b` into v_div_scale x2, v_rcp_f64, four FMAs that refine the
// reciprocal, then `q0 = a r; e = fma(-b, q0, a); q = fma(e, r, q0)` (v_div_fmas) and
// v_div_fixup: about a dozen VALU instructions.  The refinement depends on the denominator
// alone, so it is done once per denominator here and each quotient costs three instructions.
// The result is BIT-IDENTICAL to `a / b` whenever v_div_scale would not rescale and
// v_div_fixup would not special-case, i.e. for b and a within about 2^-700 .. 2^700 in
// magnitude (a may also be zero; the zero may come out with the other sign).  Not for
// denominators that can be zero, infinite or NaN where the quotient's value matters.
// ---------------------------------------------------------------------------
struct ExactRcp {
    double den, r;
};

__device__ __forceinline__ ExactRcp exact_rcp(double den)
{
    double r = __builtin_amdgcn_rcp(den);
    double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    return ExactRcp{den, r};
}

__device__ __forceinline__ double exact_div(double num, const ExactRcp &d)
{
    double q = num * d.r;
    double e = __builtin_fma(-d.den, q, num);
    return __builtin_fma(e, d.r, q);
}

// One division by a denominator that is used once: still five instructions fewer than `a / b`
// (no v_div_scale pair, no v_div_fixup); same validity range as exact_div.
__device__ __forceinline__ double exact_quot(double num, double den) { return exact_div(num, exact_rcp(den)); }

// Correctly rounded fp64 square root without the compiler's range handling.  `sqrt(x)` expands
// to a compare + two v_ldexp (rescaling for x < 2^-767), v_rsq_f64, the Goldschmidt iteration
// below, and a class test + two selects that pass zero and infinity through: 18 instructions.
// exact_sqrt_pos is the iteration alone (10): BIT-IDENTICAL to sqrt(x) for finite x >= 2^-767
// (NaN for negative x, as sqrt); x = 0 and x = inf, where the iteration produces NaN, must not
// reach it.  exact_sqrt0 adds the zero pass-through (+-0 returns itself, as sqrt does).
__device__ __forceinline__ double exact_sqrt_pos(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double s = x * y;
    double h = y * 0.5;
    double r = __builtin_fma(-h, s, 0.5);
    s = __builtin_fma(s, r, s);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-s, s, x);
    s = __builtin_fma(d, h, s);
    d = __builtin_fma(-s, s, x);
    return __builtin_fma(d, h, s);
}

__device__ __forceinline__ double exact_sqrt0(double x)
{
    double s = exact_sqrt_pos(x);
    return x == 0.0 ? x : s;
}

// exp(x) for the sink profile exp(-(r/r_sink)^4), -700 < x <= 0, reached by a handful of waves per launch.
// Same range reduction, polynomial and coefficients as the device library's exp (the results are bit-identical
// in this range; no overflow/underflow handling is needed here), but the 13 constants are read from constant
// memory inside the call, through a pointer whose origin is hidden from the optimiser: written as exp(), the
// compiler hoists them out of the march loop into 18 vector registers that every iteration then pays for.
__constant__ double EXP_TABLE[13] = {
    0x1.71547652b82fep+0,   // log2(e)
    -0x1.62e42fefa39efp-1,  // -ln2, high part
    -0x1.abc9e3b39803fp-56, // -ln2, low part
    0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22, 0x1.71dee623fde64p-19, 0x1.a01997c89e6b0p-16,
    0x1.a01a014761f6ep-13, 0x1.6c16c1852b7b0p-10, 0x1.1111111122322p-7, 0x1.55555555502a1p-5,
    0x1.5555555555511p-3, 0x1.000000000000bp-1,
};

__device__ __forceinline__ double exp_cold(double x)
{
    const double *c = EXP_TABLE;
    asm volatile("" : "+s"(c));
    double n = __builtin_rint(x * c[0]);
    double r = __builtin_fma(n, c[1], x);
    r = __builtin_fma(n, c[2], r);
    double p = __builtin_fma(r, c[3], c[4]);
#pragma unroll
    for (int k = 5; k < 13; ++k) p = __builtin_fma(r, p, c[k]);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return ldexp(p, (int)n);
}

// The wavespeed folds keep a NaN from either side, as numpy's max does, which is what the reference applies to
// its per-zone wavespeed array (cbdiso_2d.py:174): a blown-up state must give a NaN dt, not a finite one.
__device__ __forceinline__ double max_nan(double a, double s) { return (s > a || s != s) ? s : a; }

// max of a non-negative double over the wave, result valid in every lane (NaN if any lane holds one)
__device__ __forceinline__ double wave_max_nonneg(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max_nan(v, __shfl_xor(v, off, 64));
    return v;
}

// Non-negative doubles order like their bit patterns, so a 64-bit integer
// atomic max implements the floating-point max exactly (bit-exact dt); a NaN's
// pattern ranks above every finite value and infinity, so it survives too.
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v)
{
    // A zero cannot raise a non-negative maximum: skip it (a NaN is not skipped).  This matters for the fused
    // epilogue of the stage kernels, whose filter leaves most waves without any zone worth an exact evaluation:
    // the same-address atomics of a launch serialise at ~8 ns each, 8 us for the ~1000 waves of a 512^2 launch
    // that otherwise takes 15.  (A plain load first, to skip values below the CURRENT maximum, does not help: the
    // waves of a one-round launch all finish together and all read the initial zero.)
    if (!(v <= 0.0))
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
