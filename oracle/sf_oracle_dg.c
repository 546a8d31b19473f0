/*
 * sf_oracle_dg.c -- CPU restatement of sailfish's cbdisodg_2d kernels: third-order
 * modal discontinuous-Galerkin update of the isothermal binary-accretion problem.
 *
 * TEST INFRASTRUCTURE ONLY (see sf_oracle.c): parity oracle for the HIP kernels in
 * sailfish_amd/csrc/sf_dg.hpp.  Pinned bit-for-bit against the reference's own C
 * compiled in place (oracle/_ref/cbdisodg_2d_*.so, tests/test_oracle_vs_ref.py) and
 * by the reference-generated fixtures tests/golden/dg_*.npz.
 *
 * Reference: /root/reference/sailfish/solvers/cbdisodg_2d.c (line numbers below).
 * Layout: weights arrays (ni+2, nj+2, 3, 3, 3) = zone (one guard zone per side),
 * conserved field q, Legendre orders m (x) and n (y); only the six slots with
 * m + n < 3 take part in the scheme, the other three are carried along untouched.
 *
 * Behaviours of the reference that look unintended and are reproduced because they
 * decide the output bits (each is noted again where it happens):
 *   - minmod_simple is declared `int` (:455): the limited slope is truncated to an integer;
 *   - the "gradient" basis tables on faces normal to the derivative direction hold
 *     face VALUES, because basis_phi_1d tests for an endpoint before it looks at
 *     `deriv` (:383-390);
 *   - add_viscous_flux ignores its `multiplier` argument (:326-361): both sides of
 *     a face add their full viscous stress.
 */
#include <math.h>
#include <string.h>

#define NQ 3
#define NW 27   /* doubles per zone: 3 fields x 3 x 3 */
#define SLOT(q, m, n) ((q) * 9 + (m) * 3 + (n))

typedef struct {
    double x, y, vx, vy, mass, rsoft, sink_rate, sink_radius;
    int sink_model;
} dg_mass_t; /* cbdisodg_2d.c:28-38 */

typedef struct {
    double surface_density, central_mass, driving_rate, outer_radius, onset_width;
    int is_enabled;
} dg_buffer_t; /* cbdisodg_2d.c:44-51 */

/* Scaled Legendre polynomials: values at the three Gauss nodes, derivatives there,
 * and values at the two interval ends (cbdisodg_2d.c:366-381). */
static const double P_NODE[3][3] = {
    {+1.000000000000000, +1.00000000000000, +1.000000000000000},
    {-1.341640786499873, +0.00000000000000, +1.341640786499873},
    {+0.894427190999914, -1.11803398874990, +0.894427190999914},
};
static const double P_DERIV[3][3] = {
    {+0.000000000000000, +0.000000000000000, +0.000000000000000},
    {+1.732050807568877, +1.732050807568877, +1.732050807568877},
    {-5.196152422706629, +0.000000000000000, +5.196152422706629},
};
static const double P_LEFT[3] = {+1.000000000000000, -1.732050807568877, +2.23606797749979};
static const double P_RIGHT[3] = {+1.000000000000000, +1.732050807568877, +2.23606797749979};
static const double GAUSS_W[3] = {0.555555555555556, 0.888888888888889, 0.555555555555556}; /* :648 */
static const double GAUSS_X[3] = {-0.774596669241483, 0.000000000000000, 0.774596669241483}; /* :649 */

static inline double lo2(double a, double b) { return a < b ? a : b; } /* :18 */
static inline double hi2(double a, double b) { return a > b ? a : b; } /* :19 */

/* One-dimensional basis factor for position code `at`: 0..2 a Gauss node (value or
 * derivative), 3 the left end, 4 the right end.  At an end the reference returns
 * the VALUE whatever `deriv` says (:383-390). */
static inline double basis_1d(int at, int m, int deriv)
{
    if (at == 3) return P_LEFT[m];
    if (at == 4) return P_RIGHT[m];
    return deriv ? P_DERIV[m][at] : P_NODE[m][at];
}

/* sum over the six used slots of w[q][m][n] * (bx[m] * by[n]), in the reference's
 * order and starting from 0.0 (reconstruct_2d / reconstruct_1d, :411-453) */
static void evaluate(const double *w, int at_x, int dx_, int at_y, int dy_, double *out)
{
    for (int q = 0; q < NQ; ++q) {
        double s = 0.0;
        for (int m = 0; m < 3; ++m)
            for (int n = 0; n < 3 - m; ++n)
                s += w[SLOT(q, m, n)] * (basis_1d(at_x, m, dx_) * basis_1d(at_y, n, dy_));
        out[q] = s;
    }
}

/* :56-81 */
static double potential(const dg_mass_t *m, double x, double y)
{
    double phi = 0.0;
    for (int p = 0; p < 2; ++p) {
        if (m[p].mass > 0.0) {
            double dx = x - m[p].x;
            double dy = y - m[p].y;
            double r2 = dx * dx + dy * dy;
            phi -= m[p].mass / sqrt(r2 + m[p].rsoft * m[p].rsoft);
        }
    }
    return phi;
}

/* :171-188 */
static double sound_speed2(double cs2, double mach2, int eos, double x, double y, const dg_mass_t *m)
{
    if (eos == 1) return cs2;
    if (eos == 2) return -potential(m, x, y) / mach2;
    return 1.0;
}

/* :83-152 -- gravity and sink of one mass; `prim` = (sigma, vx, vy) */
static void mass_source(const dg_mass_t *ms, double x, double y, double dt, const double *prim, double *du)
{
    double sigma = prim[0];
    double dx = x - ms->x;
    double dy = y - ms->y;
    double r2 = dx * dx + dy * dy;
    double dr = sqrt(r2);
    double r_sink = ms->sink_radius;
    double r_soft = ms->rsoft;
    double fnum = sigma * ms->mass * pow(r2 + r_soft * r_soft, -1.5);
    double fx = -fnum * dx;
    double fy = -fnum * dy;
    double rate = (dr < 4.0 * r_sink) ? ms->sink_rate * exp(-pow(dr / r_sink, 4.0)) : 0.0;
    double mdot = sigma * rate * -1.0;

    du[0] += 0.0;
    du[1] += fx * dt;
    du[2] += fy * dt;

    switch (ms->sink_model) {
    case 1: /* acceleration-free */
        du[0] += dt * mdot;
        du[1] += dt * mdot * prim[1];
        du[2] += dt * mdot * prim[2];
        break;
    case 2: { /* torque-free */
        double rhx = dx / (dr + 1e-12);
        double rhy = dy / (dr + 1e-12);
        double dv = (prim[1] - ms->vx) * rhx + (prim[2] - ms->vy) * rhy;
        double vxs = dv * rhx + ms->vx;
        double vys = dv * rhy + ms->vy;
        du[0] += dt * mdot;
        du[1] += dt * mdot * vxs;
        du[2] += dt * mdot * vys;
        break;
    }
    case 3: /* force-free */
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

/* :190-223 -- driving toward a Keplerian state in the outer buffer; acts on conserved */
static void buffer_source(const dg_buffer_t *b, double x, double y, double dt, const double *cons, double *du)
{
    if (!b->is_enabled) return;
    double rc = sqrt(x * x + y * y);
    double onset = b->outer_radius - b->onset_width;
    if (rc > onset) {
        double vk = sqrt(b->central_mass / rc);
        double u0[NQ] = {b->surface_density, b->surface_density * (-y / rc) * vk, b->surface_density * (+x / rc) * vk};
        double omega = sqrt(b->central_mass * pow(onset, -3.0));
        double rate = b->driving_rate * omega * (rc - onset) / (b->outer_radius - onset);
        for (int q = 0; q < NQ; ++q)
            du[q] -= (cons[q] - u0[q]) * rate * dt;
    }
}

/* :227-241 */
static void cons_to_prim(const double *u, double *p, double vmax)
{
    p[0] = u[0];
    p[1] = copysign(1.0, u[1]) * lo2(fabs(u[1] / u[0]), vmax);
    p[2] = copysign(1.0, u[2]) * lo2(fabs(u[2] / u[0]), vmax);
}

/* :255-269 */
static void flux_of(const double *p, const double *u, double cs2, int dir, double *f)
{
    double vn = p[1 + dir];
    double pressure = p[0] * cs2;
    f[0] = vn * u[0];
    f[1] = vn * u[1] + pressure * (double)(dir == 0);
    f[2] = vn * u[2] + pressure * (double)(dir == 1);
}

/* :295-324 -- HLLE on conserved states */
static void hlle(const double *ul, const double *ur, double cs2, double vmax, int dir, double *f)
{
    double pl[NQ], pr[NQ], fl[NQ], fr[NQ];
    cons_to_prim(ul, pl, vmax);
    cons_to_prim(ur, pr, vmax);
    flux_of(pl, ul, cs2, dir, fl);
    flux_of(pr, ur, cs2, dir, fr);
    double cs = sqrt(cs2);
    double am = lo2(0.0, lo2(pl[1 + dir] - cs, pr[1 + dir] - cs));
    double ap = hi2(0.0, hi2(pl[1 + dir] + cs, pr[1 + dir] + cs));
    for (int q = 0; q < NQ; ++q)
        f[q] = (fl[q] * ap - fr[q] * am - (ul[q] - ur[q]) * ap * am) / (ap - am);
}

/* :326-361 -- viscous stress subtracted from the x and/or y flux.  The reference's
 * `multiplier` argument is never used in its body, so there is none here. */
static void viscous(const double *u, const double *ux, const double *uy, double *fx, double *fy,
                    double nu, double dx, double dy)
{
    double dvxdx = 2.0 * (ux[1] / u[0] - u[1] * ux[0] / u[0] / u[0]) / dx;
    double dvydx = 2.0 * (ux[2] / u[0] - u[2] * ux[0] / u[0] / u[0]) / dx;
    double dvxdy = 2.0 * (uy[1] / u[0] - u[1] * uy[0] / u[0] / u[0]) / dy;
    double dvydy = 2.0 * (uy[2] / u[0] - u[2] * uy[0] / u[0] / u[0]) / dy;
    double mu = u[0] * nu;
    double sxx = 4.0 / 3.0 * dvxdx - 2.0 / 3.0 * dvydy;
    double syy = 4.0 / 3.0 * dvydy - 2.0 / 3.0 * dvxdx;
    double sxy = dvxdy + dvydx;
    if (fx) {
        fx[1] -= mu * sxx;
        fx[2] -= mu * sxy;
    }
    if (fy) {
        fy[1] -= mu * sxy;
        fy[2] -= mu * syy;
    }
}

/* :455-493.  The reference declares this function `int`: whatever slope it picks is
 * truncated toward zero on return (x86: out-of-range and NaN become INT_MIN). */
static double minmod_truncated(double w1, double w0l, double w0, double w0r)
{
    double a = w1;
    double b = (w0 - w0l) * 1.0 / sqrt(3.0);
    double c = (w0r - w0) * 1.0 / sqrt(3.0);
    double pick = 0.0; /* mixed signs (or a NaN): zero */
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

/* ------------------------------------------------------------------------------------
 * cbdisodg_2d_slope_limit, :498-605.  Troubled-cell indicator of Fu & Shu on the surface
 * density; troubled zones get every field's higher moments zeroed and the two slopes
 * replaced by the (truncated) minmod values.  The mesh and point-mass arguments are not
 * used by the reference either.
 * ------------------------------------------------------------------------------------ */
void sfo_cbdisodg_2d_slope_limit(int ni, int nj, double xl, double xr, double yl, double yr,
                                 double m1x, double m1y, double m2x, double m2y,
                                 const double *weights1, double *weights2)
{
    (void)xl; (void)xr; (void)yl; (void)yr; (void)m1x; (void)m1y; (void)m2x; (void)m2y;
    const long si = (long)NW * (nj + 2), sj = NW;
    const double s3 = sqrt(3.0), s5 = sqrt(5.0);
#pragma omp parallel for
    for (int i = 0; i < ni; ++i) {
        for (int j = 0; j < nj; ++j) {
            const double *cc = weights1 + (i + 1) * si + (j + 1) * sj;
            const double *li = cc - si, *ri = cc + si, *lj = cc - sj, *rj = cc + sj;
            double *out = weights2 + (i + 1) * si + (j + 1) * sj;
            memcpy(out, cc, NW * sizeof(double));

            const int t00 = SLOT(0, 0, 0), t01 = SLOT(0, 0, 1), t10 = SLOT(0, 1, 0), t02 = SLOT(0, 0, 2), t20 = SLOT(0, 2, 0);
            double big = hi2(hi2(fabs(cc[t00]), fabs(li[t00])),
                             hi2(fabs(ri[t00]), hi2(fabs(lj[t00]), fabs(rj[t00]))));
            double a = 4.0 * li[t00] + 8.0 * s3 * li[t10] + 24.0 * s5 * li[t20];
            double b = 4.0 * ri[t00] - 8.0 * s3 * ri[t10] + 24.0 * s5 * ri[t20];
            double c = 4.0 * lj[t00] + 8.0 * s3 * lj[t01] + 24.0 * s5 * lj[t02];
            double d = 4.0 * rj[t00] - 8.0 * s3 * rj[t01] + 24.0 * s5 * rj[t02];
            double tci = (fabs(cc[t00] - a / 4.0) + fabs(cc[t00] - b / 4.0) + fabs(cc[t00] - c / 4.0) +
                          fabs(cc[t00] - d / 4.0)) / big;
            if (tci > 0.1) {
                for (int q = 0; q < NQ; ++q) {
                    const int p00 = SLOT(q, 0, 0), p01 = SLOT(q, 0, 1), p10 = SLOT(q, 1, 0);
                    double wx = minmod_truncated(cc[p10], li[p00], cc[p00], ri[p00]);
                    double wy = minmod_truncated(cc[p01], lj[p00], cc[p00], rj[p00]);
                    if (wx != cc[p10] || wy != cc[p01]) {
                        for (int k = 1; k < 9; ++k)
                            out[q * 9 + k] = 0.0;
                        out[p10] = wx;
                        out[p01] = wy;
                    }
                }
            }
        }
    }
}

/* HLLE flux through one face node plus, for nu > 0, the viscous stress of both sides.
 * `lo` is the zone on the low side of the face, `hi` the one on the high side;
 * `node` the Gauss node along the face.  (:848-961, one of the four blocks) */
static void face_flux(const double *lo, const double *hi, int dir, int node, double cs2, double vmax, double nu,
                      double dx, double dy, double *fhat)
{
    double um[NQ], up[NQ], gx[NQ], gy[NQ];
    if (dir == 0) {
        evaluate(hi, 3, 0, node, 0, up);
        evaluate(lo, 4, 0, node, 0, um);
    } else {
        evaluate(hi, node, 0, 3, 0, up);
        evaluate(lo, node, 0, 4, 0, um);
    }
    hlle(um, up, cs2, vmax, dir, fhat);
    if (nu > 0.0) {
        double *fx = dir == 0 ? fhat : 0, *fy = dir == 1 ? fhat : 0;
        if (dir == 0) {
            evaluate(hi, 3, 1, node, 0, gx);
            evaluate(hi, 3, 0, node, 1, gy);
            viscous(up, gx, gy, fx, fy, nu, dx, dy);
            evaluate(lo, 4, 1, node, 0, gx);
            evaluate(lo, 4, 0, node, 1, gy);
            viscous(um, gx, gy, fx, fy, nu, dx, dy);
        } else {
            evaluate(hi, node, 1, 3, 0, gx);
            evaluate(hi, node, 0, 3, 1, gy);
            viscous(up, gx, gy, fx, fy, nu, dx, dy);
            evaluate(lo, node, 1, 4, 0, gx);
            evaluate(lo, node, 0, 4, 1, gy);
            viscous(um, gx, gy, fx, fy, nu, dx, dy);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * cbdisodg_2d_advance_rk, :607-983.
 * ------------------------------------------------------------------------------------ */
void sfo_cbdisodg_2d_advance_rk(
    int ni, int nj, double xl, double xr, double yl, double yr,
    const double *weights0, const double *weights1, double *weights2,
    double buffer_surface_density, double buffer_central_mass, double buffer_driving_rate,
    double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled,
    double x1, double y1, double vx1, double vy1, double mass1, double rsoft1, double rate1, double rsink1, int model1,
    double x2, double y2, double vx2, double vy2, double mass2, double rsoft2, double rate2, double rsink2, int model2,
    double soundspeed2, double mach_squared, int eos_type, double nu, double rk_param, double dt,
    double velocity_ceiling)
{
    const double dx = (xr - xl) / ni;
    const double dy = (yr - yl) / nj;
    const double cell_volume = dx * dy;
    const long si = (long)NW * (nj + 2), sj = NW;
    const dg_buffer_t buffer = {buffer_surface_density, buffer_central_mass, buffer_driving_rate,
                                buffer_outer_radius, buffer_onset_width, buffer_is_enabled};
    const dg_mass_t masses[2] = {{x1, y1, vx1, vy1, mass1, rsoft1, rate1, rsink1, model1},
                                 {x2, y2, vx2, vy2, mass2, rsoft2, rate2, rsink2, model2}};
    const double vmax = velocity_ceiling;

#pragma omp parallel for
    for (int i = 0; i < ni; ++i) {
        for (int j = 0; j < nj; ++j) {
            const long at = (i + 1) * si + (j + 1) * sj;
            const double *cc = weights1 + at;
            const double *li = cc - si, *ri = cc + si, *lj = cc - sj, *rj = cc + sj;
            double vol[NW] = {0.0}, srf[NW] = {0.0}, src[NW] = {0.0}; /* equation_19, equation_20, source_weights */
            const double xc = xl + (i + 0.5) * dx;
            const double yc = yl + (j + 0.5) * dy;

            /* volume integral and sources, :776-844 */
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) {
                    double x = xc + 0.5 * GAUSS_X[a] * dx;
                    double y = yc + 0.5 * GAUSS_X[b] * dy;
                    double gw = GAUSS_W[a] * GAUSS_W[b];
                    double u[NQ], p[NQ], fx[NQ], fy[NQ];
                    double cs2 = sound_speed2(soundspeed2, mach_squared, eos_type, x, y, masses);
                    evaluate(cc, a, 0, b, 0, u);
                    cons_to_prim(u, p, vmax);
                    flux_of(p, u, cs2, 0, fx);
                    flux_of(p, u, cs2, 1, fy);
                    if (nu > 0.0) {
                        double gx[NQ], gy[NQ];
                        evaluate(cc, a, 1, b, 0, gx);
                        evaluate(cc, a, 0, b, 1, gy);
                        viscous(u, gx, gy, fx, fy, nu, dx, dy);
                    }
                    for (int m = 0; m < 3; ++m) {
                        for (int n = 0; n < 3 - m; ++n) {
                            double dpx = basis_1d(a, m, 1) * basis_1d(b, n, 0);
                            double dpy = basis_1d(a, m, 0) * basis_1d(b, n, 1);
                            for (int q = 0; q < NQ; ++q) {
                                vol[SLOT(q, m, n)] += dx * fx[q] * dpx * gw;
                                vol[SLOT(q, m, n)] += dy * fy[q] * dpy * gw;
                            }
                        }
                    }
                    double du[NQ] = {0.0, 0.0, 0.0};
                    buffer_source(&buffer, x, y, dt, u, du);
                    mass_source(&masses[0], x, y, dt, p, du);
                    mass_source(&masses[1], x, y, dt, p, du);
                    for (int q = 0; q < NQ; ++q)
                        for (int m = 0; m < 3; ++m)
                            for (int n = 0; n < 3 - m; ++n)
                                src[SLOT(q, m, n)] += 0.25 * du[q] * (basis_1d(a, m, 0) * basis_1d(b, n, 0)) * gw;
                }
            }

            /* surface integral, :848-961: per Gauss node the faces xl, xr, yl, yr in that order */
            for (int k = 0; k < 3; ++k) {
                double fhat[NQ];
                for (int face = 0; face < 4; ++face) {
                    const int dir = face >> 1, high = face & 1;
                    double x, y;
                    if (dir == 0) {
                        x = high ? xc + 0.5 * dx : xc - 0.5 * dx;
                        y = yc + 0.5 * GAUSS_X[k] * dy;
                    } else {
                        x = xc + 0.5 * GAUSS_X[k] * dx;
                        y = high ? yc + 0.5 * dy : yc - 0.5 * dy;
                    }
                    double cs2 = sound_speed2(soundspeed2, mach_squared, eos_type, x, y, masses);
                    const double *lo = high ? cc : (dir == 0 ? li : lj);
                    const double *hi = high ? (dir == 0 ? ri : rj) : cc;
                    face_flux(lo, hi, dir, k, cs2, vmax, nu, dx, dy, fhat);
                    const double len = dir == 0 ? dy : dx;
                    for (int q = 0; q < NQ; ++q) {
                        for (int m = 0; m < 3; ++m) {
                            for (int n = 0; n < 3 - m; ++n) {
                                double phi = dir == 0 ? basis_1d(3 + high, m, 0) * basis_1d(k, n, 0)
                                                      : basis_1d(k, m, 0) * basis_1d(3 + high, n, 0);
                                double term = len * fhat[q] * phi * GAUSS_W[k];
                                if (high) srf[SLOT(q, m, n)] += term;
                                else srf[SLOT(q, m, n)] -= term;
                            }
                        }
                    }
                }
            }

            /* :963-981 -- only the six used slots of each field are written */
            const double *w0 = weights0 + at;
            double *w2 = weights2 + at;
            for (int q = 0; q < NQ; ++q) {
                for (int m = 0; m < 3; ++m) {
                    for (int n = 0; n < 3 - m; ++n) {
                        const int s = SLOT(q, m, n);
                        double w = cc[s] + (vol[s] - srf[s]) * 0.5 * dt / cell_volume + src[s];
                        w2[s] = (1.0 - rk_param) * w + rk_param * w0[s];
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * cbdisodg_2d_wavespeed, :1108-1176: from the zone averages (slot [q][0][0]).
 * ------------------------------------------------------------------------------------ */
void sfo_cbdisodg_2d_wavespeed(
    int ni, int nj, double xl, double xr, double yl, double yr,
    double soundspeed2, double mach_squared, int eos_type,
    double x1, double y1, double vx1, double vy1, double mass1, double rsoft1, double rate1, double rsink1, int model1,
    double x2, double y2, double vx2, double vy2, double mass2, double rsoft2, double rate2, double rsink2, int model2,
    double velocity_ceiling, const double *weights, double *wavespeed)
{
    const double dx = (xr - xl) / ni;
    const double dy = (yr - yl) / nj;
    const dg_mass_t masses[2] = {{x1, y1, vx1, vy1, mass1, rsoft1, rate1, rsink1, model1},
                                 {x2, y2, vx2, vy2, mass2, rsoft2, rate2, rsink2, model2}};
#pragma omp parallel for
    for (int i = 0; i < ni; ++i) {
        for (int j = 0; j < nj; ++j) {
            const double *cc = weights + ((long)(i + 1) * (nj + 2) + (j + 1)) * NW;
            double x = xl + (i + 0.5) * dx;
            double y = yl + (j + 0.5) * dy;
            double u[NQ] = {cc[0], cc[9], cc[18]}, p[NQ];
            cons_to_prim(u, p, velocity_ceiling);
            double cs = sqrt(sound_speed2(soundspeed2, mach_squared, eos_type, x, y, masses));
            double ax = hi2(fabs(p[1] - cs), fabs(p[1] + cs));
            double ay = hi2(fabs(p[2] - cs), fabs(p[2] + cs));
            wavespeed[(long)(i + 1) * (nj + 2) + (j + 1)] = hi2(ax, ay);
        }
    }
}
