/*
 * sf_oracle.c -- CPU restatement of sailfish's cbdiso_2d / cbdgam_2d per-zone
 * finite-volume update.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP kernels
 * in sailfish_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it; the product path
 * (sailfish_amd/) never does and fails loudly when the HIP library is missing.
 *
 * Pinning: the reference ships no golden vectors for this path (SURVEY.md §4,
 * §8c).  The oracle is pinned against the reference's own C kernels compiled in
 * place (oracle/build_ref.py -> oracle/_ref/ shared objects): tests/test_oracle_vs_ref.py
 * demands BIT-IDENTICAL output on every branch combination, and the committed
 * fixtures under tests/golden/ were produced by running the reference (C
 * kernels + Python solver classes) in the build container
 * (tests/golden/generate.py).
 *
 * It is a restatement, not a transcription: the reference evaluates the four
 * faces of every zone from that zone's point of view (every face twice); this
 * file evaluates slopes once per zone and fluxes once per face -- the same
 * decomposition the HIP kernels use -- and relies on the fact (checked bitwise
 * against oracle/_ref) that both points of view perform the identical IEEE
 * operations.  All arithmetic is fp64, compiled with -ffp-contract=off.
 *
 * Array layout (reference: cbdiso_2d.c:443-445): C-contiguous
 * (ni+4, nj+4, NQ), two guard zones per side, q fastest.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NG 2

typedef struct {
    double x, y, vx, vy, mass, rsoft, sink_rate, sink_radius;
    int sink_model;
} sfo_mass_t; /* cbdiso_2d.c:42-52, cbdgam_2d.c:46-56 */

static inline double dmin(double a, double b) { return a < b ? a : b; } /* cbdiso_2d.c:16 */
static inline double dmax(double a, double b) { return a > b ? a : b; } /* cbdiso_2d.c:17 */
static inline double sgn(double x) { return copysign(1.0, x); }        /* cbdiso_2d.c:20 */

/* generalized minmod slope: cbdiso_2d.c:23-29 (theta 1.8), cbdgam_2d.c:27-33 (theta 1.5) */
static inline double slope(double yl, double y0, double yr, double theta)
{
    double a = (y0 - yl) * theta;
    double b = (yr - yl) * 0.5;
    double c = (yr - y0) * theta;
    double m = dmin(fabs(a), dmin(fabs(b), fabs(c)));
    return 0.25 * fabs(sgn(a) + sgn(b)) * (sgn(a) + sgn(c)) * m;
}

/* softened potential of up to two masses: cbdiso_2d.c:70-95 */
static double potential(const sfo_mass_t *m, double x, double y)
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

/* cbdiso_2d.c:194-211 */
static double iso_cs2(double cs2, double mach2, int eos, double x, double y, const sfo_mass_t *m)
{
    if (eos == 1) return cs2;
    if (eos == 2) return -potential(m, x, y) / mach2;
    return 1.0;
}

/* HLLE for the isothermal system: cbdiso_2d.c:285-338, 352-380.  dir 0 = x, 1 = y. */
static void iso_hlle(const double *pl, const double *pr, double cs2, int dir, double *f)
{
    double ul[3] = {pl[0], pl[1] * pl[0], pl[2] * pl[0]};
    double ur[3] = {pr[0], pr[1] * pr[0], pr[2] * pr[0]};
    double vl = pl[1 + dir], vr = pr[1 + dir];
    double prl = pl[0] * cs2, prr = pr[0] * cs2;
    double fl[3], fr[3];
    fl[0] = vl * ul[0];
    fl[1] = vl * ul[1] + prl * (dir == 0);
    fl[2] = vl * ul[2] + prl * (dir == 1);
    fr[0] = vr * ur[0];
    fr[1] = vr * ur[1] + prr * (dir == 0);
    fr[2] = vr * ur[2] + prr * (dir == 1);
    double cs = sqrt(cs2);
    double am = dmin(0.0, dmin(vl - cs, vr - cs));
    double ap = dmax(0.0, dmax(vl + cs, vr + cs));
    for (int q = 0; q < 3; ++q)
        f[q] = (fl[q] * ap - fr[q] * am - (ul[q] - ur[q]) * ap * am) / (ap - am);
}

/* strain components from limited slopes: cbdiso_2d.c:248-263 == cbdgam_2d.c:278-293
 * s[0]=xx s[1]=xy s[2]=yx s[3]=yy */
static void strain(const double *gx, const double *gy, double dx, double dy, double *s)
{
    s[0] = 4.0 / 3.0 * gx[1] / dx - 2.0 / 3.0 * gy[2] / dy;
    s[3] = -2.0 / 3.0 * gx[1] / dx + 4.0 / 3.0 * gy[2] / dy;
    s[1] = 1.0 / 1.0 * gx[2] / dx + 1.0 / 1.0 * gy[1] / dy;
    s[2] = s[1];
}

/* one point mass, isothermal: cbdiso_2d.c:97-175 */
static void iso_mass_source(const sfo_mass_t *m, double x1, double y1, double dt, const double *prim, double *dc)
{
    double sigma = prim[0];
    double dx = x1 - m->x;
    double dy = y1 - m->y;
    double r2 = dx * dx + dy * dy;
    double dr = sqrt(r2);
    double r_sink = m->sink_radius;
    double r_soft = m->rsoft;
    double fnum = sigma * m->mass * pow(r2 + r_soft * r_soft, -1.5);
    double fx = -fnum * dx;
    double fy = -fnum * dy;
    double rate = (dr < 4.0 * r_sink) ? m->sink_rate * exp(-pow(dr / r_sink, 4.0)) : 0.0;
    double mdot = 0.0;
    if (rate > 0.0) mdot = -rate * sigma;
    else if (rate < 0.0) mdot = -rate;

    dc[0] += 0.0;
    dc[1] += fx * dt;
    dc[2] += fy * dt;
    if (m->sink_model == 1) {
        dc[0] += dt * mdot;
        dc[1] += dt * mdot * prim[1];
        dc[2] += dt * mdot * prim[2];
    } else if (m->sink_model == 2) {
        double rhx = dx / (dr + 1e-12);
        double rhy = dy / (dr + 1e-12);
        double dv = (prim[1] - m->vx) * rhx + (prim[2] - m->vy) * rhy;
        double vxs = dv * rhx + m->vx;
        double vys = dv * rhy + m->vy;
        dc[0] += dt * mdot;
        dc[1] += dt * mdot * vxs;
        dc[2] += dt * mdot * vys;
    } else if (m->sink_model == 3) {
        dc[0] += dt * mdot;
        dc[1] += 0.0;
        dc[2] += 0.0;
    } else {
        dc[0] += 0.0;
        dc[1] += 0.0;
        dc[2] += 0.0;
    }
}

/* keplerian wave-damping buffer, isothermal: cbdiso_2d.c:213-246 */
static void iso_buffer(double sd, double mc, double rate, double r_out, double w, int on,
                       double xc, double yc, double dt, const double *u, double *dc)
{
    if (!on) return;
    double rc = sqrt(xc * xc + yc * yc);
    double r_on = r_out - w;
    if (rc > r_on) {
        double vk = sqrt(mc / rc);
        double u0[3] = {sd, sd * (-yc / rc) * vk, sd * (+xc / rc) * vk};
        double om = sqrt(mc * pow(r_on, -3.0));
        double br = rate * om * (rc - r_on) / (r_out - r_on);
        for (int q = 0; q < 3; ++q) dc[q] -= (u[q] - u0[q]) * br * dt;
    }
}

/* cbdiso_2d.c:268-283 */
static void iso_c2p(const double *u, double *p, double vmax, double dfloor)
{
    double rho = dmax(u[0], dfloor);
    p[0] = rho;
    p[1] = sgn(u[1]) * dmin(fabs(u[1] / rho), vmax);
    p[2] = sgn(u[2]) * dmin(fabs(u[2] / rho), vmax);
}

/*
 * One RK stage of the isothermal solver.  Restates cbdiso_2d_advance_rk,
 * cbdiso_2d.c:385-604 (same argument list, same order).
 */
void sfo_cbdiso_2d_advance_rk(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    const double *conserved_rk, const double *primitive_rd, double *primitive_wr,
    double buffer_surface_density, double buffer_central_mass, double buffer_driving_rate,
    double buffer_outer_radius, double buffer_onset_width, int buffer_is_enabled,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    double cs2, double mach_squared, int eos_type, double nu, double a, double dt,
    double velocity_ceiling, double density_floor)
{
    const double theta = 1.8; /* cbdiso_2d.c:11 */
    const int NQ = 3;
    sfo_mass_t m[2] = {
        {x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1},
        {x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2}};
    const double dx = (patch_xr - patch_xl) / ni; /* cbdiso_2d.c:440-441 */
    const double dy = (patch_yr - patch_yl) / nj;
    const long NI = ni + 2 * NG, NJ = nj + 2 * NG;
    const long si = NQ * NJ, sj = NQ;
    const double *P = primitive_rd;
#define Z(I, J) ((I) * si + (J) * sj)

    /* pass A: limited slopes of every zone one ring outside the interior (cbdiso_2d.c:525-534) */
    double *gx = calloc((size_t)(NI * NJ * NQ), sizeof(double));
    double *gy = calloc((size_t)(NI * NJ * NQ), sizeof(double));
    /* pass B storage: x-face flux F[I][J] is the face between (I-1,J) and (I,J); G likewise in J */
    double *F = calloc((size_t)(NI * NJ * NQ), sizeof(double));
    double *G = calloc((size_t)(NI * NJ * NQ), sizeof(double));

    _Pragma("omp parallel for")
    for (long I = 1; I < NI - 1; ++I)
        for (long J = 1; J < NJ - 1; ++J)
            for (int q = 0; q < NQ; ++q) {
                gx[Z(I, J) + q] = slope(P[Z(I - 1, J) + q], P[Z(I, J) + q], P[Z(I + 1, J) + q], theta);
                gy[Z(I, J) + q] = slope(P[Z(I, J - 1) + q], P[Z(I, J) + q], P[Z(I, J + 1) + q], theta);
            }

    /* pass B: one HLLE flux per face (cbdiso_2d.c:536-587) */
    _Pragma("omp parallel for")
    for (long I = NG; I <= ni + NG; ++I)
        for (long J = NG; J < nj + NG; ++J) {
            long L = Z(I - 1, J), R = Z(I, J);
            double pm[3], pp[3];
            for (int q = 0; q < NQ; ++q) {
                pm[q] = P[L + q] + 0.5 * gx[L + q];
                pp[q] = P[R + q] - 0.5 * gx[R + q];
            }
            double xf = patch_xl + ((double)(I - NG)) * dx;
            double yc = patch_yl + ((J - NG) + 0.5) * dy;
            double c2 = iso_cs2(cs2, mach_squared, eos_type, xf, yc, m);
            double *f = &F[R];
            iso_hlle(pm, pp, c2, 0, f);
            if (nu > 0.0) {
                double sl[4], sr[4];
                strain(&gx[L], &gy[L], dx, dy, sl);
                strain(&gx[R], &gy[R], dx, dy, sr);
                f[1] -= 0.5 * nu * (P[L] * sl[0] + P[R] * sr[0]);
                f[2] -= 0.5 * nu * (P[L] * sl[1] + P[R] * sr[1]);
            }
        }
    _Pragma("omp parallel for")
    for (long I = NG; I < ni + NG; ++I)
        for (long J = NG; J <= nj + NG; ++J) {
            long L = Z(I, J - 1), R = Z(I, J);
            double pm[3], pp[3];
            for (int q = 0; q < NQ; ++q) {
                pm[q] = P[L + q] + 0.5 * gy[L + q];
                pp[q] = P[R + q] - 0.5 * gy[R + q];
            }
            double xc = patch_xl + ((I - NG) + 0.5) * dx;
            double yf = patch_yl + ((double)(J - NG)) * dy;
            double c2 = iso_cs2(cs2, mach_squared, eos_type, xc, yf, m);
            double *g = &G[R];
            iso_hlle(pm, pp, c2, 1, g);
            if (nu > 0.0) {
                double sl[4], sr[4];
                strain(&gx[L], &gy[L], dx, dy, sl);
                strain(&gx[R], &gy[R], dx, dy, sr);
                g[1] -= 0.5 * nu * (P[L] * sl[2] + P[R] * sr[2]);
                g[2] -= 0.5 * nu * (P[L] * sl[3] + P[R] * sr[3]);
            }
        }

    /* pass C: sources, divergence, RK blend, primitive recovery (cbdiso_2d.c:588-602) */
    _Pragma("omp parallel for")
    for (long I = NG; I < ni + NG; ++I)
        for (long J = NG; J < nj + NG; ++J) {
            long c = Z(I, J);
            const double *pc = &P[c];
            double xc = patch_xl + ((I - NG) + 0.5) * dx;
            double yc = patch_yl + ((J - NG) + 0.5) * dy;
            double u[3] = {pc[0], pc[1] * pc[0], pc[2] * pc[0]};
            double dc[3] = {0.0, 0.0, 0.0};
            iso_buffer(buffer_surface_density, buffer_central_mass, buffer_driving_rate,
                       buffer_outer_radius, buffer_onset_width, buffer_is_enabled, xc, yc, dt, u, dc);
            iso_mass_source(&m[0], xc, yc, dt, pc, dc);
            iso_mass_source(&m[1], xc, yc, dt, pc, dc);
            for (int q = 0; q < NQ; ++q)
                dc[q] -= ((F[Z(I + 1, J) + q] - F[c + q]) / dx + (G[Z(I, J + 1) + q] - G[c + q]) / dy) * dt;
            for (int q = 0; q < NQ; ++q) {
                u[q] += dc[q];
                u[q] = (1.0 - a) * u[q] + a * conserved_rk[c + q];
            }
            iso_c2p(u, &primitive_wr[c], velocity_ceiling, density_floor);
        }
    free(gx); free(gy); free(F); free(G);
#undef Z
}

/* cbdiso_2d.c:606-624 */
void sfo_cbdiso_2d_primitive_to_conserved(int ni, int nj, const double *primitive, double *conserved)
{
    const long NJ = nj + 2 * NG;
    _Pragma("omp parallel for")
    for (long I = NG; I < ni + NG; ++I)
        for (long J = NG; J < nj + NG; ++J) {
            const double *p = &primitive[(I * NJ + J) * 3];
            double *u = &conserved[(I * NJ + J) * 3];
            u[0] = p[0];
            u[1] = p[1] * p[0];
            u[2] = p[2] * p[0];
        }
}

/* cbdiso_2d.c:340-350 == cbdgam_2d.c:401-409 */
static double max_speed(double vx, double vy, double cs2)
{
    double cs = sqrt(cs2);
    double ax = dmax(fabs(vx - cs), fabs(vx + cs));
    double ay = dmax(fabs(vy - cs), fabs(vy + cs));
    return dmax(ax, ay);
}

/* cbdiso_2d.c:666-722 (same argument list) */
void sfo_cbdiso_2d_wavespeed(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double soundspeed2, double mach_squared, int eos_type,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    const double *primitive, double *wavespeed)
{
    sfo_mass_t m[2] = {
        {x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1},
        {x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2}};
    const long NJ = nj + 2 * NG;
    const double dx = (patch_xr - patch_xl) / ni;
    const double dy = (patch_yr - patch_yl) / nj;
    _Pragma("omp parallel for")
    for (long I = NG; I < ni + NG; ++I)
        for (long J = NG; J < nj + NG; ++J) {
            const double *p = &primitive[(I * NJ + J) * 3];
            double x = patch_xl + ((I - NG) + 0.5) * dx;
            double y = patch_yl + ((J - NG) + 0.5) * dy;
            double c2 = iso_cs2(soundspeed2, mach_squared, eos_type, x, y, m);
            wavespeed[I * NJ + J] = max_speed(p[1], p[2], c2);
        }
}

/* cbdiso_2d.c:626-664: per-zone rate of ONE mass with dt = 1 (diagnostics) */
void sfo_cbdiso_2d_point_mass_source_term(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    const double *primitive, double *cons_rate)
{
    sfo_mass_t m = {x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1};
    const long NJ = nj + 2 * NG;
    const double dx = (patch_xr - patch_xl) / ni;
    const double dy = (patch_yr - patch_yl) / nj;
    _Pragma("omp parallel for")
    for (long I = NG; I < ni + NG; ++I)
        for (long J = NG; J < nj + NG; ++J) {
            double xc = patch_xl + ((I - NG) + 0.5) * dx;
            double yc = patch_yl + ((J - NG) + 0.5) * dy;
            iso_mass_source(&m, xc, yc, 1.0, &primitive[(I * NJ + J) * 3], &cons_rate[(I * NJ + J) * 3]);
        }
}

/* ======================================================================== */
/*                         gamma-law solver (cbdgam_2d)                      */
/* ======================================================================== */

/* local disk scale height: cbdgam_2d.c:75-106 */
static double gam_height(const sfo_mass_t *m, double x, double y, const double *prim)
{
    if (m[0].mass == 0.0 && m[1].mass == 0.0) return 1.0;
    double om2 = 0.0;
    for (int p = 0; p < 2; ++p) {
        if (m[p].mass > 0.0) {
            double dx = x - m[p].x;
            double dy = y - m[p].y;
            double r2 = dx * dx + dy * dy + 1e-12;
            double r = sqrt(r2);
            om2 += m[p].mass * pow(r, -3.0);
        }
    }
    return sqrt(prim[3] / prim[0]) / sqrt(om2);
}

/* cbdgam_2d.c:347-362 */
static void gam_p2c(const double *p, double *u, double gamma)
{
    u[0] = p[0];
    u[1] = p[1] * p[0];
    u[2] = p[2] * p[0];
    u[3] = p[3] / (gamma - 1.0) + 0.5 * p[0] * (p[1] * p[1] + p[2] * p[2]);
}

/* cbdgam_2d.c:374-387 */
static void gam_flux(const double *p, const double *u, int dir, double *f)
{
    double vn = p[1 + dir];
    f[0] = vn * u[0];
    f[1] = vn * u[1] + p[3] * (dir == 0);
    f[2] = vn * u[2] + p[3] * (dir == 1);
    f[3] = vn * (u[3] + p[3]);
}

/*
 * HLLE at one face, evaluated with the two sound speeds the reference uses for
 * it: zone R (for which this is the lower face) takes cs2 of zone L, zone L
 * (for which it is the upper face) takes cs2 of zone R -- cbdgam_2d.c:613-622.
 * f_for_R / f_for_L receive the respective flux vectors.
 */
static void gam_hlle2(const double *pm, const double *pp, double cs2_L, double cs2_R, int dir,
                      double gamma, double *f_for_R, double *f_for_L)
{
    double ul[4], ur[4], fl[4], fr[4];
    gam_p2c(pm, ul, gamma);
    gam_p2c(pp, ur, gamma);
    gam_flux(pm, ul, dir, fl);
    gam_flux(pp, ur, dir, fr);
    double vl = pm[1 + dir], vr = pp[1 + dir];
    for (int k = 0; k < 2; ++k) {
        double cs = sqrt(k == 0 ? cs2_L : cs2_R);
        double *f = k == 0 ? f_for_R : f_for_L;
        double am = dmin(0.0, dmin(vl - cs, vr - cs));
        double ap = dmax(0.0, dmax(vl + cs, vr + cs));
        for (int q = 0; q < 4; ++q)
            f[q] = (fl[q] * ap - fr[q] * am - (ul[q] - ur[q]) * ap * am) / (ap - am);
    }
}

/* one point mass, gamma-law: cbdgam_2d.c:108-203 (assigns, does not accumulate) */
static void gam_mass_source(const sfo_mass_t *m, double x1, double y1, double dt, const double *prim,
                            double h, double *dc, int constant_softening, double gamma)
{
    double sigma = prim[0];
    double eps = prim[3] / sigma / (gamma - 1.0);
    double dx = x1 - m->x;
    double dy = y1 - m->y;
    double r2 = dx * dx + dy * dy;
    double dr = sqrt(r2);
    double r_sink = m->sink_radius;
    double r_soft;
    if (constant_softening) {
        r_soft = m->rsoft;
    } else if (dr > r_sink) {
        r_soft = 0.5 * h;
    } else {
        double tr = pow(1.0 - pow(dr / r_sink, 2.0), 2.0);
        r_soft = tr * r_sink + (1.0 - tr) * 0.5 * h;
    }
    double rate = (dr < 4.0 * r_sink) ? m->sink_rate * exp(-pow(dr / r_sink, 4.0)) : 0.0;
    double fnum = sigma * m->mass * pow(r2 + r_soft * r_soft, -1.5);
    double fx = -fnum * dx;
    double fy = -fnum * dy;
    double mdot = sigma * rate * -1.0;
    double vx = prim[1], vy = prim[2];
    if (m->sink_model == 1) {
        dc[0] = dt * mdot;
        dc[1] = dt * mdot * prim[1] + dt * fx;
        dc[2] = dt * mdot * prim[2] + dt * fy;
        dc[3] = dt * (mdot * eps + 0.5 * mdot * (vx * vx + vy * vy)) + dt * (fx * vx + fy * vy);
    } else if (m->sink_model == 2) {
        double rhx = dx / (dr + 1e-12);
        double rhy = dy / (dr + 1e-12);
        double dv = (vx - m->vx) * rhx + (vy - m->vy) * rhy;
        double vxs = dv * rhx + m->vx;
        double vys = dv * rhy + m->vy;
        dc[0] = dt * mdot;
        dc[1] = dt * mdot * vxs + dt * fx;
        dc[2] = dt * mdot * vys + dt * fy;
        dc[3] = dt * (mdot * eps + 0.5 * mdot * (vxs * vxs + vys * vys)) + dt * (fx * vx + fy * vy);
    } else if (m->sink_model == 3) {
        dc[0] = dt * mdot;
        dc[1] = dt * fx;
        dc[2] = dt * fy;
        dc[3] = dt * (fx * vx + fy * vy);
    } else {
        dc[0] = dc[1] = dc[2] = dc[3] = 0.0;
    }
}

/* cbdgam_2d.c:238-276 (acts on the conserved state directly) */
static void gam_buffer(double sd, double sp, double mc, double rate, double r_out, double w, int on,
                       double xc, double yc, double dt, double *u, double gamma)
{
    if (!on) return;
    double rc = sqrt(xc * xc + yc * yc);
    double r_on = r_out - w;
    if (rc > r_on) {
        double pf = sd * sqrt(mc / rc);
        double px = pf * (-yc / rc);
        double py = pf * (xc / rc);
        double ke = 0.5 * (px * px + py * py) / sd;
        double en = sp / (gamma - 1.0) + ke;
        double u0[4] = {sd, px, py, en};
        double om = sqrt(mc * pow(r_on, -3.0));
        double br = rate * om * (rc - r_on) / (r_out - r_on);
        for (int q = 0; q < 4; ++q) u[q] -= (u[q] - u0[q]) * br * dt;
    }
}

/* cbdgam_2d.c:298-317 */
static void gam_cooling(double coeff, double mach_ceiling, double dt, const double *prim, double *u, double gamma)
{
    double sigma = prim[0];
    double eps = prim[3] / prim[0] / (gamma - 1.0);
    double ec = eps * pow(1.0 + 3.0 * coeff * pow(sigma, -2.0) * pow(eps, 3.0) * dt, -1.0 / 3.0);
    double vx = prim[1], vy = prim[2];
    double ek = 0.5 * (vx * vx + vy * vy);
    ec = dmax(ec, 2.0 * ek / gamma / (gamma - 1.0) * pow(mach_ceiling, -2.0));
    u[3] += sigma * (ec - eps);
}

/* cbdgam_2d.c:319-345 */
static void gam_c2p(const double *u, double *p, double vmax, double dfloor, double pfloor, double gamma)
{
    double pres = dmax(pfloor, (u[3] - 0.5 * (u[1] * u[1] + u[2] * u[2]) / u[0]) * (gamma - 1.0));
    double vx = sgn(u[1]) * dmin(fabs(u[1] / u[0]), vmax);
    double vy = sgn(u[2]) * dmin(fabs(u[2] / u[0]), vmax);
    double rho = u[0];
    if (u[0] < dfloor) {
        rho = dfloor;
        vx = 0.0;
        vy = 0.0;
        pres = pfloor;
    }
    p[0] = rho; p[1] = vx; p[2] = vy; p[3] = pres;
}

/*
 * One RK stage of the gamma-law solver.  Restates cbdgam_2d_advance_rk,
 * cbdgam_2d.c:439-683 (same argument list, same order).
 */
void sfo_cbdgam_2d_advance_rk(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    const double *conserved_rk, const double *primitive_rd, double *primitive_wr,
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
    const double theta = 1.5; /* cbdgam_2d.c:15 */
    const int NQ = 4;
    const double gamma = gamma_law_index;
    sfo_mass_t m[2] = {
        {x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1},
        {x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2}};
    const double dx = (patch_xr - patch_xl) / ni;
    const double dy = (patch_yr - patch_yl) / nj;
    const long NI = ni + 2 * NG, NJ = nj + 2 * NG;
    const long si = NQ * NJ, sj = NQ;
    const double *P = primitive_rd;
#define Z(I, J) ((I) * si + (J) * sj)
    size_t n = (size_t)(NI * NJ * NQ);
    double *gx = calloc(n, sizeof(double));
    double *gy = calloc(n, sizeof(double));
    /* FR[I][J]: flux through face (I-1|I) as zone (I,J) sees it; FL: as zone (I-1,J) sees it */
    double *FR = calloc(n, sizeof(double)), *FL = calloc(n, sizeof(double));
    double *GR = calloc(n, sizeof(double)), *GL = calloc(n, sizeof(double));
    /* cell-centred alpha viscosity nu_cc = alpha * h(xc,yc,pcc) * sqrt(cs2cc), cbdgam_2d.c:638,644 */
    double *nucc = calloc((size_t)(NI * NJ), sizeof(double));

    _Pragma("omp parallel for")
    for (long I = 1; I < NI - 1; ++I)
        for (long J = 1; J < NJ - 1; ++J) {
            for (int q = 0; q < NQ; ++q) {
                gx[Z(I, J) + q] = slope(P[Z(I - 1, J) + q], P[Z(I, J) + q], P[Z(I + 1, J) + q], theta);
                gy[Z(I, J) + q] = slope(P[Z(I, J - 1) + q], P[Z(I, J) + q], P[Z(I, J + 1) + q], theta);
            }
        }
    if (alpha > 0.0)
        _Pragma("omp parallel for")
        for (long I = NG; I < ni + NG; ++I)
            for (long J = NG; J < nj + NG; ++J) {
                const double *pc = &P[Z(I, J)];
                double xc = patch_xl + ((I - NG) + 0.5) * dx;
                double yc = patch_yl + ((J - NG) + 0.5) * dy;
                double c2 = pc[3] / pc[0] * gamma;
                nucc[I * NJ + J] = alpha * gam_height(m, xc, yc, pc) * sqrt(c2);
            }

    for (int dir = 0; dir < 2; ++dir) {
        const double *g = dir == 0 ? gx : gy;
        double *TR = dir == 0 ? FR : GR;
        double *TL = dir == 0 ? FL : GL;
        long Ihi = ni + NG + (dir == 0), Jhi = nj + NG + (dir == 1);
        _Pragma("omp parallel for")
        for (long I = NG; I < Ihi; ++I)
            for (long J = NG; J < Jhi; ++J) {
                long L = dir == 0 ? Z(I - 1, J) : Z(I, J - 1), R = Z(I, J);
                double pm[4], pp[4];
                for (int q = 0; q < NQ; ++q) {
                    pm[q] = P[L + q] + 0.5 * g[L + q];
                    pp[q] = P[R + q] - 0.5 * g[R + q];
                }
                double c2L = P[L + 3] / P[L] * gamma; /* cbdgam_2d.c:231-236 */
                double c2R = P[R + 3] / P[R] * gamma;
                double *fR = &TR[R], *fL = &TL[R];
                gam_hlle2(pm, pp, c2L, c2R, dir, gamma, fR, fL);
                if (alpha > 0.0) { /* cbdgam_2d.c:624-667 */
                    double sl[4], sr[4];
                    strain(&gx[L], &gy[L], dx, dy, sl);
                    strain(&gx[R], &gy[R], dx, dy, sr);
                    /* face position; the reference evaluates the neighbour's height there */
                    double xf = dir == 0 ? patch_xl + ((double)(I - NG)) * dx : patch_xl + ((I - NG) + 0.5) * dx;
                    double yf = dir == 0 ? patch_yl + ((J - NG) + 0.5) * dy : patch_yl + ((double)(J - NG)) * dy;
                    double nufL = alpha * gam_height(m, xf, yf, &P[L]) * sqrt(c2L);
                    double nufR = alpha * gam_height(m, xf, yf, &P[R]) * sqrt(c2R);
                    long cL = dir == 0 ? (I - 1) * NJ + J : I * NJ + (J - 1);
                    double nuL = nucc[cL], nuR = nucc[I * NJ + J];
                    int k1 = dir == 0 ? 0 : 2; /* tau^dir_x */
                    int k2 = dir == 0 ? 1 : 3; /* tau^dir_y */
                    /* zone R's view: neighbour L at the face, itself at its centre */
                    fR[1] -= 0.5 * (nufL * P[L] * sl[k1] + nuR * P[R] * sr[k1]);
                    fR[2] -= 0.5 * (nufL * P[L] * sl[k2] + nuR * P[R] * sr[k2]);
                    fR[3] -= 0.5 * (nufL * P[L] * sl[k1] * P[L + 1] + nuR * P[R] * sr[k1] * P[R + 1]);
                    fR[3] -= 0.5 * (nufL * P[L] * sl[k2] * P[L + 2] + nuR * P[R] * sr[k2] * P[R + 2]);
                    /* zone L's view: itself at its centre, neighbour R at the face */
                    fL[1] -= 0.5 * (nuL * P[L] * sl[k1] + nufR * P[R] * sr[k1]);
                    fL[2] -= 0.5 * (nuL * P[L] * sl[k2] + nufR * P[R] * sr[k2]);
                    fL[3] -= 0.5 * (nuL * P[L] * sl[k1] * P[L + 1] + nufR * P[R] * sr[k1] * P[R + 1]);
                    fL[3] -= 0.5 * (nuL * P[L] * sl[k2] * P[L + 2] + nufR * P[R] * sr[k2] * P[R + 2]);
                }
            }
    }

    _Pragma("omp parallel for")
    for (long I = NG; I < ni + NG; ++I)
        for (long J = NG; J < nj + NG; ++J) {
            long c = Z(I, J);
            const double *pc = &P[c];
            double xc = patch_xl + ((I - NG) + 0.5) * dx;
            double yc = patch_yl + ((J - NG) + 0.5) * dy;
            double u[4], dc[4];
            double hcc = gam_height(m, xc, yc, pc);
            gam_p2c(pc, u, gamma);
            gam_buffer(buffer_surface_density, buffer_surface_pressure, buffer_central_mass, buffer_driving_rate,
                       buffer_outer_radius, buffer_onset_width, buffer_is_enabled, xc, yc, dt, u, gamma);
            for (int p = 0; p < 2; ++p) {
                gam_mass_source(&m[p], xc, yc, dt, pc, hcc, dc, constant_softening, gamma);
                for (int q = 0; q < NQ; ++q) u[q] += dc[q];
            }
            gam_cooling(cooling_coefficient, mach_ceiling, dt, pc, u, gamma);
            for (int q = 0; q < NQ; ++q) {
                /* lower faces as this zone sees them (FR/GR at c); upper faces as seen from below (FL/GL at +1) */
                double fri = FL[Z(I + 1, J) + q], fli = FR[c + q];
                double frj = GL[Z(I, J + 1) + q], flj = GR[c + q];
                u[q] -= ((fri - fli) / dx + (frj - flj) / dy) * dt;
                u[q] = (1.0 - a) * u[q] + a * conserved_rk[c + q];
            }
            gam_c2p(u, &primitive_wr[c], velocity_ceiling, density_floor, pressure_floor, gamma);
        }
    free(gx); free(gy); free(FR); free(FL); free(GR); free(GL); free(nucc);
#undef Z
}

/* cbdgam_2d.c:685-708 */
void sfo_cbdgam_2d_wavespeed(int ni, int nj, const double *primitive, double *wavespeed, double gamma_law_index)
{
    const long NJ = nj + 2 * NG;
    _Pragma("omp parallel for")
    for (long I = NG; I < ni + NG; ++I)
        for (long J = NG; J < nj + NG; ++J) {
            const double *p = &primitive[(I * NJ + J) * 4];
            double c2 = p[3] / p[0] * gamma_law_index;
            wavespeed[I * NJ + J] = max_speed(p[1], p[2], c2);
        }
}

/* cbdgam_2d.c:710-729 */
void sfo_cbdgam_2d_primitive_to_conserved(int ni, int nj, const double *primitive, double *conserved, double gamma_law_index)
{
    const long NJ = nj + 2 * NG;
    for (long I = NG; I < ni + NG; ++I)
        for (long J = NG; J < nj + NG; ++J)
            gam_p2c(&primitive[(I * NJ + J) * 4], &conserved[(I * NJ + J) * 4], gamma_law_index);
}

/* cbdgam_2d.c:731-784 */
void sfo_cbdgam_2d_point_mass_source_term(
    int ni, int nj, double patch_xl, double patch_xr, double patch_yl, double patch_yr,
    double x1, double y1, double vx1, double vy1, double mass1, double softening_length1,
    double sink_rate1, double sink_radius1, int sink_model1,
    double x2, double y2, double vx2, double vy2, double mass2, double softening_length2,
    double sink_rate2, double sink_radius2, int sink_model2,
    int which_mass, const double *primitive, double *cons_rate, int constant_softening, double gamma_law_index)
{
    sfo_mass_t m[2] = {
        {x1, y1, vx1, vy1, mass1, softening_length1, sink_rate1, sink_radius1, sink_model1},
        {x2, y2, vx2, vy2, mass2, softening_length2, sink_rate2, sink_radius2, sink_model2}};
    const long NJ = nj + 2 * NG;
    const double dx = (patch_xr - patch_xl) / ni;
    const double dy = (patch_yr - patch_yl) / nj;
    _Pragma("omp parallel for")
    for (long I = NG; I < ni + NG; ++I)
        for (long J = NG; J < nj + NG; ++J) {
            double xc = patch_xl + ((I - NG) + 0.5) * dx;
            double yc = patch_yl + ((J - NG) + 0.5) * dy;
            const double *pc = &primitive[(I * NJ + J) * 4];
            double h = gam_height(m, xc, yc, pc);
            gam_mass_source(&m[which_mass - 1], xc, yc, 1.0, pc, h, &cons_rate[(I * NJ + J) * 4],
                            constant_softening, gamma_law_index);
        }
}

/*
 * Per-function probes (tests/test_function_probes.py): the static helpers above, one call per array element, so
 * that slope, HLLE flux and primitive recovery can be held against the reference's own helpers (oracle/_ref,
 * through oracle/ref_probes.c) and against the device functions (tests/probes/) in isolation.
 */
void sfo_probe_slope(int n, double theta, const double *yl, const double *y0, const double *yr, double *g)
{
    for (int i = 0; i < n; ++i) g[i] = slope(yl[i], y0[i], yr[i], theta);
}

void sfo_probe_iso_hlle(int n, int dir, const double *pl, const double *pr, const double *cs2, double *flux)
{
    for (int i = 0; i < n; ++i) iso_hlle(pl + 3 * i, pr + 3 * i, cs2[i], dir, flux + 3 * i);
}

void sfo_probe_gam_hlle2(int n, int dir, const double *pm, const double *pp, const double *cs2_below, const double *cs2_above,
                         double gamma, double *flux_for_above, double *flux_for_below)
{
    for (int i = 0; i < n; ++i)
        gam_hlle2(pm + 4 * i, pp + 4 * i, cs2_below[i], cs2_above[i], dir, gamma, flux_for_above + 4 * i, flux_for_below + 4 * i);
}

void sfo_probe_iso_c2p(int n, const double *cons, double vmax, double dfloor, double *prim)
{
    for (int i = 0; i < n; ++i) iso_c2p(cons + 3 * i, prim + 3 * i, vmax, dfloor);
}

void sfo_probe_gam_c2p(int n, const double *cons, double vmax, double dfloor, double pfloor, double gamma, double *prim)
{
    for (int i = 0; i < n; ++i) gam_c2p(cons + 4 * i, prim + 4 * i, vmax, dfloor, pfloor, gamma);
}
