/*
 * ref_probes.c -- TEST INFRASTRUCTURE.  Exported loops over the reference's file-local helpers.
 *
 * oracle/build_ref.py appends this text to the reference's own translation unit (KERNEL_LIB_HEADER + cbdiso_2d.c
 * or cbdgam_2d.c, read where they lie under /root/reference) for the extra builds oracle/_ref/<solver>_probes.so.
 * The helpers it calls -- plm_gradient_scalar, riemann_hlle, conserved_to_primitive (cbdiso_2d.c:23-29, 352-380,
 * 268-283; cbdgam_2d.c:27-33, 416-434, 319-345) -- are `static` there and cannot be reached from outside; these
 * loops make them callable one array element at a time and add nothing else.  NCONS is the reference's macro
 * (3: cbdiso_2d, 4: cbdgam_2d), which also tells the two argument lists apart.
 */
void probe_plm_gradient_scalar(int n, const double *yl, const double *y0, const double *yr, double *g)
{
    for (int i = 0; i < n; ++i) g[i] = plm_gradient_scalar(yl[i], y0[i], yr[i]);
}

void probe_riemann_hlle(int n, int direction, const double *pl, const double *pr, const double *cs2, double gamma_law_index, double *flux)
{
    for (int i = 0; i < n; ++i) {
#if NCONS == 3
        (void)gamma_law_index;
        riemann_hlle(pl + NCONS * i, pr + NCONS * i, flux + NCONS * i, cs2[i], direction);
#else
        riemann_hlle(pl + NCONS * i, pr + NCONS * i, flux + NCONS * i, cs2[i], direction, gamma_law_index);
#endif
    }
}

void probe_conserved_to_primitive(int n, const double *cons, double velocity_ceiling, double density_floor, double pressure_floor,
                                  double gamma_law_index, double *prim)
{
    for (int i = 0; i < n; ++i) {
#if NCONS == 3
        (void)pressure_floor; (void)gamma_law_index;
        conserved_to_primitive(cons + NCONS * i, prim + NCONS * i, velocity_ceiling, density_floor);
#else
        conserved_to_primitive(cons + NCONS * i, prim + NCONS * i, velocity_ceiling, density_floor, pressure_floor, gamma_law_index);
#endif
    }
}
