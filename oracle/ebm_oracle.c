/* CPU oracle (plain C, fp64) for the energy-balance time-stepping hot path.
 *
 * TEST INFRASTRUCTURE ONLY — never linked into or called from the shipped HIP path.  Used by
 * tests/ (parity checker at sizes the NumPy oracle is too slow for) and by bench.py's
 * `cpu_baseline` leg (kind "port": this is a restatement, not the Julia package).
 *
 * PARITY UNPINNED: the Julia reference cannot run in this pipeline and its golden file is
 * absent (see oracle/ebm_oracle.py header).  This file is pinned to oracle/ebm_oracle.py bit
 * for bit (tests/test_oracle.py), which in turn is pinned to the known answers the reference
 * text provides.
 *
 * Operation order follows the reference expression by expression; build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math
 * so no FMA contraction or re-association happens.  Citations are to /root/reference.
 *
 * Layout: every state array is [ncol][nx], latitude contiguous (one Julia Vec per column).
 *
 * Two builds of this one source (oracle/Makefile): `real` = double, entry points ebmo_* — THE oracle;
 * and -DEBMO_LONG: `real` = long double (x87 80-bit), entry points ebmol_* — the same formulas, the
 * same fp64 inputs, outputs rounded to fp64 once at the end of a run, ~2000x less rounding error in
 * between.  The extended build is a measuring stick for rounding error only (how far the fp64 oracle
 * and the GPU each are from the exactly-evaluated discrete model, tests/test_error_budget.py); it is
 * not a second opinion on the transcription.
 */
#include <math.h>
#ifdef EBMO_LONG
typedef long double real;
#define EBMO(name) ebmol_##name
#define R_POW powl
#define R_COPYSIGN copysignl
#else
typedef double real;
#define EBMO(name) ebmo_##name
#define R_POW pow
#define R_COPYSIGN copysign
#endif
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { P_D, P_A, P_B, P_cw, P_S0, P_S1, P_S2, P_a0, P_a2, P_ai, P_Fb, P_k, P_Lf, P_F, P_cg,
       P_tau, P_Tm, P_m1, P_m2, P_alpha, P_rl, P_Dmin, P_Dmax, P_hmin, P_kappa, P_COUNT };

#define MAX_NEWTON 1000   /* NonlinearSolve's default maxiters (src/miz.jl:55-60 passes none) */

/* Julia min(::Float64, ::Float64): NaN-propagating, -0.0 < +0.0 */
static inline real jl_min(real x, real y) {
    real diff = x - y;
    real am = signbit(diff) ? x : y;
    return (isnan(x) || isnan(y)) ? diff : am;
}
/* Julia clamp(x, lo, hi) = ifelse(x > hi, hi, ifelse(x < lo, lo, x)) */
static inline real jl_clamp(real x, real lo, real hi) {
    return x > hi ? hi : (x < lo ? lo : x);
}
/* Julia *(x::Float64, b::Bool) */
static inline real bool_mul(real x, int b) { return b ? x : R_COPYSIGN((real)0.0, x); }

/* ---- diffusion geometry, src/infrastructure.jl:477-492 (identity), :509-518 (other) ---- */
typedef struct {
    int kind, nx;              /* kind 0 = identity, 1 = non-uniform ("sin") */
    real D;
    real *c0, *c1, *c2, *c3, *c4; /* identity: sub, diag, sup ; else mph, mmh, dxp, dxm, w */
    real *lo, *di, *up;      /* plain tridiagonal coefficients (Jacobian assembly only) */
} Geom;

static void geom_free(Geom *g) {
    free(g->c0); free(g->c1); free(g->c2); free(g->c3); free(g->c4);
    free(g->lo); free(g->di); free(g->up);
}

static void geom_init(Geom *g, int kind, int nx, const real *x, real D) {
    g->kind = kind; g->nx = nx; g->D = D;
    size_t nb = sizeof(real) * (size_t)nx;
    g->c0 = calloc(1, nb); g->c1 = calloc(1, nb); g->c2 = calloc(1, nb);
    g->c3 = calloc(1, nb); g->c4 = calloc(1, nb);
    g->lo = calloc(1, nb); g->di = calloc(1, nb); g->up = calloc(1, nb);
    if (kind == 0) {
        real dx = 1.0 / nx;
        real *lam = calloc(1, nb);
        for (int i = 1; i < nx; ++i) {
            real xb = (real)i / nx;
            lam[i - 1] = (1.0 - xb * xb) / (dx * dx);
        }
        for (int k = 0; k < nx; ++k) {
            real sub = k > 0 ? lam[k - 1] : 0.0;
            real sup = k < nx - 1 ? lam[k] : 0.0;
            real l1 = k > 0 ? -lam[k - 1] : 0.0;
            real l2 = k < nx - 1 ? -lam[k] : 0.0;
            real l3 = (-l1) - l2;
            g->c0[k] = D * sub; g->c1[k] = D * (-l3); g->c2[k] = D * sup;
            g->lo[k] = g->c0[k]; g->di[k] = g->c1[k]; g->up[k] = g->c2[k];
        }
        free(lam);
    } else {
        for (int k = 0; k < nx; ++k) {
            real xk = x[k];
            real xm = k > 0 ? x[k - 1] : -x[0];
            real xp = k < nx - 1 ? x[k + 1] : 2.0 - x[nx - 1];
            real xxph = (xp + xk) / 2.0, xxmh = (xk + xm) / 2.0;
            g->c0[k] = 1.0 - xxph * xxph;
            g->c1[k] = 1.0 - xxmh * xxmh;
            g->c2[k] = xp - xk;
            g->c3[k] = xk - xm;
            g->c4[k] = xxph - xxmh;
            real up = D * g->c0[k] / (g->c2[k] * g->c4[k]);
            real lo = D * g->c1[k] / (g->c3[k] * g->c4[k]);
            if (k == nx - 1) up = 0.0;
            if (k == 0) lo = 0.0;
            g->lo[k] = lo; g->up[k] = up; g->di[k] = -(lo + up);
        }
    }
}

/* base + D d/dx[(1-x^2) d temp/dx] at cell k; src/infrastructure.jl:495-497 and :521-524 */
static inline real diffusion_add(const Geom *g, real base, const real *temp, int k) {
    int nx = g->nx;
    if (g->kind == 0) {
        real y = 0.0;
        if (k > 0) y = y + g->c0[k] * temp[k - 1];
        y = y + g->c1[k] * temp[k];
        if (k < nx - 1) y = y + g->c2[k] * temp[k + 1];
        return base + y;
    }
    real dTp = k < nx - 1 ? temp[k + 1] - temp[k] : 0.0;
    real dTm = k > 0 ? temp[k] - temp[k - 1] : 0.0;
    return base + (g->D * ((g->c0[k] * dTp) / g->c2[k] - (g->c1[k] * dTm) / g->c3[k])) / g->c4[k];
}

/* Thomas algorithm, same operation order as oracle/ebm_oracle.py:thomas */
static void thomas(int n, const real *a, const real *b, const real *c, const real *d,
                   real *cp, real *dp, real *xs) {
    cp[0] = c[0] / b[0];
    dp[0] = d[0] / b[0];
    for (int i = 1; i < n; ++i) {
        real den = b[i] - a[i] * cp[i - 1];
        cp[i] = c[i] / den;
        dp[i] = (d[i] - a[i] * dp[i - 1]) / den;
    }
    xs[n - 1] = dp[n - 1];
    for (int i = n - 2; i >= 0; --i) xs[i] = dp[i] - cp[i] * xs[i + 1];
}

static void thomas_export(int n, const real *a, const real *b, const real *c, const real *d,
                 real *xs) {
    real *cp = malloc(sizeof(real) * n), *dp = malloc(sizeof(real) * n);
    thomas(n, a, b, c, d, cp, dp, xs);
    free(cp); free(dp);
}

typedef struct {
    real *Tw, *Ti, *hp, *dd, *r, *rhs, *g, *a, *b, *c, *d, *cp, *dp, *v, *tb;
    unsigned char *s;
} Work;

static void work_init(Work *w, int nx) {
    size_t nb = sizeof(real) * (size_t)nx;
    real **ps[] = {&w->Tw, &w->Ti, &w->hp, &w->dd, &w->r, &w->rhs, &w->g, &w->a, &w->b,
                     &w->c, &w->d, &w->cp, &w->dp, &w->v, &w->tb};
    for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]); ++i) *ps[i] = malloc(nb);
    w->s = malloc((size_t)nx);
}
static void work_free(Work *w) {
    free(w->Tw); free(w->Ti); free(w->hp); free(w->dd); free(w->r); free(w->rhs); free(w->g);
    free(w->a); free(w->b); free(w->c); free(w->d); free(w->cp); free(w->dp); free(w->v);
    free(w->tb); free(w->s);
}

static inline real insolation(const real *p, real x, real ct) {
    /* src/miz.jl:11 */
    return p[P_S0] - p[P_S1] * x * ct - p[P_S2] * (x * x);
}

/* One MIZ step for one column; src/miz.jl:150-196.  Returns number of tridiagonal solves,
 * negative if the active-set iteration hit MAX_NEWTON. */
static int miz_step_col(const Geom *g, const real *p, int nx, const real *x, real dt,
                        real ct, real f, real *Ei, real *Ew, real *h, real *D,
                        real *phi, real *T0, real *Tw_o, real *Ti_o, real *n_o,
                        real *E_o, real *T_o, Work *w, int imex, const real *zon) {
    const real Tm = p[P_Tm], cw = p[P_cw], A = p[P_A], B = p[P_B], ai = p[P_ai];
    const real Lf = p[P_Lf], alpha = p[P_alpha], hmin = p[P_hmin], Dmin = p[P_Dmin];
    const real Dmax = p[P_Dmax], Fb = p[P_Fb];
    /* water temperature, src/miz.jl:30,157 */
    for (int k = 0; k < nx; ++k) {
        real tw = Tm + Ew[k] / ((1.0 - phi[k]) * cw);
        w->Tw[k] = isnan(tw) ? 0.0 : tw;
    }
    /* solveTi, src/miz.jl:47-68 — active-set Newton on the piecewise-linear T0eq (:33-45) */
    for (int k = 0; k < nx; ++k) {
        w->hp[k] = (h[k] == 0.0) ? hmin : h[k];
        w->dd[k] = p[P_k] / w->hp[k] + B;
        w->r[k] = (1.0 - phi[k]) * (w->Tw[k] - Tm);
        w->s[k] = T0[k] < Tm;
    }
    for (int k = 0; k < nx; ++k) {
        real rm = k > 0 ? w->r[k - 1] : 0.0, rp = k < nx - 1 ? w->r[k + 1] : 0.0;
        w->rhs[k] = ai * insolation(p, x[k], ct) - A + (g->lo[k] * rm + g->di[k] * w->r[k] + g->up[k] * rp) + f;
        w->d[k] = -w->rhs[k];
    }
    int nit = 0, ok = 0;
    while (nit < MAX_NEWTON) {
        ++nit;
        for (int k = 0; k < nx; ++k) w->g[k] = w->s[k] ? phi[k] : 0.0;
        for (int k = 0; k < nx; ++k) {
            real gm = k > 0 ? w->g[k - 1] : 0.0, gp = k < nx - 1 ? w->g[k + 1] : 0.0;
            w->a[k] = g->lo[k] * gm;
            w->c[k] = g->up[k] * gp;
            w->b[k] = g->di[k] * w->g[k] - w->dd[k];
        }
        thomas(nx, w->a, w->b, w->c, w->d, w->cp, w->dp, w->v);
        int same = 1;
        for (int k = 0; k < nx; ++k) {
            unsigned char sn = w->v[k] < 0.0;
            if (sn != w->s[k]) same = 0;
            w->s[k] = sn;
        }
        if (same) { ok = 1; break; }
    }
    for (int k = 0; k < nx; ++k) {
        T0[k] = w->v[k] + Tm;
        real ti = jl_min(T0[k], Tm);
        w->Ti[k] = (h[k] == 0.0) ? 0.0 : ti;
        w->tb[k] = w->Ti[k] * phi[k] + (1.0 - phi[k]) * w->Tw[k];           /* Tbar :21-26 */
    }
    const real Tm_pow = R_POW(Tm, p[P_m2]);
    const real c_latmelt = -M_PI / 2.0 * alpha;
    const real c_dn = Lf * alpha * (Dmin * Dmin) * hmin;
    const real c_weld = p[P_kappa] * alpha / 4.0;
    const real c_ht = -1.0 / Lf;
    const real two_rl = 2.0 * p[P_rl];
    if (imex) {
        /* EXTENSION, not in the reference: oracle/ebm_oracle.py:implicit_diffusion_correction (its
         * definition).  dE = dt*(phi*Fvi + (1-phi)*Fvw), (I - (dt/cw)*Dif) dE_new = dE, and the diffusion
         * term of both vertical fluxes below is corrected by (dE_new - dE)/dt.  Same expressions as the
         * main loop. */
        const real theta = dt / cw;
        for (int k = 0; k < nx; ++k) {
            const real xk = x[k], ph = phi[k];
            real S = insolation(p, xk, ct);
            real L = A + B * (w->tb[k] - Tm);
            real sol_i = 0.0 + ai * S;
            real sol_w = 0.0 + (p[P_a0] - p[P_a2] * (xk * xk)) * S;
            real dif = diffusion_add(g, 0.0, w->tb, k);
            if (zon) dif = dif + zon[k];                                   /* coupling experiment only (miz2d_run_impl) */
            real Fvi = sol_i - L + dif + Fb + f;
            real Fvw = sol_w - L + dif + Fb + f;
            w->d[k] = (ph * Fvi + (1.0 - ph) * Fvw) * dt;
            w->a[k] = -(theta * g->lo[k]);
            w->c[k] = -(theta * g->up[k]);
            w->b[k] = 1.0 + theta * (g->lo[k] + g->up[k]);
        }
        thomas(nx, w->a, w->b, w->c, w->d, w->cp, w->dp, w->g);
        for (int k = 0; k < nx; ++k) w->g[k] = (w->g[k] - w->d[k]) / dt;
    }
    for (int k = 0; k < nx; ++k) {
        const real xk = x[k], ph = phi[k], hk = h[k], Dk = D[k], Tw = w->Tw[k], Ti = w->Ti[k];
        /* num :83-87 */
        real n = ph / (alpha * (Dk * Dk));
        if (Dk == 0.0) n = 0.0;
        /* vert_flux :96-101 */
        real S = insolation(p, xk, ct);
        real L = A + B * (w->tb[k] - Tm);
        real sol_i = 0.0 + ai * S;
        real sol_w = 0.0 + (p[P_a0] - p[P_a2] * (xk * xk)) * S;
        real dif = diffusion_add(g, 0.0, w->tb, k);
        if (zon) dif = dif + zon[k];
        if (imex) dif = dif + w->g[k];
        real Fvi = sol_i - L + dif + Fb + f;
        real Fvw = sol_w - L + dif + Fb + f;
        /* lat_flux :103-107, wlat :71 */
        real wl = p[P_m1] * (Tw - Tm_pow);
        real Flat = ph * hk * Lf * wl * M_PI / (alpha * Dk);
        if (Dk == 0.0) Flat = 0.0;
        /* forward Euler + redistribution :166-170, :109-117 */
        real rEi = Ei[k] + (ph * Fvi + Flat) * dt;
        real rEw = Ew[k] + ((1.0 - ph) * Fvw - Flat) * dt;
        real cEi = jl_clamp(rEi, -INFINITY, 0.0);
        real cEw = jl_clamp(rEw, 0.0, INFINITY);
        real psiEidt = rEi - cEi, psiEwdt = rEw - cEw;
        real Ei_n = cEi + psiEwdt, Ew_n = cEw + psiEidt;
        /* area_lead :90-93 */
        real Dr = Dk + two_rl;
        real ring = alpha * n * (Dr * Dr - Dk * Dk);
        real Al = jl_min(ring, 1.0 - ph);
        /* split_psiEw :120-125 on psiEwdt/dt (:173) */
        real psi = psiEwdt / dt;
        real Ql = Al / (1.0 - ph) * psi;
        if (ph == 1.0) Ql = 0.0;
        real Qp = psi - Ql;
        /* psinplus :127, :174 */
        real dn = dt * (-Qp / c_dn);
        /* D_t :140-146 */
        real lat_melt = c_latmelt * wl;
        real lat_grow = -Dk / (2.0 * Lf * hk * ph) * Ql;
        real weld = c_weld * ph * (Dk * Dk * Dk);
        if (hk == 0.0) lat_grow = 0.0;
        real rD = Dk + (lat_melt + lat_grow + weld) * dt;
        /* average :129-134, clamp!, zeroref! (:176-178) */
        real total = n + dn;
        real D_n = (n * rD + dn * Dmin) / total;
        if (total == 0.0) D_n = 0.0;
        D_n = jl_clamp(D_n, Dmin, Dmax);
        if (Ei_n == 0.0) D_n = 0.0;
        /* thickness :179-181 */
        real rh = hk + (c_ht * Fvi) * dt;
        rh = jl_clamp(rh, 0.0, INFINITY);
        real h_n = (n * rh + dn * hmin) / total;
        if (total == 0.0) h_n = 0.0;
        /* concentration :74-80 */
        real phi_n = -Ei_n / (Lf * h_n);
        if (h_n == 0.0) phi_n = 0.0;
        if (phi_n > 1.0) phi_n = 1.0;
        if (h_n == 0.0) Ei_n = 0.0;                                        /* :185 */
        E_o[k] = phi_n * Ei_n + (1.0 - phi_n) * Ew_n;                      /* :186 */
        T_o[k] = Ti * phi_n + (1.0 - phi_n) * Tw;                          /* :187 */
        Ti_o[k] = (Ei_n == 0.0) ? NAN : Ti;                                /* :193 */
        Tw_o[k] = (phi_n > 0.99) ? NAN : Tw;                               /* :194 */
        n_o[k] = n;
        Ei[k] = Ei_n; Ew[k] = Ew_n; h[k] = h_n; D[k] = D_n; phi[k] = phi_n;
    }
    return ok ? nit : -nit;
}

/* nsteps MIZ steps on ncol independent columns.  ct[s] = cos(2.0*pi*t_s), ft[s] the scalar
 * forcing of step s, fcol[c] a per-column offset (forcing = ft[s] + fcol[c]; pass NULL for 0).
 * counters[0] += tridiagonal solves, counters[1] += steps whose iteration hit MAX_NEWTON. */
static int miz_run_impl(int kind, int nx, int ncol, const real *x, const real *par, real dt,
                 int nsteps, const real *ct, const real *ft, const real *fcol,
                 real *Ei, real *Ew, real *h, real *D, real *phi, real *T0,
                 real *Tw, real *Ti, real *n, real *E, real *T, long long *counters,
                 int nthreads, int imex) {
    Geom g;
    geom_init(&g, kind, nx, x, par[P_D]);
    long long solves = 0, fails = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel reduction(+ : solves, fails)
#endif
    {
        Work w;
        work_init(&w, nx);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int c = 0; c < ncol; ++c) {
            size_t o = (size_t)c * nx;
            for (int s = 0; s < nsteps; ++s) {
                real f = fcol ? ft[s] + fcol[c] : ft[s];
                int r = miz_step_col(&g, par, nx, x, dt, ct[s], f, Ei + o, Ew + o, h + o, D + o,
                                     phi + o, T0 + o, Tw + o, Ti + o, n + o, E + o, T + o, &w, imex, NULL);
                solves += r < 0 ? -r : r;
                fails += r < 0;
            }
        }
        work_free(&w);
    }
    geom_free(&g);
    if (counters) { counters[0] += solves; counters[1] += fails; }
    (void)nthreads;
    return 0;
}

/* ---- EXTENSION, not in the reference: the zonal diffusion substep (ebm_zonal_diffusion, defined in include/ebm_hip.h) ----
 * ncol = nmember*nlon columns, column = member*nlon + longitude, periodic in longitude.  For every member and latitude k:
 *     (1 + 2 a_k) U_l - a_k (U_{l-1} + U_{l+1}) = T_l,   a_k = (dt/cw) D / ((1 - x_k)(1 + x_k) dlambda^2),   dlambda = 2 pi / nlon
 *     Z_l = (U_l - T_l) cw/dt
 * The checker's restatement solves the periodic system by Thomas + Sherman-Morrison (two solves with the matrix whose
 * corners are folded into the first and last diagonal entries), a different algorithm from the kernel's. */
static void zonal_impl(int nx, int nlon, int nmember, const real *x, real D, real cw, real dt, const real *T, real *U,
                       real *Z) {
    const real dl = 2.0 * M_PI / nlon, theta = dt / cw;
    real *a = malloc(sizeof(real) * (size_t)nlon * 8);
    real *sub = a, *dia = a + nlon, *sup = a + 2 * nlon, *rhs = a + 3 * nlon, *cp = a + 4 * nlon, *dp = a + 5 * nlon,
         *y = a + 6 * nlon, *q = a + 7 * nlon;
    for (int m = 0; m < nmember; ++m)
        for (int k = 0; k < nx; ++k) {
            const real ak = theta * D / (((1.0 - x[k]) * (1.0 + x[k])) * (dl * dl));
            const real B = 1.0 + 2.0 * ak, gam = -B;
            for (int l = 0; l < nlon; ++l) {
                sub[l] = -ak; sup[l] = -ak; dia[l] = B;
                rhs[l] = T[((size_t)m * nlon + l) * nx + k];
            }
            dia[0] = B - gam;
            dia[nlon - 1] = B - (ak * ak) / gam;
            thomas(nlon, sub, dia, sup, rhs, cp, dp, y);
            for (int l = 0; l < nlon; ++l) rhs[l] = 0.0;
            rhs[0] = gam;
            rhs[nlon - 1] = -ak;
            thomas(nlon, sub, dia, sup, rhs, cp, dp, q);
            /* v = (1, 0, ..., 0, -a/gam) */
            const real fact = (y[0] + (-ak / gam) * y[nlon - 1]) / (1.0 + q[0] + (-ak / gam) * q[nlon - 1]);
            for (int l = 0; l < nlon; ++l) {
                const size_t o = ((size_t)m * nlon + l) * nx + k;
                const real u = y[l] - fact * q[l];
                if (U) U[o] = u;
                if (Z) Z[o] = (u - T[o]) * (cw / dt);
            }
        }
    free(a);
}

/* CHECKER-ONLY EXPERIMENT (tests/test_oracle_zonal.py; nothing in the library corresponds to it): the zonal substep
 * coupled to the column step by operator splitting, on nmember grids of nlon longitudes — per step the zonal substep on
 * the T the previous step left (T is input AND output), then the column step of the implicit-diffusion extension with Z
 * added to the diffusion term of both vertical fluxes.  Converges on open water, unstable over thin new ice: the
 * measurement behind the header's statement of why the library ships the operator and not such a model. */
static int miz2d_run_impl(int kind, int nx, int nlon, int nmember, const real *x, const real *par, real dt, int nsteps,
                          const real *ct, const real *ft, const real *fcol, real *Ei, real *Ew, real *h, real *D,
                          real *phi, real *T0, real *Tw, real *Ti, real *n, real *E, real *T, long long *counters) {
    Geom g;
    geom_init(&g, kind, nx, x, par[P_D]);
    const int ncol = nlon * nmember;
    real *zon = malloc(sizeof(real) * (size_t)ncol * nx);
    Work w;
    work_init(&w, nx);
    long long solves = 0, fails = 0;
    for (int s = 0; s < nsteps; ++s) {
        zonal_impl(nx, nlon, nmember, x, par[P_D], par[P_cw], dt, T, NULL, zon);
        for (int c = 0; c < ncol; ++c) {
            size_t o = (size_t)c * nx;
            real f = fcol ? ft[s] + fcol[c] : ft[s];
            int r = miz_step_col(&g, par, nx, x, dt, ct[s], f, Ei + o, Ew + o, h + o, D + o, phi + o, T0 + o, Tw + o,
                                 Ti + o, n + o, E + o, T + o, &w, 1, zon + o);
            solves += r < 0 ? -r : r;
            fails += r < 0;
        }
    }
    work_free(&w);
    free(zon);
    geom_free(&g);
    if (counters) { counters[0] += solves; counters[1] += fails; }
    return 0;
}

/* Residual of the reference's T0eq (src/miz.jl:33-45) at a given T0, one column — lets tests
 * check a T0 against the reference solver's own acceptance criterion (abstol 1e-8). */
static void T0eq_impl(int kind, int nx, const real *x, const real *par, real ct, real f,
               const real *h, const real *Ew, const real *phi, const real *T0,
               real *res) {
    Geom g;
    geom_init(&g, kind, nx, x, par[P_D]);
    const real Tm = par[P_Tm];
    real *tb = malloc(sizeof(real) * nx);
    for (int k = 0; k < nx; ++k) {
        real tw = Tm + Ew[k] / ((1.0 - phi[k]) * par[P_cw]);
        if (isnan(tw)) tw = 0.0;
        tb[k] = jl_min(T0[k], Tm) * phi[k] + (1.0 - phi[k]) * tw;
    }
    for (int k = 0; k < nx; ++k) {
        real hp = (h[k] == 0.0) ? par[P_hmin] : h[k];
        real v = par[P_k] * (Tm - T0[k]) / hp;
        v = v + par[P_ai] * insolation(par, x[k], ct);
        v = v + ((-par[P_A]) - par[P_B] * (T0[k] - Tm));
        v = diffusion_add(&g, v, tb, k);
        res[k] = v + f;
    }
    free(tb);
    geom_free(&g);
}

/* ---- classic model, src/classic.jl ---- */
static int classic_run_impl(int nx, int ncol, const real *x, const real *par, real dt, int nsteps,
                     const real *ct_i, const real *ct_ip1, const real *ft,
                     const real *fcol, real *E, real *Tg, real *T, real *h,
                     int nthreads) {
    Geom g;
    geom_init(&g, 0, nx, x, 1.0);                       /* get_diffop(nx), unscaled */
    const real A = par[P_A], cw = par[P_cw], ai = par[P_ai], Lf = par[P_Lf], Fb = par[P_Fb];
    const real cg_tau = par[P_cg] / par[P_tau];       /* get_statics :18-29 */
    const real dt_tau = dt / par[P_tau];
    const real dc = dt_tau * cg_tau;
    const real dtD = dt * par[P_D];
    const real one = 1.0 + dt_tau;
    const real M = par[P_B] + cg_tau;
    const real kLf = par[P_k] * par[P_Lf];
    size_t nb = sizeof(real) * (size_t)nx;
    real *ksub = malloc(nb), *kdiag = malloc(nb), *ksup = malloc(nb), *aw = malloc(nb),
           *Sb = malloc(nb);
    for (int k = 0; k < nx; ++k) {
        ksub[k] = 0.0 - (dtD * g.c0[k]) / par[P_cg];
        ksup[k] = 0.0 - (dtD * g.c2[k]) / par[P_cg];
        kdiag[k] = one - (dtD * g.c1[k]) / par[P_cg];
        aw[k] = par[P_a0] - par[P_a2] * (x[k] * x[k]);
        Sb[k] = par[P_S0] - par[P_S2] * (x[k] * x[k]);
    }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        real *b = malloc(nb), *d = malloc(nb), *cp = malloc(nb), *dp = malloc(nb),
               *xs = malloc(nb);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int c = 0; c < ncol; ++c) {
            real *Ec = E + (size_t)c * nx, *Tgc = Tg + (size_t)c * nx;
            real *Tc = T + (size_t)c * nx, *hc = h + (size_t)c * nx;
            for (int s = 0; s < nsteps; ++s) {
                real f = fcol ? ft[s] + fcol[c] : ft[s];
                for (int k = 0; k < nx; ++k) {          /* src/classic.jl:47-65 */
                    real Ek = Ec[k];
                    real S_i = Sb[k] - (par[P_S1] * ct_i[s]) * x[k];
                    real S_ip1 = Sb[k] - (par[P_S1] * ct_ip1[s]) * x[k];
                    real alpha = bool_mul(aw[k], Ek > 0.0) + bool_mul(ai, Ek < 0.0);
                    real C = alpha * S_i + cg_tau * Tgc[k] - A + f;
                    real T0 = C / (M - kLf / Ek);
                    real Tk = bool_mul(Ek / cw, Ek >= 0.0) + bool_mul(bool_mul(T0, Ek < 0.0), T0 < 0.0);
                    Ek = Ek + dt * (C - M * Tk + Fb);
                    real den = M - kLf / Ek;
                    real q = bool_mul(bool_mul(dc / den, T0 < 0.0), Ek < 0.0);
                    d[k] = Tgc[k] + dt_tau * (bool_mul(Ek / cw, Ek >= 0.0) +
                               bool_mul(bool_mul((ai * S_ip1 - A + f) / den, T0 < 0.0), Ek < 0.0));
                    b[k] = kdiag[k] - q;
                    Ec[k] = Ek; Tc[k] = Tk;
                    hc[k] = bool_mul(-Ek / Lf, Ek < 0.0);
                }
                thomas(nx, ksub, b, ksup, d, cp, dp, xs);
                memcpy(Tgc, xs, nb);
            }
        }
        free(b); free(d); free(cp); free(dp); free(xs);
    }
    free(ksub); free(kdiag); free(ksup); free(aw); free(Sb);
    geom_free(&g);
    (void)nthreads;
    return 0;
}

/* Geometry export so tests can compare the C and NumPy restatements coefficient by
 * coefficient.  out: 8 arrays of nx (c0..c4, lo, di, up). */
static void geometry_impl(int kind, int nx, const real *x, real D, real *out) {
    Geom g;
    geom_init(&g, kind, nx, x, D);
    real *src[8] = {g.c0, g.c1, g.c2, g.c3, g.c4, g.lo, g.di, g.up};
    for (int i = 0; i < 8; ++i) memcpy(out + (size_t)i * nx, src[i], sizeof(real) * nx);
    geom_free(&g);
}

int EBMO(max_threads)(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- exported entry points: fp64 arrays in and out ----------------------------------------------
 * double build: the implementations above, called directly (no copies).  Extended build: inputs are
 * widened exactly, the whole run is carried in `real`, results are rounded to fp64 once at the end. */
#ifdef EBMO_LONG
static real *widen(const double *v, size_t n) {
    if (!v) return NULL;
    real *o = malloc(sizeof(real) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) o[i] = v[i];
    return o;
}
static void narrow(double *dst, const real *src, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i];
}
#define WIDEN(name, n) real *name##_r = widen(name, n)
#define RARG(name) name##_r
#define NARROW_FREE(name, n) do { if (name) narrow(name, name##_r, n); free(name##_r); } while (0)
#define JUST_FREE(name) free(name##_r)
#else
#define WIDEN(name, n) (void)0
#define RARG(name) name
#define NARROW_FREE(name, n) (void)0
#define JUST_FREE(name) (void)0
#endif

void EBMO(thomas)(int n, const double *a, const double *b, const double *c, const double *d, double *xs) {
    WIDEN(a, n); WIDEN(b, n); WIDEN(c, n); WIDEN(d, n); WIDEN(xs, n);
    thomas_export(n, RARG(a), RARG(b), RARG(c), RARG(d), RARG(xs));
    JUST_FREE(a); JUST_FREE(b); JUST_FREE(c); JUST_FREE(d); NARROW_FREE(xs, n);
}

/* imex != 0: the implicit-diffusion EXTENSION (see miz_step_col), not the reference's scheme */
int EBMO(miz_run)(int kind, int nx, int ncol, const double *x, const double *par, double dt,
                  int nsteps, const double *ct, const double *ft, const double *fcol,
                  double *Ei, double *Ew, double *h, double *D, double *phi, double *T0,
                  double *Tw, double *Ti, double *n, double *E, double *T, long long *counters,
                  int nthreads, int imex) {
    const size_t N = (size_t)ncol * nx;
    (void)N;
    WIDEN(x, nx); WIDEN(par, P_COUNT); WIDEN(ct, nsteps); WIDEN(ft, nsteps); WIDEN(fcol, ncol);
    WIDEN(Ei, N); WIDEN(Ew, N); WIDEN(h, N); WIDEN(D, N); WIDEN(phi, N); WIDEN(T0, N);
    WIDEN(Tw, N); WIDEN(Ti, N); WIDEN(n, N); WIDEN(E, N); WIDEN(T, N);
    int rc = miz_run_impl(kind, nx, ncol, RARG(x), RARG(par), dt, nsteps, RARG(ct), RARG(ft), RARG(fcol),
                          RARG(Ei), RARG(Ew), RARG(h), RARG(D), RARG(phi), RARG(T0), RARG(Tw), RARG(Ti),
                          RARG(n), RARG(E), RARG(T), counters, nthreads, imex);
    JUST_FREE(x); JUST_FREE(par); JUST_FREE(ct); JUST_FREE(ft); JUST_FREE(fcol);
    NARROW_FREE(Ei, N); NARROW_FREE(Ew, N); NARROW_FREE(h, N); NARROW_FREE(D, N); NARROW_FREE(phi, N);
    NARROW_FREE(T0, N); NARROW_FREE(Tw, N); NARROW_FREE(Ti, N); NARROW_FREE(n, N); NARROW_FREE(E, N);
    NARROW_FREE(T, N);
    return rc;
}

/* the zonal diffusion substep (ebm_zonal_diffusion): U and/or Z (either may be NULL), [nmember*nlon][nx] */
void EBMO(zonal)(int nx, int nlon, int nmember, const double *x, double D, double cw, double dt, const double *T,
                 double *U, double *Z) {
    const size_t N = (size_t)nlon * nmember * nx;
    (void)N;
    WIDEN(x, nx); WIDEN(T, N);
#ifdef EBMO_LONG
    real *U_r = U ? malloc(sizeof(real) * N) : NULL, *Z_r = Z ? malloc(sizeof(real) * N) : NULL;
#endif
    zonal_impl(nx, nlon, nmember, RARG(x), D, cw, dt, RARG(T), RARG(U), RARG(Z));
#ifdef EBMO_LONG
    if (U) { narrow(U, U_r, N); free(U_r); }
    if (Z) { narrow(Z, Z_r, N); free(Z_r); }
#endif
    JUST_FREE(x); JUST_FREE(T);
}

/* the coupling experiment: T is the previous step's output temperature on entry and the last step's on exit */
int EBMO(miz2d_run)(int kind, int nx, int nlon, int nmember, const double *x, const double *par, double dt,
                    int nsteps, const double *ct, const double *ft, const double *fcol,
                    double *Ei, double *Ew, double *h, double *D, double *phi, double *T0,
                    double *Tw, double *Ti, double *n, double *E, double *T, long long *counters) {
    const int ncol = nlon * nmember;
    const size_t N = (size_t)ncol * nx;
    (void)N;
    WIDEN(x, nx); WIDEN(par, P_COUNT); WIDEN(ct, nsteps); WIDEN(ft, nsteps); WIDEN(fcol, ncol);
    WIDEN(Ei, N); WIDEN(Ew, N); WIDEN(h, N); WIDEN(D, N); WIDEN(phi, N); WIDEN(T0, N);
    WIDEN(Tw, N); WIDEN(Ti, N); WIDEN(n, N); WIDEN(E, N); WIDEN(T, N);
    int rc = miz2d_run_impl(kind, nx, nlon, nmember, RARG(x), RARG(par), dt, nsteps, RARG(ct), RARG(ft), RARG(fcol),
                            RARG(Ei), RARG(Ew), RARG(h), RARG(D), RARG(phi), RARG(T0), RARG(Tw), RARG(Ti),
                            RARG(n), RARG(E), RARG(T), counters);
    JUST_FREE(x); JUST_FREE(par); JUST_FREE(ct); JUST_FREE(ft); JUST_FREE(fcol);
    NARROW_FREE(Ei, N); NARROW_FREE(Ew, N); NARROW_FREE(h, N); NARROW_FREE(D, N); NARROW_FREE(phi, N);
    NARROW_FREE(T0, N); NARROW_FREE(Tw, N); NARROW_FREE(Ti, N); NARROW_FREE(n, N); NARROW_FREE(E, N);
    NARROW_FREE(T, N);
    return rc;
}

void EBMO(T0eq)(int kind, int nx, const double *x, const double *par, double ct, double f,
                const double *h, const double *Ew, const double *phi, const double *T0, double *res) {
    WIDEN(x, nx); WIDEN(par, P_COUNT); WIDEN(h, nx); WIDEN(Ew, nx); WIDEN(phi, nx); WIDEN(T0, nx); WIDEN(res, nx);
    T0eq_impl(kind, nx, RARG(x), RARG(par), ct, f, RARG(h), RARG(Ew), RARG(phi), RARG(T0), RARG(res));
    JUST_FREE(x); JUST_FREE(par); JUST_FREE(h); JUST_FREE(Ew); JUST_FREE(phi); JUST_FREE(T0); NARROW_FREE(res, nx);
}

int EBMO(classic_run)(int nx, int ncol, const double *x, const double *par, double dt, int nsteps,
                      const double *ct_i, const double *ct_ip1, const double *ft, const double *fcol,
                      double *E, double *Tg, double *T, double *h, int nthreads) {
    const size_t N = (size_t)ncol * nx;
    (void)N;
    WIDEN(x, nx); WIDEN(par, P_COUNT); WIDEN(ct_i, nsteps); WIDEN(ct_ip1, nsteps); WIDEN(ft, nsteps); WIDEN(fcol, ncol);
    WIDEN(E, N); WIDEN(Tg, N); WIDEN(T, N); WIDEN(h, N);
    int rc = classic_run_impl(nx, ncol, RARG(x), RARG(par), dt, nsteps, RARG(ct_i), RARG(ct_ip1), RARG(ft),
                              RARG(fcol), RARG(E), RARG(Tg), RARG(T), RARG(h), nthreads);
    JUST_FREE(x); JUST_FREE(par); JUST_FREE(ct_i); JUST_FREE(ct_ip1); JUST_FREE(ft); JUST_FREE(fcol);
    NARROW_FREE(E, N); NARROW_FREE(Tg, N); NARROW_FREE(T, N); NARROW_FREE(h, N);
    return rc;
}

void EBMO(geometry)(int kind, int nx, const double *x, double D, double *out) {
    WIDEN(x, nx); WIDEN(out, (size_t)8 * nx);
    geometry_impl(kind, nx, RARG(x), D, RARG(out));
    JUST_FREE(x); NARROW_FREE(out, (size_t)8 * nx);
}
