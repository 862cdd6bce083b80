/* A plain-C caller of the C ABI (include/ebm_hip.h) — what any FFI (Julia's ccall, cgo, JNI ...) does.
 *
 *   gcc -std=c99 -Wall -Iinclude examples/c_abi_example.c -o c_abi_example \
 *       -Lenergybalancemodel.jl_amd -lebm_hip -lm -Wl,-rpath,$PWD/energybalancemodel.jl_amd
 *   ./c_abi_example                       (needs an MI355X; without one ebm_create fails loudly)
 *
 * The reference test's configuration (test/runtests.jl:22-32): SpaceTime{sin}(180, 2000, 1), Forcing(0.0),
 * default_parameters(:MIZ), zero state; ten steps (the index the reference test compares, :40-41), the
 * first five as single `step!`s (ebm_step), the rest through the resident loop (ebm_run); prints T.
 */
#define _USE_MATH_DEFINES
#define _DEFAULT_SOURCE          /* M_PI under -std=c99 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ebm_hip.h"

#define NX 180
#define NT 2000

static void die(const char *what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, ebm_last_error());
    exit(1);
}

int main(void) {
    /* default_parval, src/infrastructure.jl:407-433, in enum ebm_param order */
    double par[EBM_P_COUNT];
    par[EBM_P_D] = 0.6;  par[EBM_P_A] = 193.0;  par[EBM_P_B] = 2.1;  par[EBM_P_cw] = 9.8;
    par[EBM_P_S0] = 420.0;  par[EBM_P_S1] = 338.0;  par[EBM_P_S2] = 240.0;  par[EBM_P_a0] = 0.7;
    par[EBM_P_a2] = 0.1;  par[EBM_P_ai] = 0.4;  par[EBM_P_Fb] = 4.0;  par[EBM_P_k] = 2.0;
    par[EBM_P_Lf] = 9.5;  par[EBM_P_F] = 0.0;  par[EBM_P_cg] = 0.01 * 9.8;  par[EBM_P_tau] = 1e-5;
    par[EBM_P_Tm] = 0.0;  par[EBM_P_m1] = 1.6e-6 * 31536000;  par[EBM_P_m2] = 1.36;
    par[EBM_P_alpha] = 0.66;  par[EBM_P_rl] = 0.5;  par[EBM_P_Dmin] = 1.0;  par[EBM_P_Dmax] = 156.0;
    par[EBM_P_hmin] = 0.1;  par[EBM_P_kappa] = 0.01 * 31536000;

    /* SpaceTime{sin}(180, 2000, 1): x = sin(u), u the midpoints of 180 equal parts of (0, pi/2) */
    double x[NX], ctab[NT];
    const double du = (M_PI / 2.0) / NX;
    for (int k = 0; k < NX; ++k) x[k] = sin(du / 2.0 + k * du);
    for (int i = 0; i < NT; ++i) ctab[i] = cos(2.0 * M_PI * ((2.0 * i + 1.0) / (2.0 * NT)));

    ebm_handle_t h = NULL;
    int rc = ebm_create(&h, EBM_MODEL_MIZ, EBM_GRID_NONUNIFORM, NX, 1, x, par, 1.0 / NT, 0);
    if (rc) die("ebm_create", rc);
    printf("%s\n", ebm_version());
    for (int i = 0; i < 5; ++i)                                  /* step!(Val(:MIZ), t_i, 0.0, vars, st, par) */
        if ((rc = ebm_step(h, ctab[i], 0.0, 0.0, 1))) die("ebm_step", rc);
    if ((rc = ebm_set_time_table(h, NT, ctab))) die("ebm_set_time_table", rc);
    if ((rc = ebm_run(h, 5, 5, NULL, 1))) die("ebm_run", rc);       /* steps 6..10, state resident */
    double T[NX], phi[NX];
    if ((rc = ebm_get_field(h, EBM_F_T, T))) die("ebm_get_field(T)", rc);
    if ((rc = ebm_get_field(h, EBM_F_phi, phi))) die("ebm_get_field(phi)", rc);
    long long cnt[4];
    if ((rc = ebm_get_counters(h, cnt))) die("ebm_get_counters", rc);
    printf("steps %lld solves %lld cap_hits %lld launches %lld\n", cnt[0], cnt[1], cnt[2], cnt[3]);
    for (int k = 0; k < NX; ++k) printf("T[%d] %.17g phi %.17g\n", k, T[k], phi[k]);
    ebm_destroy(h);
    return 0;
}
