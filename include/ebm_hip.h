/* ebm_hip.h — C ABI of the MI355X-native energy-balance time-stepping path.
 *
 * This is the drop-in boundary for the hot path of waylonwh/EnergyBalanceModel.jl:
 *
 *   Infrastructure.step!(::Val{:MIZ}|::Val{:Classic}, t, f, vars, st, par; debug, verbose)
 *       stub   src/infrastructure.jl:594
 *       MIZ    src/miz.jl:150-196        classic  src/classic.jl:37-71
 *   Infrastructure.integrate(model, st, forcing, par, init; lastonly, debug, verbose)
 *       src/infrastructure.jl:615-636 (+ savesol! :549-591, annual_mean :536-544)
 *
 * The reference is pure Julia and has no FFI of its own; the entry points below are what a
 * Julia `ccall` shim binds (julia/EBMHip.jl).  The shim's two supported calls keep
 * the reference's signatures and model symbols, as functions of the shim module:
 *     EBMHip.integrate(:MIZ | :Classic, st, forcing, par, init; lastonly, verbose)   -> ebm_integrate
 *     EBMHip.step!(Val(:MIZ | :Classic), t, f, vars, st, par; verbose)               -> ebm_step
 * (no new model tag: the reference decides on the symbol's value what a run stores and which
 * parameters it gets, src/infrastructure.jl:621-624, :473-474).  Plain C: opaque handle,
 * `double*`/`int` only, no C++/torch types.  All arrays are fp64, latitude contiguous:
 * a field is `[ncol][nlat]` in C order == Julia `Array{Float64,2}(nlat, ncol)`; a column is
 * one independent meridian (a longitude of a 2-D grid and/or an ensemble member).
 *
 * Threading: all state — including the T0 warm start the reference hides in a module-level
 * closure (src/miz.jl:47,64) — is owned by the handle.  Different handles may be used from
 * different host threads; one handle must not be used concurrently.
 *
 * Errors: every function returns 0 on success, a negative ebm_status otherwise; the message
 * is available (per host thread) from ebm_last_error().  Numerical events are not errors:
 * NaN sentinels are data (src/miz.jl:193-194) and a T0 iteration that hits its cap is only
 * counted (the reference merely warns, src/miz.jl:61-63) — see ebm_get_counters.
 */
#ifndef EBM_HIP_H
#define EBM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ebm_ctx *ebm_handle_t;

enum ebm_status {
    EBM_OK = 0,
    EBM_ERR_ARG = -1,      /* bad argument */
    EBM_ERR_HIP = -2,      /* HIP runtime error */
    EBM_ERR_UNSUPPORTED = -3,
    EBM_ERR_NO_DEVICE = -4, /* no usable GPU: the library never falls back to the CPU */
    EBM_ERR_STALE = -5      /* the requested field is older than the state (see "Validity" below) */
};

/* model tag == the Val{...} the reference dispatches step! on (src/infrastructure.jl:594).
 *
 * EBM_MODEL_MIZ_IMEX is an EXTENSION with no counterpart in the reference (and therefore no parity to
 * claim): the MIZ model with its meridional diffusion treated linearly implicitly.  Each step the explicit
 * increment of every cell's total enthalpy, dE = dt*(phi*Fvi + (1-phi)*Fvw), goes through one tridiagonal
 * solve per meridian, (I - (dt/cw)*Dif) dE_new = dE, and the diffusion term of both vertical fluxes
 * (src/miz.jl:96-101) is corrected by (dE_new - dE)/dt; everything else is the reference's step.  This lifts
 * the explicit limit dt <= cw*dx^2/(2D) (more than 800,000 steps per year at 4096 latitudes) and converges
 * to the reference's scheme as dt -> 0.  Dif is the operator D d/dx[(1-x^2) d/dx] as a plain tridiagonal
 * matrix (zero-flux ends); assuming that the surface temperature follows the increment with the water's
 * heat capacity cw is an upper bound of the true response (heat that melts or grows ice changes no
 * temperature), which is what makes the scheme stable.  THIS TEXT IS THE DEFINITION.
 * Same fields, parameters and entry points as EBM_MODEL_MIZ, ebm_run_fused included. */
enum ebm_model { EBM_MODEL_MIZ = 0, EBM_MODEL_CLASSIC = 1, EBM_MODEL_MIZ_IMEX = 2 };

/* SpaceTime{identity} uses the sparse uniform-x operator (src/infrastructure.jl:495-497);
 * every other SpaceTime{F} (e.g. sin) the flux-form stencil (:505-526). */
enum ebm_grid { EBM_GRID_IDENTITY = 0, EBM_GRID_NONUNIFORM = 1 };

/* order of the 25 entries of default_parval (src/infrastructure.jl:407-433) */
enum ebm_param {
    EBM_P_D = 0, EBM_P_A, EBM_P_B, EBM_P_cw, EBM_P_S0, EBM_P_S1, EBM_P_S2, EBM_P_a0, EBM_P_a2,
    EBM_P_ai, EBM_P_Fb, EBM_P_k, EBM_P_Lf, EBM_P_F, EBM_P_cg, EBM_P_tau, EBM_P_Tm, EBM_P_m1,
    EBM_P_m2, EBM_P_alpha, EBM_P_rl, EBM_P_Dmin, EBM_P_Dmax, EBM_P_hmin, EBM_P_kappa,
    EBM_P_COUNT
};

/* fields of `vars` (src/infrastructure.jl:604-605, 621-624) plus the hidden warm start */
enum ebm_field {
    EBM_F_Ei = 0, EBM_F_Ew, EBM_F_h, EBM_F_D, EBM_F_phi, /* MIZ prognostics (init)          */
    EBM_F_T0,                                             /* MIZ warm start (src/miz.jl:47), see below */
    EBM_F_Tw, EBM_F_Ti, EBM_F_n, EBM_F_E, EBM_F_T,        /* MIZ diagnostics; E,T also classic */
    EBM_F_Tg,                                             /* classic ghost layer             */
    EBM_F_COUNT
};
/* classic uses EBM_F_E, EBM_F_Tg (prognostic) and EBM_F_T, EBM_F_h (diagnostic).
 *
 * EBM_F_T0: the reference keeps the previous T0 solution as the starting iterate of its nonlinear
 * solve.  The active-set iteration used here depends on that vector only through its sign pattern
 * [T0 < Tm], so between steps the library carries that pattern (one bit per cell) and writes the
 * fp64 EBM_F_T0 field together with the diagnostics (write_diag / diag_last != 0).
 * ebm_set_field(EBM_F_T0, ...) rebuilds the pattern from the given values. */

/* ---- lifetime ------------------------------------------------------------------------ */

/* Create a stepping context on HIP device `device` for `ncol` independent meridians of
 * `nlat` cells (2 <= nlat <= 4096: one workgroup owns a whole meridian; longer ones are refused with
 * EBM_ERR_UNSUPPORTED).  x[nlat] is st.x, params[EBM_P_COUNT] the parameter values (entries a model
 * does not use are ignored), dt = st.dt.  All state starts at zero (as the reference's T0
 * warm start does).  Fails with EBM_ERR_NO_DEVICE when no GPU is present. */
int ebm_create(ebm_handle_t *out, int model, int grid, int nlat, int ncol, const double *x,
               const double *params, double dt, int device);

/* Launch options.  The library reads NO environment variable: whatever changes how a handle runs is passed
 * here.  Zero-initialise (or call ebm_options_default) and set struct_bytes = sizeof(ebm_options); fields the
 * caller's struct does not have keep their defaults, so the struct can grow.
 *
 * cells_per_thread: latitudes per thread of the one workgroup that owns a meridian — 4 (default: 32
 *   contiguous bytes per lane and field, the throughput geometry) or 2 (nlat <= 1536: twice as many waves
 *   per meridian, for latency-bound runs of a FEW short meridians, e.g. one 180-band column).  The
 *   tridiagonal partition, hence the ROUNDING of the T0 / Tg solves, depends on this number and on nothing
 *   else a caller controls: the geometry is a function of (nlat, cells_per_thread) only — never of ncol — so
 *   a member gives the same bits alone, inside a large ensemble, and under any sharding over GPUs, as long as
 *   every rank passes the same options.
 * use_graph: replay hipGraphs of 64 captured step launches in ebm_run (-1 = by size: on for steps of at most
 *   262,144 cells, which are launch-bound; 0 = off; 1 = on).  Bit-identical either way.
 * prefetch_cols: L2-prefetch distance of the MIZ step kernel in columns (-1 = the successor workgroup on the
 *   same XCD when at most two workgroups fit a CU, else off; 0 = off).  Performance only.
 * launch_chains: 1 (and -1, the default) = every step is one launch over all columns; 2 = the two halves of the columns
 *   are stepped by two independent chains of launches on two streams, which fill each other's launch boundaries and store
 *   tails: worth it where a CU holds a single workgroup (meridians of more than 2048 cells) and every chain still fills
 *   the chip several times over (4096 x 2048: 0.1656 -> 0.1594 ms per step).  Columns are independent: bit-identical either
 *   way.  ebm_get_counters counts one launch per chain and step; ignored with graph replay.
 * fused_state_in_lds: where the fused-K launches (ebm_run_fused) of the reference's step keep the state when both of its
 *   kernels exist (four cells per thread, meridians of up to 2048 cells): 0 = in registers (fewest LDS round trips: the
 *   fastest for a few columns), 1 = in LDS (128 registers per lane, so that up to four workgroups share a CU and fill each
 *   other's barrier stalls: 1.15 ... 1.45 x the throughput on launches of many columns), -1 (default) = in LDS when the
 *   handle has more columns than the register kernel runs in one round (one 256-thread workgroup per compute unit, four of
 *   64 threads).  Performance only: the two kernels compute the same bits.
 * integrate_steps_per_launch: how ebm_integrate / ebm_integrate_hemispheric step through the stretches of a year that need
 *   nothing but the annual-mean sums (no raw snapshot, no seasonal snapshot, not a year's last step): -1 (default) and
 *   values > 1 = that many steps fused into one launch with the state resident on the chip and the sums taken from
 *   every step (default 64); 1 = one launch per step everywhere.  Bit-identical either way.  MIZ and MIZ_IMEX (not two
 *   cells per thread on meridians of more than 1024 cells); other handles always step one launch at a time. */
typedef struct ebm_options {
    int struct_bytes;
    int cells_per_thread;
    int use_graph;
    int prefetch_cols;
    int launch_chains;
    int integrate_steps_per_launch;
    int fused_state_in_lds;
} ebm_options;
int ebm_options_default(ebm_options *opt);
/* ebm_create with explicit options (opt == NULL: the defaults, i.e. exactly ebm_create). */
int ebm_create_ex(ebm_handle_t *out, int model, int grid, int nlat, int ncol, const double *x,
                  const double *params, double dt, int device, const ebm_options *opt);
int ebm_destroy(ebm_handle_t h);
const char *ebm_last_error(void);
const char *ebm_version(void);

/* ---- state --------------------------------------------------------------------------- */

/* Copy a whole field host<->device ([ncol][nlat] doubles, synchronous).  The copies run through a pinned
 * staging ring owned by the handle (device -> pinned by DMA while the previous piece is copied on to the
 * caller's pageable buffer by a few host threads).
 *
 * Validity.  The prognostic fields are always current.  The diagnostic fields (MIZ: Tw, Ti, n, E, T; classic:
 * T, h) and the fp64 warm start EBM_F_T0 are written only by steps that were asked to (ebm_step with
 * write_diag, the last step of ebm_run / ebm_run_fused with diag_last, the seasonal and last steps of
 * ebm_integrate).  A read of one of them — ebm_get_field, ebm_get_field_device, ebm_hemispheric_mean*,
 * ebm_field_device_ptr — while the state is NEWER than the field (steps taken since without diagnostics, or
 * a prognostic field overwritten with ebm_set_field) fails with EBM_ERR_STALE; the message names the step
 * that last wrote the field and the state's step.  Nothing stale is ever returned silently — in particular
 * not the T0 a caller would checkpoint as the warm start (src/miz.jl:47,64).  ebm_field_step reports both
 * steps; ebm_get_field_as_of returns the field as of an EXPLICITLY named step (the 0-based global index of
 * the step that wrote it) however far the state has moved on since, and fails with EBM_ERR_STALE if that is
 * not the step that wrote it.  ebm_set_field(EBM_F_T0) makes T0 current; setting a diagnostic field directly
 * makes that field current as well (it is the caller's statement of what it holds). */
int ebm_set_field(ebm_handle_t h, int field, const double *host);
int ebm_get_field(ebm_handle_t h, int field, double *host);
/* *written_step: 0-based global index of the step that last wrote the field (-1: never written; for a field
 * set by the caller: the index of the last step taken before that, -1 if none); *state_step: the same for
 * the prognostic state.  The field is current iff the two are equal and no prognostic field has been
 * overwritten since (*current != 0).  Any output pointer may be NULL. */
int ebm_field_step(ebm_handle_t h, int field, long long *written_step, long long *state_step, int *current);
int ebm_get_field_as_of(ebm_handle_t h, int field, long long step, double *host);
/* hemispheric_mean (src/utilities.jl:397-403) of a field, per column, reduced on the device in the
 * reference's summation order (bit-identical): out[ncol] on the host.  Ensemble diagnostics are
 * O(columns) instead of O(state).  Synchronous. */
int ebm_hemispheric_mean(ebm_handle_t h, int field, double *out);
/* Device-to-device variants for callers that keep their own device buffers (e.g. the payload of an
 * RCCL gather): dev_out[ncol] / dev_out[ncol][nlat] packed, on the handle's device.  Synchronous. */
int ebm_hemispheric_mean_device(ebm_handle_t h, int field, double *dev_out);
int ebm_get_field_device(ebm_handle_t h, int field, double *dev_out);
/* The meridional diffusion operator on its own — diffusion!(base, temp, st, par) / diffusion(T, st, par)
 * = D∇², src/infrastructure.jl:495-533: out = base + D d/dx[(1-x^2) d temp/dx] per column, with the
 * handle's grid kind (identity: the CSC product of par.D*get_diffop, :495-497; any other grid: the flux
 * form, :505-526) and the very device functions the step kernels fuse — bit for bit the reference's
 * operation order.  temp, base (NULL = zeros), out: [ncol][nlat] host arrays.  MIZ handles only.
 * Synchronous. */
int ebm_diffusion(ebm_handle_t h, const double *temp, const double *base, double *out);
/* The ZONAL partner of the meridional operator above, as an implicit substep — an EXTENSION with no counterpart in the
 * reference (the reference has no longitude axis; the nearest text is the meridional operator,
 * src/infrastructure.jl:505-526), hence no parity to claim.  The columns of the handle are read as nmember = ncol / nlon
 * latitude-longitude grids of nlon equally spaced longitudes (column = member*nlon + longitude, periodic).  The zonal part
 * of the spherical diffusion operator, D/(1-x^2) d^2/dlambda^2, cannot be taken explicitly near the pole (1-x^2 = 6e-7 at
 * the last of 1024 latitudes); this is its backward-Euler step over the handle's dt on a field with the water's heat
 * capacity cw: for every member and latitude k solve the periodic tridiagonal system along the latitude circle
 *     (1 + 2 a_k) U_l - a_k (U_{l-1} + U_{l+1}) = temp_l,    a_k = (dt/cw) D / ((1 - x_k)(1 + x_k) dlambda^2),  dlambda = 2 pi/nlon
 * (1 - x^2 is evaluated as (1 - x)(1 + x), which does not cancel near the pole) and form the zonal heat-flux convergence
 *     Z_l = (U_l - temp_l) cw/dt      ( = D/((1-x_k^2) dlambda^2) (U_{l-1} - 2 U_l + U_{l+1}) ).
 * THIS TEXT IS THE DEFINITION.  The solve is free arithmetic (its result is defined by the linear system, like T0's): one
 * lane per (member, latitude) walks the longitudes, so every access is along the contiguous latitude axis; circles of 256
 * longitudes and more are cut into 4 ... 32 segments (a function of nlon only) with a reduced periodic system of their ends.
 * temp, out_U, out_Z: [ncol][nlat] host arrays (either output may be NULL); nlon >= 3 must divide ncol.  MIZ-family
 * handles.  Synchronous.
 * Why this is an operator and not a model: coupling it to the column step by operator splitting (Z of the previous step's
 * output temperature added to the diffusion term of both vertical fluxes) is stable on open water — there it converges
 * to the decay of the spherical harmonics P_l^m(x) cos(m lambda) at second order — but not over thin new ice, whose surface
 * temperature answers an enthalpy change ~60 times more strongly than water's does and without delay: zonal differences
 * then grow ~50-fold per step until the ice has thickened (measured with the checker's restatement,
 * tests/test_oracle_zonal.py).  A stable coupling has to put the zonal term inside the T0 balance of src/miz.jl:33-45 — a
 * two-dimensional elliptic solve per step — which this library does not contain. */
int ebm_zonal_diffusion(ebm_handle_t h, int nlon, const double *temp, double *out_U, double *out_Z);
/* Device pointer of a field and its row pitch in elements (>= nlat), for zero-copy users
 * (e.g. a torch tensor view).  The pointer stays valid until ebm_destroy.  For the MIZ diagnostic fields the view
 * shows the field as of this call: steps that write them afterwards store them in a layout private to the library
 * until the next read through this interface (call again after such a step).  Fails with EBM_ERR_STALE like
 * ebm_get_field. */
int ebm_field_device_ptr(ebm_handle_t h, int field, double **dptr, long long *pitch);
/* Per-column forcing offset added to the per-step scalar forcing (forcing = f + fcol[col];
 * NULL clears it).  This is how ensemble members / longitudes get perturbed forcings — the
 * reference has a single scalar `f` per step (src/infrastructure.jl:631). */
int ebm_set_column_forcing(ebm_handle_t h, const double *fcol);
/* Per-column forcing SCHEDULES: column c is additionally forced by its own Forcing{false}
 * (src/infrastructure.jl:208-241), evaluated on the device at the model time T of every step as
 * the reference's call operator does (:294-307):
 *     T < d1: base;  T < d2: base + up*(T - d1);  T < d3: peak;  T < d4: peak + down*(T - d3);  else cool
 * sched[ncol][9] = {base, peak, cool, up, down, d1, d2, d3, d4} (d = Forcing.domain[2..5]); NULL
 * clears.  T of 0-based global step n is st.T[n+1] = (2n+1)/(2 nt): ebm_run takes n from
 * first_step, ebm_step and ebm_integrate use and advance the handle's step clock (0 after ebm_create, set by
 * ebm_set_step_clock, left at the step after the last one by every stepping call — so a run chunked into several
 * ebm_integrate calls of whole years sees the same model time as one call).  Needs the time table (its length is nt).  The three contributions add:
 * forcing = f + fcol[c] + schedule_c(T). */
int ebm_set_column_schedule(ebm_handle_t h, const double *sched);
int ebm_set_step_clock(ebm_handle_t h, long long step);
/* Table of cos(2.0*pi*st.t[i]), i = 1..nt (src/miz.jl:11, src/classic.jl:24), needed by
 * ebm_run/ebm_integrate.  Computed by the caller so that host and device agree bit for bit. */
int ebm_set_time_table(ebm_handle_t h, int nt, const double *cos2pit);

/* ---- stepping ------------------------------------------------------------------------ */

/* One step!: cos2pit = cos(2.0*pi*t) for this step; cos2pit_next is column i+1 of the
 * classic model's S table (src/classic.jl:25,61; ignored for MIZ); f the scalar forcing.
 * write_diag != 0 also writes the diagnostic fields (Tw,Ti,n,E,T / T,h).  Asynchronous on
 * the handle's stream. */
int ebm_step(ebm_handle_t h, double cos2pit, double cos2pit_next, double f, int write_diag);

/* nsteps consecutive steps, one kernel launch per step (K = 1), starting at 0-based global
 * step index `first_step` (time-of-year index = first_step mod nt into the time table).
 * f_steps[nsteps] are the per-step scalar forcings (NULL = 0.0).  Diagnostics are written
 * on the last step only when diag_last != 0.  Asynchronous. */
int ebm_run(ebm_handle_t h, long long first_step, int nsteps, const double *f_steps,
            int diag_last);

/* The same nsteps steps with `steps_per_launch` (K) consecutive steps fused into one kernel launch:
 * the time loop of integrate (src/infrastructure.jl:630-634) for callers that need no per-step
 * output.  Between the steps of a launch the whole state stays on the chip — in registers for
 * meridians of up to 2048 cells, in LDS for longer ones and for EBM_MODEL_MIZ_IMEX; the per-step
 * scalars come from a device table.  Results are bit-identical to ebm_run for every model and size
 * (ebm_get_counters reports the launches actually made: ceil(nsteps / K), twice that with two launch
 * chains).  Asynchronous. */
int ebm_run_fused(ebm_handle_t h, long long first_step, int nsteps, const double *f_steps,
                  int diag_last, int steps_per_launch);

/* integrate + savesol! (src/infrastructure.jl:549-591, 615-636) with state resident on the
 * device: runs nt*dur steps from the current state.  `fields[nvars]` selects the saved
 * variables; outputs are host buffers (any may be NULL to skip):
 *   raw    [nvars][nraw][ncol][nlat]   nraw = lastonly ? nt : nt*dur
 *   winter, summer, avg  [nvars][dur][ncol][nlat]
 * winter_inx/summer_inx are the 1-based in-year indices st.winter.inx / st.summer.inx.
 * f_steps[nt*dur] as in ebm_run.  `fields` must be solution variables (not the hidden EBM_F_T0),
 * each at most once.  savesol! runs inside the step kernel: the annual-mean sums and the raw snapshot are taken
 * from the step's registers — one launch per step on the steps whose snapshot leaves the device (raw, winter,
 * summer, a year's last step), ebm_options.integrate_steps_per_launch steps per launch in between, with the
 * same bits.  The annual mean is sum / nt with the sum
 * taken per cell sequentially in step order; the reference's crossmean (src/utilities.jl:390-395) is
 * Statistics.mean over the year's snapshots, i.e. Julia's pairwise, SIMD-reassociated sum — the two agree up to
 * summation-order rounding (the parity tests hold avg to 1e-8 of the oracle's), not bit for bit.
 * The time-of-year index starts at 1 (the call begins a year); model time (per-column schedules) continues from
 * the handle's step clock.  Host output overlaps the stepping (device-to-device snapshot, then DMA through the
 * handle's pinned ring on a stream of its own).  Synchronous: returns when all outputs are in the caller's arrays. */
int ebm_integrate(ebm_handle_t h, int nt, int dur, const double *f_steps, int lastonly,
                  int winter_inx, int summer_inx, int nvars, const int *fields, double *raw,
                  double *winter, double *summer, double *avg);

/* The same integration when only the HEMISPHERIC MEANS of the seasonal outputs are wanted — the numbers
 * behind the reference's hysteresis plot (plot_seasonal, src/plot.jl:173-225: hemispheric_mean of
 * seasonal.avg.T[year] and of seasonal.{avg,winter,summer}.phi[year]): per saved variable, year and column,
 * hemispheric_mean (src/utilities.jl:397-403, the reference's summation order) of the winter snapshot, the
 * summer snapshot and the annual mean, reduced on the device.  Outputs are [nvars][dur][ncol] host arrays
 * (any may be NULL): O(columns x years) bytes cross the bus instead of O(state x years) — ensembles.
 * Bit-identical to applying hemispheric_mean to ebm_integrate's winter / summer / avg outputs. */
int ebm_integrate_hemispheric(ebm_handle_t h, int nt, int dur, const double *f_steps, int winter_inx,
                              int summer_inx, int nvars, const int *fields, double *hm_winter,
                              double *hm_summer, double *hm_avg);

int ebm_sync(ebm_handle_t h);

/* ---- measurement / diagnostics ------------------------------------------------------- */

/* counters[0] steps, [1] tridiagonal solves summed over columns, [2] column-steps whose T0
 * active-set iteration hit its cap, [3] kernel launches.  Synchronises the stream. */
int ebm_get_counters(ebm_handle_t h, long long *counters);
int ebm_reset_counters(ebm_handle_t h);
/* HIP-event timing on the handle's stream (the stream the kernels are launched on). */
int ebm_timer_start(ebm_handle_t h);
int ebm_timer_stop(ebm_handle_t h, float *elapsed_ms);
/* Launch geometry chosen for this handle: info[0] threads per workgroup, [1] cells per
 * thread, [2] dynamic LDS bytes per workgroup, [3] workgroups per launch. */
int ebm_launch_info(ebm_handle_t h, int *info);
/* Self-test: q[i] = a[i] / b[i] computed on the device with the division routine the physics
 * kernels use (bit-exact IEEE fp64 division is part of the parity contract). */
int ebm_selftest_divide(int device, int n, const double *a, const double *b, double *q);

#ifdef __cplusplus
}
#endif
#endif /* EBM_HIP_H */
