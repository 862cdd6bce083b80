// Internal declarations shared by the HIP kernels (ebm_kernels.hip) and the host runtime
// behind the C ABI (ebm_runtime.hip).  Not part of the public interface (include/ebm_hip.h).
#pragma once
#include <hip/hip_runtime.h>

namespace ebm {

// Parameter block, resident in device memory and read through scalar loads.  The first 25
// entries mirror default_parval (reference src/infrastructure.jl:407-433); the derived
// constants are evaluated once on the host in the reference's operation order so that every
// cell sees the same rounded value.
struct Params {
    double D, A, B, cw, S0, S1, S2, a0, a2, ai, Fb, k, Lf, F, cg, tau, Tm, m1, m2, alpha, rl,
        Dmin, Dmax, hmin, kappa;
    double dt;          // st.dt
    // MIZ
    double Tm_pow_m2;   // Tm^m2                                   src/miz.jl:71
    double c_latmelt;   // -pi/2.0*alpha                           src/miz.jl:141
    double c_dn;        // Lf*alpha*Dmin^2*hmin                    src/miz.jl:127
    double c_weld;      // kappa*alpha/4                           src/miz.jl:143
    double c_ht;        // -1/Lf                                   src/miz.jl:139
    double two_rl;      // 2.0*rl                                  src/miz.jl:91
    // classic (get_statics, src/classic.jl:18-29)
    double cg_tau, dt_tau, dc, M, kLf;
    // Refined reciprocals of two constant divisors, produced ON THE DEVICE by the same routine the
    // physics' IEEE division uses (derive_params_kernel), so that dividing through them gives the
    // bits an in-kernel division gives: 1/dt (src/miz.jl:173) and 1/c_dn (src/miz.jl:127).
    double rcp_dt, rcp_cdn;
    double theta_imex;  // dt/cw: the implicit-diffusion extension's matrix is I - theta*Dif (EBM_MODEL_MIZ_IMEX)
};

// Device state: one slab, field slot s at state + s*fstride, each [ncol][pitch] with
// pitch = threads*cells >= nlat (latitude contiguous, padding cells kept at zero).
enum MizSlot { S_Ei = 0, S_Ew, S_h, S_D, S_phi, S_T0, S_Tw, S_Ti, S_n, S_E, S_T, S_MIZ_COUNT };
enum ClassicSlot { C_E = 0, C_Tg, C_T, C_h, C_COUNT };

// The quantities a step produces per cell, in the order the kernels hold them in registers
// (savesol! fusion: StepArgs::var_of maps each one to a saved-variable index or -1).
enum MizQuantity { Q_Ei = 0, Q_Ew, Q_h, Q_D, Q_phi, Q_n, Q_E, Q_T, Q_Ti, Q_Tw, Q_MIZ_COUNT };
enum ClassicQuantity { QC_E = 0, QC_Tg, QC_T, QC_h, QC_COUNT };
constexpr int kMaxQuantities = 12;

// What a step launch writes besides the prognostic state.
enum OutMode {
    OUT_STATE = 0,   // prognostics (+ warm-start mask) only
    OUT_DIAG = 1,    // + T0 and the diagnostic fields
    OUT_SAVE = 2,    // savesol! fused into the step: annual-mean running sums and/or a raw snapshot
                     // from registers; the diagnostic fields only if write_diag
    OUT_LOOP = 3,    // nfused steps in one launch, the whole state on the chip between them (miz_fused_kernel: in registers,
                     // meridians of up to 2048 cells; miz_resident_kernel: in LDS, longer ones and the extension; classic:
                     // registers, any); the diagnostic fields after the last step if write_diag
    OUT_LOOP_SAVE = 4,   // OUT_LOOP with savesol!'s running sums taken from every step (miz_resident_kernel<SAVE>; four cells
                         // per thread; ebm_integrate's stretches without snapshots)
};

// Per-latitude constant tables: one slab, table i at geom + i*gstride (gstride = pitch).
//   G_X              st.x
//   G_0..G_4         physics stencil, bit-exact restatement of the reference:
//                      identity grid: sub, diag, sup of par.D*get_diffop  (infrastructure.jl:480-497)
//                      other grids:   mxxph, mxxmh, diffx[i], diffx[i-1], phmmh  (:509-518)
//   G_LO, G_DI, G_UP the same operator as plain tridiagonal coefficients (T0 / Tg solves only)
//   G_KSUB..G_SB     classic: kappa's three diagonals, aw, S base (src/classic.jl:21-28)
enum GeomTable { G_X = 0, G_0, G_1, G_2, G_3, G_4, G_LO, G_DI, G_UP, G_KSUB, G_KDIAG, G_KSUP, G_AW, G_SB, G_COUNT };

// Per-step scalars of a graph-replayed step: kernel node `slot` of the replayed graph reads entry
// `slot` of a small device table that the host refills before every replay.
struct StepSched {
    double ct, ct_next, ft, tyear;
};

struct StepArgs {
    double *state;
    long long fstride;
    const double *geom;
    long long gstride;
    const double *fcol;              // per-column forcing offset or nullptr
    const double *fsched;            // per-column Forcing schedules [ncol][kSchedWords] or nullptr
    const Params *p;                 // device memory
    unsigned long long *counters;    // 64 shards x {solves, cap hits} (MIZ)
    unsigned short *amask;           // MIZ warm start as an active set: [ncol][threads], bit i <=> T0 < Tm in cell i of the thread
    int pitch, nlat, ncol;
    int col0;                        // first column of this launch (workgroup b steps column col0 + b): launch chains
    double ct, ct_next, ft;          // cos(2 pi t) [MIZ / classic column i], classic column i+1, forcing
    double tyear;                    // model time of the step in years (st.T[tinx]), for the schedules
    const StepSched *sched;          // if non-null, ct/ct_next/ft come from sched[slot] instead (graph replay)
    int slot;
    int write_diag;
    int prefetch;                    // MIZ: L2 prefetch distance in columns (0 = off)
    int nfused;                      // fused launches: steps in this launch, scalars from sched[slot .. slot+nfused)
    // savesol! fused into the step (OUT_SAVE), src/infrastructure.jl:549-591:
    double *sums;                    // annual-mean running sums [nvars][ncol*pitch] in the pair-split layout, or nullptr
    long long sum_stride;            // ncol*pitch
    double *stage;                   // raw snapshots [nvars][chunk][ncol][pitch], or nullptr
    long long stage_var_stride;      // chunk*ncol*pitch
    long long stage_offset;          // snapshot index * ncol*pitch
    signed char var_of[kMaxQuantities];   // quantity -> saved-variable index, -1 = not saved
    unsigned long long *stamps;      // diagnostic builds only (EBM_STAMPS), else nullptr
};

struct LaunchCfg {
    int threads;        // workgroup size (multiple of 64)
    int cells;          // cells per thread (C)
    size_t lds_bytes;   // dynamic LDS per workgroup
    // fused-K launches of the reference's step where both kernels exist (four cells per thread, <= kFusedRegThreads threads):
    // state resident in LDS (128 VGPRs: up to four workgroups per CU fill each other's barrier stalls — launches of many
    // columns) instead of in registers (fewer LDS round trips: a few columns).  Same bits; set by the runtime, not by
    // choose_launch — the GEOMETRY stays a function of (nlat, cells) only.
    bool fused_in_lds;
};

constexpr int kCounterShards = 64;
constexpr int kSchedWords = 9;     // base, peak, cool, rate up, rate down, domain[1..4]
constexpr int kMaxNewton = 1000;   // the cap of the T0 iteration: NonlinearSolve's default maxiters (src/miz.jl:55-60 passes none)

constexpr int kMaxLat = 4096;      // one workgroup of <= 1024 threads x 4 cells owns a whole meridian
constexpr int kFusedRegThreads = 512;   // up to here the fused-K kernel keeps the whole state in registers (4 cells per thread)
constexpr int kFusedRegThreads2 = 768;  // ... with 2 cells per thread (168 VGPRs: three waves per SIMD)
constexpr int kMaxLat2 = 1536;          // longest meridian stepped with 2 cells per thread

// cells_requested: 2 (honoured for nlat <= kMaxLat2) or anything else = 4.  A function of nlat and the request only.
LaunchCfg choose_launch(int nlat, int cells_requested);
hipError_t prepare_kernels(const LaunchCfg &cfg);   // raises the dynamic-LDS limit if needed
// One workgroup per column.  mode: OutMode; OUT_LOOP runs a.nfused steps per launch.
// The per-step MIZ kernels, one function per build part of ebm_kernels.hip (EBM_PART); nullptr = not compiled
using KernelFn = void (*)(const StepArgs);
KernelFn miz_step_kernels_identity(int cells, int mode, int threads);
KernelFn miz_step_kernels_nonuniform(int cells, int mode, int threads);
KernelFn miz_step_kernels_imex(int grid_kind, int mode, int threads);
// fused-K with the state resident in LDS (miz_resident_kernel): four cells per thread; the reference's step beyond
// kFusedRegThreads threads, the extension at every size
KernelFn miz_resident_kernels(int grid_kind, int threads, bool imex);
KernelFn miz_resident_save_kernels(int grid_kind, int threads, bool imex);   // ... with savesol!'s sums, every size
KernelFn miz_fused2_save_kernels(int grid_kind, int threads);                // two cells per thread: miz_fused_kernel<2, ..., SAVE>

bool has_miz_kernel(const LaunchCfg &cfg, int grid_kind, int mode, bool imex);   // is this (geometry, mode) compiled?
// `count` workgroups, stepping columns first ... first + count - 1
hipError_t launch_miz_step(const StepArgs &a, int grid_kind, int mode, const LaunchCfg &cfg, bool imex, int first, int count,
                           hipStream_t s);
hipError_t launch_classic_step(const StepArgs &a, int mode, const LaunchCfg &cfg, int first, int count, hipStream_t s);
// rcp_dt / rcp_cdn of the device-resident parameter block (see Params)
hipError_t launch_derive_params(Params *p_dev, hipStream_t s);
// active set from the T0 field (after ebm_set_field(T0))
hipError_t launch_mask_from_t0(const StepArgs &a, int ncol, const LaunchCfg &cfg, hipStream_t s);
hipError_t launch_divide(const double *a, const double *b, double *q, int n, hipStream_t s);
// out[col] = hemispheric_mean(field[col], x), src/utilities.jl:397-403 (sequential sum, bit-exact)
hipError_t launch_hemispheric_mean(const double *field, const double *x, int pitch, int nlat, int ncol, double *out,
                                   hipStream_t s);
// out = base + D d/dx[(1-x^2) d temp/dx] per column ([ncol][pitch] device arrays; base may be null)
hipError_t launch_diffusion(const double *temp, const double *base, double *out, const double *geom, long long gstride,
                            const Params *p, int grid_kind, int pitch, int nlat, int ncol, hipStream_t s);
// annual_mean (src/infrastructure.jl:536-544): dst[v][col][k] = sum[v]/nt with `sum` in the pair-split
// layout of the step kernels, then sum = 0; all nvars variables (var_stride apart) in one launch
hipError_t launch_finish_mean(double *dst, double *sum, double nt, int ncol, int nvars, long long var_stride,
                              const LaunchCfg &cfg, hipStream_t s);
// Zonal diffusion substep (ebm_zonal_diffusion): per member and latitude the periodic tridiagonal solve along the circle.
// Everything lives in the handle's store index space p (4 cells per thread: pair-split, p = j*2T + 2t + q <-> latitude
// k = 4t + 2j + q; 2 cells per thread: p = k), so that all accesses are contiguous: T, out_Z, out_U [ncol][pitch];
// zM, zE [nlon][pitch]; za, zW [pitch].  out_Z is also the scratch of the forward sweep; out_U may be null.
hipError_t launch_zonal_sweep(const double *T, double *out_Z, double *out_U, const double *zM, const double *zE,
                              const double *za, const double *zW, int nlon, int nmember, int pitch, double rtheta,
                              hipStream_t s);
// The same systems partitioned along the circle into S segments (circles of >= 256 longitudes): chain tables cM, cE
// [nlon/S - 1][pitch] of one segment, reduced-system tables rM, rE [S - 1][pitch], za, za2, rW [pitch]; su, sg, sy are
// scratch [nmember][S][pitch].  Three launches.
hipError_t launch_zonal_sweep_segmented(const double *T, double *out_Z, double *out_U, const double *cM, const double *cE,
                                        const double *rM, const double *rE, const double *za, const double *za2,
                                        const double *rW, double *su, double *sg, double *sy, int nlon, int S, int nmember,
                                        int pitch, double rtheta, hipStream_t s);
// natural <-> pair-split layout of whole fields ([ncol][pitch], 4 cells per thread; a no-op with 2), in place
hipError_t launch_split_fields(double *fields, long long field_stride, int nfields, int ncol, const LaunchCfg &cfg,
                               hipStream_t s);
// the diagnostic fields of a 4-cells-per-thread step launch, pair-split -> natural layout, in place
hipError_t launch_unsplit_fields(double *fields, long long field_stride, int nfields, int ncol, const LaunchCfg &cfg,
                                 hipStream_t s);

}  // namespace ebm
