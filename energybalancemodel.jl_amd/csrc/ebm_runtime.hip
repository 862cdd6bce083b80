// Host runtime behind the C ABI of include/ebm_hip.h: handle/state ownership, per-latitude
// constant tables, the step / run / integrate drivers and HIP-event timing.  Device work is in
// ebm_kernels.hip.  There is deliberately no CPU fallback: without a GPU every entry point
// fails with EBM_ERR_NO_DEVICE.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ebm_hip.h"
#include "ebm_hostcopy.h"
#include "ebm_internal.h"

using ebm_host::CopyJob;
using ebm_host::HostCopier;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    // a failed runtime call leaves its code as the thread's "last error", which the next kernel launch's
    // hipGetLastError() check would report as its own: the failure has been reported here, so clear it
    if (code == EBM_ERR_HIP) (void)hipGetLastError();
    return code;
}
#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(EBM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

}  // namespace

struct ebm_ctx {
    int model = 0, grid = 0, nlat = 0, ncol = 0, device = 0;
    bool imex = false;                             // EBM_MODEL_MIZ_IMEX: model == EBM_MODEL_MIZ plus the implicit-diffusion extension
    long long pitch = 0;
    double dt = 0.0;
    ebm::Params p{};
    ebm::LaunchCfg cfg{};
    ebm::Params *p_dev = nullptr;                  // parameter block in device memory
    double *geom = nullptr;                        // per-latitude tables, G_COUNT x gstride
    long long gstride = 0;
    double *state = nullptr;                       // field slab, nslots x fstride
    long long fstride = 0;
    int nslots = 0;
    double *field[EBM_F_COUNT] = {nullptr};        // views into the slab (null: not in this model)
    double *fcol = nullptr;
    double *fsched = nullptr;                      // per-column Forcing schedules
    long long clock = 0;                           // global index of the next step (model time of ebm_step)
    unsigned long long *stamps = nullptr;          // diagnostic builds only
    int num_cus = 0;
    int prefetch = 0;                 // L2 prefetch distance of the MIZ kernel, columns (0 = off)
    double *hm_dev = nullptr;         // ebm_hemispheric_mean: per-column results on the device
    // per-step scalars of the fused-K launches: two device tables of kFusedTable entries used in turn, each with the event
    // that marks the end of the launches that read it — a table is refilled only after that event, so consecutive fused
    // calls neither wait for each other nor synchronise the stream
    struct SchedTable { ebm::StepSched *dev = nullptr; hipEvent_t done = nullptr; bool in_use = false; } sched_tab[2];
    int sched_next = 0;
    int integrate_spl = 64;                  // ebm_options::integrate_steps_per_launch (1 = one launch per step)
    // hipGraph replay for launch-bound shapes (small grids): kGraphSteps step kernels per replay
    ebm::StepSched *sched_dev = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool use_graph = false;
    std::vector<double> ttab;                      // cos(2*pi*t_i), host copy
    unsigned long long *counters = nullptr;        // device, kCounterShards x 2
    unsigned short *amask = nullptr;               // MIZ warm-start active set, ncol x threads
    long long n_steps = 0, n_launches = 0;
    hipStream_t stream = nullptr;                  // THE stream of the handle: everything is ordered on it (see main_stream)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Two chains of step launches.  Long meridians leave room for ONE workgroup per CU, so within a launch nothing runs under
    // a workgroup's load, solve and store phases, and a launch cannot start before the slowest workgroup of the previous one
    // has ended.  Columns are independent: the first half of them is stepped on `stream`, the second on `stream2`, each half
    // its own chain of launches; the chains drift apart and fill each other's gaps (measured on 4096 x 2048: 0.1656 ->
    // 0.1594 ms per step, tests/tools/ab_two_handles.py).  `forked` = the chains are running apart; any other use of the
    // handle's stream joins them first (main_stream).
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int split_col = 0;                             // 0: one chain; else the first column of the second chain
    bool forked = false;
    // Validity of the fields that only some steps write (diagnostics, the fp64 T0): `epoch` counts every change
    // of the prognostic state (steps taken, prognostic fields overwritten), `state_step` is the global index of
    // the last step taken (-1: none); a field is current iff written_epoch[f] == epoch.
    long long epoch = 0, state_step = -1;
    long long written_epoch[EBM_F_COUNT], written_step[EBM_F_COUNT];
    // The MIZ step kernels (4 cells per thread) store the five diagnostic fields in the pair-split layout (whole
    // 128-B lines per store instruction, csrc/ebm_kernels.hip); whoever reads one of them gets the natural layout:
    // the first reader after such a step runs the in-place un-permutation once.
    bool diag_split = false;
    // ebm_zonal_diffusion: the tables of the last nlon used, kept between calls
    int nlon = 0, zseg = 1;                        // zseg: segments a circle is cut into (a function of nlon only)
    double *ztab = nullptr;                        // chain tables | reduced-system tables | per-latitude scalars | scratch (build_zonal_tables)
    double *zM = nullptr, *zE = nullptr, *zrM = nullptr, *zrE = nullptr, *za = nullptr, *za2 = nullptr, *zW = nullptr,
           *zsu = nullptr, *zsg = nullptr, *zsy = nullptr;       // views into ztab
    std::vector<double> xhost;                     // st.x (the zonal tables are built on demand)
    HostCopier *copier = nullptr;                  // pinned staging ring, lazily created by the first host transfer
    double *scratch = nullptr;                     // ebm_diffusion: temp | base | out, kept between calls
    // ebm_integrate's device buffers, kept between calls while the shape stays the same
    double *ig_sums = nullptr, *ig_mean = nullptr, *ig_snap = nullptr, *ig_stage = nullptr, *ig_hm = nullptr;
    size_t ig_sums_n = 0, ig_mean_n = 0, ig_snap_n = 0, ig_stage_n = 0, ig_hm_n = 0;
};

namespace {

// The handle's stream for everything that is not a step launch.  If the two chains of step launches are running apart
// (ebm_ctx::forked), the second one is joined first: whatever is enqueued next is ordered after all steps of all columns.
hipStream_t main_stream(ebm_ctx *h) {
    if (h->forked) {
        (void)hipEventRecord(h->ev_join, h->stream2);
        (void)hipStreamWaitEvent(h->stream, h->ev_join, 0);
        h->forked = false;
    }
    return h->stream;
}

// slab slot of a public field id for this model, -1 if the model does not have it
int slot_of(int model, int f) {
    if (model == EBM_MODEL_MIZ) return (f >= EBM_F_Ei && f <= EBM_F_T) ? f : -1;   // same order
    switch (f) {
        case EBM_F_E: return ebm::C_E;
        case EBM_F_Tg: return ebm::C_Tg;
        case EBM_F_T: return ebm::C_T;
        case EBM_F_h: return ebm::C_h;
        default: return -1;
    }
}
bool has_field(const ebm_ctx *h, int f) { return f >= 0 && f < EBM_F_COUNT && slot_of(h->model, f) >= 0; }

// fields that only diagnostic steps write (everything else is prognostic and always current)
bool is_diagnostic(const ebm_ctx *h, int f) {
    if (h->model == EBM_MODEL_MIZ)
        return f == EBM_F_T0 || f == EBM_F_Tw || f == EBM_F_Ti || f == EBM_F_n || f == EBM_F_E || f == EBM_F_T;
    return f == EBM_F_T || f == EBM_F_h;
}
const char *field_name(int f) {
    static const char *names[EBM_F_COUNT] = {"Ei", "Ew", "h", "D", "phi", "T0", "Tw", "Ti", "n", "E", "T", "Tg"};
    return (f >= 0 && f < EBM_F_COUNT) ? names[f] : "?";
}
// a step (or a run of steps ending at global index `step`) has been launched
void note_steps(ebm_ctx *h, long long nsteps, long long last_step, bool wrote_diag) {
    h->epoch += nsteps;
    h->state_step = last_step;
    if (wrote_diag)
        for (int f = 0; f < EBM_F_COUNT; ++f)
            if (has_field(h, f) && is_diagnostic(h, f)) {
                h->written_epoch[f] = h->epoch;
                h->written_step[f] = last_step;
            }
}
// EBM_OK if `f` may be read now, else EBM_ERR_STALE with the two steps in the message
int check_current(const ebm_ctx *h, int f, const char *who) {
    if (!is_diagnostic(h, f) || h->written_epoch[f] == h->epoch) return EBM_OK;
    std::string msg = std::string(who) + ": field " + field_name(f) + " is stale — ";
    if (h->written_epoch[f] < 0) msg += "it has never been written";
    else if (h->written_step[f] < 0) msg += "it holds what it held before the first step";
    else msg += "last written by step " + std::to_string(h->written_step[f]);
    msg += "; the state is at step " + std::to_string(h->state_step) +
           (h->written_step[f] == h->state_step && h->written_epoch[f] >= 0 ? " with prognostic fields overwritten since" : "") +
           " (take a step with write_diag / diag_last, or name the step: ebm_get_field_as_of)";
    return fail(EBM_ERR_STALE, msg);
}

// Per-latitude constants.  Same expressions, in the same order, as the reference:
// get_diffop (src/infrastructure.jl:480-492), the non-uniform cache (:509-518) and
// get_statics (src/classic.jl:18-29).
int build_tables(ebm_ctx *h, const double *x) {
    const int nx = h->nlat;
    const ebm::Params &p = h->p;
    std::vector<double> xv(x, x + nx), g0(nx), g1(nx), g2(nx), g3(nx, 0.0), g4(nx, 0.0), lo(nx), di(nx), up(nx);
    const bool uniform = (h->grid == EBM_GRID_IDENTITY) || (h->model == EBM_MODEL_CLASSIC);
    // classic: get_statics always uses get_diffop, whatever the grid type (src/classic.jl:21)
    const double Dscale = (h->model == EBM_MODEL_CLASSIC) ? 1.0 : p.D;
    if (uniform) {
        const double dx = 1.0 / nx;
        std::vector<double> lam(nx > 1 ? nx - 1 : 0);
        for (int i = 1; i < nx; ++i) {
            double xb = (double)i / nx;
            lam[i - 1] = (1.0 - xb * xb) / (dx * dx);
        }
        for (int k = 0; k < nx; ++k) {
            double sub = k > 0 ? lam[k - 1] : 0.0;
            double sup = k < nx - 1 ? lam[k] : 0.0;
            double l1 = k > 0 ? -lam[k - 1] : 0.0;
            double l2 = k < nx - 1 ? -lam[k] : 0.0;
            double l3 = (-l1) - l2;
            g0[k] = Dscale * sub;
            g1[k] = Dscale * (-l3);
            g2[k] = Dscale * sup;
            lo[k] = g0[k];
            di[k] = g1[k];
            up[k] = g2[k];
        }
    } else {
        for (int k = 0; k < nx; ++k) {
            double xk = x[k];
            double xm = k > 0 ? x[k - 1] : -x[0];
            double xp = k < nx - 1 ? x[k + 1] : 2.0 - x[nx - 1];
            double xxph = (xp + xk) / 2.0, xxmh = (xk + xm) / 2.0;
            g0[k] = 1.0 - xxph * xxph;
            g1[k] = 1.0 - xxmh * xxmh;
            g2[k] = xp - xk;
            g3[k] = xk - xm;
            g4[k] = xxph - xxmh;
            double u = p.D * g0[k] / (g2[k] * g4[k]);
            double l = p.D * g1[k] / (g3[k] * g4[k]);
            if (k == nx - 1) u = 0.0;
            if (k == 0) l = 0.0;
            lo[k] = l;
            up[k] = u;
            di[k] = -(l + u);
        }
    }
    // one zero-padded slab: table i at geom + i*gstride
    h->gstride = h->pitch;
    std::vector<double> slab((size_t)ebm::G_COUNT * h->gstride, 0.0);
    auto put = [&](int table, const std::vector<double> &v) {
        std::memcpy(slab.data() + (size_t)table * h->gstride, v.data(), sizeof(double) * v.size());
    };
    put(ebm::G_X, xv); put(ebm::G_0, g0); put(ebm::G_1, g1); put(ebm::G_2, g2); put(ebm::G_3, g3);
    put(ebm::G_4, g4); put(ebm::G_LO, lo); put(ebm::G_DI, di); put(ebm::G_UP, up);
    if (h->model == EBM_MODEL_CLASSIC) {
        std::vector<double> ksub(nx), kdiag(nx), ksup(nx), aw(nx), Sb(nx);
        const double dtD = h->dt * p.D;
        const double one = 1.0 + p.dt_tau;
        for (int k = 0; k < nx; ++k) {
            ksub[k] = 0.0 - (dtD * g0[k]) / p.cg;
            ksup[k] = 0.0 - (dtD * g2[k]) / p.cg;
            kdiag[k] = one - (dtD * g1[k]) / p.cg;
            aw[k] = p.a0 - p.a2 * (x[k] * x[k]);
            Sb[k] = p.S0 - p.S2 * (x[k] * x[k]);
        }
        put(ebm::G_KSUB, ksub); put(ebm::G_KDIAG, kdiag); put(ebm::G_KSUP, ksup);
        put(ebm::G_AW, aw); put(ebm::G_SB, Sb);
    }
    HIPCHK(hipMalloc(&h->geom, sizeof(double) * slab.size()));
    HIPCHK(hipMemcpy(h->geom, slab.data(), sizeof(double) * slab.size(), hipMemcpyHostToDevice));
    return EBM_OK;
}

void fill_params(ebm::Params &p, const double *v, double dt) {
    p.D = v[EBM_P_D]; p.A = v[EBM_P_A]; p.B = v[EBM_P_B]; p.cw = v[EBM_P_cw];
    p.S0 = v[EBM_P_S0]; p.S1 = v[EBM_P_S1]; p.S2 = v[EBM_P_S2]; p.a0 = v[EBM_P_a0];
    p.a2 = v[EBM_P_a2]; p.ai = v[EBM_P_ai]; p.Fb = v[EBM_P_Fb]; p.k = v[EBM_P_k];
    p.Lf = v[EBM_P_Lf]; p.F = v[EBM_P_F]; p.cg = v[EBM_P_cg]; p.tau = v[EBM_P_tau];
    p.Tm = v[EBM_P_Tm]; p.m1 = v[EBM_P_m1]; p.m2 = v[EBM_P_m2]; p.alpha = v[EBM_P_alpha];
    p.rl = v[EBM_P_rl]; p.Dmin = v[EBM_P_Dmin]; p.Dmax = v[EBM_P_Dmax]; p.hmin = v[EBM_P_hmin];
    p.kappa = v[EBM_P_kappa];
    p.dt = dt;
    p.Tm_pow_m2 = std::pow(p.Tm, p.m2);
    p.c_latmelt = -M_PI / 2.0 * p.alpha;
    p.c_dn = p.Lf * p.alpha * (p.Dmin * p.Dmin) * p.hmin;
    p.c_weld = p.kappa * p.alpha / 4.0;
    p.c_ht = -1.0 / p.Lf;
    p.two_rl = 2.0 * p.rl;
    p.cg_tau = p.cg / p.tau;
    p.dt_tau = dt / p.tau;
    p.dc = p.dt_tau * p.cg_tau;
    p.M = p.B + p.cg_tau;
    p.kLf = p.k * p.Lf;
    p.theta_imex = dt / p.cw;            // EBM_MODEL_MIZ_IMEX: the solve's matrix is I - theta*Dif
}

// Segments a latitude circle of nlon unknowns is cut into (zonal_seg_* kernels): a power of two between 4 and 32 that
// leaves segments of at least 64 unknowns, else 1 (zonal_sweep_kernel walks the whole circle).  A function of nlon ONLY —
// like the column geometry, never of how many members share the handle.
int zonal_segments(int nlon) {
    int S = 1;
    for (int c = 4; c <= 32; c *= 2)
        if (nlon % c == 0 && nlon / c >= 64) S = c;
    return S;
}

// Data-independent part of the periodic Thomas elimination of the system (-a, B, -a) of n unknowns (see zonal_sweep_kernel):
// m_l and ep_l for l = 0 .. n-2 into M / E (stride P), the reciprocal of the reduced last diagonal into *W.
void periodic_tables(double a, double B, int n, double *M, double *E, size_t P, double *W) {
    double cp_prev = 0.0, ep_prev = 0.0, gW = 0.0, f = -a, cp = 0.0, ep = 0.0;
    for (int l = 0; l <= n - 2; ++l) {
        const double m = 1.0 / (l == 0 ? B : B - a * cp_prev);
        cp = a * m;
        ep = l == 0 ? cp : a * ep_prev * m;
        M[(size_t)l * P] = m;
        E[(size_t)l * P] = ep;
        if (l <= n - 3) {
            gW += f * ep;
            f = -a * ep;                // f_{l+1} = f_l cp_l = -a ep_l
        }
        cp_prev = cp;
        ep_prev = ep;
    }
    *W = 1.0 / (B + gW + (f - a) * (cp + ep));       // f = f_{n-2}, cp / ep = those of row n-2
}

// Tables of the zonal substep (ebm_zonal_diffusion, include/ebm_hip.h; kernels: zonal_sweep_kernel, zonal_seg_*): per
// latitude the coefficient a_k = (dt/cw) D / ((1 - x_k)(1 + x_k) dlambda^2) and the data-independent part of the
// elimination, stored in the handle's store index space: with 4 cells per thread entry p = j*2T + 2t + q belongs to
// latitude k = 4t + 2j + q (pair-split), with 2 cells p = k; padding latitudes get a = 0 (U = temp, Z = 0).  One segment
// (S = 1): the whole circle's chain with its wrap closure.  S > 1: the chain of ONE segment of m = nlon/S unknowns (open
// ends), and the reduced periodic system of the S segment ends, (-a'', B'', -a'') with a'' = a ep_{m-2},
// B'' = B - a cp_{m-2} - a alpha, alpha = sum_i P_i ep_i.  Built on first use and whenever nlon changes.
int build_zonal_tables(ebm_ctx *h, int nlon) {
    if (h->ztab && h->nlon == nlon) return EBM_OK;
    const int S = zonal_segments(nlon), m = nlon / S, P = (int)h->pitch, T = h->cfg.threads;
    const int nmember = h->ncol / nlon;
    const double *x = h->xhost.data();
    const double dl = 2.0 * M_PI / nlon, theta = h->dt / h->p.cw;
    const size_t chain_rows = (size_t)(S == 1 ? nlon : m), red_rows = (size_t)(S == 1 ? 0 : S);
    const size_t scratch = S == 1 ? 0 : 3 * (size_t)nmember * S * P;
    std::vector<double> tab(2 * chain_rows * P + 2 * red_rows * P + 3 * (size_t)P, 0.0);
    double *zM = tab.data(), *zE = zM + chain_rows * P, *rM = zE + chain_rows * P, *rE = rM + red_rows * P,
           *za = rE + red_rows * P, *za2 = za + P, *zW = za2 + P;
    for (int p = 0; p < P; ++p) {
        int k = p;
        if (h->cfg.cells == 4) {
            const int j = p / (2 * T), rem = p % (2 * T), t = rem / 2, q = rem % 2;
            k = 4 * t + 2 * j + q;
        }
        double a = 0.0;
        if (k < h->nlat) {
            const double mm = (1.0 - x[k]) * (1.0 + x[k]);       // 1 - x^2 without the cancellation near the pole
            if (!(mm > 0.0)) return fail(EBM_ERR_ARG, "ebm_zonal_diffusion: needs |x| < 1 at every cell centre (the zonal coefficient is D/(1-x^2))");
            a = theta * h->p.D / (mm * (dl * dl));
        }
        const double B = 1.0 + 2.0 * a;
        za[p] = a;
        if (S == 1) {
            periodic_tables(a, B, nlon, zM + p, zE + p, (size_t)P, &zW[p]);
            continue;
        }
        double unused;
        periodic_tables(a, B, m, zM + p, zE + p, (size_t)P, &unused);      // the open chain's m_l, ep_l are the same recurrences
        double alpha = 0.0;
        for (int i = 0; i <= m - 2; ++i) alpha += (i == 0 ? 1.0 : zE[(size_t)(i - 1) * P + p]) * zE[(size_t)i * P + p];
        const double cp_last = a * zM[(size_t)(m - 2) * P + p], ep_last = zE[(size_t)(m - 2) * P + p];
        const double a2 = a * ep_last, B2 = B - a * cp_last - a * alpha;
        za2[p] = a2;
        periodic_tables(a2, B2, S, rM + p, rE + p, (size_t)P, &zW[p]);
    }
    HIPCHK(hipStreamSynchronize(main_stream(h)));
    if (h->ztab) (void)hipFree(h->ztab);
    h->ztab = nullptr;
    h->nlon = 0;
    HIPCHK(hipMalloc(&h->ztab, sizeof(double) * (tab.size() + scratch)));
    HIPCHK(hipMemcpy(h->ztab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    h->zM = h->ztab; h->zE = h->zM + chain_rows * P; h->zrM = h->zE + chain_rows * P; h->zrE = h->zrM + red_rows * P;
    h->za = h->zrE + red_rows * P; h->za2 = h->za + P; h->zW = h->za2 + P;
    h->zsu = h->zW + P; h->zsg = h->zsu + scratch / 3; h->zsy = h->zsg + scratch / 3;
    h->nlon = nlon;
    h->zseg = S;
    return EBM_OK;
}
hipError_t zonal_sweep(ebm_ctx *h, const double *T, double *outZ, double *outU) {
    const int nmember = h->ncol / h->nlon;
    const double rtheta = h->p.cw / h->dt;
    if (h->zseg == 1)
        return ebm::launch_zonal_sweep(T, outZ, outU, h->zM, h->zE, h->za, h->zW, h->nlon, nmember, (int)h->pitch, rtheta,
                                       main_stream(h));
    return ebm::launch_zonal_sweep_segmented(T, outZ, outU, h->zM, h->zE, h->zrM, h->zrE, h->za, h->za2, h->zW, h->zsu, h->zsg,
                                             h->zsy, h->nlon, h->zseg, nmember, (int)h->pitch, rtheta, main_stream(h));
}

ebm::StepArgs base_args(const ebm_ctx *h) {
    ebm::StepArgs a{};
    a.state = h->state; a.fstride = h->fstride; a.geom = h->geom; a.gstride = h->gstride;
    a.fcol = h->fcol; a.fsched = h->fsched; a.p = h->p_dev; a.counters = h->counters; a.amask = h->amask;
    a.pitch = (int)h->pitch; a.nlat = h->nlat; a.ncol = h->ncol;
    a.stamps = h->stamps;
    a.prefetch = h->prefetch;
    a.nfused = 1;
    std::memset(a.var_of, -1, sizeof(a.var_of));
    return a;
}

constexpr int kGraphSteps = 64;


// mode: ebm::OutMode.  The classic kernel decides about T, h at run time (write_diag).
hipError_t launch_columns(ebm_ctx *h, const ebm::StepArgs &a, int mode, int first, int count, hipStream_t s) {
    return (h->model == EBM_MODEL_MIZ) ? ebm::launch_miz_step(a, h->grid, mode, h->cfg, h->imex, first, count, s)
                                       : ebm::launch_classic_step(a, mode, h->cfg, first, count, s);
}
hipError_t launch_step(ebm_ctx *h, const ebm::StepArgs &a, int mode) {
    if (!h->split_col) return launch_columns(h, a, mode, 0, h->ncol, main_stream(h));
    if (!h->forked) {            // the second chain starts after everything the handle's stream has been given so far
        hipError_t e = hipEventRecord(h->ev_fork, h->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(h->stream2, h->ev_fork, 0);
        if (e != hipSuccess) return e;
        h->forked = true;
    }
    hipError_t e = launch_columns(h, a, mode, 0, h->split_col, h->stream);
    if (e == hipSuccess) e = launch_columns(h, a, mode, h->split_col, h->ncol - h->split_col, h->stream2);
    return e;
}

// Capture kGraphSteps step kernels (node i reads sched_dev[i]) into a graph, once per handle.
int build_graph(ebm_ctx *h) {
    HIPCHK(hipMalloc(&h->sched_dev, sizeof(ebm::StepSched) * kGraphSteps));
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(main_stream(h), hipStreamCaptureModeThreadLocal));
    hipError_t e = hipSuccess;
    for (int i = 0; i < kGraphSteps && e == hipSuccess; ++i) {
        ebm::StepArgs a = base_args(h);
        a.sched = h->sched_dev;
        a.slot = i;
        a.write_diag = 0;
        e = launch_step(h, a, ebm::OUT_STATE);
    }
    hipError_t e2 = hipStreamEndCapture(main_stream(h), &graph);
    if (e != hipSuccess || e2 != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return fail(EBM_ERR_HIP, std::string("graph capture: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    }
    e = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    return EBM_OK;
}

// model time of 0-based global step `step`: st.T[step+1] = (2 step + 1)/(2 nt), correctly rounded
double year_time(const ebm_ctx *h, long long step) {
    const double nt = (double)h->ttab.size();
    return nt > 0.0 ? (double)(2 * step + 1) / (2.0 * nt) : 0.0;
}

// drop the captured graph: its kernel nodes hold the argument values of the time of capture
void invalidate_graph(ebm_ctx *h) {
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
    if (h->sched_dev) (void)hipFree(h->sched_dev);
    h->sched_dev = nullptr;
}

// savesol! fused into a step launch (ebm::OUT_SAVE): where the running sums and the raw snapshot go
struct SaveTarget {
    double *sums = nullptr;
    long long sum_stride = 0;
    double *stage = nullptr;
    long long stage_var_stride = 0, stage_offset = 0;
    signed char var_of[ebm::kMaxQuantities];
};

int do_step(ebm_ctx *h, double ct, double ct_next, double f, int write_diag, long long step,
            const SaveTarget *save = nullptr) {
    ebm::StepArgs a = base_args(h);
    a.ct = ct; a.ct_next = ct_next; a.ft = f; a.write_diag = write_diag;
    a.tyear = year_time(h, step);
    int mode = write_diag ? ebm::OUT_DIAG : ebm::OUT_STATE;
    if (save) {
        mode = ebm::OUT_SAVE;
        a.sums = save->sums; a.sum_stride = save->sum_stride;
        a.stage = save->stage; a.stage_var_stride = save->stage_var_stride; a.stage_offset = save->stage_offset;
        std::memcpy(a.var_of, save->var_of, sizeof(a.var_of));
    }
    hipError_t e = launch_step(h, a, mode);
    if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    h->n_steps += 1;
    h->n_launches += h->split_col ? 2 : 1;
    h->clock = step + 1;
    note_steps(h, 1, step, write_diag != 0);
    // the 4-cells-per-thread MIZ step kernels leave the diagnostic fields pair-split (ensure_natural undoes it)
    if (write_diag && h->model == EBM_MODEL_MIZ) h->diag_split = h->cfg.cells == 4;
    return EBM_OK;
}

// Readers of a diagnostic field get the natural layout: un-permute in place once after a step that stored them split.
int ensure_natural(ebm_ctx *h) {
    if (!h->diag_split) return EBM_OK;
    hipError_t e = ebm::launch_unsplit_fields(h->field[EBM_F_Tw], h->fstride, 5, h->ncol, h->cfg, main_stream(h));
    if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("unsplit_fields: ") + hipGetErrorString(e));
    h->diag_split = false;
    return EBM_OK;
}
bool is_split_field(const ebm_ctx *h, int f) {
    return h->model == EBM_MODEL_MIZ && (f == EBM_F_Tw || f == EBM_F_Ti || f == EBM_F_n || f == EBM_F_E || f == EBM_F_T);
}

int get_copier(ebm_ctx *h) {
    if (h->copier) return EBM_OK;
    HostCopier *c = new HostCopier();
    hipError_t e = c->init(h->device);
    if (e != hipSuccess) {
        c->shutdown();
        delete c;
        return fail(EBM_ERR_HIP, std::string("pinned staging ring: ") + hipGetErrorString(e));
    }
    h->copier = c;
    return EBM_OK;
}
// (re)size one of the handle's kept device buffers
hipError_t keep_buffer(double **buf, size_t *have, size_t want) {
    if (*buf && *have >= want) return hipSuccess;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    hipError_t e = hipMalloc(buf, sizeof(double) * want);
    if (e == hipSuccess) *have = want;
    return e;
}

// quantity index (ebm::MizQuantity / ClassicQuantity) of a public field id, -1 if the step kernels
// do not produce it (the hidden warm start T0 is not a solution variable)
int quantity_of(int model, int f) {
    if (model == EBM_MODEL_MIZ) {
        switch (f) {
            case EBM_F_Ei: return ebm::Q_Ei;
            case EBM_F_Ew: return ebm::Q_Ew;
            case EBM_F_h: return ebm::Q_h;
            case EBM_F_D: return ebm::Q_D;
            case EBM_F_phi: return ebm::Q_phi;
            case EBM_F_n: return ebm::Q_n;
            case EBM_F_E: return ebm::Q_E;
            case EBM_F_T: return ebm::Q_T;
            case EBM_F_Ti: return ebm::Q_Ti;
            case EBM_F_Tw: return ebm::Q_Tw;
            default: return -1;
        }
    }
    switch (f) {
        case EBM_F_E: return ebm::QC_E;
        case EBM_F_Tg: return ebm::QC_Tg;
        case EBM_F_T: return ebm::QC_T;
        case EBM_F_h: return ebm::QC_h;
        default: return -1;
    }
}

constexpr int kFusedTable = 16384;     // per-step scalars resident on the device at a time (512 KiB)

}  // namespace

extern "C" {

const char *ebm_last_error(void) { return g_err.c_str(); }
const char *ebm_version(void) { return "ebm_hip 0.1 (gfx950)"; }

int ebm_options_default(ebm_options *opt) {
    if (!opt) return fail(EBM_ERR_ARG, "ebm_options_default: null argument");
    opt->struct_bytes = (int)sizeof(ebm_options);
    opt->cells_per_thread = 0;
    opt->use_graph = -1;
    opt->prefetch_cols = -1;
    opt->launch_chains = -1;
    opt->integrate_steps_per_launch = -1;
    opt->fused_state_in_lds = -1;
    return EBM_OK;
}

int ebm_create(ebm_handle_t *out, int model, int grid, int nlat, int ncol, const double *x,
               const double *params, double dt, int device) {
    return ebm_create_ex(out, model, grid, nlat, ncol, x, params, dt, device, nullptr);
}

int ebm_create_ex(ebm_handle_t *out, int model, int grid, int nlat, int ncol, const double *x,
                  const double *params, double dt, int device, const ebm_options *user_opt) {
    if (!out || !x || !params) return fail(EBM_ERR_ARG, "ebm_create: null argument");
    *out = nullptr;
    // options: the defaults, overwritten by as many fields as the caller's struct has (no environment is read)
    ebm_options opt;
    (void)ebm_options_default(&opt);
    if (user_opt) {
        if (user_opt->struct_bytes < (int)sizeof(int) || user_opt->struct_bytes > 4096)
            return fail(EBM_ERR_ARG, "ebm_create_ex: options.struct_bytes must be sizeof(ebm_options)");
        std::memcpy(&opt, user_opt, std::min((size_t)user_opt->struct_bytes, sizeof(opt)));
        opt.struct_bytes = (int)sizeof(opt);
    }
    if (opt.cells_per_thread != 0 && opt.cells_per_thread != 2 && opt.cells_per_thread != 4)
        return fail(EBM_ERR_ARG, "ebm_create_ex: cells_per_thread must be 0 (default), 2 or 4");
    if (opt.use_graph < -1 || opt.use_graph > 1) return fail(EBM_ERR_ARG, "ebm_create_ex: use_graph must be -1, 0 or 1");
    if (opt.prefetch_cols < -1) return fail(EBM_ERR_ARG, "ebm_create_ex: prefetch_cols must be -1 (default), 0 or a distance");
    if (opt.launch_chains != -1 && opt.launch_chains != 1 && opt.launch_chains != 2)
        return fail(EBM_ERR_ARG, "ebm_create_ex: launch_chains must be -1 (default), 1 or 2");
    if (opt.fused_state_in_lds < -1 || opt.fused_state_in_lds > 1)
        return fail(EBM_ERR_ARG, "ebm_create_ex: fused_state_in_lds must be -1 (default), 0 or 1");
    if (opt.integrate_steps_per_launch < -1)
        return fail(EBM_ERR_ARG, "ebm_create_ex: integrate_steps_per_launch must be -1 (default), 1 or a number of steps");
    if (model != EBM_MODEL_MIZ && model != EBM_MODEL_CLASSIC && model != EBM_MODEL_MIZ_IMEX)
        return fail(EBM_ERR_ARG, "ebm_create: unknown model");
    const bool imex = model == EBM_MODEL_MIZ_IMEX;        // the extension is the MIZ model with one more solve per step
    if (imex) model = EBM_MODEL_MIZ;
    if (grid != EBM_GRID_IDENTITY && grid != EBM_GRID_NONUNIFORM) return fail(EBM_ERR_ARG, "ebm_create: unknown grid kind");
    if (nlat < 2 || ncol < 1) return fail(EBM_ERR_ARG, "ebm_create: need nlat >= 2 and ncol >= 1");
    if (!(dt > 0.0)) return fail(EBM_ERR_ARG, "ebm_create: dt must be positive");
    if (model == EBM_MODEL_MIZ && params[EBM_P_Tm] < 0.0 && params[EBM_P_m2] != std::floor(params[EBM_P_m2]))
        return fail(EBM_ERR_ARG, "ebm_create: Tm^m2 with Tm < 0 and non-integer m2 (DomainError in the reference, src/miz.jl:71)");
    if (opt.cells_per_thread == 2 && (imex || nlat > ebm::kMaxLat2))
        return fail(EBM_ERR_ARG, "ebm_create_ex: cells_per_thread = 2 needs nlat <= 1536 and is not built for the IMEX extension");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(EBM_ERR_NO_DEVICE, "ebm_create: no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(EBM_ERR_ARG, "ebm_create: device index out of range");
    ebm::LaunchCfg cfg = ebm::choose_launch(nlat, opt.cells_per_thread);      // a function of nlat and the option only
    if (cfg.threads == 0)
        return fail(EBM_ERR_UNSUPPORTED, "ebm_create: nlat > 4096 is not supported (one workgroup owns a whole meridian)");
    HIPCHK(hipSetDevice(device));
    HIPCHK(ebm::prepare_kernels(cfg));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    ebm_ctx *h = new ebm_ctx();
    h->model = model; h->grid = grid; h->nlat = nlat; h->ncol = ncol; h->device = device;
    h->dt = dt; h->cfg = cfg; h->imex = imex;
    h->integrate_spl = opt.integrate_steps_per_launch <= 0 ? 64 : opt.integrate_steps_per_launch;
    for (int f = 0; f < EBM_F_COUNT; ++f) {
        h->written_epoch[f] = -1;
        h->written_step[f] = -1;
    }
    h->written_epoch[EBM_F_T0] = 0;                       // the warm start begins at zero, like the reference's (src/miz.jl:47)
    h->num_cus = prop.multiProcessorCount;
    // more columns than the register kernel runs in one round (it holds one 256-thread workgroup per CU, four of 64 threads:
    // tests/tools/r3/fused_choice_sweep.py): from there on occupancy beats latency.  The choice changes no bit.
    h->cfg.fused_in_lds = opt.fused_state_in_lds >= 0 ? opt.fused_state_in_lds != 0
                                                      : ncol > h->num_cus * std::max(1, 256 / h->cfg.threads);
    // a step of fewer than ~256K cells is launch-bound: replay graphs in ebm_run
    h->use_graph = opt.use_graph >= 0 ? opt.use_graph != 0 : ((long long)nlat * ncol <= 262144);
    {
        // One or two workgroups per CU (a long meridian fills the CU's LDS): little or nothing
        // overlaps the input loads of a workgroup, so each workgroup prefetches into L2 the inputs
        // of the one that follows it on its XCD (workgroups go round-robin over the XCDs and in
        // order within one).
        int per_cu = (int)((160u * 1024u) / cfg.lds_bytes);                  // workgroups a CU holds: LDS ...
        if (per_cu > 2048 / cfg.threads) per_cu = 2048 / cfg.threads;         // ... and wave slots
        const int ahead = h->num_cus * per_cu;                                // the successor on the same XCD
        // measured: -3.5 % time at one workgroup per CU, -2.5 % at two, nothing beyond
        h->prefetch = opt.prefetch_cols >= 0 ? opt.prefetch_cols : (per_cu <= 2 && ncol > ahead ? ahead : 0);
        // two chains of launches (see ebm_ctx::stream2): on request only — the default stays one launch per step, whose
        // duration a profiler reports as such; never with graph replay (one captured stream)
        h->split_col = (opt.launch_chains == 2 && !h->use_graph && ncol >= 2) ? ncol / 2 : 0;
    }
    h->pitch = (long long)cfg.threads * cfg.cells;     // >= nlat; padding cells stay zero
    fill_params(h->p, params, dt);
    h->xhost.assign(x, x + nlat);
    int rc = build_tables(h, x);
    if (rc) { ebm_destroy(h); return rc; }
    h->fstride = (long long)ncol * h->pitch;
    h->nslots = (model == EBM_MODEL_MIZ) ? (int)ebm::S_MIZ_COUNT : (int)ebm::C_COUNT;
    const size_t nbytes = sizeof(double) * (size_t)h->nslots * (size_t)h->fstride;
    hipError_t e = hipMalloc(&h->state, nbytes);
    if (e == hipSuccess) e = hipMemset(h->state, 0, nbytes);
    if (e == hipSuccess && model == EBM_MODEL_MIZ) {
        const size_t mb = sizeof(unsigned short) * (size_t)ncol * cfg.threads;
        e = hipMalloc(&h->amask, mb);
        if (e == hipSuccess) e = hipMemset(h->amask, 0, mb);
    }
    if (e == hipSuccess) e = hipMalloc(&h->p_dev, sizeof(ebm::Params));
    if (e == hipSuccess) e = hipMemcpy(h->p_dev, &h->p, sizeof(ebm::Params), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = ebm::launch_derive_params(h->p_dev, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMalloc(&h->hm_dev, sizeof(double) * (size_t)ncol);
    if (e != hipSuccess) { ebm_destroy(h); return fail(EBM_ERR_HIP, std::string("state allocation: ") + hipGetErrorString(e)); }
    for (int f = 0; f < EBM_F_COUNT; ++f) {
        const int slot = slot_of(model, f);
        h->field[f] = slot >= 0 ? h->state + (size_t)slot * h->fstride : nullptr;
    }
    e = hipMalloc(&h->counters, sizeof(unsigned long long) * 2 * ebm::kCounterShards);
    if (e == hipSuccess) e = hipMemset(h->counters, 0, sizeof(unsigned long long) * 2 * ebm::kCounterShards);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess && h->split_col) e = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking);
    if (e == hipSuccess && h->split_col) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess && h->split_col) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) { ebm_destroy(h); return fail(EBM_ERR_HIP, std::string("ebm_create: ") + hipGetErrorString(e)); }
    *out = h;
    return EBM_OK;
}

int ebm_destroy(ebm_handle_t h) {
    if (!h) return EBM_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(main_stream(h));
    if (h->copier) {
        h->copier->shutdown();
        delete h->copier;
    }
    for (double *b : {h->scratch, h->ig_sums, h->ig_mean, h->ig_snap, h->ig_stage, h->ig_hm, h->ztab})
        if (b) (void)hipFree(b);
    if (h->geom) (void)hipFree(h->geom);
    if (h->state) (void)hipFree(h->state);
    if (h->p_dev) (void)hipFree(h->p_dev);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->sched_dev) (void)hipFree(h->sched_dev);
    for (auto &tb : h->sched_tab) {
        if (tb.dev) (void)hipFree(tb.dev);
        if (tb.done) (void)hipEventDestroy(tb.done);
    }
    if (h->hm_dev) (void)hipFree(h->hm_dev);
    if (h->amask) (void)hipFree(h->amask);
    if (h->fcol) (void)hipFree(h->fcol);
    if (h->fsched) (void)hipFree(h->fsched);
    if (h->counters) (void)hipFree(h->counters);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return EBM_OK;
}

int ebm_set_field(ebm_handle_t h, int field, const double *host) {
    if (!h || !host) return fail(EBM_ERR_ARG, "ebm_set_field: null argument");
    if (!has_field(h, field)) return fail(EBM_ERR_ARG, "ebm_set_field: field not part of this model");
    HIPCHK(hipSetDevice(h->device));
    int rc = get_copier(h);
    if (rc) return rc;
    if (is_split_field(h, field)) {
        rc = ensure_natural(h);                   // the other diagnostic fields keep their values, in the natural layout
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(main_stream(h)));
    HIPCHK(h->copier->wait_all());
    HIPCHK(h->copier->upload(h->field[field], (size_t)h->pitch, host, (size_t)h->nlat, (size_t)h->ncol));
    if (field == EBM_F_T0 && h->model == EBM_MODEL_MIZ) {
        // the stepping kernels carry the warm start as its active set: rebuild it from the new T0
        hipError_t e = ebm::launch_mask_from_t0(base_args(h), h->ncol, h->cfg, main_stream(h));
        if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("mask_from_t0: ") + hipGetErrorString(e));
        HIPCHK(hipStreamSynchronize(main_stream(h)));
    }
    if (is_diagnostic(h, field)) {                // the caller's statement of what the field holds: current as of now
        h->written_epoch[field] = h->epoch;
        h->written_step[field] = h->state_step;
    } else {
        h->epoch += 1;                            // the prognostic state changed: every diagnostic field is older than it now
    }
    return EBM_OK;
}

// device -> host through the pinned ring (synchronous)
static int download_field(ebm_handle_t h, int field, double *host, const char *who) {
    int rc = get_copier(h);
    if (rc) return rc;
    if (is_split_field(h, field)) {
        rc = ensure_natural(h);
        if (rc) return rc;
    }
    HostCopier *c = h->copier;
    HIPCHK(c->wait_all());
    HIPCHK(c->order_after(main_stream(h)));
    CopyJob j;
    j.src = h->field[field]; j.src_pitch = (size_t)h->pitch; j.row_elems = (size_t)h->nlat; j.nrows = (size_t)h->ncol; j.dst = host;
    hipError_t e = c->run(j);
    if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
    return EBM_OK;
}

int ebm_get_field(ebm_handle_t h, int field, double *host) {
    if (!h || !host) return fail(EBM_ERR_ARG, "ebm_get_field: null argument");
    if (!has_field(h, field)) return fail(EBM_ERR_ARG, "ebm_get_field: field not part of this model");
    int rc = check_current(h, field, "ebm_get_field");
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    return download_field(h, field, host, "ebm_get_field");
}

int ebm_get_field_as_of(ebm_handle_t h, int field, long long step, double *host) {
    if (!h || !host) return fail(EBM_ERR_ARG, "ebm_get_field_as_of: null argument");
    if (!has_field(h, field)) return fail(EBM_ERR_ARG, "ebm_get_field_as_of: field not part of this model");
    const long long have = is_diagnostic(h, field) ? (h->written_epoch[field] >= 0 ? h->written_step[field] : -2) : h->state_step;
    if (have != step)
        return fail(EBM_ERR_STALE, std::string("ebm_get_field_as_of: field ") + field_name(field) + " is not as of step " +
                                       std::to_string(step) + (have == -2 ? " (it has never been written)"
                                                                          : " (it was last written by step " + std::to_string(have) + ")"));
    HIPCHK(hipSetDevice(h->device));
    return download_field(h, field, host, "ebm_get_field_as_of");
}

int ebm_field_step(ebm_handle_t h, int field, long long *written_step, long long *state_step, int *current) {
    if (!h) return fail(EBM_ERR_ARG, "ebm_field_step: null handle");
    if (!has_field(h, field)) return fail(EBM_ERR_ARG, "ebm_field_step: field not part of this model");
    const bool diag = is_diagnostic(h, field);
    if (written_step) *written_step = diag ? (h->written_epoch[field] >= 0 ? h->written_step[field] : -1) : h->state_step;
    if (state_step) *state_step = h->state_step;
    if (current) *current = (!diag || h->written_epoch[field] == h->epoch) ? 1 : 0;
    return EBM_OK;
}

int ebm_hemispheric_mean(ebm_handle_t h, int field, double *out) {
    if (!h || !out) return fail(EBM_ERR_ARG, "ebm_hemispheric_mean: null argument");
    if (!has_field(h, field)) return fail(EBM_ERR_ARG, "ebm_hemispheric_mean: field not part of this model");
    int rc = check_current(h, field, "ebm_hemispheric_mean");
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    if (is_split_field(h, field) && (rc = ensure_natural(h))) return rc;
    hipError_t e = ebm::launch_hemispheric_mean(h->field[field], h->geom + (size_t)ebm::G_X * h->gstride, (int)h->pitch,
                                                h->nlat, h->ncol, h->hm_dev, main_stream(h));
    if (e == hipSuccess) e = hipMemcpyAsync(out, h->hm_dev, sizeof(double) * (size_t)h->ncol, hipMemcpyDeviceToHost, main_stream(h));
    if (e == hipSuccess) e = hipStreamSynchronize(main_stream(h));
    if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("ebm_hemispheric_mean: ") + hipGetErrorString(e));
    return EBM_OK;
}

int ebm_hemispheric_mean_device(ebm_handle_t h, int field, double *dev_out) {
    if (!h || !dev_out) return fail(EBM_ERR_ARG, "ebm_hemispheric_mean_device: null argument");
    if (!has_field(h, field)) return fail(EBM_ERR_ARG, "ebm_hemispheric_mean_device: field not part of this model");
    int rc = check_current(h, field, "ebm_hemispheric_mean_device");
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    if (is_split_field(h, field) && (rc = ensure_natural(h))) return rc;
    hipError_t e = ebm::launch_hemispheric_mean(h->field[field], h->geom + (size_t)ebm::G_X * h->gstride, (int)h->pitch,
                                                h->nlat, h->ncol, dev_out, main_stream(h));
    if (e == hipSuccess) e = hipStreamSynchronize(main_stream(h));
    if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("ebm_hemispheric_mean_device: ") + hipGetErrorString(e));
    return EBM_OK;
}

int ebm_get_field_device(ebm_handle_t h, int field, double *dev_out) {
    if (!h || !dev_out) return fail(EBM_ERR_ARG, "ebm_get_field_device: null argument");
    if (!has_field(h, field)) return fail(EBM_ERR_ARG, "ebm_get_field_device: field not part of this model");
    int rc = check_current(h, field, "ebm_get_field_device");
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    if (is_split_field(h, field) && (rc = ensure_natural(h))) return rc;
    HIPCHK(hipMemcpy2DAsync(dev_out, sizeof(double) * h->nlat, h->field[field], sizeof(double) * h->pitch,
                            sizeof(double) * h->nlat, h->ncol, hipMemcpyDeviceToDevice, main_stream(h)));
    HIPCHK(hipStreamSynchronize(main_stream(h)));
    return EBM_OK;
}

int ebm_diffusion(ebm_handle_t h, const double *temp, const double *base, double *out) {
    if (!h || !temp || !out) return fail(EBM_ERR_ARG, "ebm_diffusion: null argument");
    if (h->model != EBM_MODEL_MIZ)
        return fail(EBM_ERR_ARG, "ebm_diffusion: needs a MIZ handle (the classic model carries get_diffop unscaled inside kappa, src/classic.jl:21)");
    HIPCHK(hipSetDevice(h->device));
    const size_t npitch = (size_t)h->ncol * h->pitch;
    if (!h->scratch) {                                   // temp | base | out, [ncol][pitch] each: kept until ebm_destroy
        HIPCHK(hipMalloc(&h->scratch, sizeof(double) * npitch * 3));
        HIPCHK(hipMemsetAsync(h->scratch, 0, sizeof(double) * npitch * 3, main_stream(h)));       // padding cells stay zero
    }
    double *buf = h->scratch;
    auto up = [&](double *dst, const double *src) {
        return hipMemcpy2DAsync(dst, sizeof(double) * h->pitch, src, sizeof(double) * h->nlat, sizeof(double) * h->nlat,
                                h->ncol, hipMemcpyHostToDevice, main_stream(h));
    };
    hipError_t e = up(buf, temp);
    if (e == hipSuccess && base) e = up(buf + npitch, base);
    if (e == hipSuccess)
        e = ebm::launch_diffusion(buf, base ? buf + npitch : nullptr, buf + 2 * npitch, h->geom, h->gstride, h->p_dev,
                                  h->grid, (int)h->pitch, h->nlat, h->ncol, main_stream(h));
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(out, sizeof(double) * h->nlat, buf + 2 * npitch, sizeof(double) * h->pitch,
                             sizeof(double) * h->nlat, h->ncol, hipMemcpyDeviceToHost, main_stream(h));
    if (e == hipSuccess) e = hipStreamSynchronize(main_stream(h));
    if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("ebm_diffusion: ") + hipGetErrorString(e));
    return EBM_OK;
}

int ebm_zonal_diffusion(ebm_handle_t h, int nlon, const double *temp, double *out_U, double *out_Z) {
    if (!h || !temp || (!out_U && !out_Z)) return fail(EBM_ERR_ARG, "ebm_zonal_diffusion: null argument");
    if (h->model != EBM_MODEL_MIZ) return fail(EBM_ERR_ARG, "ebm_zonal_diffusion: needs a MIZ handle (cw and D are MIZ parameters of this operator)");
    if (nlon < 3) return fail(EBM_ERR_ARG, "ebm_zonal_diffusion: needs nlon >= 3 (longitudes per member)");
    if (h->ncol % nlon) return fail(EBM_ERR_ARG, "ebm_zonal_diffusion: the handle's column count must be a multiple of nlon");
    HIPCHK(hipSetDevice(h->device));
    int rc = build_zonal_tables(h, nlon);
    if (rc) return rc;
    const size_t npitch = (size_t)h->ncol * h->pitch;
    if (!h->scratch) {                                   // temp | U | Z, [ncol][pitch] each: kept until ebm_destroy
        HIPCHK(hipMalloc(&h->scratch, sizeof(double) * npitch * 3));
        HIPCHK(hipMemsetAsync(h->scratch, 0, sizeof(double) * npitch * 3, main_stream(h)));       // padding cells stay zero
    }
    double *buf = h->scratch;
    hipError_t e = hipMemsetAsync(buf, 0, sizeof(double) * npitch, main_stream(h));            // (an earlier call left it permuted)
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(buf, sizeof(double) * h->pitch, temp, sizeof(double) * h->nlat, sizeof(double) * h->nlat,
                             h->ncol, hipMemcpyHostToDevice, main_stream(h));
    if (e == hipSuccess) e = ebm::launch_split_fields(buf, 0, 1, h->ncol, h->cfg, main_stream(h));
    if (e == hipSuccess) e = zonal_sweep(h, buf, buf + 2 * npitch, buf + npitch);
    if (e == hipSuccess) e = ebm::launch_unsplit_fields(buf + npitch, (long long)npitch, 2, h->ncol, h->cfg, main_stream(h));
    auto down = [&](double *dst, const double *src) {
        return hipMemcpy2DAsync(dst, sizeof(double) * h->nlat, src, sizeof(double) * h->pitch, sizeof(double) * h->nlat,
                                h->ncol, hipMemcpyDeviceToHost, main_stream(h));
    };
    if (e == hipSuccess && out_U) e = down(out_U, buf + npitch);
    if (e == hipSuccess && out_Z) e = down(out_Z, buf + 2 * npitch);
    if (e == hipSuccess) e = hipStreamSynchronize(main_stream(h));
    if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("ebm_zonal_diffusion: ") + hipGetErrorString(e));
    return EBM_OK;
}

int ebm_field_device_ptr(ebm_handle_t h, int field, double **dptr, long long *pitch) {
    if (!h || !dptr) return fail(EBM_ERR_ARG, "ebm_field_device_ptr: null argument");
    if (!has_field(h, field)) return fail(EBM_ERR_ARG, "ebm_field_device_ptr: field not part of this model");
    int rc = check_current(h, field, "ebm_field_device_ptr");
    if (rc) return rc;
    if (is_split_field(h, field)) {                      // the view is of the natural layout as of this call
        HIPCHK(hipSetDevice(h->device));
        if ((rc = ensure_natural(h))) return rc;
        HIPCHK(hipStreamSynchronize(main_stream(h)));
    }
    *dptr = h->field[field];
    if (pitch) *pitch = h->pitch;
    return EBM_OK;
}

int ebm_set_column_forcing(ebm_handle_t h, const double *fcol) {
    if (!h) return fail(EBM_ERR_ARG, "ebm_set_column_forcing: null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(main_stream(h)));
    if (!fcol) {
        if (h->fcol) { HIPCHK(hipFree(h->fcol)); invalidate_graph(h); }
        h->fcol = nullptr;
        return EBM_OK;
    }
    if (!h->fcol) { HIPCHK(hipMalloc(&h->fcol, sizeof(double) * h->ncol)); invalidate_graph(h); }
    HIPCHK(hipMemcpy(h->fcol, fcol, sizeof(double) * h->ncol, hipMemcpyHostToDevice));
    return EBM_OK;
}

int ebm_set_column_schedule(ebm_handle_t h, const double *sched) {
    if (!h) return fail(EBM_ERR_ARG, "ebm_set_column_schedule: null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(main_stream(h)));
    const size_t nb = sizeof(double) * ebm::kSchedWords * (size_t)h->ncol;
    if (!sched) {
        if (h->fsched) { HIPCHK(hipFree(h->fsched)); invalidate_graph(h); }
        h->fsched = nullptr;
        return EBM_OK;
    }
    for (int c = 0; c < h->ncol; ++c) {
        const double *w = sched + (size_t)ebm::kSchedWords * c;
        if (!(w[5] <= w[6] && w[6] <= w[7] && w[7] <= w[8]))
            return fail(EBM_ERR_ARG, "ebm_set_column_schedule: breakpoints must be non-decreasing");
    }
    if (!h->fsched) { HIPCHK(hipMalloc(&h->fsched, nb)); invalidate_graph(h); }
    HIPCHK(hipMemcpy(h->fsched, sched, nb, hipMemcpyHostToDevice));
    return EBM_OK;
}

int ebm_set_step_clock(ebm_handle_t h, long long step) {
    if (!h || step < 0) return fail(EBM_ERR_ARG, "ebm_set_step_clock: bad argument");
    h->clock = step;
    return EBM_OK;
}

int ebm_set_time_table(ebm_handle_t h, int nt, const double *cos2pit) {
    if (!h || !cos2pit || nt < 1) return fail(EBM_ERR_ARG, "ebm_set_time_table: bad argument");
    h->ttab.assign(cos2pit, cos2pit + nt);
    return EBM_OK;
}

int ebm_step(ebm_handle_t h, double cos2pit, double cos2pit_next, double f, int write_diag) {
    if (!h) return fail(EBM_ERR_ARG, "ebm_step: null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->fsched && h->ttab.empty()) return fail(EBM_ERR_ARG, "ebm_step: column schedules need the time table (ebm_set_time_table)");
    return do_step(h, cos2pit, cos2pit_next, f, write_diag, h->clock);
}

int ebm_run(ebm_handle_t h, long long first_step, int nsteps, const double *f_steps, int diag_last) {
    if (!h || nsteps < 0 || first_step < 0) return fail(EBM_ERR_ARG, "ebm_run: bad argument");
    if (h->ttab.empty()) return fail(EBM_ERR_ARG, "ebm_run: call ebm_set_time_table first");
    HIPCHK(hipSetDevice(h->device));
    const long long nt = (long long)h->ttab.size();
    int s = 0;
    if (h->use_graph && nsteps >= 2 * kGraphSteps) {
        // launch-bound shapes: replay a captured graph of kGraphSteps launches (still one launch
        // per step); the per-step scalars travel through a small device table
        if (!h->graph_exec) {
            int rc = build_graph(h);
            if (rc) return rc;
        }
        std::vector<ebm::StepSched> sched(kGraphSteps);
        const int last_graph_step = nsteps - (diag_last ? 1 : 0);     // a diagnostic last step is launched directly
        for (; s + kGraphSteps <= last_graph_step; s += kGraphSteps) {
            for (int i = 0; i < kGraphSteps; ++i) {
                const long long ti = (first_step + s + i) % nt;
                sched[i].ct = h->ttab[ti];
                sched[i].ct_next = h->ttab[(ti + 1) % nt];
                sched[i].ft = f_steps ? f_steps[s + i] : 0.0;
                sched[i].tyear = year_time(h, first_step + s + i);
            }
            // pageable source: the copy is staged before the call returns, so `sched` can be refilled
            HIPCHK(hipMemcpyAsync(h->sched_dev, sched.data(), sizeof(ebm::StepSched) * kGraphSteps,
                                  hipMemcpyHostToDevice, main_stream(h)));
            HIPCHK(hipGraphLaunch(h->graph_exec, main_stream(h)));
            h->n_steps += kGraphSteps;
            h->n_launches += kGraphSteps;
            h->clock = first_step + s + kGraphSteps;
            note_steps(h, kGraphSteps, first_step + s + kGraphSteps - 1, false);
        }
    }
    for (; s < nsteps; ++s) {
        const long long ti = (first_step + s) % nt;
        const double f = f_steps ? f_steps[s] : 0.0;
        int rc = do_step(h, h->ttab[ti], h->ttab[(ti + 1) % nt], f, diag_last && s == nsteps - 1, first_step + s);
        if (rc) return rc;
    }
    return EBM_OK;
}

// nsteps steps, steps_per_launch to a launch, the per-step scalars from a device table: time-table entry tab_first + i and
// model-time step clock_first + i for step i.  save: savesol!'s running sums from every step (OUT_LOOP_SAVE), else plain
// fused stepping (OUT_LOOP).
static int fused_range(ebm_ctx *h, long long tab_first, long long clock_first, int nsteps, const double *f_steps, int diag_last,
                       int steps_per_launch, const SaveTarget *save) {
    const long long first_step = clock_first;
    const long long nt = (long long)h->ttab.size();
    std::vector<ebm::StepSched> sched;
    for (int s0 = 0; s0 < nsteps; s0 += kFusedTable) {
        const int n = std::min(kFusedTable, nsteps - s0);
        sched.resize(n);
        for (int i = 0; i < n; ++i) {
            const long long ti = (tab_first + s0 + i) % nt;
            sched[i].ct = h->ttab[ti];
            sched[i].ct_next = h->ttab[(ti + 1) % nt];
            sched[i].ft = f_steps ? f_steps[s0 + i] : 0.0;
            sched[i].tyear = year_time(h, first_step + s0 + i);
        }
        // the table used two batches ago: its launches must have ended before it is refilled (normally long since).  The copy is
        // synchronous for the host but not ordered with the handle's (non-blocking) streams.
        auto &tb = h->sched_tab[h->sched_next];
        h->sched_next ^= 1;
        if (!tb.dev) {
            HIPCHK(hipMalloc(&tb.dev, sizeof(ebm::StepSched) * kFusedTable));
            HIPCHK(hipEventCreateWithFlags(&tb.done, hipEventDisableTiming));
        }
        if (tb.in_use) HIPCHK(hipEventSynchronize(tb.done));
        HIPCHK(hipMemcpy(tb.dev, sched.data(), sizeof(ebm::StepSched) * (size_t)n, hipMemcpyHostToDevice));
        for (int i = 0; i < n; i += steps_per_launch) {
            ebm::StepArgs a = base_args(h);
            a.sched = tb.dev;
            a.slot = i;
            a.nfused = std::min(steps_per_launch, n - i);
            a.prefetch = 0;
            a.write_diag = (diag_last && s0 + i + a.nfused == nsteps) ? 1 : 0;
            if (save) {
                a.sums = save->sums; a.sum_stride = save->sum_stride;
                a.stage = nullptr;
                std::memcpy(a.var_of, save->var_of, sizeof(a.var_of));
            }
            hipError_t e = launch_step(h, a, save ? ebm::OUT_LOOP_SAVE : ebm::OUT_LOOP);
            if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("fused launch: ") + hipGetErrorString(e));
            h->n_launches += h->split_col ? 2 : 1;
            note_steps(h, a.nfused, first_step + s0 + i + a.nfused - 1, a.write_diag != 0);
            if (a.write_diag && h->model == EBM_MODEL_MIZ) h->diag_split = false;      // the fused kernel stores them in the natural layout
        }
        HIPCHK(hipEventRecord(tb.done, main_stream(h)));     // (both launch chains, joined)
        tb.in_use = true;
        h->n_steps += n;
        h->clock = first_step + s0 + n;
    }
    return EBM_OK;
}

int ebm_run_fused(ebm_handle_t h, long long first_step, int nsteps, const double *f_steps, int diag_last,
                  int steps_per_launch) {
    if (!h || nsteps < 0 || first_step < 0 || steps_per_launch < 1) return fail(EBM_ERR_ARG, "ebm_run_fused: bad argument");
    if (h->ttab.empty()) return fail(EBM_ERR_ARG, "ebm_run_fused: call ebm_set_time_table first");
    // every shape has a fused-K kernel: the state in registers up to kFusedRegThreads threads per meridian (2048 cells at
    // 4 per thread; kFusedRegThreads2 at 2 per thread), resident in LDS for longer meridians and for the extension
    if (steps_per_launch == 1) return ebm_run(h, first_step, nsteps, f_steps, diag_last);
    HIPCHK(hipSetDevice(h->device));
    return fused_range(h, first_step, first_step, nsteps, f_steps, diag_last, steps_per_launch, nullptr);
}

// integrate + savesol! (ebm_integrate) with, optionally, the per-column hemispheric means of the seasonal
// outputs reduced on the device (ebm_integrate_hemispheric): hm_* are [nvars][dur][ncol] host arrays.
//
// Host output never stalls the stepping: what has to leave the device is first copied device -> device into a
// buffer of its own on the compute stream (seasonal snapshots; the annual means come out of ONE finish-mean
// launch; raw snapshots are written by the step kernel into one half of a two-part staging buffer), then the
// handle's copier moves it to the caller's arrays — DMA into the pinned ring on its own stream, host threads
// from there — while the following steps run.  A buffer is reused only after the job that reads it has finished.
static int integrate_impl(ebm_handle_t h, int nt, int dur, const double *f_steps, int lastonly,
                          int winter_inx, int summer_inx, int nvars, const int *fields, double *raw,
                          double *winter, double *summer, double *avg, double *hm_winter, double *hm_summer,
                          double *hm_avg) {
    if (!h || nt < 1 || dur < 1 || nvars < 0 || nvars > ebm::kMaxQuantities || (nvars > 0 && !fields))
        return fail(EBM_ERR_ARG, "ebm_integrate: bad argument");
    if ((hm_winter || hm_summer || hm_avg) && nvars < 1) return fail(EBM_ERR_ARG, "ebm_integrate_hemispheric: no variables");
    if ((long long)h->ttab.size() != nt) return fail(EBM_ERR_ARG, "ebm_integrate: time table length must equal nt");
    SaveTarget save;
    std::memset(save.var_of, -1, sizeof(save.var_of));
    for (int v = 0; v < nvars; ++v) {
        const int q = has_field(h, fields[v]) ? quantity_of(h->model, fields[v]) : -1;
        if (q < 0) return fail(EBM_ERR_ARG, "ebm_integrate: not a solution variable of this model");
        if (save.var_of[q] >= 0) return fail(EBM_ERR_ARG, "ebm_integrate: a variable is listed twice");
        save.var_of[q] = (signed char)v;
    }
    HIPCHK(hipSetDevice(h->device));
    int rc = get_copier(h);
    if (rc) return rc;
    HostCopier *cp = h->copier;
    HIPCHK(cp->wait_all());
    const size_t ncell = (size_t)h->ncol * h->nlat;          // packed cells per snapshot (host side)
    const size_t npitch = (size_t)h->ncol * h->pitch;        // device elements per field
    const long long total = (long long)nt * dur;
    const long long nraw = lastonly ? nt : total;
    const bool want_hm = (hm_winter || hm_summer || hm_avg) && nvars > 0;
    const bool want_sums = (avg || hm_avg) && nvars > 0;
    const bool want_snap = (winter || summer) && nvars > 0;
    // Any failure: let the copier finish what it was given (it reads this call's device buffers) before returning.
#define EBM_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (void)cp->wait_all();                                                         \
            return fail(EBM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
        }                                                                                 \
    } while (0)
    // Device buffers (kept in the handle between calls): raw snapshots are staged as two halves of
    // [var][chunk][ncol][pitch]; the annual-mean sums are [var][ncol*pitch] (pair-split layout), the means and the
    // seasonal snapshots [var][ncol*pitch] in the natural layout.
    long long chunk = 0;
    if (raw && nvars > 0) {
        chunk = (long long)((128ull << 20) / (sizeof(double) * npitch * (size_t)nvars));
        if (chunk < 1) chunk = 1;
        if (chunk > nraw) chunk = nraw;
        EBM_TRY(keep_buffer(&h->ig_stage, &h->ig_stage_n, 2 * npitch * (size_t)nvars * (size_t)chunk));
    }
    if (want_hm) EBM_TRY(keep_buffer(&h->ig_hm, &h->ig_hm_n, (size_t)h->ncol * (size_t)nvars));
    if (want_sums) {
        EBM_TRY(keep_buffer(&h->ig_sums, &h->ig_sums_n, npitch * (size_t)nvars));
        EBM_TRY(hipMemsetAsync(h->ig_sums, 0, sizeof(double) * npitch * (size_t)nvars, main_stream(h)));
        EBM_TRY(keep_buffer(&h->ig_mean, &h->ig_mean_n, npitch * (size_t)nvars));
    }
    if (want_snap) EBM_TRY(keep_buffer(&h->ig_snap, &h->ig_snap_n, npitch * (size_t)nvars));
    double *const sums = want_sums ? h->ig_sums : nullptr, *const mean = h->ig_mean, *const snap = h->ig_snap;
    double *const hm = h->ig_hm;
    double *const stage = (raw && nvars > 0) ? h->ig_stage : nullptr;
    // hemispheric_mean (src/utilities.jl:397-403) of every saved variable of a padded device field set,
    // reduced on the device, [nvars][ncol] -> out[v][year][col]
    auto means_to_host = [&](double *out, long long year, auto field_of) -> hipError_t {
        for (int v = 0; v < nvars; ++v) {
            hipError_t e = ebm::launch_hemispheric_mean(field_of(v), h->geom + (size_t)ebm::G_X * h->gstride, (int)h->pitch,
                                                        h->nlat, h->ncol, hm + (size_t)v * h->ncol, main_stream(h));
            if (e != hipSuccess) return e;
        }
        hipError_t e = hipStreamSynchronize(main_stream(h));
        for (int v = 0; v < nvars && e == hipSuccess; ++v)
            e = hipMemcpy(out + ((size_t)v * dur + (size_t)(year - 1)) * h->ncol, hm + (size_t)v * h->ncol,
                          sizeof(double) * (size_t)h->ncol, hipMemcpyDeviceToHost);
        return e;
    };
    save.sums = sums;
    save.sum_stride = (long long)npitch;
    save.stage_var_stride = chunk * (long long)npitch;
    // one asynchronous job per saved variable: [ncol][pitch] on the device -> packed [ncol][nlat] at dst
    auto fields_to_host = [&](double *dst_base, long long year, const double *dev_base) -> hipError_t {
        hipError_t e = cp->order_after(main_stream(h));
        for (int v = 0; v < nvars && e == hipSuccess; ++v) {
            CopyJob j;
            j.src = dev_base + (size_t)v * npitch; j.src_pitch = (size_t)h->pitch; j.row_elems = (size_t)h->nlat;
            j.nrows = (size_t)h->ncol; j.dst = dst_base + ((size_t)v * dur + (size_t)(year - 1)) * ncell;
            cp->submit(j);
        }
        return e;
    };
    // seasonal snapshot: the state fields of this step, device -> device, then out
    auto season_to_host = [&](double *dst_base, long long year) -> hipError_t {
        hipError_t e = cp->wait_all();                                       // the previous snapshot has left `snap`
        for (int v = 0; v < nvars && e == hipSuccess; ++v)
            e = hipMemcpyAsync(snap + (size_t)v * npitch, h->field[fields[v]], sizeof(double) * npitch, hipMemcpyDeviceToDevice, main_stream(h));
        if (e == hipSuccess) e = fields_to_host(dst_base, year, snap);
        return e;
    };
    long long staged = 0, raw_base = 0;   // snapshots in the current half of the staging buffer; raw index of its first
    int half = 0;
    const long long clock0 = h->clock;    // model time continues from the handle's step clock (0 after ebm_create)
    auto flush = [&]() -> hipError_t {
        if (!staged) return hipSuccess;
        // the half just filled goes out while the steps fill the other one — whose previous contents must have left
        hipError_t e = cp->wait_all();
        if (e == hipSuccess) e = cp->order_after(main_stream(h));
        for (int v = 0; v < nvars && e == hipSuccess; ++v) {
            CopyJob j;      // `staged` snapshots of ncol rows each: (staged * ncol) rows of nlat doubles, pitch apart
            j.src = stage + (size_t)half * (size_t)nvars * chunk * npitch + (size_t)v * chunk * npitch;
            j.src_pitch = (size_t)h->pitch; j.row_elems = (size_t)h->nlat; j.nrows = (size_t)staged * h->ncol;
            j.dst = raw + ((size_t)v * nraw + raw_base) * ncell;
            cp->submit(j);
        }
        raw_base += staged;
        staged = 0;
        half ^= 1;
        return e;
    };
    // Stretches that need nothing but the running sums (no raw snapshot, no seasonal snapshot, not a year's last step, not the
    // run's last step) are fused, integrate_spl steps to a launch, with the state resident on the chip
    // (miz_resident_kernel<SAVE>, miz_fused_kernel<2, ..., SAVE>; plain fused stepping when no mean is asked for): MIZ and
    // MIZ_IMEX, every geometry but two cells per thread at 768 threads.
    const bool may_fuse = h->integrate_spl > 1 && h->model == EBM_MODEL_MIZ &&
                          (!sums || ebm::has_miz_kernel(h->cfg, h->grid, ebm::OUT_LOOP_SAVE, h->imex));
    auto plain_step = [&](long long t) {
        const long long ti_ = (t - 1) % nt + 1;
        if (t >= total || ti_ == nt) return false;
        if (stage && (!lastonly || t > total - nt)) return false;
        if ((ti_ == winter_inx && (winter || hm_winter)) || (ti_ == summer_inx && (summer || hm_summer))) return false;
        return true;
    };
    for (long long tinx = 1; tinx <= total; ++tinx) {              // 1-based, as the reference
        if (may_fuse && plain_step(tinx) && plain_step(tinx + 1)) {
            long long n = 2;
            while (n < (1 << 30) && plain_step(tinx + n)) ++n;
            rc = fused_range(h, tinx - 1, clock0 + tinx - 1, (int)n, f_steps ? f_steps + (tinx - 1) : nullptr, 0, h->integrate_spl,
                             sums ? &save : nullptr);
            if (rc) { (void)cp->wait_all(); return rc; }
            tinx += n - 1;
            continue;
        }
        const long long ti = (tinx - 1) % nt + 1;
        const long long year = (tinx - 1) / nt + 1;                // ceil(st.T[tinx])
        const double f = f_steps ? f_steps[tinx - 1] : 0.0;
        // savesol!, src/infrastructure.jl:549-591, from the step kernel's registers: the annual-mean
        // sums on every step, the raw snapshot on the steps that are kept; the diagnostic FIELDS are
        // only stored on steps whose snapshot is copied out of them (seasons) and on the last one
        const bool want_raw = stage && (!lastonly || tinx > total - nt);
        const bool want_season = (ti == winter_inx && (winter || hm_winter)) || (ti == summer_inx && (summer || hm_summer));
        const int diag = (want_season || tinx == total) ? 1 : 0;
        save.stage = want_raw ? stage + (size_t)half * (size_t)nvars * chunk * npitch : nullptr;
        save.stage_offset = staged * (long long)npitch;
        rc = do_step(h, h->ttab[ti - 1], h->ttab[ti % nt], f, diag, clock0 + tinx - 1, (sums || want_raw) ? &save : nullptr);
        if (rc) { (void)cp->wait_all(); return rc; }
        if (want_raw && ++staged == chunk) EBM_TRY(flush());
        auto state_field = [&](int v) { return (const double *)h->field[fields[v]]; };
        if (want_season && (rc = ensure_natural(h))) { (void)cp->wait_all(); return rc; }
        if (ti == winter_inx) {
            if (winter) EBM_TRY(season_to_host(winter, year));
            if (hm_winter) EBM_TRY(means_to_host(hm_winter, year, state_field));
        } else if (ti == summer_inx) {
            if (summer) EBM_TRY(season_to_host(summer, year));
            if (hm_summer) EBM_TRY(means_to_host(hm_summer, year, state_field));
        } else if (ti == nt) {
            if (sums) {
                EBM_TRY(cp->wait_all());                                     // last year's means have left `mean`
                EBM_TRY(ebm::launch_finish_mean(mean, sums, (double)nt, h->ncol, nvars, (long long)npitch, h->cfg, main_stream(h)));
                if (avg) EBM_TRY(fields_to_host(avg, year, mean));
                if (hm_avg) EBM_TRY(means_to_host(hm_avg, year, [&](int v) { return (const double *)(mean + (size_t)v * npitch); }));
            }
        }
        if (sums && ti == nt && !(ti != winter_inx && ti != summer_inx))   // year ended on a seasonal index:
            EBM_TRY(hipMemsetAsync(sums, 0, sizeof(double) * npitch * (size_t)nvars, main_stream(h)));  // no mean is taken, restart sums
    }
    EBM_TRY(flush());
    EBM_TRY(hipStreamSynchronize(main_stream(h)));
    EBM_TRY(cp->wait_all());
#undef EBM_TRY
    return EBM_OK;
}

int ebm_integrate(ebm_handle_t h, int nt, int dur, const double *f_steps, int lastonly,
                  int winter_inx, int summer_inx, int nvars, const int *fields, double *raw,
                  double *winter, double *summer, double *avg) {
    return integrate_impl(h, nt, dur, f_steps, lastonly, winter_inx, summer_inx, nvars, fields, raw, winter, summer, avg,
                          nullptr, nullptr, nullptr);
}

int ebm_integrate_hemispheric(ebm_handle_t h, int nt, int dur, const double *f_steps, int winter_inx, int summer_inx,
                              int nvars, const int *fields, double *hm_winter, double *hm_summer, double *hm_avg) {
    if (!hm_winter && !hm_summer && !hm_avg) return fail(EBM_ERR_ARG, "ebm_integrate_hemispheric: no output requested");
    return integrate_impl(h, nt, dur, f_steps, 1, winter_inx, summer_inx, nvars, fields, nullptr, nullptr, nullptr, nullptr,
                          hm_winter, hm_summer, hm_avg);
}

int ebm_sync(ebm_handle_t h) {
    if (!h) return fail(EBM_ERR_ARG, "ebm_sync: null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(main_stream(h)));
    return EBM_OK;
}

int ebm_get_counters(ebm_handle_t h, long long *counters) {
    if (!h || !counters) return fail(EBM_ERR_ARG, "ebm_get_counters: null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(main_stream(h)));
    unsigned long long host[2 * ebm::kCounterShards];
    HIPCHK(hipMemcpy(host, h->counters, sizeof(host), hipMemcpyDeviceToHost));
    long long solves = 0, caps = 0;
    for (int i = 0; i < ebm::kCounterShards; ++i) {
        solves += (long long)host[2 * i];
        caps += (long long)host[2 * i + 1];
    }
    counters[0] = h->n_steps;
    counters[1] = solves;
    counters[2] = caps;
    counters[3] = h->n_launches;
    return EBM_OK;
}

int ebm_reset_counters(ebm_handle_t h) {
    if (!h) return fail(EBM_ERR_ARG, "ebm_reset_counters: null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(main_stream(h)));
    HIPCHK(hipMemset(h->counters, 0, sizeof(unsigned long long) * 2 * ebm::kCounterShards));
    h->n_steps = 0;
    h->n_launches = 0;
    return EBM_OK;
}

int ebm_timer_start(ebm_handle_t h) {
    if (!h) return fail(EBM_ERR_ARG, "ebm_timer_start: null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev0, main_stream(h)));
    return EBM_OK;
}

int ebm_timer_stop(ebm_handle_t h, float *elapsed_ms) {
    if (!h || !elapsed_ms) return fail(EBM_ERR_ARG, "ebm_timer_stop: null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev1, main_stream(h)));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return EBM_OK;
}

#ifdef EBM_STAMPS
// Diagnostic build only: allocate / fetch the phase stamps: per workgroup (ncol x 16), then per wave
// (ncol x 16 waves x 8).
int ebm_debug_stamps(ebm_handle_t h, unsigned long long *host) {
    if (!h) return fail(EBM_ERR_ARG, "ebm_debug_stamps: null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(main_stream(h)));
    const size_t nb = sizeof(unsigned long long) * (16 + 128) * (size_t)h->ncol;
    if (!h->stamps) {
        HIPCHK(hipMalloc(&h->stamps, nb));
        HIPCHK(hipMemset(h->stamps, 0, nb));
        return EBM_OK;
    }
    if (host) HIPCHK(hipMemcpy(host, h->stamps, nb, hipMemcpyDeviceToHost));
    return EBM_OK;
}
#endif

int ebm_selftest_divide(int device, int n, const double *a, const double *b, double *q) {
    if (n < 0 || !a || !b || !q) return fail(EBM_ERR_ARG, "ebm_selftest_divide: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(EBM_ERR_NO_DEVICE, "ebm_selftest_divide: no HIP device available");
    HIPCHK(hipSetDevice(device));
    double *da = nullptr, *db = nullptr, *dq = nullptr;
    const size_t nb = sizeof(double) * (size_t)n;
    HIPCHK(hipMalloc(&da, nb)); HIPCHK(hipMalloc(&db, nb)); HIPCHK(hipMalloc(&dq, nb));
    HIPCHK(hipMemcpy(da, a, nb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, nb, hipMemcpyHostToDevice));
    hipError_t e = ebm::launch_divide(da, db, dq, n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(q, dq, nb, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dq);
    if (e != hipSuccess) return fail(EBM_ERR_HIP, std::string("ebm_selftest_divide: ") + hipGetErrorString(e));
    return EBM_OK;
}

int ebm_launch_info(ebm_handle_t h, int *info) {
    if (!h || !info) return fail(EBM_ERR_ARG, "ebm_launch_info: null argument");
    info[0] = h->cfg.threads;
    info[1] = h->cfg.cells;
    info[2] = (int)h->cfg.lds_bytes;
    info[3] = h->ncol;
    return EBM_OK;
}

}  // extern "C"
