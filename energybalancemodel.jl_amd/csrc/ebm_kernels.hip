// HIP kernels (gfx950 / CDNA4, wave64) for the energy-balance time-stepping hot path.
//
// One workgroup of T threads integrates one meridian (column) for one step; thread t owns the
// C contiguous cells t*C .. t*C+C-1 ("chunk ownership") for the whole step.  A lane reads its
// 8*C contiguous bytes with 16-byte loads, so a wave covers 64*8*C contiguous bytes of the
// latitude axis per field.  Everything the reference does in ~60 temporary vectors per step
// (src/miz.jl:150-196) is fused into one kernel:
//
//   phase A  loads; water temperature; right-hand side of the T0 system (3-point stencil: the
//            chunk interior comes from registers, the two halo cells from the neighbouring
//            lanes through LDS, zero-flux at equator and pole)
//   phase B  T0 solve: active-set Newton on the piecewise-linear system of src/miz.jl:33-45.
//            Each linear system is tridiagonal and is solved per meridian by a chunk
//            partition (each thread Thomas-eliminates its C rows in registers) followed by
//            parallel cyclic reduction of the T-row interface system in LDS
//   phase D  Tbar stencil, radiative + lateral fluxes, enthalpy Euler step, redistribution,
//            floe size / thickness / concentration update, 16-byte stores
//
// Kernels in this file:
//   miz_step_kernel<C, GRID, OUT, T, IMEX>   one step per launch; OUT: state only / + diagnostics / savesol! from
//                                       registers (annual-mean sums, raw snapshots); IMEX: the implicit-diffusion
//                                       extension (one more tridiagonal solve per step, see include/ebm_hip.h)
//   miz_fused_kernel<C, GRID, T>        K steps per launch, the whole state in registers (<= 512 threads; 768 with C = 2)
//   miz_resident_kernel<GRID, T, IMEX, SAVE>  K steps per launch, the state resident in LDS (more than 512 threads; the extension;
//                                       launches of many columns at any size; SAVE: with savesol!'s sums, for ebm_integrate)
//   classic_step_kernel<C, MODE>        WE15 model: single step / savesol! / K steps per launch
//   diffusion_kernel<GRID>              the diffusion operator on its own (ebm_diffusion)
//   finish_mean, hemispheric_mean, mask_from_t0, derive_params, divide: small helpers
// C = cells per thread (4; 2 for a few short meridians), GRID = 0 identity / 1 any other grid, T =
// workgroup size as a compile-time constant.  Every one of the 443 instantiations uses 0 bytes of scratch
// (tests/tools/resource_usage.py).
//
// Arithmetic policy.  Everything outside the tridiagonal solves is a bit-exact restatement of
// the reference's expressions (IEEE division, no FMA contraction: build with
// -ffp-contract=off; the order of operations is the reference's).  The solves are free to use
// any arithmetic (explicit FMAs, v_rcp_f64 + Newton): their result is defined by the linear
// system, not by an operation order.
//
// No MFMA: there is no dense contraction on this path; it is HBM-/fp64-VALU-bound.
#include "ebm_internal.h"

// EBM_PART: csrc/Makefile compiles this file once per part, in parallel, and links the objects — the MIZ
// step kernel alone has 246 instantiations.  1 = its instantiations on the identity grid, 2 = on every other
// grid, 3 = the implicit-diffusion extension, 4 = the LDS-resident fused-K kernel, 5 = the same with savesol!'s sums,
// 0 = all other kernels and the launchers.  Undefined: one
// translation unit with everything (tests/tools/resource_usage.py, A/B builds).
#if !defined(EBM_PART) || EBM_PART == 0
#define EBM_PART_MAIN 1
#endif
#if !defined(EBM_PART) || EBM_PART == 1
#define EBM_PART_G0 1
#endif
#if !defined(EBM_PART) || EBM_PART == 2
#define EBM_PART_G1 1
#endif
#if !defined(EBM_PART) || EBM_PART == 3
#define EBM_PART_IMEX 1
#endif
#if !defined(EBM_PART) || EBM_PART == 4
#define EBM_PART_LOOP 1
#endif
#if !defined(EBM_PART) || EBM_PART == 5
#define EBM_PART_LOOPSAVE 1
#endif

namespace ebm {

// Outputs are streamed: written once per step and not read again before the next launch.  With
// the non-temporal policy they do not allocate in L2 and drain faster (0.238 -> 0.217 ms per step on
// the 4096 x 2048 workload; the same policy on the loads was slower and is not used).
#if defined(EBM_TIMING_NO_STORES)
// Timing-only A/B build (never shipped, results are garbage): the streamed output stores are dropped, what
// remains is loads + arithmetic — the ceiling a design that hid every store would reach.  The values
// are kept alive so that the arithmetic is not eliminated.
#define EBM_STORE2(ptr, v)                                                            \
    do {                                                                              \
        asm volatile("" ::"v"((v).x), "v"((v).y), "s"(ptr));                          \
    } while (0)
#elif !defined(EBM_PLAIN_STORES)
typedef double ebm_dvec2 __attribute__((ext_vector_type(2)));
#define EBM_STORE2(ptr, v)                                                            \
    do {                                                                              \
        ebm_dvec2 t_;                                                                 \
        t_.x = (v).x;                                                                 \
        t_.y = (v).y;                                                                 \
        __builtin_nontemporal_store(t_, reinterpret_cast<ebm_dvec2 *>(ptr));          \
    } while (0)
#else
#define EBM_STORE2(ptr, v) (*reinterpret_cast<double2 *>(ptr) = (v))
#endif

// Diagnostic build only (-DEBM_STAMPS): wave 0 of every workgroup records s_memtime at phase
// boundaries into a.stamps[col*16 + n].  Never enabled in the shipped library.
#ifdef EBM_STAMPS
#define EBM_STAMP(n)                                                                   \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        unsigned long long t_;                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if (threadIdx.x == 0 && a.stamps) a.stamps[(size_t)blockIdx.x * 16 + (n)] = t_; \
    } while (0)
// per-wave variant: lane 0 of every wave records into a.stamps[ncol*16 + (col*16 + wave)*8 + n]
#define EBM_STAMPW(n)                                                                  \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        unsigned long long t_;                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if ((threadIdx.x & 63) == 0 && a.stamps)                                       \
            a.stamps[(size_t)a.ncol * 16 + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (n)] = t_; \
    } while (0)
#else
#define EBM_STAMP(n) do {} while (0)
#define EBM_STAMPW(n) do {} while (0)
#endif

// The parameter block is never written by a kernel: read it through the constant address space so
// that every access is a scalar load, also after the kernel's own global stores (through a plain
// pointer hipcc falls back to per-lane vector loads once the kernel has stored anything).
typedef const __attribute__((address_space(4))) Params ConstParams;

// ---- Julia IEEE semantics ----------------------------------------------------------------
__device__ __forceinline__ double jl_min(double x, double y) {
    // Base.min(::Float64, ::Float64): NaN-propagating, -0.0 < +0.0
    double diff = x - y;
    double am = __builtin_signbit(diff) ? x : y;
    return (__builtin_isnan(x) || __builtin_isnan(y)) ? diff : am;
}
__device__ __forceinline__ double jl_clamp(double x, double lo, double hi) {
    return x > hi ? hi : (x < lo ? lo : x);
}
__device__ __forceinline__ double bool_mul(double x, bool b) {
    return b ? x : __builtin_copysign(0.0, x);   // Bool "strong zero"
}

// IEEE fp64 division for the bit-exact physics.  hipcc expands a/b to v_div_scale x2, v_rcp_f64,
// two Newton steps, a residual correction, v_div_fmas and v_div_fixup.  The two scalings and
// div_fmas only act when an operand or the quotient is near the exponent limits; for every other
// input the sequence below (the same instructions without the scaling) returns the same bits, and
// v_div_fixup still produces the IEEE results for zero, infinite and NaN operands.
// -DEBM_FULL_DIV selects the compiler's expansion instead.
__device__ __forceinline__ double div_rcp(double b) {     // refined reciprocal of the sequence
#if defined(EBM_TIMING_NO_TRANS)        // TIMING ONLY (results garbage): what the v_rcp_f64 themselves cost
    const double r0 = __builtin_bit_cast(double, 0x7FDE6238502484BAll - __builtin_bit_cast(long long, b));   // +-12 %
#else
    const double r0 = __builtin_amdgcn_rcp(b);
#endif
#if defined(EBM_TIMING_CHEAP_DIV)       // TIMING ONLY (results garbage): what all the refinement work costs
    return r0;
#else
    const double e0 = __builtin_fma(-b, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-b, r1, 1.0);
    return __builtin_fma(r1, e1, r1);
#endif
}
__device__ __forceinline__ double div_with_rcp(double a, double b, double r2) {
#if defined(EBM_FULL_DIV)
    return a / b;
#elif defined(EBM_TIMING_CHEAP_DIV)
    return a * r2;
#else
    const double q0 = a * r2;
    const double rem = __builtin_fma(-b, q0, a);
    const double q = __builtin_fma(rem, r2, q0);
    return __builtin_amdgcn_div_fixup(q, b, a);
#endif
}
__device__ __forceinline__ double ieee_div(double a, double b) {
#ifdef EBM_FULL_DIV
    return a / b;
#else
    return div_with_rcp(a, b, div_rcp(b));
#endif
}

// ---- solver arithmetic (not order-constrained) ---------------------------------------------
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// Forcing of one column at one step: the step's scalar, plus the column's constant offset, plus the
// column's own Forcing{false} schedule (src/infrastructure.jl:208-241) evaluated at the model time
// T of the step exactly as the reference does (:294-307): hold, ramp up, hold, ramp down, hold.
__device__ __forceinline__ double column_forcing(const StepArgs &a, int col, double ft, double tyear) {
    double f = a.fcol ? ft + a.fcol[col] : ft;
    if (a.fsched) {
        const double *w = a.fsched + (size_t)kSchedWords * col;     // wave-uniform: scalar loads
        const double base = w[0], peak = w[1], cool = w[2], up = w[3], down = w[4];
        const double d1 = w[5], d2 = w[6], d3 = w[7], d4 = w[8];
        double v = cool;
        if (tyear < d1) v = base;
        else if (tyear < d2) v = base + up * (tyear - d1);
        else if (tyear < d3) v = peak;
        else if (tyear < d4) v = peak + down * (tyear - d3);
        f = f + v;
    }
    return f;
}

// ---- chunk loads / stores: 8*C contiguous bytes per lane, 16-byte accesses ------------------
// `f` is a wave-uniform base (kept in SGPRs), `k0` the lane's first cell: the access compiles to
// the saddr + voffset form, so no per-lane 64-bit pointers are kept alive.
template <int C>
__device__ __forceinline__ void load_chunk(const double *__restrict__ f, unsigned k0, double (&v)[C]) {
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        double2 d = *reinterpret_cast<const double2 *>(f + (k0 + 2 * j));
        v[2 * j] = d.x;
        v[2 * j + 1] = d.y;
    }
}
template <int C>
__device__ __forceinline__ void store_chunk(double *__restrict__ f, const double (&v)[C], unsigned k0, int nlat) {
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        double2 d;
        d.x = ((int)k0 + 2 * j < nlat) ? v[2 * j] : 0.0;           // padding cells stay zero
        d.y = ((int)k0 + 2 * j + 1 < nlat) ? v[2 * j + 1] : 0.0;
        EBM_STORE2(f + (k0 + 2 * j), d);
    }
}

// Halo exchange: every thread publishes its first and last value; returns the last value of
// the previous chunk and the first value of the next chunk (0 outside the meridian).
__device__ __forceinline__ void halo_exchange(double *E0, double *E1, int t, int T, double first,
                                              double last, double &left, double &right) {
    E0[t] = first;
    E1[t] = last;
    __syncthreads();
    const int tl = t > 0 ? t - 1 : 0, tr = t + 1 < T ? t + 1 : t;
    const double l = E1[tl], r = E0[tr];
    left = t > 0 ? l : 0.0;
    right = t + 1 < T ? r : 0.0;
}

// The same exchange for the kernel whose LDS holds the state: inside a wave through the lane crossbar, between waves
// through 32 words of E (wave w: last value at E[w], first value at E[16 + w]; T <= 1024).
__device__ __forceinline__ void halo_exchange_waves(double *E, int t, int T, double first, double last,
                                                    double &left, double &right) {
    const int lane = t & 63, w = t >> 6, nw = T >> 6;
    // (ds_bpermute with the lane taken from t, not __shfl_up / __shfl_down: their own lane id is loop-invariant and
    // would be kept in a register across the caller's step loop)
    auto from_lane = [](int src_lane, double v) {
        const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
        const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
        return __hiloint2double(hi, lo);
    };
    double l = from_lane((lane + 63) & 63, last), r = from_lane((lane + 1) & 63, first);
    if (lane == 63) E[w] = last;
    if (lane == 0) E[16 + w] = first;
    __syncthreads();
    const double pl = E[w > 0 ? w - 1 : 0], nr = E[16 + (w + 1 < nw ? w + 1 : w)];
    if (lane == 0) l = w > 0 ? pl : 0.0;
    if (lane == 63) r = w + 1 < nw ? nr : 0.0;
    left = l;
    right = r;
}

// ---- tridiagonal solve of one meridian, T threads x C rows -----------------------------------
// Row k: a_k x_{k-1} + b_k x_k + c_k x_{k+1} = d_k.  Thread t owns rows t*C..t*C+C-1.
//  1. Thomas-eliminate the C-1 leading rows of the chunk with the left interface value
//     L = x_{t*C-1} carried as a parameter:  x_i = dp_i + lp_i*L - cp_i*x_{i+1}.
//  2. Collapse that to the chunk's first unknown as an affine function of (L, R = x_{t*C+C-1}).
//  3. The chunk's last row, with x_{C-2} and the next chunk's first unknown substituted, is a
//     tridiagonal system in the T interface values y_t = x_{t*C+C-1}: solve it by parallel
//     cyclic reduction (normalised rows: one reciprocal per row per level) in LDS.
//  4. Back-substitute inside the chunk.
// P0/P1: 3T doubles each.  On entry P1 must be free and P0 free after the first barrier inside;
// on exit other threads may still be reading P0/P1 (callers put a barrier before reuse).
// Rows per thread of the second level: 4 at every workgroup size.  Measured with each variant passing
// test_every_workgroup_size (tests/tools/ab_variants.sh): 2 rows 0.1688, 8 rows 0.1681 against 0.1644 ms on the
// 4096 x 2048 shape; 2 rows for T <= 128 only (the 180-band shapes): within 0.3 % of 4 rows once the benchmark's state
// is pinned (--preroll 0).  -DEBM_SECOND_LEVEL_ROWS=n (2, 4 or 8) for A/B builds.
constexpr int second_level_rows(int /*T*/) {
#ifdef EBM_SECOND_LEVEL_ROWS
    return EBM_SECOND_LEVEL_ROWS;
#else
    return 4;
#endif
}

//
// COMPACT (miz_resident_kernel, whose LDS holds the state): the same arithmetic in 4T doubles instead of 6T — P0 = 3T,
// P1 = T.  The chunk summaries go through P0 as well (one more barrier before the interface rows overwrite them), the
// second level's summaries and the interface solution through P1, the reduction's two buffers through P0.  On entry
// P0 must be free and P1 free after the first barrier inside; on exit P0 is free and P1 may still be read.
template <int C, int R, bool COMPACT = false>
__device__ __forceinline__ void partition_solve_r(const double (&a)[C], const double (&b)[C],
                                                  const double (&c)[C], const double (&d)[C],
                                                  double (&x)[C], int t, int T, double *P0,
                                                  double *P1) {
    double cp[C - 1], dp[C - 1], lp[C - 1];
    {
        const double w = fast_rcp(b[0]);
        cp[0] = c[0] * w;
        dp[0] = d[0] * w;
        lp[0] = -a[0] * w;
    }
#pragma unroll
    for (int i = 1; i < C - 1; ++i) {
        const double w = fast_rcp(__builtin_fma(-a[i], cp[i - 1], b[i]));
        cp[i] = c[i] * w;
        dp[i] = __builtin_fma(-a[i], dp[i - 1], d[i]) * w;
        lp[i] = -(a[i] * lp[i - 1]) * w;
    }
    double u = dp[C - 2], v = lp[C - 2], wr = -cp[C - 2];
#pragma unroll
    for (int i = C - 3; i >= 0; --i) {
        u = __builtin_fma(-cp[i], u, dp[i]);
        v = __builtin_fma(-cp[i], v, lp[i]);
        wr = -cp[i] * wr;
    }
    double *const W1 = COMPACT ? P0 : P1;
    W1[t] = u;
    W1[T + t] = v;
    W1[2 * T + t] = wr;
    __syncthreads();
    const bool has_next = t + 1 < T;
    const int tn = has_next ? t + 1 : t;
    double un = W1[tn], vn = W1[T + tn], wn = W1[2 * T + tn];
    un = has_next ? un : 0.0;
    vn = has_next ? vn : 0.0;
    wn = has_next ? wn : 0.0;
    double pa, pc, pd;
    {
        const double ae = a[C - 1], be = b[C - 1], ce = c[C - 1], de = d[C - 1];
        const double RA = ae * lp[C - 2];
        const double RB = __builtin_fma(ce, vn, __builtin_fma(-ae, cp[C - 2], be));
        const double RC = ce * wn;
        const double RD = __builtin_fma(-ce, un, __builtin_fma(-ae, dp[C - 2], de));
        const double rinv = fast_rcp(RB);
        pa = RA * rinv;
        pc = RC * rinv;
        pd = RD * rinv;
    }
    // Second partition level: the T interface rows (unit diagonal) are handed to the first
    // G = T/R threads, R consecutive rows each, which repeat steps 1-3 on them; only the
    // G second-level interface rows go through parallel cyclic reduction.  Waves beyond the first
    // G threads only take part in the barriers.  Rows are exchanged through LDS transposed
    // (row q of group g at [q*G + g]) so that both sides access consecutive words.
    const int G = T / R;
    const bool lvl2 = t < G;
    if (COMPACT) __syncthreads();                         // the neighbours' summaries are read: P0 takes the rows
    {
        const int q = t % R, g = t / R;
        P0[q * G + g] = pa;
        P0[T + q * G + g] = pc;
        P0[2 * T + q * G + g] = pd;
    }
    __syncthreads();
    double a2[R], c2[R], d2[R], cq[R - 1], dq[R - 1], lq[R - 1];
    // U is dead once every second-level thread has read its neighbour's entry, i.e. after the barrier that follows the
    // S0 writes: the reduction's second buffer reuses it, and P1's 3T doubles suffice for R = 2 as well (6G = 3T)
    double *U = P1, *S0 = COMPACT ? P0 : P1 + 3 * G, *S1 = COMPACT ? P0 + 3 * G : P1;
    double *const Y = COMPACT ? P1 : P0;                  // the interface solution, row q of group g at [q*G + g]
    static_assert(R == 2 || R == 4 || R == 8, "second-level rows: P1 holds 6 T / R doubles");
    static_assert(!COMPACT || R >= 4, "compact: the 3 T / R second-level summaries share P1's T doubles");
    if (lvl2) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            a2[i] = P0[i * G + t];
            c2[i] = P0[T + i * G + t];
            d2[i] = P0[2 * T + i * G + t];
        }
        cq[0] = c2[0];
        dq[0] = d2[0];
        lq[0] = -a2[0];
#pragma unroll
        for (int i = 1; i < R - 1; ++i) {
            const double w = fast_rcp(__builtin_fma(-a2[i], cq[i - 1], 1.0));
            cq[i] = c2[i] * w;
            dq[i] = __builtin_fma(-a2[i], dq[i - 1], d2[i]) * w;
            lq[i] = -(a2[i] * lq[i - 1]) * w;
        }
        double u2 = dq[R - 2], v2 = lq[R - 2], w2 = -cq[R - 2];
#pragma unroll
        for (int i = R - 3; i >= 0; --i) {
            u2 = __builtin_fma(-cq[i], u2, dq[i]);
            v2 = __builtin_fma(-cq[i], v2, lq[i]);
            w2 = -cq[i] * w2;
        }
        U[t] = u2;
        U[G + t] = v2;
        U[2 * G + t] = w2;
    }
    __syncthreads();
    double qa = 0.0, qc = 0.0, qd = 0.0;
    if (lvl2) {
        const bool nxt = t + 1 < G;
        const int gn = nxt ? t + 1 : t;
        double un2 = U[gn], vn2 = U[G + gn], wn2 = U[2 * G + gn];
        un2 = nxt ? un2 : 0.0;
        vn2 = nxt ? vn2 : 0.0;
        wn2 = nxt ? wn2 : 0.0;
        const double ae = a2[R - 1], ce = c2[R - 1], de = d2[R - 1];
        const double RA = ae * lq[R - 2];
        const double RB = __builtin_fma(ce, vn2, __builtin_fma(-ae, cq[R - 2], 1.0));
        const double RC = ce * wn2;
        const double RD = __builtin_fma(-ce, un2, __builtin_fma(-ae, dq[R - 2], de));
        const double rinv = fast_rcp(RB);
        qa = RA * rinv;
        qc = RC * rinv;
        qd = RD * rinv;
        S0[t] = qa;
        S0[G + t] = qc;
        S0[2 * G + t] = qd;
    }
    __syncthreads();
    // Out-of-range neighbours need no special case: by induction qa == 0 exactly whenever row
    // t-s does not exist (and qc == 0 when t+s does not), so reading a clamped, finite row and
    // multiplying by that zero contributes nothing.
    double *src = S0, *dst = S1;
    for (int s = 1; s < G; s <<= 1) {
        if (lvl2) {
            const int im = t - s >= 0 ? t - s : t, ip = t + s < G ? t + s : t;
            const double am = src[im], cm = src[G + im], dm = src[2 * G + im];
            const double ap = src[ip], cn = src[G + ip], dn = src[2 * G + ip];
            const double r = fast_rcp(__builtin_fma(-qc, ap, __builtin_fma(-qa, cm, 1.0)));
            const double nqd = __builtin_fma(-qc, dn, __builtin_fma(-qa, dm, qd)) * r;
            const double nqa = -(qa * am) * r;
            const double nqc = -(qc * cn) * r;
            qa = nqa;
            qc = nqc;
            qd = nqd;
            dst[t] = qa;
            dst[G + t] = qc;
            dst[2 * G + t] = qd;
        }
        __syncthreads();
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
    if (lvl2) {
        const double L2raw = src[2 * G + (t > 0 ? t - 1 : 0)];
        const double L2 = t > 0 ? L2raw : 0.0;
        double y = qd;
        Y[(R - 1) * G + t] = y;   // the level-1 rows in P0 (compact: the summaries in P1) were consumed before the barriers above
#pragma unroll
        for (int i = R - 2; i >= 0; --i) {
            y = __builtin_fma(-cq[i], y, __builtin_fma(lq[i], L2, dq[i]));
            Y[i * G + t] = y;
        }
    }
    __syncthreads();
    pd = Y[(t % R) * G + t / R];
    const int tm = t > 0 ? t - 1 : 0;
    const double Lraw = Y[(tm % R) * G + tm / R];
    const double L = t > 0 ? Lraw : 0.0;
    x[C - 1] = pd;
#pragma unroll
    for (int i = C - 2; i >= 0; --i) x[i] = __builtin_fma(-cp[i], x[i + 1], __builtin_fma(lp[i], L, dp[i]));
}
// TT: the workgroup size if it is a compile-time constant (the MIZ kernels), 0 if only known at run time (classic)
template <int C, int TT = 0, bool COMPACT = false>
__device__ __forceinline__ void partition_solve(const double (&a)[C], const double (&b)[C],
                                                const double (&c)[C], const double (&d)[C],
                                                double (&x)[C], int t, int T, double *P0, double *P1) {
    // (run-time T: one copy of the solve only — two would take the classic K-step kernel past its 128 VGPRs)
    partition_solve_r<C, second_level_rows(TT != 0 ? TT : 1024), COMPACT>(a, b, c, d, x, t, T, P0, P1);
}

// ---- MIZ pointwise physics (one cell), bit-exact restatement of src/miz.jl:160-194 ----------
struct MizCellOut {
    double q[Q_MIZ_COUNT];     // indexed by MizQuantity
};

__device__ __forceinline__ MizCellOut miz_cell_update(ConstParams &p, double f, double S, double xk,
                                                     double dif, double tb, double Ei, double Ew,
                                                     double hk, double Dk, double ph, double Tw,
                                                     double Ti) {
    const double Tm = p.Tm, Lf = p.Lf, alpha = p.alpha, dt = p.dt;
    // num, src/miz.jl:83-87
    double n = ieee_div(ph, alpha * (Dk * Dk));
    if (Dk == 0.0) n = 0.0;
    // vert_flux, src/miz.jl:96-101 (called twice in the reference with the same Tbar/diffusion)
    const double L = p.A + p.B * (tb - Tm);
    const double sol_i = 0.0 + p.ai * S;
    const double sol_w = 0.0 + (p.a0 - p.a2 * (xk * xk)) * S;
    const double Fvi = sol_i - L + dif + p.Fb + f;
    const double Fvw = sol_w - L + dif + p.Fb + f;
    // wlat :71, lat_flux :103-107
    const double wl = p.m1 * (Tw - p.Tm_pow_m2);
    double Flat = ieee_div(ph * hk * Lf * wl * M_PI, alpha * Dk);
    if (Dk == 0.0) Flat = 0.0;
    // forward Euler (:137-138,148,166-167) and redistributeE (:109-117)
    const double rEi = Ei + (ph * Fvi + Flat) * dt;
    const double rEw = Ew + ((1.0 - ph) * Fvw - Flat) * dt;
    const double cEi = jl_clamp(rEi, -INFINITY, 0.0);
    const double cEw = jl_clamp(rEw, 0.0, INFINITY);
    const double psiEidt = rEi - cEi, psiEwdt = rEw - cEw;
    double Ei_n = cEi + psiEwdt;
    const double Ew_n = cEw + psiEidt;
    // area_lead :90-93
    const double Dr = Dk + p.two_rl;
    const double ring = alpha * n * (Dr * Dr - Dk * Dk);
    const double Al = jl_min(ring, 1.0 - ph);
    // split_psiEw :120-125 applied to psiEwdt/dt (:173); the divisor is a constant of the run: its
    // refined reciprocal comes from the parameter block (same routine, same bits as ieee_div)
    const double psi = div_with_rcp(psiEwdt, dt, p.rcp_dt);
    double Ql = ieee_div(Al, 1.0 - ph) * psi;
    if (ph == 1.0) Ql = 0.0;
    const double Qp = psi - Ql;
    // psinplus :127, :174
    const double dn = dt * div_with_rcp(-Qp, p.c_dn, p.rcp_cdn);
    // D_t :140-146
    const double lat_melt = p.c_latmelt * wl;
    double lat_grow = ieee_div(-Dk, 2.0 * Lf * hk * ph) * Ql;
    const double weld = p.c_weld * ph * (Dk * Dk * Dk);
    if (hk == 0.0) lat_grow = 0.0;
    const double rD = Dk + (lat_melt + lat_grow + weld) * dt;
    // average :129-134, clamp!, zeroref! (:175-178)
    const double total = n + dn;
    const double rtotal = div_rcp(total);                 // D_n and h_n divide by the same total
    double D_n = div_with_rcp(n * rD + dn * p.Dmin, total, rtotal);
    if (total == 0.0) D_n = 0.0;
    D_n = jl_clamp(D_n, p.Dmin, p.Dmax);
    if (Ei_n == 0.0) D_n = 0.0;
    // thickness :179-181
    double rh = hk + (p.c_ht * Fvi) * dt;
    rh = jl_clamp(rh, 0.0, INFINITY);
    double h_n = div_with_rcp(n * rh + dn * p.hmin, total, rtotal);
    if (total == 0.0) h_n = 0.0;
    // concentration :74-80
    double phi_n = ieee_div(-Ei_n, Lf * h_n);
    if (h_n == 0.0) phi_n = 0.0;
    if (phi_n > 1.0) phi_n = 1.0;
    if (h_n == 0.0) Ei_n = 0.0;   // :185
    MizCellOut o;
    o.q[Q_Ei] = Ei_n;
    o.q[Q_Ew] = Ew_n;
    o.q[Q_h] = h_n;
    o.q[Q_D] = D_n;
    o.q[Q_phi] = phi_n;
    o.q[Q_n] = n;
    o.q[Q_E] = phi_n * Ei_n + (1.0 - phi_n) * Ew_n;          // :186
    o.q[Q_T] = Ti * phi_n + (1.0 - phi_n) * Tw;              // :187 (old Ti, Tw; new phi)
    o.q[Q_Ti] = (Ei_n == 0.0) ? __builtin_nan("") : Ti;      // :193
    o.q[Q_Tw] = (phi_n > 0.99) ? __builtin_nan("") : Tw;     // :194
    return o;
}

// Uniform-x operator par.D*get_diffop(nx) applied at cell k in the CSC SpMV order of
// src/infrastructure.jl:495-497 (row k accumulates columns k-1, k, k+1 in that order); tbm/tbp = T
// at k-1 / k+1, g0/g1/g2 the sub-, main and super-diagonal.
__device__ __forceinline__ double diffusion_uniform(int k, int nlat, double g0, double g1, double g2,
                                                    double tbm, double tbk, double tbp) {
    double y = 0.0;
    y = (k > 0) ? y + g0 * tbm : y;
    y = y + g1 * tbk;
    y = (k < nlat - 1) ? y + g2 * tbp : y;
    return 0.0 + y;
}

// Flux through the interface between cells kI-1 (x = xa, T = tba) and kI (x = xb, T = tbb) of the
// non-uniform stencil, src/infrastructure.jl:510-524: (1 - xx^2) dT / dx with the ghost cells
// [-x[1]; x; 2-x[end]] and dT = 0 at the two ends.  Cell k-1 computes it as (mxxph*diffT[i])/diffx[i]
// and cell k as (mxxmh*diffT[i-1])/diffx[i-1]: the same operands in the same order, hence the same
// bits — so it is evaluated once per interface instead of twice.  Also returns xx, the interface
// position (xxph of the left cell, xxmh of the right one).
__device__ __forceinline__ double interface_flux(int kI, int nlat, double xa, double xb, double tba,
                                                 double tbb, double &xx) {
    double lo_x = xa, hi_x = xb, dT = tbb - tba;
    if (kI <= 0) {             // equator: xm = -x[1], diffT[1] = 0
        lo_x = -xb;
        dT = 0.0;
    }
    if (kI >= nlat) {          // pole: xp = 2 - x[end], diffT[end] = 0
        hi_x = 2.0 - xa;
        dT = 0.0;
    }
    xx = (hi_x + lo_x) / 2.0;
    return ieee_div((1.0 - xx * xx) * dT, hi_x - lo_x);
}

// ---- pieces of the T0 system shared by the per-step and the fused-K kernels --------------------
// (one definition each, so that every kernel forms the rows with the same operations)

// water_temp (src/miz.jl:30) with the NaN -> 0 of :157
__device__ __forceinline__ double water_temperature(ConstParams &p, double Ew, double ph) {
    const double tw = p.Tm + ieee_div(Ew, (1.0 - ph) * p.cw);
    return __builtin_isnan(tw) ? 0.0 : tw;
}
// k/hp + B with hp = (h == 0 ? hmin : h), src/miz.jl:39,41,51
__device__ __forceinline__ double t0_diag_excess(ConstParams &p, double hk) {
    return __builtin_fma(p.k, fast_rcp((hk == 0.0) ? p.hmin : hk), p.B);
}
// right-hand side -(ai S - A + Dif((1-phi)(Tw-Tm)) + f), src/miz.jl:39-43: independent of the active set
__device__ __forceinline__ double t0_rhs(ConstParams &p, double S, double lo, double up, double rm,
                                         double rk, double rp, double f) {
    const double dif = __builtin_fma(up, rp - rk, lo * (rm - rk));
    return -((p.ai * S - p.A) + dif + f);
}
__device__ __forceinline__ double insolation(ConstParams &p, double xk, double ct) {
    return p.S0 - p.S1 * xk * ct - p.S2 * (xk * xk);                         // src/miz.jl:11
}

// One active-set Newton iteration (src/miz.jl:33-68): rows for the active set `smask` (bit i <=>
// T0 < Tm in cell i of this thread), tridiagonal solve, new active set; returns whether any thread's
// set changed.  P0/P1 must be free on entry; on exit every thread has passed a barrier after its last
// LDS access.
// COMPACT: P0 = 3T, P1 = T doubles (partition_solve_r); the halo goes through the lane crossbar and P1, the "any set
// changed" vote through words of P0 that nothing writes before the next barrier — no static LDS.
template <int C, int TT, bool COMPACT = false>
__device__ __forceinline__ bool newton_iteration(const double (&lo)[C], const double (&up)[C],
                                                 const double (&dd)[C], const double (&ph)[C],
                                                 const double (&rd)[C], double (&xs)[C], unsigned &smask,
                                                 int t, int T, unsigned k0, int nlat, double *P0, double *P1) {
    double g[C];
#pragma unroll
    for (int i = 0; i < C; ++i) g[i] = ((smask >> i) & 1u) ? ph[i] : 0.0;
    double gl, gr;
    if constexpr (COMPACT) halo_exchange_waves(P1, t, T, g[0], g[C - 1], gl, gr);
    else halo_exchange(P0, P0 + T, t, T, g[0], g[C - 1], gl, gr);
    double ra[C], rb[C], rc[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        ra[i] = lo[i] * (i > 0 ? g[i > 0 ? i - 1 : 0] : gl);
        rc[i] = up[i] * (i < C - 1 ? g[i < C - 1 ? i + 1 : i] : gr);
        rb[i] = -__builtin_fma(lo[i] + up[i], g[i], dd[i]);
    }
    partition_solve<C, TT, COMPACT>(ra, rb, rc, rd, xs, t, T, P0, P1);
    unsigned snew = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) snew |= (xs[i] < 0.0) ? (1u << i) : 0u;
    const int nvalid = nlat - (int)k0;                // padding rows never count as a change
    snew &= nvalid >= C ? ~0u : (nvalid > 0 ? (1u << nvalid) - 1u : 0u);
    const int changed = snew != smask;
    smask = snew;
    if constexpr (COMPACT) {
        // one word per wave in the last third of P0: free here (the reduction's buffers were read before the solve's
        // last barrier) and next written — by a solve's chunk summaries — only after a halo exchange's barrier
        static_assert(TT > 0 && TT % 64 == 0, "whole waves");
        int *const F = reinterpret_cast<int *>(P0 + 2 * T);
        const bool wave_changed = __builtin_amdgcn_ballot_w64(changed != 0) != 0;
        if ((t & 63) == 0) F[t >> 6] = wave_changed ? 1 : 0;
        __syncthreads();
        int any = 0;
#pragma unroll
        for (int w = 0; w < TT / 64; ++w) any |= F[w];
        return any != 0;
    } else {
        return __syncthreads_or(changed) != 0;
    }
}

// ---- savesol! from registers (src/infrastructure.jl:549-591) -----------------------------------
// One pair of cells (2j, 2j+1 of this thread) of every saved quantity: running sum for the annual
// mean (crossmean, src/utilities.jl:390-395: per-cell sum over the year's steps in step order) and/or
// the raw snapshot.  The sums live in a layout private to the library ("pair-split": pair j of
// thread t at col*pitch + j*2T + 2t), so that a wave's read-modify-write covers whole 128-B lines;
// finish_mean_kernel undoes it.  Snapshots use the natural layout (kp = cell index of the pair).
template <int NQ, typename Q>
__device__ __forceinline__ void save_pair(const StepArgs &a, size_t col_off, unsigned split, unsigned kp,
                                          const Q &c0, const Q &c1, bool v0, bool v1) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int v = a.var_of[q];                        // wave-uniform (kernel argument)
        if (v < 0) continue;
        const double x0 = v0 ? c0.q[q] : 0.0, x1 = v1 ? c1.q[q] : 0.0;     // padding cells stay zero
        if (a.sums) {
            double2 *sp = reinterpret_cast<double2 *>(a.sums + (size_t)v * a.sum_stride + col_off + split);
            double2 s = *sp;
            s.x = s.x + x0;
            s.y = s.y + x1;
            *sp = s;
        }
        if (a.stage) {
            double2 d;
            d.x = x0;
            d.y = x1;
            EBM_STORE2(a.stage + (size_t)v * a.stage_var_stride + a.stage_offset + col_off + kp, d);
        }
    }
}

// MIZ step: one workgroup per meridian.
//
// Geometry (choose_launch): C = 4 cells per thread, T = ceil(nlat/4 / 64)*64 <= 1024 threads (<= 128
// VGPRs).  A 4096-cell fp64 meridian fills a CU (512 KiB of VGPRs + 160 KiB of LDS): one workgroup per
// CU; shorter meridians run several workgroups per CU, which overlap each other.  Longer meridians
// do not fit one workgroup and are refused (EBM_ERR_UNSUPPORTED).
//
// LDS map (doubles; per-cell arrays hold cell i of thread t at i*T + t: lane-consecutive,
// conflict-free):
//   P0 = [0,3T), P1 = [3T,6T)    cyclic-reduction ping-pong; idle otherwise, then borrowed for
//                                the r / g / Tbar halo exchanges
//   sEw, sh, sTw = [6T, 6T+3CT)  Ew, h, Tw of every cell, parked across the T0 solve so that the
//                                solve has the register file to itself
//
// OUT (OutMode): what is written besides the prognostics (OUT_STATE, OUT_DIAG, OUT_SAVE).
// TT: workgroup size, a compile-time constant (LDS offsets become immediates).
//
// IMEX: the implicit-diffusion EXTENSION (model EBM_MODEL_MIZ_IMEX; not in the reference — defined in
// include/ebm_hip.h): before the cell updates the explicit increment of
// every cell's total enthalpy, dE = dt*(phi*Fvi + (1-phi)*Fvw), goes through one more tridiagonal solve per
// meridian, (I - (dt/cw)*Dif) dE_new = dE — the same partition + cyclic reduction as the T0 system — and the
// diffusion term of both vertical fluxes is corrected by (dE_new - dE)/dt.  Lifts the explicit limit
// dt <= cw*dx^2/(2D) of the reference's step.
// __launch_bounds__(TT, 4): the workgroup size is the compile-time TT, never more; four waves per SIMD keep the
// 128-VGPR budget that lets a 1024-thread workgroup (and four 256-thread ones) share a CU.
template <int C, int GRID, int OUT, int TT, bool IMEX>
__global__ void __launch_bounds__(TT, 4) miz_step_kernel(const StepArgs a) {
    static_assert(C == 2 || C == 4, "cells per thread");
    static_assert(OUT == OUT_STATE || OUT == OUT_DIAG || OUT == OUT_SAVE, "per-step kernel");
    constexpr bool MAYDIAG = OUT != OUT_STATE;            // diagnostic stores compiled in
    extern __shared__ double smem[];
    constexpr int T = TT;
    const int t = threadIdx.x, col = a.col0 + (int)blockIdx.x;
    const int nlat = a.nlat;
    const unsigned k0 = (unsigned)t * C;
    double *P0 = smem, *P1 = smem + 3 * T;
    double *sEw = smem + 6 * T + t, *sh = sEw + C * T, *sTw = sh + C * T;
    ConstParams &p = *reinterpret_cast<ConstParams *>(reinterpret_cast<uintptr_t>(a.p));
    const double *const gX = a.geom + G_X * a.gstride;
    double *const st = a.state + (size_t)col * (size_t)a.pitch;         // wave-uniform
    const double Tm = p.Tm;
    EBM_STAMP(0);
    EBM_STAMPW(0);                                        // per wave: first instruction
    // Warm start (src/miz.jl:47,52-54,64).  The reference carries T0 itself between steps; the
    // active-set iteration only uses its sign pattern, so between steps the library carries that
    // pattern (one bit per cell) and writes the fp64 T0 field on diagnostic launches only.
    unsigned short *const cmask = a.amask + (size_t)col * T;            // wave-uniform
    const double ct = a.sched ? a.sched[a.slot].ct : a.ct;              // per-step scalars (scalar loads)
    const double ft = a.sched ? a.sched[a.slot].ft : a.ft;
    const double f = column_forcing(a, col, ft, a.sched ? a.sched[a.slot].tyear : a.tyear);
    const bool diag = OUT == OUT_DIAG || (MAYDIAG && a.write_diag);

    // ---------------- phase A: loads, water temperature, T0-system coefficients ----------
    // Only phi and the right-hand side stay in registers across the solve; Ew, h, Tw wait in the
    // LDS stash.  The second and later Newton iterations (a changed active set: < 0.1 % of column-steps at the
    // reference's time steps, every second one with the extension's long ones) form the rows again from the stashed h
    // and the two coefficient tables and call the same solve — one copy of the solve in the kernel.
    double ph[C], rd[C], xs[C];
    unsigned smask = cmask[t];                            // active set: bit i <=> T0_i < Tm
    int it = 0;
    bool again;
    do {
        double tlo[C], tup[C], dd[C];
        // (the lane's cell index is made opaque inside the loop: hoisted out of it, the six addresses
        // would live as per-lane 64-bit pointers instead of the wave-uniform base + 32-bit offset form)
        unsigned kl = k0;
        asm volatile("" : "+v"(kl));
        load_chunk<C>(a.geom + G_LO * a.gstride, kl, tlo);
        load_chunk<C>(a.geom + G_UP * a.gstride, kl, tup);
        if (it == 0) {
            double Ew[C], hk[C], xk[C], r[C];
            load_chunk<C>(st + S_Ew * a.fstride, kl, Ew);
            load_chunk<C>(st + S_phi * a.fstride, kl, ph);
            load_chunk<C>(st + S_h * a.fstride, kl, hk);
            load_chunk<C>(gX, kl, xk);
            // Padding cells (k >= nlat) need no special case in phases A and B: their state and table
            // entries are zero, so their rows are decoupled (lo = up = 0, g = phi = 0) and finite.
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const double tw = water_temperature(p, Ew[i], ph[i]);
                sEw[i * T] = Ew[i];
                sh[i * T] = hk[i];
                sTw[i * T] = tw;
                dd[i] = t0_diag_excess(p, hk[i]);
                r[i] = (1.0 - ph[i]) * (tw - Tm);
            }
            EBM_STAMP(1);
            EBM_STAMPW(1);                                    // per wave: inputs arrived
            double rl, rr;
            halo_exchange(P0, P0 + T, t, T, r[0], r[C - 1], rl, rr);
            EBM_STAMP(2);
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const double rm = i > 0 ? r[i > 0 ? i - 1 : 0] : rl;
                const double rp = i < C - 1 ? r[i < C - 1 ? i + 1 : i] : rr;
                rd[i] = t0_rhs(p, insolation(p, xk[i], ct), tlo[i], tup[i], rm, r[i], rp, f);
            }
            __syncthreads();                                  // r halo reads done before P0 is reused
        } else {
            // second and later iterations (a changed active set): the right-hand side does not depend on the set and
            // is still in registers, like phi; only the rows' ingredients are fetched / formed again — the same values
#pragma unroll
            for (int i = 0; i < C; ++i) dd[i] = t0_diag_excess(p, sh[i * T]);
        }
        EBM_STAMP(3);
        // ---------------- phase B: active-set Newton, src/miz.jl:33-68 --------------------
        ++it;
        again = newton_iteration<C, TT>(tlo, tup, dd, ph, rd, xs, smask, t, T, k0, nlat, P0, P1);
    } while (again && it < kMaxNewton);
    if (t == 0 && a.counters) {
        unsigned long long *cnt = a.counters + 2 * (col % kCounterShards);
        atomicAdd(cnt, (unsigned long long)it);
        if (again) atomicAdd(cnt + 1, 1ull);
    }
    {
        // (the lane index is made opaque so that the mask word's per-lane 64-bit address is formed here
        // again instead of being kept — and spilled — across the solve)
        unsigned tl = (unsigned)t;
        asm volatile("" : "+v"(tl));
        cmask[tl] = (unsigned short)smask;                // new warm start, src/miz.jl:64
    }
    EBM_STAMP(6);
    EBM_STAMPW(2);                                        // per wave: phase D starts

    // ---------------- phase D: fluxes and state update ---------------------------------------
    double xk[C];
    load_chunk<C>(gX, k0, xk);
    const double xl = gX[k0 > 0 ? k0 - 1 : 0], xr = gX[k0 + C];     // zero-padded table; unused at the ends
    double g0[GRID == 0 ? C : 1], g1[GRID == 0 ? C : 1], g2[GRID == 0 ? C : 1];
    if constexpr (GRID == 0 && !IMEX) {
        // sub-, main and super-diagonal of par.D*get_diffop: on the identity grid the physics stencil
        // and the solver's plain coefficients are the same three tables (build_tables)
        load_chunk<C>(a.geom + G_LO * a.gstride, k0, g0);
        load_chunk<C>(a.geom + G_DI * a.gstride, k0, g1);
        load_chunk<C>(a.geom + G_UP * a.gstride, k0, g2);
    }
    double tb[C];
    {
        double T0[C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            T0[i] = xs[i] + Tm;                                       // new warm start, :64
            const double ti = jl_min(T0[i], Tm);                      // ice_temp, :31,65
            xs[i] = (sh[i * T] == 0.0) ? 0.0 : ti;                    // Ti: zeroref!, :66
            tb[i] = xs[i] * ph[i] + (1.0 - ph[i]) * sTw[i * T];       // Tbar, :21-26
        }
        if (MAYDIAG && diag) store_chunk<C>(st + S_T0 * a.fstride, T0, k0, nlat);
    }
    double tbl, tbr;
    halo_exchange(P0, P0 + T, t, T, tb[0], tb[C - 1], tbl, tbr);
    EBM_STAMP(7);
    EBM_STAMPW(3);                                        // per wave: Tbar halo done
    // Whole-line stores.  A lane owns 8*C contiguous bytes of every field; written pair by pair,
    // each 128-B line would reach L2 in two halves ~10^4 cycles apart and be written back to HBM
    // twice.  The first pair's new prognostics are parked in LDS words that are dead by then (cells
    // 0,1 of the stash, the idle tail of the cyclic-reduction buffers) and all 32 bytes of a lane go
    // out in two back-to-back 16-B stores once the second pair is done.
    double Fl = 0.0, xxl = 0.0;                           // flux / position of the interface left of the current cell
    if (GRID == 1) Fl = interface_flux((int)k0, nlat, xl, xk[0], tbl, tb[0], xxl);
    double difx[IMEX ? C : 1];                            // IMEX: the corrected diffusion term of every cell
    if constexpr (IMEX) {
        // explicit D d/dx[(1-x^2) dTbar/dx] of every cell (the expressions of the loop below) and the explicit
        // increment of its total enthalpy, dE = dt*(phi*Fvi + (1-phi)*Fvw)
        auto increments = [&](double (&dif)[C], double (&dE)[C], const double hl, const double hr) {
            double Fl_ = 0.0, xxl_ = 0.0;
            if constexpr (GRID == 0) {                    // the three diagonals are fetched per use, not kept across the solve
                unsigned kg = k0;
                asm volatile("" : "+v"(kg));
                load_chunk<C>(a.geom + G_LO * a.gstride, kg, g0);
                load_chunk<C>(a.geom + G_DI * a.gstride, kg, g1);
                load_chunk<C>(a.geom + G_UP * a.gstride, kg, g2);
            }
            if (GRID == 1) Fl_ = interface_flux((int)k0, nlat, xl, xk[0], hl, tb[0], xxl_);
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const int k = (int)k0 + i;
                const double tbm = i > 0 ? tb[i > 0 ? i - 1 : 0] : hl;
                const double tbp = i < C - 1 ? tb[i < C - 1 ? i + 1 : i] : hr;
                const double xp = i < C - 1 ? xk[i < C - 1 ? i + 1 : i] : xr;
                if (GRID == 0) {
                    dif[i] = diffusion_uniform(k, nlat, g0[GRID == 0 ? i : 0], g1[GRID == 0 ? i : 0],
                                               g2[GRID == 0 ? i : 0], tbm, tb[i], tbp);
                } else {
                    double xxr;
                    const double Fr = interface_flux(k + 1, nlat, xk[i], xp, tb[i], tbp, xxr);
                    dif[i] = 0.0 + ieee_div(p.D * (Fr - Fl_), xxr - xxl_);               // :524
                    Fl_ = Fr;
                    xxl_ = xxr;
                }
                const double S = insolation(p, xk[i], ct);
                const double L = p.A + p.B * (tb[i] - Tm);
                const double sol_i = 0.0 + p.ai * S;
                const double sol_w = 0.0 + (p.a0 - p.a2 * (xk[i] * xk[i])) * S;
                const double Fvi = sol_i - L + dif[i] + p.Fb + f;
                const double Fvw = sol_w - L + dif[i] + p.Fb + f;
                // padding cells (k >= nlat; on a non-uniform grid their stencil is 0/0) must not reach the solve:
                // their rows are decoupled but a NaN right-hand side would still spread through the elimination
                dE[i] = k < nlat ? (ph[i] * Fvi + (1.0 - ph[i]) * Fvw) * p.dt : 0.0;
            }
        };
        double sol[C];
        {
            // rows of I - (dt/cw)*Dif (padding rows: lo = up = 0, decoupled) and the right-hand side
            double ra[C], rb[C], rc[C], dE[C], dif[C], tlo[C], tup[C];
            load_chunk<C>(a.geom + G_LO * a.gstride, k0, tlo);
            load_chunk<C>(a.geom + G_UP * a.gstride, k0, tup);
            increments(dif, dE, tbl, tbr);
#pragma unroll
            for (int i = 0; i < C; ++i) {
                // the explicit diffusion term waits in the Tw words of the stash (Tw is formed again below from the stashed
                // Ew and phi — one division — instead of the whole stencil a second time)
                sTw[i * T] = dif[i];
                ra[i] = -(p.theta_imex * tlo[i]);
                rc[i] = -(p.theta_imex * tup[i]);
                rb[i] = 1.0 + p.theta_imex * (tlo[i] + tup[i]);
            }
            partition_solve<C, TT>(ra, rb, rc, dE, sol, t, T, P0, P1);
        }
        __syncthreads();                                  // the solve's LDS reads are done before P0 is reused
        {
            // Only Ti (xs) and the solution crossed the solve in registers: phi and x are fetched again (the lane's cell
            // index made opaque, so that the reloads are real), Tw and Tbar are formed again, the parked diffusion term comes
            // back from the stash, and the explicit increment is evaluated a second time from it — same operands, same
            // operations, same bits — for the correction (dE_new - dE)/dt.
            unsigned kl = k0;
            asm volatile("" : "+v"(kl));
            load_chunk<C>(st + S_phi * a.fstride, kl, ph);
            load_chunk<C>(gX, kl, xk);
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const int k = (int)k0 + i;
                const double dif0 = sTw[i * T];
                const double tw = water_temperature(p, sEw[i * T], ph[i]);
                sTw[i * T] = tw;                                      // the stash holds Tw again for the cell updates
                tb[i] = xs[i] * ph[i] + (1.0 - ph[i]) * tw;
                const double S = insolation(p, xk[i], ct);
                const double L = p.A + p.B * (tb[i] - Tm);
                const double sol_i = 0.0 + p.ai * S;
                const double sol_w = 0.0 + (p.a0 - p.a2 * (xk[i] * xk[i])) * S;
                const double Fvi = sol_i - L + dif0 + p.Fb + f;
                const double Fvw = sol_w - L + dif0 + p.Fb + f;
                const double dE = k < nlat ? (ph[i] * Fvi + (1.0 - ph[i]) * Fvw) * p.dt : 0.0;
                difx[IMEX ? i : 0] = dif0 + div_with_rcp(sol[i] - dE, p.dt, p.rcp_dt);
            }
        }
    }
    double *const park0 = P0 + 2 * T + t;                 // P0[2T..3T), P1[0..3T): clear of the halo words
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        MizCellOut o[2];
        __builtin_amdgcn_sched_barrier(0);
        const double2 Ei2 = *reinterpret_cast<const double2 *>(st + S_Ei * a.fstride + (k0 + 2 * j));
        const double2 Dk2 = *reinterpret_cast<const double2 *>(st + S_D * a.fstride + (k0 + 2 * j));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_sched_barrier(0);
            const int i = 2 * j + q, k = (int)k0 + i;
            const double tbm = i > 0 ? tb[i > 0 ? i - 1 : 0] : tbl;
            const double tbp = i < C - 1 ? tb[i < C - 1 ? i + 1 : i] : tbr;
            const double xp = i < C - 1 ? xk[i < C - 1 ? i + 1 : i] : xr;
            const double S = insolation(p, xk[i], ct);
            double dif;
            if constexpr (IMEX) {
                dif = difx[IMEX ? i : 0];
            } else if (GRID == 0) {
                dif = diffusion_uniform(k, nlat, g0[GRID == 0 ? i : 0], g1[GRID == 0 ? i : 0],
                                        g2[GRID == 0 ? i : 0], tbm, tb[i], tbp);
            } else {
                double xxr;
                const double Fr = interface_flux(k + 1, nlat, xk[i], xp, tb[i], tbp, xxr);
                dif = 0.0 + ieee_div(p.D * (Fr - Fl), xxr - xxl);         // :524
                Fl = Fr;
                xxl = xxr;
            }
            o[q] = miz_cell_update(p, f, S, xk[i], dif, tb[i], q ? Ei2.y : Ei2.x, sEw[i * T], sh[i * T],
                                   q ? Dk2.y : Dk2.x, ph[i], sTw[i * T], xs[i]);
            if (C == 4 && i == C - 2) {
                // L2 prefetch for the workgroup that follows this one on the XCD (column + a.prefetch):
                // one 4-byte LDS-DMA load per 32-B sector of its phase-A inputs, issued once this
                // thread's own loads have all been consumed and hidden under the last cell's
                // arithmetic.  The data lands in stash words of cell C-2 that this wave has just
                // finished with and is never read.
                // (hipcc waits vmcnt(0) at the first use of any earlier load's result while an LDS-DMA is
                // in flight: retire the one load not consumed yet before issuing it)
                asm volatile("" ::"v"(xr), "v"(o[q].q[Q_Ei]), "v"(o[q].q[Q_Ew]), "v"(o[q].q[Q_h]), "v"(o[q].q[Q_D]),
                             "v"(o[q].q[Q_phi]));
                __builtin_amdgcn_sched_barrier(0);
                if (a.prefetch > 0 && col + a.prefetch < a.ncol) {
                    const double *nxt = a.state + (size_t)(col + a.prefetch) * (size_t)a.pitch +
                                        ((unsigned)(t >> 6) * 256u + (unsigned)(t & 63) * 4u);
                    auto *sink = (__attribute__((address_space(3))) void *)(smem + 6 * T + (C - 2) * T + (t & ~63));
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nxt + S_Ew * a.fstride), sink, 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nxt + S_phi * a.fstride), sink, 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nxt + S_h * a.fstride), sink, 4, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned kp = k0 + 2 * j;
        const bool v0 = (int)kp < nlat, v1 = (int)kp + 1 < nlat;
#define EBM_PUT(slot_, qi)                                                                         \
        {                                                                                          \
            double2 d_;                                                                            \
            d_.x = v0 ? o[0].q[qi] : 0.0;                                                          \
            d_.y = v1 ? o[1].q[qi] : 0.0;                                                          \
            EBM_STORE2(st + (slot_) * a.fstride + kp, d_);                                         \
        }
        if constexpr (C == 2) {
            // two cells per thread (short meridians of latency-bound runs): the lane's 16 bytes are the
            // pair; a wave's store already covers whole lines
            EBM_PUT(S_Ei, Q_Ei) EBM_PUT(S_Ew, Q_Ew) EBM_PUT(S_h, Q_h) EBM_PUT(S_D, Q_D) EBM_PUT(S_phi, Q_phi)
        } else if (j == 0) {
            // pair 0 of Ei, Ew -> P words; h, D, phi -> stash words of cells 0, 1 (all read already)
            park0[0] = v0 ? o[0].q[Q_Ei] : 0.0;  park0[T] = v1 ? o[1].q[Q_Ei] : 0.0;
            park0[2 * T] = v0 ? o[0].q[Q_Ew] : 0.0;  park0[3 * T] = v1 ? o[1].q[Q_Ew] : 0.0;
            sEw[0] = v0 ? o[0].q[Q_h] : 0.0;  sEw[T] = v1 ? o[1].q[Q_h] : 0.0;
            sh[0] = v0 ? o[0].q[Q_D] : 0.0;  sh[T] = v1 ? o[1].q[Q_D] : 0.0;
            sTw[0] = v0 ? o[0].q[Q_phi] : 0.0;  sTw[T] = v1 ? o[1].q[Q_phi] : 0.0;
        } else {
            // the LDS-DMA prefetch must have landed before this wave can end (its LDS is released
            // with the workgroup); it was issued a whole cell update ago
            EBM_STAMPW(5);                                // per wave: arithmetic done, before the stores
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define EBM_PUT4(slot_, qi, w0, w1)                                                                \
            {                                                                                      \
                double2 a_, b_;                                                                    \
                a_.x = (w0);                                                                       \
                a_.y = (w1);                                                                       \
                b_.x = v0 ? o[0].q[qi] : 0.0;                                                      \
                b_.y = v1 ? o[1].q[qi] : 0.0;                                                      \
                EBM_STORE2(st + (slot_) * a.fstride + k0, a_);                                       \
                EBM_STORE2(st + (slot_) * a.fstride + kp, b_);                                       \
            }
            EBM_PUT4(S_Ei, Q_Ei, park0[0], park0[T])
            EBM_PUT4(S_Ew, Q_Ew, park0[2 * T], park0[3 * T])
            EBM_PUT4(S_h, Q_h, sEw[0], sEw[T])
            EBM_PUT4(S_D, Q_D, sh[0], sh[T])
            EBM_PUT4(S_phi, Q_phi, sTw[0], sTw[T])
#undef EBM_PUT4
        }
        if (MAYDIAG && diag) {
            // The diagnostic fields are outputs only: they are stored in the PAIR-SPLIT layout (pair j of thread t at
            // j*2T + 2t, the layout of the annual-mean sums), in which every store instruction of a wave covers whole
            // 128-B lines — stored in the natural layout they left pair by pair, i.e. as the two halves of each lane's
            // 32-byte sector ~10^4 cycles apart (no LDS is left to park five more fields): 0.36 ms for a diagnostic step
            // of the 4096 x 2048 shape against 0.164 state-only.  The runtime un-permutes in place before the first
            // read (unsplit_fields_kernel; ebm_ctx::diag_split).  With two cells per thread the pair IS the chunk and
            // the two layouts coincide.
            const unsigned ks = (unsigned)(j * 2 * T + 2 * t);
#define EBM_PUTP(slot_, qi)                                                                        \
            {                                                                                      \
                double2 d_;                                                                        \
                d_.x = v0 ? o[0].q[qi] : 0.0;                                                      \
                d_.y = v1 ? o[1].q[qi] : 0.0;                                                      \
                EBM_STORE2(st + (slot_) * a.fstride + ks, d_);                                     \
            }
            EBM_PUTP(S_n, Q_n) EBM_PUTP(S_E, Q_E) EBM_PUTP(S_T, Q_T) EBM_PUTP(S_Ti, Q_Ti) EBM_PUTP(S_Tw, Q_Tw)
#undef EBM_PUTP
        }
#undef EBM_PUT
        if constexpr (OUT == OUT_SAVE)
            save_pair<Q_MIZ_COUNT>(a, (size_t)col * (size_t)a.pitch, (unsigned)(j * 2 * T + 2 * t), kp, o[0], o[1], v0, v1);
        if (j < 2) EBM_STAMP(8 + j);
        if (j == 0) EBM_STAMPW(4);                        // per wave: first pair done
    }
    EBM_STAMP(15);
    EBM_STAMPW(6);                                        // per wave: stores issued
}

// Fused-K MIZ stepping for meridians of up to 4*kFusedRegThreads cells: a.nfused steps in one launch
// with the whole state (5 prognostics, the active set, the per-latitude tables) in registers between
// steps — 256 VGPRs per lane at <= 512 threads with four cells per thread, 166 at 768 threads with two
// (1025 ... 1536-cell meridians) — and LDS used only by the solve and the halo
// exchanges.  Global memory is touched at the start (state in), at the end (state out, diagnostics of
// the last step if write_diag) and by the scalar loads of the per-step table.  Every step performs the
// operations of miz_step_kernel in the same order on the same values: bit-identical results
// (tests: test_fused_run_equals_single_steps).
// SAVE (two cells per thread only — what a caller asks for to run ONE short meridian): savesol!'s running sums from every
// step of the launch, as in miz_resident_kernel<SAVE>; with two cells per thread the thread's chunk IS the pair.
template <int C, int GRID, int TT, bool SAVE = false>
__global__ void __launch_bounds__(TT) miz_fused_kernel(const StepArgs a) {
    static_assert(!SAVE || C == 2, "the savesol! variant of the register kernel exists for two cells per thread");
    static_assert(C == 2 || C == 4, "cells per thread");
    constexpr int T = TT;
    extern __shared__ double smem[];
    const int t = threadIdx.x, col = a.col0 + (int)blockIdx.x;
    const int nlat = a.nlat;
    const unsigned k0 = (unsigned)t * C;
    double *P0 = smem, *P1 = smem + 3 * T;
    ConstParams &p = *reinterpret_cast<ConstParams *>(reinterpret_cast<uintptr_t>(a.p));
    const double *const gX = a.geom + G_X * a.gstride;
    double *const st = a.state + (size_t)col * (size_t)a.pitch;
    const double Tm = p.Tm;
    unsigned short *const wmask = a.amask + (size_t)col * T + t;
    unsigned smask = *wmask;
    double Ei[C], Ew[C], hk[C], Dk[C], ph[C], xk[C], tlo[C], tup[C];
    double g1[GRID == 0 ? C : 1];
    load_chunk<C>(st + S_Ei * a.fstride, k0, Ei);
    load_chunk<C>(st + S_Ew * a.fstride, k0, Ew);
    load_chunk<C>(st + S_h * a.fstride, k0, hk);
    load_chunk<C>(st + S_D * a.fstride, k0, Dk);
    load_chunk<C>(st + S_phi * a.fstride, k0, ph);
    load_chunk<C>(gX, k0, xk);
    load_chunk<C>(a.geom + G_LO * a.gstride, k0, tlo);   // == G_0 / G_2 on the identity grid (build_tables)
    load_chunk<C>(a.geom + G_UP * a.gstride, k0, tup);
    if constexpr (GRID == 0) load_chunk<C>(a.geom + G_DI * a.gstride, k0, g1);
    const double xl = gX[k0 > 0 ? k0 - 1 : 0], xr = gX[k0 + C];
    int nit = 0, nfail = 0;
    const int nloop = a.nfused;
    for (int step = 0; step < nloop; ++step) {
        const StepSched sc = a.sched[a.slot + step];                   // scalar loads
        if constexpr (TT > 256) {
            // 256 VGPRs per lane: not enough to also keep the step-invariant stencil geometry (interface
            // positions, their reciprocals) that the compiler would hoist out of the step loop — make x
            // opaque once per step so that it is recomputed like in the per-step kernel
#pragma unroll
            for (int i = 0; i < C; ++i) asm volatile("" : "+v"(xk[i]));
        }
        const double ct = sc.ct;
        const double f = column_forcing(a, col, sc.ft, sc.tyear);
        const bool diag = a.write_diag && step == nloop - 1;
        // phase A
        double tw[C], dd[C], r[C], rd[C], xs[C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            tw[i] = water_temperature(p, Ew[i], ph[i]);
            dd[i] = t0_diag_excess(p, hk[i]);
            r[i] = (1.0 - ph[i]) * (tw[i] - Tm);
        }
        double rl, rr;
        halo_exchange(P0, P0 + T, t, T, r[0], r[C - 1], rl, rr);
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const double rm = i > 0 ? r[i > 0 ? i - 1 : 0] : rl;
            const double rp = i < C - 1 ? r[i < C - 1 ? i + 1 : i] : rr;
            rd[i] = t0_rhs(p, insolation(p, xk[i], ct), tlo[i], tup[i], rm, r[i], rp, f);
        }
        __syncthreads();
        // phase B
        int it = 0;
        bool again = true;
        while (again && it < kMaxNewton) {
            ++it;
            again = newton_iteration<C, TT>(tlo, tup, dd, ph, rd, xs, smask, t, T, k0, nlat, P0, P1);
        }
        nit += it;
        nfail += again ? 1 : 0;
        // phase D
        double tb[C];
        {
            double T0[C];
#pragma unroll
            for (int i = 0; i < C; ++i) {
                T0[i] = xs[i] + Tm;
                const double ti = jl_min(T0[i], Tm);
                xs[i] = (hk[i] == 0.0) ? 0.0 : ti;
                tb[i] = xs[i] * ph[i] + (1.0 - ph[i]) * tw[i];
            }
            if (diag) store_chunk<C>(st + S_T0 * a.fstride, T0, k0, nlat);
        }
        double tbl, tbr;
        halo_exchange(P0, P0 + T, t, T, tb[0], tb[C - 1], tbl, tbr);
        double Fl = 0.0, xxl = 0.0;
        if (GRID == 1) Fl = interface_flux((int)k0, nlat, xl, xk[0], tbl, tb[0], xxl);
        [[maybe_unused]] MizCellOut o_even;                            // SAVE: the pair's first cell waits for its second
        [[maybe_unused]] bool v_even = false;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            __builtin_amdgcn_sched_barrier(0);                         // one cell at a time: bounded live ranges
            const int k = (int)k0 + i;
            const double tbm = i > 0 ? tb[i > 0 ? i - 1 : 0] : tbl;
            const double tbp = i < C - 1 ? tb[i < C - 1 ? i + 1 : i] : tbr;
            const double xp = i < C - 1 ? xk[i < C - 1 ? i + 1 : i] : xr;
            const double S = insolation(p, xk[i], ct);
            double dif;
            if (GRID == 0) {
                dif = diffusion_uniform(k, nlat, tlo[i], g1[GRID == 0 ? i : 0], tup[i], tbm, tb[i], tbp);
            } else {
                double xxr;
                const double Fr = interface_flux(k + 1, nlat, xk[i], xp, tb[i], tbp, xxr);
                dif = 0.0 + ieee_div(p.D * (Fr - Fl), xxr - xxl);
                Fl = Fr;
                xxl = xxr;
            }
            const MizCellOut o = miz_cell_update(p, f, S, xk[i], dif, tb[i], Ei[i], Ew[i], hk[i], Dk[i], ph[i],
                                                 tw[i], xs[i]);
            const bool valid = k < nlat;                               // padding cells stay zero
            if constexpr (SAVE) {
                if (i == 0) {
                    o_even = o;
                    v_even = valid;
                } else {
                    save_pair<Q_MIZ_COUNT>(a, (size_t)col * (size_t)a.pitch, 2u * (unsigned)t, k0, o_even, o, v_even, valid);
                }
            }
            Ei[i] = valid ? o.q[Q_Ei] : 0.0;
            Ew[i] = valid ? o.q[Q_Ew] : 0.0;
            hk[i] = valid ? o.q[Q_h] : 0.0;
            Dk[i] = valid ? o.q[Q_D] : 0.0;
            ph[i] = valid ? o.q[Q_phi] : 0.0;
            if (diag) {                                                // last step of the run only
                st[S_n * a.fstride + k] = valid ? o.q[Q_n] : 0.0;
                st[S_E * a.fstride + k] = valid ? o.q[Q_E] : 0.0;
                st[S_T * a.fstride + k] = valid ? o.q[Q_T] : 0.0;
                st[S_Ti * a.fstride + k] = valid ? o.q[Q_Ti] : 0.0;
                st[S_Tw * a.fstride + k] = valid ? o.q[Q_Tw] : 0.0;
            }
        }
        __syncthreads();                                               // halo words are rewritten by the next step
    }
    store_chunk<C>(st + S_Ei * a.fstride, Ei, k0, nlat);
    store_chunk<C>(st + S_Ew * a.fstride, Ew, k0, nlat);
    store_chunk<C>(st + S_h * a.fstride, hk, k0, nlat);
    store_chunk<C>(st + S_D * a.fstride, Dk, k0, nlat);
    store_chunk<C>(st + S_phi * a.fstride, ph, k0, nlat);
    *wmask = (unsigned short)smask;
    if (t == 0 && a.counters) {
        unsigned long long *cnt = a.counters + 2 * (col % kCounterShards);
        atomicAdd(cnt, (unsigned long long)nit);
        if (nfail) atomicAdd(cnt + 1, (unsigned long long)nfail);
    }
}

// Fused-K MIZ stepping for the meridians the register kernel above cannot hold (more than kFusedRegThreads threads
// at four cells per thread: 2049 ... 4096 cells), for the implicit-diffusion extension at every size, and for launches of
// many shorter meridians (LaunchCfg::fused_in_lds: at 128 VGPRs several workgroups share a CU): a.nfused steps
// in one launch with the state RESIDENT IN LDS — Ei, Ew, h, D of every cell (cell i of thread t at i*T + t, 16 T doubles),
// phi in registers.  What the per-step kernel spends its LDS on is cut to fit beside that: the solve runs in 4 T doubles
// instead of 6 T (partition_solve_r<COMPACT>: one more barrier), the halo exchanges go through the lane crossbar and 32
// words, the "any set changed" vote through words of the solve's buffer instead of the compiler's static LDS, and Tw is
// formed again for the cell updates (one division) instead of being stashed: 20 T doubles = exactly the CU's 160 KiB at
// T = 1024.  Global memory is touched at the start (state in), at the end (state out, diagnostics of the last step if
// write_diag) and by the per-step table loads (L2 hits).  Every step performs the operations of miz_step_kernel in the
// same order on the same values: bit-identical results (tests: test_fused_run_equals_single_steps, test_every_workgroup_size).
//
// SAVE: savesol!'s annual-mean running sums taken from every step of the launch (save_pair, as in miz_step_kernel<OUT_SAVE>:
// the same per-cell sum in step order, the same bits) — what ebm_integrate launches for the stretches of a year that need
// nothing else (no raw snapshot, no seasonal snapshot).  At four waves per SIMD for every workgroup size: this variant is
// bound by the sums' read-modify-write traffic (16 B per saved variable and cell-step) and wants the occupancy.
// Up to 512 threads every variant is held to four waves per SIMD (128 VGPRs): there the kernel exists FOR its occupancy —
// several workgroups per CU fill each other's barrier stalls (0.1278 -> 0.1121 ms per step on 2048 x 4096 against the
// register kernel, 0.304 -> 0.209 on 1024 x 16384) — and is chosen for launches of many columns (LaunchCfg::fused_in_lds).
template <int GRID, int TT, bool IMEX, bool SAVE = false>
__global__ void __launch_bounds__(TT, (SAVE || TT <= 512) ? 4 : 1) miz_resident_kernel(const StepArgs a) {
    constexpr int C = 4, T = TT;
    extern __shared__ double smem[];
    const int t = threadIdx.x, col = a.col0 + (int)blockIdx.x;
    const int nlat = a.nlat;
    const unsigned k0 = (unsigned)t * C;
    double *const PA = smem, *const PB = smem + 3 * T;    // the solve's 3T + T
    // The state words: field F (Ei, Ew, h, D) of cell i of this thread at double (4 + 4F + i)*T + t.  A ds instruction
    // reaches 64 KiB from its address register: one opaque base per 64 KiB window (three at T = 1024) and compile-time
    // offsets, instead of one address register per word kept across the step loop.
    typedef __attribute__((address_space(3))) double lds_double;
    lds_double *win[3];
    // (formed again at every phase that touches the state, from that phase's own copy of the thread index: three
    // integer additions instead of three registers held across the solves)
    auto windows = [&](int tx) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            win[j] = (lds_double *)smem + (j * 8192 + tx);
            asm volatile("" : "+v"(win[j]));
        }
    };
    windows(t);
#define EBM_RES(F, i) win[((4 + 4 * (F) + (i)) * T) >> 13][((4 + 4 * (F) + (i)) * T) & 8191]
#define sEi(i) EBM_RES(0, i)
#define sEw(i) EBM_RES(1, i)
#define sh(i) EBM_RES(2, i)
#define sD(i) EBM_RES(3, i)
    ConstParams &p = *reinterpret_cast<ConstParams *>(reinterpret_cast<uintptr_t>(a.p));
    const double *const gX = a.geom + G_X * a.gstride;
    double *const st = a.state + (size_t)col * (size_t)a.pitch;
    const double Tm = p.Tm;
    unsigned short *const cmask = a.amask + (size_t)col * T;           // wave-uniform
    unsigned smask = cmask[t];
    double ph[C];
    {
        double v[C];
        load_chunk<C>(st + S_Ei * a.fstride, k0, v);
#pragma unroll
        for (int i = 0; i < C; ++i) sEi(i) = v[i];
        load_chunk<C>(st + S_Ew * a.fstride, k0, v);
#pragma unroll
        for (int i = 0; i < C; ++i) sEw(i) = v[i];
        load_chunk<C>(st + S_h * a.fstride, k0, v);
#pragma unroll
        for (int i = 0; i < C; ++i) sh(i) = v[i];
        load_chunk<C>(st + S_D * a.fstride, k0, v);
#pragma unroll
        for (int i = 0; i < C; ++i) sD(i) = v[i];
        load_chunk<C>(st + S_phi * a.fstride, k0, ph);
    }
    int nit = 0, nfail = 0;
    const int nloop = a.nfused;
    int ts = t;
    for (int step = 0; step < nloop; ++step) {
        // the step's scalars, read through the constant address space (the table is written by the host before the launch,
        // never by a kernel): scalar loads into SGPRs — through the plain pointer they would be per-lane vector loads once
        // the kernel has stored anything, and ct and f would occupy four VGPRs for the whole step
        typedef const __attribute__((address_space(4))) StepSched ConstSched;
        ConstSched &sc = *reinterpret_cast<ConstSched *>(reinterpret_cast<uintptr_t>(a.sched + (a.slot + step)));
        const double ct = sc.ct;
        // (the column's forcing is the same in every lane: moved to SGPRs — the column offset and schedule are read
        // through plain pointers, i.e. by vector loads)
        const double fv = column_forcing(a, col, sc.ft, sc.tyear);
        const double f = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(fv)),
                                          __builtin_amdgcn_readfirstlane(__double2loint(fv)));
        const bool diag = a.write_diag != 0 && step + 1 == nloop;
        // The thread index is made opaque once per step: everything derived from it — the solve's neighbour rows at every
        // level of the reduction, the transposed interface slots — is formed again in the step (a handful of integer
        // operations) instead of being hoisted out of the step loop and kept in some thirty registers.
        asm volatile("" : "+v"(ts));
        const unsigned ks = (unsigned)ts * C;
        // ---------------- phases A and B, as in miz_step_kernel ----------------
        double rd[C], xs[C];
        int it = 0;
        bool again;
        do {
            double tlo[C], tup[C], dd[C];
            // (the lane's cell index is made opaque at every group of table loads: the per-latitude tables are fetched
            // again — L2 hits — through the wave-uniform base + 32-bit offset form, instead of living in registers, or
            // their per-lane 64-bit addresses, across the solve and the steps)
            unsigned kl = ks;
            asm volatile("" : "+v"(kl));
            windows(ts);
            load_chunk<C>(a.geom + G_LO * a.gstride, kl, tlo);
            load_chunk<C>(a.geom + G_UP * a.gstride, kl, tup);
            if (it == 0) {
                double xk[C], r[C];
                load_chunk<C>(gX, kl, xk);
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    const double tw = water_temperature(p, sEw(i), ph[i]);
                    dd[i] = t0_diag_excess(p, sh(i));
                    r[i] = (1.0 - ph[i]) * (tw - Tm);
                }
                double rl, rr;
                halo_exchange_waves(PB, ts, T, r[0], r[C - 1], rl, rr);
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    const double rm = i > 0 ? r[i > 0 ? i - 1 : 0] : rl;
                    const double rp = i < C - 1 ? r[i < C - 1 ? i + 1 : i] : rr;
                    rd[i] = t0_rhs(p, insolation(p, xk[i], ct), tlo[i], tup[i], rm, r[i], rp, f);
                }
                __syncthreads();                                  // the halo words are rewritten by the iteration
            } else {
#pragma unroll
                for (int i = 0; i < C; ++i) dd[i] = t0_diag_excess(p, sh(i));
            }
            ++it;
            again = newton_iteration<C, TT, true>(tlo, tup, dd, ph, rd, xs, smask, ts, T, ks, nlat, PA, PB);
        } while (it < kMaxNewton && again);
        nit += it;
        nfail += again ? 1 : 0;
        // ---------------- phase D ----------------
        int td = ts;                                      // phase D's own copy: nothing index-derived crosses the solve
        asm volatile("" : "+v"(td));
        unsigned kl = (unsigned)td * C;
        windows(td);
        double xk[C];
        load_chunk<C>(gX, kl, xk);
        double xl = gX[kl > 0 ? kl - 1 : 0], xr = gX[kl + C];
        double g0[GRID == 0 ? C : 1], g1[GRID == 0 ? C : 1], g2[GRID == 0 ? C : 1];
        double tb[C];
        {
            double T0[C];
#pragma unroll
            for (int i = 0; i < C; ++i) {
                T0[i] = xs[i] + Tm;                                       // new warm start, :64
                const double ti = jl_min(T0[i], Tm);                      // ice_temp, :31,65
                xs[i] = (sh(i) == 0.0) ? 0.0 : ti;                    // Ti: zeroref!, :66
                const double tw = water_temperature(p, sEw(i), ph[i]);  // the value phase A formed
                tb[i] = xs[i] * ph[i] + (1.0 - ph[i]) * tw;               // Tbar, :21-26
            }
            if (diag) store_chunk<C>(st + S_T0 * a.fstride, T0, kl, nlat);
        }
        double tbl, tbr;
        halo_exchange_waves(PB, td, T, tb[0], tb[C - 1], tbl, tbr);
        double difx[IMEX ? C : 1];
        if constexpr (IMEX) {
            // the extension's second solve, as in miz_step_kernel; nothing is parked here (no LDS is left): the explicit
            // increment is evaluated a second time after the solve — same operands, same operations, same bits
            auto increments = [&](double (&dif)[C], double (&dE)[C]) {
                double Fl_ = 0.0, xxl_ = 0.0;
                if constexpr (GRID == 0) {
                    unsigned kg = kl;
                    asm volatile("" : "+v"(kg));
                    load_chunk<C>(a.geom + G_LO * a.gstride, kg, g0);
                    load_chunk<C>(a.geom + G_DI * a.gstride, kg, g1);
                    load_chunk<C>(a.geom + G_UP * a.gstride, kg, g2);
                }
                if (GRID == 1) Fl_ = interface_flux((int)kl, nlat, xl, xk[0], tbl, tb[0], xxl_);
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    const int k = (int)kl + i;
                    const double tbm = i > 0 ? tb[i > 0 ? i - 1 : 0] : tbl;
                    const double tbp = i < C - 1 ? tb[i < C - 1 ? i + 1 : i] : tbr;
                    const double xp = i < C - 1 ? xk[i < C - 1 ? i + 1 : i] : xr;
                    if (GRID == 0) {
                        dif[i] = diffusion_uniform(k, nlat, g0[GRID == 0 ? i : 0], g1[GRID == 0 ? i : 0],
                                                   g2[GRID == 0 ? i : 0], tbm, tb[i], tbp);
                    } else {
                        double xxr;
                        const double Fr = interface_flux(k + 1, nlat, xk[i], xp, tb[i], tbp, xxr);
                        dif[i] = 0.0 + ieee_div(p.D * (Fr - Fl_), xxr - xxl_);               // :524
                        Fl_ = Fr;
                        xxl_ = xxr;
                    }
                    const double S = insolation(p, xk[i], ct);
                    const double L = p.A + p.B * (tb[i] - Tm);
                    const double sol_i = 0.0 + p.ai * S;
                    const double sol_w = 0.0 + (p.a0 - p.a2 * (xk[i] * xk[i])) * S;
                    const double Fvi = sol_i - L + dif[i] + p.Fb + f;
                    const double Fvw = sol_w - L + dif[i] + p.Fb + f;
                    dE[i] = k < nlat ? (ph[i] * Fvi + (1.0 - ph[i]) * Fvw) * p.dt : 0.0;     // padding rows stay decoupled
                }
            };
            double sol[C];
            {
                double ra[C], rb[C], rc[C], dE[C], dif[C], qlo[C], qup[C];
                load_chunk<C>(a.geom + G_LO * a.gstride, kl, qlo);
                load_chunk<C>(a.geom + G_UP * a.gstride, kl, qup);
                increments(dif, dE);
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    ra[i] = -(p.theta_imex * qlo[i]);
                    rc[i] = -(p.theta_imex * qup[i]);
                    rb[i] = 1.0 + p.theta_imex * (qlo[i] + qup[i]);
                }
                partition_solve<C, TT, true>(ra, rb, rc, dE, sol, ts, T, PA, PB);
            }
            __syncthreads();                                  // the solve's last LDS reads are done
            {
                // Only Ti (xs), phi, the two halo values of Tbar and the solution crossed the solve in registers: x is
                // fetched again (the lane's cell index made opaque, so that the reloads are real), Tw and Tbar are formed
                // again from the state words
                int tq = ts;
                asm volatile("" : "+v"(tq));
                const unsigned kq = (unsigned)tq * C;
                windows(tq);
                load_chunk<C>(gX, kq, xk);
                xl = gX[kq > 0 ? kq - 1 : 0];
                xr = gX[kq + C];
                kl = kq;
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    const double tw = water_temperature(p, sEw(i), ph[i]);
                    tb[i] = xs[i] * ph[i] + (1.0 - ph[i]) * tw;
                }
                double dif[C], dE[C];
                increments(dif, dE);
#pragma unroll
                for (int i = 0; i < C; ++i) difx[IMEX ? i : 0] = dif[i] + div_with_rcp(sol[i] - dE[i], p.dt, p.rcp_dt);
            }
        }
        double Fl = 0.0, xxl = 0.0;
        if (GRID == 1) Fl = interface_flux((int)kl, nlat, xl, xk[0], tbl, tb[0], xxl);
        [[maybe_unused]] MizCellOut o_even;                           // SAVE: the pair's first cell waits for its second
        [[maybe_unused]] bool v_even = false;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            __builtin_amdgcn_sched_barrier(0);                         // one cell at a time: bounded live ranges
            const int k = (int)kl + i;
            const double tbm = i > 0 ? tb[i > 0 ? i - 1 : 0] : tbl;
            const double tbp = i < C - 1 ? tb[i < C - 1 ? i + 1 : i] : tbr;
            const double xp = i < C - 1 ? xk[i < C - 1 ? i + 1 : i] : xr;
            const double S = insolation(p, xk[i], ct);
            double dif;
            if constexpr (IMEX) {
                dif = difx[IMEX ? i : 0];
            } else if (GRID == 0) {
                // the three diagonals of the cell's pair arrive with its first cell (16-byte loads, L2 hits), not all
                // twelve words before the loop
                if ((i & 1) == 0) {
                    const double2 q0 = *reinterpret_cast<const double2 *>(a.geom + G_LO * a.gstride + (kl + i));
                    const double2 q1 = *reinterpret_cast<const double2 *>(a.geom + G_DI * a.gstride + (kl + i));
                    const double2 q2 = *reinterpret_cast<const double2 *>(a.geom + G_UP * a.gstride + (kl + i));
                    g0[GRID == 0 ? i : 0] = q0.x;  g0[GRID == 0 ? i + 1 : 0] = q0.y;
                    g1[GRID == 0 ? i : 0] = q1.x;  g1[GRID == 0 ? i + 1 : 0] = q1.y;
                    g2[GRID == 0 ? i : 0] = q2.x;  g2[GRID == 0 ? i + 1 : 0] = q2.y;
                }
                dif = diffusion_uniform(k, nlat, g0[GRID == 0 ? i : 0], g1[GRID == 0 ? i : 0], g2[GRID == 0 ? i : 0],
                                        tbm, tb[i], tbp);
            } else {
                double xxr;
                const double Fr = interface_flux(k + 1, nlat, xk[i], xp, tb[i], tbp, xxr);
                dif = 0.0 + ieee_div(p.D * (Fr - Fl), xxr - xxl);         // :524
                Fl = Fr;
                xxl = xxr;
            }
            const double tw = water_temperature(p, sEw(i), ph[i]);      // and a third time: one division, no register
            const MizCellOut o = miz_cell_update(p, f, S, xk[i], dif, tb[i], sEi(i), sEw(i), sh(i),
                                                 sD(i), ph[i], tw, xs[i]);
            const bool valid = k < nlat;                               // padding cells stay zero
            sEi(i) = valid ? o.q[Q_Ei] : 0.0;
            sEw(i) = valid ? o.q[Q_Ew] : 0.0;
            sh(i) = valid ? o.q[Q_h] : 0.0;
            sD(i) = valid ? o.q[Q_D] : 0.0;
            ph[i] = valid ? o.q[Q_phi] : 0.0;
            if (diag) {
                // last step of the run only (wave-uniform base + the per-step opaque 32-bit cell index: no per-lane
                // 64-bit addresses for the compiler to hoist out of the step loop and keep in registers)
                (st + S_n * a.fstride)[kl + i] = valid ? o.q[Q_n] : 0.0;
                (st + S_E * a.fstride)[kl + i] = valid ? o.q[Q_E] : 0.0;
                (st + S_T * a.fstride)[kl + i] = valid ? o.q[Q_T] : 0.0;
                (st + S_Ti * a.fstride)[kl + i] = valid ? o.q[Q_Ti] : 0.0;
                (st + S_Tw * a.fstride)[kl + i] = valid ? o.q[Q_Tw] : 0.0;
            }
            if constexpr (SAVE) {
                if ((i & 1) == 0) {
                    o_even = o;
                    v_even = valid;
                } else {
                    save_pair<Q_MIZ_COUNT>(a, (size_t)col * (size_t)a.pitch, (unsigned)((i / 2) * 2 * T) + 2u * (unsigned)td,
                                           kl + (unsigned)(i - 1), o_even, o, v_even, valid);
                }
            }
        }
        __syncthreads();                                               // the halo words are rewritten by the next step
    }
    {
        // (indices made opaque: the addresses are formed here, not kept — and spilled — across the step loop)
        unsigned tl = (unsigned)ts;
        asm volatile("" : "+v"(tl));
        const unsigned ke = tl * C;
        windows((int)tl);
        double v[C];
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = sEi(i);
        store_chunk<C>(st + S_Ei * a.fstride, v, ke, nlat);
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = sEw(i);
        store_chunk<C>(st + S_Ew * a.fstride, v, ke, nlat);
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = sh(i);
        store_chunk<C>(st + S_h * a.fstride, v, ke, nlat);
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = sD(i);
        store_chunk<C>(st + S_D * a.fstride, v, ke, nlat);
        store_chunk<C>(st + S_phi * a.fstride, ph, ke, nlat);
        cmask[tl] = (unsigned short)smask;
    }
    if (ts == 0 && a.counters) {
        unsigned long long *cnt = a.counters + 2 * (col % kCounterShards);
        atomicAdd(cnt, (unsigned long long)nit);
        if (nfail) atomicAdd(cnt + 1, (unsigned long long)nfail);
    }
}
#undef sEi
#undef sEw
#undef sh
#undef sD
#undef EBM_RES

// ---- classic (WE15) step, src/classic.jl:37-71 ------------------------------------------------
// MODE: OUT_STATE (T, h written if write_diag), OUT_SAVE (savesol! from registers) or OUT_LOOP
// (a.nfused steps per launch, E and Tg in registers between steps).
struct ClassicCellOut {
    double q[QC_COUNT];
};
template <int C, int MODE>
__global__ void __launch_bounds__(1024) classic_step_kernel(const StepArgs a) {
    static_assert(C == 2 || C == 4, "cells per thread");
    constexpr bool LOOP = MODE == OUT_LOOP;
    extern __shared__ double smem[];
    const int T = blockDim.x, t = threadIdx.x, col = a.col0 + (int)blockIdx.x;
    const int nlat = a.nlat;
    const unsigned k0 = (unsigned)t * C;
    double *P0 = smem, *P1 = smem + 3 * T;
    ConstParams &p = *reinterpret_cast<ConstParams *>(reinterpret_cast<uintptr_t>(a.p));
    double *const st = a.state + (size_t)col * (size_t)a.pitch;          // wave-uniform
    double E[C], Tg[C];
    load_chunk<C>(st + C_E * a.fstride, k0, E);
    load_chunk<C>(st + C_Tg * a.fstride, k0, Tg);
    const int nloop = LOOP ? a.nfused : 1;
    for (int step = 0; step < nloop; ++step) {
        // per-latitude statics (get_statics, src/classic.jl:18-29): re-read every step (L2 hits) rather
        // than kept in 48 registers across the fused loop
        const double *ge = a.geom;
        if constexpr (LOOP) asm volatile("" : "+s"(ge));
        double xk[C], aw[C], Sb[C], kd[C], ca[C], cc[C];
        load_chunk<C>(ge + G_X * a.gstride, k0, xk);
        load_chunk<C>(ge + G_AW * a.gstride, k0, aw);
        load_chunk<C>(ge + G_SB * a.gstride, k0, Sb);
        load_chunk<C>(ge + G_KDIAG * a.gstride, k0, kd);
        load_chunk<C>(ge + G_KSUB * a.gstride, k0, ca);   // off-diagonals of kappa
        load_chunk<C>(ge + G_KSUP * a.gstride, k0, cc);
        const int slot = a.slot + step;
        const double ct = a.sched ? a.sched[slot].ct : a.ct;
        const double ct_next = a.sched ? a.sched[slot].ct_next : a.ct_next;
        const double ft = a.sched ? a.sched[slot].ft : a.ft;
        const double f = column_forcing(a, col, ft, a.sched ? a.sched[slot].tyear : a.tyear);
        double b[C], d[C], oT[C], oh[C], xs[C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const bool valid = (int)k0 + i < nlat;
            double Ek = E[i];
            const double S_i = Sb[i] - (p.S1 * ct) * xk[i];                            // :23-24
            const double S_ip1 = Sb[i] - (p.S1 * ct_next) * xk[i];
            const double alpha = bool_mul(aw[i], Ek > 0.0) + bool_mul(p.ai, Ek < 0.0); // :47
            const double Cc = alpha * S_i + p.cg_tau * Tg[i] - p.A + f;                // :48
            const double T0 = ieee_div(Cc, p.M - ieee_div(p.kLf, Ek));                 // :50
            const double Tk = bool_mul(ieee_div(Ek, p.cw), Ek >= 0.0) + bool_mul(bool_mul(T0, Ek < 0.0), T0 < 0.0);
            Ek = Ek + p.dt * (Cc - p.M * Tk + p.Fb);                                   // :53
            const double den = p.M - ieee_div(p.kLf, Ek);
            const double q = bool_mul(bool_mul(ieee_div(p.dc, den), T0 < 0.0), Ek < 0.0);       // :56
            const double rhs = Tg[i] + p.dt_tau * (bool_mul(ieee_div(Ek, p.cw), Ek >= 0.0) +
                               bool_mul(bool_mul(ieee_div(p.ai * S_ip1 - p.A + f, den), T0 < 0.0), Ek < 0.0));
            b[i] = valid ? kd[i] - q : 1.0;
            d[i] = valid ? rhs : 0.0;
            E[i] = valid ? Ek : 0.0;                                                   // padding cells stay zero
            oT[i] = Tk;
            oh[i] = bool_mul(ieee_div(-Ek, p.Lf), Ek < 0.0);                           // :65
        }
        const bool last = step == nloop - 1;
        if (last) store_chunk<C>(st + C_E * a.fstride, E, k0, nlat);
        if (a.write_diag && last) {
            store_chunk<C>(st + C_T * a.fstride, oT, k0, nlat);
            store_chunk<C>(st + C_h * a.fstride, oh, k0, nlat);
        }
        partition_solve<C>(ca, b, cc, d, xs, t, T, P0, P1);   // Implicit Euler for Tg, :55-63
#pragma unroll
        for (int i = 0; i < C; ++i) Tg[i] = ((int)k0 + i < nlat) ? xs[i] : 0.0;
        if (last) store_chunk<C>(st + C_Tg * a.fstride, Tg, k0, nlat);
        if constexpr (MODE == OUT_SAVE) {
#pragma unroll
            for (int j = 0; j < C / 2; ++j) {
                ClassicCellOut c0, c1;
                c0.q[QC_E] = E[2 * j];  c0.q[QC_Tg] = Tg[2 * j];  c0.q[QC_T] = oT[2 * j];  c0.q[QC_h] = oh[2 * j];
                c1.q[QC_E] = E[2 * j + 1];  c1.q[QC_Tg] = Tg[2 * j + 1];  c1.q[QC_T] = oT[2 * j + 1];  c1.q[QC_h] = oh[2 * j + 1];
                const unsigned kp = k0 + 2 * j;
                save_pair<QC_COUNT>(a, (size_t)col * (size_t)a.pitch, (unsigned)(j * 2 * T + 2 * t), kp, c0, c1,
                                    (int)kp < nlat, (int)kp + 1 < nlat);
            }
        }
        // (no barrier between the solves of consecutive steps: after a solve's last barrier the threads only READ P0; the
        // next solve first writes each thread's own words of P1 — which nobody reads after that last barrier — and passes
        // a barrier of its own before anything is written to P0)
    }
}

#ifdef EBM_PART_MAIN
// Active set of a T0 field (after ebm_set_field(T0)): bit i of amask[col][t] <=> T0 < Tm in cell t*C+i.
__global__ void mask_from_t0_kernel(const StepArgs a, int C) {
    const int T = blockDim.x, t = threadIdx.x, col = blockIdx.x;
    const double Tm = a.p->Tm;
    const double *T0 = a.state + S_T0 * a.fstride + (size_t)col * (size_t)a.pitch + (size_t)t * C;
    unsigned m = 0;
    for (int i = 0; i < C; ++i)
        if (t * C + i < a.nlat && T0[i] < Tm) m |= 1u << i;
    a.amask[(size_t)col * T + t] = (unsigned short)m;
}

// rcp_dt / rcp_cdn of the parameter block, with the device's own refinement sequence (see Params)
__global__ void derive_params_kernel(Params *p) {
    p->rcp_dt = div_rcp(p->dt);
    p->rcp_cdn = div_rcp(p->c_dn);
}
hipError_t launch_derive_params(Params *p_dev, hipStream_t s) {
    derive_params_kernel<<<1, 1, 0, s>>>(p_dev);
    return hipGetLastError();
}

// Self-test hook (ebm_selftest_divide): q[i] = ieee_div(a[i], b[i]) with the device routine the
// physics uses, so that tests can compare it bit for bit with host IEEE division.
__global__ void divide_kernel(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ q, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) q[i] = ieee_div(a[i], b[i]);
}
hipError_t launch_divide(const double *a, const double *b, double *q, int n, hipStream_t s) {
    divide_kernel<<<(n + 255) / 256, 256, 0, s>>>(a, b, q, n);
    return hipGetLastError();
}

// hemispheric_mean (src/utilities.jl:397-403) of one field, one workgroup per column:
//   int = 0; for i in 1:nx-1: int += (vec[i]+vec[i+1]) * (x[i+1]-x[i]) / 2.0
// The terms are formed in parallel (elementwise, exact order of operations); the accumulation is
// the reference's strictly sequential left-to-right sum, done by one lane out of LDS, so the
// result is bit-identical to the reference's loop.
__global__ void hemispheric_mean_kernel(const double *__restrict__ field, const double *__restrict__ x,
                                        int pitch, int nlat, double *__restrict__ out) {
    extern __shared__ double terms[];
    const double *v = field + (size_t)blockIdx.x * pitch;
    for (int i = threadIdx.x; i < nlat - 1; i += blockDim.x)
        terms[i] = ieee_div((v[i] + v[i + 1]) * (x[i + 1] - x[i]), 2.0);
    __syncthreads();
    if (threadIdx.x == 0) {
        double acc = 0.0;
        for (int i = 0; i < nlat - 1; ++i) acc = acc + terms[i];
        out[blockIdx.x] = acc;
    }
}
// The diffusion operator on its own: out = base + D d/dx[(1-x^2) d temp/dx], one thread per cell —
// diffusion!(base, temp, st, par) / diffusion(T, st, par), src/infrastructure.jl:495-533, with the
// same device functions (and hence the same bits) the step kernels use inside their fused physics.
template <int GRID>
__global__ void diffusion_kernel(const double *__restrict__ temp, const double *__restrict__ base,
                                 double *__restrict__ out, const double *__restrict__ geom, long long gstride,
                                 const Params *__restrict__ pp, int pitch, int nlat) {
    const int col = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nlat) return;
    const double *T = temp + (size_t)col * pitch;
    const double tk = T[k], tm = k > 0 ? T[k - 1] : 0.0, tp = k < nlat - 1 ? T[k + 1] : 0.0;
    double term;
    if (GRID == 0) {
        term = diffusion_uniform(k, nlat, geom[G_LO * gstride + k], geom[G_DI * gstride + k], geom[G_UP * gstride + k],
                                 tm, tk, tp);
    } else {
        const double *x = geom + G_X * gstride;
        const double xk = x[k], xm = k > 0 ? x[k - 1] : 0.0, xp = k < nlat - 1 ? x[k + 1] : 0.0;
        double xxl, xxr;
        const double Fl = interface_flux(k, nlat, xm, xk, tm, tk, xxl);
        const double Fr = interface_flux(k + 1, nlat, xk, xp, tk, tp, xxr);
        term = ieee_div(pp->D * (Fr - Fl), xxr - xxl);                  // :524
    }
    out[(size_t)col * pitch + k] = (base ? base[(size_t)col * pitch + k] : 0.0) + term;
}
hipError_t launch_diffusion(const double *temp, const double *base, double *out, const double *geom, long long gstride,
                            const Params *p, int grid_kind, int pitch, int nlat, int ncol, hipStream_t s) {
    dim3 grid((nlat + 255) / 256, ncol), block(256);
    if (grid_kind == 0) diffusion_kernel<0><<<grid, block, 0, s>>>(temp, base, out, geom, gstride, p, pitch, nlat);
    else diffusion_kernel<1><<<grid, block, 0, s>>>(temp, base, out, geom, gstride, p, pitch, nlat);
    return hipGetLastError();
}

// annual_mean (src/infrastructure.jl:536-544, crossmean src/utilities.jl:390-395): sum / nt, from
// the pair-split layout of save_pair to the natural [col][pitch] one; the sum restarts at zero.  blockIdx.y = saved
// variable (one launch for all of them): dst / sum advance by var_stride per variable.
__global__ void finish_mean_kernel(double *__restrict__ dst, double *__restrict__ sum, double nt, int threads,
                                   int cells, long long var_stride) {
    const int t = threadIdx.x, col = blockIdx.x;
    const size_t base = (size_t)blockIdx.y * (size_t)var_stride + (size_t)col * (size_t)threads * cells;
    for (int j = 0; j < cells / 2; ++j) {
        double2 *sp = reinterpret_cast<double2 *>(sum + base + (size_t)(j * 2 * threads + 2 * t));
        const double2 s = *sp;
        double2 m;
        m.x = s.x / nt;
        m.y = s.y / nt;
        *reinterpret_cast<double2 *>(dst + base + (size_t)(t * cells + 2 * j)) = m;
        double2 z;
        z.x = 0.0;
        z.y = 0.0;
        *sp = z;
    }
}

// Pair-split -> natural layout, in place: the diagnostic fields as the 4-cells-per-thread step kernels store them
// (pair j of thread t at j*2T + 2t) become [col][pitch] with cell k at k.  One workgroup per column holds the whole
// column in registers across a barrier, so the permutation needs no second buffer.  blockIdx.y = field: `fields`
// advances by field_stride per field.
__global__ void unsplit_fields_kernel(double *__restrict__ fields, long long field_stride, int threads) {
    const int t = threadIdx.x;
    double *f = fields + (size_t)blockIdx.y * (size_t)field_stride + (size_t)blockIdx.x * (size_t)threads * 4;
    const double2 p0 = *reinterpret_cast<const double2 *>(f + 2 * t);
    const double2 p1 = *reinterpret_cast<const double2 *>(f + 2 * threads + 2 * t);
    __syncthreads();
    *reinterpret_cast<double2 *>(f + 4 * t) = p0;
    *reinterpret_cast<double2 *>(f + 4 * t + 2) = p1;
}
// natural -> pair-split, the inverse (a field the caller set, about to be read by a kernel that expects the split layout)
__global__ void split_fields_kernel(double *__restrict__ fields, long long field_stride, int threads) {
    const int t = threadIdx.x;
    double *f = fields + (size_t)blockIdx.y * (size_t)field_stride + (size_t)blockIdx.x * (size_t)threads * 4;
    const double2 p0 = *reinterpret_cast<const double2 *>(f + 4 * t);
    const double2 p1 = *reinterpret_cast<const double2 *>(f + 4 * t + 2);
    __syncthreads();
    *reinterpret_cast<double2 *>(f + 2 * t) = p0;
    *reinterpret_cast<double2 *>(f + 2 * threads + 2 * t) = p1;
}
hipError_t launch_split_fields(double *fields, long long field_stride, int nfields, int ncol, const LaunchCfg &cfg,
                               hipStream_t s) {
    if (cfg.cells != 4) return hipSuccess;
    split_fields_kernel<<<dim3(ncol, nfields), cfg.threads, 0, s>>>(fields, field_stride, cfg.threads);
    return hipGetLastError();
}

// ---- the zonal diffusion operator as a backward-Euler substep (ebm_zonal_diffusion: an extension, defined in include/ebm_hip.h) ----
// One lane per (member, latitude): the nlon unknowns of a latitude circle are walked sequentially, so that every access
// of a wave is along the contiguous latitude axis (in the pair-split index space: lane p of the row of longitude l reads
// T[(member*nlon + l)*pitch + p]).  Periodic tridiagonal system B U_l - a (U_{l-1} + U_{l+1}) = b_l, B = 1 + 2a, by Thomas
// elimination that carries the coupling to the LAST unknown W = U_{n-1} along (no Sherman-Morrison second solve):
//     U_l = cp_l U_{l+1} + ep_l W + dp_l,     cp_l = a m_l,  ep_l = a ep_{l-1} m_l,  m_l = 1/(B - a cp_{l-1}),
//     dp_l = (b_l + a dp_{l-1}) m_l            (the only data-dependent recurrence: one fma and one multiply per unknown)
// while the last row is reduced alongside: its coefficient on U_l is f_l = -a ep_{l-1} (f_0 = -a), its right-hand side
// collects R = -sum f_l dp_l; W = (b_{n-1} + R - (f_{n-2} - a) dp_{n-2}) zW with zW the reciprocal of the reduced
// diagonal.  m_l, ep_l and zW depend on (latitude, l) only and come from tables built at ebm_create (zM, zE, zW); a = za.
// Forward sweep: dp_l is parked in the output array; backward sweep: U_l, then Z_l = (U_l - b_l) cw/dt over it.
// Free arithmetic (fma): the result is defined by the linear system.  UNR rows are loaded ahead of the recurrence.
#ifndef EBM_ZONAL_UNR
#define EBM_ZONAL_UNR 16         // rows of loads ahead of the recurrence (measured: 8 rows 203.2 us, 16 rows 194.2 us on 1024 x 512 x 32; -DEBM_ZONAL_UNR=n for A/B builds)
#endif
template <int UNR>
__global__ void __launch_bounds__(256) zonal_sweep_kernel(const double *__restrict__ T, double *__restrict__ out_Z,
                                                          double *__restrict__ out_U, const double *__restrict__ zM,
                                                          const double *__restrict__ zE, const double *__restrict__ za,
                                                          const double *__restrict__ zW, int nlon, int pitch, double rtheta) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pitch) return;
    const size_t row0 = (size_t)blockIdx.y * (size_t)nlon * (size_t)pitch + (size_t)p;
    const double *b = T + row0;
    double *d = out_Z + row0;
    double *u = out_U ? out_U + row0 : nullptr;
    const double *M = zM + p, *E = zE + p;
    const double a = za[p];
    const size_t P = (size_t)pitch;
    const int n = nlon;
    // ---- forward: l = 0 .. n-2 ----
    double dp = b[0] * M[0];
    double R = 0.0, f = -a;
    int l = 0;
    for (; l + UNR <= n - 2; l += UNR) {           // rows l+1 .. l+UNR are needed: all <= n-2
        double bb[UNR], mm[UNR], ee[UNR];
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            bb[i] = b[(size_t)(l + 1 + i) * P];
            mm[i] = M[(size_t)(l + 1 + i) * P];
            ee[i] = E[(size_t)(l + i) * P];
        }
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            d[(size_t)(l + i) * P] = dp;
            R = __builtin_fma(-f, dp, R);
            f = -a * ee[i];
            dp = __builtin_fma(a, dp, bb[i]) * mm[i];
        }
    }
    for (; l < n - 2; ++l) {
        d[(size_t)l * P] = dp;
        R = __builtin_fma(-f, dp, R);
        f = -a * E[(size_t)l * P];
        dp = __builtin_fma(a, dp, b[(size_t)(l + 1) * P]) * M[(size_t)(l + 1) * P];
    }
    // dp = dp_{n-2}, f = f_{n-2}
    const double bl = b[(size_t)(n - 1) * P];
    const double W = (bl + R - (f - a) * dp) * zW[p];
    d[(size_t)(n - 1) * P] = (W - bl) * rtheta;
    if (u) u[(size_t)(n - 1) * P] = W;
    // ---- backward: l = n-2 .. 0 (dp_{n-2} is still in the register) ----
    double Un = W;
    l = n - 2;
    {
        const double U = __builtin_fma(a * M[(size_t)l * P], Un, __builtin_fma(E[(size_t)l * P], W, dp));
        d[(size_t)l * P] = (U - b[(size_t)l * P]) * rtheta;
        if (u) u[(size_t)l * P] = U;
        Un = U;
        --l;
    }
    for (; l - UNR + 1 >= 0; l -= UNR) {
        double bb[UNR], mm[UNR], ee[UNR], dd[UNR];
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            bb[i] = b[(size_t)(l - i) * P];
            mm[i] = M[(size_t)(l - i) * P];
            ee[i] = E[(size_t)(l - i) * P];
            dd[i] = d[(size_t)(l - i) * P];
        }
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const double U = __builtin_fma(a * mm[i], Un, __builtin_fma(ee[i], W, dd[i]));
            d[(size_t)(l - i) * P] = (U - bb[i]) * rtheta;
            if (u) u[(size_t)(l - i) * P] = U;
            Un = U;
        }
    }
    for (; l >= 0; --l) {
        const double U = __builtin_fma(a * M[(size_t)l * P], Un, __builtin_fma(E[(size_t)l * P], W, d[(size_t)l * P]));
        d[(size_t)l * P] = (U - b[(size_t)l * P]) * rtheta;
        if (u) u[(size_t)l * P] = U;
        Un = U;
    }
}
hipError_t launch_zonal_sweep(const double *T, double *out_Z, double *out_U, const double *zM, const double *zE,
                              const double *za, const double *zW, int nlon, int nmember, int pitch, double rtheta,
                              hipStream_t s) {
    // one wave per workgroup: as many workgroups as the (few) lanes of this kernel allow
    dim3 grid((pitch + 63) / 64, nmember), block(64);
    zonal_sweep_kernel<EBM_ZONAL_UNR><<<grid, block, 0, s>>>(T, out_Z, out_U, zM, zE, za, zW, nlon, pitch, rtheta);
    return hipGetLastError();
}

// ---- the same periodic systems, partitioned along the circle ----------------------------------------------------------
// One lane per (member, latitude) gives nlat x nmember lanes: 64 waves for a single 4096 x 2048 grid.  For circles of 256
// longitudes and more the circle is cut into S segments of m = nlon/S unknowns (S a function of nlon only), each walked by
// its own lane — the column solve's partition, across longitude:
//   1. zonal_seg_forward: inside segment s, rows 0 .. m-2 are eliminated as above with the segment's LEFT neighbour
//      L = y_{s-1} in the role of the wrap unknown:  U_i = dp_i + ep_i L + cp_i U_{i+1}; dp_i is parked, and
//      u_s = sum_i P_i dp_i (P_i = cp_0 ... cp_{i-1}: the segment's first unknown for L = y_s = 0) and
//      g_s = b_{m-1} + a dp_{m-2} are kept per segment;
//   2. zonal_reduced_solve: the segments' last unknowns y_s obey a periodic tridiagonal system of size S with CONSTANT
//      coefficients again,  -a'' y_{s-1} + B'' y_s - a'' y_{s+1} = g_s + a u_{s+1},   a'' = a ep_{m-2},
//      B'' = B - a cp_{m-2} - a alpha,  alpha = sum_i P_i ep_i  — solved per (member, latitude) by the same elimination;
//   3. zonal_seg_backward: back-substitution inside every segment from y_s and y_{s-1}, and Z.
// Tables: the chain's m_i, ep_i for ONE segment ([m-1][pitch], cache resident whatever nlon is), the reduced system's
// ([S-1][pitch]) and a, a'', the reciprocal of the reduced last diagonal per latitude.
template <int UNR>
__global__ void __launch_bounds__(256) zonal_seg_forward_kernel(const double *__restrict__ T, double *__restrict__ out_Z,
                                                                const double *__restrict__ cM, const double *__restrict__ cE,
                                                                const double *__restrict__ za, double *__restrict__ su,
                                                                double *__restrict__ sg, int nlon, int S, int pitch) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pitch) return;
    const int seg = blockIdx.y % S, member = blockIdx.y / S, m = nlon / S;
    const size_t P = (size_t)pitch;
    const size_t row0 = ((size_t)member * (size_t)nlon + (size_t)seg * (size_t)m) * P + (size_t)p;
    const double *b = T + row0;
    double *d = out_Z + row0;
    const double *M = cM + p, *E = cE + p;
    const double a = za[p];
    double dp = b[0] * M[0];
    double u = dp;                                   // P_0 = 1
    int i = 0;
    for (; i + UNR <= m - 2; i += UNR) {             // rows i+1 .. i+UNR <= m-2
        double bb[UNR], mm[UNR], ee[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            bb[k] = b[(size_t)(i + 1 + k) * P];
            mm[k] = M[(size_t)(i + 1 + k) * P];
            ee[k] = E[(size_t)(i + k) * P];          // P_{i+1+k} = ep_{i+k}
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            d[(size_t)(i + k) * P] = dp;
            dp = __builtin_fma(a, dp, bb[k]) * mm[k];
            u = __builtin_fma(ee[k], dp, u);
        }
    }
    for (; i < m - 2; ++i) {
        d[(size_t)i * P] = dp;
        dp = __builtin_fma(a, dp, b[(size_t)(i + 1) * P]) * M[(size_t)(i + 1) * P];
        u = __builtin_fma(E[(size_t)i * P], dp, u);
    }
    d[(size_t)(m - 2) * P] = dp;                     // dp_{m-2}
    const size_t so = ((size_t)member * (size_t)S + (size_t)seg) * P + (size_t)p;
    su[so] = u;
    sg[so] = __builtin_fma(a, dp, b[(size_t)(m - 1) * P]);
}

// the reduced periodic system of the S segment ends, per (member, latitude); sy holds dp on the way and y at the end
__global__ void __launch_bounds__(256) zonal_reduced_solve_kernel(const double *__restrict__ su, const double *__restrict__ sg,
                                                                  double *__restrict__ sy, const double *__restrict__ rM,
                                                                  const double *__restrict__ rE, const double *__restrict__ za,
                                                                  const double *__restrict__ za2, const double *__restrict__ rW,
                                                                  int S, int pitch) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pitch) return;
    const size_t P = (size_t)pitch;
    const size_t o = (size_t)blockIdx.y * (size_t)S * P + (size_t)p;
    const double a = za[p], a2 = za2[p];
    auto rhs = [&](int s_) { return __builtin_fma(a, su[o + (size_t)((s_ + 1) % S) * P], sg[o + (size_t)s_ * P]); };
    const double *M = rM + p, *E = rE + p;
    double dp = rhs(0) * M[0];
    double R = 0.0, f = -a2;
    for (int l = 0; l < S - 2; ++l) {
        sy[o + (size_t)l * P] = dp;
        R = __builtin_fma(-f, dp, R);
        f = -a2 * E[(size_t)l * P];
        dp = __builtin_fma(a2, dp, rhs(l + 1)) * M[(size_t)(l + 1) * P];
    }
    const double W = (rhs(S - 1) + R - (f - a2) * dp) * rW[p];
    sy[o + (size_t)(S - 1) * P] = W;
    double Un = W;
    {
        const int l = S - 2;
        const double U = __builtin_fma(a2 * M[(size_t)l * P], Un, __builtin_fma(E[(size_t)l * P], W, dp));
        sy[o + (size_t)l * P] = U;
        Un = U;
    }
    for (int l = S - 3; l >= 0; --l) {
        const double U = __builtin_fma(a2 * M[(size_t)l * P], Un, __builtin_fma(E[(size_t)l * P], W, sy[o + (size_t)l * P]));
        sy[o + (size_t)l * P] = U;
        Un = U;
    }
}

template <int UNR>
__global__ void __launch_bounds__(256) zonal_seg_backward_kernel(const double *__restrict__ T, double *__restrict__ out_Z,
                                                                 double *__restrict__ out_U, const double *__restrict__ cM,
                                                                 const double *__restrict__ cE, const double *__restrict__ za,
                                                                 const double *__restrict__ sy, int nlon, int S, int pitch,
                                                                 double rtheta) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pitch) return;
    const int seg = blockIdx.y % S, member = blockIdx.y / S, m = nlon / S;
    const size_t P = (size_t)pitch;
    const size_t row0 = ((size_t)member * (size_t)nlon + (size_t)seg * (size_t)m) * P + (size_t)p;
    const double *b = T + row0;
    double *d = out_Z + row0;
    double *uo = out_U ? out_U + row0 : nullptr;
    const double *M = cM + p, *E = cE + p;
    const double a = za[p];
    const size_t so = (size_t)member * (size_t)S * P + (size_t)p;
    const double y = sy[so + (size_t)seg * P], L = sy[so + (size_t)((seg + S - 1) % S) * P];
    d[(size_t)(m - 1) * P] = (y - b[(size_t)(m - 1) * P]) * rtheta;
    if (uo) uo[(size_t)(m - 1) * P] = y;
    double Un = y;
    int l = m - 2;
    for (; l - UNR + 1 >= 0; l -= UNR) {
        double bb[UNR], mm[UNR], ee[UNR], dd[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            bb[k] = b[(size_t)(l - k) * P];
            mm[k] = M[(size_t)(l - k) * P];
            ee[k] = E[(size_t)(l - k) * P];
            dd[k] = d[(size_t)(l - k) * P];
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const double U = __builtin_fma(a * mm[k], Un, __builtin_fma(ee[k], L, dd[k]));
            d[(size_t)(l - k) * P] = (U - bb[k]) * rtheta;
            if (uo) uo[(size_t)(l - k) * P] = U;
            Un = U;
        }
    }
    for (; l >= 0; --l) {
        const double U = __builtin_fma(a * M[(size_t)l * P], Un, __builtin_fma(E[(size_t)l * P], L, d[(size_t)l * P]));
        d[(size_t)l * P] = (U - b[(size_t)l * P]) * rtheta;
        if (uo) uo[(size_t)l * P] = U;
        Un = U;
    }
}

hipError_t launch_zonal_sweep_segmented(const double *T, double *out_Z, double *out_U, const double *cM, const double *cE,
                                        const double *rM, const double *rE, const double *za, const double *za2,
                                        const double *rW, double *su, double *sg, double *sy, int nlon, int S, int nmember,
                                        int pitch, double rtheta, hipStream_t s) {
    dim3 block(64), gseg((pitch + 63) / 64, S * nmember), gred((pitch + 63) / 64, nmember);
    zonal_seg_forward_kernel<EBM_ZONAL_UNR><<<gseg, block, 0, s>>>(T, out_Z, cM, cE, za, su, sg, nlon, S, pitch);
    zonal_reduced_solve_kernel<<<gred, block, 0, s>>>(su, sg, sy, rM, rE, za, za2, rW, S, pitch);
    zonal_seg_backward_kernel<EBM_ZONAL_UNR><<<gseg, block, 0, s>>>(T, out_Z, out_U, cM, cE, za, sy, nlon, S, pitch, rtheta);
    return hipGetLastError();
}

hipError_t launch_unsplit_fields(double *fields, long long field_stride, int nfields, int ncol, const LaunchCfg &cfg,
                                 hipStream_t s) {
    if (cfg.cells != 4) return hipSuccess;               // two cells per thread: the layouts coincide
    unsplit_fields_kernel<<<dim3(ncol, nfields), cfg.threads, 0, s>>>(fields, field_stride, cfg.threads);
    return hipGetLastError();
}

// ---- host-side launchers ----------------------------------------------------------------------
// Cells per thread: 4 unless the caller asks for 2 (ebm_options::cells_per_thread; nlat <= kMaxLat2 = 1536: the
// fused kernel then still fits three waves per SIMD).  4 is the throughput geometry (32 contiguous bytes per lane and
// field); a run of a FEW short meridians is latency-bound on a handful of waves, and 2 cells per thread put twice as
// many SIMDs to work on every meridian.  The geometry — and with it the tridiagonal partition, i.e. the rounding of the
// solves — is a function of (nlat, cells) ONLY, never of the number of columns: a member gives the same bits alone, in
// a large ensemble and under any sharding.
LaunchCfg choose_launch(int nlat, int cells_requested) {
    LaunchCfg cfg{};
    if (nlat > kMaxLat) {
        cfg.threads = 0;
        return cfg;
    }
    const int cells = (cells_requested == 2 && nlat <= kMaxLat2) ? 2 : 4;
    const int chunks = (nlat + cells - 1) / cells;
    cfg.threads = ((chunks + 63) / 64) * 64;
    if (cells == 2 && cfg.threads > 512) cfg.threads = 768;     // the one size compiled beyond 512 (padding cells stay zero)
    cfg.cells = cells;
    // 2 x 3T cyclic reduction + the MIZ stash of Ew, h, Tw (3 C T)
    cfg.lds_bytes = sizeof(double) * (size_t)cfg.threads * (6 + 3 * (size_t)cells);
    return cfg;
}

#endif  // EBM_PART_MAIN

namespace {

// Every workgroup size is compiled as a constant: T = 64 ... 1024 in steps of one wave (two cells per
// thread: 64 ... 512, and 768 for every meridian of 1025 ... 1536 cells).
template <int C, int GRID, int OUT, bool IMEX>
[[maybe_unused]] KernelFn miz_kernel_for(int threads) {
    switch (threads) {
#define EBM_CASE(TT) case TT: return miz_step_kernel<C, GRID, OUT, TT, IMEX>;
#ifdef EBM_QUICK   // development builds (tests/tools/resource_usage.py -DEBM_QUICK): three sizes only
        EBM_CASE(64) EBM_CASE(256) EBM_CASE(512)
#else
        EBM_CASE(64) EBM_CASE(128) EBM_CASE(192) EBM_CASE(256) EBM_CASE(320) EBM_CASE(384) EBM_CASE(448) EBM_CASE(512)
#endif
        default: break;
    }
    if constexpr (C == 2) {        // 1024 < nlat <= kMaxLat2: always 768 threads (choose_launch)
        if (threads == 768) return miz_step_kernel<C, GRID, OUT, 768, IMEX>;
    }
    if constexpr (C == 4) {
        switch (threads) {
#ifdef EBM_QUICK
            EBM_CASE(1024)
#else
            EBM_CASE(576) EBM_CASE(640) EBM_CASE(704) EBM_CASE(768) EBM_CASE(832) EBM_CASE(896) EBM_CASE(960) EBM_CASE(1024)
#endif
            default: break;
        }
    }
#undef EBM_CASE
    return nullptr;
}
template <int C, int GRID, bool IMEX>
[[maybe_unused]] KernelFn miz_step_by_mode(int mode, int threads) {
    switch (mode) {
        case OUT_STATE: return miz_kernel_for<C, GRID, OUT_STATE, IMEX>(threads);
        case OUT_DIAG: return miz_kernel_for<C, GRID, OUT_DIAG, IMEX>(threads);
        case OUT_SAVE: return miz_kernel_for<C, GRID, OUT_SAVE, IMEX>(threads);
        default: return nullptr;
    }
}

}  // namespace

// The per-step MIZ kernels of one part (declared in ebm_internal.h, each defined in its own translation unit)
#ifdef EBM_PART_G0
KernelFn miz_step_kernels_identity(int cells, int mode, int threads) {
    return cells == 2 ? miz_step_by_mode<2, 0, false>(mode, threads) : miz_step_by_mode<4, 0, false>(mode, threads);
}
#endif
#ifdef EBM_PART_G1
KernelFn miz_step_kernels_nonuniform(int cells, int mode, int threads) {
    return cells == 2 ? miz_step_by_mode<2, 1, false>(mode, threads) : miz_step_by_mode<4, 1, false>(mode, threads);
}
#endif
#ifdef EBM_PART_IMEX
KernelFn miz_step_kernels_imex(int grid_kind, int mode, int threads) {        // the extension: 4 cells per thread
    return grid_kind == 0 ? miz_step_by_mode<4, 0, true>(mode, threads) : miz_step_by_mode<4, 1, true>(mode, threads);
}
#endif

#ifdef EBM_PART_LOOP
namespace {
// every size for both models: the extension has no other fused kernel; the reference's step where the register kernel ends
// (more than kFusedRegThreads threads) and, below that, where the handle prefers occupancy over latency
template <int GRID, bool IMEX>
KernelFn miz_resident_for(int threads) {
    switch (threads) {
#define EBM_CASE(TT) case TT: return miz_resident_kernel<GRID, TT, IMEX>;
#ifdef EBM_QUICK
        EBM_CASE(64) EBM_CASE(256) EBM_CASE(512) EBM_CASE(1024)
#else
        EBM_CASE(64) EBM_CASE(128) EBM_CASE(192) EBM_CASE(256) EBM_CASE(320) EBM_CASE(384) EBM_CASE(448) EBM_CASE(512)
        EBM_CASE(576) EBM_CASE(640) EBM_CASE(704) EBM_CASE(768) EBM_CASE(832) EBM_CASE(896) EBM_CASE(960) EBM_CASE(1024)
#endif
        default: break;
    }
#undef EBM_CASE
    return nullptr;
}
}  // namespace
KernelFn miz_resident_kernels(int grid_kind, int threads, bool imex) {
    if (imex) return grid_kind == 0 ? miz_resident_for<0, true>(threads) : miz_resident_for<1, true>(threads);
    return grid_kind == 0 ? miz_resident_for<0, false>(threads) : miz_resident_for<1, false>(threads);
}
#endif

#ifdef EBM_PART_LOOPSAVE
namespace {
template <int GRID, bool IMEX>
KernelFn miz_resident_save_for(int threads) {              // every size: integrate has no other fused kernel
    switch (threads) {
#define EBM_CASE(TT) case TT: return miz_resident_kernel<GRID, TT, IMEX, true>;
#ifdef EBM_QUICK
        EBM_CASE(64) EBM_CASE(256) EBM_CASE(512) EBM_CASE(1024)
#else
        EBM_CASE(64) EBM_CASE(128) EBM_CASE(192) EBM_CASE(256) EBM_CASE(320) EBM_CASE(384) EBM_CASE(448) EBM_CASE(512)
        EBM_CASE(576) EBM_CASE(640) EBM_CASE(704) EBM_CASE(768) EBM_CASE(832) EBM_CASE(896) EBM_CASE(960) EBM_CASE(1024)
#endif
        default: break;
    }
#undef EBM_CASE
    return nullptr;
}
}  // namespace
KernelFn miz_resident_save_kernels(int grid_kind, int threads, bool imex) {
    if (imex) return grid_kind == 0 ? miz_resident_save_for<0, true>(threads) : miz_resident_save_for<1, true>(threads);
    return grid_kind == 0 ? miz_resident_save_for<0, false>(threads) : miz_resident_save_for<1, false>(threads);
}
namespace {
template <int GRID>
KernelFn miz_fused2_save_for(int threads) {
    switch (threads) {
#define EBM_CASE(TT) case TT: return miz_fused_kernel<2, GRID, TT, true>;
#ifdef EBM_QUICK
        EBM_CASE(64) EBM_CASE(256) EBM_CASE(512)
#else
        EBM_CASE(64) EBM_CASE(128) EBM_CASE(192) EBM_CASE(256) EBM_CASE(320) EBM_CASE(384) EBM_CASE(448) EBM_CASE(512)
#endif
        default: break;       // (not 768 threads — meridians of 1025 ... 1536 cells: 144 B of scratch at its three waves per SIMD;
    }                         //  ebm_integrate keeps one launch per step there)
#undef EBM_CASE
    return nullptr;
}
}  // namespace
KernelFn miz_fused2_save_kernels(int grid_kind, int threads) {     // two cells per thread: the register kernel with the sums
    return grid_kind == 0 ? miz_fused2_save_for<0>(threads) : miz_fused2_save_for<1>(threads);
}
#endif

#ifdef EBM_PART_MAIN
namespace {

// which fused-K kernel steps a handle's columns (one rule for the kernel table and the LDS size): the two compute the same
// bits, so the choice is free to depend on the column count (LaunchCfg::fused_in_lds, set by the runtime)
bool fused_state_in_lds(const LaunchCfg &cfg, bool imex) {
    return imex || (cfg.cells == 4 && (cfg.threads > kFusedRegThreads || cfg.fused_in_lds));
}
template <int C, int GRID>
KernelFn miz_fused_for(int threads) {
    switch (threads) {
#define EBM_CASE(TT) case TT: return miz_fused_kernel<C, GRID, TT>;
#ifdef EBM_QUICK
        EBM_CASE(64) EBM_CASE(256) EBM_CASE(512)
#else
        EBM_CASE(64) EBM_CASE(128) EBM_CASE(192) EBM_CASE(256) EBM_CASE(320) EBM_CASE(384) EBM_CASE(448) EBM_CASE(512)
#endif
        default: break;
    }
    if constexpr (C == 2) {        // 166 VGPRs: three waves per SIMD = kFusedRegThreads2 threads
        if (threads == 768) return miz_fused_kernel<C, GRID, 768>;
    }
#undef EBM_CASE
    return nullptr;
}
KernelFn miz_kernel(const LaunchCfg &cfg, int grid_kind, int mode, bool imex) {
    const int cells = cfg.cells, threads = cfg.threads;
    if (mode == OUT_LOOP_SAVE) {
        if (cells == 4) return miz_resident_save_kernels(grid_kind, threads, imex);
        return imex ? nullptr : miz_fused2_save_kernels(grid_kind, threads);
    }
    if (mode == OUT_LOOP) {        // fused-K: state in registers where it fits, resident in LDS otherwise
        if (fused_state_in_lds(cfg, imex)) return cells != 4 ? nullptr : miz_resident_kernels(grid_kind, threads, imex);
        if (cells == 2) return grid_kind == 0 ? miz_fused_for<2, 0>(threads) : miz_fused_for<2, 1>(threads);
        return grid_kind == 0 ? miz_fused_for<4, 0>(threads) : miz_fused_for<4, 1>(threads);
    }
    if (imex) return cells != 4 ? nullptr : miz_step_kernels_imex(grid_kind, mode, threads);
    return grid_kind == 0 ? miz_step_kernels_identity(cells, mode, threads) : miz_step_kernels_nonuniform(cells, mode, threads);
}
template <int C>
KernelFn classic_kernel_c(int mode) {
    switch (mode) {
        case OUT_STATE:
        case OUT_DIAG: return classic_step_kernel<C, OUT_STATE>;
        case OUT_SAVE: return classic_step_kernel<C, OUT_SAVE>;
        case OUT_LOOP: return classic_step_kernel<C, OUT_LOOP>;
        default: return nullptr;
    }
}
KernelFn classic_kernel(int cells, int mode) { return cells == 2 ? classic_kernel_c<2>(mode) : classic_kernel_c<4>(mode); }
// LDS of a launch: the fused register kernel only needs the solve's buffers; the resident kernel 4T for the solve and
// 4 fields x 4 cells x T for the state
size_t miz_lds_bytes(const LaunchCfg &cfg, int mode, bool imex) {
    if (mode == OUT_LOOP_SAVE) return sizeof(double) * (cfg.cells == 4 ? 20 : 6) * (size_t)cfg.threads;
    if (mode == OUT_LOOP) return sizeof(double) * (fused_state_in_lds(cfg, imex) ? 20 : 6) * (size_t)cfg.threads;
    return cfg.lds_bytes;
}

}  // namespace

// Dynamic LDS above the 64 KiB default must be requested per kernel.
hipError_t prepare_kernels(const LaunchCfg &cfg) {
    auto raise = [](KernelFn fn, size_t bytes) -> hipError_t {
        if (!fn) return hipErrorInvalidValue;
        return hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    };
    for (int grid = 0; grid < 2; ++grid)
        for (int imex = 0; imex < (cfg.cells == 4 ? 2 : 1); ++imex) {
            if (cfg.lds_bytes > 64 * 1024)
                for (int mode = OUT_STATE; mode <= OUT_SAVE; ++mode) {   // (the fused register kernel needs 6T doubles <= 24 KiB)
                    hipError_t e = raise(miz_kernel(cfg, grid, mode, imex != 0), cfg.lds_bytes);
                    if (e != hipSuccess) return e;
                }
            // the resident fused-K kernels and their savesol! variants (four cells per thread): 160 T bytes
            const size_t bytes = sizeof(double) * 20 * (size_t)cfg.threads;
            if (cfg.cells != 4 || bytes <= 64 * 1024) continue;
            hipError_t e = raise(miz_resident_kernels(grid, cfg.threads, imex != 0), bytes);
            if (e == hipSuccess) e = raise(miz_resident_save_kernels(grid, cfg.threads, imex != 0), bytes);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

bool has_miz_kernel(const LaunchCfg &cfg, int grid_kind, int mode, bool imex) {
    return miz_kernel(cfg, grid_kind, mode, imex) != nullptr;
}

hipError_t launch_miz_step(const StepArgs &a, int grid_kind, int mode, const LaunchCfg &cfg, bool imex, int first, int count,
                           hipStream_t s) {
    KernelFn fn = miz_kernel(cfg, grid_kind, mode, imex);
    if (!fn || first < 0 || count < 1 || first + count > a.ncol) return hipErrorInvalidValue;
    StepArgs b = a;
    b.col0 = first;
    fn<<<dim3(count), dim3(cfg.threads), miz_lds_bytes(cfg, mode, imex), s>>>(b);
    return hipGetLastError();
}

hipError_t launch_classic_step(const StepArgs &a, int mode, const LaunchCfg &cfg, int first, int count, hipStream_t s) {
    KernelFn fn = classic_kernel(cfg.cells, mode);
    if (!fn || first < 0 || count < 1 || first + count > a.ncol) return hipErrorInvalidValue;
    StepArgs b = a;
    b.col0 = first;
    fn<<<dim3(count), dim3(cfg.threads), sizeof(double) * 6 * (size_t)cfg.threads, s>>>(b);
    return hipGetLastError();
}

hipError_t launch_mask_from_t0(const StepArgs &a, int ncol, const LaunchCfg &cfg, hipStream_t s) {
    mask_from_t0_kernel<<<ncol, cfg.threads, 0, s>>>(a, cfg.cells);
    return hipGetLastError();
}

hipError_t launch_hemispheric_mean(const double *field, const double *x, int pitch, int nlat, int ncol, double *out,
                                   hipStream_t s) {
    hemispheric_mean_kernel<<<ncol, 256, sizeof(double) * (size_t)nlat, s>>>(field, x, pitch, nlat, out);
    return hipGetLastError();
}
hipError_t launch_finish_mean(double *dst, double *sum, double nt, int ncol, int nvars, long long var_stride,
                              const LaunchCfg &cfg, hipStream_t s) {
    finish_mean_kernel<<<dim3(ncol, nvars), cfg.threads, 0, s>>>(dst, sum, nt, cfg.threads, cfg.cells, var_stride);
    return hipGetLastError();
}
#endif  // EBM_PART_MAIN

}  // namespace ebm
