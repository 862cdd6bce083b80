// HIP kernels (gfx950 / CDNA4, wave64) for the energy-balance time-stepping hot path.
//
// One workgroup integrates one meridian (column) for one step.  Everything that the reference
// does in ~60 temporary vectors per step (src/miz.jl:150-196) is fused into one kernel:
//
//   phase A  coalesced 16-B loads along latitude ("interleaved" ownership: thread t owns the
//            cell pairs t, t+T, t+2T, ...), water temperature, T0-system right-hand side
//            (3-point stencil through an LDS tile with implicit zero-flux halo)
//   phase B  T0 solve: active-set Newton on the piecewise-linear system of src/miz.jl:33-45;
//            each linear system is tridiagonal and is solved per meridian by a chunk
//            partition (each thread eliminates its C contiguous rows in registers) followed
//            by parallel cyclic reduction of the T-row interface system in LDS
//   phase C  back to interleaved ownership
//   phase D  Tbar stencil through LDS, radiative + lateral fluxes, enthalpy Euler step,
//            redistribution, floe size / thickness / concentration update, coalesced stores
//
// Arithmetic policy.  Everything outside the tridiagonal solves is a bit-exact restatement of
// the reference's expressions (IEEE division, no FMA contraction: build with
// -ffp-contract=off; the order of operations is the reference's).  The solves are free to use
// any arithmetic: their result is defined by the linear system, not by an operation order.
//
// No MFMA: there is no dense contraction on this path; it is HBM-/fp64-VALU-bound.
#include "ebm_internal.h"

namespace ebm {

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

// ---- LDS tile addressing -----------------------------------------------------------------
// A meridian tile holds T*C cells.  One pad element after every C cells makes the stride of
// a thread's chunk C+1 doubles, which is conflict-free for ds_read_b64 (stride 2(C+1) dwords,
// C+1 odd) while interleaved (lane-consecutive) accesses stay conflict-free too.
template <int C>
__device__ __forceinline__ int pidx(int k) {
    return k + k / C;
}

// interleaved ownership: thread t holds cell pairs p = t + j*T  (cells 2p, 2p+1)
template <int C>
__device__ __forceinline__ void load_il(const double *__restrict__ f, size_t base, int t, int T,
                                        int plat, double (&v)[C]) {
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        int k0 = 2 * (t + j * T);
        if (k0 < plat) {
            double2 d = *reinterpret_cast<const double2 *>(f + base + k0);
            v[2 * j] = d.x;
            v[2 * j + 1] = d.y;
        } else {
            v[2 * j] = 0.0;
            v[2 * j + 1] = 0.0;
        }
    }
}
template <int C>
__device__ __forceinline__ void store_il(double *__restrict__ f, size_t base, int t, int T,
                                         int plat, int nlat, const double (&v)[C]) {
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        int k0 = 2 * (t + j * T);
        if (k0 < plat) {
            double2 d;
            d.x = v[2 * j];
            d.y = (k0 + 1 < nlat) ? v[2 * j + 1] : 0.0;   // keep the pitch padding at zero
            *reinterpret_cast<double2 *>(f + base + k0) = d;
        }
    }
}
// chunk ownership: thread t holds cells t*C .. t*C+C-1 (arrays are padded to T*C on the host)
template <int C>
__device__ __forceinline__ void load_chunk(const double *__restrict__ f, int t, double (&v)[C]) {
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        double2 d = *reinterpret_cast<const double2 *>(f + t * C + 2 * j);
        v[2 * j] = d.x;
        v[2 * j + 1] = d.y;
    }
}
// interleaved registers -> LDS tile
template <int C>
__device__ __forceinline__ void tile_put_il(double *tile, int t, int T, const double (&v)[C]) {
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        int k0 = 2 * (t + j * T);
        tile[pidx<C>(k0)] = v[2 * j];
        tile[pidx<C>(k0 + 1)] = v[2 * j + 1];
    }
}
template <int C>
__device__ __forceinline__ void tile_get_il(const double *tile, int t, int T, double (&v)[C]) {
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        int k0 = 2 * (t + j * T);
        v[2 * j] = tile[pidx<C>(k0)];
        v[2 * j + 1] = tile[pidx<C>(k0 + 1)];
    }
}
template <int C>
__device__ __forceinline__ void tile_get_chunk(const double *tile, int t, double (&v)[C]) {
#pragma unroll
    for (int i = 0; i < C; ++i) v[i] = tile[t * (C + 1) + i];
}
template <int C>
__device__ __forceinline__ void tile_put_chunk(double *tile, int t, const double (&v)[C]) {
#pragma unroll
    for (int i = 0; i < C; ++i) tile[t * (C + 1) + i] = v[i];
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
// On entry R0/R1 (>= 3T doubles each) must be free; on exit other threads may still read them.
template <int C>
__device__ __forceinline__ void partition_solve(const double (&a)[C], const double (&b)[C],
                                                const double (&c)[C], const double (&d)[C],
                                                double (&x)[C], int t, int T, double *R0,
                                                double *R1) {
    double cp[C - 1], dp[C - 1], lp[C - 1];
    {
        double w = 1.0 / b[0];
        cp[0] = c[0] * w;
        dp[0] = d[0] * w;
        lp[0] = -a[0] * w;
    }
#pragma unroll
    for (int i = 1; i < C - 1; ++i) {
        double w = 1.0 / (b[i] - a[i] * cp[i - 1]);
        cp[i] = c[i] * w;
        dp[i] = (d[i] - a[i] * dp[i - 1]) * w;
        lp[i] = -(a[i] * lp[i - 1]) * w;
    }
    double u = dp[C - 2], v = lp[C - 2], wr = -cp[C - 2];
#pragma unroll
    for (int i = C - 3; i >= 0; --i) {
        u = dp[i] - cp[i] * u;
        v = lp[i] - cp[i] * v;
        wr = -cp[i] * wr;
    }
    R1[t] = u;
    R1[T + t] = v;
    R1[2 * T + t] = wr;
    __syncthreads();
    double un = 0.0, vn = 0.0, wn = 0.0;
    if (t + 1 < T) {
        un = R1[t + 1];
        vn = R1[T + t + 1];
        wn = R1[2 * T + t + 1];
    }
    double pa, pc, pd;
    {
        const double ae = a[C - 1], be = b[C - 1], ce = c[C - 1], de = d[C - 1];
        double RA = ae * lp[C - 2];
        double RB = be - ae * cp[C - 2] + ce * vn;
        double RC = ce * wn;
        double RD = de - ae * dp[C - 2] - ce * un;
        double rinv = 1.0 / RB;
        pa = RA * rinv;
        pc = RC * rinv;
        pd = RD * rinv;
    }
    double *src = R0, *dst = R1;
    src[t] = pa;
    src[T + t] = pc;
    src[2 * T + t] = pd;
    __syncthreads();
    for (int s = 1; s < T; s <<= 1) {
        double am = 0.0, cm = 0.0, dm = 0.0, ap = 0.0, cn = 0.0, dn = 0.0;
        if (t - s >= 0) {
            am = src[t - s];
            cm = src[T + t - s];
            dm = src[2 * T + t - s];
        }
        if (t + s < T) {
            ap = src[t + s];
            cn = src[T + t + s];
            dn = src[2 * T + t + s];
        }
        double r = 1.0 / (1.0 - pa * cm - pc * ap);
        double npd = (pd - pa * dm - pc * dn) * r;
        double npa = -(pa * am) * r;
        double npc = -(pc * cn) * r;
        pa = npa;
        pc = npc;
        pd = npd;
        dst[t] = pa;
        dst[T + t] = pc;
        dst[2 * T + t] = pd;
        __syncthreads();
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
    const double L = t > 0 ? src[2 * T + t - 1] : 0.0;
    x[C - 1] = pd;
#pragma unroll
    for (int i = C - 2; i >= 0; --i) x[i] = dp[i] + lp[i] * L - cp[i] * x[i + 1];
}

// ---- MIZ pointwise physics (one cell), bit-exact restatement of src/miz.jl:160-194 ----------
struct MizCellOut {
    double Ei, Ew, h, D, phi, n, E, T, Ti, Tw;
};

__device__ __forceinline__ MizCellOut miz_cell_update(const Params &p, double dt, double f,
                                                     double S, double xk, double dif, double tb,
                                                     double Ei, double Ew, double hk, double Dk,
                                                     double ph, double Tw, double Ti) {
    const double Tm = p.Tm, Lf = p.Lf, alpha = p.alpha;
    // num, src/miz.jl:83-87
    double n = ph / (alpha * (Dk * Dk));
    if (Dk == 0.0) n = 0.0;
    // vert_flux, src/miz.jl:96-101 (called twice in the reference with the same Tbar/diffusion)
    double L = p.A + p.B * (tb - Tm);
    double sol_i = 0.0 + p.ai * S;
    double sol_w = 0.0 + (p.a0 - p.a2 * (xk * xk)) * S;
    double Fvi = sol_i - L + dif + p.Fb + f;
    double Fvw = sol_w - L + dif + p.Fb + f;
    // wlat :71, lat_flux :103-107
    double wl = p.m1 * (Tw - p.Tm_pow_m2);
    double Flat = ph * hk * Lf * wl * M_PI / (alpha * Dk);
    if (Dk == 0.0) Flat = 0.0;
    // forward Euler (:137-138,148,166-167) and redistributeE (:109-117)
    double rEi = Ei + (ph * Fvi + Flat) * dt;
    double rEw = Ew + ((1.0 - ph) * Fvw - Flat) * dt;
    double cEi = jl_clamp(rEi, -INFINITY, 0.0);
    double cEw = jl_clamp(rEw, 0.0, INFINITY);
    double psiEidt = rEi - cEi, psiEwdt = rEw - cEw;
    double Ei_n = cEi + psiEwdt, Ew_n = cEw + psiEidt;
    // area_lead :90-93
    double Dr = Dk + p.two_rl;
    double ring = alpha * n * (Dr * Dr - Dk * Dk);
    double Al = jl_min(ring, 1.0 - ph);
    // split_psiEw :120-125 applied to psiEwdt/dt (:173)
    double psi = psiEwdt / dt;
    double Ql = Al / (1.0 - ph) * psi;
    if (ph == 1.0) Ql = 0.0;
    double Qp = psi - Ql;
    // psinplus :127, :174
    double dn = dt * (-Qp / p.c_dn);
    // D_t :140-146
    double lat_melt = p.c_latmelt * wl;
    double lat_grow = -Dk / (2.0 * Lf * hk * ph) * Ql;
    double weld = p.c_weld * ph * (Dk * Dk * Dk);
    if (hk == 0.0) lat_grow = 0.0;
    double rD = Dk + (lat_melt + lat_grow + weld) * dt;
    // average :129-134, clamp!, zeroref! (:175-178)
    double total = n + dn;
    double D_n = (n * rD + dn * p.Dmin) / total;
    if (total == 0.0) D_n = 0.0;
    D_n = jl_clamp(D_n, p.Dmin, p.Dmax);
    if (Ei_n == 0.0) D_n = 0.0;
    // thickness :179-181
    double rh = hk + (p.c_ht * Fvi) * dt;
    rh = jl_clamp(rh, 0.0, INFINITY);
    double h_n = (n * rh + dn * p.hmin) / total;
    if (total == 0.0) h_n = 0.0;
    // concentration :74-80
    double phi_n = -Ei_n / (Lf * h_n);
    if (h_n == 0.0) phi_n = 0.0;
    if (phi_n > 1.0) phi_n = 1.0;
    if (h_n == 0.0) Ei_n = 0.0;   // :185
    MizCellOut o;
    o.Ei = Ei_n;
    o.Ew = Ew_n;
    o.h = h_n;
    o.D = D_n;
    o.phi = phi_n;
    o.n = n;
    o.E = phi_n * Ei_n + (1.0 - phi_n) * Ew_n;          // :186
    o.T = Ti * phi_n + (1.0 - phi_n) * Tw;              // :187 (old Ti, Tw; new phi)
    o.Ti = (Ei_n == 0.0) ? __builtin_nan("") : Ti;      // :193
    o.Tw = (phi_n > 0.99) ? __builtin_nan("") : Tw;     // :194
    return o;
}

// D d/dx[(1-x^2) dT/dx] at cell k added to base; tbm/tbp = T at k-1 / k+1.
// GRID 0: CSC SpMV order of src/infrastructure.jl:495-497; GRID 1: flux form :521-524.
template <int GRID>
__device__ __forceinline__ double diffusion_add(double base, double D, int k, int nlat, double g0,
                                                double g1, double g2, double g3, double g4,
                                                double tbm, double tbk, double tbp) {
    if (GRID == 0) {
        double y = 0.0;
        if (k > 0) y = y + g0 * tbm;
        y = y + g1 * tbk;
        if (k < nlat - 1) y = y + g2 * tbp;
        return base + y;
    } else {
        double dTp = (k < nlat - 1) ? tbp - tbk : 0.0;
        double dTm = (k > 0) ? tbk - tbm : 0.0;
        return base + (D * ((g0 * dTp) / g2 - (g1 * dTm) / g3)) / g4;
    }
}

template <int C, int GRID>
__global__ void __launch_bounds__(1024) miz_step_kernel(const MizArgs a) {
    extern __shared__ double smem[];
    const int T = blockDim.x, t = threadIdx.x, col = blockIdx.x;
    const int nlat = a.nlat, plat = (int)a.pitch;
    double *R0 = smem;
    double *R1 = smem + T * (C + 1);
    const size_t base = (size_t)col * (size_t)a.pitch;
    const Params &p = a.p;
    const double f = a.fcol ? a.ft + a.fcol[col] : a.ft;
    const double Tm = p.Tm;

    // ---------------- phase A: loads, water temperature, T0-system coefficients ----------
    double Ei[C], Ew[C], hk[C], Dk[C], ph[C], Tw[C], xk[C];
    load_il<C>(a.Ew, base, t, T, plat, Ew);
    load_il<C>(a.phi, base, t, T, plat, ph);
    load_il<C>(a.h, base, t, T, plat, hk);
    load_il<C>(a.g.x, 0, t, T, plat, xk);
    double cdd[C], crhs[C], cphi[C], cv0[C];
    {
        double v0[C], lo[C], di[C], up[C], dd[C], r[C], rhs[C], S[C];
        load_il<C>(a.T0, base, t, T, plat, v0);
        load_il<C>(a.g.lo, 0, t, T, plat, lo);
        load_il<C>(a.g.di, 0, t, T, plat, di);
        load_il<C>(a.g.up, 0, t, T, plat, up);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int k = 2 * (t + (c >> 1) * T) + (c & 1);
            const bool valid = k < nlat;
            double tw = Tm + Ew[c] / ((1.0 - ph[c]) * p.cw);      // water_temp, src/miz.jl:30
            tw = __builtin_isnan(tw) ? 0.0 : tw;                  // :157
            Tw[c] = tw;
            double hp = (hk[c] == 0.0) ? p.hmin : hk[c];          // :51
            dd[c] = valid ? p.k / hp + p.B : -1.0;
            r[c] = valid ? (1.0 - ph[c]) * (tw - Tm) : 0.0;
            S[c] = p.S0 - p.S1 * xk[c] * a.ct - p.S2 * (xk[c] * xk[c]);   // :11
            v0[c] = valid ? v0[c] - Tm : 0.0;
            if (!valid) ph[c] = 0.0;
        }
        tile_put_il<C>(R0, t, T, r);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < C / 2; ++j) {
            const int k0 = 2 * (t + j * T);
            const double rm = k0 > 0 ? R0[pidx<C>(k0 - 1)] : 0.0;
            const double rp = k0 + 2 < T * C ? R0[pidx<C>(k0 + 2)] : 0.0;
            const bool v0ok = k0 < nlat, v1ok = k0 + 1 < nlat;
            rhs[2 * j] = v0ok ? p.ai * S[2 * j] - p.A +
                                    (lo[2 * j] * rm + di[2 * j] * r[2 * j] + up[2 * j] * r[2 * j + 1]) + f
                              : 0.0;
            rhs[2 * j + 1] = v1ok ? p.ai * S[2 * j + 1] - p.A +
                                        (lo[2 * j + 1] * r[2 * j] + di[2 * j + 1] * r[2 * j + 1] +
                                         up[2 * j + 1] * rp) + f
                                  : 0.0;
        }
        // interleaved -> chunk ownership, ping-pong between the two tiles (one barrier each)
        tile_put_il<C>(R1, t, T, dd);
        __syncthreads();
        tile_get_chunk<C>(R1, t, cdd);
        tile_put_il<C>(R0, t, T, rhs);
        __syncthreads();
        tile_get_chunk<C>(R0, t, crhs);
        tile_put_il<C>(R1, t, T, ph);
        __syncthreads();
        tile_get_chunk<C>(R1, t, cphi);
        tile_put_il<C>(R0, t, T, v0);
        __syncthreads();
        tile_get_chunk<C>(R0, t, cv0);
    }

    // ---------------- phase B: active-set Newton, src/miz.jl:47-68 ------------------------
    double xs[C];
    int nit = 0;
    bool ok = false;
    {
        double clo[C], cdi[C], cup[C];
        load_chunk<C>(a.g.lo, t, clo);
        load_chunk<C>(a.g.di, t, cdi);
        load_chunk<C>(a.g.up, t, cup);
        bool s[C];
#pragma unroll
        for (int i = 0; i < C; ++i) s[i] = cv0[i] < 0.0;
        __syncthreads();   // tiles free
        while (nit < kMaxNewton) {
            ++nit;
            double g[C];
#pragma unroll
            for (int i = 0; i < C; ++i) g[i] = s[i] ? cphi[i] : 0.0;
            R0[t] = g[0];
            R0[T + t] = g[C - 1];
            __syncthreads();
            const double gl = t > 0 ? R0[T + t - 1] : 0.0;
            const double gr = t + 1 < T ? R0[t + 1] : 0.0;
            double ra[C], rb[C], rc[C], rd[C];
#pragma unroll
            for (int i = 0; i < C; ++i) {
                ra[i] = clo[i] * (i > 0 ? g[i - 1] : gl);
                rc[i] = cup[i] * (i < C - 1 ? g[i + 1] : gr);
                rb[i] = cdi[i] * g[i] - cdd[i];
                rd[i] = -crhs[i];
            }
            partition_solve<C>(ra, rb, rc, rd, xs, t, T, R0, R1);
            int changed = 0;
#pragma unroll
            for (int i = 0; i < C; ++i) {
                bool sn = xs[i] < 0.0;
                changed |= (sn != s[i]);
                s[i] = sn;
            }
            if (!__syncthreads_or(changed)) {
                ok = true;
                break;
            }
        }
    }
    if (t == 0 && a.counters) {
        unsigned long long *cnt = a.counters + 2 * (col % kCounterShards);
        atomicAdd(cnt, (unsigned long long)nit);
        if (!ok) atomicAdd(cnt + 1, 1ull);
    }

    // ---------------- phase C: chunk -> interleaved ownership ---------------------------------
    double T0[C], Ti[C], tb[C];
    tile_put_chunk<C>(R0, t, xs);
    __syncthreads();
    tile_get_il<C>(R0, t, T, T0);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        T0[c] = T0[c] + Tm;                                       // new warm start, :64
        double ti = jl_min(T0[c], Tm);                            // ice_temp, :31,65
        Ti[c] = (hk[c] == 0.0) ? 0.0 : ti;                        // zeroref!, :66
        tb[c] = Ti[c] * ph[c] + (1.0 - ph[c]) * Tw[c];            // Tbar, :21-26
    }
    store_il<C>(a.T0, base, t, T, plat, nlat, T0);

    // ---------------- phase D: fluxes and state update ---------------------------------------
    tile_put_il<C>(R1, t, T, tb);
    load_il<C>(a.Ei, base, t, T, plat, Ei);
    load_il<C>(a.D, base, t, T, plat, Dk);
    double g0[C], g1[C], g2[C], g3[C], g4[C];
    load_il<C>(a.g.g0, 0, t, T, plat, g0);
    load_il<C>(a.g.g1, 0, t, T, plat, g1);
    load_il<C>(a.g.g2, 0, t, T, plat, g2);
    if (GRID == 1) {
        load_il<C>(a.g.g3, 0, t, T, plat, g3);
        load_il<C>(a.g.g4, 0, t, T, plat, g4);
    }
    __syncthreads();
    double oEi[C], oEw[C], oh[C], oD[C], ophi[C], on[C], oE[C], oT[C], oTi[C], oTw[C];
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        const int k0 = 2 * (t + j * T);
        const double tbm = k0 > 0 ? R1[pidx<C>(k0 - 1)] : 0.0;
        const double tbp = k0 + 2 < T * C ? R1[pidx<C>(k0 + 2)] : 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = 2 * j + q, k = k0 + q;
            const double m = q == 0 ? tbm : tb[c > 0 ? c - 1 : 0];
            const double pl = q == 0 ? tb[c + 1 < C ? c + 1 : c] : tbp;
            const double S = p.S0 - p.S1 * xk[c] * a.ct - p.S2 * (xk[c] * xk[c]);
            const double dif = diffusion_add<GRID>(0.0, p.D, k, nlat, g0[c], g1[c], g2[c],
                                                   GRID == 1 ? g3[c] : 0.0, GRID == 1 ? g4[c] : 0.0,
                                                   m, tb[c], pl);
            MizCellOut o = miz_cell_update(p, a.dt, f, S, xk[c], dif, tb[c], Ei[c], Ew[c], hk[c],
                                           Dk[c], ph[c], Tw[c], Ti[c]);
            oEi[c] = o.Ei; oEw[c] = o.Ew; oh[c] = o.h; oD[c] = o.D; ophi[c] = o.phi;
            on[c] = o.n; oE[c] = o.E; oT[c] = o.T; oTi[c] = o.Ti; oTw[c] = o.Tw;
        }
    }
    store_il<C>(a.Ei, base, t, T, plat, nlat, oEi);
    store_il<C>(a.Ew, base, t, T, plat, nlat, oEw);
    store_il<C>(a.h, base, t, T, plat, nlat, oh);
    store_il<C>(a.D, base, t, T, plat, nlat, oD);
    store_il<C>(a.phi, base, t, T, plat, nlat, ophi);
    if (a.write_diag) {
        store_il<C>(a.n, base, t, T, plat, nlat, on);
        store_il<C>(a.E, base, t, T, plat, nlat, oE);
        store_il<C>(a.T, base, t, T, plat, nlat, oT);
        store_il<C>(a.Ti, base, t, T, plat, nlat, oTi);
        store_il<C>(a.Tw, base, t, T, plat, nlat, oTw);
    }
}

// ---- classic (WE15) step, src/classic.jl:37-71 ------------------------------------------------
template <int C>
__global__ void __launch_bounds__(1024) classic_step_kernel(const ClassicArgs a) {
    extern __shared__ double smem[];
    const int T = blockDim.x, t = threadIdx.x, col = blockIdx.x;
    const int nlat = a.nlat, plat = (int)a.pitch;
    double *R0 = smem;
    double *R1 = smem + T * (C + 1);
    const size_t base = (size_t)col * (size_t)a.pitch;
    const Params &p = a.p;
    const double f = a.fcol ? a.ft + a.fcol[col] : a.ft;

    double E[C], Tg[C], xk[C], aw[C], Sb[C], kd[C];
    load_il<C>(a.E, base, t, T, plat, E);
    load_il<C>(a.Tg, base, t, T, plat, Tg);
    load_il<C>(a.g.x, 0, t, T, plat, xk);
    load_il<C>(a.g.aw, 0, t, T, plat, aw);
    load_il<C>(a.g.Sb, 0, t, T, plat, Sb);
    load_il<C>(a.g.kdiag, 0, t, T, plat, kd);
    double b[C], d[C], oT[C], oh[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int k = 2 * (t + (c >> 1) * T) + (c & 1);
        const bool valid = k < nlat;
        double Ek = E[c];
        const double S_i = Sb[c] - (p.S1 * a.ct_i) * xk[c];                        // :23-24
        const double S_ip1 = Sb[c] - (p.S1 * a.ct_ip1) * xk[c];
        const double alpha = bool_mul(aw[c], Ek > 0.0) + bool_mul(p.ai, Ek < 0.0); // :47
        const double Cc = alpha * S_i + p.cg_tau * Tg[c] - p.A + f;                // :48
        const double T0 = Cc / (p.M - p.kLf / Ek);                                 // :50
        const double Tk = bool_mul(Ek / p.cw, Ek >= 0.0) + bool_mul(bool_mul(T0, Ek < 0.0), T0 < 0.0);
        Ek = Ek + a.dt * (Cc - p.M * Tk + p.Fb);                                   // :53
        const double den = p.M - p.kLf / Ek;
        const double q = bool_mul(bool_mul(p.dc / den, T0 < 0.0), Ek < 0.0);       // :56
        const double rhs = Tg[c] + p.dt_tau * (bool_mul(Ek / p.cw, Ek >= 0.0) +
                           bool_mul(bool_mul((p.ai * S_ip1 - p.A + f) / den, T0 < 0.0), Ek < 0.0));
        b[c] = valid ? kd[c] - q : 1.0;
        d[c] = valid ? rhs : 0.0;
        E[c] = Ek;
        oT[c] = Tk;
        oh[c] = bool_mul(-Ek / p.Lf, Ek < 0.0);                                    // :65
    }
    store_il<C>(a.E, base, t, T, plat, nlat, E);
    if (a.write_diag) {
        store_il<C>(a.T, base, t, T, plat, nlat, oT);
        store_il<C>(a.h, base, t, T, plat, nlat, oh);
    }
    double cb[C], cd[C], ca[C], cc[C], xs[C];
    tile_put_il<C>(R0, t, T, b);
    __syncthreads();
    tile_get_chunk<C>(R0, t, cb);
    tile_put_il<C>(R1, t, T, d);
    __syncthreads();
    tile_get_chunk<C>(R1, t, cd);
    load_chunk<C>(a.g.ksub, t, ca);
    load_chunk<C>(a.g.ksup, t, cc);
    __syncthreads();
    partition_solve<C>(ca, cb, cc, cd, xs, t, T, R0, R1);   // Implicit Euler for Tg, :55-63
    __syncthreads();
    tile_put_chunk<C>(R0, t, xs);
    __syncthreads();
    tile_get_il<C>(R0, t, T, Tg);
    store_il<C>(a.Tg, base, t, T, plat, nlat, Tg);
}

// ---- savesol! helpers (src/infrastructure.jl:549-591, src/utilities.jl:390-395) ---------------
__global__ void accumulate_kernel(double *__restrict__ sum, const double *__restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        sum[i] = sum[i] + src[i];
}
__global__ void finish_mean_kernel(double *__restrict__ dst, double *__restrict__ sum, double nt, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        dst[i] = sum[i] / nt;
        sum[i] = 0.0;
    }
}

// ---- host-side launchers ----------------------------------------------------------------------
LaunchCfg choose_launch(int nlat) {
    LaunchCfg cfg{};
    for (int C : {4, 8}) {
        int chunks = (nlat + C - 1) / C;
        int T = ((chunks + 63) / 64) * 64;
        if (T <= 1024) {
            cfg.threads = T;
            cfg.cells = C;
            cfg.lds_bytes = sizeof(double) * 2 * (size_t)T * (C + 1);
            return cfg;
        }
    }
    cfg.threads = 0;
    return cfg;
}

// Dynamic LDS above the 64 KiB default must be requested per kernel.  The kernels also hold a
// few hundred bytes of static LDS (__syncthreads_or), so ask only for what the launch needs.
hipError_t prepare_kernels(const LaunchCfg &cfg) {
    if (cfg.lds_bytes <= 64 * 1024) return hipSuccess;
    const int bytes = (int)cfg.lds_bytes;
    hipError_t e;
#define EBM_SET(fn) \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
    if (e != hipSuccess) return e;
    if (cfg.cells == 4) {
        EBM_SET((miz_step_kernel<4, 0>))
        EBM_SET((miz_step_kernel<4, 1>))
        EBM_SET((classic_step_kernel<4>))
    } else {
        EBM_SET((miz_step_kernel<8, 0>))
        EBM_SET((miz_step_kernel<8, 1>))
        EBM_SET((classic_step_kernel<8>))
    }
#undef EBM_SET
    return hipSuccess;
}

hipError_t launch_miz_step(const MizArgs &a, int grid_kind, const LaunchCfg &cfg, hipStream_t s) {
    dim3 grid(a.ncol), block(cfg.threads);
    if (cfg.cells == 4) {
        if (grid_kind == 0) miz_step_kernel<4, 0><<<grid, block, cfg.lds_bytes, s>>>(a);
        else miz_step_kernel<4, 1><<<grid, block, cfg.lds_bytes, s>>>(a);
    } else {
        if (grid_kind == 0) miz_step_kernel<8, 0><<<grid, block, cfg.lds_bytes, s>>>(a);
        else miz_step_kernel<8, 1><<<grid, block, cfg.lds_bytes, s>>>(a);
    }
    return hipGetLastError();
}

hipError_t launch_classic_step(const ClassicArgs &a, const LaunchCfg &cfg, hipStream_t s) {
    dim3 grid(a.ncol), block(cfg.threads);
    if (cfg.cells == 4) classic_step_kernel<4><<<grid, block, cfg.lds_bytes, s>>>(a);
    else classic_step_kernel<8><<<grid, block, cfg.lds_bytes, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_accumulate(double *sum, const double *src, size_t n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    accumulate_kernel<<<blocks, 256, 0, s>>>(sum, src, n);
    return hipGetLastError();
}
hipError_t launch_finish_mean(double *dst, double *sum, double nt, size_t n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    finish_mean_kernel<<<blocks, 256, 0, s>>>(dst, sum, nt, n);
    return hipGetLastError();
}

}  // namespace ebm
