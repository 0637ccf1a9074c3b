// smallk_amd/csrc/rank2_persist.hip -- a whole RANK2 factorisation of a sparse node matrix as ONE launch.
//
// HierNMF2 on a large graph spends most of its wall time in a few medium-sized nodes that need thousands of rank-2
// iterations (the C5-shaped run: 8 400 of 9 759 iterations on two 190 k x 62 k nodes with 1 M entries, 54 us each as six
// launches of 7 - 12 us on 24 MB of data).  Here the driver loop of NmfSolve<> (nmf_solve_generic.hpp:67-139) with the
// RANK2 solver (nmf_solver_rank2.hpp:353-455) and the PG_RATIO rule (progress_estimator_generic.hpp:30-69) runs inside ONE
// resident kernel: one workgroup of 1024 threads per CU at most, TWO grid-wide barriers per iteration, no host round trip.
//
// The data flow that makes two barriers enough.  Column j of A is handled by a fixed group of lanes, row i by another:
//   phase H(t):  h_j = solve(W'W, (W'A)_j)                       -> Hc[t & 1][j], partial sums of HH'
//   -- barrier B1: HH' = sum of the partials (every workgroup adds them up itself, same order, same bits);
//      also the stopping rule of iteration t - 1, whose projected-gradient partials were written before this barrier
//   phase W(t):  r_i = sum over row i of A: a * Hc[col];  w_i = solve(HH', r_i)   -> Wc[i] (NOT yet normalised), R2c[i]
//   -- barrier B2: G = sum of the partials of W'W; nu = sqrt(diag G)
//   phase G(t):  the per-iteration NormalizeAndScale (:418-437) is applied on the fly -- the gather reads the raw W and
//      the sums are divided by nu (W'A of the normalised W = D^-1 (Wraw' A)), W'W = D^-1 G D^-1, HH' and AH' are scaled the
//      same way -- so no barrier separates the normalisation from the product that needs it:
//      (W'A)_j = D^-1 sum over column j: a * Wc[row];  projected-gradient partial sums of this iteration (row group: W side,
//      column group: H side);  and, fused, phase H(t + 1) for the same column from the sums still in registers.
// The normalised W and the scaled H are materialised once, by the epilogue, for the iteration the run ends on.
//
// Grid barrier: XCD-hierarchical (MI355X_MICROARCH.md "barrier-xcd"): arrivals per XCD, the last arriver of an XCD
// releases (buffer_wbl2 sc1: the XCD's L2 is written back once) and arrives at the top counter, the last XCD opens the
// generation word; every workgroup acquires (buffer_inv sc1) before it reads what the others wrote.  Which XCD a workgroup
// runs on is read from the hardware (HW_REG_XCC_ID), the membership counts come from one flat barrier at kernel start:
// nothing depends on dispatch order or placement.  Every single wait is bounded by the wall clock (5 s); on expiry every workgroup
// leaves through the abort word and the host runs the launch-per-kernel loop instead (the solver state was not touched).
#include "devutil.h"
#include "rank2_math.h"

#include <algorithm>

namespace smk {

#define R2P_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef GLOBAL_AS unsigned gu32_t;

enum { R2P_XTOT = 0, R2P_XCNT = 8, R2P_XGEN = 16, R2P_TOP = 24, R2P_TOPGEN = 25, R2P_FLAT = 26, R2P_ABORT = 27, R2P_WORDS = 28 };

struct GridBar {
    gu32_t* w;
    unsigned xcc, xtot, nx, epoch;
    unsigned long long deadline;
};

__device__ __forceinline__ gu32_t* bar_word(const GridBar& b, int line) { return b.w + 32 * line; }

// one lane polls one word (relaxed, agent scope); false when the run has been aborted or the deadline passed
__device__ __forceinline__ bool bar_wait(const GridBar& b, gu32_t* p, unsigned target)
{
    unsigned long long t_wait = 0;                     // when THIS wait began (a long factorisation is not a stall: the limit is per wait)
    for (unsigned spins = 0;; ++spins) {
        const unsigned v = __hip_atomic_load(p, R2P_RLX_AGENT);
        if ((int)(v - target) >= 0) return __hip_atomic_load(bar_word(b, R2P_ABORT), R2P_RLX_AGENT) == 0u;   // a late arrival of an abandoned run must not go on
        if ((spins & 63u) == 63u) {
            if (__hip_atomic_load(bar_word(b, R2P_ABORT), R2P_RLX_AGENT) != 0u) return false;
            const unsigned long long now = wall_clock64();
            if (t_wait == 0) t_wait = now;
            else if (now - t_wait > b.deadline) { __hip_atomic_store(bar_word(b, R2P_ABORT), 1u, R2P_RLX_AGENT); return false; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// every thread of every workgroup calls it; returns false when the run must be abandoned
__device__ __forceinline__ bool grid_barrier(GridBar& b, int* sh_ok)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its stores have reached the XCD's L2
    __syncthreads();
    b.epoch += 1;
    if (threadIdx.x == 0) {
        bool ok = true;
        const unsigned e = b.epoch;
        const unsigned a = __hip_atomic_fetch_add(bar_word(b, R2P_XCNT + b.xcc), 1u, R2P_RLX_AGENT) + 1u;
        if (a == e * b.xtot) {                                 // last arriver of this XCD
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(bar_word(b, R2P_TOP), 1u, R2P_RLX_AGENT) + 1u;
            if (t == e * b.nx) __hip_atomic_store(bar_word(b, R2P_TOPGEN), e, R2P_RLX_AGENT);
            else ok = bar_wait(b, bar_word(b, R2P_TOPGEN), e);
            // the members are released only by a barrier that completed; after an abort or a timeout they find the ABORT word
            // themselves (the leader's own wait has set it) instead of running on beside workgroups that have left
            if (ok) __hip_atomic_store(bar_word(b, R2P_XGEN + b.xcc), e, R2P_RLX_AGENT);
        } else {
            ok = bar_wait(b, bar_word(b, R2P_XGEN + b.xcc), e);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *sh_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return *sh_ok != 0;
}

// NV per-thread values -> their sums over the workgroup, in every thread (fixed order: wave DPP sums, then the waves in index order)
template <int NV>
__device__ __forceinline__ void block_sums(double (&v)[NV], double (*shw)[8], double* sho)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const double s = wave_sum(v[c]);
        if (lane == 0) shw[wave][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += shw[w][threadIdx.x];
        sho[threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c] = sho[c];
    __syncthreads();
}

// ---- the two products in "stream" form ------------------------------------------------------------------------------
// A side of the iteration (the columns of A, or of A') is cut into one CONTIGUOUS block of items per workgroup, so the
// block's entries are one contiguous range of the CSC arrays.  A product runs in two steps per chunk of CHUNK entries:
//   1. entry-parallel: lane e of the workgroup takes entries e, e + 1024, ... of the chunk -- value (coalesced), row index
//      (from the workgroup's LDS copy when it fits, else coalesced from memory), the gathered 16-byte row of the factor --
//      and leaves value * row in LDS.  Every gather of the chunk is in flight at once; no lane idles on a short column and
//      none serialises a long one (the gathers issued = the entries stored);
//   2. item-parallel: the lane that owns a column adds up its segment of the products in storage order (the order of the
//      reference's sparse Gemm loops, sparse_gemm_ab_impl.hpp:480-582: fixed, independent of the geometry).
// The column offsets of the block live in LDS for the whole launch (they never change), relative to the block's first entry.
constexpr int R2P_CHUNK = 4096;           // entries per chunk = 4 per lane: 64 KB of products
constexpr int R2P_MAX_ITEMS = 4096;       // items per workgroup and side (4 per lane)

struct Side {
    const unsigned* ri;      // row indices of the side's CSC
    const double* va;
    i64 first;               // first item of this workgroup's block
    int cnt;                 // items in the block
    i64 pb;                  // first entry of the block
    i64 ne;                  // entries in the block
    const unsigned* off;     // LDS: cnt + 1 offsets relative to pb
    const unsigned* idxc;    // LDS copy of the block's row indices, or nullptr
};

// acc[u] = (column first + tid + 1024 u of the side) . X for u < 4; X is N x 2 compact
__device__ __forceinline__ void stream_products(const Side& sd, const double* X, f64x2_t* prod, double (&acc)[4][2])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u][0] = acc[u][1] = 0.0;
    for (i64 c0 = 0; c0 < sd.ne; c0 += R2P_CHUNK) {
        const i64 c1 = c0 + R2P_CHUNK < sd.ne ? c0 + R2P_CHUNK : sd.ne;
        unsigned r[4];
        double v[4];
        bool in[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const i64 e = c0 + tid + 1024 * q;
            in[q] = e < c1;
            r[q] = 0u;
            v[q] = 0.0;
            if (in[q]) {
                r[q] = sd.idxc ? sd.idxc[e] : sd.ri[sd.pb + e];
                v[q] = sd.va[sd.pb + e];
            }
        }
        f64x2_t x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x[q][0] = 0.0; x[q][1] = 0.0;
            if (in[q]) x[q] = *(const f64x2_t*)(X + (i64)r[q] * 2);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (in[q]) { f64x2_t t; t[0] = v[q] * x[q][0]; t[1] = v[q] * x[q][1]; prod[tid + 1024 * q] = t; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = tid + 1024 * u;
            if (it < sd.cnt) {
                const i64 s0 = sd.off[it], s1 = sd.off[it + 1];
                const i64 a = s0 > c0 ? s0 : c0, b = s1 < c1 ? s1 : c1;
                for (i64 p = a; p < b; ++p) { const f64x2_t t = prod[p - c0]; acc[u][0] += t[0]; acc[u][1] += t[1]; }
            }
        }
        __syncthreads();
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char r2p_dyn[];

__global__ __launch_bounds__(1024) void rank2_persist_kernel(R2PersistArgs A)
{
    constexpr int KP = 8;
    __shared__ double shw[16][8];
    __shared__ double sho[8];
    __shared__ int sh_ok;
    __shared__ unsigned sh_u[4];
    const int tid = threadIdx.x, nwg = gridDim.x, wg = blockIdx.x;
    const i64 gthreads = (i64)nwg * blockDim.x, gtid = (i64)wg * blockDim.x + tid;

    // ---- who is where: membership per XCD through one flat barrier ----
    GridBar bar;
    bar.w = (gu32_t*)A.sync;
    bar.epoch = 0;
    // longest single wait, in ticks of the 100 MHz constant clock: 0.25 s for the start-up barrier (whether every workgroup
    // is resident is decided within microseconds of the launch; a grid that does not fit, e.g. under a CU mask, gives up
    // quickly), 5 s afterwards (a wait inside the loop covers a whole phase of the slowest workgroup)
    bar.deadline = 25000000ull;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        __hip_atomic_fetch_add(bar_word(bar, R2P_XTOT + xcc), 1u, R2P_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar_word(bar, R2P_FLAT), 1u, R2P_RLX_AGENT);
        const bool ok = bar_wait(bar, bar_word(bar, R2P_FLAT), (unsigned)nwg);
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += __hip_atomic_load(bar_word(bar, R2P_XTOT + x), R2P_RLX_AGENT) != 0u ? 1u : 0u;
        sh_u[0] = xcc;
        sh_u[1] = __hip_atomic_load(bar_word(bar, R2P_XTOT + xcc), R2P_RLX_AGENT);
        sh_u[2] = nx;
        sh_ok = ok ? 1 : 0;
    }
    __syncthreads();
    bar.xcc = sh_u[0]; bar.xtot = sh_u[1]; bar.nx = sh_u[2];
    bar.deadline = 500000000ull;
    if (!sh_ok) { if (gtid == 0) A.out[0] = (double)R2P_ABORTED; return; }

    // ---- this workgroup's blocks of the two sides; LDS: products | offsets of A | offsets of A' | row-index copies ----
    f64x2_t* prod = (f64x2_t*)r2p_dyn;
    Side sa, st;
    {
        size_t used = (size_t)R2P_CHUNK * sizeof(f64x2_t);
        auto block_of = [&](const i64* cp, const unsigned* ri, const double* va, i64 nitems, Side& sd) {
            const i64 ipw = (nitems + nwg - 1) / nwg;
            sd.ri = ri; sd.va = va;
            sd.first = (i64)wg * ipw;
            i64 cnt = ipw;
            if (sd.first >= nitems) { sd.first = nitems; cnt = 0; } else if (sd.first + cnt > nitems) cnt = nitems - sd.first;
            sd.cnt = (int)cnt;
            sd.pb = cp[sd.first];                                       // cp has nitems + 1 entries: valid also for an empty block
            sd.ne = cp[sd.first + cnt] - sd.pb;
            unsigned* off = (unsigned*)(r2p_dyn + used);
            used += ((size_t)(cnt + 1) * 4 + 15) / 16 * 16;
            for (i64 i = tid; i <= cnt; i += blockDim.x) off[i] = (unsigned)(cp[sd.first + i] - sd.pb);
            sd.off = off;
            sd.idxc = nullptr;
        };
        block_of(A.colptr, A.rowidx, A.val, A.n, sa);
        block_of(A.colptr_t, A.rowidx_t, A.val_t, A.m, st);
        auto cache_idx = [&](Side& sd) {                                // uniform over the workgroup
            const size_t b = ((size_t)sd.ne * 4 + 15) / 16 * 16;
            if (sd.ne == 0 || used + b > A.lds_bytes) return;
            unsigned* idx = (unsigned*)(r2p_dyn + used);
            used += b;
            for (i64 e = tid; e < sd.ne; e += blockDim.x) idx[e] = sd.ri[sd.pb + e];
            sd.idxc = idx;
        };
        cache_idx(sa);
        cache_idx(st);
        __syncthreads();
    }

    // uniform state of the iteration (identical bits in every thread of every workgroup)
    double gw00 = A.Gw0[0], gw01 = A.Gw0[1], gw11 = A.Gw0[KP + 1];       // W'W the H solve uses
    double gh00 = 0.0, gh01 = 0.0, gh11 = 0.0;                           // HH' (scaled once the norms are known)
    double nu0 = 1.0, nu1 = 1.0, inu0 = 1.0, inu1 = 1.0;
    double pg0 = 1.0, metric = 1.0;
    int success_count = 0, status = R2P_RUNNING, fail_tag = 0, count = 0, t = 0;

    // ---- phase H(0): from the products of solver.Init ----
    {
        const R2Solve sv = r2_prepare(gw00, gw01, gw11, 0);
        if (sv.bad) { status = R2P_SOLVER_FAILED; fail_tag = A.iter_tag0; }
        double hs[3] = {0.0, 0.0, 0.0};
        if (!sv.bad)
            for (int it = tid; it < sa.cnt; it += blockDim.x) {
                const i64 j = sa.first + it;
                double x0, x1;
                r2_apply(sv, 0, rhs_elem(A.R1, j, 0), rhs_elem(A.R1, j, 1), x0, x1);
                f64x2_t v; v[0] = x0; v[1] = x1;
                *(f64x2_t*)(A.Hc0 + j * 2) = v;
                hs[0] += x0 * x0; hs[1] += x0 * x1; hs[2] += x1 * x1;
            }
        block_sums<3>(hs, shw, sho);
        if (tid < 3) A.gp_h[(i64)wg * 8 + tid] = hs[tid];
    }

    // where the time goes, as seen by workgroup 0 (100 MHz ticks): waiting in B1 / phase W / waiting in B2 / phase G
    unsigned long long tk[4] = {0, 0, 0, 0}, tk0 = wall_clock64();
    auto lap = [&](int slot) { const unsigned long long now = wall_clock64(); tk[slot] += now - tk0; tk0 = now; };
    while (status == R2P_RUNNING) {
        double* Hcur = (t & 1) ? A.Hc1 : A.Hc0;
        double* Hnext = (t & 1) ? A.Hc0 : A.Hc1;
        // ---- B1: HH' of H(t); the stopping rule of iteration t - 1 ----
        if (!grid_barrier(bar, &sh_ok)) { status = R2P_ABORTED; break; }
        lap(0);
        {
            double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            if (tid < nwg) {
                const double* g = A.gp_h + (i64)tid * 8;
                v[0] = g[0]; v[1] = g[1]; v[2] = g[2];
                if (t > 0) { const double* q = A.pgp + (i64)tid * 8; v[3] = q[0]; v[4] = q[1]; }
            }
            block_sums<5>(v, shw, sho);
            if (t > 0) {
                const int p = t - 1;                     // NmfSolve<>: iteration 0 only initialises the estimator, then from min_iter on
                if (p == 0 || p >= A.min_iter) {
                    const double pg = sqrt(v[3] + v[4]);
                    if (pg != pg) { status = R2P_NAN; count = p; break; }
                    if (p == 0) { pg0 = pg; metric = 1.0; } else metric = pg / pg0;
                    if (p >= A.min_iter) {
                        if (metric <= A.tol) { if (++success_count >= A.tolcount) { status = R2P_CONVERGED; count = p; break; } }
                        else success_count = 0;
                    }
                }
            }
            gh00 = v[0]; gh01 = v[1]; gh11 = v[2];
        }
        // ---- phase W(t): (AH')' for this workgroup's rows, their solves ----
        {
            const R2Solve sv = r2_prepare(gh00, gh01, gh11, 1);
            if (sv.bad) { status = R2P_SOLVER_FAILED; fail_tag = A.iter_tag0 + t; break; }
            double acc[4][2];
            stream_products(st, Hcur, prod, acc);
            double ws[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = tid + 1024 * u;
                if (it < st.cnt) {
                    const i64 i = st.first + it;
                    double x0, x1;
                    r2_apply(sv, 1, acc[u][0], acc[u][1], x0, x1);
                    f64x2_t w; w[0] = x0; w[1] = x1;
                    f64x2_t r; r[0] = acc[u][0]; r[1] = acc[u][1];
                    *(f64x2_t*)(A.Wc + i * 2) = w;
                    *(f64x2_t*)(A.R2c + i * 2) = r;
                    ws[0] += x0 * x0; ws[1] += x0 * x1; ws[2] += x1 * x1;
                }
            }
            block_sums<3>(ws, shw, sho);
            if (tid < 3) A.gp_w[(i64)wg * 8 + tid] = ws[tid];
        }
        lap(1);
        // ---- B2: W'W of the W just solved; its column norms ----
        if (!grid_barrier(bar, &sh_ok)) { status = R2P_ABORTED; break; }
        lap(2);
        {
            double v[3] = {0.0, 0.0, 0.0};
            if (tid < nwg) { const double* g = A.gp_w + (i64)tid * 8; v[0] = g[0]; v[1] = g[1]; v[2] = g[2]; }
            block_sums<3>(v, shw, sho);
            // rank2_normalize_kernel's arithmetic, value for value
            nu0 = sqrt(v[0]); nu1 = sqrt(v[2]);
            const bool ok0 = !(fabs(nu0) < DBL_EPSILON), ok1 = !(fabs(nu1) < DBL_EPSILON);
            if (!ok0 || !ok1) { status = R2P_SOLVER_FAILED; fail_tag = -2; break; }
            inu0 = 1.0 / nu0; inu1 = 1.0 / nu1;
            gh00 *= nu0 * nu0; gh01 *= nu0 * nu1; gh11 *= nu1 * nu1;
            gw00 = v[0] / (nu0 * nu0); gw01 = v[1] / (nu0 * nu1); gw11 = v[2] / (nu1 * nu1);
        }
        // ---- phase G(t): W'A of the normalised W, both projected-gradient sums, and H(t + 1) from the same registers ----
        {
            double ps[2] = {0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {                                // W side (projected_gradient.hpp:125-171): this workgroup's rows
                const int it = tid + 1024 * u;
                if (it < st.cnt) {
                    const i64 i = st.first + it;
                    f64x2_t w = *(const f64x2_t*)(A.Wc + i * 2);
                    f64x2_t r = *(const f64x2_t*)(A.R2c + i * 2);
                    w[0] *= inu0; w[1] *= inu1;
                    r[0] *= nu0; r[1] *= nu1;
                    const double g0 = (gh00 * w[0] + gh01 * w[1]) - r[0];
                    const double g1 = (gh01 * w[0] + gh11 * w[1]) - r[1];
                    if (g0 < 0.0 || w[0] > 0.0) ps[0] += g0 * g0;
                    if (g1 < 0.0 || w[1] > 0.0) ps[0] += g1 * g1;
                }
            }
            const bool more = t + 1 < A.max_iter;
            const R2Solve sv = r2_prepare(gw00, gw01, gw11, 0);
            if (more && sv.bad) { status = R2P_SOLVER_FAILED; fail_tag = A.iter_tag0 + t + 1; break; }
            double acc[4][2];
            stream_products(sa, A.Wc, prod, acc);
            double hs[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = tid + 1024 * u;
                if (it < sa.cnt) {
                    const i64 j = sa.first + it;
                    const double r0 = acc[u][0] * inu0, r1 = acc[u][1] * inu1;
                    f64x2_t h = *(const f64x2_t*)(Hcur + j * 2);
                    h[0] *= nu0; h[1] *= nu1;
                    const double g0 = (gw00 * h[0] + gw01 * h[1]) - r0;
                    const double g1 = (gw01 * h[0] + gw11 * h[1]) - r1;
                    if (g0 < 0.0 || h[0] > 0.0) ps[1] += g0 * g0;
                    if (g1 < 0.0 || h[1] > 0.0) ps[1] += g1 * g1;
                    if (more) {
                        double x0, x1;
                        r2_apply(sv, 0, r0, r1, x0, x1);
                        f64x2_t v; v[0] = x0; v[1] = x1;
                        *(f64x2_t*)(Hnext + j * 2) = v;
                        hs[0] += x0 * x0; hs[1] += x0 * x1; hs[2] += x1 * x1;
                    }
                }
            }
            double all[5] = {ps[0], ps[1], hs[0], hs[1], hs[2]};
            block_sums<5>(all, shw, sho);
            if (tid < 2) A.pgp[(i64)wg * 8 + tid] = all[tid];
            if (tid >= 2 && tid < 5) A.gp_h[(i64)wg * 8 + (tid - 2)] = all[tid];
        }
        lap(3);
        t += 1;
        if (t == A.max_iter) {
            // the last iteration's rule: nothing runs after it (nmf_solve_generic.hpp: reaching max_iter is success)
            if (!grid_barrier(bar, &sh_ok)) { status = R2P_ABORTED; break; }
            const int p = t - 1;
            status = R2P_EXHAUSTED; count = A.max_iter;
            if (p == 0 || p >= A.min_iter) {
                double v[2] = {0.0, 0.0};
                if (tid < nwg) { const double* q = A.pgp + (i64)tid * 8; v[0] = q[0]; v[1] = q[1]; }
                block_sums<2>(v, shw, sho);
                const double pg = sqrt(v[0] + v[1]);
                if (pg != pg) { status = R2P_NAN; count = p; break; }
                if (p == 0) { pg0 = pg; metric = 1.0; } else metric = pg / pg0;
                if (p >= A.min_iter && metric <= A.tol && ++success_count >= A.tolcount) { status = R2P_CONVERGED; count = p; }
            }
            break;
        }
    }

    // ---- epilogue: the state of the iteration the run ended on, in the solver's layout ----
    const int performed = (status == R2P_CONVERGED) ? count + 1 : t;      // iterations whose result is kept
    if (status == R2P_CONVERGED || status == R2P_EXHAUSTED) {
        const double* Hfin = ((performed - 1) & 1) ? A.Hc1 : A.Hc0;
        for (i64 i = gtid; i < A.m; i += gthreads) {
            f64x2_t w = *(const f64x2_t*)(A.Wc + i * 2);
            w[0] *= inu0; w[1] *= inu1;
            *(f64x2_t*)(A.Wt + i * KP) = w;
        }
        for (i64 j = gtid; j < A.n; j += gthreads) {
            f64x2_t h = *(const f64x2_t*)(Hfin + j * 2);
            h[0] *= nu0; h[1] *= nu1;
            *(f64x2_t*)(A.H + j * KP) = h;
        }
        if (wg == 0 && tid < KP * KP) {
            const int e = tid;
            A.Gw[e] = (e == 0) ? gw00 : (e == 1 || e == KP) ? gw01 : (e == KP + 1) ? gw11 : 0.0;
        }
    }
    if (gtid == 0) {
        if (status == R2P_SOLVER_FAILED) atomicMin(A.fail_flag, fail_tag);
        A.out[1] = (double)count;
        A.out[2] = (double)performed;
        A.out[3] = pg0;
        A.out[4] = metric;
        A.out[5] = (double)fail_tag;
        for (int q = 0; q < 4; ++q) A.out[8 + q] = (double)tk[q] * 0.01;     // us
        // a run that any workgroup abandoned is ABORTED whatever this workgroup saw at its last barrier: the others never
        // wrote their rows of Wt / H (the host then repeats the run on the launch-per-kernel path from the state Init left)
        if ((status == R2P_CONVERGED || status == R2P_EXHAUSTED) && __hip_atomic_load(bar_word(bar, R2P_ABORT), R2P_RLX_AGENT) != 0u)
            status = R2P_ABORTED;
        A.out[0] = (double)status;
    }
}

constexpr size_t R2P_LDS_BYTES = 150 * 1024;      // of the CU's 160 KB (static LDS: ~1.2 KB): 64 KB of products, the offsets, the row-index copies
size_t rank2_persist_lds_bytes()
{
    static const int small = [] { const char* e = getenv("SMK_R2P_LDS"); return e && atoi(e) == 0 ? 1 : 0; }();
    // SMK_R2P_LDS=0: products and offsets only (no room for a row-index copy)
    return small ? (size_t)R2P_CHUNK * 16 + 2 * ((size_t)(R2P_MAX_ITEMS + 1) * 4 + 16) : R2P_LDS_BYTES;
}
size_t rank2_persist_sync_bytes() { return (size_t)R2P_WORDS * 32 * sizeof(unsigned); }

// workgroups of 1024 threads: about two entries per lane and chunk, no more than R2P_MAX_ITEMS items per workgroup and
// side, at most one workgroup per CU.  0: the matrix does not fit this geometry (the caller takes the other path).
int rank2_persist_workgroups(i64 m, i64 n, i64 nnz, int num_cus)
{
    static const int cap = [] { const char* e = getenv("SMK_R2P_WGS"); return e ? atoi(e) : 0; }();
    const i64 top = cap > 0 ? cap : num_cus;
    const i64 longer = m > n ? m : n;
    i64 wgs = std::max((nnz + 2047) / 2048, (longer + 1023) / 1024);
    if (wgs > top) wgs = top;
    if (wgs < 1) wgs = 1;
    if ((longer + wgs - 1) / wgs > R2P_MAX_ITEMS) return 0;
    return (int)wgs;
}

int launch_rank2_persist(const R2PersistArgs& a, int workgroups, hipStream_t st)
{
    SMK_HIP(hipMemsetAsync(a.sync, 0, rank2_persist_sync_bytes(), st));
    SMK_HIP(hipMemsetAsync(a.out, 0, 16 * sizeof(double), st));
    static std::atomic<unsigned long long> attr_set{0};       // per device (DeviceOnce)
    if (DeviceOnce once{attr_set}) {
        SMK_HIP(hipFuncSetAttribute((const void*)rank2_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R2P_LDS_BYTES));
        once.done();
    }
    {   // every workgroup has to be resident at once (two grid barriers per iteration): a grid the occupancy calculator does
        // not place is refused here (the caller takes the launch-per-kernel path) rather than after the start-up deadline
        int per_cu = 0, dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)rank2_persist_kernel, 1024, a.lds_bytes) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && (i64)per_cu * cus < workgroups)
            return 1;      // not an error: "does not fit"
    }
    rank2_persist_kernel<<<workgroups, 1024, a.lds_bytes, st>>>(a);
    SMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace smk
