// smallk_amd/csrc/spmm_seg.hip -- the gather product of sparse NMF at ranks 3 .. 128 (MU / HALS / BPP on CSC A):
//   out[:, j] = sum over the stored entries p of column j of val[p] * X[:, row[p]]
// = the reference's sparse Gemm variants in gather form (common/include/sparse_gemm_ab_impl.hpp:24-100, :480-582 for W'A with
// B = A; sparse_gemm_ba_impl.hpp:25-99 for (AH')' with B = A').
//
// Work is cut by STORED ENTRIES, not by columns: the columns of B are packed, in storage order, into segments of at most SEG
// consecutive entries -- a run of whole columns, or one piece of a column longer than SEG -- once per matrix (SegPlan, built on
// first use; the structure of A does not change between iterations).  A group of KP/2 lanes owns a segment: 16 bytes of the
// gathered row per lane, U gathers in flight per lane, the values and row indices of its entries read as broadcast loads
// (consecutive addresses, L1 lines reused 16 / 32 times), the running sum kept in two registers and written to P when the
// row index carries the "last entry of its column" flag (bit 31 of a flagged copy of the row indices; row counts are below
// 2^30, solver.cpp checks m k < 2^31).  Sums run in storage order inside a column, as in the reference's loops.  Every group
// handles <= SEG entries whatever the column lengths are: a term that occurs in 8000 documents is 125 pieces, summed in piece
// order by spmm_seg_fixup_kernel, not one lane group working 8000 entries while the rest of the chip idles.
// The round-4 kernel this replaces (kernels.hip: spmm_gather_kernel, kept for SMK_SPMM_SEG=0) gave a column to KP/4 lanes and
// walked its entries one dependent gather at a time.
#include <vector>

#include "common.h"
#include "devutil.h"
#include "gram_inverse.h"
#include "gram_body.h"

namespace smk {

static constexpr unsigned LAST_FLAG = 0x80000000u;
static constexpr unsigned NO_PIECE = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void seg_flag_rows_kernel(const i64* __restrict__ colptr, i64 ncols,
                                                            const unsigned* __restrict__ rowidx, i64 nnz,
                                                            unsigned* __restrict__ rowflag, int phase)
{
    const i64 stride = (i64)gridDim.x * blockDim.x;
    if (phase == 0) {
        for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride) rowflag[p] = rowidx[p];
    } else {
        for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < ncols; j += stride) {
            const i64 a = colptr[j], b = colptr[j + 1];
            if (b > a) rowflag[b - 1] |= LAST_FLAG;
        }
    }
}

// one segment per group of KP/2 lanes
template <int KP, int U, bool EMPTY>
__global__ __launch_bounds__(256) void spmm_seg_kernel(const i64* __restrict__ seg_p0, const unsigned* __restrict__ seg_len,
                                                       const unsigned* __restrict__ seg_col,
                                                       const unsigned* __restrict__ seg_piece, i64 nseg,
                                                       const i64* __restrict__ colptr, const unsigned* __restrict__ rowflag,
                                                       const double* __restrict__ val, const double* __restrict__ X,
                                                       double* __restrict__ P, int kpp, double* __restrict__ pieces, InvRide ride,
                                                       const double* __restrict__ gram_x, i64 gram_n, i64 gram_cpw, int gram_nblk,
                                                       double* __restrict__ gram_gp)
{
    constexpr int LPC = KP / 2;
    constexpr int GPB = 256 / LPC;
    i64 blk = blockIdx.x;
    if constexpr (KP == 16 || KP == 32) {
        // the first gram_nblk workgroups of a launch that carries a Gram matrix form its partial sums (common.h: GramRide)
        if (gram_x) {
            if (blk < gram_nblk) {
                __shared__ double red[KP * KP];
                gram_mfma_body<KP>(gram_x, gram_n, gram_cpw, gram_gp, blk, red);
                return;
            }
            blk -= gram_nblk;
        }
    }
    if constexpr (KP == 32 || KP == 64) {
        // workgroup 0 of a launch that carries the Gram inverse (common.h: InvRide) inverts; the product starts at workgroup 1
        if (ride.G) {
            if (blk == 0) {
                __shared__ __attribute__((aligned(16))) double inv_lds[GRAM_INVERSE_LDS(KP)];
                gram_inverse64_body<KP>(ride.G, ride.k, ride.Ginv, (int*)(ride.Ginv + KP * KP), inv_lds);
                return;
            }
            --blk;
        }
    }
    const i64 sg = blk * GPB + threadIdx.x / LPC;
    const int l = threadIdx.x % LPC;
    if (sg >= nseg) return;
    const i64 p0 = seg_p0[sg];
    const unsigned len = seg_len[sg];
    unsigned j = seg_col[sg];
    const unsigned piece = seg_piece[sg];
    const unsigned* __restrict__ rf = rowflag + p0;
    const double* __restrict__ vv = val + p0;
    const double* __restrict__ Xl = X + 2 * l;
    double a0 = 0.0, a1 = 0.0;
    for (unsigned e = 0; e < len; e += U) {
        unsigned ri[U];
        double v[U];
        f64x2_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned q = e + u < len ? e + u : len - 1;
            ri[u] = rf[q];
            v[u] = vv[q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = *(const f64x2_t*)(Xl + (i64)(ri[u] & ~LAST_FLAG) * KP);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + u < len) {
                a0 = __builtin_fma(v[u], x[u][0], a0);
                a1 = __builtin_fma(v[u], x[u][1], a1);
                if ((ri[u] & LAST_FLAG) && piece == NO_PIECE) {
                    if (2 * l < kpp) { f64x2_t r; r[0] = a0; r[1] = a1; *(f64x2_t*)(P + (i64)j * kpp + 2 * l) = r; }
                    a0 = a1 = 0.0;
                    ++j;
                    if constexpr (EMPTY) {            // columns without stored entries: the next entry belongs to a later column
                        if (e + u + 1 < len) { const i64 nxt = p0 + e + u + 1; while (colptr[j + 1] <= nxt) ++j; }
                    }
                }
            }
        }
    }
    if (piece != NO_PIECE) { f64x2_t r; r[0] = a0; r[1] = a1; *(f64x2_t*)(pieces + (i64)piece * KP + 2 * l) = r; }
}

// columns longer than a segment: a wave per column; the 64 / LPC lane groups take every (64 / LPC)-th piece each with four
// accumulators (all loads of a step in flight at once: the longest column of the Reuters shape is 123 pieces, which one
// lane group adding them one dependent load at a time took 20 us over), then the partial sums are joined in a fixed order
// (accumulator 0..3 of group 0, then group 1, ...): the same bits on every run
template <int KP>
__global__ __launch_bounds__(256) void spmm_seg_fixup_kernel(const unsigned* __restrict__ long_col,
                                                             const i64* __restrict__ long_piece0, i64 nlong,
                                                             const double* __restrict__ pieces, double* __restrict__ P, int kpp,
                                                             const double* __restrict__ gram_gp, int gram_nblk, double* __restrict__ gram_g)
{
    // the first KP * KP / 16 workgroups of a launch that carries a Gram matrix add up its partial sums (common.h: GramRide)
    i64 fblk = blockIdx.x;
    if (gram_gp) {
        constexpr int RB = KP * KP / 16;
        if (fblk < RB) {
            __shared__ double sh[16][17];
            gram_reduce_body(gram_gp, gram_nblk, KP * KP, gram_g, 0, nullptr, nullptr, 1.0, (int)fblk, sh);
            return;
        }
        fblk -= RB;
    }
    constexpr int LPC = KP / 2;
    constexpr int G = 64 / LPC;
    __shared__ double part[4][G][KP];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const i64 c = fblk * 4 + w;
    if (c >= nlong) return;
    const int g = lane / LPC, l = lane % LPC;
    const i64 q0 = long_piece0[c], q1 = long_piece0[c + 1];
    double a[4][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    for (i64 q = q0 + g; q < q1; q += 4 * G) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (q + (i64)u * G < q1) {
                const f64x2_t t = *(const f64x2_t*)(pieces + (q + (i64)u * G) * KP + 2 * l);
                a[u][0] += t[0];
                a[u][1] += t[1];
            }
        }
    }
    part[w][g][2 * l] = (a[0][0] + a[1][0]) + (a[2][0] + a[3][0]);
    part[w][g][2 * l + 1] = (a[0][1] + a[1][1]) + (a[2][1] + a[3][1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes have landed (one wave owns part[w]; LDS
    __builtin_amdgcn_wave_barrier();                        // operations of a wave execute in order)
    if (g == 0 && 2 * l < kpp) {
        double r0 = 0.0, r1 = 0.0;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) { r0 += part[w][gg][2 * l]; r1 += part[w][gg][2 * l + 1]; }
        f64x2_t r; r[0] = r0; r[1] = r1;
        *(f64x2_t*)(P + (i64)long_col[c] * kpp + 2 * l) = r;
    }
}

void free_seg_plan(SegPlan* s)
{
    void* bufs[] = {s->seg_p0, s->seg_len, s->seg_col, s->seg_piece, s->rowflag, s->long_col, s->long_piece0, s->pieces};
    for (void* b : bufs)
        if (b) (void)dev_free(b);
    *s = SegPlan();
}

template <typename T>
static int upload_vec(T** dst, const std::vector<T>& v, hipStream_t st)
{
    *dst = nullptr;
    SMK_HIP(dev_malloc((void**)dst, (v.empty() ? 1 : v.size()) * sizeof(T)));
    if (!v.empty()) SMK_HIP(hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return 0;
}

int spmm_seg_len()
{
    static const int seg = [] { const char* e = getenv("SMK_SPMM_SEG_LEN"); const int v = e ? atoi(e) : 64; return v < 8 ? 8 : v > 4096 ? 4096 : v; }();
    return seg;
}

int build_seg_plan(i64 ncols, i64 nnz, const i64* colptr, const unsigned* rowidx, SegPlan* out, hipStream_t st)
{
    free_seg_plan(out);
    const int SEG = spmm_seg_len();
    std::vector<i64> cp((size_t)ncols + 1);
    SMK_HIP(hipMemcpyAsync(cp.data(), colptr, cp.size() * sizeof(i64), hipMemcpyDeviceToHost, st));
    SMK_HIP(hipStreamSynchronize(st));
    const i64 base = cp[0];
    if (base != 0) { set_error("build_seg_plan: column offsets must start at 0"); return -100; }
    std::vector<i64> p0;
    std::vector<unsigned> len, col, piece, lcol;
    std::vector<i64> lp0;
    i64 npieces = 0;
    bool empty = false;
    i64 j = 0;
    while (j < ncols) {
        const i64 cl = cp[j + 1] - cp[j];
        if (cl == 0) { empty = true; ++j; continue; }
        if (cl > SEG) {                                   // one column, several pieces
            lcol.push_back((unsigned)j);
            lp0.push_back(npieces);
            for (i64 q = cp[j]; q < cp[j + 1]; q += SEG) {
                p0.push_back(q - base);
                len.push_back((unsigned)std::min<i64>(SEG, cp[j + 1] - q));
                col.push_back((unsigned)j);
                piece.push_back((unsigned)npieces++);
            }
            ++j;
            continue;
        }
        i64 jb = j + 1;                                   // whole columns while they fit
        while (jb < ncols && cp[jb + 1] - cp[jb] <= SEG && cp[jb + 1] - cp[j] <= SEG) ++jb;
        while (jb > j + 1 && cp[jb] == cp[jb - 1]) --jb;  // do not end on empty columns (the walk stops at the last entry)
        p0.push_back(cp[j] - base);
        len.push_back((unsigned)(cp[jb] - cp[j]));
        col.push_back((unsigned)j);
        piece.push_back(NO_PIECE);
        for (i64 t = j; t < jb; ++t) if (cp[t + 1] == cp[t]) empty = true;
        j = jb;
    }
    lp0.push_back(npieces);
    out->nseg = (i64)p0.size();
    out->nlong = (i64)lcol.size();
    out->npieces = npieces;
    out->has_empty = empty;
    {   // columns of nearly equal length (a k-nearest-neighbour or fixed-degree graph): nothing to balance, and the kernel that
        // gives every column its own lane group then runs 7 % faster (1 M nodes, degree 16 +- 4: 627 us against 670,
        // profiles/r05_spmm_segment_sweep.txt) -- launch_spmm_seg's callers keep it for such a matrix
        i64 longest = 0;
        for (i64 c = 0; c < ncols; ++c) longest = std::max(longest, cp[c + 1] - cp[c]);
        const double avg = ncols > 0 ? (double)nnz / (double)ncols : 0.0;
        out->uniform = avg >= 4.0 && (double)longest <= 4.0 * avg && longest <= 256;
        out->longest = longest;
    }
    out->ncols = ncols;
    out->nnz = nnz;
    int rc = 0;
    rc |= upload_vec(&out->seg_p0, p0, st);
    rc |= upload_vec(&out->seg_len, len, st);
    rc |= upload_vec(&out->seg_col, col, st);
    rc |= upload_vec(&out->seg_piece, piece, st);
    rc |= upload_vec(&out->long_col, lcol, st);
    rc |= upload_vec(&out->long_piece0, lp0, st);
    if (!rc && dev_malloc((void**)&out->rowflag, (size_t)std::max<i64>(nnz, 1) * sizeof(unsigned)) != hipSuccess) rc = -100;
    if (!rc && dev_malloc((void**)&out->pieces, (size_t)std::max<i64>(npieces, 1) * 128 * sizeof(double)) != hipSuccess) rc = -100;
    if (rc) { free_seg_plan(out); set_error("build_seg_plan: device allocation failed"); return -100; }
    if (nnz > 0) {
        const int grid = (int)std::min<i64>((nnz + 255) / 256, 4096);
        seg_flag_rows_kernel<<<grid, 256, 0, st>>>(colptr, ncols, rowidx, nnz, out->rowflag, 0);
        seg_flag_rows_kernel<<<grid, 256, 0, st>>>(colptr, ncols, rowidx, nnz, out->rowflag, 1);
    }
    SMK_HIP(hipGetLastError());
    SMK_HIP(hipStreamSynchronize(st));            // the host vectors go out of scope
    return 0;
}

int launch_spmm_seg(const SegPlan& sp, const i64* colptr, const double* val, const double* X, int k, double* P, int kpp,
                    hipStream_t st, double* pieces, const InvRide* ride_in, const GramRide* gram_in)
{
    const int KPv = kp_of(k);
    if (!pieces) pieces = sp.pieces;
    if (sp.ncols <= 0) return 0;
    if (sp.has_empty) SMK_HIP(hipMemsetAsync(P, 0, (size_t)sp.ncols * kpp * sizeof(double), st));
    if (sp.nseg == 0) return 0;
    const int gpb = 256 / (KPv / 2);
    InvRide ride;
    if (ride_in && ride_in->G && (KPv == 32 || KPv == 64)) ride = *ride_in;
    GramRide gram;
    int gnblk = 0;
    i64 gcpw = 0;
    if (gram_in && gram_in->X && gram_in->Gp && gram_in->G && (KPv == 16 || KPv == 32) && !ride.G) {
        gram = *gram_in;
        gram_partial_shape(gram.N, gram.max_blocks, &gnblk, &gcpw);
    }
    const unsigned grid = (unsigned)((sp.nseg + gpb - 1) / gpb) + (ride.G ? 1u : 0u) + (unsigned)gnblk;
    const double* v = val;
    static const int ufix = [] { const char* e = getenv("SMK_SPMM_SEG_U"); return e ? atoi(e) : 0; }();
#define SMK_SEG(U)                                                                                                                  \
    KP_DISPATCH128(KPv, (sp.has_empty ? spmm_seg_kernel<KP, U, true><<<grid, 256, 0, st>>>(sp.seg_p0, sp.seg_len, sp.seg_col, sp.seg_piece, sp.nseg, colptr, sp.rowflag, v, X, P, kpp, pieces, ride, gram.X, gram.N, gcpw, gnblk, gram.Gp) \
                                      : spmm_seg_kernel<KP, U, false><<<grid, 256, 0, st>>>(sp.seg_p0, sp.seg_len, sp.seg_col, sp.seg_piece, sp.nseg, colptr, sp.rowflag, v, X, P, kpp, pieces, ride, gram.X, gram.N, gcpw, gnblk, gram.Gp)))
    if (ufix == 4) { SMK_SEG(4); } else if (ufix == 16) { SMK_SEG(16); } else { SMK_SEG(8); }
#undef SMK_SEG
    SMK_HIP(hipGetLastError());
    if (sp.nlong > 0) {
        // (the reduction of the Gram partial sums rides here: KP * KP / 16 more workgroups)
        const unsigned g2 = (unsigned)((sp.nlong + 3) / 4) + (gram.X ? (unsigned)(KPv * KPv / 16) : 0u);
        KP_DISPATCH128(KPv, (spmm_seg_fixup_kernel<KP><<<g2, 256, 0, st>>>(sp.long_col, sp.long_piece0, sp.nlong, pieces, P, kpp,
                                                                            gram.X ? gram.Gp : nullptr, gnblk, gram.G)));
        SMK_HIP(hipGetLastError());
    } else if (gram.X) {
        const int rrc = launch_gram_reduce(gram.Gp, gnblk, k, gram.G, st, nullptr, nullptr, 1.0);
        if (rrc) return rrc;
    }
    return (ride.G ? 1 : 0) | (gram.X ? 2 : 0);
}

}  // namespace smk
