// smallk_amd/csrc/spmm_blocked.hip -- the rank-2 gather product for factors that do not fit one L2.
//
// out[:, j] = sum_p val[p] * X[row[p], :] gathers a 16-byte row of X per stored entry.  With X larger than the 4 MB L2
// of an XCD (1 M rows = 16 MB) nearly every gather is a line fetched across the fabric from the Infinity Cache / HBM: the
// product of a 16 M-entry matrix runs at the rate the fabric delivers 64 - 128 byte lines (245 us with 16 lanes per
// column), not at the rate the entries stream (192 MB: 30 us).  Here the ROWS of X are cut into nb <= 4 blocks of
// <= 256 K rows (4 MB of X), the stored entries are regrouped once per matrix by row block (a stable radix sort on the
// block number keeps the column order inside a block, so each block is a CSC of its own), and block b is processed by the
// workgroups the dispatcher places on XCD b mod 8: that XCD gathers from a slice of X that stays in ITS L2 while the
// entries stream past.  Every (block, column) pair leaves a partial sum; the consumers of the product add the nb slabs
// (they read the right-hand side through a PartialView with S slabs anyway).
#include "devutil.h"

#include <hipcub/hipcub.hpp>

namespace smk {

namespace {

__global__ __launch_bounds__(256) void bl_keys_kernel(const i64* __restrict__ colptr, const unsigned* __restrict__ rowidx, i64 ncols,
                                                      unsigned shift, unsigned* __restrict__ key, unsigned* __restrict__ col_of,
                                                      unsigned* __restrict__ pos)
{
    // one wave per column: block number of every entry, its column, its position
    const i64 wave = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const i64 nw = ((i64)gridDim.x * 256) >> 6;
    for (i64 j = wave; j < ncols; j += nw)
        for (i64 p = colptr[j] + lane; p < colptr[j + 1]; p += 64) {
            key[p] = rowidx[p] >> shift;
            col_of[p] = (unsigned)j;
            pos[p] = (unsigned)p;
        }
}

__global__ __launch_bounds__(256) void bl_gather_kernel(const unsigned* __restrict__ perm, const unsigned* __restrict__ col_of,
                                                        const unsigned* __restrict__ rowidx, const double* __restrict__ val, i64 nnz,
                                                        unsigned* __restrict__ col_b, unsigned* __restrict__ ri_b, double* __restrict__ va_b)
{
    for (i64 q = (i64)blockIdx.x * 256 + threadIdx.x; q < nnz; q += (i64)gridDim.x * 256) {
        const unsigned p = perm[q];
        col_b[q] = col_of[p];
        ri_b[q] = rowidx[p];
        va_b[q] = val[p];
    }
}

// cp[b][j] = first position q of block b (positions [boff(b), boff(b + 1)) of the sorted order) whose column is >= j
__global__ __launch_bounds__(256) void bl_offsets_kernel(const unsigned* __restrict__ key_sorted, const unsigned* __restrict__ col_b,
                                                         i64 nnz, i64 ncols, int nb, i64* __restrict__ cp)
{
    const i64 total = (i64)nb * (ncols + 1);
    for (i64 t = (i64)blockIdx.x * 256 + threadIdx.x; t < total; t += (i64)gridDim.x * 256) {
        const int b = (int)(t / (ncols + 1));
        const i64 j = t % (ncols + 1);
        // lexicographic lower bound of (b, j) in the sorted (key, column) sequence
        i64 lo = 0, hi = nnz;
        while (lo < hi) {
            const i64 mid = (lo + hi) >> 1;
            const unsigned kb = key_sorted[mid];
            const bool less = kb < (unsigned)b || (kb == (unsigned)b && (i64)col_b[mid] < j);
            if (less) lo = mid + 1; else hi = mid;
        }
        cp[t] = lo;
    }
}

// LPC lanes per (block, column) segment, U segments per lane group in flight at once; workgroup w serves block
// (w mod 8) mod nb -- the XCD it is dispatched to.  A segment is short (nnz / columns / blocks: 2 - 4 entries), so a lane
// has ONE entry of it: offsets -> (value, row) -> gathered row are three dependent round trips, and with one segment per
// lane group the wave spends them waiting (81 % of its cycles, SQ_WAIT_ANY / SQ_WAVE_CYCLES).  With U segments the three
// trips are shared by U independent chains.  (Measured, tools/gpu_round3_j.sh: U = 2 / 4 are 2 - 5 % SLOWER than U = 1 at every
// lane count, while more lanes per segment always help (753 / 564 / 514 us per root-sized iteration at 1 / 2 / 4 lanes): the
// kernel is bound by how the value / index reads coalesce, not by the depth of a lane's dependency chain.  U = 1 is the default.)
template <int LPC, int U>
__global__ __launch_bounds__(256) void spmm_blocked2_kernel(const i64* __restrict__ cp, const unsigned* __restrict__ ri,
                                                            const double* __restrict__ va, i64 ncols, i64 ncols_pad, int nb,
                                                            const double* __restrict__ X, double* __restrict__ P)
{
    constexpr int CPW = 256 / LPC;                      // columns per workgroup and step
    const int xcd = blockIdx.x & 7;
    const int b = xcd % nb;
    const i64 tile = (i64)(blockIdx.x >> 3) * (8 / nb) + xcd / nb;
    const int l = threadIdx.x % LPC;
    const i64* cpb = cp + (i64)b * (ncols + 1);
    i64 j[U], p[U], p1[U];
    double a0[U], a1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        j[u] = (tile * U + u) * CPW + threadIdx.x / LPC;
        a0[u] = a1[u] = 0.0;
        p[u] = p1[u] = 0;
        if (j[u] < ncols) { p[u] = cpb[j[u]] + l; p1[u] = cpb[j[u] + 1]; }
    }
    bool more = true;
    while (more) {
        double v[U];
        unsigned r[U];
        bool have[U];
        more = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            have[u] = p[u] < p1[u];
            v[u] = have[u] ? __builtin_nontemporal_load(va + p[u]) : 0.0;
            r[u] = have[u] ? __builtin_nontemporal_load(ri + p[u]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (have[u]) {
                const f64x2_t x = *(const f64x2_t*)(X + (i64)r[u] * 2);
                a0[u] += v[u] * x[0];
                a1[u] += v[u] * x[1];
            }
            p[u] += LPC;
            more |= p[u] < p1[u];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const double s0 = group_sum<LPC>(a0[u]), s1 = group_sum<LPC>(a1[u]);      // whole waves take part in the DPP moves
        if (l == 0 && j[u] < ncols) {
            f64x2_t o;
            o[0] = s0;
            o[1] = s1;
            *(f64x2_t*)(P + ((i64)b * ncols_pad + j[u]) * 2) = o;
        }
    }
}

}  // namespace

void free_blocked_csc(BlockedCsc* b)
{
    if (!b) return;
    if (b->cp) (void)smk::dev_free(b->cp);
    if (b->ri) (void)smk::dev_free(b->ri);
    if (b->va) (void)smk::dev_free(b->va);
    *b = BlockedCsc();
}

// number of row blocks for a gathered factor of `rows` rows (16 bytes each): 1 while it fits an L2 beside the streams
int blocked_csc_blocks(i64 rows)
{
    static const int forced = [] { const char* e = getenv("SMK_SPMM_BLOCKS"); return e ? atoi(e) : 0; }();
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
    // measured on a 1 M x 1 M, 16 M-entry matrix (tools/gpu_round3_e.sh): the product itself 244 us unblocked, 191 us with
    // 4 blocks, 215 us with 8 -- but every consumer of the result reads one slab per block (solve 26 -> 31 -> 54 us,
    // normalise 49 -> 62 -> 84 us), so 4 blocks of <= 4 MB win (533 us per iteration against 622) and 8 lose (659)
    const i64 bytes = rows * 16;
    if (bytes <= ((i64)6 << 20)) return 1;
    int nb = 2;
    while (nb < 4 && bytes / nb > ((i64)4 << 20)) nb *= 2;
    return nb;
}

int build_blocked_csc(i64 rows, i64 ncols, i64 nnz, const i64* colptr, const unsigned* rowidx, const double* val, int nb,
                      BlockedCsc* out, hipStream_t st)
{
    free_blocked_csc(out);
    if (nb < 2 || nnz <= 0 || nnz > 0x7FFFFFFF) return 1;
    // rows per block: a power of two, so that the block number is a shift
    unsigned shift = 0;
    while (((i64)1 << shift) * nb < rows) ++shift;
    int bits = 0;
    while ((1 << bits) < nb) ++bits;
    unsigned *key = nullptr, *key_sorted = nullptr, *col_of = nullptr, *pos = nullptr, *perm = nullptr, *col_b = nullptr;
    void* temp = nullptr;
    size_t tb = 0;
    int rc = 0;
    auto fail = [&](const char* what) { set_error(std::string("blocked CSC: ") + what); rc = -100; };
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key, key_sorted, pos, perm, (int)nnz, 0, bits, st) != hipSuccess) fail("size query");
    unsigned** u32s[] = {&key, &key_sorted, &col_of, &pos, &perm, &col_b};
    for (unsigned** p : u32s)
        if (!rc && smk::dev_malloc((void**)p, (size_t)nnz * 4) != hipSuccess) fail("hipMalloc");
    if (!rc && smk::dev_malloc(&temp, tb + 16) != hipSuccess) fail("hipMalloc");
    if (!rc && smk::dev_malloc((void**)&out->cp, (size_t)nb * (ncols + 1) * sizeof(i64)) != hipSuccess) fail("hipMalloc");
    if (!rc && smk::dev_malloc((void**)&out->ri, (size_t)nnz * 4) != hipSuccess) fail("hipMalloc");
    if (!rc && smk::dev_malloc((void**)&out->va, (size_t)nnz * 8) != hipSuccess) fail("hipMalloc");
    if (!rc) {
        const int g1 = (int)((ncols * 64 + 255) / 256 < 8192 ? (ncols * 64 + 255) / 256 : 8192);
        bl_keys_kernel<<<g1 > 0 ? g1 : 1, 256, 0, st>>>(colptr, rowidx, ncols, shift, key, col_of, pos);
        if (hipcub::DeviceRadixSort::SortPairs(temp, tb, key, key_sorted, pos, perm, (int)nnz, 0, bits, st) != hipSuccess) fail("radix sort");   // stable
    }
    if (!rc) {
        const int g2 = (int)((nnz + 255) / 256 < 8192 ? (nnz + 255) / 256 : 8192);
        bl_gather_kernel<<<g2, 256, 0, st>>>(perm, col_of, rowidx, val, nnz, col_b, out->ri, out->va);
        const i64 total = (i64)nb * (ncols + 1);
        const int g3 = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        bl_offsets_kernel<<<g3, 256, 0, st>>>(key_sorted, col_b, nnz, ncols, nb, out->cp);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) fail("kernels");
    }
    void* ptrs[] = {key, key_sorted, col_of, pos, perm, col_b, temp};
    for (void* p : ptrs) if (p) (void)smk::dev_free(p);
    if (rc) { free_blocked_csc(out); return rc; }
    out->nb = nb; out->rb = (i64)1 << shift; out->ncols = ncols; out->nnz = nnz;
    return 0;
}

// P: [nb][ncols_pad][2] partial products, X: the compact copy of the gathered factor (16 bytes per row)
int launch_spmm_blocked2(const BlockedCsc& b, const double* X, double* P, i64 ncols_pad, hipStream_t st)
{
    if (b.nb < 2 || b.ncols <= 0) return -100;
    const double avg = (double)b.nnz / ((double)b.ncols * b.nb);
    static const int forced = [] { const char* e = getenv("SMK_SPMM_BLOCKED_LPC"); return e ? atoi(e) : 0; }();
    const int lpc = forced ? forced : (avg <= 1.5 ? 1 : avg <= 3.0 ? 2 : avg <= 6.0 ? 4 : 8);
    static const int unroll = [] { const char* e = getenv("SMK_SPMM_UNROLL"); return e ? atoi(e) : 1; }();
    const int U = unroll >= 4 ? 4 : unroll >= 2 ? 2 : 1;
    const i64 cpw = (256 / lpc) * U;                            // columns per workgroup
    const i64 tiles = (b.ncols + cpw - 1) / cpw;
    const i64 per = 8 / b.nb;                                   // column tiles per group of 8 workgroups
    const i64 grid = (tiles + per - 1) / per * 8;
#define SMK_BL(LP, UU) spmm_blocked2_kernel<LP, UU><<<(unsigned)grid, 256, 0, st>>>(b.cp, b.ri, b.va, b.ncols, ncols_pad, b.nb, X, P)
#define SMK_BLU(LP) do { if (U == 4) SMK_BL(LP, 4); else if (U == 2) SMK_BL(LP, 2); else SMK_BL(LP, 1); } while (0)
    switch (lpc) {
        case 1: SMK_BLU(1); break;
        case 2: SMK_BLU(2); break;
        case 4: SMK_BLU(4); break;
        default: SMK_BLU(8); break;
    }
#undef SMK_BLU
#undef SMK_BL
    SMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace smk
