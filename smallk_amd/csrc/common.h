// smallk_amd/csrc/common.h -- shared host-side declarations for the MI355X NMF library.
#pragma once
#include <atomic>
#include <mutex>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <string>

namespace smk {

typedef int64_t i64;

// Device storage of the big matrix A (and its transpose).
enum Storage { STORE_F32 = 0, STORE_BF16 = 1 };

inline int elem_size(int storage) { return storage == STORE_BF16 ? 2 : 4; }

// Row padding of every stored big matrix (zero filled): lets the streaming
// kernel read whole 64/128-row stages without bounds checks.
constexpr i64 ROW_PAD = 128;
// Column padding = columns per workgroup tile of the streaming kernel.
constexpr i64 COL_PAD = 256;

inline i64 round_up(i64 x, i64 m) { return (x + m - 1) / m * m; }

// padded k handled by the column-per-thread kernels
// padded rank: the narrow kernels are built for 8 / 16 / 32 / 64 / 128; above that ("wide", wide.hip) a multiple of 64
inline int kp_of(int k) { return k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : k <= 64 ? 64 : k <= 128 ? 128 : (k + 63) / 64 * 64; }
constexpr int MAX_K = 2048;                    // larger ranks: SMK_UNSUPPORTED (the widest instantiation of wide.hip: 32 values per lane)
constexpr int MAX_K_BPP = MAX_K;               // block pivoting: the Gram-inverse route at every rank (SMK_NNLS_INV=0: the direct form only); scratch at
                                               // KP = 2048: (num_cus x workgroups per CU + 2) x KP^2 doubles = about 8.6 GB per solver (nnls_wide_scratch_elems)
constexpr int MAX_GROUPS = MAX_K / 64;         // the streaming product takes 64 factor rows per pass over A
inline bool is_wide(int k) { return k > 128; }
// block pivoting takes the tile kernels of wide.hip (and their scratch layout) from this rank on: everything above 128, and
// k in (64, 128] unless SMK_NNLS_TILE128=0 asks for nnls_bpp_inv128_kernel
bool nnls_uses_tiles(int k);
// number of 32-wide k tiles of the streaming product
// devmem.cpp: hipMalloc / hipFree with a per-device cache of freed blocks (solver, subset and sort workspaces come and go per
// node of a clustering run); dev_trim returns the current device's cached blocks to the runtime
hipError_t dev_malloc(void** p, size_t bytes);
template <typename T> inline hipError_t dev_malloc(T** p, size_t bytes) { return dev_malloc((void**)p, bytes); }
hipError_t dev_free(void* p);
void dev_trim();
void dev_cache_stats(unsigned long long* hits, unsigned long long* misses, size_t* cached_bytes);
inline int kt_of(int k) { return (k + 31) / 32; }
// "do this once" for things that are per DEVICE (hipFuncSetAttribute applies to the function on the current device): one
// process may drive several devices (smk_nmf_dense_sharded, bench.py --single-process, HierNMF2 with SMK_CLUST_DEVICES), and a
// process-wide flag would opt the kernel in on the first device only.  Usage:
//     static std::atomic<unsigned long long> attr_set{0};
//     if (DeviceOnce once{attr_set}) { SMK_HIP(hipFuncSetAttribute(...)); once.done(); }
// The device's bit is set by done(), i.e. only after the set-up has succeeded (an early return leaves it clear and the next
// call tries again), and the first caller holds a lock until then: a second host thread on the same device (shards on one GPU,
// HierNMF2 workers) waits instead of launching with > 64 KB of dynamic LDS before the opt-in has taken effect.
inline std::mutex& device_once_mutex() { static std::mutex mu; return mu; }
struct DeviceOnce {
    std::atomic<unsigned long long>& mask;
    unsigned long long bit = 0;
    std::unique_lock<std::mutex> lk;
    bool first = false;
    explicit DeviceOnce(std::atomic<unsigned long long>& m) : mask(m)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        bit = 1ull << (dev & 63);
        if (mask.load(std::memory_order_acquire) & bit) return;
        lk = std::unique_lock<std::mutex>(device_once_mutex());
        first = !(mask.load(std::memory_order_acquire) & bit);
    }
    explicit operator bool() const { return first; }
    void done() { mask.fetch_or(bit, std::memory_order_release); }
};
// doubles per column of the partial products of a dense pass: the k tiles of 32 rows -- but 8 / 16 for k <= 8 / 16 (half or a
// quarter of the bytes written by the streaming pass and read by the update kernel behind it: C2 is k = 16)
inline int kpp_of(int k) { return k <= 8 ? 8 : k <= 16 ? 16 : kt_of(k) * 32; }

void set_error(const std::string& msg);

#define SMK_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ::smk::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));        \
            return -100; /* SMK_DEVICE_ERROR */                                         \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------------------
// kernel launchers (kernels.hip).  All asynchronous on `st`.
// X-side matrices are fp64, column-major k x N with leading dimension k
// ("H layout"; W is kept transposed as Wt, k x m).
// P = fp64 partial products from the streaming kernel: [S][ncols_pad][KPP].
// ---------------------------------------------------------------------------
struct PartialView {
    const void* p;    // base: doubles (kernel output) or floats (after the cross-GPU all-reduce)
    int S;            // number of partial slabs to sum
    i64 slab;         // elements between slabs (= ncols_pad * kpp)
    int kpp;          // padded k of a column in P (multiple of 32)
    int f64;          // element type of p
};

int launch_fill_uniform(void* buf, int storage, i64 ld, i64 rows, i64 cols, i64 rows_pad, i64 cols_pad,
                        i64 r0, i64 c0, i64 gheight, uint64_t seed, int quant, hipStream_t st);
// planted low-rank + noise, keyed by the global element index (kernels.hip: fill_planted_kernel)
int launch_fill_planted(void* buf, int storage, i64 ld, i64 rows, i64 cols, i64 rows_pad, i64 cols_pad, i64 c0,
                        i64 gheight, uint64_t seed, int kstar, double thr, double noise, int quant, hipStream_t st);
int launch_convert_f64(const double* src, i64 ld_src, void* dst, int storage, i64 ld_dst, i64 rows, i64 cols,
                       hipStream_t st);
int launch_transpose_store(const void* src, i64 ld_src, void* dst, i64 ld_dst, int storage, i64 rows, i64 cols,
                           hipStream_t st);
// dst[:, j] = src[:, cols[j]] (whole padded columns; byte strides, multiples of 16)
int launch_fill_factor_uniform(double* X, int k, i64 N, unsigned long long seed, int transposed, hipStream_t st);
int launch_guard_compare(PartialView fast, const unsigned* cols, int ncols, const double* acc, int S_acc, i64 slab_acc, int kpp, int k,
                         double* out2, hipStream_t st);
int launch_gather_cols(const void* src, i64 ld_src_bytes, const unsigned* cols_dev, i64 ncols, void* dst,
                       i64 ld_dst_bytes, i64 col_bytes, hipStream_t st);
int launch_transpose_f64(const double* src, i64 ld_src, double* dst, i64 ld_dst, i64 rows, i64 cols, hipStream_t st);

// packed MFMA operand of X (k x N): bytes needed
// nsplit: 1..3 = bf16 terms of the skinny operand; NSPLIT_F16X2 = two fp16 terms with per-row power-of-two scales
constexpr int NSPLIT_F16X2 = 4;
// the accurate form: A's entries (exact in fp64) times the fp64 factor itself on the fp64 matrix cores (v_mfma_f64_16x16x4),
// fp64 accumulation throughout -- no packed operand; Xp of launch_bigprod is then the factor (KP doubles per column, rows
// from the plan's k0 on)
constexpr int NSPLIT_F64 = 8;
size_t packed_bytes(int storage, int k, i64 N, int nsplit);
size_t packed_row_offset(int storage, int kg, int nsplit, i64 r0);
// a row-sharded factor: X holds this rank's `nblocks` blocks of `blk` rows back to back (N valid rows, the rest packs as
// zeros); block j lands at rows ((j world + rank) blk ...) of the operand starting at `out`
int launch_pack_own_blocks(const double* X, int ldx, int k0, int kg, i64 N, i64 blk, int nblocks, int world, int rank,
                           int storage, int nsplit, void* out, hipStream_t st, const double* xscale = nullptr);
// k in (8, 16], fp16 two-term form: the Gram partials of an NNLS launch reduced AND the operand packed in one launch (returns 1
// for any other shape)
int launch_reduce_pack_f16x2(const double* Gp, int nblk, int k, double* G, double* xscale, double* oscale, double ascale,
                             const double* X, i64 N, int storage, void* out, hipStream_t st);
int launch_pack(const double* X, int k, i64 N, int storage, int nsplit, void* out, hipStream_t st, const double* xscale = nullptr);
// rows [k0, k0 + kg) of a factor stored with leading dimension ldx
int launch_pack_rows(const double* X, int ldx, int k0, int kg, i64 N, int storage, int nsplit, void* out, hipStream_t st,
                     const double* xscale = nullptr);

// The Gram inverse the NEXT block-pivoting launch needs, formed by one more workgroup of a product launch (gram_inverse.h; k in
// (16, 64] only): launch_spmm_seg / launch_spmm_gather / launch_bigprod return 1 instead of 0 when the launch carried it, so that the
// caller knows whether it still has to launch one
struct InvRide {
    const double* G = nullptr;       // the Gram matrix, KP x KP (complete before this launch starts)
    int k = 0;
    double* Ginv = nullptr;          // KP x KP doubles, then the int status (launch_gram_inverse's layout)
};

// streaming product: P[s][j][:] = sum over the rows of split s of X[:,row] * B[row, j]
struct BigProdPlan {
    // k > 64 runs as groups of 64 factor rows (one pass over the big matrix per group); this plan describes ONE group:
    // k0 = first factor row, pstride = doubles per column of P (the padded k of the whole factor)
    int k0 = 0, kg = 0, pstride = 0;
    size_t pack_offset = 0;   // bytes from the start of the packed operand to this group's fragments
    // fp16 two-term form (nsplit == NSPLIT_F16X2): A is multiplied by ascale (a power of two) before the split, the
    // packed operand carries per-row scales, and row r of the result is multiplied by oscale[r] = 1 / (xscale[r] ascale)
    const double* oscale = nullptr;   // device, indexed by the factor row (0 .. k-1 of the WHOLE factor)
    float ascale = 1.0f;
    int ldx = 0;    // NSPLIT_F64: doubles per column of the factor (its padded rank KP)
    int accum = 0;  // != 0: the launch adds to P instead of overwriting it (a later row chunk of the same product)
    // cache policy of the streamed loads of A: 0 = non-temporal (A is read once per pass and must not displace the operand slices:
    // right whenever A and A' together exceed the 256 MB Infinity Cache -- C3 7 %, a 1 GB matrix 20 % faster than with the default
    // policy), 1 = default policy (both copies stay cache resident between the passes: C2, 268 MB of A + A', 8 % faster);
    // the solver decides by the bytes an iteration streams (profiles/r05_cache_policy_ab.txt)
    int temporal = 0;
    // transposed source (bf16 storage, single-copy matrices): B is A itself for the H*A' pass -- the tile's columns are rows of A,
    // a stage is 64 columns of A, ldb the column stride of A (bigprod.hip: TRB); plan_bigprod_tr fills the plan
    int tr = 0;
    int mb = 0, nb = 0;   // rows per stage / columns per workgroup tile of the chosen kernel variant
    int S;          // row splits
    i64 len = 0;    // NSPLIT_F64: contraction length (rows past it read as zero from the factor)
    i64 stages;     // total stages = ceil(len / MB)
    i64 nst;        // stages per split
    i64 tiles;      // column tiles of 128
    int kt, nsplit, storage, variant;
    i64 ncols_pad;
    size_t p_elems; // doubles needed for P
    // optional tail (bigprod_supports_tail): 16 extra workgroups of the launch add up tail_nblk partial 16 x 16 Gram matrices
    // (the layout nnls_bpp_kernel<16> leaves) into tail_g, in gram_reduce_kernel's order, while the product streams
    const double* tail_gp = nullptr;
    int tail_nblk = 0;
    double* tail_g = nullptr;
    // ... and one more forms the totals of a deferred progress check (solver.cpp: check_rides_in_nnls; what launch_pg_defer_sum does
    // as a launch of its own): the sum of `n` per-workgroup projected-gradient sums, the failure flag and W'W of the checked
    // iteration, written to the device scalars, the pinned slot and the snapshot
    struct TailCheck {
        const double* part = nullptr;
        int n = 0, flag_slot = 0, tag_limit = 0, kk = 0;
        double* out = nullptr;
        double* host_out = nullptr;
        const int* flag = nullptr;
        const double* G = nullptr;
        double* snap_g = nullptr;
        double tag = 0.0;            // != 0: stored into host_out[7] last, system scope (the host polls the slot instead of an event)
    } tail_check;
    InvRide inv_ride;                // bigprod_supports_ride: eight more workgroups, the first one inverts (launch_bigprod returns 1)
};
bool bigprod_supports_tail(const BigProdPlan& pl);
bool bigprod_supports_ride(const BigProdPlan& pl);
BigProdPlan plan_bigprod(int storage, int k, i64 len, i64 ncols, int nsplit, int num_cus);
// the groups of a k-row factor (1 for k <= 64, 2 up to 128): same row splits, P laid out [S][ncols_pad][32 kt_of(k)]
int plan_bigprod_groups(int storage, int k, i64 len, i64 ncols, int nsplit, int num_cus, BigProdPlan* out /* 2 */);
int launch_bigprod(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st);
// the groups of the H*A' pass taken from A itself: len = columns of A (the contraction), ncols = rows of A; bf16 storage with
// nsplit 1..3, or fp32 storage with the fp16 two-term form / bf16x3
int plan_bigprod_groups_tr(int storage, int k, i64 len, i64 ncols, int nsplit, int num_cus, BigProdPlan* out);

int launch_reduce_partials(PartialView pv, int k, i64 c0, i64 N, void* out /* [.][kpp] */, int out_f64, hipStream_t st);

// xscale / oscale (KP doubles each, optional): per-row scales of the fp16 two-term operand, derived from the diagonal
// in the same reduce launch: xscale[r] = 2^e with sqrt(G_rr) 2^e in [2^13, 2^14], oscale[r] = 1 / (xscale[r] * ascale)
int launch_gram(const double* X, int k, i64 N, double* G /* KP x KP */, double* scratch, int max_blocks, hipStream_t st,
                double* xscale = nullptr, double* oscale = nullptr, double ascale = 1.0);
size_t gram_scratch_elems(int k, int max_blocks);
// the two halves of launch_gram, for a factor given as several row segments (one partials call per segment with a
// scratch offset of nblk * KP * KP doubles, then one reduce); launch_gram_scales: the fp16 row scales from a finished G
int launch_gram_partials(const double* X, int k, i64 N, double* scratch, int max_blocks, int* nblk_out, hipStream_t st);
int launch_gram_reduce(const double* scratch, int nblk, int k, double* G, hipStream_t st, double* xscale = nullptr,
                       double* oscale = nullptr, double ascale = 1.0);
int launch_gram_scales(const double* G, int k, double* xscale, double* oscale, double ascale, hipStream_t st);
// *out = bits of max |A[i]| (NaN entries are ignored by fmaxf)
int launch_absmax_f32(const float* A, i64 elems, unsigned* out, hipStream_t st);
// out2[0] = bits of the largest column maximum of |A|, out2[1] = bits of the smallest non-zero one (0xFFFFFFFF: none)
int launch_colrange(const void* A, int storage, i64 ld, i64 rows, i64 cols, unsigned* out2, hipStream_t st);
// out[0] = the largest sum of squares of a column of A (rows x cols, column-major with leading dimension ld); one pass
int launch_colnorm2_max(const void* A, int storage, i64 ld, i64 rows, i64 cols, double* out, hipStream_t st);
// chunk pairs of the packed operand of an N-row factor in the fp16 two-term form (what NnlsPack::nq wants)
i64 packed_chunk_pairs_f16x2(int storage, i64 N);
// G = X X' and the packed streaming operand of X in one launch (k <= 64, bf16 fragments); returns 1 if this shape has
// no fused kernel.  The ticket word at scratch[max_blocks * KP * KP] must be zero before the first call.
int launch_gram_pack(const double* X, int k, i64 N, double* G, double* scratch, int max_blocks, int storage, int nsplit,
                     void* packed, hipStream_t st);

// ---- k > 128 (wide.hip): one wave per column, the Gram matrix through the caches; reached through the launch_*
// functions below and in nnls.hip, which branch on is_wide(k)
int launch_gram_wide_partials(const double* X, int KP, i64 N, double* scratch, int max_blocks, int* nblk_out, hipStream_t st);
int gram_wide_blocks(int KP, i64 N, int max_blocks);
// tmp (optional, N x KP doubles): Y = X G as one product on the f64 matrix cores instead of a matrix-vector product per column
int launch_mu_update_wide(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st, double* tmp);
int launch_hals_sweep_wide(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st);
int launch_grad_pg_wide(const double* X, int k, i64 N, PartialView R, const double* G, double* grad_out, double* pg_partials,
                        int* grid_out, hipStream_t st, double* tmp);
int hals_w_wide_blocks(i64 M);
// k > 64: the W sweep by blocks of 16 columns (one product over W per block + a small launch per column)
size_t hals_w_blocked_scratch_elems(int k, i64 M);
int launch_hals_w_update_blocked(double* Wt, int k, i64 M, PartialView R, const double* G, double* scratch, hipStream_t st);
int launch_hals_w_update_wide(double* Wt, int k, i64 M, PartialView R, const double* G, double* scratch, hipStream_t st);
int launch_spmm_gather_wide(const i64* colptr, const unsigned* rowidx, const double* val, i64 ncols, const double* X, int k,
                            double* P, int kpp, hipStream_t st);
size_t nnls_wide_scratch_elems(int k, int num_cus, i64 ncols);   // ncols: the most columns one launch will solve
int launch_nnls_bpp_wide(double* X, double* Y, int k, i64 col_begin, i64 col_end, PartialView R, const double* G, int* fail_flag,
                         int iter_tag, double* scratch, int inverse_ready, int num_cus, hipStream_t st);
// L, Ginv and the status word of `scratch` from G (what launch_nnls_bpp_wide does first unless inverse_ready)
int launch_gram_inverse_wide(const double* G, int k, double* scratch, int num_cus, hipStream_t st);

int launch_mu_update(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st, double* wide_tmp = nullptr);
// epilogue of the two HALS sweeps (kernels.hip: tile_pack_gram): the sweep also writes the packed operand of the product that
// follows (bf16 fragments, nsplit terms) and one partial Gram matrix per workgroup into Gp; on return done says whether it did
// (nblk partials, for launch_gram_reduce) -- when not, the caller runs the separate Gram / pack launch
struct HalsEpilogue {
    unsigned char* pack_out = nullptr;
    double* Gp = nullptr;
    int KT = 1, nsplit = 3, max_blocks = 0, nblk = 0;
    i64 nq = 0;
    bool done = false;
};
int launch_hals_sweep(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st, HalsEpilogue* ep = nullptr);
// gradient G*X - R, optional store, projected-gradient partial sums -> pg_accum[slot] += sum
int launch_grad_pg(const double* X, int k, i64 N, PartialView R, const double* G, double* grad_out,
                   double* pg_partials, double* pg_accum, int slot, hipStream_t st, double* wide_tmp = nullptr);
int launch_grad_pg_partials(const double* X, int k, i64 N, PartialView R, const double* G, double* grad_out,
                            double* pg_partials, int* grid_out, hipStream_t st, double* wide_tmp = nullptr);
int launch_sum_partials(const double* partials, int n, double* out, hipStream_t st);
// both factors in two launches; also mirrors *flag into pg_accum[flag_slot] (as a double)
int launch_grad_pg2(const double* X1, i64 N1, PartialView R1, const double* G1, double* part1, const double* X2, i64 N2,
                    PartialView R2, const double* G2, double* part2, int k, double* pg_accum, const int* flag,
                    int flag_slot, hipStream_t st);
// the same + the snapshot of (W', H, W'W) in the gradient launch, and the totals written to `host_out` (pinned host memory, 8
// doubles) by the summing launch as well: two launches, no copy packet, per checked iteration.  snap may be NULL.  skip1: the
// projected-gradient sum of side 1 is known to be zero (BPP: the gradient is the NNLS's own dual), only its snapshot is taken.
int launch_grad_pg2_fused(const double* X1, i64 N1, PartialView R1, const double* G1, double* part1, const double* X2, i64 N2,
                          PartialView R2, const double* G2, double* part2, int k, double* pg_accum, const int* flag,
                          int flag_slot, double* snap, double* host_out, hipStream_t st, int skip1 = 0, double host_tag = 0.0);
// (host_tag != 0: stored into host_out[7] behind the totals, system scope -- the host polls the slot instead of waiting for an event)
// projected-gradient sum from an existing gradient array
int launch_pg_from_grad(const double* X, const double* Y, int k, i64 N, double* pg_partials, double* pg_accum,
                        int slot, hipStream_t st);
// HALS W update (all k columns, k+1 launches); norms scratch: [k][nblocks] + ...
int launch_hals_w_update(double* Wt, int k, i64 M, PartialView R, const double* G, double* scratch, int num_cus,
                         int* fail_flag, int parity, int force_multi, hipStream_t st, HalsEpilogue* ep = nullptr);
int hals_w_scratch_init(double* scratch, int k, i64 M, hipStream_t st);
size_t hals_w_scratch_elems(int k, i64 M);
// BPP / NNLS block principal pivoting over all columns
// scratch: nnls_scratch_elems(k) doubles (k > 32: inverse of G + path selector), may be NULL (slow path only)
// gram_partials / gram_nblk (optional): k in (8, 16], all columns from 0, at most NNLS_GRAM_MAX workgroups: the launch also
// leaves partial Gram matrices X X' of the solved columns ([*gram_nblk][16 * 16], for launch_gram_reduce); *gram_nblk = 0 otherwise
constexpr int NNLS_GRAM_MAX = 1024;
// pack (optional, only together with gram_partials): the launch also writes the solved factor as the packed operand of the
// fp16 two-term product (pack_f16x2_kernel's layout, KT = 1) with row scales that need no pass over the result: the system
// matrix G is the Gram matrix of the OTHER factor F >= 0, and at a KKT point x_r |f_r| <= |F x| <= |a|, so
// x_r <= anorm / sqrt(G_rr) with anorm = the largest 2-norm of a column (row) of A.  xscale / oscale receive the scales
// (KP doubles each); an entry beyond fp16's range after scaling (cannot happen while the bound holds) sets *fail_flag = -4.
struct NnlsPack {
    unsigned char* out = nullptr;     // nullptr: off
    double* xscale = nullptr;
    double* oscale = nullptr;
    double anorm = 0.0, ascale = 1.0;
    i64 nq = 0;                       // 1-KiB chunk pairs of the operand (rows padded to ROW_PAD): the tail past the last column is zeroed
};
constexpr int NNLS_PACK_OVERFLOW = -4;
// riders of a k <= 16 block-pivoting launch that serve the CHECKED iteration loop (round 6, solver.cpp: deferred progress check):
//   pg_part  the launch first forms the gradient G x - r of its WARM START (= the factor and the products the previous iteration
//            left: exactly gradH of that iteration, nmf_solver_bpp.hpp:370-371) and leaves one projected-gradient partial sum per
//            workgroup (projected_gradient.hpp:125-171) -- the stopping rule of iteration i costs no pass of its own in iteration i + 1
//   snap_x   the solved factor is stored a second time, compact (k2 = k rounded up to even values per column): the snapshot the
//            driver restores when the rule fires one iteration late
struct NnlsRiders {
    double* pg_part = nullptr;
    int* pg_nblk = nullptr;           // out: partials written
    double* snap_x = nullptr;
    int k2 = 0;
};
int launch_nnls_bpp(double* X, double* Y, int k, i64 col_begin, i64 col_end, PartialView R, const double* G,
                    int* fail_flag, int iter_tag, double* scratch, int inverse_ready, int num_cus, hipStream_t st,
                    double* gram_partials = nullptr, int* gram_nblk = nullptr, const NnlsPack* pack = nullptr,
                    unsigned* defer_ws = nullptr, const NnlsRiders* riders = nullptr);
// totals of a deferred check: out / host_out [0] = 0 (side 1: BPP's dual), [1] = sum of the n partials, [flag_slot] = the failure
// flag if it names an iteration <= tag_limit (else "none"); also copies the kk doubles of G into snap_g (may be NULL)
int launch_pg_defer_sum(const double* part, int n, double* out, double* host_out, const int* flag, int flag_slot, int tag_limit,
                        const double* G, double* snap_g, int kk, hipStream_t st, double host_tag = 0.0);
// k in (32, 64]: work list of the four-columns-per-wave kernel (nnls_g16.hip), nnls_defer_elems(ncols) unsigneds per launch
// in flight; without it launch_nnls_bpp keeps the wave-per-column kernel for every column
inline size_t nnls_defer_elems(i64 ncols) { return (size_t)ncols + 4; }
int launch_nnls_bpp_g16(double* X, double* Y, int k, i64 col_begin, i64 col_end, PartialView R, const double* G, const double* Ginv,
                        const int* status, int* fail_flag, int iter_tag, unsigned* defer, int num_cus, hipStream_t st,
                        unsigned long long* stats);
unsigned long long* nnls_stats_ptr();
// k > 32: the inverse of G into scratch, ahead of launch_nnls_bpp(..., inverse_ready = 1, ...) (any stream)
int launch_gram_inverse(const double* G, int k, double* scratch, hipStream_t st);
size_t nnls_scratch_elems(int k);
// diagnostics: the 256 counters of nnls.hip (SMK_NNLS_STATS=1), -1 when the counters are off
int nnls_stats_read(unsigned long long* out256, int reset);
bool nnls_inverse_at_32();      // k in (16, 32] solves through the inverse of the Gram matrix too (SMK_NNLS_INV32=0: not)
// normalisation: scale Wt rows by 1/nu_c, H rows by nu_c where nu_c^2 = G[c][c]
int launch_scale_rows(double* X, int k, i64 N, const double* G, int invert, int* fail_flag, hipStream_t st);
// delta-fnorm: out[0] = sum (W - Wprev)^2, out[1] = sum W^2 ; then Wprev = W
int launch_delta_fnorm(const double* W, double* Wprev, i64 count, double* partials, double* out2, hipStream_t st);
int launch_zero_f64(double* p, i64 n, hipStream_t st);
// sharded runs: scal[6..7] <- (scal[1], failed ? 1 : 0) before the all-reduce (unpack = 0); scal[1] <- scal[6] and
// flag <- min(flag, tag) when any rank failed, after it (unpack = 1)
int launch_dist_scalars(double* scal, int* flag, int tag, int unpack, int wpart, hipStream_t st);
// dst (k x N, ld k) = first k rows of src (KP x N, ld KP)
int launch_compact_rows(const double* src, int KP, double* dst, int k, i64 N, hipStream_t st);
// live rows of W' and H + the Gram matrix <-> one compact buffer (pack != 0: factors -> buffer)
size_t snapshot_elems(int k, i64 m, i64 n);
int launch_snapshot(double* Wt, i64 m, double* H, i64 n, double* G, double* buf, int k, int pack, hipStream_t st);
// ---- RANK2 (rank2.hip): the iteration as few fused launches
// X <- closed-form 2x2 solve + optimal active set of G X = R (side 0: H, side 1: W'); Xc (optional): compact N x 2 copy.
// The left-hand side: finished in Gin (gin_nb == 0) or gin_nb partial sums in Gin_p, which the kernel sums itself (its
// workgroup 0 then stores the finished matrix in Gin).  Gp_out (optional): partial sums of X X' of the result for the next
// consumer (*nb_out of them), or -- finish != 0 / too many -- summed into Gout by a second launch (*nb_out = 0).
int launch_rank2_solve(double* X, double* Xc, i64 N, PartialView R, double* Gin, const double* Gin_p, int gin_nb, int side,
                       int* fail_flag, int iter_tag, double* Gp_out, int* nb_out, double* Gout, int finish, hipStream_t st);
size_t rank2_gram_scratch_elems(i64 N);
// per-iteration NormalizeAndScale in one launch (H, W, the stored AH', HH'); Graw = W'W of the W just solved, finished
// or as partial sums; writes Gw = W'W of the normalised W (Graw_ij / (nu_i nu_j)) and, optionally, the compact copy of W
int launch_rank2_normalize(double* H, i64 n, double* Wt, double* Wc, i64 m, PartialView R, double* Gh, const double* Graw,
                           const double* Graw_p, int graw_nb, double* Gw, int* fail_flag, hipStream_t st);
int launch_rank2_compact(const double* X, double* Xc, i64 N, hipStream_t st);
// both projected-gradient sums as per-workgroup partials [blocks][2] (W, H) with the failure flag behind them, optional
// snapshot of (W, H, Gw) in launch_snapshot's layout
int launch_rank2_progress(const double* Wt, i64 m, PartialView R2, const double* Gh, const double* H, i64 n, PartialView R1,
                          const double* Gw, double* partials, const int* flag, double* snap, hipStream_t st);
int rank2_progress_blocks(i64 m, i64 n);
size_t rank2_progress_scratch_elems(i64 m, i64 n);
// ---- rank2_persist.hip: a whole RANK2 factorisation of a sparse matrix (driver loop, stopping rule, final state) in ONE
// launch of resident workgroups with two grid-wide barriers per iteration
enum { R2P_RUNNING = 0, R2P_CONVERGED = 1, R2P_EXHAUSTED = 2, R2P_SOLVER_FAILED = 3, R2P_NAN = 4, R2P_ABORTED = 5 };   // out[0]
struct R2PersistArgs {
    const i64 *colptr, *colptr_t;              // CSC of A (n columns) and of A' (m columns)
    const unsigned *rowidx, *rowidx_t;
    const double *val, *val_t;
    i64 m, n;
    const double* Gw0;                         // W'W of the start (KP = 8 layout)
    PartialView R1;                            // W'A of the start
    double *Wc, *Hc0, *Hc1, *R2c;              // compact N x 2 work arrays
    double *gp_h, *gp_w, *pgp;                 // [workgroups][8] partial sums
    unsigned* sync;                            // rank2_persist_sync_bytes(), zeroed by the launch function
    int min_iter, max_iter, tolcount;
    double tol;
    int iter_tag0;                             // iterations done before this run (failure tags count from it)
    double *Wt, *H, *Gw;                       // results (KP = 8 layout), written by the epilogue only
    int* fail_flag;
    unsigned lds_bytes;                        // dynamic LDS per workgroup (rank2_persist_lds_bytes())
    double* out;                               // [16] device: status, NmfStats::iteration_count, iterations performed, pg0, last metric, failure tag; [8..11]: us spent in B1 / phase W / B2 / phase G (workgroup 0)
};
size_t rank2_persist_sync_bytes();
size_t rank2_persist_lds_bytes();
int rank2_persist_workgroups(i64 m, i64 n, i64 nnz, int num_cus);      // 0: not for this matrix
int launch_rank2_persist(const R2PersistArgs& a, int workgroups, hipStream_t st);
// sparse A (CSC): out[:, j] = sum_p val[p] * X[:, row[p]] over the nonzeros of column j
// X: the gathered factor, row pitch ldx doubles (KP, or 2 for the compact copy of a rank-2 factor)
// nnz_hint: number of stored entries (picks the lanes per column of the rank-2 kernel; <= 0: unknown)
int launch_spmm_gather(const i64* colptr, const unsigned* rowidx, const double* val, i64 ncols, i64 nnz_hint, const double* X,
                       int ldx, int k, double* P, int kpp, hipStream_t st, const InvRide* ride = nullptr);

// spmm_seg.hip: the gather product at ranks 3 .. 128, work cut by stored entries.  The plan of one CSC (A or A'): segments of
// <= spmm_seg_len() consecutive entries = whole columns, or pieces of one long column (summed by a fix-up launch)
struct SegPlan {
    i64 nseg = 0, nlong = 0, npieces = 0, ncols = 0, nnz = 0;
    bool has_empty = false;          // some column has no stored entry (P is cleared first, the column walk skips them)
    bool uniform = false;            // column lengths within 4 x the mean: the column-per-lane-group kernel is the faster one
    i64 longest = 0;
    i64* seg_p0 = nullptr;           // first entry
    unsigned* seg_len = nullptr;     // entries
    unsigned* seg_col = nullptr;     // column of the first entry
    unsigned* seg_piece = nullptr;   // 0xFFFFFFFF: whole columns; else the slot of this piece in `pieces`
    unsigned* rowflag = nullptr;     // row indices, bit 31 set on the last entry of a column
    unsigned* long_col = nullptr;    // columns longer than a segment ...
    i64* long_piece0 = nullptr;      // ... and their first piece (nlong + 1 entries)
    double* pieces = nullptr;        // npieces x 128 doubles
};
int spmm_seg_len();
int build_seg_plan(i64 ncols, i64 nnz, const i64* colptr, const unsigned* rowidx, SegPlan* out, hipStream_t st);
void free_seg_plan(SegPlan* s);
// `pieces`: the caller's own npieces x 128 doubles for the partial sums of long columns (a solver owns one per pass, so that two
// solvers on one matrix and different streams do not share it); nullptr: the plan's buffer (one stream at a time)
// `gram`: the Gram matrix of a factor (k in (8, 32]) formed along the way -- its partial sums by extra workgroups of the gather launch
// (gram_body.h; both only read a factor), their reduction by extra workgroups of the fix-up launch (a launch of its own when the
// matrix has no long columns).  Return value: bit 0 = the Gram inverse was carried, bit 1 = the Gram matrix was formed.
struct GramRide {
    const double* X = nullptr;       // the factor, KP x N
    i64 N = 0;
    int max_blocks = 0;              // the blocking of launch_gram_partials(.., max_blocks, ..): the same partial sums in the same order
    double* Gp = nullptr;            // partial sums ([nblk][KP * KP])
    double* G = nullptr;             // the result, KP x KP
};
void gram_partial_shape(i64 N, int max_blocks, int* nblk_out, i64* cpw_out);
int launch_spmm_seg(const SegPlan& sp, const i64* colptr, const double* val, const double* X, int k, double* P, int kpp,
                    hipStream_t st, double* pieces = nullptr, const InvRide* ride = nullptr, const GramRide* gram = nullptr);

// spmm_blocked.hip: the rank-2 gather product with the gathered factor cut into row blocks that stay in one XCD's L2.
// A matrix regrouped by row block: block b is a CSC of its own (cp[b * (ncols + 1) + j] .. are absolute positions in ri / va)
struct BlockedCsc {
    int nb = 0;              // row blocks: 2, 4 or 8 (0: not built)
    i64 rb = 0;              // rows per block (a power of two)
    i64 ncols = 0, nnz = 0;
    i64* cp = nullptr;
    unsigned* ri = nullptr;
    double* va = nullptr;
};
int blocked_csc_blocks(i64 rows);      // 1: the factor fits an L2, no blocking
int build_blocked_csc(i64 rows, i64 ncols, i64 nnz, const i64* colptr, const unsigned* rowidx, const double* val, int nb,
                      BlockedCsc* out, hipStream_t st);
void free_blocked_csc(BlockedCsc* b);
// P: [nb][ncols_pad][2] partial products (the consumers add the slabs), X: compact copy of the factor (16 B per row)
int launch_spmm_blocked2(const BlockedCsc& b, const double* X, double* P, i64 ncols_pad, hipStream_t st);

// sparse_subset.hip: CSC(A[:, cols]) and CSC(A[:, cols]') with unused rows dropped, assembled on the device
// from the resident CSC(A) / CSC(A').  `cols` must be strictly increasing.  The six output arrays are
// hipMalloc'ed for the caller; new_to_old_host (capacity src.m) receives the kept rows.
struct SparseDev {
    i64 m = 0, n = 0, nnz = 0;
    i64 *colptr = nullptr, *colptr_t = nullptr;
    unsigned *rowidx = nullptr, *rowidx_t = nullptr;
    double *val = nullptr, *val_t = nullptr;
};
int device_sparse_subset(const SparseDev& src, const unsigned* cols_host, i64 ncols, SparseDev* out,
                         unsigned* new_to_old_host, hipStream_t st);

// sort.hip: stable descending radix sort of host vectors on the device (argsort when idx_host[v] != nullptr,
// keys-only into sorted_host[v] otherwise)
int device_sort_desc(const double* const* keys_host, int* const* idx_host, double* const* sorted_host, int count, i64 n,
                     hipStream_t st);
// compute_priority (clust_hier_util.hpp:105-173) on the device: wp, wc[0..n), wc[n..2n) host vectors in, the score out;
// n_part = number of nonzeros of wp.  0 on success (negative: not available, take the host path).  The workspace is kept
// per host thread between calls; device_priority_release() frees it.
int device_priority_score(const double* wp, const double* wc, i64 n, i64 n_part, double* score, hipStream_t st);
void device_priority_release();
// CSC(A') from CSC(A), all device arrays (colptr_t: height + 1 offsets); entry order = the host counting sort's.
// Returns 0, or non-zero when the caller should take the host path.
int device_csc_transpose(i64 height, i64 ncols, i64 nnz, const i64* colptr, const unsigned* rowidx, const double* val,
                         i64* colptr_t, unsigned* rowidx_t, double* val_t, hipStream_t st);

}  // namespace smk
