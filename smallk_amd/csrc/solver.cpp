// smallk_amd/csrc/solver.cpp -- host side of the C ABI (include/smallk_amd.h):
// device-resident A, the NmfSolve<> driver loop and the three per-iteration
// schedules (MU / HALS / BPP) expressed as launches of the kernels in kernels.hip.
//
// Device data layout (all in HBM, fp64 unless noted):
//   A   : m_pad x n_pad   bf16|f32, column-major, zero padded (rows to 128, cols to 128)
//   At  : n_pad x m_pad   same dtype: the explicit transpose (the reference's BPP keeps one
//         too, nmf_solver_bpp.hpp:319) so BOTH streaming products contract down the
//         contiguous dimension
//   H   : KP x n (ld KP)    Wt : KP x m (ld KP) -- W is kept transposed on the device; KP = k padded
//         to 8/16/32/64, pad rows are zero
//   Gw = W'W, Gh = HH' : KP x KP (KP = 8/16/32/64 padded)
//   P1 : S1 slabs of n_pad x kpp fp64 = W'A partials,   P2 : S2 slabs of m_pad x kpp = (AH')' partials
//   packW / packH : MFMA operand fragments of W' / H
#include "common.h"
#include "comm.h"
#include "../../include/smallk_amd.h"

#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <deque>
#include <algorithm>
#include <thread>
#include <mutex>

struct smk_matrix;

namespace smk {

// One device context per process by default; the single-process multi-GPU driver (smk_nmf_dense_sharded) runs
// one host thread per shard and gives each its own context through t_ctx.
struct DeviceCtx {
    bool init = false;
    int cus = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int live_solvers = 0;               // solver handles cache the stream: it cannot change under them
    std::vector<struct ::smk_matrix*> mats;   // live matrices: they follow the context's stream when it is replaced
};
static DeviceCtx g_ctx;
static thread_local DeviceCtx* t_ctx = nullptr;
static inline DeviceCtx& ctx() { return t_ctx ? *t_ctx : g_ctx; }
#define g_init (ctx().init)
#define g_cus (ctx().cus)
#define g_stream (ctx().stream)
#define g_own_stream (ctx().own_stream)
#define g_live_solvers (ctx().live_solvers)
static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }

static inline double wall_us()
{
    using namespace std::chrono;
    return (double)duration_cast<nanoseconds>(steady_clock::now().time_since_epoch()).count() * 1e-3;
}

template <typename T>
static int dev_alloc(T** p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    SMK_HIP(smk::dev_malloc((void**)p, count * sizeof(T)));
    // debugging aid: SMK_POISON=1 fills every fresh workspace with 0xFF bytes (NaN as fp64 / fp32, -1 as int), so that a
    // kernel reading memory nobody wrote shows up in every run instead of once in a hundred
    static const bool poison = [] { const char* e = getenv("SMK_POISON"); return e && atoi(e) != 0; }();
    if (poison) { SMK_HIP(hipMemset(*p, 0xFF, count * sizeof(T))); SMK_HIP(hipDeviceSynchronize()); }   // the fill must not trail work on the non-blocking streams
    return 0;
}

}  // namespace smk

using namespace smk;

struct smk_matrix {
    i64 m = 0, n_global = 0, c0 = 0, n = 0;
    int storage = SMK_STORE_F32;
    hipStream_t st = nullptr;                        // stream of the context that created it
    smk::DeviceCtx* owner = nullptr;                 // the context whose registry lists it (nullptr once that context is gone)
    mutable float ascale = 0.f;                      // fp16 two-term products: power of two with max|A| ascale in [2^13, 2^14); 0 = not yet measured
    mutable int col_spread_log2 = -1;                // log2(largest / smallest non-zero column maximum of |A|); -1 = not yet measured
    mutable double colnorm_max = -1.0, rownorm_max = -1.0;   // largest 2-norm of a column / a row of A (dense; NnlsPack's bound); < 0 = not yet measured
    void* A = nullptr;  i64 ldA = 0, colsA = 0;      // m_pad x n_pad
    void* At = nullptr; i64 ldAt = 0, colsAt = 0;    // n_pad x m_pad
    // single copy (MU / HALS): no stored transpose -- the H*A' pass contracts down the strided direction of A itself
    // (bigprod.hip: TRB for bf16, TAIL = 2 for fp32), as the reference's MU / HALS do (Gemm(NORMAL, TRANSPOSE) on A, nmf_solver_mu.hpp:121-164,
    // nmf_solver_hals.hpp:166-199); half the footprint, no transpose pass at load time
    bool single = false;
    // sparse A: CSC of the local columns and CSC of its transpose (fp64 values, 64-bit offsets)
    bool sparse = false;
    i64 nnz = 0;
    i64 *colptr = nullptr, *colptr_t = nullptr;
    unsigned *rowidx = nullptr, *rowidx_t = nullptr;
    double *val = nullptr, *val_t = nullptr;
    // host copy of the CSC (column subsets for HierNMF2 nodes are cut on the host)
    // RANK2 on a factor larger than an L2: the entries regrouped by row block (spmm_blocked.hip), built on first use
    mutable BlockedCsc bA, bAt;
    mutable bool blocked_tried = false;
    // ranks 3 .. 128 on sparse A: the entry-balanced segments of CSC(A) / CSC(A') (spmm_seg.hip), built on first use
    mutable SegPlan segA, segAt;
    mutable bool seg_tried = false;
    mutable std::vector<unsigned> h_colptr, h_rowidx;     // fetched on first use (ensure_host_csc)
    mutable std::vector<double> h_val;
};

static const int MAX_CHUNKS = 8;
static void repoint_matrices(hipStream_t st);

// The registries of all contexts share one lock: a matrix may be destroyed from another thread than the one that created
// it (Python's collector, the workers of smk_nmf_dense_sharded), and it leaves the registry of the context that OWNS it.
static std::mutex g_mats_mu;
static void repoint_matrices(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_mats_mu);
    for (smk_matrix* a : ctx().mats) a->st = st;
}
// the context goes away: its matrices stay alive without an owner (they take the next context's stream)
static void orphan_matrices()
{
    std::lock_guard<std::mutex> lk(g_mats_mu);
    for (smk_matrix* a : ctx().mats) { a->st = nullptr; a->owner = nullptr; }
    ctx().mats.clear();
}
static void register_matrix(smk_matrix* a)
{
    std::lock_guard<std::mutex> lk(g_mats_mu);
    a->owner = &ctx();
    a->owner->mats.push_back(a);
}
static void unregister_matrix(smk_matrix* a)
{
    std::lock_guard<std::mutex> lk(g_mats_mu);
    if (!a->owner) return;
    auto& v = a->owner->mats;
    v.erase(std::remove(v.begin(), v.end(), a), v.end());
    a->owner = nullptr;
}

struct smk_solver {
    smk_options o;
    const smk_matrix* a = nullptr;
    int k = 0, KP = 0, kpp = 0, nsplit = 3;
    i64 m = 0, n = 0;
    hipStream_t st = nullptr;
    double *H = nullptr, *Wt = nullptr, *Gw = nullptr, *Gh = nullptr, *gram_scratch = nullptr;
    double* seg_pieces[2] = {nullptr, nullptr};     // sparse A, spmm_seg.hip: partial sums of the long columns of pass 0 / 1
    double *Wprev = nullptr, *hals_scratch = nullptr, *pg_partials = nullptr, *scal = nullptr, *tmpW = nullptr;
    double* wide_tmp = nullptr;           // k > 128: max(m, n) x KP, the product X G of the MU rule and of the gradients
    double* tmpH = nullptr;               // k x n compact copy of H for the host (get_factors)
    // RANK2 (rank2.hip): scratch of the fused solve / progress kernels (ticket + partial sums), W'W of the W just solved
    // (before its normalisation), and -- sparse A -- compact N x 2 copies of the factors for the gather products
    double *r2_scratch = nullptr, *r2_prog = nullptr, *Graw = nullptr, *Hc = nullptr, *Wc = nullptr;
    static constexpr int PROG_SLOTS = 4;         // progress checks in flight + 1 (solver_run_once / smk_solver_iterate_checked)
    double* pin_r2[PROG_SLOTS] = {nullptr, nullptr, nullptr, nullptr};      // pinned copies of the progress partials (the host sums them)
    // the whole RANK2 factorisation as one resident launch (rank2_persist.hip): second H buffer, the rows of (AH')', partial
    // sums, barrier words, result slots (device + pinned); latched off after an aborted launch
    double *r2p_hc1 = nullptr, *r2p_r2c = nullptr, *r2p_part = nullptr, *r2p_out = nullptr, *r2p_pin = nullptr;
    unsigned* r2p_sync = nullptr;
    bool r2p_off = false;
    // run-time guard of the product form (guard_step): a column sample of A, its accurate-form product, the comparison scalars
    // + both Gram matrices on their way to the host
    void* guard_As = nullptr;
    unsigned* guard_cols = nullptr;
    double *guard_P = nullptr, *guard_dev = nullptr, *guard_pin = nullptr;
    hipEvent_t guard_ev = nullptr;
    BigProdPlan guard_pl[MAX_GROUPS];
    int guard_ncols = 0, guard_checks = 0, guard_fired = 0;
    bool guard_pending = false, guard_off = false;
    double guard_last = 0.0;               // cond * delta of the last check
    bool wc_valid = false;
    double* nnls_scratch = nullptr;       // BPP: inverses of W'W and HH' + path selectors (k > 32), two halves
    unsigned* nnls_defer = nullptr;       // BPP, k in (32, 64]: work list between nnls_bpp_g16_kernel and the wave-per-column kernel
    int hals_ep_blocks = 0;               // HALS, k <= 32: Gram partials the sweeps' epilogues may write into gram_scratch (0: epilogues off)
    // deferred progress check (BPP, k <= 16; check_rides_in_nnls): the slot whose totals the NEXT H-side NNLS launch produces, the
    // iteration tag up to which a failure counts for it, whether its snapshot is being written by this iteration's NNLS launches
    int pg_defer_slot = -1, pg_defer_tag = 0, pg_defer_nblk = 0, iter_snap_slot = -1;
    bool pg_defer_snap = false;
    unsigned check_routes[4] = {0, 0, 0, 0}; // progress checks formed so far by route (smk_solver_kernel_name(2)): 1 own launches, 2 NNLS riders + totals launch, 3 riders + pass tail
    int pg_totals_slot = -1;                 // >= 0: the H-side launch has left the partial sums of this slot's check; its totals are due
    hipStream_t st_inv = nullptr;         // the 0.1 ms single-workgroup inversions run here, beside the streaming products
    hipEvent_t ev_g[2] = {nullptr, nullptr}, ev_inv[2] = {nullptr, nullptr};
    bool inv_pending[2] = {false, false};
    bool gram_ride[2] = {false, false};   // sparse, k in (8, 32]: this factor's Gram matrix is due and rides in the two launches of the gather product that follows (gram_factor, timed_spmm)
    bool inv_ride[2] = {false, false};    // sparse BPP, k in (16, 64]: this side's Gram matrix is new, its inverse is to ride in the product launch that follows (timed_spmm)
    double *xscale[2] = {nullptr, nullptr}, *oscale[2] = {nullptr, nullptr};   // fp16 two-term products: row scales of W / H (from the Gram diagonal) and their inverses
    bool packed_fresh[2] = {false, false};   // the fused Gram kernel has already written packW / packH for the next product
    int nnls_gram_nblk[2] = {0, 0};          // > 0: the NNLS launch of this side left that many Gram partials in gram_scratch (k <= 16)
    // k in (8, 16], BPP, fp16 form, one GPU (C2): the NNLS launch also PACKS the factor it solves (row scales from an a-priori
    // bound, NnlsPack) and the reduction of its Gram partials rides in the streaming pass that follows (BigProdPlan::tail_*),
    // so nothing stands between the solve and the product.  Indexed by factor: 0 = W, 1 = H.
    bool pack_in_solve = false;              // the shape qualifies (decided with the plans)
    bool pack_in_solve_off = false;          // latched by pack_fail_soft
    bool from_nnls[2] = {false, false};      // the factor is the output of an NNLS launch of this run (hence >= 0)
    bool nnls_packed[2] = {false, false};    // the last NNLS launch packed this factor
    int tail_nblk[2] = {0, 0};               // > 0: the next product of this factor carries the reduction of that many partials
    // HALS: the fused W sweep needs every workgroup resident; if its bounded polls ever expire (flag -3) the run is
    // repeated from the initial factors on the one-launch-per-column path, latched for the life of the handle
    double *W0c = nullptr, *H0c = nullptr;
    bool hals_multi = false;
    int hals_calls = 0;
    double *Gh_own = nullptr, *scal_own = nullptr, *Wt_own = nullptr;
    void *packW = nullptr, *packH = nullptr;
    double *P1 = nullptr, *P2 = nullptr;
    float* R2red = nullptr;
    BigProdPlan pl1, pl2;                 // first group of each pass (row splits, P layout)
    BigProdPlan pg1[MAX_GROUPS], pg2[MAX_GROUPS];   // all groups: k > 64 streams the big matrix once per 64 factor rows
    int ng = 1;
    int* fail_flag = nullptr;
    int iter = 0;
    bool have_factors = false, inited = false, normalized = false;
    double pg0 = 1.0, last_metric = 1.0;
    size_t pg_half = 2048;
    // comm: a native communicator (RCCL or the in-process stand-in, comm.cpp) or -- test hook -- a host callback
    int rank = 0, world = 1;
    smk_allreduce_fn ar = nullptr;
    void* ar_user = nullptr;
    smk_comm* comm = nullptr;
    void* comm_ws = nullptr;              // owned workspace when a native communicator is attached
    // Native communicator: EVERY collective is issued on st2 (one stream per communicator), tied to the main stream by
    // events.  The rows of A (= columns of A', rows of W) are cut into `nchunk` chunks of world * blk rows; block r of a
    // chunk belongs to rank r (block-cyclic), so a chunk is at once a contiguous range of the H*At pass, the send buffer
    // of one reduce-scatter / all-reduce and the receive buffer of one all-gather: the exchange of chunk j runs on st2
    // while the streaming product works on chunk j + 1.
    int nchunk = 1;
    i64 blk = 0, rows_cap = 0;            // rows per (chunk, rank) block (multiple of 256); world * nchunk * blk >= m_pad
    bool r2_alias = false;                // the H*At pass writes ONE slab of fp64 partial products: the collectives work on it directly (no copy)
    bool red_f64 = false;                 // element type of the summed (AH')' on the wire (native communicator: fp64 unless SMK_COMM_F64=0)
    bool w_sharded = false;               // BPP: every rank solves (and holds current) only its own blocks of W
    bool w_full = true;                   // all rows of the fp64 W on this rank are current
    // a row-sharded W: this rank's blocks back to back (KP x nchunk*blk; the n_own valid rows are a prefix because only the
    // last non-empty block of a rank can be short) and, in the same order, its rows of the summed (AH')'
    double* Wown = nullptr;
    void* R2own = nullptr;
    i64 n_own = 0;
    hipStream_t st2 = nullptr;
    hipEvent_t ev_gram = nullptr, ev_gh = nullptr, ev_x = nullptr, ev_y = nullptr;
    hipEvent_t ev_c[MAX_CHUNKS] = {}, ev_r[MAX_CHUNKS] = {}, ev_a[MAX_CHUNKS] = {};
    bool gh_pending = false;
    bool r2_pending = false;              // the chunk exchanges of the last H*At pass have not been joined by the main stream yet
    bool inv_done[2] = {false, false};    // the inverse of this side's current Gram matrix is in place (ordered before the main stream)
    // stopping rule evaluated one iteration late (smk_solver_run): pinned result slots, events, and a
    // snapshot of (W, H, W'W) per checked iteration so that a speculative iteration can be undone
    struct ProgSlot { double h[8]; int flag; int fused; };     // fused: the flag travels in h[5]
    ProgSlot* pin = nullptr;
    hipEvent_t pev[PROG_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    double poll_tag[PROG_SLOTS] = {0, 0, 0, 0};      // != 0: the kernel stores this into h[7] behind the result; progress_end polls the slot (no event)
    double* snap[PROG_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    // timing
    bool timing = false;
    // a pair of event records around a launch costs ~11 us of idle time (5.7 us in front of the kernel, 5.8 behind it: measured
    // on C2, where that was 23 of 119 us per iteration): passes shorter than ~0.2 ms are timed one launch in `timing_stride`
    // and the totals scaled back up, so that measuring does not change what is measured
    int timing_stride = 1;
    unsigned pass_counter[2] = {0, 0}, pass_sampled[2] = {0, 0};     // passes seen / passes that carried events since enable_timing
    bool pass_timed[2] = {false, false};     // this W'A / H*At pass (all of its launches, and the collectives behind it) is a timed sample
    struct TimedSpan { hipEvent_t e0, e1; int counts; };     // counts: this span completes one launch (a pass cut into chunks is ONE launch)
    // 0: W'A passes, 1: H*At passes, 2: the big collectives of a sharded run (on st2), 3: what the MAIN stream spends waiting
    // for events of the collective stream (the exposed part of the exchange: the bracket holds nothing but the wait)
    // 4: the same bracket around a wait for an event that completed long ago -- what a bracket costs by itself (three packets
    // through the command processor, ~15 us): exposure = slot 3 - brackets x the average of slot 4
    std::vector<TimedSpan> ev[6];       // 0 / 1: the passes, 2 - 4: collectives, waits, calibration, 5: the block-pivoting launches
    double acc_ms[6] = {0, 0, 0, 0, 0, 0};
    int launches[6] = {0, 0, 0, 0, 0, 0};
    hipEvent_t ev_cal = nullptr;          // recorded once on the collective stream
    unsigned cal_counter = 0;
};

static const int GRAM_BLOCKS = 256;


extern "C" {

int smk_initialize(int device_ordinal)
{
    if (device_ordinal >= 0) SMK_HIP(hipSetDevice(device_ordinal));
    int dev = 0;
    SMK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SMK_HIP(hipGetDeviceProperties(&prop, dev));
    g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (!g_stream) {
        SMK_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        g_own_stream = true;
    }
    g_init = true;
    return SMK_OK;
}

int smk_is_initialized(void) { return g_init ? SMK_INITIALIZED : SMK_NOTINITIALIZED; }

void smk_finalize(void)
{
    if (g_stream) (void)hipStreamSynchronize(g_stream);
    if (g_own_stream && g_stream) (void)hipStreamDestroy(g_stream);
    orphan_matrices();                    // a matrix that outlives the context takes the next context's stream
    dev_trim();                           // cached device blocks of this device go back to the runtime
    g_stream = nullptr;
    g_own_stream = false;
    g_init = false;
}

// A host thread that drives a device of its own (the second device of a two-device HierNMF2 run, hierclust.cpp): its
// library state -- stream, CU count, live handles -- is separate from the process-wide context from here to _end().
int smk_thread_context_begin(int device_ordinal)
{
    if (t_ctx) { set_error("this thread already has a context of its own"); return SMK_BAD_PARAM; }
    t_ctx = new DeviceCtx;
    const int rc = smk_initialize(device_ordinal);
    if (rc != SMK_OK) { delete t_ctx; t_ctx = nullptr; }
    return rc;
}
void smk_thread_context_end(void)
{
    if (!t_ctx) return;
    smk_finalize();
    delete t_ctx;
    t_ctx = nullptr;
}
int smk_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
int smk_current_device(void)
{
    int d = 0;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}

size_t smk_device_trim(void)
{
    size_t cached = 0;
    smk::dev_cache_stats(nullptr, nullptr, &cached);
    smk::dev_trim();
    return cached;
}

int smk_device_synchronize(void)
{
    SMK_HIP(hipDeviceSynchronize());
    return SMK_OK;
}

const char* smk_last_error(void) { return g_err.c_str(); }
int smk_device_cu_count(void) { return g_cus; }

int smk_set_stream(void* hip_stream)
{
    if (g_live_solvers > 0) { set_error("smk_set_stream: destroy every solver handle first"); return SMK_BAD_PARAM; }
    if (g_stream) (void)hipStreamSynchronize(g_stream);     // resident-matrix work queued on the old stream
    if (g_own_stream && g_stream) (void)hipStreamDestroy(g_stream);
    g_stream = (hipStream_t)hip_stream;
    g_own_stream = false;
    repoint_matrices(g_stream);           // resident matrices were created under the old stream
    return SMK_OK;
}

// IsValid, common/src/nmf_options.cpp:23-112 (same checks, same messages)
int smk_is_valid(const smk_options* o, int validate_matrix)
{
    if (!o) return 0;
    if (o->k <= 0) { fprintf(stderr, "nmflib error: k-value must be a positive integer\n"); return 0; }
    if (validate_matrix) {
        if (o->height <= 0) { fprintf(stderr, "nmflib error: matrix height must be a positive integer\n"); return 0; }
        if (o->width <= 0) { fprintf(stderr, "nmflib error: matrix width must be a positive integer\n"); return 0; }
        if (o->k > o->width) { fprintf(stderr, "nmflib error: k value cannot exceed the number of columns\n"); return 0; }
    }
    if (o->tol <= 0.0 || o->tol >= 1.0) { fprintf(stderr, "nmflib error: tolerance must be in the interval (0.0, 1.0)\n"); return 0; }
    if (o->min_iter <= 0) { fprintf(stderr, "nmflib error: miniter must be a positive integer\n"); return 0; }
    if (o->max_iter <= 0) { fprintf(stderr, "nmflib error: maxiter must be a positive integer\n"); return 0; }
    if (o->tolcount <= 0) { fprintf(stderr, "nmflib error: tolcount must be a positive integer\n"); return 0; }
    if (o->algorithm != SMK_ALG_MU && o->algorithm != SMK_ALG_HALS && o->algorithm != SMK_ALG_RANK2 &&
        o->algorithm != SMK_ALG_BPP) {
        fprintf(stderr, "nmflib error: unknown NMF algorithm specified\n");
        return 0;
    }
    if (o->algorithm == SMK_ALG_RANK2 && o->k != 2) { fprintf(stderr, "nmflib error: RANK2 algorithm requires k == 2\n"); return 0; }
    if (o->prog_est_algorithm != SMK_PROG_PG_RATIO && o->prog_est_algorithm != SMK_PROG_DELTA_FNORM) {
        fprintf(stderr, "nmflib error: unknown stopping criterion specified\n");
        return 0;
    }
    return 1;
}

// same generator as the device fill (kernels.hip) and the oracle, on the host
static inline uint64_t h_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void smk_uniform_fill_host(double* buf, int64_t ld, int64_t rows, int64_t cols, int64_t r0, int64_t c0,
                           int64_t gheight, uint64_t seed, int quant)
{
    for (int64_t c = 0; c < cols; ++c)
        for (int64_t r = 0; r < rows; ++r) {
            uint64_t h = h_mix64(seed * 0xD1342543DE82EF95ull + (uint64_t)((c0 + c) * gheight + (r0 + r)));
            float f = (float)(h >> 40) * (1.0f / 16777216.0f);
            if (quant == 1) {
                uint32_t b;
                memcpy(&b, &f, 4);
                b += 0x7FFFu + ((b >> 16) & 1u);
                b &= 0xFFFF0000u;
                memcpy(&f, &b, 4);
            }
            buf[c * ld + r] = (double)f;
        }
}

// ------------------------------------------------------------------------------------------
// matrix
// ------------------------------------------------------------------------------------------
static thread_local bool g_create_single = false;
int smk_matrix_create(smk_matrix** out, int64_t height, int64_t width_global, int64_t col0, int64_t ncols_local,
                      int storage)
{
    if (!out) return SMK_BAD_PARAM;
    *out = nullptr;
    if (!g_init) { set_error("smk_initialize() has not been called"); return SMK_NOTINITIALIZED; }
    if (height <= 0 || width_global <= 0 || ncols_local <= 0 || col0 < 0 || col0 + ncols_local > width_global ||
        (storage != SMK_STORE_F32 && storage != SMK_STORE_BF16))
        return SMK_BAD_PARAM;
    smk_matrix* a = new smk_matrix;
    a->m = height; a->n_global = width_global; a->c0 = col0; a->n = ncols_local; a->storage = storage;
    a->st = g_stream;
    register_matrix(a);
    // rows of A padded to COL_PAD, not ROW_PAD: a single-copy matrix is also read through the transposed source, whose tiles are
    // 128 ROWS of A and whose chunked passes (sharded runs, chunk_rows) run to round_up(m, COL_PAD) -- with 128-row padding a
    // height with 0 < m mod 256 <= 128 let the last tile read 128 rows past the column (the next column's data; past the
    // allocation in the last column).  The pad rows are zero like every other pad.
    a->ldA = round_up(height, COL_PAD);      a->colsA = round_up(ncols_local, COL_PAD);
    a->ldAt = round_up(ncols_local, ROW_PAD); a->colsAt = round_up(height, COL_PAD);
    const size_t es = (size_t)elem_size(storage);
    // A column stride that is a multiple of 1 MiB gets ROW_PAD more (zero) rows: with the 128 columns of a workgroup's stage
    // exactly 2^20 bytes apart the W'A pass of C4 runs 3 % slower (11.3 -> 10.95 ms) and that of a C4 shard 8 % (1.60 ->
    // 1.47 ms; bench.py --emulate-world 8: 3.28 -> 3.15 ms per rank).  Smaller power-of-two strides are best left alone
    // (C3: 128 KiB and 32 KiB strides, skewed: 1200 -> 1130 / 980 it/s); 256 and 384 rows more gain less than 128.
    // SMK_LD_SKEW=0 turns it off, =n asks for n rows.  (profiles/r04_leading_dimension_skew.txt)
    {
        static const i64 skew = [] { const char* e = getenv("SMK_LD_SKEW"); return e ? (i64)atoll(e) / ROW_PAD * ROW_PAD : ROW_PAD; }();
        if (skew > 0 && ((size_t)a->ldA * es) % ((size_t)1 << 20) == 0) a->ldA += skew;
        if (skew > 0 && ((size_t)a->ldAt * es) % ((size_t)1 << 20) == 0) a->ldAt += skew;
    }
    {   // SMK_SINGLE_COPY=1: dense matrices are created without the stored transpose (smk_matrix_create_single_copy asks for it explicitly)
        const char* esc = getenv("SMK_SINGLE_COPY");
        a->single = g_create_single || (esc && esc[0] == '1');
    }
    hipError_t e1 = smk::dev_malloc(&a->A, (size_t)a->ldA * a->colsA * es);
    hipError_t e2 = (e1 == hipSuccess && !a->single) ? smk::dev_malloc(&a->At, (size_t)a->ldAt * a->colsAt * es) : e1;
    if (e1 == hipSuccess && e2 != hipSuccess && !a->single) {
        // A fits, A and A' together do not: the matrix becomes a single copy (MU, HALS and BPP with the 16-bit product forms run
        // on it as they are; RANK2 and the accurate form will ask for the transpose and report the allocation failure then)
        (void)hipGetLastError();
        a->At = nullptr;
        a->single = true;
        e2 = hipSuccess;
    }
    if (e1 != hipSuccess || e2 != hipSuccess) {
        set_error(std::string("smk::dev_malloc(A): ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        if (a->A) (void)smk::dev_free(a->A);
        delete a;
        return SMK_DEVICE_ERROR;
    }
    e1 = hipMemsetAsync(a->A, 0, (size_t)a->ldA * a->colsA * es, g_stream);
    if (e1 == hipSuccess && a->At) e1 = hipMemsetAsync(a->At, 0, (size_t)a->ldAt * a->colsAt * es, g_stream);
    if (e1 != hipSuccess) {
        set_error(std::string("hipMemsetAsync(A): ") + hipGetErrorString(e1));
        smk_matrix_destroy(a);
        return SMK_DEVICE_ERROR;
    }
    *out = a;
    return SMK_OK;
}

int smk_matrix_create_single_copy(smk_matrix** out, int64_t height, int64_t width_global, int64_t col0, int64_t ncols_local,
                                  int storage)
{
    g_create_single = true;
    const int rc = smk_matrix_create(out, height, width_global, col0, ncols_local, storage);
    g_create_single = false;
    return rc;
}
int smk_matrix_is_single_copy(const smk_matrix* a) { return a && a->single ? 1 : 0; }
// bytes of HBM the resident matrix occupies (A, the stored transpose when there is one, the CSC arrays of a sparse matrix)
int64_t smk_matrix_device_bytes(const smk_matrix* a)
{
    if (!a) return 0;
    if (a->sparse) return (int64_t)((size_t)(a->n + 1 + a->m + 1) * sizeof(i64) + 2 * (size_t)a->nnz * (sizeof(unsigned) + sizeof(double)));
    const size_t es = (size_t)elem_size(a->storage);
    return (int64_t)((size_t)a->ldA * a->colsA * es + (a->At ? (size_t)a->ldAt * a->colsAt * es : 0));
}

// a single-copy matrix meets a consumer of the stored transpose (BPP, RANK2, the accurate form, column subsets): allocate and fill
// it now; solvers already planned on the transposed source keep reading A (their plans say so)
static int matrix_materialize_transpose(const smk_matrix* ca)
{
    smk_matrix* a = const_cast<smk_matrix*>(ca);       // the lazily built parts of a matrix (scales, blocked CSC, segment plans) are filled the same way
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!a->single || a->At) return 0;
    const size_t es = (size_t)elem_size(a->storage);
    hipStream_t st = a->st ? a->st : g_stream;
    if (smk::dev_malloc(&a->At, (size_t)a->ldAt * a->colsAt * es) != hipSuccess) { a->At = nullptr; set_error("no memory for the stored transpose of a single-copy matrix"); return SMK_DEVICE_ERROR; }
    SMK_HIP(hipMemsetAsync(a->At, 0, (size_t)a->ldAt * a->colsAt * es, st));
    const int rc = launch_transpose_store(a->A, a->ldA, a->At, a->ldAt, a->storage, a->m, a->n, st);
    if (rc) return rc;
    SMK_HIP(hipStreamSynchronize(st));
    a->single = false;
    return 0;
}

static int matrix_make_transpose(smk_matrix* a)
{
    if (a->single) return 0;
    return launch_transpose_store(a->A, a->ldA, a->At, a->ldAt, a->storage, a->m, a->n, g_stream);
}

// ---- host fp64 -> resident matrix ---------------------------------------------------------------------------------------
// The reference wraps the caller's buffer as a view, no copy (common/src/nmf.cpp:224-226); here A has to cross PCIe once, and this
// is the path every reference caller takes (nmf/src/main.cpp:218-233, smallk.cpp:604-619, smallk_lib.pyx:769).  The loop is plain:
// hipMemcpy2DAsync straight from the caller's pageable buffer into one 64 MB device staging buffer, conversion to the stored type,
// synchronise, next chunk; the stored transpose in one device pass at the end.  Round 6 MEASURED it before replacing it
// (profiles/r06_upload_rates.txt, bench.py --api-path): 50 - 55 GB/s on C3's 8.6 GB, on a C4 shard's 17 GB and on C2's 0.27 GB -- the
// runtime pins the pageable pages in place piece by piece and the per-chunk synchronisation costs nothing measurable.  Two
// pipelined variants were built and timed against it on the same box: pinned staging buffers filled by 2 - 16 host threads with the
// transfer and the conversion overlapped (41 GB/s whatever the thread count, 10 - 19 GB/s on C2's matrix: the pinned allocations)
// and hipHostRegister of each chunk of the caller's buffer (52 GB/s).  Both slower, both removed.
int smk_matrix_upload_f64(smk_matrix* a, const double* host, int64_t ld)
{
    if (a) { a->ascale = 0.f; a->col_spread_log2 = -1; a->colnorm_max = a->rownorm_max = -1.0; }     // new contents: scale, column spread and norms are measured again on first use
    if (!a || !host || ld < a->m || a->sparse) return SMK_BAD_PARAM;
    const size_t budget = (size_t)64 << 20;   // staging bytes
    i64 chunk = (i64)(budget / ((size_t)a->m * sizeof(double)));
    if (chunk < 1) chunk = 1;
    if (chunk > a->n) chunk = a->n;
    double* stage = nullptr;
    int rc = dev_alloc(&stage, (size_t)a->m * chunk);
    if (rc) return rc;
    struct Free { void* p; ~Free() { if (p) (void)smk::dev_free(p); } } stage_guard{stage};   // also on the error returns
    const size_t es = (size_t)elem_size(a->storage);
    for (i64 c = 0; c < a->n; c += chunk) {
        const i64 nc = (a->n - c < chunk) ? (a->n - c) : chunk;
        SMK_HIP(hipMemcpy2DAsync(stage, (size_t)a->m * sizeof(double), host + c * ld, (size_t)ld * sizeof(double),
                                 (size_t)a->m * sizeof(double), (size_t)nc, hipMemcpyHostToDevice, g_stream));
        rc = launch_convert_f64(stage, a->m, (unsigned char*)a->A + (size_t)c * a->ldA * es, a->storage, a->ldA,
                                a->m, nc, g_stream);
        if (rc) return rc;
        SMK_HIP(hipStreamSynchronize(g_stream));
    }
    rc = matrix_make_transpose(a);
    if (rc) return rc;
    SMK_HIP(hipStreamSynchronize(g_stream));
    return SMK_OK;
}

int smk_matrix_fill_uniform(smk_matrix* a, uint64_t seed)
{
    if (a) { a->ascale = 0.f; a->col_spread_log2 = -1; a->colnorm_max = a->rownorm_max = -1.0; }
    if (!a || a->sparse) return SMK_BAD_PARAM;
    int rc = launch_fill_uniform(a->A, a->storage, a->ldA, a->m, a->n, a->ldA, a->colsA, 0, a->c0, a->m, seed,
                                 a->storage == SMK_STORE_BF16 ? 1 : 0, g_stream);
    if (rc) return rc;
    rc = matrix_make_transpose(a);
    if (rc) return rc;
    SMK_HIP(hipStreamSynchronize(g_stream));
    return SMK_OK;
}

int smk_matrix_fill_planted(smk_matrix* a, uint64_t seed, int kstar, double threshold, double noise)
{
    if (a) { a->ascale = 0.f; a->col_spread_log2 = -1; a->colnorm_max = a->rownorm_max = -1.0; }
    if (!a || a->sparse || kstar < 1 || kstar > 4096 || !(threshold >= 0.0 && threshold < 1.0) || !(noise >= 0.0)) return SMK_BAD_PARAM;
    int rc = launch_fill_planted(a->A, a->storage, a->ldA, a->m, a->n, a->ldA, a->colsA, a->c0, a->m, seed, kstar, threshold,
                                 noise, a->storage == SMK_STORE_BF16 ? 1 : 0, g_stream);
    if (rc) return rc;
    rc = matrix_make_transpose(a);
    if (rc) return rc;
    SMK_HIP(hipStreamSynchronize(g_stream));
    return SMK_OK;
}

int smk_matrix_download_f64(const smk_matrix* a, double* host, int64_t ld)
{
    if (!a || !host || ld < a->m || a->sparse) return SMK_BAD_PARAM;
    const size_t es = (size_t)elem_size(a->storage);
    std::vector<unsigned char> col((size_t)a->m * es);
    SMK_HIP(hipStreamSynchronize(g_stream));
    for (i64 c = 0; c < a->n; ++c) {
        SMK_HIP(hipMemcpy(col.data(), (const unsigned char*)a->A + (size_t)c * a->ldA * es, (size_t)a->m * es,
                          hipMemcpyDeviceToHost));
        if (a->storage == SMK_STORE_BF16) {
            const uint16_t* p = (const uint16_t*)col.data();
            for (i64 r = 0; r < a->m; ++r) {
                uint32_t b = ((uint32_t)p[r]) << 16;
                float f;
                memcpy(&f, &b, 4);
                host[c * ld + r] = (double)f;
            }
        } else {
            const float* p = (const float*)col.data();
            for (i64 r = 0; r < a->m; ++r) host[c * ld + r] = (double)p[r];
        }
    }
    return SMK_OK;
}

void smk_matrix_destroy(smk_matrix* a)
{
    if (!a) return;
    unregister_matrix(a);
    free_blocked_csc(&a->bA);
    free_blocked_csc(&a->bAt);
    free_seg_plan(&a->segA);
    free_seg_plan(&a->segAt);
    void* ptrs[] = {a->A, a->At, a->colptr, a->colptr_t, a->rowidx, a->rowidx_t, a->val, a->val_t};
    for (void* p : ptrs)
        if (p) (void)smk::dev_free(p);
    delete a;
}

// A copy of a resident matrix in the CALLING thread's context (its current device and stream): the second device of a
// two-device HierNMF2 run holds one (hierclust.cpp).  Device-to-device copies; works across devices and on one.
int smk_matrix_clone(const smk_matrix* src, smk_matrix** out)
{
    if (!src || !out) return SMK_BAD_PARAM;
    *out = nullptr;
    if (!g_init) { set_error("smk_initialize() has not been called"); return SMK_NOTINITIALIZED; }
    smk_matrix* a = new smk_matrix;
    a->m = src->m; a->n_global = src->n_global; a->c0 = src->c0; a->n = src->n; a->storage = src->storage;
    a->ascale = src->ascale; a->col_spread_log2 = src->col_spread_log2;
    a->colnorm_max = src->colnorm_max; a->rownorm_max = src->rownorm_max;
    a->ldA = src->ldA; a->colsA = src->colsA; a->ldAt = src->ldAt; a->colsAt = src->colsAt;
    a->sparse = src->sparse; a->nnz = src->nnz; a->single = src->single;
    a->st = g_stream;
    register_matrix(a);
    bool ok = true;
    auto dup = [&](void** dst, const void* from, size_t bytes) {
        if (!ok || !from) return;
        if (bytes == 0) bytes = 8;
        if (smk::dev_malloc(dst, bytes) != hipSuccess || hipMemcpy(*dst, from, bytes, hipMemcpyDefault) != hipSuccess) ok = false;
    };
    if (src->sparse) {
        dup((void**)&a->colptr, src->colptr, (size_t)(src->n + 1) * sizeof(i64));
        dup((void**)&a->colptr_t, src->colptr_t, (size_t)(src->m + 1) * sizeof(i64));
        dup((void**)&a->rowidx, src->rowidx, (size_t)src->nnz * sizeof(unsigned));
        dup((void**)&a->rowidx_t, src->rowidx_t, (size_t)src->nnz * sizeof(unsigned));
        dup((void**)&a->val, src->val, (size_t)src->nnz * sizeof(double));
        dup((void**)&a->val_t, src->val_t, (size_t)src->nnz * sizeof(double));
    } else {
        const size_t es = (size_t)elem_size(src->storage);
        dup(&a->A, src->A, (size_t)src->ldA * src->colsA * es);
        dup(&a->At, src->At, (size_t)src->ldAt * src->colsAt * es);
    }
    if (!ok) { set_error("smk_matrix_clone: device allocation or copy failed"); smk_matrix_destroy(a); return SMK_DEVICE_ERROR; }
    *out = a;
    return SMK_OK;
}

// ---- host-side CSC bookkeeping (no device involved; pinned against the reference's own SparseMatrix code
// compiled in place, oracle/_ref/libref_sparse.so, by tests/test_ref_sparse.py) -------------------------
// Transpose(SparseMatrix), common/include/sparse_matrix_ops.hpp:36-127: counting sort by row; inside a
// row of the result the entries keep the source's column order.
int smk_csc_transpose(int64_t height, int64_t width, const unsigned* col_offsets, const unsigned* row_indices,
                      const double* data, unsigned* out_col_offsets /* height+1 */, unsigned* out_row_indices,
                      double* out_data)
{
    if (height < 0 || width < 0 || !col_offsets || !out_col_offsets) return SMK_BAD_PARAM;
    const unsigned base = col_offsets[0];
    const int64_t nnz = (int64_t)col_offsets[width] - base;
    if (nnz > 0 && (!row_indices || !data || !out_row_indices || !out_data)) return SMK_BAD_PARAM;
    std::vector<i64> cnt((size_t)height + 1, 0);
    for (int64_t p = 0; p < nnz; ++p) {
        if ((int64_t)row_indices[base + p] >= height) { set_error("row index out of range"); return SMK_BAD_PARAM; }
        cnt[(size_t)row_indices[base + p] + 1] += 1;
    }
    for (int64_t r = 0; r < height; ++r) cnt[(size_t)r + 1] += cnt[(size_t)r];
    for (int64_t r = 0; r <= height; ++r) out_col_offsets[r] = (unsigned)cnt[(size_t)r];
    std::vector<i64> fill(cnt.begin(), cnt.end() - 1);
    for (int64_t c = 0; c < width; ++c)
        for (i64 p = (i64)col_offsets[c] - base; p < (i64)col_offsets[c + 1] - base; ++p) {
            const i64 q = fill[row_indices[base + p]]++;
            out_row_indices[q] = (unsigned)c;
            out_data[q] = data[base + p];
        }
    return SMK_OK;
}

// SparseMatrix::SubMatrixColsCompact, common/include/sparse_matrix_impl.hpp:478-592: the listed columns in
// the listed order, rows without a stored entry dropped and the rest renumbered in increasing order.
// Call once with out_* NULL for the sizes (*out_nnz, *new_height), then with arrays of that capacity.
// old_to_new (height entries, 0xFFFFFFFF = dropped) and new_to_old may be NULL.
int smk_csc_subset_cols_compact(int64_t height, int64_t width, const unsigned* col_offsets, const unsigned* row_indices,
                                const double* data, const unsigned* cols, int64_t ncols, unsigned* out_col_offsets,
                                unsigned* out_row_indices, double* out_data, unsigned* old_to_new, unsigned* new_to_old,
                                int64_t* new_height, int64_t* out_nnz)
{
    if (height <= 0 || width <= 0 || !col_offsets || !cols || ncols <= 0) { set_error("SubMatrixColsCompact: empty column set"); return SMK_BAD_PARAM; }
    const unsigned UNUSED = 0xFFFFFFFFu;
    std::vector<unsigned> o2n((size_t)height, UNUSED);
    int64_t total = 0;
    for (int64_t j = 0; j < ncols; ++j) {
        if ((int64_t)cols[j] >= width) { set_error("SubMatrixColsCompact: column index out of range"); return SMK_BAD_PARAM; }
        for (unsigned p = col_offsets[cols[j]]; p < col_offsets[cols[j] + 1]; ++p) o2n[row_indices[p]] = 0;
        total += col_offsets[cols[j] + 1] - col_offsets[cols[j]];
    }
    if (total == 0) { set_error("SparseMatrix::SubMatrixColsCompact: submatrix is the zero matrix"); return SMK_BAD_PARAM; }
    int64_t nh = 0;
    for (int64_t r = 0; r < height; ++r)
        if (o2n[(size_t)r] != UNUSED) {
            o2n[(size_t)r] = (unsigned)nh;
            if (new_to_old) new_to_old[nh] = (unsigned)r;
            ++nh;
        }
    if (old_to_new) std::copy(o2n.begin(), o2n.end(), old_to_new);
    if (new_height) *new_height = nh;
    if (out_nnz) *out_nnz = total;
    if (!out_col_offsets) return SMK_OK;
    if (!out_row_indices || !out_data) return SMK_BAD_PARAM;
    unsigned q = 0;
    for (int64_t j = 0; j < ncols; ++j) {
        out_col_offsets[j] = q;
        for (unsigned p = col_offsets[cols[j]]; p < col_offsets[cols[j] + 1]; ++p, ++q) {
            out_row_indices[q] = o2n[row_indices[p]];
            out_data[q] = data[p];
        }
    }
    out_col_offsets[ncols] = q;
    return SMK_OK;
}

// read a resident sparse matrix (or the stored CSC of its transpose) back to the host (tests)
int smk_matrix_download_csc(const smk_matrix* a, int transposed, unsigned* col_offsets, unsigned* row_indices, double* data)
{
    if (!a || !a->sparse || !col_offsets) return SMK_BAD_PARAM;
    const i64 nc = transposed ? a->m : a->n;
    std::vector<i64> cp((size_t)nc + 1);
    SMK_HIP(hipStreamSynchronize(g_stream));
    SMK_HIP(hipMemcpy(cp.data(), transposed ? a->colptr_t : a->colptr, cp.size() * sizeof(i64), hipMemcpyDeviceToHost));
    for (i64 c = 0; c <= nc; ++c) col_offsets[c] = (unsigned)cp[(size_t)c];
    if (a->nnz > 0 && row_indices && data) {
        SMK_HIP(hipMemcpy(row_indices, transposed ? a->rowidx_t : a->rowidx, (size_t)a->nnz * sizeof(unsigned), hipMemcpyDeviceToHost));
        SMK_HIP(hipMemcpy(data, transposed ? a->val_t : a->val, (size_t)a->nnz * sizeof(double), hipMemcpyDeviceToHost));
    }
    return SMK_OK;
}
static int ensure_seg_plans(const smk_matrix* a)
{
    // lazily built part of a shared, nominally const matrix: same lock discipline as matrix_materialize_transpose
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (a->seg_tried) return 0;
    a->seg_tried = true;
    static const bool seg_on = [] { const char* e = getenv("SMK_SPMM_SEG"); return !(e && e[0] == '0'); }();
    hipStream_t bst = a->st ? a->st : g_stream;
    if (seg_on && (build_seg_plan(a->n, a->nnz, a->colptr, a->rowidx, &a->segA, bst) ||
                   build_seg_plan(a->m, a->nnz, a->colptr_t, a->rowidx_t, &a->segAt, bst))) {
        free_seg_plan(&a->segA);
        free_seg_plan(&a->segAt);
    }
    return 0;
}

// The sparse Gemm of the reference by itself (common/include/sparse_gemm_ab_impl.hpp / sparse_gemm_ba_impl.hpp in gather
// form): out (k x ncols(B)) = X (k x rows(B)) * B with B = A (transposed == 0: W'A from X = W') or B = A' (transposed != 0:
// (AH')' from X = H), on the kernel the solver would take at rank k.  `reps` launches are timed with HIP events (avg_ms, may be NULL).
int smk_matrix_sparse_product(const smk_matrix* a, int transposed, int k, const double* X, int64_t ldx, double* out,
                              int64_t ldo, int reps, double* avg_ms)
{
    if (!a || !a->sparse || k < 1 || k > MAX_K || !X || !out || ldx < k || ldo < k) return SMK_BAD_PARAM;
    const i64 rows = transposed ? a->n : a->m, ncols = transposed ? a->m : a->n;
    const int KP = kp_of(k);
    const int kpp = (k <= 2) ? 2 : KP;
    const int ldx_dev = (k <= 2) ? 2 : KP;
    hipStream_t st = a->st ? a->st : g_stream;
    if (k > 2 && !is_wide(k)) ensure_seg_plans(a);
    std::vector<double> xp((size_t)rows * ldx_dev, 0.0), pp((size_t)ncols * kpp);
    for (i64 r = 0; r < rows; ++r)
        for (int c = 0; c < k; ++c) xp[(size_t)r * ldx_dev + c] = X[r * ldx + c];
    double *dX = nullptr, *dP = nullptr;
    int rc = dev_alloc(&dX, xp.size());
    if (!rc) rc = dev_alloc(&dP, pp.size());
    struct Free { double *&a, *&b; ~Free() { if (a) (void)smk::dev_free(a); if (b) (void)smk::dev_free(b); } } guard{dX, dP};
    if (rc) return rc;
    SMK_HIP(hipMemcpyAsync(dX, xp.data(), xp.size() * sizeof(double), hipMemcpyHostToDevice, st));
    const i64* cp = transposed ? a->colptr_t : a->colptr;
    const unsigned* ri = transposed ? a->rowidx_t : a->rowidx;
    const double* va = transposed ? a->val_t : a->val;
    const SegPlan& seg = transposed ? a->segAt : a->segA;
    auto once = [&]() -> int {
        if (k > 2 && !is_wide(k) && seg.rowflag && seg.ncols == ncols && !seg.uniform) return launch_spmm_seg(seg, cp, va, dX, k, dP, kpp, st);
        return launch_spmm_gather(cp, ri, va, ncols, a->nnz, dX, ldx_dev, k, dP, kpp, st);
    };
    rc = once();
    if (rc) return rc;
    if (reps > 0 && avg_ms) {
        hipEvent_t e0, e1;
        SMK_HIP(hipEventCreate(&e0));
        SMK_HIP(hipEventCreate(&e1));
        SMK_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < reps && !rc; ++i) rc = once();
        SMK_HIP(hipEventRecord(e1, st));
        SMK_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms = (double)ms / reps;
        if (rc) return rc;
    }
    SMK_HIP(hipMemcpyAsync(pp.data(), dP, pp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    SMK_HIP(hipStreamSynchronize(st));
    for (i64 j = 0; j < ncols; ++j)
        for (int c = 0; c < k; ++c) out[j * ldo + c] = pp[(size_t)j * kpp + c];
    return SMK_OK;
}
int64_t smk_matrix_nnz(const smk_matrix* a) { return a ? a->nnz : 0; }
int64_t smk_matrix_height(const smk_matrix* a) { return a ? a->m : 0; }

// CSC shard (columns [col0, col0+ncols_local) of a height x width_global matrix) -> HBM, plus the
// CSC of its transpose built on the host by a counting sort (SparseMatrix::Transpose,
// sparse_matrix_ops.hpp:37-127).  Duplicate entries are kept (they add up in every product, as in
// the reference's Compress(), sparse_matrix_impl.hpp:184-260).
int smk_matrix_create_sparse(smk_matrix** out, int64_t height, int64_t width_global, int64_t col0,
                             int64_t ncols_local, int64_t nnz, const unsigned* col_offsets,
                             const unsigned* row_indices, const double* data)
{
    if (!out) return SMK_BAD_PARAM;
    *out = nullptr;
    if (!g_init) { set_error("smk_initialize() has not been called"); return SMK_NOTINITIALIZED; }
    if (height <= 0 || width_global <= 0 || ncols_local <= 0 || col0 < 0 || col0 + ncols_local > width_global ||
        nnz < 0 || !col_offsets || (nnz > 0 && (!row_indices || !data)))
        return SMK_BAD_PARAM;
    if ((int64_t)col_offsets[ncols_local] - (int64_t)col_offsets[0] != nnz) { set_error("col_offsets do not span nnz"); return SMK_BAD_PARAM; }
    const unsigned base = col_offsets[0];
    std::vector<i64> cp((size_t)ncols_local + 1);
    for (int64_t c = 0; c <= ncols_local; ++c) {
        if (c > 0 && col_offsets[c] < col_offsets[c - 1]) { set_error("col_offsets not monotone"); return SMK_BAD_PARAM; }
        cp[(size_t)c] = (i64)col_offsets[c] - base;
    }
    for (int64_t p = 0; p < nnz; ++p)
        if ((int64_t)row_indices[base + p] >= height) { set_error("row index out of range"); return SMK_BAD_PARAM; }
    smk_matrix* a = new smk_matrix;
    a->m = height; a->n_global = width_global; a->c0 = col0; a->n = ncols_local; a->storage = SMK_STORE_F32;
    a->sparse = true; a->nnz = nnz;
    a->st = g_stream;
    register_matrix(a);
    int rc = 0;
    rc |= dev_alloc(&a->colptr, (size_t)ncols_local + 1);
    rc |= dev_alloc(&a->colptr_t, (size_t)height + 1);
    rc |= dev_alloc(&a->rowidx, (size_t)nnz);
    rc |= dev_alloc(&a->rowidx_t, (size_t)nnz);
    rc |= dev_alloc(&a->val, (size_t)nnz);
    rc |= dev_alloc(&a->val_t, (size_t)nnz);
    if (rc) { smk_matrix_destroy(a); return SMK_DEVICE_ERROR; }
    hipError_t e = hipMemcpy(a->colptr, cp.data(), cp.size() * sizeof(i64), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0) e = hipMemcpy(a->rowidx, row_indices + base, (size_t)nnz * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0) e = hipMemcpy(a->val, data + base, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice);
    // the transpose: a stable radix sort by row on the device (sort.hip) -- the entry order of the host counting sort --
    // or, if that is not available (SMK_TRANSPOSE=host forces it), the host routine and a second upload
    static const bool host_tr = [] { const char* ev = getenv("SMK_TRANSPOSE"); return ev && ev[0] == 'h'; }();
    bool done = false;
    if (e == hipSuccess && !host_tr)
        done = device_csc_transpose(height, ncols_local, nnz, a->colptr, a->rowidx, a->val, a->colptr_t, a->rowidx_t, a->val_t, g_stream) == 0;
    if (e == hipSuccess && !done) {
        std::vector<unsigned> rit((size_t)(nnz > 0 ? nnz : 1)), cpt32((size_t)height + 1);
        std::vector<double> vt((size_t)(nnz > 0 ? nnz : 1));
        const int trc = smk_csc_transpose(height, ncols_local, col_offsets, row_indices, data, cpt32.data(), rit.data(), vt.data());
        if (trc != SMK_OK) { smk_matrix_destroy(a); return trc; }
        std::vector<i64> cpt((size_t)height + 1);
        for (int64_t r = 0; r <= height; ++r) cpt[(size_t)r] = cpt32[(size_t)r];
        e = hipMemcpy(a->colptr_t, cpt.data(), cpt.size() * sizeof(i64), hipMemcpyHostToDevice);
        if (e == hipSuccess && nnz > 0) e = hipMemcpy(a->rowidx_t, rit.data(), (size_t)nnz * sizeof(unsigned), hipMemcpyHostToDevice);
        if (e == hipSuccess && nnz > 0) e = hipMemcpy(a->val_t, vt.data(), (size_t)nnz * sizeof(double), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        set_error(std::string("hipMemcpy(CSC): ") + hipGetErrorString(e));
        smk_matrix_destroy(a);
        return SMK_DEVICE_ERROR;
    }
    *out = a;
    return SMK_OK;
}

// host copy of a resident CSC (32-bit offsets), fetched on first use: only column subsets whose list is not strictly
// increasing are cut on the host
static int ensure_host_csc(const smk_matrix* a)
{
    if (!a->h_colptr.empty()) return SMK_OK;
    a->h_colptr.resize((size_t)a->n + 1);
    a->h_rowidx.resize((size_t)(a->nnz > 0 ? a->nnz : 1));
    a->h_val.resize((size_t)(a->nnz > 0 ? a->nnz : 1));
    const int rc = smk_matrix_download_csc(a, 0, a->h_colptr.data(), a->h_rowidx.data(), a->h_val.data());
    if (rc != SMK_OK) { a->h_colptr.clear(); return rc; }
    return SMK_OK;
}

// Column subset of a resident matrix as a new matrix (HierNMF2 node, SubMatrixColsCompact).
// Dense (dense_matrix_impl.hpp:224-281): all rows kept, columns gathered HBM -> HBM, transpose rebuilt
// on the device.  Sparse (sparse_matrix_impl.hpp:479-590): rows without a stored entry in the selected
// columns are dropped; the cut is made on the host copy of the CSC and uploaded.
int smk_matrix_gather_cols(const smk_matrix* src, const unsigned* cols, int64_t ncols, smk_matrix** out,
                           unsigned* new_to_old_rows, int64_t* new_height)
{
    if (!out) return SMK_BAD_PARAM;
    *out = nullptr;
    if (!src || !cols || ncols <= 0) { set_error("SubMatrixColsCompact: empty column set"); return SMK_BAD_PARAM; }
    for (int64_t j = 0; j < ncols; ++j)
        if ((i64)cols[j] >= src->n) { set_error("SubMatrixColsCompact: column index out of range"); return SMK_BAD_PARAM; }
    if (!src->sparse) {
        smk_matrix* a = nullptr;
        int rc = smk_matrix_create(&a, src->m, ncols, 0, ncols, src->storage);
        if (rc) return rc;
        unsigned* dcols = nullptr;
        rc = dev_alloc(&dcols, (size_t)ncols);
        if (rc) { smk_matrix_destroy(a); return rc; }
        const i64 es = elem_size(src->storage);
        hipError_t e = hipMemcpyAsync(dcols, cols, (size_t)ncols * sizeof(unsigned), hipMemcpyHostToDevice, g_stream);
        if (e == hipSuccess) {
            rc = launch_gather_cols(src->A, src->ldA * es, dcols, ncols, a->A, a->ldA * es, src->ldA * es, g_stream);
            if (!rc) rc = matrix_make_transpose(a);
            if (!rc) e = hipStreamSynchronize(g_stream);
        }
        (void)smk::dev_free(dcols);
        if (e != hipSuccess) { set_error(std::string("gather_cols: ") + hipGetErrorString(e)); rc = SMK_DEVICE_ERROR; }
        if (rc) { smk_matrix_destroy(a); return rc; }
        if (new_to_old_rows) for (i64 r = 0; r < src->m; ++r) new_to_old_rows[r] = (unsigned)r;
        if (new_height) *new_height = src->m;
        *out = a;
        return SMK_OK;
    }
    // strictly increasing column lists (every HierNMF2 document list): cut on the device, nothing but
    // the row map crosses PCIe (sparse_subset.hip).  SMK_SPARSE_SUBSET=host forces the host cut below.
    bool increasing = true;
    for (int64_t j = 1; j < ncols && increasing; ++j) increasing = cols[j] > cols[j - 1];
    static const bool force_host = [] { const char* e = getenv("SMK_SPARSE_SUBSET"); return e && e[0] == 'h'; }();
    if (increasing && !force_host) {
        SparseDev sd, od;
        sd.m = src->m; sd.n = src->n; sd.nnz = src->nnz;
        sd.colptr = src->colptr; sd.rowidx = src->rowidx; sd.val = src->val;
        sd.colptr_t = src->colptr_t; sd.rowidx_t = src->rowidx_t; sd.val_t = src->val_t;
        std::vector<unsigned> n2o_tmp;
        unsigned* n2o = new_to_old_rows;
        if (!n2o) { n2o_tmp.resize((size_t)src->m); n2o = n2o_tmp.data(); }
        const int rc = device_sparse_subset(sd, cols, ncols, &od, n2o, g_stream);
        if (rc) return rc == -3 ? SMK_BAD_PARAM : SMK_DEVICE_ERROR;
        smk_matrix* a = new smk_matrix;
        a->m = od.m; a->n_global = ncols; a->c0 = 0; a->n = ncols; a->storage = SMK_STORE_F32;
        a->sparse = true; a->nnz = od.nnz;
        a->st = g_stream;
        register_matrix(a);
        a->colptr = od.colptr; a->rowidx = od.rowidx; a->val = od.val;
        a->colptr_t = od.colptr_t; a->rowidx_t = od.rowidx_t; a->val_t = od.val_t;
        if (new_height) *new_height = od.m;
        *out = a;
        return SMK_OK;
    }
    int64_t nh = 0, nz = 0;
    int rc = ensure_host_csc(src);
    if (rc != SMK_OK) return rc;
    rc = smk_csc_subset_cols_compact(src->m, src->n, src->h_colptr.data(), src->h_rowidx.data(), src->h_val.data(), cols,
                                         ncols, nullptr, nullptr, nullptr, nullptr, nullptr, &nh, &nz);
    if (rc != SMK_OK) return rc;
    std::vector<unsigned> cp((size_t)ncols + 1), ri((size_t)nz);
    std::vector<double> va((size_t)nz);
    rc = smk_csc_subset_cols_compact(src->m, src->n, src->h_colptr.data(), src->h_rowidx.data(), src->h_val.data(), cols,
                                     ncols, cp.data(), ri.data(), va.data(), nullptr, new_to_old_rows, &nh, &nz);
    if (rc != SMK_OK) return rc;
    if (new_height) *new_height = nh;
    return smk_matrix_create_sparse(out, nh, ncols, 0, ncols, nz, cp.data(), ri.data(), va.data());
}

// ------------------------------------------------------------------------------------------
// solver
// ------------------------------------------------------------------------------------------
static PartialView view1(const smk_solver* s)
{
    return PartialView{s->P1, s->pl1.S, (i64)s->pl1.ncols_pad * s->kpp, s->kpp, 1};
}
static inline bool is_dist(const smk_solver* s) { return s->ar != nullptr || s->comm != nullptr; }
// BPP with a native communicator: every rank solves only its own blocks of W, so it needs only those rows of the summed
// (AH')' -- a reduce-scatter instead of an all-reduce (half the bytes on the wire) -- and the other ranks receive the
// packed streaming operand of those rows (4 B per entry in the fp16 form) instead of the fp64 values
static inline bool w_rows_sharded(const smk_solver* s) { return s->w_sharded; }

static PartialView view2(const smk_solver* s)
{
    if (is_dist(s)) return PartialView{s->R2red, 1, 0, s->kpp, s->red_f64 ? 1 : 0};
    return PartialView{s->P2, s->pl2.S, (i64)s->pl2.ncols_pad * s->kpp, s->kpp, 1};
}

static PartialView view_own(const smk_solver* s) { return PartialView{s->R2own, 1, 0, s->kpp, s->red_f64 ? 1 : 0}; }

// rows [r0, r1) of chunk j (clipped to the padded row count of the H*At pass) and this rank's block [a, b) of it
// (valid rows only: b <= m; empty when b <= a)
static inline void chunk_rows(const smk_solver* s, int j, i64* r0, i64* r1)
{
    *r0 = (i64)j * s->world * s->blk;
    *r1 = std::min<i64>(*r0 + (i64)s->world * s->blk, s->pl2.ncols_pad);
}
static inline void own_block(const smk_solver* s, int j, i64* a, i64* b)
{
    *a = ((i64)j * s->world + s->rank) * s->blk;
    *b = std::min<i64>(*a + s->blk, s->m);
}

// workspace layout: [R2red: rows x kpp f32|f64][Gh: KP*KP f64][scal: 8 f64][Wt: KP x rows f64].  rows = the padded
// row count (callback hook: one all-reduce per buffer, any world) or world * nchunk * blk with a native communicator
// (equal blocks for the reduce-scatter / all-gather)
static inline i64 comm_rows(const smk_solver* s) { return s->comm ? std::max<i64>(s->rows_cap, s->pl2.ncols_pad) : s->pl2.ncols_pad; }
static size_t comm_bytes(const smk_solver* s)
{
    size_t b = (size_t)comm_rows(s) * s->kpp * (s->red_f64 ? sizeof(double) : sizeof(float));
    b = (b + 255) / 256 * 256;
    b += (size_t)s->KP * s->KP * sizeof(double);
    b = (b + 255) / 256 * 256;
    b += 8 * sizeof(double);
    b = (b + 255) / 256 * 256;
    b += (size_t)s->KP * (size_t)comm_rows(s) * sizeof(double);
    return b;
}

// One pass over a dense A at HBM rate, once per matrix contents: the column maxima of |A| give
//   ascale          power of two with max |A| ascale in [2^13, 2^14) (fp16 two-term products; 1 for an all-zero matrix)
//   col_spread_log2 log2 of (largest / smallest non-zero column maximum): how far apart the column scales are
static int matrix_measure_scale(const smk_matrix* a, hipStream_t st)
{
    unsigned* d = nullptr;
    SMK_HIP(smk::dev_malloc((void**)&d, 2 * sizeof(unsigned)));
    unsigned bits[2] = {0, 0};
    int rc = launch_colrange(a->A, a->storage, a->ldA, a->m, a->n, d, st);
    if (!rc && hipMemcpyAsync(bits, d, sizeof(bits), hipMemcpyDeviceToHost, st) != hipSuccess) rc = SMK_DEVICE_ERROR;
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = SMK_DEVICE_ERROR;
    (void)smk::dev_free(d);
    if (rc) { set_error("could not measure max |A|"); return rc; }
    float mx, mn;
    memcpy(&mx, &bits[0], sizeof(mx));
    memcpy(&mn, &bits[1], sizeof(mn));
    float sc = 1.f;
    int spread = 0;
    if (mx > 0.f && std::isfinite(mx)) {
        int ex = 0;
        (void)frexpf(mx, &ex);                     // mx = f 2^ex, f in [0.5, 1)
        sc = ldexpf(1.f, 14 - ex);
        if (bits[1] != 0xFFFFFFFFu && mn > 0.f) { int en = 0; (void)frexpf(mn, &en); spread = ex - en; }
    }
    a->ascale = sc;
    a->col_spread_log2 = spread;
    return 0;
}

// One pass over A and one over A' at HBM rate, once per matrix contents: the largest 2-norm of a column and of a row
// (what bounds the NNLS solutions from above, NnlsPack)
static int matrix_measure_norms(const smk_matrix* a, hipStream_t st)
{
    double* d = nullptr;
    SMK_HIP(smk::dev_malloc((void**)&d, 2 * sizeof(double)));
    double v[2] = {0.0, 0.0};
    int rc = launch_colnorm2_max(a->A, a->storage, a->ldA, a->m, a->n, d, st);
    if (!rc) rc = launch_colnorm2_max(a->At, a->storage, a->ldAt, a->n, a->m, d + 1, st);
    if (!rc && hipMemcpyAsync(v, d, sizeof(v), hipMemcpyDeviceToHost, st) != hipSuccess) rc = SMK_DEVICE_ERROR;
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = SMK_DEVICE_ERROR;
    (void)smk::dev_free(d);
    if (rc) { set_error("could not measure the column / row norms of A"); return rc; }
    a->colnorm_max = std::sqrt(v[0]);
    a->rownorm_max = std::sqrt(v[1]);
    return 0;
}

// What the transposed-source kernels cover: MU, HALS and BPP with the 16-bit product forms (the reference's BPP keeps a transpose
// itself, nmf_solver_bpp.hpp:319 -- its W-side right-hand side H A' is the same product as MU's and HALS's, so it does not have
// to; RANK2 and the accurate form contract down the contiguous direction of A' on the vector ALUs / fp64 matrix cores).
static bool single_copy_serves(const smk_solver* s)
{
    const int alg = s->o.algorithm;
    const bool ok_alg = alg == SMK_ALG_MU || alg == SMK_ALG_HALS || alg == SMK_ALG_BPP;
    const bool ok_form = s->a->storage == SMK_STORE_BF16 ? (s->nsplit >= 1 && s->nsplit <= 3) : (s->nsplit == 3 || s->nsplit == NSPLIT_F16X2);
    return ok_alg && ok_form;
}

// Everything that depends on the product form (s->nsplit): the launch plans of both passes and the fp16 row scales ...
static int plan_products(smk_solver* s)
{
    const smk_matrix* a = s->a;
    if (s->nsplit == NSPLIT_F16X2 && a->ascale == 0.f) {
        const int rc0 = matrix_measure_scale(a, s->st);
        if (rc0) return rc0;
    }
    if (a->single && !a->sparse && !single_copy_serves(s)) {
        // every (re-)plan passes here -- smk_solver_create, the form agreement of attach_comm, guard_resolve, smk_solver_nnls_hals:
        // a single-copy matrix whose run leaves what the transposed-source kernels cover gets its stored transpose NOW, once
        // (the matrix is an ordinary one afterwards; solvers already planned on the transposed source keep reading A)
        const int trc = matrix_materialize_transpose(a);
        if (trc) return trc;
    }
    s->ng = plan_bigprod_groups(a->storage, s->k, s->m, s->n, s->nsplit, g_cus, s->pg1);
    if (a->single) {                                                                                        // H*A' from A itself
        if (plan_bigprod_groups_tr(a->storage, s->k, s->n, s->m, s->nsplit, g_cus, s->pg2) < 0) {
            set_error("no transposed-source kernel for this product form");
            return SMK_UNSUPPORTED;
        }
    } else (void)plan_bigprod_groups(a->storage, s->k, s->n, s->m, s->nsplit, g_cus, s->pg2);
    if (s->nsplit == NSPLIT_F64)
        for (int g = 0; g < s->ng; ++g) { s->pg1[g].ldx = s->KP; s->pg2[g].ldx = s->KP; }
    {
        // Cache policy of the streamed loads (BigProdPlan::temporal): what one iteration streams -- A and, unless the matrix is a
        // single copy, A' -- either stays in the 256 MB Infinity Cache between the passes or it does not.  Measured crossover:
        // profiles/r05_cache_policy_ab.txt.  SMK_BP_TEMPORAL=0/1 forces a policy.
        const double streamed = (double)s->m * (double)s->n * elem_size(a->storage) * (a->single ? 1.0 : 2.0) * s->ng;
        static const int forced = [] { const char* e = getenv("SMK_BP_TEMPORAL"); return e ? atoi(e) : -1; }();
        const int temporal = forced >= 0 ? (forced ? 1 : 0) : (streamed <= 300.0e6 ? 1 : 0);
        for (int g = 0; g < s->ng; ++g) { s->pg1[g].temporal = temporal; s->pg2[g].temporal = temporal; }
    }
    s->pl1 = s->pg1[0];
    s->pl2 = s->pg2[0];
    if (s->nsplit == NSPLIT_F16X2) {
        int rc = 0;
        for (int side = 0; side < 2 && !rc; ++side) {
            if (!s->xscale[side]) rc |= dev_alloc(&s->xscale[side], (size_t)MAX_K);
            if (!s->oscale[side]) rc |= dev_alloc(&s->oscale[side], (size_t)MAX_K);
        }
        if (rc) return SMK_DEVICE_ERROR;
        for (int g = 0; g < s->ng; ++g) {
            s->pg1[g].oscale = s->oscale[0] + s->pg1[g].k0;  s->pg1[g].ascale = a->ascale;
            s->pg2[g].oscale = s->oscale[1] + s->pg2[g].k0;  s->pg2[g].ascale = a->ascale;
        }
        s->pl1 = s->pg1[0];
        s->pl2 = s->pg2[0];
    }
    const char* epk = getenv("SMK_NNLS_PACK");                      // read per plan, like SMK_NSPLIT (0 = the separate reduce-and-pack launch)
    const bool pack_env = !(epk && epk[0] == '0');
    s->pack_in_solve = pack_env && !s->pack_in_solve_off && s->o.algorithm == SMK_ALG_BPP && s->KP == 16 && s->nsplit == NSPLIT_F16X2 &&
                       !a->sparse && !a->single && s->ng == 1 && !is_dist(s) && !s->comm && bigprod_supports_tail(s->pl1) && bigprod_supports_tail(s->pl2);
    if (s->pack_in_solve && (a->colnorm_max < 0.0 || a->rownorm_max < 0.0)) {
        const int rc0 = matrix_measure_norms(a, s->st);
        if (rc0) return rc0;
    }
    return 0;
}
// ... and the buffers sized by those plans: the packed operands and the partial products (dense A)
static int alloc_product_buffers(smk_solver* s)
{
    int rc = 0;
    void** bufs[] = {&s->packW, &s->packH, (void**)&s->P1, (void**)&s->P2};
    for (void** b : bufs)
        if (*b) { (void)smk::dev_free(*b); *b = nullptr; }
    if (!s->a->sparse) {
        rc |= dev_alloc((unsigned char**)&s->packW, packed_bytes(s->a->storage, s->k, s->m, s->nsplit));
        rc |= dev_alloc((unsigned char**)&s->packH, packed_bytes(s->a->storage, s->k, s->n, s->nsplit));
    }
    rc |= dev_alloc(&s->P1, s->pl1.p_elems);
    rc |= dev_alloc(&s->P2, s->pl2.p_elems);
    return rc;
}

int smk_solver_create(smk_solver** out, const smk_options* opts, const smk_matrix* a)
{
    if (!out) return SMK_BAD_PARAM;
    *out = nullptr;
    if (!g_init) { set_error("smk_initialize() has not been called"); return SMK_NOTINITIALIZED; }
    if (!opts || !a) return SMK_BAD_PARAM;
    if (!smk_is_valid(opts, 1)) return SMK_BAD_PARAM;
    if (opts->height != a->m || opts->width != a->n_global) { set_error("options/matrix dimension mismatch"); return SMK_BAD_PARAM; }
    if (opts->k > MAX_K || (opts->algorithm == SMK_ALG_BPP && opts->k > MAX_K_BPP)) { set_error("device path supports k <= 2048"); return SMK_UNSUPPORTED; }
    // W and H element counts must fit the reference's 32-bit index (nmf.cpp:194-210)
    if ((uint64_t)a->m * (uint64_t)opts->k > 0x7FFFFFFFull) { fprintf(stderr, "W matrix size too large\n"); return SMK_SIZE_TOO_LARGE; }
    if ((uint64_t)a->n_global * (uint64_t)opts->k > 0x7FFFFFFFull) { fprintf(stderr, "H matrix size too large\n"); return SMK_SIZE_TOO_LARGE; }

    smk_solver* s = new smk_solver;
    ++g_live_solvers;
    s->o = *opts;
    s->a = a;
    s->k = opts->k;
    s->KP = kp_of(s->k);
    s->kpp = kpp_of(s->k);
    s->m = a->m;
    s->n = a->n;
    s->st = a->st ? a->st : g_stream;
    const char* env = getenv("SMK_NSPLIT");
    // fp32 A: the fp16 two-term form (3 MFMAs per product, 2^-22 operand error) unless SMK_NSPLIT picks the bf16 forms
    // (3 = bf16x3, 6 MFMAs, power-bound; 2 = two bf16 terms, 2^-16); bf16 A: three bf16 terms of the factor
    // HALS amplifies the product error several thousand times (its W update is a difference of nearly equal terms per
    // column, more so at high rank): the 24-bit bf16x3 operands stay inside the parity bar where the 22-bit fp16 ones
    // do not always (tools/fuzz_parity.py: k = 117 .. 253 1e-4 .. 5e-4 against < 1e-4, one k = 62 case at 1.1e-4 after
    // 17 iterations), so HALS keeps bf16x3; MU and BPP take the fp16 form
    // The amplification grows with the rank for every algorithm (BPP k = 191 after 34 iterations: 1.26e-4 with the
    // fp16 form), so above k = 64 the 24-bit operands are kept as well; C2 (k = 16) and C4 (k = 64) take the fp16 form.
    const bool hals = opts->algorithm == SMK_ALG_HALS;
    // HALS above k = 64: the 1e-8-class products of the 16-bit forms, amplified ~2x per five iterations at these ranks, leave
    // the 1e-4 parity bar in runs of 30+ iterations (1.4e-4 .. 1.6e-4 at k = 100 .. 150 after 30) -- with either 16-bit form
    // and with bf16 A alike, so it is the accumulation, not the operands.  Those runs take the ACCURATE form (NSPLIT_F64:
    // the stored A against the fp64 factor on the fp64 matrix cores), ~3x the product time.  SMK_NSPLIT=8 selects it anywhere.
    int nsplit_default = (a->storage == SMK_STORE_F32 && !hals && opts->k <= 64) ? NSPLIT_F16X2 : 3;
    if (hals && opts->k > 64 && !a->sparse) nsplit_default = NSPLIT_F64;
    // Block pivoting above k = 64 as well: the Gram matrix of a uniform start has condition ~3k, so 1e-8-class products are
    // 2e-6 in the factors after ONE iteration at k = 100, and on data with sparse planted factors that grows ~1.3x per iteration
    // (tools/wide_long_run.py: 1200 x 1000, k = 100, bf16x3: 1.2e-4 after 20 iterations, 1.1e-3 after 30; k = 160: 4e-5 after 25;
    // the accurate form: 8.5e-12 after 30).  At k <= 64 the same data stays below 2e-5 over 30 iterations with the fp16 form.
    if (opts->algorithm == SMK_ALG_BPP && opts->k > 64 && !a->sparse) nsplit_default = NSPLIT_F64;
    // ... and at k in (32, 64] where the accurate form costs nothing measurable (A of at most 2^24 entries: the streaming pass
    // is a 10 - 30 us launch next to a 100 us NNLS).  Block pivoting at these ranks amplifies the product error ~3000 x while
    // the passive sets are still moving: on data with sparse planted factors (1500 x 1100, k = 64) the distance to the oracle's
    // trajectory peaks at 0.4e-4 .. 2e-4 around iteration 100 with the fp16 form whatever its fold interval (3e-8 .. 9e-8
    // products), 5e-6 with bf16x3, 2e-12 with the accurate form, and contracts to 4e-7 by iteration 500
    // (profiles/r04_long_runs_500_iterations.txt).  Larger matrices keep the fp16 form (C4: 3 x the pass time otherwise);
    // SMK_NSPLIT=8 / 3 select the other forms anywhere.
    // (SMK_BPP_SMALL_ACCURATE=0 keeps the fp16 form: the test suite sets it, its small cases stand in for C4's path.)
    const char* esa = getenv("SMK_BPP_SMALL_ACCURATE");             // read per solver, like SMK_NSPLIT
    const bool small_accurate = !(esa && esa[0] == '0');
    if (small_accurate && opts->algorithm == SMK_ALG_BPP && opts->k > 32 && !a->sparse && a->m * a->n_global <= ((i64)1 << 24))
        nsplit_default = NSPLIT_F64;
    // Dense RANK2 (every node factorisation of HierNMF2 / flatclust on dense A runs 100 .. 1000 iterations to a tight
    // tolerance): the accurate form costs nothing at two factor rows -- bigprod_f64_k2_kernel does its 2 fp64 multiply-adds
    // per stored entry on the vector ALUs at the streaming rate -- and leaves only summation order between this path and
    // the reference's fp64 arithmetic (common/src/nmf.cpp:33).
    if (opts->algorithm == SMK_ALG_RANK2 && !a->sparse) nsplit_default = NSPLIT_F64;
    // Column scales of A more than 2^28 apart: the small columns fall below what fp32-class products resolve next to the
    // large ones (HALS / BPP leave the bar at 2^+-20, tests/test_gpu_parity.py) -- the accurate form as well.
    if (!env && !a->sparse && opts->algorithm != SMK_ALG_RANK2) {
        if (a->col_spread_log2 < 0) { const int rc0 = matrix_measure_scale(a, a->st ? a->st : g_stream); if (rc0) { --g_live_solvers; delete s; return rc0; } }
        if (a->col_spread_log2 > 28) nsplit_default = NSPLIT_F64;
    }
    s->nsplit = env ? atoi(env) : nsplit_default;
    if (s->nsplit != NSPLIT_F64 && (s->nsplit < 1 || s->nsplit > NSPLIT_F16X2)) s->nsplit = nsplit_default;
    if (s->nsplit == NSPLIT_F64 && a->sparse) s->nsplit = 3;      // sparse A: gather products in fp64 already
    // the fp16 two-term form applies to fp32 storage; RANK2 keeps its Gram matrices inside its own solve kernel
    if (s->nsplit == NSPLIT_F16X2 && (a->storage != SMK_STORE_F32 || a->sparse || opts->algorithm == SMK_ALG_RANK2)) s->nsplit = 3;
    // (a single-copy matrix outside what the transposed-source kernels cover gets its stored transpose inside plan_products)
    int rc = plan_products(s);
    if (rc) { smk_solver_destroy(s); return rc; }
    if (a->sparse) {   // gather products write one slab, as dense as the factor layout (KP values per column; RANK2: the 2 live ones)
        s->kpp = (opts->algorithm == SMK_ALG_RANK2) ? 2 : s->KP;
        int nb1 = 1, nb2 = 1;
        if (opts->algorithm == SMK_ALG_RANK2) {
            // a factor that does not fit an XCD's L2 is gathered block by block (one partial slab per row block)
            if (!a->blocked_tried) {
                a->blocked_tried = true;
                const int b1 = blocked_csc_blocks(a->m), b2 = blocked_csc_blocks(a->n);
                hipStream_t bst = a->st ? a->st : g_stream;
                if (b1 > 1) (void)build_blocked_csc(a->m, a->n, a->nnz, a->colptr, a->rowidx, a->val, b1, &a->bA, bst);
                if (b2 > 1) (void)build_blocked_csc(a->n, a->m, a->nnz, a->colptr_t, a->rowidx_t, a->val_t, b2, &a->bAt, bst);
            }
            nb1 = a->bA.nb > 1 ? a->bA.nb : 1;
            nb2 = a->bAt.nb > 1 ? a->bAt.nb : 1;
        } else if (!is_wide(opts->k)) {
            // MU / HALS / BPP at ranks up to 128: the gather products work on entry-balanced segments (spmm_seg.hip);
            // SMK_SPMM_SEG=0 keeps the column-per-lane-group kernel of round 4
            ensure_seg_plans(a);
        }
        s->pl1.S = nb1; s->pl1.p_elems = (size_t)nb1 * s->pl1.ncols_pad * s->kpp;
        s->pl2.S = nb2; s->pl2.p_elems = (size_t)nb2 * s->pl2.ncols_pad * s->kpp;
    }

    const size_t kk = (size_t)s->KP * s->KP;
    rc |= dev_alloc(&s->H, (size_t)s->KP * s->n);
    rc |= dev_alloc(&s->Wt_own, (size_t)s->KP * s->m);
    rc |= dev_alloc(&s->Gw, kk);
    rc |= dev_alloc(&s->Gh_own, kk);
    {
        size_t gs = gram_scratch_elems(s->k, GRAM_BLOCKS);
        if (opts->algorithm == SMK_ALG_HALS && !a->sparse && (s->KP == 16 || s->KP == 32)) {
            // the sweeps' own Gram partials (HalsEpilogue): one per workgroup -- 1024 / KP columns of H, 256 rows of W
            const i64 need = std::max<i64>((s->n * s->KP + 1023) / 1024, (s->m + 255) / 256);
            if (need <= 4096) { s->hals_ep_blocks = (int)std::max<i64>(need, 1); gs = std::max(gs, gram_scratch_elems(s->k, s->hals_ep_blocks)); }
        }
        if (s->KP == 16 && opts->algorithm == SMK_ALG_BPP) gs = std::max(gs, (size_t)NNLS_GRAM_MAX * (256 + 16) + 8);   // partials from the NNLS launch
        rc |= dev_alloc(&s->gram_scratch, gs);
        if (s->o.algorithm == SMK_ALG_RANK2) {
            const size_t e1 = rank2_gram_scratch_elems(std::max(s->m, s->n)), e2 = rank2_progress_scratch_elems(s->m, s->n);
            rc |= dev_alloc(&s->r2_scratch, e1);
            rc |= dev_alloc(&s->r2_prog, e2);
            rc |= dev_alloc(&s->Graw, (size_t)s->KP * s->KP);
            if (a->sparse) { rc |= dev_alloc(&s->Hc, (size_t)2 * s->n); rc |= dev_alloc(&s->Wc, (size_t)2 * s->m); }

        }
        if (!rc && hipMemsetAsync(s->gram_scratch, 0, gs * sizeof(double), s->st) != hipSuccess) rc |= 1;   // incl. the ticket word
    }
    rc |= dev_alloc(&s->tmpW, (size_t)s->KP * s->m);
    if (is_wide(s->k)) rc |= dev_alloc(&s->wide_tmp, (size_t)s->KP * std::max(s->m, s->n));
    // one partial per workgroup of the column-tile kernels (grid = N*(KP/4)/256 blocks) and at most
    // 2 x 512 for delta_fnorm
    s->pg_half = (size_t)((std::max(s->m, s->n) * (s->KP / 4) + 255) / 256) + 1024;
    if (is_wide(s->k)) s->pg_half = std::max(s->pg_half, (size_t)((std::max(s->m, s->n) + 3) / 4) + 1024);   // one partial per 4 columns
    rc |= dev_alloc(&s->pg_partials, 2 * s->pg_half);
    rc |= dev_alloc(&s->scal_own, (size_t)8);
    rc |= dev_alloc(&s->fail_flag, (size_t)1);
    rc |= alloc_product_buffers(s);
    if (opts->algorithm == SMK_ALG_HALS) {
        rc |= dev_alloc(&s->hals_scratch, hals_w_scratch_elems(s->k, s->m));
        rc |= dev_alloc(&s->W0c, (size_t)s->KP * s->m);
        rc |= dev_alloc(&s->H0c, (size_t)s->KP * s->n);
        if (!rc) rc = hals_w_scratch_init(s->hals_scratch, s->k, s->m, s->st);
    }
    if (s->pack_in_solve && !s->W0c) {      // pack_fail_soft goes back to the initial factors
        rc |= dev_alloc(&s->W0c, (size_t)s->KP * s->m);
        rc |= dev_alloc(&s->H0c, (size_t)s->KP * s->n);
    }
    if (opts->algorithm == SMK_ALG_BPP) {
        // k <= 128: two (inverse + selector) halves; above: one Cholesky panel per resident workgroup (wide.hip)
        rc |= dev_alloc(&s->nnls_scratch, nnls_uses_tiles(s->k) ? nnls_wide_scratch_elems(s->k, g_cus, std::max(s->m, s->n)) : 2 * nnls_scratch_elems(s->k));
        if (s->KP == 64 && !nnls_uses_tiles(s->k)) rc |= dev_alloc(&s->nnls_defer, nnls_defer_elems(std::max(s->m, s->n)));
        // (above k = 128 the inverse stays in stream order: beside the product it gained 1-2 % -- measured -- and the two sides
        // share one scratch there)
        // The second stream pays two event hops per solve (~8 us each way on the main stream): worth it where it hides the 47 us
        // inversion of k in (32, 64] behind a dense pass; at k in (16, 32] (13 us) stream order is as fast or faster (4096 x 2048,
        // k = 32: 115 -> 92 us per iteration; 8192 x 4096: 131 -> 125; 32768 x 8192: equal), and with a sparse matrix the inverse
        // rides in the gather product's launch (start_inverse).  SMK_INV_STREAM=1: the second stream everywhere, =0: nowhere.
        static const int inv_env = [] { const char* e = getenv("SMK_INV_STREAM"); return e ? atoi(e) : -1; }();
        const bool inv_beside = inv_env >= 0 ? inv_env != 0 : (!a->sparse && s->KP >= 64);
        if (inv_beside && (s->KP >= 64 || (s->KP == 32 && nnls_inverse_at_32())) && !nnls_uses_tiles(s->k)) {
            if (hipStreamCreateWithFlags(&s->st_inv, hipStreamNonBlocking) != hipSuccess) rc |= 1;
            for (int i = 0; i < 2 && !rc; ++i) {
                if (hipEventCreateWithFlags(&s->ev_g[i], hipEventDisableTiming) != hipSuccess) rc |= 1;
                if (hipEventCreateWithFlags(&s->ev_inv[i], hipEventDisableTiming) != hipSuccess) rc |= 1;
            }
        }
    }
    if (opts->prog_est_algorithm == SMK_PROG_DELTA_FNORM) rc |= dev_alloc(&s->Wprev, (size_t)s->KP * s->m);
    if (rc) { smk_solver_destroy(s); return SMK_DEVICE_ERROR; }
    s->Gh = s->Gh_own;
    s->scal = s->scal_own;
    s->Wt = s->Wt_own;
    *out = s;
    return SMK_OK;
}

void smk_solver_destroy(smk_solver* s)
{
    if (!s) return;
    if (s->st2) (void)hipStreamSynchronize(s->st2);
    if (s->st_inv) (void)hipStreamSynchronize(s->st_inv);
    void* ptrs[] = {s->H, s->Wt_own, s->Gw, s->Gh_own, s->gram_scratch, s->tmpW, s->pg_partials, s->scal_own,
                    s->fail_flag, s->packW, s->packH, s->P1, s->P2, s->hals_scratch, s->Wprev, s->tmpH, s->nnls_scratch, s->W0c, s->H0c,
                    s->xscale[0], s->xscale[1], s->oscale[0], s->oscale[1], s->r2_scratch, s->r2_prog, s->Graw, s->Hc, s->Wc, s->wide_tmp};
    for (void* p : ptrs)
        if (p) (void)smk::dev_free(p);
    for (int w = 0; w < 6; ++w)
        for (auto& e : s->ev[w]) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
    if (s->ev_cal) (void)hipEventDestroy(s->ev_cal);
    for (int b = 0; b < smk_solver::PROG_SLOTS; ++b) {
        if (s->snap[b]) (void)smk::dev_free(s->snap[b]);
        if (s->pev[b]) (void)hipEventDestroy(s->pev[b]);
    }
    if (s->pin) (void)hipHostFree(s->pin);
    { void* gq[] = {s->guard_As, s->guard_cols, s->guard_P, s->guard_dev}; for (void* q : gq) if (q) (void)smk::dev_free(q); }
    if (s->guard_pin) (void)hipHostFree(s->guard_pin);
    if (s->guard_ev) (void)hipEventDestroy(s->guard_ev);
    { void* r2p[] = {s->r2p_hc1, s->r2p_r2c, s->r2p_part, s->r2p_out, s->r2p_sync}; for (void* q : r2p) if (q) (void)smk::dev_free(q); }
    if (s->r2p_pin) (void)hipHostFree(s->r2p_pin);
    for (int b = 0; b < smk_solver::PROG_SLOTS; ++b) if (s->pin_r2[b]) (void)hipHostFree(s->pin_r2[b]);
    for (int b = 0; b < 2; ++b) if (s->seg_pieces[b]) (void)smk::dev_free(s->seg_pieces[b]);
    if (s->nnls_defer) (void)smk::dev_free(s->nnls_defer);
    if (s->comm_ws) (void)smk::dev_free(s->comm_ws);
    if (s->Wown) (void)smk::dev_free(s->Wown);
    if (s->R2own) (void)smk::dev_free(s->R2own);
    if (s->st2) (void)hipStreamDestroy(s->st2);
    if (s->st_inv) (void)hipStreamDestroy(s->st_inv);
    for (int i = 0; i < 2; ++i) {
        if (s->ev_g[i]) (void)hipEventDestroy(s->ev_g[i]);
        if (s->ev_inv[i]) (void)hipEventDestroy(s->ev_inv[i]);
    }
    hipEvent_t single[] = {s->ev_gram, s->ev_gh, s->ev_x, s->ev_y};
    for (hipEvent_t e : single) if (e) (void)hipEventDestroy(e);
    for (int j = 0; j < MAX_CHUNKS; ++j) {
        if (s->ev_c[j]) (void)hipEventDestroy(s->ev_c[j]);
        if (s->ev_r[j]) (void)hipEventDestroy(s->ev_r[j]);
        if (s->ev_a[j]) (void)hipEventDestroy(s->ev_a[j]);
    }
    --g_live_solvers;
    delete s;
}

int smk_solver_comm_workspace_bytes(const smk_solver* s, size_t* bytes)
{
    if (!s || !bytes) return SMK_BAD_PARAM;
    *bytes = comm_bytes(s);
    return SMK_OK;
}

static void carve_workspace(smk_solver* s, void* workspace)
{
    unsigned char* p = (unsigned char*)workspace;
    s->R2red = (float*)p;
    size_t b = (size_t)comm_rows(s) * s->kpp * (s->red_f64 ? sizeof(double) : sizeof(float));      // same layout as comm_bytes()
    b = (b + 255) / 256 * 256;
    s->Gh = (double*)(p + b);
    b += (size_t)s->KP * s->KP * sizeof(double);
    b = (b + 255) / 256 * 256;
    s->scal = (double*)(p + b);
    b += 8 * sizeof(double);
    b = (b + 255) / 256 * 256;
    s->Wt = (double*)(p + b);
}

int smk_solver_set_comm(smk_solver* s, int rank, int world, smk_allreduce_fn fn, void* user, void* workspace,
                        size_t workspace_bytes)
{
    if (!s || world < 1 || rank < 0 || rank >= world) return SMK_BAD_PARAM;
    if (world > 1 && !fn) return SMK_BAD_PARAM;
    if (s->comm) { set_error("a native communicator is attached"); return SMK_BAD_PARAM; }
    s->rank = rank; s->world = world;
    if (fn && (!workspace || workspace_bytes < comm_bytes(s))) return SMK_BAD_PARAM;
    s->ar = fn; s->ar_user = user;
    if (fn) {
        carve_workspace(s, workspace);
    } else {
        s->Gh = s->Gh_own;
        s->scal = s->scal_own;
        s->Wt = s->Wt_own;
    }
    return SMK_OK;
}

static int progress_prealloc(smk_solver* s);

// Native path: every collective is issued from C on the solver's second stream (RCCL over xGMI, or the in-process
// stand-in).  Call before set_factors().  The communicator must outlive the solver.
int smk_solver_attach_comm(smk_solver* s, smk_comm* comm)
{
    if (!s || !comm) return SMK_BAD_PARAM;
    if (s->ar || s->comm) { set_error("solver already has a communicator"); return SMK_BAD_PARAM; }
    s->comm = comm;
    s->rank = comm->rank; s->world = comm->world;
    // MEASUREMENT HOOK (bench.py --emulate-world N, one GPU): the geometry of rank 0 of N ranks -- row blocks, chunk
    // sizes, per-rank NNLS / Gram / packing work -- while the collectives run through the real one-rank communicator.
    // Times a rank's work without the transfers; the blocks of the other ranks never arrive, so the factors mean nothing.
    if (const char* e = getenv("SMK_COMM_EMULATE_WORLD"))
        if (comm->world == 1 && atoi(e) > 1 && atoi(e) <= 64) { s->world = atoi(e); s->rank = 0; }
    // The ranks must run ONE exchange protocol.  The product form is chosen per solver, and one input of that choice -- the
    // spread of the column scales -- is measured on the rank's LOCAL column shard: a rank whose shard alone spans more than
    // 2^28 would take the accurate form (and with it other buffers and other collectives) while its peers do not.  So the
    // form is agreed here, first thing, over the communicator: any rank that wants the accurate form moves all of them.
    // Every rank of the communicator calls attach, so this is a collective like the ones that follow.
    if (comm->world > 1) {
        double want = s->nsplit == NSPLIT_F64 ? 1.0 : 0.0;
        SMK_HIP(hipMemcpy(s->scal_own, &want, sizeof(double), hipMemcpyHostToDevice));
        int arc = comm_allreduce(comm, s->scal_own, 1, 1, s->st);
        if (arc) { s->comm = nullptr; return arc; }
        SMK_HIP(hipStreamSynchronize(s->st));
        SMK_HIP(hipMemcpy(&want, s->scal_own, sizeof(double), hipMemcpyDeviceToHost));
        if (want > 0.0 && s->nsplit != NSPLIT_F64 && !s->a->sparse) {
            s->nsplit = NSPLIT_F64;
            arc = plan_products(s);
            if (!arc && alloc_product_buffers(s)) arc = SMK_DEVICE_ERROR;
            if (arc) { s->comm = nullptr; return arc; }
        }
    }
    // chunk geometry: blocks of >= 4096 rows, at most 4 chunks (SMK_COMM_CHUNKS overrides: 1 .. 8), block a multiple of
    // 256 rows (the column tile of the streaming kernels and a whole number of packed chunk pairs)
    {
        const i64 mpad = s->pl2.ncols_pad;
        i64 c = mpad / ((i64)s->world * 4096);
        c = std::max<i64>(1, std::min<i64>(c, 4));
        if (const char* e = getenv("SMK_COMM_CHUNKS")) c = std::max(1, std::min(atoi(e), MAX_CHUNKS));
        s->blk = round_up((mpad + (i64)s->world * c - 1) / ((i64)s->world * c), 256);
        s->nchunk = (int)((mpad + (i64)s->world * s->blk - 1) / ((i64)s->world * s->blk));      // chunks that hold rows
        s->rows_cap = (i64)s->nchunk * s->world * s->blk;
    }
    {
        // fp64 on the wire unless SMK_COMM_F64=0.  SURVEY 8e suggested fp32 (half the bytes); measured at the full C4 size on
        // the bench's noise-like data (tools/shard8_fullsize.py: 8 shards against one GPU): fp32 costs 1.1e-4 in W after ONE
        // iteration -- the sums over 65536 columns are ~500, their fp32 rounding 3e-5, and HH' of such data amplifies it a
        // thousand times -- against 6e-7 with fp64.  With the exchange pipelined behind the pass the extra bytes are hidden
        // except in the last chunk.
        const char* e = getenv("SMK_COMM_F64");
        s->red_f64 = !(e && atoi(e) == 0);
    }
    // BPP and MU update the rows of W independently of each other: every rank takes its own blocks (HALS normalises column by
    // column over ALL rows inside its sweep and keeps the replicated update)
    // (also under the accurate product form: its W'A pass reads the fp64 factor itself, so the fp64 own blocks are gathered per
    // chunk in place of the packed operand -- prod1_sharded)
    s->w_sharded = (s->o.algorithm == SMK_ALG_BPP || s->o.algorithm == SMK_ALG_MU) && !s->a->sparse && (s->world > 1 || comm_forced());
    const size_t bytes = comm_bytes(s);
    if (smk::dev_malloc(&s->comm_ws, bytes) != hipSuccess) { s->comm = nullptr; s->w_sharded = false; set_error("smk::dev_malloc(comm workspace)"); return SMK_DEVICE_ERROR; }
    SMK_HIP(hipMemsetAsync(s->comm_ws, 0, bytes, s->st));
    carve_workspace(s, s->comm_ws);
    if (!s->a->sparse && s->pl2.S == 1 && s->red_f64) {
        // one row split and fp64 on the wire: the partial products ARE the send buffer.  They get room for the equal
        // blocks of the last chunk (rows past the padded row count are never written and stay zero).
        if (s->P2) (void)smk::dev_free(s->P2);
        s->P2 = nullptr;
        const size_t pe = (size_t)comm_rows(s) * s->kpp;
        if (dev_alloc(&s->P2, pe)) return SMK_DEVICE_ERROR;
        SMK_HIP(hipMemsetAsync(s->P2, 0, pe * sizeof(double), s->st));
        s->R2red = (float*)s->P2;
        s->r2_alias = true;
    }
    if (s->w_sharded) {
        // the packed operand of W is gathered in equal blocks: room for rows_cap rows, groups laid out for that length
        if (s->packW) (void)smk::dev_free(s->packW);
        s->packW = nullptr;
        const size_t pb = packed_bytes(s->a->storage, s->k, s->rows_cap, s->nsplit);
        if (dev_alloc((unsigned char**)&s->packW, pb)) return SMK_DEVICE_ERROR;
        SMK_HIP(hipMemsetAsync(s->packW, 0, pb, s->st));
        size_t off = 0;
        for (int g = 0; g < s->ng; ++g) {
            s->pg1[g].pack_offset = off;
            off += packed_bytes(s->a->storage, s->pg1[g].kg, s->rows_cap, s->nsplit);
        }
        s->pl1.pack_offset = s->pg1[0].pack_offset;
        const size_t own_rows = (size_t)s->nchunk * s->blk;
        const size_t rb = s->red_f64 ? sizeof(double) : sizeof(float);
        if (dev_alloc(&s->Wown, own_rows * s->KP)) return SMK_DEVICE_ERROR;
        if (dev_alloc((unsigned char**)&s->R2own, own_rows * s->kpp * rb)) return SMK_DEVICE_ERROR;
        SMK_HIP(hipMemsetAsync(s->Wown, 0, own_rows * s->KP * sizeof(double), s->st));
        SMK_HIP(hipMemsetAsync(s->R2own, 0, own_rows * s->kpp * rb, s->st));
        s->n_own = 0;
        for (int j = 0; j < s->nchunk; ++j) {
            i64 a, b;
            own_block(s, j, &a, &b);
            if (b > a) s->n_own += b - a;
        }
    }
    SMK_HIP(hipStreamCreateWithFlags(&s->st2, hipStreamNonBlocking));
    hipEvent_t* evs[] = {&s->ev_gram, &s->ev_gh, &s->ev_x, &s->ev_y};
    for (hipEvent_t* e : evs) SMK_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (int j = 0; j < MAX_CHUNKS; ++j) {
        SMK_HIP(hipEventCreateWithFlags(&s->ev_c[j], hipEventDisableTiming));
        SMK_HIP(hipEventCreateWithFlags(&s->ev_r[j], hipEventDisableTiming));
        SMK_HIP(hipEventCreateWithFlags(&s->ev_a[j], hipEventDisableTiming));
    }
    // nothing is allocated once the collectives are in flight (several ranks may live in one process)
    return progress_prealloc(s);
}

// this rank's blocks of the full W (every rank holds all of it after set_factors / a gather / the final scaling) -> Wown
static int scatter_own(smk_solver* s)
{
    for (int j = 0; j < s->nchunk; ++j) {
        i64 a, b;
        own_block(s, j, &a, &b);
        if (b > a)
            SMK_HIP(hipMemcpyAsync(s->Wown + (i64)j * s->blk * s->KP, s->Wt + a * s->KP, (size_t)(b - a) * s->KP * sizeof(double),
                                   hipMemcpyDeviceToDevice, s->st));
    }
    return 0;
}

int smk_solver_set_factors(smk_solver* s, const double* W0, int64_t ldW, const double* H0, int64_t ldH)
{
    if (!s || !W0 || !H0) return SMK_BAD_PARAM;
    if (ldW < s->m || ldH < s->k) { set_error("leading dimension too small"); return SMK_BAD_PARAM; }
    if (s->nsplit == NSPLIT_F16X2 && (s->a->ascale == 0.f || s->a->ascale != s->pg1[0].ascale)) {
        // the matrix was refilled after this solver was created: its fp16 product scale follows the new contents
        if (s->a->ascale == 0.f) { const int rc0 = matrix_measure_scale(s->a, s->st); if (rc0) return rc0; }
        for (int g = 0; g < s->ng; ++g) s->pg1[g].ascale = s->pg2[g].ascale = s->a->ascale;
        s->pl1.ascale = s->pl2.ascale = s->a->ascale;
    }
    if (s->pack_in_solve && (s->a->colnorm_max < 0.0 || s->a->rownorm_max < 0.0)) {      // ... and so do the norms behind the packing NNLS launch
        const int rc0 = matrix_measure_norms(s->a, s->st);
        if (rc0) return rc0;
    }
    // pad rows of the KP x N device layout must be (and stay) zero
    SMK_HIP(hipMemsetAsync(s->Wt, 0, (size_t)s->KP * s->m * sizeof(double), s->st));
    SMK_HIP(hipMemsetAsync(s->H, 0, (size_t)s->KP * s->n * sizeof(double), s->st));
    // W0 (m x k, host) -> tmpW (m x k, ld m) -> Wt (KP x m)
    SMK_HIP(hipMemcpy2DAsync(s->tmpW, (size_t)s->m * sizeof(double), W0, (size_t)ldW * sizeof(double),
                             (size_t)s->m * sizeof(double), (size_t)s->k, hipMemcpyHostToDevice, s->st));
    int rc = launch_transpose_f64(s->tmpW, s->m, s->Wt, s->KP, s->m, s->k, s->st);
    if (rc) return rc;
    SMK_HIP(hipMemcpy2DAsync(s->H, (size_t)s->KP * sizeof(double), H0, (size_t)ldH * sizeof(double),
                             (size_t)s->k * sizeof(double), (size_t)s->n, hipMemcpyHostToDevice, s->st));
    const int big = INT_MAX;
    SMK_HIP(hipMemcpyAsync(s->fail_flag, &big, sizeof(int), hipMemcpyHostToDevice, s->st));
    if (s->W0c) {
        SMK_HIP(hipMemcpyAsync(s->W0c, s->Wt, (size_t)s->KP * s->m * sizeof(double), hipMemcpyDeviceToDevice, s->st));
        SMK_HIP(hipMemcpyAsync(s->H0c, s->H, (size_t)s->KP * s->n * sizeof(double), hipMemcpyDeviceToDevice, s->st));
    }
    if (s->w_sharded) { rc = scatter_own(s); if (rc) return rc; }
    SMK_HIP(hipStreamSynchronize(s->st));
    s->w_full = true;
    s->wc_valid = false;
    s->have_factors = true;
    s->inited = false;
    s->normalized = false;
    s->iter = 0;
    s->pg0 = 1.0;
    s->last_metric = 1.0;
    return SMK_OK;
}

// The same start as smk_solver_set_factors(W0, H0) with W0 = smk_uniform_fill_host(m x k, seed_w), H0 = smk_uniform_fill_host
// (k x n, seed_h) -- the RandomMatrix stand-in of every caller in this library -- generated on the device: no host fill, no
// upload.  (HierNMF2 draws two such matrices per node.)  Unsharded solvers only.
int smk_solver_set_factors_uniform(smk_solver* s, uint64_t seed_w, uint64_t seed_h)
{
    if (!s) return SMK_BAD_PARAM;
    if (is_dist(s) || s->comm) { set_error("set_factors_uniform: not for sharded solvers"); return SMK_UNSUPPORTED; }
    if (s->nsplit == NSPLIT_F16X2 && (s->a->ascale == 0.f || s->a->ascale != s->pg1[0].ascale)) {
        if (s->a->ascale == 0.f) { const int rc0 = matrix_measure_scale(s->a, s->st); if (rc0) return rc0; }
        for (int g = 0; g < s->ng; ++g) s->pg1[g].ascale = s->pg2[g].ascale = s->a->ascale;
        s->pl1.ascale = s->pl2.ascale = s->a->ascale;
    }
    if (s->pack_in_solve && (s->a->colnorm_max < 0.0 || s->a->rownorm_max < 0.0)) {      // ... and so do the norms behind the packing NNLS launch
        const int rc0 = matrix_measure_norms(s->a, s->st);
        if (rc0) return rc0;
    }
    int rc = launch_fill_factor_uniform(s->Wt, s->k, s->m, seed_w, 1, s->st);
    if (!rc) rc = launch_fill_factor_uniform(s->H, s->k, s->n, seed_h, 0, s->st);
    if (rc) return rc;
    const int big = INT_MAX;
    SMK_HIP(hipMemcpyAsync(s->fail_flag, &big, sizeof(int), hipMemcpyHostToDevice, s->st));
    if (s->W0c) {
        SMK_HIP(hipMemcpyAsync(s->W0c, s->Wt, (size_t)s->KP * s->m * sizeof(double), hipMemcpyDeviceToDevice, s->st));
        SMK_HIP(hipMemcpyAsync(s->H0c, s->H, (size_t)s->KP * s->n * sizeof(double), hipMemcpyDeviceToDevice, s->st));
    }
    SMK_HIP(hipStreamSynchronize(s->st));
    s->w_full = true;
    s->wc_valid = false;
    s->have_factors = true;
    s->inited = false;
    s->normalized = false;
    s->iter = 0;
    s->pg0 = 1.0;
    s->last_metric = 1.0;
    return SMK_OK;
}

// ---- collectives -------------------------------------------------------------------------------
// callback hook: one host call per buffer, in stream order on the main stream
static int dist_allreduce_cb(smk_solver* s, void* ptr, i64 count, int f64)
{
    if (s->ar && s->ar(s->ar_user, ptr, (int64_t)count, f64)) { set_error("all-reduce callback failed"); return SMK_DEVICE_ERROR; }
    return 0;
}

// native communicator: st2 picks up after everything enqueued on the main stream so far
static int comm_fork(smk_solver* s, hipEvent_t ev)
{
    SMK_HIP(hipEventRecord(ev, s->st));
    SMK_HIP(hipStreamWaitEvent(s->st2, ev, 0));
    return 0;
}
// The main stream waits for `n` events of the collective stream.  On a timed pass the wait is bracketed by two events on the
// main stream: their distance is the time the main stream stood still for the exchange -- measured exposure instead of the
// "step time minus products" subtraction (slot 3 of smk_solver_kernel_time; the two records themselves cost a few us of idle
// time, so an exchange that is hidden completely still reads ~5 us per wait).
static int main_waits_for_comm(smk_solver* s, const hipEvent_t* evs, int n)
{
    const bool timed = s->timing && (s->pass_timed[0] || s->pass_timed[1]);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed && (s->cal_counter++ & 3u) == 0 && s->st2) {        // every fourth bracket is preceded by a calibration bracket
        if (!s->ev_cal) {
            SMK_HIP(hipEventCreateWithFlags(&s->ev_cal, hipEventDisableTiming));
            SMK_HIP(hipEventRecord(s->ev_cal, s->st2));
        } else {
            hipEvent_t c0 = nullptr, c1 = nullptr;
            SMK_HIP(hipEventCreate(&c0));
            if (hipEventCreate(&c1) != hipSuccess) { (void)hipEventDestroy(c0); set_error("hipEventCreate failed"); return SMK_DEVICE_ERROR; }
            (void)hipEventRecord(c0, s->st);
            (void)hipStreamWaitEvent(s->st, s->ev_cal, 0);
            (void)hipEventRecord(c1, s->st);
            s->ev[4].push_back({c0, c1, 1});
        }
    }
    if (timed) {
        SMK_HIP(hipEventCreate(&e0));
        if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); set_error("hipEventCreate failed"); return SMK_DEVICE_ERROR; }
        (void)hipEventRecord(e0, s->st);
    }
    for (int i = 0; i < n; ++i)
        if (hipStreamWaitEvent(s->st, evs[i], 0) != hipSuccess) {
            if (timed) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
            set_error("hipStreamWaitEvent failed");
            return SMK_DEVICE_ERROR;
        }
    if (timed) {
        (void)hipEventRecord(e1, s->st);
        s->ev[3].push_back({e0, e1, 1});
    }
    return 0;
}
// ... and the main stream waits for what st2 has been given so far
static int comm_join(smk_solver* s, hipEvent_t ev)
{
    SMK_HIP(hipEventRecord(ev, s->st2));
    return main_waits_for_comm(s, &ev, 1);
}

// a collective on st2 bracketed by events when timing is on (slot 2 of smk_solver_kernel_time)
static int timed_collective(smk_solver* s, int pass, const std::function<int()>& issue)
{
    if (!s->timing || !s->pass_timed[pass]) return issue();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    SMK_HIP(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); set_error("hipEventCreate failed"); return SMK_DEVICE_ERROR; }
    (void)hipEventRecord(e0, s->st2);
    const int rc = issue();
    if (rc) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
    (void)hipEventRecord(e1, s->st2);
    s->ev[2].push_back({e0, e1, 1});
    return 0;
}

// a small in-place sum that the main stream needs right away (scalars of the stopping rule, W'W of a row-sharded W)
static int dist_allreduce_now(smk_solver* s, void* ptr, i64 count, int f64)
{
    if (!s->comm) return dist_allreduce_cb(s, ptr, count, f64);
    int rc = comm_fork(s, s->ev_x);
    if (rc) return rc;
    rc = comm_allreduce(s->comm, ptr, count, f64, s->st2);
    if (rc) return rc;
    return comm_join(s, s->ev_y);
}

// HH' is needed only after the H*At pass: with a native communicator its all-reduce runs on the second stream
// beside that pass; wait_gh() joins it back.
static int allreduce_gh(smk_solver* s)
{
    if (!is_dist(s)) return 0;
    if (!s->comm) return dist_allreduce_cb(s, s->Gh, (i64)s->KP * s->KP, 1);
    int rc = comm_fork(s, s->ev_gram);
    if (rc) return rc;
    rc = comm_allreduce(s->comm, s->Gh, (i64)s->KP * s->KP, 1, s->st2);
    if (rc) return rc;
    SMK_HIP(hipEventRecord(s->ev_gh, s->st2));
    s->gh_pending = true;
    return 0;
}
static int wait_gh(smk_solver* s)
{
    if (!s->gh_pending) return 0;
    const int rc = main_waits_for_comm(s, &s->ev_gh, 1);
    if (rc) return rc;
    s->gh_pending = false;
    return 0;
}
// the main stream joins the chunk exchanges of the last H*At pass (every consumer of the summed (AH')' calls this)
static int wait_r2(smk_solver* s)
{
    if (!s->r2_pending) return 0;
    // (the chunk exchanges were recorded on ONE in-order stream: the last event covers the others -- one barrier packet on the
    // main stream instead of nchunk; tools/mb/mb_event_hop.hip prices them)
    const int rc = main_waits_for_comm(s, &s->ev_r[s->nchunk - 1], 1);
    if (rc) return rc;
    s->r2_pending = false;
    return 0;
}

// BPP, k > 32: the inverse of a Gram matrix (side 0: W'W for the H solve, side 1: HH' for the W solve) is taken on a
// side stream as soon as the matrix exists; the streaming product that follows on the main stream hides it.
static inline double* inv_scratch(smk_solver* s, int side)
{
    return nnls_uses_tiles(s->k) ? s->nnls_scratch : s->nnls_scratch + (size_t)side * nnls_scratch_elems(s->k);
}
// `after`: the event that makes G final when that is not the main stream's current position (the HH' all-reduce of a
// sharded run finishes on the second stream)
static int start_inverse(smk_solver* s, int side, const double* G, hipEvent_t after = nullptr)
{
    s->inv_done[side] = false;
    s->inv_ride[side] = false;
    if (s->o.algorithm != SMK_ALG_BPP) return 0;
    {
        // The inverse is formed by one more workgroup of the product launch that follows this Gram matrix in every BPP schedule
        // (spmm_seg.hip / kernels.hip / bigprod.hip: InvRide) where that launch can carry it: a second stream pays two event hops
        // per solve, ~8 us each way on the main stream -- more than the 13 us inversion of k <= 32 they hide (the Reuters shape: 125
        // us per iteration beside the product, 118 in stream order, 91 riding; dense 4096 x 2048, k = 32: 115 / 92 / 70).  A launch
        // that cannot carry it leaves it to launch_nnls_bpp, in stream order, or to the second stream.  SMK_INV_RIDE=0: never rides.
        static const bool ride = [] { const char* e = getenv("SMK_INV_RIDE"); return !(e && e[0] == '0'); }();
        const bool route = !nnls_uses_tiles(s->k) && (s->KP == 64 || (s->KP == 32 && nnls_inverse_at_32()));
        const bool carrier = s->a->sparse ? true : (s->ng == 1 && bigprod_supports_ride(side == 0 ? s->pg1[0] : s->pg2[0]));
        if (ride && route && carrier && !after && !is_dist(s) && !s->comm && !s->w_sharded) { s->inv_ride[side] = true; return 0; }
    }
    if (!s->st_inv) return 0;
    if (after) {
        SMK_HIP(hipStreamWaitEvent(s->st_inv, after, 0));
    } else {
        SMK_HIP(hipEventRecord(s->ev_g[side], s->st));
        SMK_HIP(hipStreamWaitEvent(s->st_inv, s->ev_g[side], 0));
    }
    int rc = launch_gram_inverse(G, s->k, inv_scratch(s, side), s->st_inv);
    if (rc) return rc;
    SMK_HIP(hipEventRecord(s->ev_inv[side], s->st_inv));
    s->inv_pending[side] = true;
    return 0;
}
// The checked iteration loop on latency-bound BPP problems (C2: 73 us per iteration, 16 us per check): the stopping rule's gradient
// of iteration i is gradH_i = W_i'W_i H_i - W_i'A (gradW is the W-side NNLS's dual: projected-gradient sum exactly 0) -- and those
// three operands are precisely the system matrix, the warm start and the right-hand side of iteration i + 1's H-side NNLS launch.
// So that launch forms the sum on its way in (NnlsRiders::pg_part), both NNLS launches of an iteration store their result a second
// time into the snapshot slot (NnlsRiders::snap_x), and a check costs ONE small launch (the totals, written into the pinned slot)
// instead of a gradient pass over 16 slabs of W'A, a snapshot copy and a sum.  The driver already evaluates the rule one iteration
// late; when no further iteration follows, progress_end forms the check the old way.  SMK_PROGRESS_DEFER=0: off.
static bool guard_applies(const smk_solver* s);
static bool check_rides_in_nnls(const smk_solver* s)
{
    static const bool on = [] { const char* e = getenv("SMK_PROGRESS_DEFER"); return !(e && e[0] == '0'); }();
    static const bool fused = [] { const char* e = getenv("SMK_PROGRESS_FUSED"); return !(e && e[0] == '0'); }();
    static const bool dual = [] { const char* e = getenv("SMK_BPP_GRADW"); return !(e && e[0] == '1'); }();
    static const bool guard_env = [] { const char* e = getenv("SMK_GUARD_EVERY"); return e && atoi(e) > 0; }();
    if (!on || !fused || !dual || guard_env) return false;
    if (s->o.algorithm != SMK_ALG_BPP || s->KP > 16 || s->o.prog_est_algorithm != SMK_PROG_PG_RATIO) return false;
    if (is_dist(s) || s->comm || s->w_sharded) return false;
    const i64 gpb = 256 / s->KP;
    return (s->n + gpb - 1) / gpb <= (i64)s->pg_half;          // one partial per workgroup of the H-side launch
}

static int nnls_side(smk_solver* s, int side, double* X, i64 c0, i64 c1, PartialView R, const double* G)
{
    if (s->inv_pending[side]) {
        SMK_HIP(hipStreamWaitEvent(s->st, s->ev_inv[side], 0));
        s->inv_pending[side] = false;
        s->inv_done[side] = true;
    }
    // k in (8, 16], fp16 form, one GPU: the launch that solves ALL columns of a factor also leaves its Gram partials (the
    // gram_x that follows in every BPP schedule then only reduces them)
    static const bool fuse = [] { const char* e = getenv("SMK_NNLS_GRAM"); return !(e && e[0] == '0'); }();
    const i64 N = side == 0 ? s->n : s->m;
    const bool want = fuse && s->KP == 16 && s->nsplit == NSPLIT_F16X2 && !is_dist(s) && c0 == 0 && c1 == N && (X == s->H || X == s->Wt);
    s->nnls_gram_nblk[side] = 0;
    // ... and packs it, when the system matrix is the Gram matrix of a factor that an NNLS launch of this run produced (>= 0:
    // the bound behind the row scales needs that; the first solve of a run works from the caller's factor and packs the old way)
    const int fx = side == 0 ? 1 : 0;                // the factor being solved (0 = W, 1 = H); the other one is `side`
    NnlsPack pk;
    const bool pack = want && s->pack_in_solve && !s->pack_in_solve_off && !s->comm && s->from_nnls[side] && s->a->colnorm_max >= 0.0 && s->a->rownorm_max >= 0.0;
    if (pack) {
        pk.out = (unsigned char*)(fx == 0 ? s->packW : s->packH);
        pk.xscale = s->xscale[fx];
        pk.oscale = s->oscale[fx];
        pk.anorm = side == 0 ? s->a->colnorm_max : s->a->rownorm_max;     // H's columns solve for columns of A, W's rows for rows
        pk.ascale = (double)s->a->ascale;
        pk.nq = packed_chunk_pairs_f16x2(s->a->storage, N);
        // TEST HOOK: shrink the bound so that the scaled entries leave fp16's range (the launch must flag it, the run must be
        // repeated without the packing: pack_fail_soft)
        if (const char* e = getenv("SMK_NNLS_PACK_TEST_ANORM")) pk.anorm *= atof(e);
    }
    s->nnls_packed[fx] = false;
    // slot 5 of smk_solver_kernel_time: the block-pivoting launches (both kernels of a k > 16 solve), sampled with the pass that fed them
    hipEvent_t te0 = nullptr, te1 = nullptr;
    if (s->timing && s->pass_timed[side] && c1 > c0) {
        if (hipEventCreate(&te0) == hipSuccess && hipEventCreate(&te1) == hipSuccess) (void)hipEventRecord(te0, s->st);
        else { if (te0) (void)hipEventDestroy(te0); te0 = te1 = nullptr; }
    }
    struct Stamp { smk_solver* s; hipEvent_t a, b; ~Stamp() { if (a) { (void)hipEventRecord(b, s->st); s->ev[5].push_back({a, b, 1}); } } } stamp{s, te0, te1};
    // riders of the checked loop (check_rides_in_nnls): the previous iteration's projected-gradient sum, this iteration's snapshot
    NnlsRiders rd;
    bool riders = false;
    const int k2 = (s->k + 1) / 2 * 2;
    if (c0 == 0 && c1 == N && (X == s->H || X == s->Wt) && (s->pg_defer_slot >= 0 || s->iter_snap_slot >= 0) && check_rides_in_nnls(s)) {
        if (side == 0 && s->pg_defer_slot >= 0) { rd.pg_part = s->pg_partials + s->pg_half; rd.pg_nblk = &s->pg_defer_nblk; riders = true; }
        if (s->iter_snap_slot >= 0) {
            const int b = s->iter_snap_slot;
            if (!s->snap[b]) { const int arc = dev_alloc(&s->snap[b], snapshot_elems(s->k, s->m, s->n)); if (arc) return arc; }
            rd.snap_x = s->snap[b] + (side == 0 ? (size_t)s->m * k2 : 0);        // snapshot_kernel's layout: [W'][H][W'W]
            rd.k2 = k2;
            riders = true;
        }
    }
    const int rc = launch_nnls_bpp(X, nullptr, s->k, c0, c1, R, G, s->fail_flag, s->iter, inv_scratch(s, side), s->inv_done[side] ? 1 : 0, g_cus, s->st,
                                   want ? s->gram_scratch : nullptr, want ? &s->nnls_gram_nblk[side] : nullptr, pack ? &pk : nullptr, s->nnls_defer,
                                   riders ? &rd : nullptr);
    if (!rc && rd.pg_part) {
        // the totals of the deferred check are due: they ride in the tail of the pass that follows (prod2), or go out as a launch
        // of their own right in front of it (check_totals)
        if (s->pg_defer_nblk <= 0) { set_error("the deferred progress check was not carried by the NNLS launch"); return SMK_FAILURE; }
        s->pg_totals_slot = s->pg_defer_slot;
        s->pg_defer_slot = -1;
    }
    if (!rc && (X == s->H || X == s->Wt)) {
        s->from_nnls[fx] = c0 == 0 && c1 == N;
        s->nnls_packed[fx] = pack && s->nnls_gram_nblk[side] > 0;
    }
    // without the side stream (k <= 32) a first launch at k > 32 computes the inverse itself, in stream order
    if (!rc && c1 > c0 && (s->KP >= 64 || (s->KP == 32 && nnls_inverse_at_32()))) s->inv_done[side] = true;
    return rc;
}

// ---- building blocks -----------------------------------------------------------------------
// every pass starts here: is it one of the timed samples?
static inline void begin_pass(smk_solver* s, int which)
{
    if (!s->timing) { s->pass_timed[which] = false; return; }
    const unsigned c = s->pass_counter[which]++;
    s->pass_timed[which] = s->timing_stride <= 1 || (c % (unsigned)s->timing_stride) == 0;
    if (s->pass_timed[which]) ++s->pass_sampled[which];
}

// one launch of the streaming product, bracketed by events when timing is on; `counts`: this launch completes a pass
static int timed_bigprod(smk_solver* s, int which, const BigProdPlan& pl_in, const void* B, i64 ldb, const void* Xp,
                         double* P, int counts = 1)
{
    BigProdPlan pl = pl_in;
    if (s->inv_ride[which] && bigprod_supports_ride(pl)) {          // start_inverse: this side's Gram inverse rides in the launch
        pl.inv_ride.G = which == 0 ? s->Gw : s->Gh;
        pl.inv_ride.k = s->k;
        pl.inv_ride.Ginv = inv_scratch(s, which);
    }
    s->inv_ride[which] = false;
    if (s->timing && s->pass_timed[which]) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        SMK_HIP(hipEventCreate(&e0));
        if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); set_error("hipEventCreate failed"); return SMK_DEVICE_ERROR; }
        s->ev[which].push_back({e0, e1, counts});   // owned by the solver from here on (destroyed with it)
        SMK_HIP(hipEventRecord(e0, s->st));
        int rc = launch_bigprod(pl, B, ldb, Xp, P, s->st);
        if (rc == 1) { s->inv_done[which] = true; rc = 0; }        // the launch carried the inverse
        if (rc) { s->ev[which].pop_back(); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
        SMK_HIP(hipEventRecord(e1, s->st));
        return 0;
    }
    int rc = launch_bigprod(pl, B, ldb, Xp, P, s->st);
    if (rc == 1) { s->inv_done[which] = true; rc = 0; }
    return rc;
}

// R1 = W'A  (k x n, local columns)
static int timed_spmm(smk_solver* s, int which, const i64* colptr, const unsigned* rowidx, const double* val, i64 ncols,
                      const double* X, int ldx, double* P)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = s->timing && s->pass_timed[which];
    if (timed) {
        SMK_HIP(hipEventCreate(&e0));
        if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); set_error("hipEventCreate failed"); return SMK_DEVICE_ERROR; }
        (void)hipEventRecord(e0, s->st);
    }
    int rc;
    const BlockedCsc& blk = (which == 0) ? s->a->bA : s->a->bAt;
    const SegPlan& seg = (which == 0) ? s->a->segA : s->a->segAt;
    InvRide ride;                                                   // start_inverse: this side's Gram inverse rides in the launch
    if (s->inv_ride[which]) { ride.G = which == 0 ? s->Gw : s->Gh; ride.k = s->k; ride.Ginv = inv_scratch(s, which); }
    s->inv_ride[which] = false;
    GramRide gram;                                                  // gram_factor: this factor's Gram matrix is due
    const bool seg_route = ldx == s->KP && s->k > 2 && !is_wide(s->k) && seg.ncols == ncols && seg.rowflag && !seg.uniform;
    if (s->gram_ride[which]) {
        s->gram_ride[which] = false;
        const double* F = which == 0 ? s->Wt : s->H;
        const i64 FN = which == 0 ? s->m : s->n;
        double* G = which == 0 ? s->Gw : s->Gh;
        if (seg_route) { gram.X = F; gram.N = FN; gram.max_blocks = GRAM_BLOCKS; gram.Gp = s->gram_scratch; gram.G = G; }
        else { const int grc = launch_gram(F, s->k, FN, G, s->gram_scratch, GRAM_BLOCKS, s->st); if (grc) return grc; }
    }
    if (ldx == 2 && blk.nb > 1) rc = launch_spmm_blocked2(blk, X, P, which == 0 ? s->pl1.ncols_pad : s->pl2.ncols_pad, s->st);
    else if (seg_route) {
        // the partial sums of long columns live in the SOLVER (two solvers on one sparse matrix run on their own streams)
        if (seg.npieces > 0 && !s->seg_pieces[which] && smk::dev_malloc((void**)&s->seg_pieces[which], (size_t)seg.npieces * 128 * sizeof(double)) != hipSuccess) {
            s->seg_pieces[which] = nullptr;
            set_error("no memory for the long-column partial sums");
            return SMK_DEVICE_ERROR;
        }
        rc = launch_spmm_seg(seg, colptr, val, X, s->k, P, s->kpp, s->st, s->seg_pieces[which], &ride, &gram);
        if (rc > 0 && (rc & 2)) rc &= ~2;                           // ... and the Gram matrix
        else if (rc >= 0 && gram.X) {                               // nothing was launched (a matrix without stored entries): the matrix by itself
            const int grc = launch_gram(gram.X, s->k, gram.N, gram.G, s->gram_scratch, GRAM_BLOCKS, s->st);
            if (grc) return grc;
        }
    }
    else rc = launch_spmm_gather(colptr, rowidx, val, ncols, s->a->nnz, X, ldx, s->k, P, s->kpp, s->st, &ride);
    if (rc == 1) { s->inv_done[which] = true; rc = 0; }             // the launch carried the inverse
    if (timed) {
        if (rc) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
        (void)hipEventRecord(e1, s->st);
        s->ev[which].push_back({e0, e1, 1});
    }
    return rc;
}

// W'A with a row-sharded W (BPP, native communicator): every rank packs its own blocks, the blocks of chunk j are
// all-gathered on st2 (packed operand: 4 B per entry in the fp16 form, half of the fp64 values) and the product is
// taken chunk by chunk down the contraction, each launch adding to P1, as soon as its chunk of the operand has landed.
static int prod1_sharded(smk_solver* s)
{
    begin_pass(s, 0);
    const int storage = s->a->storage;
    const size_t es = (size_t)elem_size(storage);
    const bool f64 = s->nsplit == NSPLIT_F64;       // the accurate form reads the fp64 factor: gather the fp64 own blocks into W itself
    int rc = 0;
    // one launch per group of 64 factor rows: this rank's blocks into their places in the operand (padding rows as zeros)
    for (int g = 0; g < s->ng && !rc && !f64; ++g)
        rc = launch_pack_own_blocks(s->Wown, s->KP, s->pg1[g].k0, s->pg1[g].kg, s->n_own, s->blk, s->nchunk, s->world, s->rank, storage,
                                    s->nsplit, (unsigned char*)s->packW + s->pg1[g].pack_offset, s->st, s->xscale[0]);
    if (rc) return rc;
    rc = comm_fork(s, s->ev_x);
    if (rc) return rc;
    for (int j = 0; j < s->nchunk; ++j) {
        i64 r0, r1;
        chunk_rows(s, j, &r0, &r1);
        rc = timed_collective(s, 0, [&] {
            if (f64)        // block r of chunk j from every rank, straight into its rows of the full fp64 W
                return comm_allgather_to(s->comm, s->Wown + (i64)j * s->blk * s->KP, s->Wt + r0 * s->KP, s->blk * s->KP, 1, s->st2);
            for (int g = 0; g < s->ng; ++g) {
                const size_t per = packed_row_offset(storage, s->pg1[g].kg, s->nsplit, s->blk);          // bytes per block
                unsigned char* base = (unsigned char*)s->packW + s->pg1[g].pack_offset + packed_row_offset(storage, s->pg1[g].kg, s->nsplit, r0);
                const int grc = comm_allgather(s->comm, base, (i64)(per / 4), 0, s->st2);
                if (grc) return grc;
            }
            return 0;
        });
        if (rc) return rc;
        SMK_HIP(hipEventRecord(s->ev_a[j], s->st2));
    }
    bool first = true;
    for (int j = 0; j < s->nchunk; ++j) {
        i64 r0, r1;
        chunk_rows(s, j, &r0, &r1);
        { const int wrc = main_waits_for_comm(s, &s->ev_a[j], 1); if (wrc) return wrc; }
        const i64 rows = std::min<i64>(r1, s->m) - r0;
        if (rows <= 0) continue;
        const bool last = (j == s->nchunk - 1) || (std::min<i64>(r1, s->m) >= s->m);
        for (int g = 0; g < s->ng; ++g) {
            BigProdPlan pl = s->pg1[g];
            pl.stages = (rows + pl.mb - 1) / pl.mb;
            pl.nst = (pl.stages + pl.S - 1) / pl.S;
            pl.accum = first ? 0 : 1;
            const void* Xp = (const unsigned char*)s->packW + pl.pack_offset + packed_row_offset(storage, pl.kg, s->nsplit, r0);
            if (f64) { pl.len = rows; Xp = s->Wt + r0 * s->KP + pl.k0; }     // rows [r0, r0 + rows) of the gathered W, this group's factor rows
            rc = timed_bigprod(s, 0, pl, (const unsigned char*)s->a->A + (size_t)r0 * es, s->a->ldA, Xp, s->P1 + pl.k0, last ? 1 : 0);
            if (rc) return rc;
        }
        first = false;
    }
    if (f64) s->w_full = true;       // every row of the fp64 W is current again (a by-product of the accurate form's gather)
    return 0;
}

// the product of factor `side` (0 = W, 1 = H) takes the pending reduction of that factor's Gram partials along (gram_factor)
static inline void take_tail(smk_solver* s, int side, BigProdPlan* pl)
{
    if (s->tail_nblk[side] <= 0) return;
    pl->tail_gp = s->gram_scratch;
    pl->tail_nblk = s->tail_nblk[side];
    pl->tail_g = side == 0 ? s->Gw : s->Gh;
    s->tail_nblk[side] = 0;
}

// The totals of a deferred progress check (nnls_side left the per-workgroup sums): W'W of the checked iteration is still in place
// until the W-side solve of this iteration has run, so they may be formed anywhere in front of it.  pl != nullptr and the pass
// carries a tail already: one more tail workgroup forms them (the caller records the slot's event behind the pass: returns 1);
// otherwise a launch of their own, now.  SMK_PROGRESS_TAIL=0: always the launch.
// The result of a check is awaited by polling its pinned slot (the kernel that writes the totals stores a tag behind them, system
// scope) instead of an event: an event record between two launches of a 70 us iteration costs 3 - 4 us of idle stream (C2 checked:
// 12 650 -> 13 350 it/s).  SMK_PROGRESS_POLL=0: the event.
static bool progress_polls()
{
    static const bool poll = [] { const char* e = getenv("SMK_PROGRESS_POLL"); return !(e && e[0] == '0'); }();
    return poll;
}

static int check_totals(smk_solver* s, BigProdPlan* pl)
{
    if (s->pg_totals_slot < 0) return 0;
    static const bool ride = [] { const char* e = getenv("SMK_PROGRESS_TAIL"); return !(e && e[0] == '0'); }();
    const bool poll = progress_polls();
    const int b = s->pg_totals_slot, k2 = (s->k + 1) / 2 * 2;
    double* snap_g = s->pg_defer_snap ? s->snap[b] + (size_t)(s->m + s->n) * k2 : nullptr;
    const double* part = s->pg_partials + s->pg_half;
    const double tag = poll ? (double)(s->pg_defer_tag + 2) : 0.0;
    s->poll_tag[b] = tag;
    if (ride && pl && pl->tail_nblk > 0) {
        BigProdPlan::TailCheck& tc = pl->tail_check;
        tc.part = part;  tc.n = s->pg_defer_nblk;  tc.flag_slot = 5;  tc.tag_limit = s->pg_defer_tag;  tc.kk = s->KP * s->KP;
        tc.out = s->scal;  tc.host_out = s->pin[b].h;  tc.flag = s->fail_flag;  tc.G = s->Gw;  tc.snap_g = snap_g;  tc.tag = tag;
        ++s->check_routes[3];
        return 1;
    }
    s->pg_totals_slot = -1;
    ++s->check_routes[2];
    const int rc = launch_pg_defer_sum(part, s->pg_defer_nblk, s->scal, s->pin[b].h, s->fail_flag, 5, s->pg_defer_tag, s->Gw, snap_g,
                                       s->KP * s->KP, s->st, tag);
    if (rc) return rc;
    if (tag == 0.0) SMK_HIP(hipEventRecord(s->pev[b], s->st));
    return 0;
}

static int prod1(smk_solver* s)
{
    if (s->w_sharded) return prod1_sharded(s);
    begin_pass(s, 0);
    if (s->a->sparse) {
        if (s->Wc) {        // RANK2: gather from the compact copy of W (16 B per row)
            if (!s->wc_valid) { const int crc = launch_rank2_compact(s->Wt, s->Wc, s->m, s->st); if (crc) return crc; s->wc_valid = true; }
            return timed_spmm(s, 0, s->a->colptr, s->a->rowidx, s->a->val, s->n, s->Wc, 2, s->P1);
        }
        return timed_spmm(s, 0, s->a->colptr, s->a->rowidx, s->a->val, s->n, s->Wt, s->KP, s->P1);
    }
    int rc = 0;
    const bool f64 = s->nsplit == NSPLIT_F64;                 // the accurate form reads the fp64 factor itself
    if (!s->packed_fresh[0] && !f64) rc = launch_pack(s->Wt, s->k, s->m, s->a->storage, s->nsplit, s->packW, s->st, s->xscale[0]);
    s->packed_fresh[0] = false;
    if (rc) return rc;
    for (int g = 0; g < s->ng; ++g) {
        const void* Xp = f64 ? (const void*)(s->Wt + s->pg1[g].k0) : (const void*)((const unsigned char*)s->packW + s->pg1[g].pack_offset);
        BigProdPlan pl = s->pg1[g];
        take_tail(s, 0, &pl);
        rc = timed_bigprod(s, 0, pl, s->a->A, s->a->ldA, Xp, s->P1 + s->pg1[g].k0);
        if (rc) return rc;
    }
    return 0;
}

// R2 = H At = (A H')'  (k x m), summed over ranks when sharded.  With a native communicator the pass runs chunk by
// chunk over the rows of A and the sum of chunk j (reduce-scatter for the row-sharded BPP solve, all-reduce otherwise)
// travels on st2 while the product streams chunk j + 1; consumers call wait_r2().
static int prod2(smk_solver* s)
{
    begin_pass(s, 1);
    int rc = 0;
    const PartialView pv{s->P2, s->pl2.S, (i64)s->pl2.ncols_pad * s->kpp, s->kpp, 1};
    if (s->pg_totals_slot >= 0 && (s->a->sparse || s->comm || s->ng != 1 || s->tail_nblk[1] <= 0)) {
        rc = check_totals(s, nullptr);                  // no tail to ride in
        if (rc) return rc;
    }
    if (s->a->sparse) {
        rc = s->Hc ? timed_spmm(s, 1, s->a->colptr_t, s->a->rowidx_t, s->a->val_t, s->m, s->Hc, 2, s->P2)
                   : timed_spmm(s, 1, s->a->colptr_t, s->a->rowidx_t, s->a->val_t, s->m, s->H, s->KP, s->P2);
        if (rc) return rc;
        rc = wait_gh(s);
        if (rc || !is_dist(s)) return rc;
        rc = launch_reduce_partials(pv, s->k, 0, s->pl2.ncols_pad, s->R2red, s->red_f64 ? 1 : 0, s->st);
        if (rc) return rc;
        if (!s->comm) return dist_allreduce_cb(s, s->R2red, (i64)s->pl2.ncols_pad * s->kpp, 0);
        rc = comm_fork(s, s->ev_c[0]);
        if (rc) return rc;
        rc = timed_collective(s, 1, [&] { return comm_allreduce(s->comm, s->R2red, (i64)s->pl2.ncols_pad * s->kpp, s->red_f64 ? 1 : 0, s->st2); });
        if (rc) return rc;
        return comm_join(s, s->ev_r[0]);
    }
    const bool f64 = s->nsplit == NSPLIT_F64;
    if (!s->packed_fresh[1] && !f64) rc = launch_pack(s->H, s->k, s->n, s->a->storage, s->nsplit, s->packH, s->st, s->xscale[1]);
    s->packed_fresh[1] = false;
    if (rc) return rc;
    auto xh = [&](const BigProdPlan& pl) -> const void* {
        return f64 ? (const void*)(s->H + pl.k0) : (const void*)((const unsigned char*)s->packH + pl.pack_offset);
    };
    if (!s->comm) {
        for (int g = 0; g < s->ng && !rc; ++g) {
            BigProdPlan pl = s->pg2[g];
            take_tail(s, 1, &pl);
            const int ride = check_totals(s, &pl);
            if (ride < 0 || ride > 1) return ride;
            rc = pl.tr ? timed_bigprod(s, 1, pl, s->a->A, s->a->ldA, xh(s->pg2[g]), s->P2 + s->pg2[g].k0)
                       : timed_bigprod(s, 1, pl, s->a->At, s->a->ldAt, xh(s->pg2[g]), s->P2 + s->pg2[g].k0);
            if (ride == 1 && !rc) {
                if (s->poll_tag[s->pg_totals_slot] == 0.0) SMK_HIP(hipEventRecord(s->pev[s->pg_totals_slot], s->st));
                s->pg_totals_slot = -1;
            }
        }
        if (rc) return rc;
        rc = wait_gh(s);
        if (rc || !is_dist(s)) return rc;
        rc = launch_reduce_partials(pv, s->k, 0, s->pl2.ncols_pad, s->R2red, 0, s->st);
        if (rc) return rc;
        return dist_allreduce_cb(s, s->R2red, (i64)s->pl2.ncols_pad * s->kpp, 0);
    }
    const size_t es = (size_t)elem_size(s->a->storage), rb = s->red_f64 ? sizeof(double) : sizeof(float);
    for (int j = 0; j < s->nchunk; ++j) {
        i64 r0, r1;
        chunk_rows(s, j, &r0, &r1);
        for (int g = 0; g < s->ng; ++g) {
            BigProdPlan pl = s->pg2[g];
            pl.tiles = (r1 - r0 + pl.nb - 1) / pl.nb;
            if (pl.tr && r0 + pl.tiles * pl.nb > s->a->ldA) { set_error("transposed-source chunk runs past the padded rows of A"); return SMK_FAILURE; }
            // rows [r0, r1) of A: columns of the stored transpose, or -- single copy -- a row offset into A itself
            const unsigned char* Bsrc = pl.tr ? (const unsigned char*)s->a->A + (size_t)r0 * es
                                              : (const unsigned char*)s->a->At + (size_t)r0 * s->a->ldAt * es;
            rc = timed_bigprod(s, 1, pl, Bsrc, pl.tr ? s->a->ldA : s->a->ldAt,
                               xh(pl), s->P2 + pl.k0 + r0 * pl.pstride, j == s->nchunk - 1 ? 1 : 0);
            if (rc) return rc;
        }
        rc = comm_fork(s, s->ev_c[j]);
        if (rc) return rc;
        // off the main stream: the row splits of the chunk summed (and rounded to the wire type), then its exchange
        if (!s->r2_alias) { rc = launch_reduce_partials(pv, s->k, r0, r1 - r0, s->R2red, s->red_f64 ? 1 : 0, s->st2); if (rc) return rc; }
        unsigned char* base = (unsigned char*)s->R2red + (size_t)r0 * s->kpp * rb;
        rc = timed_collective(s, 1, [&] {
            if (s->w_sharded)       // every rank receives the sum of ITS block, next to its other blocks
                return comm_reduce_scatter_to(s->comm, base, (unsigned char*)s->R2own + (size_t)j * s->blk * s->kpp * rb, s->blk * s->kpp,
                                              s->red_f64 ? 1 : 0, s->st2);
            return comm_allreduce(s->comm, base, (r1 - r0) * s->kpp, s->red_f64 ? 1 : 0, s->st2);
        });
        if (rc) return rc;
        SMK_HIP(hipEventRecord(s->ev_r[j], s->st2));
    }
    s->r2_pending = true;
    return wait_gh(s);
}

// Gram matrix of a factor; for dense A the same launch also writes the packed operand of the product that follows
// (every schedule calls gram_x and prod_x back to back)
static int gram_factor(smk_solver* s, int side)
{
    const double* X = side == 0 ? s->Wt : s->H;
    const i64 N = side == 0 ? s->m : s->n;
    double* G = side == 0 ? s->Gw : s->Gh;
    s->packed_fresh[side] = false;
    s->gram_ride[side] = false;
    {
        // Sparse A: the gather product that follows in every schedule reads the same factor, and its two launches (segments, long-column
        // fix-up) carry the Gram matrix along -- partial sums as extra workgroups of the first, their reduction as extra workgroups of
        // the second (spmm_seg.hip, gram_body.h): four launches become two (the Reuters shape under HALS: 102 -> 84 us per iteration).
        // Not where block pivoting needs the inverse of this matrix in the SAME launch (k in (16, 32]: that one rides, start_inverse).
        // A product that takes another route forms the matrix first, as before (timed_spmm).  SMK_GRAM_RIDE=0: never rides.
        static const bool ride = [] { const char* e = getenv("SMK_GRAM_RIDE"); return !(e && e[0] == '0'); }();
        const bool inverse_next = s->o.algorithm == SMK_ALG_BPP && (s->KP == 64 || (s->KP == 32 && nnls_inverse_at_32()));
        if (ride && s->a->sparse && !is_dist(s) && !s->comm && !s->w_sharded && (s->KP == 16 || s->KP == 32) && s->k > 2 && !inverse_next &&
            s->o.algorithm != SMK_ALG_RANK2) {
            s->gram_ride[side] = true;
            return 0;
        }
    }
    if (side == 0 && s->w_sharded) {
        // W'W from this rank's own rows, summed over the ranks; the fp16 row scales follow the finished matrix
        int rc = s->n_own > 0 ? launch_gram(s->Wown, s->k, s->n_own, G, s->gram_scratch, GRAM_BLOCKS, s->st)
                              : launch_zero_f64(G, (i64)s->KP * s->KP, s->st);
        if (rc) return rc;
        rc = dist_allreduce_now(s, G, (i64)s->KP * s->KP, 1);
        if (rc) return rc;
        if (s->nsplit == NSPLIT_F16X2) return launch_gram_scales(G, s->k, s->xscale[0], s->oscale[0], (double)s->a->ascale, s->st);
        return 0;
    }
    if (s->nsplit == NSPLIT_F16X2) {     // the reduce launch also derives the row scales of the operand packed next
        const int ns = side == 0 ? 1 : 0;            // the NNLS launch that solved THIS factor (side 1 solves W, side 0 solves H)
        if (s->nnls_gram_nblk[ns] > 0) {
            const int nb = s->nnls_gram_nblk[ns];
            s->nnls_gram_nblk[ns] = 0;
            // the solve packed the operand itself: nothing is needed before the product, which takes the reduction along
            if (s->nnls_packed[side]) {
                s->nnls_packed[side] = false;
                s->packed_fresh[side] = true;
                s->tail_nblk[side] = nb;
                return 0;
            }
            // ... and packs it in the same launch (the packing workgroups add up the 16 diagonal entries themselves)
            static const bool fuse_pack = [] { const char* e = getenv("SMK_REDUCE_PACK"); return !(e && e[0] == '0'); }();
            if (fuse_pack && !s->a->sparse) {
                const int rc = launch_reduce_pack_f16x2(s->gram_scratch, nb, s->k, G, s->xscale[side], s->oscale[side], (double)s->a->ascale, X, N,
                                                        s->a->storage, side == 0 ? s->packW : s->packH, s->st);
                if (rc == 0) { s->packed_fresh[side] = true; return 0; }
                if (rc != 1) return rc;
            }
            return launch_gram_reduce(s->gram_scratch, nb, s->k, G, s->st, s->xscale[side], s->oscale[side], (double)s->a->ascale);
        }
        return launch_gram(X, s->k, N, G, s->gram_scratch, GRAM_BLOCKS, s->st, s->xscale[side], s->oscale[side], (double)s->a->ascale);
    }
    // (k > 16.  Leaving the partial sums to the pack launch -- every pack thread adding up its diagonal entry, workgroup 0
    // finishing the matrix -- costs 12 us where reduce + pack cost 9.5; what works, at k <= 16 above, is separate reducer
    // workgroups in the same launch and packers that read a compact copy of the diagonal partials.)
    if (!s->a->sparse && s->o.algorithm != SMK_ALG_RANK2 && s->nsplit != NSPLIT_F64) {
        const int rc = launch_gram_pack(X, s->k, N, G, s->gram_scratch, GRAM_BLOCKS, s->a->storage, s->nsplit,
                                        side == 0 ? s->packW : s->packH, s->st);
        if (rc == 0) { s->packed_fresh[side] = true; return 0; }
        if (rc != 1) return rc;
    }
    return launch_gram(X, s->k, N, G, s->gram_scratch, GRAM_BLOCKS, s->st);
}

static int gram_w(smk_solver* s)
{
    int rc = gram_factor(s, 0);
    if (rc) return rc;
    return start_inverse(s, 0, s->Gw);
}

static int gram_h(smk_solver* s)
{
    int rc = gram_factor(s, 1);
    if (rc) return rc;
    rc = allreduce_gh(s);
    if (rc) return rc;
    // sharded: HH' is final only after its all-reduce -- in stream order with the callback hook, on the second stream
    // (event ev_gh) with a native communicator; either way the inversion runs beside the H*At pass
    return start_inverse(s, 1, s->Gh, (is_dist(s) && s->gh_pending) ? s->ev_gh : nullptr);
}

// row-sharded W: bring every row of the fp64 W to every rank (results, DELTA_FNORM); one in-place all-gather per chunk
static int gather_w(smk_solver* s)
{
    if (!s->w_sharded || s->w_full) return 0;
    int rc = comm_fork(s, s->ev_x);
    if (rc) return rc;
    for (int j = 0; j < s->nchunk && !rc; ++j)
        rc = comm_allgather_to(s->comm, s->Wown + (i64)j * s->blk * s->KP, s->Wt + (i64)j * s->world * s->blk * s->KP, s->blk * s->KP, 1, s->st2);
    if (rc) return rc;
    s->w_full = true;
    return comm_join(s, s->ev_y);
}

// solver.Init (mu :98-114, hals :142-159, bpp :310-335) + progress_est->Init
static int solver_init(smk_solver* s)
{
    int rc = 0;
    // the factors are the caller's (or a restored snapshot): nothing is known to come from an NNLS launch, nothing is pending
    s->from_nnls[0] = s->from_nnls[1] = false;
    s->nnls_packed[0] = s->nnls_packed[1] = false;
    s->tail_nblk[0] = s->tail_nblk[1] = 0;
    s->pg_defer_slot = s->iter_snap_slot = s->pg_totals_slot = -1;
    if (s->o.algorithm == SMK_ALG_HALS) {
        rc = gram_h(s);  if (rc) return rc;
        rc = prod2(s);   if (rc) return rc;
    } else {   // MU, BPP, RANK2: WtA and WtW from W0
        rc = gram_w(s);  if (rc) return rc;
        rc = prod1(s);   if (rc) return rc;
    }
    if (s->o.prog_est_algorithm == SMK_PROG_DELTA_FNORM) {
        rc = gather_w(s); if (rc) return rc;
        SMK_HIP(hipMemcpyAsync(s->Wprev, s->Wt, (size_t)s->KP * s->m * sizeof(double), hipMemcpyDeviceToDevice, s->st));
    }
    s->inited = true;
    return 0;
}

// one solver iteration (the body of `solver(A,W,H,gradW,gradH)`); gradients are formed on
// demand by update_progress() from the same R1/R2/Gw/Gh the reference uses.
static int solver_iteration(smk_solver* s)
{
    int rc = 0;
    s->normalized = false;          // the factors are about to change
    const PartialView r1 = view1(s), r2 = view2(s);
    switch (s->o.algorithm) {
        case SMK_ALG_MU:   // nmf_solver_mu.hpp:121-164
            rc = launch_mu_update(s->H, s->k, s->n, r1, s->Gw, s->st, s->wide_tmp);  if (rc) return rc;
            rc = gram_h(s);   if (rc) return rc;
            rc = prod2(s);    if (rc) return rc;
            rc = wait_r2(s);  if (rc) return rc;
            if (s->w_sharded) {       // own blocks only (back to back in Wown, their rows of the summed (AH')' in R2own)
                if (s->n_own > 0) { rc = launch_mu_update(s->Wown, s->k, s->n_own, view_own(s), s->Gh, s->st, s->wide_tmp); if (rc) return rc; }
                s->w_full = false;
            } else {
                rc = launch_mu_update(s->Wt, s->k, s->m, r2, s->Gh, s->st, s->wide_tmp); if (rc) return rc;
            }
            rc = gram_w(s);   if (rc) return rc;
            rc = prod1(s);    if (rc) return rc;
            break;
        case SMK_ALG_HALS: { // nmf_solver_hals.hpp:166-199
            // round 6: at k <= 32 on one GPU both sweeps also leave the packed operand of the product that follows and their Gram
            // partials (kernels.hip: tile_pack_gram) -- the separate gram_pack launches (11 us each at C3) go, only the 5 us
            // reductions stay.  SMK_HALS_EPILOGUE=0: the launches of before.
            static const bool ep_on = [] { const char* e = getenv("SMK_HALS_EPILOGUE"); return !(e && e[0] == '0'); }();
            const bool bf16_frag = s->a->storage == SMK_STORE_BF16 || s->nsplit >= 2;
            const bool ep_ok = ep_on && !s->a->sparse && !is_dist(s) && !s->comm && (s->KP == 16 || s->KP == 32) && bf16_frag && s->nsplit >= 1 && s->nsplit <= 3 &&
                               s->ng == 1 && s->hals_ep_blocks > 0;
            HalsEpilogue epw, eph;
            if (ep_ok) {
                epw.pack_out = (unsigned char*)s->packW; epw.Gp = s->gram_scratch; epw.KT = kt_of(s->k); epw.nsplit = s->nsplit;
                epw.max_blocks = s->hals_ep_blocks; epw.nq = round_up(s->m, 128) / 16;
                eph = epw;
                eph.pack_out = (unsigned char*)s->packH; eph.nq = round_up(s->n, 128) / 16;
            }
            rc = wait_r2(s);  if (rc) return rc;
            rc = launch_hals_w_update(s->Wt, s->k, s->m, r2, s->Gh, s->hals_scratch, g_cus, s->fail_flag, s->hals_calls++, s->hals_multi ? 1 : 0, s->st,
                                      ep_ok ? &epw : nullptr); if (rc) return rc;
            if (epw.done) { rc = launch_gram_reduce(s->gram_scratch, epw.nblk, s->k, s->Gw, s->st, nullptr, nullptr, 1.0); s->packed_fresh[0] = true; }
            else rc = gram_w(s);
            if (rc) return rc;
            rc = prod1(s);    if (rc) return rc;
            rc = launch_hals_sweep(s->H, s->k, s->n, r1, s->Gw, s->st, ep_ok ? &eph : nullptr); if (rc) return rc;
            if (eph.done) { rc = launch_gram_reduce(s->gram_scratch, eph.nblk, s->k, s->Gh, s->st, nullptr, nullptr, 1.0); s->packed_fresh[1] = true; }
            else rc = gram_h(s);
            if (rc) return rc;
            rc = prod2(s);    if (rc) return rc;
            break;
        }
        case SMK_ALG_BPP:  // nmf_solver_bpp.hpp:342-377
            rc = nnls_side(s, 0, s->H, 0, s->n, r1, s->Gw); if (rc) return rc;
            rc = gram_h(s);   if (rc) return rc;
            rc = prod2(s);    if (rc) return rc;
            if (s->w_sharded) {
                // W is replicated in the algorithm but its rows are independent NNLS problems: every rank solves its own
                // blocks (held back to back, one launch) once the last slice of the reduce-scatter has landed
                rc = wait_r2(s);  if (rc) return rc;
                rc = nnls_side(s, 1, s->Wown, 0, s->n_own, view_own(s), s->Gh); if (rc) return rc;
                s->w_full = false;
            } else if (s->ar && s->world > 1) {
                // callback hook (one primitive only): every rank solves a row range, zeroes the rest and sum-all-reduces
                const i64 base = s->m / s->world, extra = s->m % s->world;
                const i64 i0 = s->rank * base + (s->rank < extra ? s->rank : extra);
                const i64 i1 = i0 + base + (s->rank < extra ? 1 : 0);
                rc = nnls_side(s, 1, s->Wt, i0, i1, r2, s->Gh); if (rc) return rc;
                if (i0 > 0) SMK_HIP(hipMemsetAsync(s->Wt, 0, (size_t)i0 * s->KP * sizeof(double), s->st));
                if (i1 < s->m) SMK_HIP(hipMemsetAsync(s->Wt + i1 * s->KP, 0, (size_t)(s->m - i1) * s->KP * sizeof(double), s->st));
                rc = dist_allreduce_cb(s, s->Wt, (i64)s->KP * s->m, 1); if (rc) return rc;
            } else {
                rc = wait_r2(s);  if (rc) return rc;
                rc = nnls_side(s, 1, s->Wt, 0, s->m, r2, s->Gh); if (rc) return rc;
            }
            rc = gram_w(s);   if (rc) return rc;
            rc = prod1(s);    if (rc) return rc;
            break;
        case SMK_ALG_RANK2:  // nmf_solver_rank2.hpp:353-455
            // each closed-form solve also emits partial sums of the Gram matrix of its result (no second pass over H / W),
            // which the kernel that needs the matrix adds up itself (rank2.hip)
            {
                double *GpH = s->r2_scratch, *GpW = s->r2_scratch + rank2_gram_scratch_elems(std::max(s->m, s->n)) / 2;
                int nbh = 0, nbw = 0;
                rc = launch_rank2_solve(s->H, s->Hc, s->n, r1, s->Gw, nullptr, 0, 0, s->fail_flag, s->iter, GpH, &nbh, s->Gh,
                                        is_dist(s) ? 1 : 0, s->st); if (rc) return rc;       // sharded: HH' is finished here and summed over the ranks
                rc = allreduce_gh(s); if (rc) return rc;
                rc = prod2(s);    if (rc) return rc;
                rc = wait_r2(s);  if (rc) return rc;
                rc = launch_rank2_solve(s->Wt, nullptr, s->m, r2, s->Gh, GpH, nbh, 1, s->fail_flag, s->iter, GpW, &nbw, s->Graw, 0, s->st); if (rc) return rc;
                // NormalizeAndScale(W, H, ScaleFactors) every iteration, norms from the Gram matrix of the new W;
                // also rescales HH' and AH', leaves Gw = W'W of the normalised W (D^-1 Graw D^-1) and the compact copy of W
                rc = launch_rank2_normalize(s->H, s->n, s->Wt, s->Wc, s->m, r2, s->Gh, s->Graw, GpW, nbw, s->Gw, s->fail_flag, s->st); if (rc) return rc;
                s->wc_valid = s->Wc != nullptr;
            }
            rc = prod1(s);    if (rc) return rc;
            break;
        default:
            return SMK_UNSUPPORTED;
    }
    s->iter += 1;
    return 0;
}

static int resolve_events(smk_solver* s)
{
    for (int w = 0; w < 6; ++w) {
        for (auto& e : s->ev[w]) {
            float ms = 0.f;
            SMK_HIP(hipEventElapsedTime(&ms, e.e0, e.e1));
            s->acc_ms[w] += (double)ms;                                // sampled totals; smk_solver_kernel_time scales them by passes seen / passes sampled
            s->launches[w] += e.counts;
            (void)hipEventDestroy(e.e0);
            (void)hipEventDestroy(e.e1);
        }
        s->ev[w].clear();
    }
    return 0;
}

// wait for the stream; translate device-side failure flags into Result codes
static int sync_and_check(smk_solver* s, int* fail_iter)
{
    int flag = INT_MAX;
    int rc = wait_r2(s);                 // chunk exchanges still on the second stream
    if (rc) return rc;
    SMK_HIP(hipMemcpyAsync(&flag, s->fail_flag, sizeof(int), hipMemcpyDeviceToHost, s->st));
    SMK_HIP(hipStreamSynchronize(s->st));
    rc = resolve_events(s);
    if (rc) return rc;
    if (fail_iter) *fail_iter = flag;
    if (flag != INT_MAX) return SMK_FAILURE;
    return SMK_OK;
}

// sharded: sum the H-side scalar over the ranks and make the failure flag the same everywhere, so that every
// rank takes the same branch of the driver loop (a rank that stopped alone would strand the others in a collective)
static int dist_agree(smk_solver* s)
{
    const int wpart = w_rows_sharded(s) ? 1 : 0;      // the W-side sum covers this rank's rows only
    int rc = launch_dist_scalars(s->scal, s->fail_flag, s->iter > 0 ? s->iter - 1 : 0, 0, wpart, s->st);
    if (rc) return rc;
    rc = dist_allreduce_now(s, s->scal + 5, 3, 1);
    if (rc) return rc;
    return launch_dist_scalars(s->scal, s->fail_flag, s->iter > 0 ? s->iter - 1 : 0, 1, wpart, s->st);
}

// progress_est->Update(iter, W, H, gradW, gradH): returns the metric (synchronises)
// the kernels (and, sharded, the scalar all-reduce) of one progress evaluation; results land in s->scal
static int enqueue_progress_kernels(smk_solver* s)
{
    int rc = wait_r2(s);
    if (rc) return rc;
    if (s->o.prog_est_algorithm == SMK_PROG_DELTA_FNORM) {
        rc = gather_w(s);             // a row-sharded W: the norm runs over all of it (BPP's default rule is PG_RATIO)
        if (rc) return rc;
        rc = launch_delta_fnorm(s->Wt, s->Wprev, (i64)s->KP * s->m, s->pg_partials, s->scal + 2, s->st);
        if (rc) return rc;
    } else {
        // gradW = W*HHt - AHt  (slot 0, replicated), gradH = WtW*H - WtA (slot 1, local shard)
        if (w_rows_sharded(s)) {      // only this rank's rows of the summed (AH')' exist here: partial sum, joined in dist_agree
            rc = s->n_own > 0 ? launch_grad_pg(s->Wown, s->k, s->n_own, view_own(s), s->Gh, nullptr, s->pg_partials, s->scal, 0, s->st, s->wide_tmp)
                              : launch_zero_f64(s->scal, 1, s->st);
        } else {
            rc = launch_grad_pg(s->Wt, s->k, s->m, view2(s), s->Gh, nullptr, s->pg_partials, s->scal, 0, s->st, s->wide_tmp);
        }
        if (rc) return rc;
        rc = launch_grad_pg(s->H, s->k, s->n, view1(s), s->Gw, nullptr, s->pg_partials + s->pg_half, s->scal, 1, s->st, s->wide_tmp);
        if (rc) return rc;
    }
    if (is_dist(s)) { rc = dist_agree(s); if (rc) return rc; }
    return 0;
}

// metric from the four scalars (PG sums for W and H, delta-Fnorm numerator / denominator)
static int evaluate_progress(smk_solver* s, const double* h, int iter_index, double* metric)
{
    if (s->o.prog_est_algorithm == SMK_PROG_DELTA_FNORM) {
        *metric = std::sqrt(h[2]) / std::sqrt(h[3]);
    } else {
        const double pg = std::sqrt(h[0] + h[1]);
        if (std::isnan(pg)) { set_error("ProjectedGradientNorm: NaN"); return SMK_FAILURE; }   // reference throws
        if (iter_index == 0) { s->pg0 = pg; *metric = 1.0; }
        else *metric = pg / s->pg0;
    }
    s->last_metric = *metric;
    return SMK_OK;
}

// progress_est->Update(iter, W, H, gradW, gradH): returns the metric (synchronises)
static int update_progress(smk_solver* s, int iter_index, double* metric)
{
    int rc = enqueue_progress_kernels(s);
    if (rc) return rc;
    double h[4] = {0, 0, 0, 0};
    SMK_HIP(hipMemcpyAsync(h, s->scal, 4 * sizeof(double), hipMemcpyDeviceToHost, s->st));
    int fail_iter = INT_MAX;
    rc = sync_and_check(s, &fail_iter);
    if (rc) return rc;
    return evaluate_progress(s, h, iter_index, metric);
}

// ---- the same evaluation, one iteration late ------------------------------------------------
// Enqueue the progress kernels of the iteration that has just been enqueued, copy their scalars and
// the failure flag into pinned slot `b` and record an event; when `snapshot` is set also keep
// (W, H, W'W) of this iteration so that the NEXT, speculatively enqueued iteration can be undone.
static int progress_prealloc(smk_solver* s)
{
    if (!s->pin) {
        SMK_HIP(hipHostMalloc((void**)&s->pin, smk_solver::PROG_SLOTS * sizeof(smk_solver::ProgSlot)));
        for (int i = 0; i < smk_solver::PROG_SLOTS; ++i) SMK_HIP(hipEventCreateWithFlags(&s->pev[i], hipEventDisableTiming));
    }
    for (int b = 0; b < 2; ++b)            // a sharded run keeps ONE check in flight (two slots)
        if (!s->snap[b]) { const int rc = dev_alloc(&s->snap[b], snapshot_elems(s->k, s->m, s->n)); if (rc) return rc; }
    return 0;
}

static int progress_begin(smk_solver* s, int b, bool snapshot, bool allow_defer = true)
{
    if (!s->pin) {
        SMK_HIP(hipHostMalloc((void**)&s->pin, smk_solver::PROG_SLOTS * sizeof(smk_solver::ProgSlot)));
        for (int i = 0; i < smk_solver::PROG_SLOTS; ++i) SMK_HIP(hipEventCreateWithFlags(&s->pev[i], hipEventDisableTiming));
    }
    int rc = wait_r2(s);
    if (rc) return rc;
    s->poll_tag[b] = 0.0;
    s->pin[b].h[7] = 0.0;
    // deferred (check_rides_in_nnls): nothing is enqueued now -- the next H-side NNLS launch forms the sum; a snapshot can be
    // promised only if this iteration's NNLS launches have been writing it (iter_snap_slot)
    if (allow_defer && s->pg_defer_slot < 0 && (!snapshot || s->iter_snap_slot == b) && check_rides_in_nnls(s)) {
        s->pg_defer_slot = b;
        s->pg_defer_snap = snapshot;
        s->pg_defer_tag = s->iter - 1;              // the iteration just enqueued
        s->pin[b].fused = 1;
        return 0;
    }
    ++s->check_routes[1];
    if (s->o.algorithm == SMK_ALG_RANK2 && s->o.prog_est_algorithm == SMK_PROG_PG_RATIO && !is_dist(s)) {
        // one launch: both gradient sums, the failure flag and the snapshot (rank2.hip)
        if (snapshot && !s->snap[b]) { rc = dev_alloc(&s->snap[b], snapshot_elems(s->k, s->m, s->n)); if (rc) return rc; }
        const size_t pe = rank2_progress_scratch_elems(s->m, s->n);
        if (!s->pin_r2[b]) SMK_HIP(hipHostMalloc((void**)&s->pin_r2[b], pe * sizeof(double)));
        rc = launch_rank2_progress(s->Wt, s->m, view2(s), s->Gh, s->H, s->n, view1(s), s->Gw, s->r2_prog, s->fail_flag,
                                   snapshot ? s->snap[b] : nullptr, s->st);
        if (rc) return rc;
        s->pin[b].fused = 2;             // per-workgroup partials: progress_end adds them up in index order
        SMK_HIP(hipMemcpyAsync(s->pin_r2[b], s->r2_prog, pe * sizeof(double), hipMemcpyDeviceToHost, s->st));
        SMK_HIP(hipEventRecord(s->pev[b], s->st));
        return 0;
    }
    static const bool fused_check = [] { const char* e = getenv("SMK_PROGRESS_FUSED"); return !(e && e[0] == '0'); }();
    // BPP: gradW is the dual of the W-side NNLS (nmf_solver_bpp.hpp:362-366), whose projected-gradient sum is exactly zero after a
    // solve that reached optimality (a solve that did not has raised the failure flag); SMK_BPP_GRADW=1 forms HH' W' - (AH')' anyway
    static const bool bpp_dual_is_gradient = [] { const char* e = getenv("SMK_BPP_GRADW"); return !(e && e[0] == '1'); }();
    if (fused_check && s->o.prog_est_algorithm == SMK_PROG_PG_RATIO && !is_dist(s) && !is_wide(s->k) && !s->w_sharded) {
        // round 6: gradients + snapshot in one launch, sums + failure flag written into the pinned slot by a second (kernels.hip:
        // grad_pg2_snap_kernel, sum_partials2_host_kernel); SMK_PROGRESS_FUSED=0: the four stream operations of before
        if (snapshot && !s->snap[b]) { rc = dev_alloc(&s->snap[b], snapshot_elems(s->k, s->m, s->n)); if (rc) return rc; }
        const double tag = progress_polls() ? (double)(s->iter + 1) : 0.0;
        rc = launch_grad_pg2_fused(s->Wt, s->m, view2(s), s->Gh, s->pg_partials, s->H, s->n, view1(s), s->Gw, s->pg_partials + s->pg_half,
                                   s->k, s->scal, s->fail_flag, 5, snapshot ? s->snap[b] : nullptr, s->pin[b].h, s->st,
                                   (s->o.algorithm == SMK_ALG_BPP && bpp_dual_is_gradient) ? 1 : 0, tag);
        if (rc) return rc;
        s->pin[b].fused = 1;
        s->poll_tag[b] = tag;
        if (tag == 0.0) SMK_HIP(hipEventRecord(s->pev[b], s->st));
        return 0;
    }
    if (s->o.prog_est_algorithm == SMK_PROG_PG_RATIO && !is_dist(s) && !is_wide(s->k)) {
        // both gradients in one launch, both sums + the failure flag in a second, one 64-byte read-back
        rc = launch_grad_pg2(s->Wt, s->m, view2(s), s->Gh, s->pg_partials, s->H, s->n, view1(s), s->Gw,
                             s->pg_partials + s->pg_half, s->k, s->scal, s->fail_flag, 5, s->st);
        if (rc) return rc;
        s->pin[b].fused = 1;
        SMK_HIP(hipMemcpyAsync(s->pin[b].h, s->scal, 8 * sizeof(double), hipMemcpyDeviceToHost, s->st));
    } else {
        rc = enqueue_progress_kernels(s);
        if (rc) return rc;
        s->pin[b].fused = 0;
        SMK_HIP(hipMemcpyAsync(s->pin[b].h, s->scal, 4 * sizeof(double), hipMemcpyDeviceToHost, s->st));
        SMK_HIP(hipMemcpyAsync(&s->pin[b].flag, s->fail_flag, sizeof(int), hipMemcpyDeviceToHost, s->st));
    }
    if (snapshot) {
        if (!s->snap[b]) { rc = dev_alloc(&s->snap[b], snapshot_elems(s->k, s->m, s->n)); if (rc) return rc; }
        rc = s->w_sharded ? launch_snapshot(s->Wown, s->n_own, s->H, s->n, s->Gw, s->snap[b], s->k, 1, s->st)
                          : launch_snapshot(s->Wt, s->m, s->H, s->n, s->Gw, s->snap[b], s->k, 1, s->st);
        if (rc) return rc;
    }
    SMK_HIP(hipEventRecord(s->pev[b], s->st));
    return 0;
}

// How many progress checks may be in flight.  ONE is the default, as in rounds 1 - 5: the host enqueues iteration i + 1, then waits
// for the event of check i.  Round 6 suspected that wait of starving the device on small problems (C2: 13 500 it/s unchecked,
// 11 000 checked) and generalised the loop to a ring of slots (pinned result, event, snapshot per in-flight check; the rule still
// evaluated for every iteration in order, a run that converges at p restored from p's snapshot) -- measured with two and three
// checks in flight: 10 980 / 11 100 it/s against 11 010 (profiles/r06_progress_check.txt).  The host is not the limit; what a
// checked iteration costs is the gradient pass itself, which re-reads the S slabs of both right-hand sides (32 MB at C2).
// SMK_PROGRESS_DEPTH=2|3 keeps the deeper pipeline selectable; sharded runs always use one (their checks carry collectives).
static int progress_depth(const smk_solver* s)
{
    static const int forced = [] { const char* e = getenv("SMK_PROGRESS_DEPTH"); return e ? atoi(e) : 0; }();
    if (is_dist(s) || s->comm || forced <= 1) return 1;
    if (snapshot_elems(s->k, s->m, s->n) * sizeof(double) > ((size_t)1 << 30)) return 1;
    return std::min(forced, smk_solver::PROG_SLOTS - 1);
}

// wait for slot b; SMK_FAILURE when the device flagged a solver failure up to that iteration
static int progress_end(smk_solver* s, int b, int iter_index, double* metric)
{
    if (s->pg_defer_slot == b) {                    // no further iteration was enqueued: the check is formed the direct way now
        s->pg_defer_slot = -1;
        const int frc = progress_begin(s, b, false, false);
        if (frc) return frc;
    }
    if (s->poll_tag[b] != 0.0) {
        // the kernel that forms the totals stores the tag behind them (system scope); the stream going idle without it is an error
        const volatile double* tagp = &s->pin[b].h[7];
        const double want = s->poll_tag[b];
        for (unsigned long spins = 1; *tagp != want; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((spins & 0x3FFFF) == 0) {
                const hipError_t q = hipStreamQuery(s->st);
                if (q == hipErrorNotReady) continue;
                if (q == hipSuccess && *tagp == want) break;
                set_error(q == hipSuccess ? "the progress check's result never arrived in its pinned slot" : hipGetErrorString(q));
                return q == hipSuccess ? SMK_FAILURE : SMK_DEVICE_ERROR;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        s->poll_tag[b] = 0.0;
    } else {
        SMK_HIP(hipEventSynchronize(s->pev[b]));
    }
    if (s->pin[b].fused == 2) {
        const int nb = rank2_progress_blocks(s->m, s->n);
        const double* p = s->pin_r2[b];
        double sw = 0.0, sh = 0.0;
        for (int i = 0; i < nb; ++i) { sw += p[2 * i]; sh += p[2 * i + 1]; }
        s->pin[b].h[0] = sw;
        s->pin[b].h[1] = sh;
        s->pin[b].flag = (int)p[2 * (size_t)nb];
    } else if (s->pin[b].fused) s->pin[b].flag = (int)s->pin[b].h[5];
    if (s->pin[b].flag != INT_MAX) return SMK_FAILURE;
    return evaluate_progress(s, s->pin[b].h, iter_index, metric);
}

static int progress_restore(smk_solver* s, int b)
{
    int rc = s->w_sharded ? launch_snapshot(s->Wown, s->n_own, s->H, s->n, s->Gw, s->snap[b], s->k, 0, s->st)
                          : launch_snapshot(s->Wt, s->m, s->H, s->n, s->Gw, s->snap[b], s->k, 0, s->st);
    if (rc) return rc;
    // whatever the undone iteration did to the failure flag is void; products / HH' are stale
    const int big = INT_MAX;
    SMK_HIP(hipMemcpyAsync(s->fail_flag, &big, sizeof(int), hipMemcpyHostToDevice, s->st));
    SMK_HIP(hipStreamSynchronize(s->st));
    s->inited = false;
    s->wc_valid = false;
    if (s->w_sharded) s->w_full = false;    // the snapshot holds this rank's own rows; the full copy is stale
    return 0;
}

static int normalize_device(smk_solver* s)
{
    if (s->normalized) return 0;
    { const int grc = gather_w(s); if (grc) return grc; }
    // nu_c^2 = (W'W)[c][c]; Gw is current in all three schedules after the last W update
    int rc = launch_scale_rows(s->H, s->k, s->n, s->Gw, 0, s->fail_flag, s->st);
    if (rc) return rc;
    rc = launch_scale_rows(s->Wt, s->k, s->m, s->Gw, 1, s->fail_flag, s->st);
    if (rc) return rc;
    if (s->w_sharded) { rc = scatter_own(s); if (rc) return rc; }      // the own rows follow the scaled full copy
    s->wc_valid = false;
    s->normalized = true;
    // Gw/Gh and the stored products describe the un-normalised factors: a later iterate()/run() on this
    // handle starts from solver.Init on the scaled (W, H), exactly like a fresh solver given them
    s->inited = false;
    return 0;
}

// The fused HALS W sweep reported expired polls (-3: some workgroup was not resident): go back to the initial
// factors and latch the one-launch-per-column path.  Returns 1 when the caller should repeat its work.
static int hals_fail_soft(smk_solver* s)
{
    if (s->o.algorithm != SMK_ALG_HALS || s->hals_multi || !s->W0c) return 0;
    int flag = INT_MAX;
    if (hipMemcpy(&flag, s->fail_flag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || flag != -3) return 0;
    fprintf(stderr, "smallk_amd: fused HALS W sweep could not keep its workgroups resident; repeating the run on the per-column path\n");
    s->hals_multi = true;
    const int big = INT_MAX;
    (void)hipMemcpyAsync(s->Wt, s->W0c, (size_t)s->KP * s->m * sizeof(double), hipMemcpyDeviceToDevice, s->st);
    (void)hipMemcpyAsync(s->H, s->H0c, (size_t)s->KP * s->n * sizeof(double), hipMemcpyDeviceToDevice, s->st);
    (void)hipMemcpyAsync(s->fail_flag, &big, sizeof(int), hipMemcpyHostToDevice, s->st);
    (void)hals_w_scratch_init(s->hals_scratch, s->k, s->m, s->st);
    (void)hipStreamSynchronize(s->st);
    s->inited = false;
    s->normalized = false;
    s->iter = 0;
    s->pg0 = 1.0;
    s->last_metric = 1.0;
    return 1;
}

// An NNLS launch that packs its own result (NnlsPack) met an entry beyond fp16's range after scaling (flag -4): the a-priori
// bound behind its row scales did not hold -- it cannot while the other factor is non-negative and the solve reaches a KKT
// point, so this is a safety net.  Back to the initial factors with the packing latched off for the life of the handle.
// Returns 1 when the caller should repeat its work.
static int pack_fail_soft(smk_solver* s)
{
    if (!s->pack_in_solve || s->pack_in_solve_off || !s->W0c) return 0;
    int flag = INT_MAX;
    if (hipMemcpy(&flag, s->fail_flag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || flag != NNLS_PACK_OVERFLOW) return 0;
    fprintf(stderr, "smallk_amd: an NNLS launch could not pack its result inside fp16's range; repeating the run with the separate pack launch\n");
    s->pack_in_solve_off = true;
    const int big = INT_MAX;
    (void)hipMemcpyAsync(s->Wt, s->W0c, (size_t)s->KP * s->m * sizeof(double), hipMemcpyDeviceToDevice, s->st);
    (void)hipMemcpyAsync(s->H, s->H0c, (size_t)s->KP * s->n * sizeof(double), hipMemcpyDeviceToDevice, s->st);
    (void)hipMemcpyAsync(s->fail_flag, &big, sizeof(int), hipMemcpyHostToDevice, s->st);
    (void)hipStreamSynchronize(s->st);
    s->inited = false;
    s->normalized = false;
    s->iter = 0;
    s->pg0 = 1.0;
    s->last_metric = 1.0;
    return 1;
}

// ---- run-time guard of the product form ------------------------------------------------------------------------------
// Which product form a run takes is decided when the solver is created (rank, algorithm, the spread of the column scales:
// thresholds found by sweeps).  A matrix outside the swept families must not leave the 1e-4 bar silently, so the choice is
// can be re-examined while the run goes on (OPT-IN: SMK_GUARD_EVERY=16; default 0 = off, see the note at the end).  Every
// SMK_GUARD_EVERY iterations of a BPP run the solver forms W'A for a sample
// of 64 columns in the ACCURATE form (fp64 factor against the stored data, bigprod_f64_kernel) and compares it with what the
// fast form has just produced for the same columns: delta = ||P_fast - P_acc||_F / ||P_acc||_F, the product error ON THIS
// DATA (4e-8 for well-scaled data; much more when small entries sit next to large ones).  To first order an error delta in
// the right-hand sides moves the solution of the k x k normal equations by at most cond(G) delta, G = W'W or HH', and the
// condition numbers are cheap: both Gram matrices come back with the two sums (k <= 256: Cholesky + inverse on the host,
// 1-norm).  When max(cond) * delta exceeds SMK_GUARD_TAU (default 1e-4: ONE iteration could move the factors by the parity
// bar) the solver changes to the accurate form for the rest of the run: plans and buffers are rebuilt and solver.Init is
// repeated on the current factors, exactly what a fresh solver given them would do.  The check is asynchronous -- enqueued
// behind one iteration, read GUARD_EVERY iterations later -- so it never stalls the stream (C4: one 67 MB pass per 16
// iterations of 137 GB each; measured -1.1 % on C4, -8 % on the 0.09 ms iterations of C2 where the host runs at most 16
// iterations ahead).  Not for sharded runs (every rank would have to agree; their form is agreed once, at attach).
// WHY IT IS NOT ON BY DEFAULT (round 4, profiles/r04_guard_cases.txt): cond(G) delta is a worst-case bound, and the
// non-negativity constraints take the ill-conditioned directions out of the problems that are actually solved.  HALS on
// uniform noise at k = 8 shows cond(HH') = 2.3e4, delta = 1.2e-8, product 2.6e-4 > tau, while the run ends 1.7e-6 from the
// oracle after 40 iterations; with the guard on, C3 (HALS, k = 32) changes to the accurate form and runs at 253 instead of
// 1160 iterations/s.  For block pivoting -- where the bound describes the solves that are really done -- the cases measured
// separate cleanly (noise: 2e-6 .. 7e-6 over 40 iterations; a planted factor with nearly collinear columns: > 1e-2), so the
// guard is offered for BPP runs on data outside the swept families, and the static rules of smk_solver_create stay the default.
static bool guard_applies(const smk_solver* s)
{
    static const int every = [] { const char* e = getenv("SMK_GUARD_EVERY"); return e ? atoi(e) : 0; }();
    if (every <= 0 || s->guard_off || s->a->sparse || s->nsplit == NSPLIT_F64 || is_dist(s) || s->comm) return false;
    if (s->o.algorithm != SMK_ALG_BPP) return false;
    return s->k <= 256 && s->n >= 64 && s->iter > 0 && s->iter % every == 0;
}

// 1-norm condition number of the live k x k block of a KP x KP symmetric matrix (inf when it is not positive definite)
static double cond1_spd(const double* G, int KP, int k)
{
    std::vector<double> L((size_t)k * k, 0.0), X((size_t)k * k, 0.0);
    double n1 = 0.0;
    for (int c = 0; c < k; ++c) { double sum = 0.0; for (int r = 0; r < k; ++r) sum += std::fabs(G[(size_t)c * KP + r]); n1 = std::max(n1, sum); }
    for (int j = 0; j < k; ++j) {                                  // G = L L'
        double d = G[(size_t)j * KP + j];
        for (int p = 0; p < j; ++p) d -= L[(size_t)j * k + p] * L[(size_t)j * k + p];
        if (!(d > 0.0)) return INFINITY;
        const double ljj = std::sqrt(d);
        L[(size_t)j * k + j] = ljj;
        for (int i = j + 1; i < k; ++i) {
            double v = G[(size_t)j * KP + i];
            for (int p = 0; p < j; ++p) v -= L[(size_t)i * k + p] * L[(size_t)j * k + p];
            L[(size_t)i * k + j] = v / ljj;
        }
    }
    double ninv = 0.0;
    std::vector<double> y((size_t)k);
    for (int c = 0; c < k; ++c) {                                  // column c of the inverse: L y = e_c, L' x = y
        for (int i = 0; i < k; ++i) {
            double v = i == c ? 1.0 : 0.0;
            for (int p = 0; p < i; ++p) v -= L[(size_t)i * k + p] * y[(size_t)p];
            y[(size_t)i] = v / L[(size_t)i * k + i];
        }
        double sum = 0.0;
        for (int i = k - 1; i >= 0; --i) {
            double v = y[(size_t)i];
            for (int p = i + 1; p < k; ++p) v -= L[(size_t)p * k + i] * X[(size_t)p];
            X[(size_t)i] = v / L[(size_t)i * k + i];
            sum += std::fabs(X[(size_t)i]);
        }
        ninv = std::max(ninv, sum);
    }
    return n1 * ninv;
}

static int guard_resolve(smk_solver* s)
{
    if (!s->guard_pending) return 0;
    s->guard_pending = false;
    SMK_HIP(hipEventSynchronize(s->guard_ev));
    static const double tau = [] { const char* e = getenv("SMK_GUARD_TAU"); return e ? atof(e) : 1e-4; }();
    const double* p = s->guard_pin;
    const double delta = p[1] > 0.0 ? std::sqrt(p[0] / p[1]) : 0.0;
    const size_t kk = (size_t)s->KP * s->KP;
    const double cond = std::max(cond1_spd(p + 2, s->KP, s->k), cond1_spd(p + 2 + kk, s->KP, s->k));
    s->guard_checks += 1;
    s->guard_last = cond * delta;
    static const bool verbose = [] { const char* e = getenv("SMK_GUARD_VERBOSE"); return e && atoi(e) != 0; }();
    if (verbose) fprintf(stderr, "[smk guard] iteration %d: delta %.3e, cond %.3e, product %.3e (tau %.1e)\n", s->iter, delta, cond, cond * delta, tau);
    if (!(cond * delta > tau)) return 0;
    // change to the accurate form: new plans and buffers, then solver.Init on the current factors
    s->guard_fired += 1;
    s->nsplit = NSPLIT_F64;
    int rc = plan_products(s);
    if (!rc && alloc_product_buffers(s)) rc = SMK_DEVICE_ERROR;
    if (rc) return rc;
    s->packed_fresh[0] = s->packed_fresh[1] = false;
    s->nnls_gram_nblk[0] = s->nnls_gram_nblk[1] = 0;
    const int iter_keep = s->iter;
    rc = solver_init(s);
    s->iter = iter_keep;
    return rc;
}

static int guard_enqueue(smk_solver* s)
{
    const size_t kk = (size_t)s->KP * s->KP;
    if (!s->guard_As) {                    // first use: the sample (64 columns, evenly spaced), plans, buffers
        const int nc = 64;
        std::vector<unsigned> cols((size_t)nc);
        for (int i = 0; i < nc; ++i) cols[(size_t)i] = (unsigned)((i64)i * s->n / nc);
        const i64 es = elem_size(s->a->storage);
        int rc = dev_alloc(&s->guard_cols, (size_t)nc);
        rc |= dev_alloc((unsigned char**)&s->guard_As, (size_t)s->a->ldA * COL_PAD * es);
        rc |= dev_alloc(&s->guard_dev, 2 + 2 * kk);
        if (rc) return SMK_DEVICE_ERROR;
        SMK_HIP(hipMemsetAsync(s->guard_As, 0, (size_t)s->a->ldA * COL_PAD * es, s->st));
        SMK_HIP(hipMemcpyAsync(s->guard_cols, cols.data(), (size_t)nc * sizeof(unsigned), hipMemcpyHostToDevice, s->st));
        rc = launch_gather_cols(s->a->A, s->a->ldA * es, s->guard_cols, nc, s->guard_As, s->a->ldA * es, s->a->ldA * es, s->st);
        if (rc) return rc;
        SMK_HIP(hipStreamSynchronize(s->st));              // `cols` leaves scope
        const int ng = plan_bigprod_groups(s->a->storage, s->k, s->m, nc, NSPLIT_F64, g_cus, s->guard_pl);
        for (int g = 0; g < ng; ++g) s->guard_pl[g].ldx = s->KP;
        if (dev_alloc(&s->guard_P, s->guard_pl[0].p_elems)) return SMK_DEVICE_ERROR;
        SMK_HIP(hipHostMalloc((void**)&s->guard_pin, (2 + 2 * kk) * sizeof(double)));
        SMK_HIP(hipEventCreateWithFlags(&s->guard_ev, hipEventDisableTiming));
        s->guard_ncols = nc;
    }
    // the full fp64 W (s->Wt) against the sampled columns, one launch per group of 64 factor rows
    for (int g = 0; g < s->ng; ++g) {
        const int rc = launch_bigprod(s->guard_pl[g], s->guard_As, s->a->ldA, s->Wt + s->guard_pl[g].k0, s->guard_P + s->guard_pl[g].k0, s->st);
        if (rc) return rc;
    }
    const BigProdPlan& gp = s->guard_pl[0];
    int rc = launch_guard_compare(view1(s), s->guard_cols, s->guard_ncols, s->guard_P, gp.S, (i64)gp.ncols_pad * gp.pstride, gp.pstride, s->k,
                                  s->guard_dev, s->st);
    if (rc) return rc;
    SMK_HIP(hipMemcpyAsync(s->guard_dev + 2, s->Gw, kk * sizeof(double), hipMemcpyDeviceToDevice, s->st));
    SMK_HIP(hipMemcpyAsync(s->guard_dev + 2 + kk, s->Gh, kk * sizeof(double), hipMemcpyDeviceToDevice, s->st));
    SMK_HIP(hipMemcpyAsync(s->guard_pin, s->guard_dev, (2 + 2 * kk) * sizeof(double), hipMemcpyDeviceToHost, s->st));
    SMK_HIP(hipEventRecord(s->guard_ev, s->st));
    s->guard_pending = true;
    return 0;
}

// after every iteration: read the check enqueued GUARD_EVERY iterations ago, enqueue the next one
static int guard_step(smk_solver* s)
{
    if (!guard_applies(s)) return 0;
    int rc = guard_resolve(s);
    if (rc || s->nsplit == NSPLIT_F64) return rc;
    rc = wait_r2(s);                      // HH' / the stored products are final
    if (rc) return rc;
    return guard_enqueue(s);
}

int smk_solver_iterate(smk_solver* s, int iters)
{
    if (!s || iters < 0) return SMK_BAD_PARAM;
    if (!s->have_factors) { set_error("set_factors() first"); return SMK_BAD_PARAM; }
    int rc = 0;
    if (!s->inited) { rc = solver_init(s); if (rc) return rc; }
    for (int i = 0; i < iters; ++i) {
        rc = solver_iteration(s);
        if (rc) return rc;
        rc = guard_step(s);
        if (rc) return rc;
    }
    return SMK_OK;
}

// `iters` iterations as NmfSolve<> runs them past min_iter (nmf_solve_generic.hpp:98-121): after every iteration the stopping rule's
// metric is formed (gradients for PG_RATIO, nmf_solver_bpp.hpp:370-377 / nmf_solver_mu.hpp:151-164; W - Wprev for DELTA_FNORM) and
// read back -- by the same one-iteration-late scheme as smk_solver_run (snapshot, pinned slot, event), with a tolerance that never
// fires.  What bench.py --check-every-iteration times.  *last_metric (optional): the metric of the last iteration.
int smk_solver_iterate_checked(smk_solver* s, int iters, double* last_metric)
{
    if (!s || iters < 0) return SMK_BAD_PARAM;
    if (!s->have_factors) { set_error("set_factors() first"); return SMK_BAD_PARAM; }
    int rc = 0;
    if (!s->inited) { rc = solver_init(s); if (rc) return rc; }
    if (s->o.algorithm == SMK_ALG_RANK2) { set_error("iterate_checked: MU / HALS / BPP"); return SMK_UNSUPPORTED; }
    const int depth = progress_depth(s), NS = smk_solver::PROG_SLOTS;
    const int base = s->iter;                       // the very first iteration of a run initialises the estimator (PG_RATIO: pg0)
    std::deque<int> pend;                           // iterations whose check is outstanding, oldest first
    double metric = 1.0;
    for (int i = 0; i < iters; ++i) {
        s->iter_snap_slot = check_rides_in_nnls(s) ? i % NS : -1;      // this iteration's NNLS launches also write its snapshot
        rc = solver_iteration(s);
        s->iter_snap_slot = rc ? -1 : s->iter_snap_slot;
        if (rc) return rc;
        rc = guard_step(s);
        if (rc) { s->iter_snap_slot = -1; return rc; }
        rc = progress_begin(s, i % NS, true);
        s->iter_snap_slot = -1;
        if (rc) return rc;
        while ((int)pend.size() >= depth) {
            rc = progress_end(s, pend.front() % NS, base + pend.front(), &metric);
            if (rc) return rc;
            pend.pop_front();
        }
        pend.push_back(i);
    }
    while (!pend.empty()) {
        rc = progress_end(s, pend.front() % NS, base + pend.front(), &metric);
        if (rc) return rc;
        pend.pop_front();
    }
    if (last_metric) *last_metric = metric;
    return SMK_OK;
}

// which product form the solver is using now (SMK_NSPLIT numbering; 8 = the accurate form), how often the run-time guard has
// looked and how often it changed the form, and cond * delta of its last look
int smk_solver_product_form(const smk_solver* s, int* guard_checks, int* guard_fired, double* guard_last)
{
    if (!s) return SMK_BAD_PARAM;
    if (guard_checks) *guard_checks = s->guard_checks;
    if (guard_fired) *guard_fired = s->guard_fired;
    if (guard_last) *guard_last = s->guard_last;
    return s->nsplit;
}

int smk_solver_sync(smk_solver* s)
{
    if (!s) return SMK_BAD_PARAM;
    const int target = s->iter;
    int rc = gather_w(s);                 // every rank calls sync: the fp64 W is whole again afterwards
    if (rc) return rc;
    rc = sync_and_check(s, nullptr);
    if (rc == SMK_FAILURE && !is_dist(s) && (hals_fail_soft(s) || pack_fail_soft(s))) {
        rc = smk_solver_iterate(s, target);
        if (rc == SMK_OK) rc = sync_and_check(s, nullptr);
    }
    return rc;
}

int smk_solver_progress(smk_solver* s, double* metric)
{
    if (!s || !metric) return SMK_BAD_PARAM;
    if (!s->inited) return SMK_BAD_PARAM;
    return update_progress(s, s->iter > 0 ? s->iter - 1 : 0, metric);
}

int smk_solver_iteration_count(const smk_solver* s) { return s ? s->iter : 0; }

static int solver_run_once(smk_solver* s, smk_stats* stats);

// NmfSolve<>, common/include/nmf_solve_generic.hpp:34-140
int smk_solver_run(smk_solver* s, smk_stats* stats)
{
    int rc = solver_run_once(s, stats);
    if (rc == SMK_FAILURE && s && !is_dist(s) && (hals_fail_soft(s) || pack_fail_soft(s))) rc = solver_run_once(s, stats);
    return rc;
}

// ---- RANK2 on sparse A: the driver loop below as ONE launch (rank2_persist.hip) -------------------------------------------
// For small and medium node matrices an iteration is six 7 - 12 us launches on a few MB: launch-latency bound.  The
// resident kernel carries the loop, the stopping rule (PG_RATIO, min_iter / tolcount as NmfSolve<>), the per-iteration
// normalisation and the final state; the host waits for one 128-byte result.  Measured on the C5-shaped run
// (profiles/r04_c5_*): 54 -> 37 us per iteration on the 1 M-entry nodes (the two gather products alone are 24 us at the
// chip's ~84 G random 16-byte gathers per second), and it also wins on the large nodes (9 M entries: 495 -> 291 us, the
// 16 M-entry root 645 -> 556 us) because its entry-parallel products issue exactly one gather per stored entry with all of
// a chunk's gathers in flight -- so there is no size limit by default (SMK_R2_PERSIST_NNZ sets one, SMK_R2_PERSIST=0 turns
// the kernel off).
// devices on which a resident launch could not synchronise its workgroups (a CU mask, another process holding CUs): latched
// for the PROCESS, not per solver handle -- HierNMF2 and flatclust create a solver per node and per trial, and each would pay the
// start-up deadline again and print the warning again
static std::atomic<unsigned long long> g_r2p_off_devices{0};
static bool rank2_persist_eligible(const smk_solver* s)
{
    static const int mode = [] { const char* e = getenv("SMK_R2_PERSIST"); return e ? atoi(e) : 1; }();
    static const i64 max_nnz = [] { const char* e = getenv("SMK_R2_PERSIST_NNZ"); return e ? (i64)atoll(e) : (i64)1 << 40; }();
    if (!mode || s->r2p_off) return false;
    if (g_r2p_off_devices.load(std::memory_order_relaxed) & (1ull << (smk_current_device() & 63))) return false;
    if (s->o.algorithm != SMK_ALG_RANK2 || !s->a->sparse || s->o.prog_est_algorithm != SMK_PROG_PG_RATIO) return false;
    if (is_dist(s) || s->comm || s->o.verbose || s->timing || !s->Hc || !s->Wc) return false;
    if (rank2_persist_workgroups(s->m, s->n, s->a->nnz, g_cus) < 1) return false;      // more than 4096 rows per workgroup
    return s->a->nnz <= max_nnz || mode == 2;
}

// returns 0 with *status = one of R2P_*: CONVERGED / EXHAUSTED (factors, W'W, counters in place), SOLVER_FAILED / NAN (the
// failure flag is set as the launch-per-kernel loop sets it), ABORTED (nothing was touched: the caller runs the classic loop)
static int rank2_persist_run(smk_solver* s, int* status, int* count)
{
    const int nwg = rank2_persist_workgroups(s->m, s->n, s->a->nnz, g_cus);
    if (nwg < 1 || nwg > 1024) { *status = R2P_ABORTED; return 0; }
    // TEST HOOK: behave as if the kernel's workgroups had not all become resident (the caller must then finish the run on the
    // launch-per-kernel loop from the state solver.Init left)
    if (const char* e = getenv("SMK_R2P_TEST_ABORT")) if (atoi(e) != 0) { *status = R2P_ABORTED; return 0; }
    if (!s->r2p_sync) {
        int rc = dev_alloc(&s->r2p_hc1, (size_t)2 * s->n);
        rc |= dev_alloc(&s->r2p_r2c, (size_t)2 * s->m);
        rc |= dev_alloc(&s->r2p_part, (size_t)3 * 8 * 1024);                  // three arrays of [workgroups <= 1024][8]
        rc |= dev_alloc(&s->r2p_out, (size_t)16);
        rc |= dev_alloc((unsigned char**)&s->r2p_sync, rank2_persist_sync_bytes());
        if (rc) return SMK_DEVICE_ERROR;
        SMK_HIP(hipHostMalloc((void**)&s->r2p_pin, 16 * sizeof(double)));
    }
    R2PersistArgs a;
    a.colptr = s->a->colptr; a.rowidx = s->a->rowidx; a.val = s->a->val;
    a.colptr_t = s->a->colptr_t; a.rowidx_t = s->a->rowidx_t; a.val_t = s->a->val_t;
    a.m = s->m; a.n = s->n;
    a.Gw0 = s->Gw; a.R1 = view1(s);
    a.Wc = s->Wc; a.Hc0 = s->Hc; a.Hc1 = s->r2p_hc1; a.R2c = s->r2p_r2c;
    a.gp_h = s->r2p_part; a.gp_w = s->r2p_part + 8 * 1024; a.pgp = s->r2p_part + 2 * 8 * 1024;
    a.sync = s->r2p_sync;
    a.min_iter = s->o.min_iter; a.max_iter = s->o.max_iter; a.tolcount = s->o.tolcount; a.tol = s->o.tol;
    a.iter_tag0 = s->iter;
    a.Wt = s->Wt; a.H = s->H; a.Gw = s->Gw;
    a.fail_flag = s->fail_flag;
    a.lds_bytes = (unsigned)rank2_persist_lds_bytes();
    a.out = s->r2p_out;
    int rc = launch_rank2_persist(a, nwg, s->st);
    if (rc == 1) { *status = R2P_ABORTED; return 0; }       // the grid does not fit the device as it is now
    if (rc) return rc;
    SMK_HIP(hipMemcpyAsync(s->r2p_pin, s->r2p_out, 16 * sizeof(double), hipMemcpyDeviceToHost, s->st));
    SMK_HIP(hipStreamSynchronize(s->st));
    *status = (int)s->r2p_pin[0];
    *count = (int)s->r2p_pin[1];
    {
        static const bool prof = [] { const char* e = getenv("SMK_R2P_PROFILE"); return e && atoi(e) != 0; }();
        if (prof) {
            const double it = std::max(1.0, s->r2p_pin[2]);
            fprintf(stderr, "[r2p] %ld x %ld nnz %ld: %d workgroups, status %d, %.0f iterations; per iteration (workgroup 0): "
                    "B1 %.1f us, phase W %.1f, B2 %.1f, phase G %.1f\n", (long)s->m, (long)s->n, (long)s->a->nnz, nwg, *status, it,
                    s->r2p_pin[8] / it, s->r2p_pin[9] / it, s->r2p_pin[10] / it, s->r2p_pin[11] / it);
        }
    }
    if (*status == R2P_CONVERGED || *status == R2P_EXHAUSTED) {
        s->iter += (int)s->r2p_pin[2];
        s->pg0 = s->r2p_pin[3];
        s->last_metric = s->r2p_pin[4];
        s->normalized = false;
        s->inited = false;             // HH' and the stored products were never materialised: a later run starts from solver.Init
        s->wc_valid = false;
    }
    return 0;
}

static int solver_run_once(smk_solver* s, smk_stats* stats)
{
    if (!s) return SMK_BAD_PARAM;
    set_error("");                                  // smk_last_error() describes THIS run afterwards
    if (!s->have_factors) { set_error("set_factors() first"); return SMK_BAD_PARAM; }
    const smk_options& o = s->o;
    const double t0 = wall_us();
    int rc = 0, result = SMK_OK;
    bool success = false;
    int iter = 0, success_count = 0, fail_iter = INT_MAX;
    static std::mutex r2p_device_mu[64];
    std::unique_lock<std::mutex> r2p_lock;

    if (!s->inited) { rc = solver_init(s); if (rc) { result = rc; goto done; } }

    // One resident kernel per device at a time: two of them launched together (two host threads with a context each on ONE
    // device, the two-device HierNMF2 test) would each get part of the CUs and wait for workgroups that cannot start.  The
    // second caller WAITS (it does not take the other path: which path runs must not depend on timing, the two differ in the
    // last bits and the tree search is discrete).  Contexts on different devices do not meet here.
    if (rank2_persist_eligible(s)) {
        const int dev = smk_current_device();
        r2p_lock = std::unique_lock<std::mutex>(r2p_device_mu[dev >= 0 && dev < 64 ? dev : 0]);
    }
    if (r2p_lock.owns_lock()) {
        int status = R2P_ABORTED, count = 0;
        rc = rank2_persist_run(s, &status, &count);
        r2p_lock.unlock();
        if (rc) { result = rc; goto done; }
        if (status == R2P_CONVERGED || status == R2P_EXHAUSTED) {
            success = true;
            iter = count;
            goto finish;
        }
        if (status == R2P_NAN) { set_error("ProjectedGradientNorm: NaN"); iter = count; result = SMK_FAILURE; goto done; }
        if (status == R2P_SOLVER_FAILED) { result = SMK_FAILURE; goto failed_check; }
        // ABORTED: some workgroup never became resident (or the deadline passed); the solver state is the one solver.Init
        // left, so the launch-per-kernel loop below takes over, latched for the life of the handle
        s->r2p_off = true;
        s->wc_valid = false;           // the compact copy of W served as the kernel's work array
        const unsigned long long bit = 1ull << (smk_current_device() & 63);
        if (!(g_r2p_off_devices.fetch_or(bit, std::memory_order_relaxed) & bit))       // once per device and process
            fprintf(stderr, "smallk_amd: the resident RANK2 kernel could not synchronise its workgroups on device %d; this process continues on the launch-per-kernel path there\n", smk_current_device());
    }

    // The stopping rule of iteration i (NmfSolve, nmf_solve_generic.hpp:81-121) is evaluated AFTER
    // iteration i+1 has been enqueued: the host never leaves the GPU idle waiting for a scalar.  When
    // the rule fires for i, the state of i is restored from its snapshot and i+1 is discarded, so
    // results and iteration counts are those of the check-every-iteration loop (SMK_SYNC_PROGRESS=1
    // runs that loop instead).
    {
        static const bool sync_mode = [] { const char* e = getenv("SMK_SYNC_PROGRESS"); return e && atoi(e) != 0; }();
        const int depth = progress_depth(s), NS = smk_solver::PROG_SLOTS;
        std::deque<int> pend;                          // iterations whose check is outstanding, oldest first (at most `depth`)
        auto resolve = [&](int p, bool speculated) -> int {      // 0: go on, 1: converged at p, < 0: error
            double metric = 1.0;
            int prc = progress_end(s, p % NS, p, &metric);
            if (prc) { result = prc; return -1; }
            if (p < o.min_iter) return 0;              // iteration 0 only initialises the estimator
            if (o.verbose && ((p + 1 < 10) || ((p + 1) % 10 == 0)))
                printf("%d:\tprogress metric:\t%g\n", p + 1, metric);     // nmf_progress_estimation.hpp:22-33
            if (metric <= o.tol) {
                if (++success_count >= o.tolcount) {
                    if (o.verbose) printf("\nSolution converged after %d iterations.\n\n", p + 1);
                    if (speculated) { prc = progress_restore(s, p % NS); if (prc) { result = prc; return -1; } s->iter = p + 1; }
                    return 1;
                }
            } else {
                success_count = 0;
            }
            return 0;
        };
        for (iter = 0; iter < o.max_iter; ++iter) {
            // (an iteration whose check will want a snapshot lets its NNLS launches write it: check_rides_in_nnls)
            s->iter_snap_slot = (!sync_mode && iter >= o.min_iter && check_rides_in_nnls(s)) ? iter % NS : -1;
            rc = solver_iteration(s);
            if (rc) { s->iter_snap_slot = -1; result = rc; goto done; }
            rc = guard_step(s);
            if (rc) { s->iter_snap_slot = -1; result = rc; goto done; }
            const bool check = (iter == 0) || (iter >= o.min_iter);
            if (sync_mode) {
                if (!check) { if (o.verbose) printf("%d:\tprogress metric: \t(min_iter)\n", iter + 1); continue; }
                rc = progress_begin(s, iter % NS, false);
                if (rc) { result = rc; goto done; }
                const int r = resolve(iter, false);
                if (r < 0) goto failed_check;
                if (iter < o.min_iter && o.verbose) printf("%d:\tprogress metric: \t(min_iter)\n", iter + 1);
                if (r == 1) { success = true; break; }
                continue;
            }
            if (check) {
                rc = progress_begin(s, iter % NS, iter >= o.min_iter);
                s->iter_snap_slot = -1;
                if (rc) { result = rc; goto done; }
            }
            s->iter_snap_slot = -1;
            // the oldest outstanding checks, once `depth` of them are in flight (depth 1: the check of the previous iteration,
            // as in rounds 1 - 5); an unchecked iteration (0 < iter < min_iter) leaves nothing behind, as before
            while (!pend.empty() && ((int)pend.size() >= depth || !check)) {
                const int p = pend.front();
                pend.pop_front();
                const int r = resolve(p, true);
                if (r < 0) goto failed_check;
                if (r == 1) { success = true; iter = p; pend.clear(); break; }
            }
            if (success) break;
            if (iter < o.min_iter && o.verbose) printf("%d:\tprogress metric: \t(min_iter)\n", iter + 1);
            if (check) pend.push_back(iter);
        }
        while (!success && !pend.empty()) {            // what is still outstanding when the iterations are used up
            const int p = pend.front();
            pend.pop_front();
            const int r = resolve(p, !pend.empty());   // the very last one: nothing ran after it
            if (r < 0) goto failed_check;
            if (r == 1) { success = true; iter = p; pend.clear(); }
        }
    }

finish:
    rc = gather_w(s);
    if (rc) { result = rc; goto done; }
    if (o.normalize) { rc = normalize_device(s); if (rc) { result = rc; goto done; } }
    if (is_dist(s)) { rc = dist_agree(s); if (rc) { result = rc; goto done; } }
    rc = sync_and_check(s, &fail_iter);
    if (rc) { result = rc; goto failed_check; }
    if (!success && iter == o.max_iter) success = true;
    result = success ? SMK_OK : SMK_FAILURE;
    goto done;

failed_check:
    if (result == SMK_FAILURE) {
        // which iteration set the device flag (BPP pivot limit / non-SPD / zero column norm)
        int flag = INT_MAX;
        (void)hipMemcpy(&flag, s->fail_flag, sizeof(int), hipMemcpyDeviceToHost);
        if (flag != INT_MAX && flag >= 0) {
            iter = flag;
            fprintf(stderr, "\tNMF solver failure on iteration %d\n", iter + 1);
        }
    }
done:
    if (stats) {
        stats->elapsed_us = (unsigned long long)(wall_us() - t0);
        stats->iteration_count = iter;
    }
    return result;
}

// NnlsHals (common/include/nnls.hpp:249-316): W stays fixed, H is swept with the HALS row update until
// the projected-gradient norm of H drops below tol * (its value after the first sweep).  W'A is
// formed once by the streaming product, every iteration is two column-tile kernels over H.
// Returns SMK_OK on convergence (W, H then normalised like the reference does) or SMK_FAILURE when
// the iteration limit is reached.
int smk_solver_nnls_hals(smk_solver* s, double tol, int verbose, int max_iter, int* iterations)
{
    if (!s || max_iter < 0) return SMK_BAD_PARAM;
    if (!s->have_factors) { set_error("set_factors() first"); return SMK_BAD_PARAM; }
    if (is_dist(s)) { set_error("NnlsHals: not available on a sharded solver"); return SMK_UNSUPPORTED; }
    if (verbose) printf("\nRunning NNLS solver...\n");
    int rc = 0;
    // W'A is formed ONCE and every sweep of the loop below reads it: the accurate product form (the fp64 product of the stored
    // data) for the price of one slower pass, so that the converged H differs from the reference's by summation order only
    if (!s->a->sparse && s->nsplit != NSPLIT_F64 && !getenv("SMK_NSPLIT")) {
        s->nsplit = NSPLIT_F64;
        rc = plan_products(s);
        if (!rc && alloc_product_buffers(s)) rc = SMK_DEVICE_ERROR;
        if (rc) return rc;
    }
    rc = gram_w(s);
    if (!rc) rc = prod1(s);
    if (rc) return rc;
    bool success = false;
    double pg0 = 0.0;
    int i = 0;
    for (i = 0; i < max_iter; ++i) {
        rc = launch_hals_sweep(s->H, s->k, s->n, view1(s), s->Gw, s->st);
        if (!rc) rc = launch_grad_pg(s->H, s->k, s->n, view1(s), s->Gw, nullptr, s->pg_partials + s->pg_half, s->scal, 1, s->st, s->wide_tmp);
        if (rc) return rc;
        double sum = 0.0;
        SMK_HIP(hipMemcpyAsync(&sum, s->scal + 1, sizeof(double), hipMemcpyDeviceToHost, s->st));
        SMK_HIP(hipStreamSynchronize(s->st));
        const double pg = std::sqrt(sum);
        if (std::isnan(pg)) { set_error("ProjectedGradientNorm: NaN"); return SMK_FAILURE; }
        if (i == 0) {
            pg0 = pg;
            if (verbose) printf("1:\tprogress metric:\t%g\n", 1.0);
            continue;
        }
        if (verbose && ((i + 1 < 10) || ((i + 1) % 10 == 0))) printf("%d:\tprogress metric:\t%g\n", i + 1, pg / pg0);
        if (pg < tol * pg0) {
            success = true;
            s->normalized = false;
            rc = normalize_device(s);
            if (rc) return rc;
            break;
        }
    }
    if (iterations) *iterations = success ? i + 1 : i;
    s->inited = false;       // Gw/R1 no longer describe a solver schedule
    rc = sync_and_check(s, nullptr);
    if (rc) return rc;
    if (!success) fprintf(stderr, "NNLS solver reached iteration limit.\n");
    return success ? SMK_OK : SMK_FAILURE;
}

// NnlsBlockpivot(LHS, RHS, X, Y), common/include/nnls.hpp:144-244, on its own: min ||.|| s.t. X >= 0 for
// LHS X = RHS with LHS k x k SPD, warm start X (passive set = X > 0), dual Y = LHS X - RHS.
int smk_nnls_blockpivot(int k, int64_t ncols, const double* LHS, int64_t ldL, const double* RHS, int64_t ldR, double* X,
                        int64_t ldX, double* Y, int64_t ldY)
{
    if (!g_init) { set_error("smk_initialize() has not been called"); return SMK_NOTINITIALIZED; }
    if (k <= 0 || ncols <= 0 || !LHS || !RHS || !X || ldL < k || ldR < k || ldX < k || (Y && ldY < k)) return SMK_BAD_PARAM;
    if (k > MAX_K_BPP) { set_error("device path supports k <= 2048"); return SMK_UNSUPPORTED; }
    const int KP = kp_of(k);
    std::vector<double> hg((size_t)KP * KP, 0.0), hr((size_t)KP * ncols, 0.0), hx((size_t)KP * ncols, 0.0);
    for (int c = 0; c < k; ++c)
        for (int r = 0; r < k; ++r) hg[(size_t)c * KP + r] = LHS[(size_t)c * ldL + r];
    for (int64_t c = 0; c < ncols; ++c)
        for (int r = 0; r < k; ++r) {
            hr[(size_t)c * KP + r] = RHS[(size_t)c * ldR + r];
            hx[(size_t)c * KP + r] = X[(size_t)c * ldX + r];
        }
    double *dg = nullptr, *dr = nullptr, *dx = nullptr, *dy = nullptr, *dscratch = nullptr;
    int* dflag = nullptr;
    int rc = 0;
    rc |= dev_alloc(&dg, hg.size());
    rc |= dev_alloc(&dr, hr.size());
    rc |= dev_alloc(&dx, hx.size());
    rc |= dev_alloc(&dy, hx.size());
    rc |= dev_alloc(&dscratch, nnls_uses_tiles(k) ? nnls_wide_scratch_elems(k, g_cus, ncols) : nnls_scratch_elems(k));
    rc |= dev_alloc(&dflag, (size_t)1);
    unsigned* ddefer = nullptr;
    if (KP == 64 && !nnls_uses_tiles(k)) rc |= dev_alloc(&ddefer, nnls_defer_elems(ncols));
    struct Free { std::vector<void*> p; ~Free() { for (void* q : p) if (q) (void)smk::dev_free(q); } } guard{{dg, dr, dx, dy, dscratch, dflag, ddefer}};
    if (rc) return SMK_DEVICE_ERROR;
    const int big = INT_MAX;
    SMK_HIP(hipMemcpyAsync(dg, hg.data(), hg.size() * sizeof(double), hipMemcpyHostToDevice, g_stream));
    SMK_HIP(hipMemcpyAsync(dr, hr.data(), hr.size() * sizeof(double), hipMemcpyHostToDevice, g_stream));
    SMK_HIP(hipMemcpyAsync(dx, hx.data(), hx.size() * sizeof(double), hipMemcpyHostToDevice, g_stream));
    SMK_HIP(hipMemsetAsync(dy, 0, hx.size() * sizeof(double), g_stream));
    SMK_HIP(hipMemcpyAsync(dflag, &big, sizeof(int), hipMemcpyHostToDevice, g_stream));
    const PartialView pv{dr, 1, 0, KP, 1};
    rc = launch_nnls_bpp(dx, dy, k, 0, ncols, pv, dg, dflag, 0, dscratch, 0, g_cus, g_stream, nullptr, nullptr, nullptr, ddefer);
    if (rc) return rc;
    int flag = INT_MAX;
    SMK_HIP(hipMemcpyAsync(hx.data(), dx, hx.size() * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    SMK_HIP(hipMemcpyAsync(hr.data(), dy, hx.size() * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    SMK_HIP(hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, g_stream));
    SMK_HIP(hipStreamSynchronize(g_stream));
    for (int64_t c = 0; c < ncols; ++c)
        for (int r = 0; r < k; ++r) {
            X[(size_t)c * ldX + r] = hx[(size_t)c * KP + r];
            if (Y) Y[(size_t)c * ldY + r] = hr[(size_t)c * KP + r];
        }
    return flag == INT_MAX ? SMK_OK : SMK_FAILURE;
}

int smk_solver_get_factors(smk_solver* s, int normalize, double* W, int64_t ldW, double* H, int64_t ldH)
{
    if (!s || !W || !H) return SMK_BAD_PARAM;
    if (ldW < s->m || ldH < s->k) { set_error("leading dimension too small"); return SMK_BAD_PARAM; }
    int rc = gather_w(s);                 // a collective when W is row-sharded and stale: every rank must be here
    if (rc) return rc;
    if (normalize) { rc = normalize_device(s); if (rc) return rc; }
    rc = launch_transpose_f64(s->Wt, s->KP, s->tmpW, s->m, s->k, s->m, s->st);
    if (rc) return rc;
    // Device-to-host copies are kept contiguous: a strided hipMemcpy2DAsync into pageable memory leaves
    // ~190 KiB of runtime staging behind per call on this ROCm (1200 one-shot sparse runs: 230 MiB).
    if (ldW == s->m) {
        SMK_HIP(hipMemcpyAsync(W, s->tmpW, (size_t)s->m * s->k * sizeof(double), hipMemcpyDeviceToHost, s->st));
    } else {
        for (int c = 0; c < s->k; ++c)
            SMK_HIP(hipMemcpyAsync(W + (size_t)c * ldW, s->tmpW + (size_t)c * s->m, (size_t)s->m * sizeof(double),
                                   hipMemcpyDeviceToHost, s->st));
    }
    if (s->k == s->KP && ldH == s->k) {
        SMK_HIP(hipMemcpyAsync(H, s->H, (size_t)s->k * s->n * sizeof(double), hipMemcpyDeviceToHost, s->st));
    } else {
        if (!s->tmpH) { rc = dev_alloc(&s->tmpH, (size_t)s->k * s->n); if (rc) return rc; }
        rc = launch_compact_rows(s->H, s->KP, s->tmpH, s->k, s->n, s->st);
        if (rc) return rc;
        if (ldH == s->k) {
            SMK_HIP(hipMemcpyAsync(H, s->tmpH, (size_t)s->k * s->n * sizeof(double), hipMemcpyDeviceToHost, s->st));
        } else {      // caller's leading dimension exceeds k: compact on the device, scatter on the host
            std::vector<double> tmp((size_t)s->k * s->n);
            SMK_HIP(hipMemcpyAsync(tmp.data(), s->tmpH, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, s->st));
            SMK_HIP(hipStreamSynchronize(s->st));
            for (i64 c = 0; c < s->n; ++c)
                std::copy(tmp.begin() + c * s->k, tmp.begin() + (c + 1) * s->k, H + c * ldH);
        }
    }
    return sync_and_check(s, nullptr);
}

int smk_solver_enable_timing(smk_solver* s, int on)
{
    if (!s) return SMK_BAD_PARAM;
    s->timing = on != 0;
    // one pass in 16 under 1 GB of streamed matrix per pass, one in 8 under 32 GB (a C4 shard in chunks: 8 launches and 8
    // collectives per iteration would carry ~0.1 ms of event gaps in 3.4 ms), every pass above
    {
        const double bytes = (double)s->m * (double)s->n * (s->a->sparse ? 12.0 : (double)elem_size(s->a->storage));
        s->timing_stride = bytes < (double)((i64)1 << 30) ? 16 : bytes < 32.0 * (double)((i64)1 << 30) ? 8 : 1;
    }
    if (const char* e = getenv("SMK_TIMING_STRIDE")) s->timing_stride = std::max(1, atoi(e));
    s->pass_counter[0] = s->pass_counter[1] = 0;
    s->pass_sampled[0] = s->pass_sampled[1] = 0;
    for (int w = 0; w < 6; ++w) { s->acc_ms[w] = 0.0; s->launches[w] = 0; }
    return SMK_OK;
}

int smk_solver_kernel_time(smk_solver* s, int which, double* total_ms, int* launches)
{
    if (!s || which < 0 || which > 5) return SMK_BAD_PARAM;      // 2: the (AH')' sum and the W all-gather of a sharded run; 3: main-stream waits for them; 5: block pivoting
    if (which == 4) {       // calibration brackets: unscaled (their average is what matters)
        if (total_ms) *total_ms = s->acc_ms[4];
        if (launches) *launches = s->launches[4];
        return SMK_OK;
    }
    // one pass in `timing_stride` carries events: totals are scaled by the TRUE ratio passes seen / passes sampled (100 passes
    // at stride 8 are 13 samples standing for 100, not for 104); the collectives (slot 2) are sampled with either pass
    const unsigned seen = which < 2 ? s->pass_counter[which] : s->pass_counter[0] + s->pass_counter[1];
    const unsigned sampled = which < 2 ? s->pass_sampled[which] : s->pass_sampled[0] + s->pass_sampled[1];
    const double f = sampled > 0 ? (double)seen / (double)sampled : 1.0;
    if (total_ms) *total_ms = s->acc_ms[which] * f;
    if (launches) *launches = (int)llround((double)s->launches[which] * f);
    return SMK_OK;
}

// the kernel pass `which` launches (what the bench lines and profiles attribute their time to); same decision as timed_spmm /
// launch_bigprod
int smk_solver_kernel_name(const smk_solver* s, int which, char* out, int cap)
{
    if (!s || which < 0 || which > 2 || !out || cap < 8) return SMK_BAD_PARAM;
    std::string name;
    if (which == 2) {
        static const char* const route[4] = {"", "launches of its own behind the iteration",
                                             "sums in the next H-side NNLS launch, totals as one launch",
                                             "sums in the next H-side NNLS launch, totals in the tail of the pass behind it"};
        for (int r = 3; r >= 1; --r)
            if (s->check_routes[r]) name += (name.empty() ? "" : "; ") + std::to_string(s->check_routes[r]) + " x " + route[r];
        if (name.empty()) name = "none formed yet";
    } else if (s->a->sparse) {
        const BlockedCsc& blk = (which == 0) ? s->a->bA : s->a->bAt;
        const SegPlan& seg = (which == 0) ? s->a->segA : s->a->segAt;
        const i64 ncols = which == 0 ? s->n : s->m;
        const int ldx = s->Wc ? 2 : s->KP;
        if (s->r2p_sync && s->o.algorithm == SMK_ALG_RANK2) name = "smk::rank2_persist_kernel";
        else if (ldx == 2 && blk.nb > 1) name = "smk::spmm_blocked2_kernel";
        else if (ldx == s->KP && s->k > 2 && !is_wide(s->k) && seg.ncols == ncols && seg.rowflag && !seg.uniform) name = "smk::spmm_seg_kernel";
        else name = "smk::spmm_gather_kernel";
    } else {
        const BigProdPlan& pl = which == 0 ? s->pl1 : s->pl2;
        name = s->nsplit == NSPLIT_F64 ? (s->k <= 2 ? "smk::bigprod_f64_k2_kernel" : "smk::bigprod_f64_kernel")
             : s->a->storage == SMK_STORE_F32 ? "smk::bigprod_f3_kernel" : "smk::bigprod_kernel";
        name += " variant " + std::to_string(pl.variant) + (pl.tr ? " (transposed source)" : "");
    }
    snprintf(out, (size_t)cap, "%s", name.c_str());
    return SMK_OK;
}

int smk_debug_nnls_stats(unsigned long long* out256, int reset)
{
    if (!g_init) return SMK_NOTINITIALIZED;
    const int rc = nnls_stats_read(out256, reset);
    return rc == 0 ? SMK_OK : rc == -1 ? SMK_UNSUPPORTED : SMK_DEVICE_ERROR;
}

int smk_solver_kernel_work(const smk_solver* s, int which, double* bytes, double* flops)
{
    if (!s || which < 0 || which > 1) return SMK_BAD_PARAM;
    if (s->a->sparse) {   // per nonzero: 12 bytes of A (value + row index) + one KP-row of X gathered
        if (bytes) *bytes = (double)s->a->nnz * (12.0 + 8.0 * (s->Wc ? 2 : s->KP));      // RANK2 gathers 16 B rows of the compact copy
        if (flops) *flops = 2.0 * (double)s->a->nnz * s->k;
        return SMK_OK;
    }
    const double mn = (double)s->m * (double)s->n;
    if (bytes) *bytes = mn * elem_size(s->a->storage);
    if (flops) *flops = 2.0 * mn * s->k / s->ng;      // per launch: k > 64 streams the matrix once per group of 64 rows
    return SMK_OK;
}

// Result Nmf(...), common/src/nmf.cpp:173-229
int smk_nmf_dense(const smk_options* opts, const double* A, int64_t ldA, double* W, int64_t ldW, double* H,
                  int64_t ldH, smk_stats* stats, int storage)
{
    if (!g_init) {
        fprintf(stderr, "nmflib error: nmf_initialize() must be called prior to any factorization routine\n\n");
        return SMK_NOTINITIALIZED;
    }
    if (!opts || !smk_is_valid(opts, 1)) return SMK_BAD_PARAM;
    if (!A || !W || !H) return SMK_BAD_PARAM;
    if (opts->k > MAX_K || (opts->algorithm == SMK_ALG_BPP && opts->k > MAX_K_BPP)) { set_error("device path supports k <= 2048"); return SMK_UNSUPPORTED; }     // before anything is uploaded
    const int64_t m = opts->height, n = opts->width;
    if (ldA < m || ldW < m || ldH < opts->k) { set_error("leading dimension too small"); return SMK_BAD_PARAM; }
    smk_matrix* a = nullptr;
    smk_solver* s = nullptr;
    int rc = smk_matrix_create(&a, m, n, 0, n, storage);
    if (rc == SMK_OK) rc = smk_matrix_upload_f64(a, A, ldA);
    if (rc == SMK_OK) rc = smk_solver_create(&s, opts, a);
    if (rc == SMK_OK) rc = smk_solver_set_factors(s, W, ldW, H, ldH);
    int run_rc = SMK_OK;
    if (rc == SMK_OK) {
        run_rc = smk_solver_run(s, stats);
        // like the reference, W/H hold the last iterate even when the solver reports failure
        if (run_rc == SMK_OK || run_rc == SMK_FAILURE) (void)smk_solver_get_factors(s, 0, W, ldW, H, ldH);
        rc = run_rc;
    }
    smk_solver_destroy(s);
    smk_matrix_destroy(a);
    return rc;
}

// Result Nmf(...) on `nshards` column shards of A, one host thread + one HIP device per shard: A and H
// column-sharded, W replicated, RCCL collectives issued from C on each shard's streams.  `devices` NULL: shard r on
// device r.  local_stub != 0: every shard on the CURRENT device with the in-process stand-in for RCCL (which refuses
// two ranks on one device) -- for boxes with fewer GPUs than shards, and for the tests.
int smk_nmf_dense_sharded(const smk_options* opts, const double* A, int64_t ldA, double* W, int64_t ldW, double* H,
                          int64_t ldH, smk_stats* stats, int storage, int nshards, const int* devices, int local_stub)
{
    if (!ctx().init) {
        fprintf(stderr, "nmflib error: nmf_initialize() must be called prior to any factorization routine\n\n");
        return SMK_NOTINITIALIZED;
    }
    if (!opts || !smk_is_valid(opts, 1)) return SMK_BAD_PARAM;
    if (!A || !W || !H || nshards < 1 || nshards > 16) return SMK_BAD_PARAM;
    if (opts->k > MAX_K || (opts->algorithm == SMK_ALG_BPP && opts->k > MAX_K_BPP)) { set_error("device path supports k <= 2048"); return SMK_UNSUPPORTED; }
    const int64_t m = opts->height, n = opts->width;
    if (ldA < m || ldW < m || ldH < opts->k) { set_error("leading dimension too small"); return SMK_BAD_PARAM; }
    if (nshards > n) nshards = (int)n;
    if (nshards == 1) return smk_nmf_dense(opts, A, ldA, W, ldW, H, ldH, stats, storage);
    int dev0 = 0;
    SMK_HIP(hipGetDevice(&dev0));
    if (!local_stub) {
        int ndev = 0;
        SMK_HIP(hipGetDeviceCount(&ndev));
        for (int r = 0; r < nshards; ++r)
            if ((devices ? devices[r] : r) >= ndev) { set_error("smk_nmf_dense_sharded: not enough devices for the shards"); return SMK_BAD_PARAM; }
    }
    std::vector<smk_comm*> comms((size_t)nshards, nullptr);
    std::vector<int> devs((size_t)nshards);
    for (int r = 0; r < nshards; ++r) devs[(size_t)r] = local_stub ? dev0 : (devices ? devices[r] : r);
    int rc = local_stub ? smk_comm_init_local(comms.data(), nshards) : smk_comm_init_all(comms.data(), nshards, devs.data());
    if (rc != SMK_OK) return rc;

    std::vector<int> rcs((size_t)nshards, SMK_OK);
    std::vector<smk_stats> sts((size_t)nshards, smk_stats{0, 0});
    std::vector<std::string> errs((size_t)nshards);
    std::vector<double> Wcopy((size_t)m * opts->k);          // every shard starts from the same W0
    for (int c = 0; c < opts->k; ++c) std::copy(W + (size_t)c * ldW, W + (size_t)c * ldW + m, Wcopy.begin() + (size_t)c * m);
    auto worker = [&](int r) {
        DeviceCtx local;
        t_ctx = &local;
        int wrc = SMK_OK;
        smk_matrix* a = nullptr;
        smk_solver* s = nullptr;
        const int64_t base = n / nshards, extra = n % nshards;
        const int64_t c0 = r * base + std::min<int64_t>(r, extra), nc = base + (r < extra ? 1 : 0);
        wrc = smk_initialize(devs[(size_t)r]);
        if (wrc == SMK_OK) wrc = smk_matrix_create(&a, m, n, c0, nc, storage);
        if (wrc == SMK_OK) wrc = smk_matrix_upload_f64(a, A + (size_t)c0 * ldA, ldA);
        if (wrc == SMK_OK) wrc = smk_solver_create(&s, opts, a);
        if (wrc == SMK_OK) wrc = smk_solver_attach_comm(s, comms[(size_t)r]);
        if (wrc == SMK_OK) wrc = smk_solver_set_factors(s, Wcopy.data(), m, H + (size_t)c0 * ldH, ldH);
        // a shard that failed before the first collective would strand the others: agree on the setup first
        {
            double ok = (wrc == SMK_OK) ? 0.0 : 1.0, *dflag = nullptr;
            if (smk::dev_malloc((void**)&dflag, sizeof(double)) == hipSuccess) {
                (void)hipMemcpy(dflag, &ok, sizeof(double), hipMemcpyHostToDevice);
                (void)comm_allreduce(comms[(size_t)r], dflag, 1, 1, ctx().stream);
                (void)hipStreamSynchronize(ctx().stream);
                (void)hipMemcpy(&ok, dflag, sizeof(double), hipMemcpyDeviceToHost);
                (void)smk::dev_free(dflag);
            }
            if (ok != 0.0 && wrc == SMK_OK) wrc = SMK_FAILURE;
        }
        if (wrc == SMK_OK) {
            wrc = smk_solver_run(s, &sts[(size_t)r]);
            // anything but an agreed result (OK / FAILURE are all-reduced) means this rank left the loop alone: release the
            // peers that wait for it in a collective
            if (wrc != SMK_OK && wrc != SMK_FAILURE) smk_comm_abort(comms[(size_t)r]);
            if (wrc == SMK_OK || wrc == SMK_FAILURE) {
                // W is replicated: rank 0 returns it; every rank returns its own columns of H
                std::vector<double> Wl(r == 0 ? 0 : (size_t)m * opts->k);
                (void)smk_solver_get_factors(s, 0, r == 0 ? W : Wl.data(), r == 0 ? ldW : m, H + (size_t)c0 * ldH, ldH);
            }
        }
        errs[(size_t)r] = g_err;
        smk_solver_destroy(s);
        smk_matrix_destroy(a);
        smk_finalize();
        rcs[(size_t)r] = wrc;
        t_ctx = nullptr;
    };
    std::vector<std::thread> th;
    for (int r = 0; r < nshards; ++r) th.emplace_back(worker, r);
    for (auto& t : th) t.join();
    for (smk_comm* c : comms) smk_comm_destroy(c);
    (void)hipSetDevice(dev0);
    int result = SMK_OK;
    for (int r = 0; r < nshards; ++r)
        if (rcs[(size_t)r] != SMK_OK && result == SMK_OK) { result = rcs[(size_t)r]; set_error(errs[(size_t)r]); }
    if (stats) *stats = sts[0];
    return result;
}

// Result NmfSparse(...), common/src/nmf.cpp:232-300 (CSC input, 32-bit indices as in the reference)
int smk_nmf_sparse(const smk_options* opts, unsigned height, unsigned width, unsigned nz, const unsigned* col_offsets,
                   const unsigned* row_indices, const double* data, double* W, int64_t ldW, double* H, int64_t ldH,
                   smk_stats* stats)
{
    if (!g_init) {
        fprintf(stderr, "nmflib error: nmf_initialize() must be called prior to any factorization routine\n\n");
        return SMK_NOTINITIALIZED;
    }
    if (!opts || !smk_is_valid(opts, 1)) return SMK_BAD_PARAM;
    if (!col_offsets || !row_indices || !data || !W || !H) return SMK_BAD_PARAM;
    if (opts->k > MAX_K || (opts->algorithm == SMK_ALG_BPP && opts->k > MAX_K_BPP)) { set_error("device path supports k <= 2048"); return SMK_UNSUPPORTED; }
    if ((int64_t)height != opts->height || (int64_t)width != opts->width) return SMK_BAD_PARAM;
    if (ldW < opts->height || ldH < opts->k) { set_error("leading dimension too small"); return SMK_BAD_PARAM; }
    smk_matrix* a = nullptr;
    smk_solver* s = nullptr;
    int rc = smk_matrix_create_sparse(&a, height, width, 0, width, nz, col_offsets, row_indices, data);
    if (rc == SMK_OK) rc = smk_solver_create(&s, opts, a);
    if (rc == SMK_OK) rc = smk_solver_set_factors(s, W, ldW, H, ldH);
    if (rc == SMK_OK) {
        rc = smk_solver_run(s, stats);
        if (rc == SMK_OK || rc == SMK_FAILURE) (void)smk_solver_get_factors(s, 0, W, ldW, H, ldH);
    }
    smk_solver_destroy(s);
    smk_matrix_destroy(a);
    return rc;
}

}  // extern "C"
