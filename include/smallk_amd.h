/*
 * include/smallk_amd.h -- C ABI of the MI355X-native dense NMF solver.
 *
 * This is the drop-in boundary for the smallk hot path: plain
 * pointers and sizes, no C++ or torch types.  Every entry point names the
 * reference interface it stands in for (paths relative to /root/reference).
 * The C++ facade (include/smallk.hpp, include/nmf.hpp) and the Python mirror of
 * pysmallk (smallk_amd/api.py) are both thin layers over these symbols.
 *
 * Conventions (same as the reference's inner seam, common/include/nmf.hpp:77-81):
 *   - host matrices are fp64, column-major, leading dimension >= height;
 *   - W (m x k) and H (k x n) are in/out: initial guess in, factors out;
 *   - return values are `Result` codes (nmf.hpp:17-26) plus two negative
 *     extensions for device errors; smk_last_error() gives the text;
 *   - not thread safe (the reference is not either: smallk.cpp:46-67).
 */
#ifndef SMALLK_AMD_H
#define SMALLK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum Result, common/include/nmf.hpp:17-26 */
enum {
    SMK_OK = 0,
    SMK_NOTINITIALIZED = -1,
    SMK_INITIALIZED = -2,
    SMK_BAD_PARAM = -3,
    SMK_FAILURE = -4,
    SMK_SIZE_TOO_LARGE = -5,
    SMK_FLATCLUST_FAILURE = -6,
    /* extensions (never produced by the reference) */
    SMK_DEVICE_ERROR = -100, /* HIP runtime error, see smk_last_error() */
    SMK_UNSUPPORTED = -101   /* valid for the reference, not built here (k > 2048) */
};

/* enum NmfAlgorithm, common/include/nmf.hpp:28-34 (NOT smallk::Algorithm's order) */
enum { SMK_ALG_MU = 0, SMK_ALG_HALS = 1, SMK_ALG_RANK2 = 2, SMK_ALG_BPP = 3 };

/* enum NmfProgressAlgorithm, common/include/nmf.hpp:37-41 */
enum { SMK_PROG_PG_RATIO = 0, SMK_PROG_DELTA_FNORM = 1 };

/* how A is held in HBM (device-side option; the host API stays fp64) */
enum { SMK_STORE_F32 = 0, SMK_STORE_BF16 = 1 };

/* struct NmfOptions, common/include/nmf.hpp:55-69 (bool -> int) */
typedef struct smk_options {
    double tol;
    int algorithm;
    int prog_est_algorithm;
    int height;
    int width;
    int k;
    int min_iter;
    int max_iter;
    int tolcount;
    int max_threads; /* accepted, unused on the device path */
    int verbose;
    int normalize;
} smk_options;

/* struct NmfStats, common/include/nmf.hpp:43-53 */
typedef struct smk_stats {
    unsigned long long elapsed_us;
    int iteration_count;
} smk_stats;

typedef struct smk_matrix smk_matrix; /* A (and A') resident in HBM, possibly a column shard */
typedef struct smk_solver smk_solver; /* one NmfSolve<> instance */

/* ---- lifecycle: NmfInitialize / NmfIsInitialized / NmfFinalize, nmf.hpp:71-73 ------------ */
int smk_initialize(int device_ordinal); /* -1: keep the current HIP device */
int smk_is_initialized(void);           /* SMK_INITIALIZED or SMK_NOTINITIALIZED */
void smk_finalize(void);
const char* smk_last_error(void);
int smk_device_cu_count(void);

/* IsValid(const NmfOptions&, bool validate_matrix), common/src/nmf_options.cpp:23-112 */
int smk_is_valid(const smk_options* opts, int validate_matrix);

/* Use an existing HIP stream (e.g. torch's current stream) instead of the library's own. */
int smk_set_stream(void* hip_stream);

/* ---- one shot: Result Nmf(const NmfOptions&, double* A, int ldA, double* W, int ldW,
 *      double* H, int ldH, NmfStats&), common/include/nmf.hpp:77-81 / src/nmf.cpp:173-229.
 *      `storage` chooses how A sits in HBM (the reference has no such knob). */
int smk_nmf_dense(const smk_options* opts, const double* A, int64_t ldA, double* W, int64_t ldW,
                  double* H, int64_t ldH, smk_stats* stats, int storage);

/* ---- one shot, sparse: Result NmfSparse(const NmfOptions&, height, width, nz, col_offsets, row_indices,
 *      data, W, ldW, H, ldH, NmfStats&), common/include/nmf.hpp:83-92 / src/nmf.cpp:232-300 */
int smk_nmf_sparse(const smk_options* opts, unsigned height, unsigned width, unsigned nz,
                   const unsigned* col_offsets, const unsigned* row_indices, const double* data,
                   double* W, int64_t ldW, double* H, int64_t ldH, smk_stats* stats);

/* ---- device-resident A (replaces the DenseMatrix view of buf_a, nmf.cpp:224) ---------------
 * A matrix object holds the column shard [col0, col0 + ncols_local) of a height x width_global
 * matrix, plus its transpose (the reference's BPP solver keeps At too, nmf_solver_bpp.hpp:319). */
int smk_matrix_create(smk_matrix** out, int64_t height, int64_t width_global, int64_t col0,
                      int64_t ncols_local, int storage);
/* The same without the stored transpose ("single copy"): the reference's MU and HALS call Gemm(NORMAL, TRANSPOSE)
 * on A itself (nmf_solver_mu.hpp:121-164, nmf_solver_hals.hpp:166-199) -- only its BPP keeps A' -- and so does this matrix: the
 * H*A' pass contracts down the strided direction of A (bf16: transposing LDS reads; fp32: strided 4-byte reads).  Half the footprint (twice the problem per GPU) and
 * no transpose pass at load time.  Solvers on it: MU, HALS and BPP with the 16-bit product forms read A only; the first solver that needs
 * the transpose (RANK2, the accurate form) makes the matrix allocate and fill it, after which it is an ordinary matrix
 * (smk_matrix_is_single_copy / smk_matrix_device_bytes tell).  SMK_SINGLE_COPY=1 makes smk_matrix_create do this for every dense matrix,
 * and smk_matrix_create does it by itself when A fits the device memory and A + A' do not. */
int smk_matrix_create_single_copy(smk_matrix** out, int64_t height, int64_t width_global, int64_t col0,
                                  int64_t ncols_local, int storage);
int smk_matrix_is_single_copy(const smk_matrix* a);
/* bytes of HBM the resident matrix occupies (A, its transpose when stored, or the CSC arrays) */
int64_t smk_matrix_device_bytes(const smk_matrix* a);
/* copy a host fp64 column-major block (height x ncols_local, the shard) into HBM */
int smk_matrix_upload_f64(smk_matrix* a, const double* host, int64_t ld);
/* fill with the counter-based uniform [0,1) generator (matrixgen UNIFORM, matrixgen/src/main.cpp:64-72);
 * element (r, c) depends only on (seed, global index), so shards agree with the whole. */
int smk_matrix_fill_uniform(smk_matrix* a, uint64_t seed);
/* structured synthetic data (the planted-low-rank variant): A = Ws Hs + noise U, rounded to the storage type, with
 * Ws (height x kstar) / Hs (kstar x width_global) the uniform matrices of seed + 1 / seed + 2 with entries <= threshold set to
 * zero and U the uniform matrix of `seed`; keyed by the global element index like smk_matrix_fill_uniform, fp64 sums in
 * increasing j (oracle twin: orc_fill_planted, same bits) */
int smk_matrix_fill_planted(smk_matrix* a, uint64_t seed, int kstar, double threshold, double noise);
/* read the shard back as fp64 (tests) */
int smk_matrix_download_f64(const smk_matrix* a, double* host, int64_t ld);
void smk_matrix_destroy(smk_matrix* a);
/* a copy of a resident matrix in the calling thread's context (current device): device-to-device, also across devices */
int smk_matrix_clone(const smk_matrix* src, smk_matrix** out);
/* a host thread with a device context of its own (stream, handles) on `device_ordinal`, separate from the process-wide
 * one, until smk_thread_context_end(); smk_device_count / smk_current_device: the HIP runtime's answers */
int smk_thread_context_begin(int device_ordinal);
void smk_thread_context_end(void);
int smk_device_count(void);
int smk_current_device(void);
int smk_device_synchronize(void);   /* hipDeviceSynchronize on the calling thread's current device */
/* The library keeps freed device workspaces of up to 512 MB each (at most 4 GB per device, SMK_DEVMEM_CACHE_MB) for reuse; they are
 * invisible to other allocators in the process (torch, RCCL).  smk_device_trim returns those of the current device to the HIP
 * runtime (call it before creating communicators or large torch tensors); the return value is the number of bytes that were cached. */
size_t smk_device_trim(void);
/* sparse A in CSC (replaces SparseMatrix<double>, common/include/sparse_matrix_decl.hpp:21-132): the local
 * columns [col0, col0+ncols_local); 32-bit indices as in the reference, duplicates allowed (they add up).
 * The transpose is built here too (the reference does it in Solver_Generic_BPP::Init, nmf_solver_bpp.hpp:319). */
int smk_matrix_create_sparse(smk_matrix** out, int64_t height, int64_t width_global, int64_t col0,
                             int64_t ncols_local, int64_t nnz_local, const unsigned* col_offsets,
                             const unsigned* row_indices, const double* data);
/* Host-side CSC bookkeeping behind the sparse matrix objects (no device needed).
 * smk_csc_transpose: Transpose(SparseMatrix), common/include/sparse_matrix_ops.hpp:36-127 (counting sort).
 * smk_csc_subset_cols_compact: SparseMatrix::SubMatrixColsCompact, common/include/sparse_matrix_impl.hpp:478-592;
 *   call with out_col_offsets == NULL for the sizes, then with arrays of *out_nnz / ncols+1 entries;
 *   old_to_new (height entries, 0xFFFFFFFF = dropped row) and new_to_old (*new_height entries) may be NULL.
 * smk_matrix_download_csc: the resident CSC (transposed != 0: the CSC of A') back on the host (tests). */
int smk_csc_transpose(int64_t height, int64_t width, const unsigned* col_offsets, const unsigned* row_indices,
                      const double* data, unsigned* out_col_offsets, unsigned* out_row_indices, double* out_data);
int smk_csc_subset_cols_compact(int64_t height, int64_t width, const unsigned* col_offsets, const unsigned* row_indices,
                                const double* data, const unsigned* cols, int64_t ncols, unsigned* out_col_offsets,
                                unsigned* out_row_indices, double* out_data, unsigned* old_to_new, unsigned* new_to_old,
                                int64_t* new_height, int64_t* out_nnz);
int smk_matrix_download_csc(const smk_matrix* a, int transposed, unsigned* col_offsets, unsigned* row_indices,
                            double* data);
/* The reference's sparse Gemm by itself (common/include/sparse_gemm_ab_impl.hpp:24-100, :480-582; sparse_gemm_ba_impl.hpp:25-99),
 * in the gather form the solver uses: out (k x ncols(B), ld ldo) = X (k x rows(B), ld ldx) * B, B = A (transposed == 0, i.e. W'A
 * from X = W') or B = A' (transposed != 0, i.e. (AH')' from X = H), on the kernel the solver takes at rank k.  reps > 0 and
 * avg_ms != NULL: that many more launches timed with HIP events. */
int smk_matrix_sparse_product(const smk_matrix* a, int transposed, int k, const double* X, int64_t ldx, double* out,
                              int64_t ldo, int reps, double* avg_ms);
int64_t smk_matrix_nnz(const smk_matrix* a);
int64_t smk_matrix_height(const smk_matrix* a);
/* same generator on the host, for W0/H0 (RandomMatrix stand-in, smallk.cpp:533,554) */
void smk_uniform_fill_host(double* buf, int64_t ld, int64_t rows, int64_t cols, int64_t r0, int64_t c0,
                           int64_t global_height, uint64_t seed, int quant /* 0: 24 bit, 1: bf16 */);

/* ---- solver object = NmfSolve<> (common/include/nmf_solve_generic.hpp:34-140) ---------------- */
int smk_solver_create(smk_solver** out, const smk_options* opts, const smk_matrix* a);
void smk_solver_destroy(smk_solver* s);
/* W0 (m x k) and the local H0 shard (k x ncols_local); runs solver.Init + progress_est->Init (:62-63) */
int smk_solver_set_factors(smk_solver* s, const double* W0, int64_t ldW, const double* H0, int64_t ldH);
/* the same with W0 / H0 = the counter-based uniform matrices of smk_uniform_fill_host(seed_w) / (seed_h), generated on the device
 * (no host fill, no upload); unsharded solvers */
int smk_solver_set_factors_uniform(smk_solver* s, uint64_t seed_w, uint64_t seed_h);
/* the whole driver loop with stopping rule, final NormalizeAndScale, stats (:67-139) */
int smk_solver_run(smk_solver* s, smk_stats* stats);
/* enqueue `iters` solver iterations without convergence checks (the min_iter branch, :81-95);
 * asynchronous: returns before the GPU finishes. */
int smk_solver_iterate(smk_solver* s, int iters);
/* the same with the stopping rule's metric formed and read back after EVERY iteration, as NmfSolve<> does past min_iter
 * (common/include/nmf_solve_generic.hpp:98-121; gradients every iteration: nmf_solver_mu.hpp:151-164, nmf_solver_bpp.hpp:370-377),
 * never stopping: the per-iteration cost of the reference's default run.  MU / HALS / BPP.  last_metric may be NULL. */
int smk_solver_iterate_checked(smk_solver* s, int iters, double* last_metric);
int smk_solver_sync(smk_solver* s); /* wait + report solver failures (Result code) */
/* progress metric of the last iteration that computed one (PG ratio or delta-Fnorm) */
int smk_solver_progress(smk_solver* s, double* metric);
/* optional final NormalizeAndScale (normalize.hpp:118-140), then copy factors to the host */
int smk_solver_get_factors(smk_solver* s, int normalize, double* W, int64_t ldW, double* H, int64_t ldH);
int smk_solver_iteration_count(const smk_solver* s);
/* the product form in use (SMK_NSPLIT numbering: 3 = bf16x3, 4 = fp16 two-term, 8 = the accurate fp64 form) and what the
 * opt-in run-time guard (BPP, SMK_GUARD_EVERY=n) has done so far: every n iterations it compares the fast form with the
 * accurate one on a column sample and changes to the accurate form when cond(Gram) x (product discrepancy) says one
 * iteration could move the factors by 1e-4 */
int smk_solver_product_form(const smk_solver* s, int* guard_checks, int* guard_fired, double* guard_last);

/* bool NnlsBlockpivot(LHS, RHS, X, Y), common/include/nnls.hpp:144-244, by itself (the reference's
 * tests/src/test_bpp.cpp drives the solver this way): LHS k x k SPD, RHS k x ncols, X in/out (warm start:
 * passive set = X > 0), Y = LHS X - RHS out (may be NULL).  SMK_FAILURE = the reference's `false`
 * (pivot limit 5k reached, or a passive block that is not positive definite). */
int smk_nnls_blockpivot(int k, int64_t ncols, const double* LHS, int64_t ldL, const double* RHS, int64_t ldR,
                        double* X, int64_t ldX, double* Y, int64_t ldY);

/* measurement: HIP events around the streaming-product launches (stream of the solver).  A pair of event records costs
 * ~11 us of idle time around the launch, so passes over less than 1 GB are timed one launch in 16 and the totals scaled
 * back up (SMK_TIMING_STRIDE overrides) */
int smk_solver_enable_timing(smk_solver* s, int on);
/* which: 0 = W'A pass, 1 = H*At pass (a pass that the multi-GPU schedule cuts into chunks counts as ONE launch per group of
 * 64 factor rows; its time is the sum of its chunk launches), 2 = the per-chunk collectives of a sharded run, timed on the
 * collective stream (the sums of (AH')' and, BPP, the all-gathers of the packed W); 3 = the time the MAIN stream stood waiting
 * for events of the collective stream (each wait bracketed by two events on the main stream: the measured, not inferred, exposed
 * part of the exchange); 4 = the same bracket around a wait for an event that completed long ago (what a bracket costs by itself,
 * ~15 us: subtract brackets x its average from slot 3); 5 = the block-pivoting (NNLS) launches of BPP, sampled with the passes.
 * Total ms and count since enable. */
int smk_solver_kernel_time(smk_solver* s, int which, double* total_ms, int* launches);
/* name of the kernel pass `which` (0 = W'A, 1 = H*At) launches for this solver -- the streaming product and its variant, or which of the
 * sparse gather products the plan chose (spmm_seg_kernel on ragged columns, spmm_gather_kernel on fixed-degree graphs, ...): what
 * bench.py attributes `roofline.achieved` to.  which = 2: how the stopping-rule checks of this solver were formed so far, with counts
 * (launches of their own, or riding in the next iteration's NNLS launch and pass: DESIGN.md 6) */
int smk_solver_kernel_name(const smk_solver* s, int which, char* out, int cap);
/* diagnostics of the block-pivoting kernels (csrc/nnls.hip), live only in a process started with SMK_NNLS_STATS=1 (else
 * SMK_UNSUPPORTED): 256 counters -- [0..15] exchanges per column (nnls.hpp:192-241 trips), [16..80] size of a column's first
 * compact solve, [96..160] of its later ones, [176] / [177] solves in the complement / direct form, [178] columns, [179] / [180]
 * solves with all / no variables passive.  Synchronises the device.  reset != 0 clears the counters after the read. */
int smk_debug_nnls_stats(unsigned long long* out256, int reset);
/* algorithmic bytes / flops one launch of pass `which` moves (len*ncols*sizeof(elt), 2*k*len*ncols) */
int smk_solver_kernel_work(const smk_solver* s, int which, double* bytes, double* flops);

/* ---- multi-GPU: A and H column sharded, W replicated in the algorithm.  Exchange steps per iteration:
 *   - sum-all-reduce of HH' (k x k, fp64), beside the H*At pass;
 *   - the sum of H*At = (AH')' (k x m; fp64 on the wire, SMK_COMM_F64=0: fp32): the pass runs in row chunks and the
 *     sum of chunk j travels while the product streams chunk j + 1 -- an all-reduce for MU / HALS (replicated W
 *     update), a reduce-scatter for BPP, whose W rows are independent NNLS problems: block r of every chunk belongs
 *     to rank r (block-cyclic), every rank solves its own blocks;
 *   - BPP: sum-all-reduce of W'W (k x k) built from the own blocks, and an all-gather per chunk of the PACKED
 *     streaming operand of those blocks (4 B per entry in the fp16 form; the fp64 rows are gathered only when results
 *     or the DELTA_FNORM rule need them); the W'A pass then runs chunk by chunk down the rows as the operand lands;
 *   - one 3-element all-reduce when the stopping rule is evaluated.
 * All of them are issued from C on ONE second HIP stream per solver (a communicator is never driven from two streams)
 * and tied to the main stream by events.  Communicators:
 *   - RCCL over xGMI: one process per GPU (smk_comm_unique_id on rank 0, broadcast the 128 bytes by any means,
 *     smk_comm_init_rank everywhere) or one process driving several GPUs (smk_comm_init_all, one host thread per
 *     device -- what smk_nmf_dense_sharded does);
 *   - an in-process stand-in with the same semantics for several shards on ONE device (smk_comm_init_local: RCCL
 *     refuses two ranks on a device), used by the tests and by boxes with fewer GPUs than shards.
 * Environment: SMK_COMM_CHUNKS=1..8 (row chunks of the exchange; default: blocks of >= 4096 rows, at most 4 chunks),
 * SMK_COMM_F64=0 (fp32 on the wire), SMK_COMM_FORCE=1 (a world of ONE rank issues every collective too: tests of the real nccl* calls).
 * The reference has no distributed mode (sphinx/source/pages_installation.rst:38). */
typedef struct smk_comm smk_comm;
int smk_comm_unique_id(void* id128 /* 128 bytes out */);
int smk_comm_init_rank(smk_comm** out, const void* id128, int rank, int world);   /* on the CURRENT HIP device */
int smk_comm_init_all(smk_comm** out /* ndev handles */, int ndev, const int* devices /* NULL: 0..ndev-1 */);
int smk_comm_init_local(smk_comm** out /* nranks handles */, int nranks);
/* every rank calls it: known values through an all-reduce (fp64, fp32), an all-gather and a reduce-scatter, checked on the host */
int smk_comm_selftest(smk_comm* c);
int smk_comm_rank(const smk_comm* c);
int smk_comm_world(const smk_comm* c);
void smk_comm_destroy(smk_comm* c);
/* a rank that gives up calls this so that peers blocked in a collective are released (ncclCommAbort / the stand-in's
 * failure flag); the handle is still destroyed with smk_comm_destroy */
void smk_comm_abort(smk_comm* c);
/* attach before smk_solver_set_factors(); the communicator must outlive the solver */
int smk_solver_attach_comm(smk_solver* s, smk_comm* comm);
/* Result Nmf(...) (common/src/nmf.cpp:173-229) on `nshards` column shards, one host thread and one device per shard;
 * devices NULL: shard r on device r; local_stub != 0: all shards on the current device through the stand-in. */
int smk_nmf_dense_sharded(const smk_options* opts, const double* A, int64_t ldA, double* W, int64_t ldW, double* H,
                          int64_t ldH, smk_stats* stats, int storage, int nshards, const int* devices, int local_stub);

/* Test hook kept from round 1: the host supplies the all-reduce (e.g. torch.distributed on gloo);
 * `ptr` is device memory inside the registered workspace. */
typedef int (*smk_allreduce_fn)(void* user, void* ptr, int64_t count, int dtype /*0 f32, 1 f64*/);
int smk_solver_comm_workspace_bytes(const smk_solver* s, size_t* bytes);
int smk_solver_set_comm(smk_solver* s, int rank, int world, smk_allreduce_fn fn, void* user,
                        void* workspace, size_t workspace_bytes);

/* ---- HierNMF2: rank-2 hierarchical clustering --------------------------------------
 * Clust / ClustSparse (hierclust/include/clust.hpp:45-58, hierclust/src/clust.cpp:97-203) with
 * ClustOptions (clust.hpp:27-37), ClustStats (:20-25) and the result Tree<T> (hierclust/include/tree.hpp).
 * Every node factorisation is the RANK2 solver above on a column subset of the resident A
 * (SubMatrixColsCompact). */
typedef struct smk_clust_options {
    smk_options nmf;       /* height/width = A's; k, algorithm are forced to 2 / RANK2 */
    int maxterms;
    double unbalanced;     /* [0, 1) */
    int trial_allowance;
    int num_clusters;      /* >= 2 */
    int verbose;
    int flat;              /* also run ClustFlat (clust_flat_generic.hpp:33-74) on the leaves */
} smk_clust_options;
typedef struct smk_clust_stats { int nmf_count; int max_count; } smk_clust_stats;
typedef struct smk_tree smk_tree;
typedef struct smk_tree_node {
    double priority;
    unsigned parent, left_child, right_child;   /* SMK_TREE_NONE when absent */
    int is_valid, is_left_child, is_leaf;
    int64_t doc_count;
} smk_tree_node;
#define SMK_TREE_NONE 0xFFFFFFFFu

int smk_clust_is_valid(const smk_clust_options* opts, int validate_matrix);   /* clust_options.cpp:16-110 */
/* column subset of a resident matrix (dense: all rows kept; sparse: unused rows dropped, the kept
 * rows are listed in new_to_old_rows[0 .. *new_height), capacity = height). */
int smk_matrix_gather_cols(const smk_matrix* src, const unsigned* cols, int64_t ncols, smk_matrix** out,
                           unsigned* new_to_old_rows, int64_t* new_height);
/* Returns SMK_OK, or SMK_FLATCLUST_FAILURE with *tree still valid (opts.flat and the flat step failed:
 * fewer leaves than clusters, or NnlsHals did not converge), or an error with *tree == NULL.
 * Random initialisers: matrix i of the run is the counter-based uniform block with seed
 * `seed + 0x9E37 * (++*draws)` (W then H per attempt); `initdir` non-empty: Winit_<i>.csv / Hinit_<i>.csv
 * (full size, i = 1, 2, ...; clust_hier_util.hpp:206-241) are used instead.  `draws` may be NULL. */
int smk_clust_dense(const smk_clust_options* opts, const double* A, int64_t ldA, int storage, uint64_t seed,
                    uint64_t* draws, const char* initdir, smk_tree** tree, smk_clust_stats* stats);
int smk_clust_sparse(const smk_clust_options* opts, int64_t nnz, const unsigned* col_offsets,
                     const unsigned* row_indices, const double* data, uint64_t seed, uint64_t* draws,
                     const char* initdir, smk_tree** tree, smk_clust_stats* stats);
/* on a matrix that already lives in HBM (smk_matrix_create[_sparse]); options height/width must match it */
int smk_clust_resident(const smk_clust_options* opts, const smk_matrix* a, uint64_t seed, uint64_t* draws,
                       const char* initdir, smk_tree** tree, smk_clust_stats* stats);
void smk_tree_destroy(smk_tree* t);
int smk_tree_node_count(const smk_tree* t);
int64_t smk_tree_term_count(const smk_tree* t);
int64_t smk_tree_doc_count(const smk_tree* t);
int smk_tree_get_node(const smk_tree* t, int q, smk_tree_node* out);
int smk_tree_node_docs(const smk_tree* t, int q, unsigned* out /* doc_count */);
int smk_tree_node_topic(const smk_tree* t, int q, double* out /* term_count */);
int smk_tree_node_terms(const smk_tree* t, int q, int* out /* maxterms */);   /* returns entries written */
int64_t smk_tree_assignments(const smk_tree* t, unsigned* out /* doc_count; SMK_TREE_NONE = outlier */);
int64_t smk_tree_outliers(const smk_tree* t, unsigned* out /* NULL: count only */);
int smk_tree_write_assignments(const smk_tree* t, const char* path);            /* tree.hpp:388-423 */
/* format 0 = XML, 1 = JSON (hierclust_xml_writer.cpp / hierclust_json_writer.cpp, byte compatible) */
int smk_tree_write(const smk_tree* t, const char* path, int format, const char* const* dictionary,
                   int64_t dictionary_size);
/* opts.flat: the flat factors (W m x num_clusters, H num_clusters x n) computed after the tree search */
int smk_tree_flat_factors(const smk_tree* t, double* W, int64_t ldW, double* H, int64_t ldH);
/* the priority score of a split: compute_priority(), clust_hier_util.hpp:105-173 */
double smk_clust_priority(const double* w_parent, const double* w_child /* n x 2, ld n */, int64_t n);

/* ---- flat clustering ------------------------------------------------------------------
 * FlatClust / FlatClustSparse (flatclust/include/flat_clust.hpp, flatclust/src/flat_clust.cpp:118-264):
 * NmfSolve<> restricted to HALS / RANK2 / BPP; same argument meaning as smk_nmf_dense / smk_nmf_sparse. */
int smk_flatclust_dense(const smk_options* opts, const double* A, int64_t ldA, double* W, int64_t ldW, double* H,
                        int64_t ldH, smk_stats* stats, int storage);
int smk_flatclust_sparse(const smk_options* opts, unsigned height, unsigned width, unsigned nz,
                         const unsigned* col_offsets, const unsigned* row_indices, const double* data, double* W,
                         int64_t ldW, double* H, int64_t ldH, smk_stats* stats);
/* NnlsHals (common/include/nnls.hpp:249-316): H-only HALS sweeps with W fixed, until
 * PG(H) < tol * PG(H after sweep 1); normalises W, H on success.  SMK_FAILURE at the iteration limit. */
int smk_solver_nnls_hals(smk_solver* s, double tol, int verbose, int max_iter, int* iterations);
/* assignments.hpp:32-113, terms.hpp:62-108 (host post-processing of the returned factors) */
int smk_compute_assignments(const double* H, unsigned ldH, unsigned k, unsigned n, unsigned* labels /* n */);
int smk_compute_fuzzy_assignments(const double* H, unsigned ldH, unsigned k, unsigned n, float* probabilities);
int smk_top_terms(int maxterms, const double* W, unsigned ldim, unsigned height, unsigned width,
                  int* term_indices /* maxterms * width */);
/* common/src/assignments.cpp:23-70; return 1 on success like the reference's bool */
int smk_write_assignments_file(const unsigned* labels, unsigned n, const char* path);
int smk_write_fuzzy_assignments_file(const float* probabilities, unsigned k, unsigned n, const char* path);
/* FlatClustWriteResults, common/src/flat_clust_output.cpp:56-141; format 0 = XML, 1 = JSON */
int smk_flatclust_write_results(const char* assignfile, const char* fuzzyfile, const char* resultfile,
                                const unsigned* assignments, unsigned num_assignments, const float* probabilities,
                                const char* const* dictionary, int64_t dictionary_size, const int* term_indices,
                                int64_t num_term_indices, int format, unsigned maxterms, unsigned num_docs,
                                unsigned num_clusters);

/* ---- CSV files: WriteDelimitedFile / LoadDelimitedFile, common/include/delimited_file.hpp:49-135 ----
 * (row-major text, scientific notation; used for w.csv / h.csv and init files) */
int smk_write_csv(const double* buf, unsigned ldim, unsigned height, unsigned width, const char* filename,
                  unsigned precision);
int smk_load_csv(const char* filename, double* out, unsigned long capacity, unsigned* height, unsigned* width);
/* MatrixMarket coordinate files -> CSC (LoadMatrixMarketFile, common/include/sparse_matrix_io.hpp:118-262:
 * real/integer/pattern x general/symmetric/skew-symmetric).  Two calls: sizes first (arrays NULL), then data.
 * Returns 1 on success, 0 on failure. */
int smk_load_matrix_market(const char* filename, unsigned* height, unsigned* width, unsigned* nnz,
                           unsigned* col_offsets /* width+1 or NULL */, unsigned* row_indices /* nnz or NULL */,
                           double* data /* nnz or NULL */);

/* ---- flat handles onto the public C++ API `namespace smallk` (include/smallk.hpp), one per entry of
 * pysmallk's extern block (pysmallk/interface/smallk_lib.pyx:42-88), for bindings that cannot call C++
 * (ctypes, cgo, JNI).  Functions that can throw return 0 = ok, 1 = std::logic_error,
 * 2 = std::runtime_error; the message is in smk_api_last_exception(). */
const char* smk_api_last_exception(void);
int smk_api_initialize(void);                /* smallk::Initialize   smallk.cpp:114-119 */
int smk_api_is_initialized(void);            /* smallk::IsInitialized */
void smk_api_finalize(void);                 /* smallk::Finalize */
void smk_api_reset(void);                    /* smallk::Reset        smallk.cpp:81-111 */
void smk_api_seed_rng(int seed);             /* smallk::SeedRNG */
unsigned smk_api_get_major_version(void);
unsigned smk_api_get_minor_version(void);
unsigned smk_api_get_patch_level(void);
int smk_api_load_matrix_file(const char* path);                                 /* LoadMatrix(string) :163 */
int smk_api_load_matrix_dense(const double* buf, unsigned ldim, unsigned height, unsigned width); /* :204 */
int smk_api_load_matrix_sparse(unsigned height, unsigned width, unsigned nz, const double* data,
                               const unsigned* row_indices, const unsigned* col_offsets);       /* :268 */
int smk_api_is_matrix_loaded(void);
int smk_api_set_output_dir(const char* dir);  /* SetOutputDir :348-380 */
const char* smk_api_get_output_dir(void);
void smk_api_set_output_precision(unsigned digits);
unsigned smk_api_get_output_precision(void);
int smk_api_set_nmf_tolerance(double tol);
double smk_api_get_nmf_tolerance(void);
void smk_api_set_max_iter(unsigned v);
unsigned smk_api_get_max_iter(void);
void smk_api_set_min_iter(unsigned v);
unsigned smk_api_get_min_iter(void);
void smk_api_set_max_threads(unsigned v);
unsigned smk_api_get_max_threads(void);
void smk_api_set_max_terms(unsigned v);
unsigned smk_api_get_max_terms(void);
void smk_api_set_output_format(int xml0_json1);
int smk_api_get_output_format(void);
int smk_api_set_hiernmf2_tolerance(double tol);
double smk_api_get_hiernmf2_tolerance(void);
void smk_api_set_device_storage(int storage);   /* extension: SMK_STORE_F32 / SMK_STORE_BF16 */
int smk_api_get_device_storage(void);
unsigned smk_api_get_iteration_count(void);
/* smallk::Nmf(k, algorithm, initfile_w, initfile_h), smallk.cpp:471-650; `algorithm` is numbered
 * like smallk::Algorithm: MU 0, BPP 1, HALS 2, RANK2 3 (smallk.hpp:34-40) */
int smk_api_nmf(unsigned k, int algorithm, const char* initfile_w, const char* initfile_h);
const double* smk_api_locked_buffer_w(unsigned* ldim, unsigned* height, unsigned* width);   /* :653-661 */
const double* smk_api_locked_buffer_h(unsigned* ldim, unsigned* height, unsigned* width);   /* :664-672 */
int smk_api_hiernmf2(unsigned num_clusters);            /* smallk::HierNmf2, smallk.cpp:859-862 */
int smk_api_hiernmf2_with_flat(unsigned num_clusters);  /* smallk::HierNmf2WithFlat, smallk.cpp:865-868 */
int smk_api_load_dictionary_file(const char* path);     /* LoadDictionary(string), smallk.cpp:675-691 */
int smk_api_load_dictionary(const char* const* terms, unsigned count);          /* smallk.cpp:694-707 */

#ifdef __cplusplus
}
#endif
#endif /* SMALLK_AMD_H */
