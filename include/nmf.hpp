// include/nmf.hpp -- the inner (POD) seam of the reference NMF library, served by the MI355X solver.
//
// Same names, values, member order and signatures as the reference seam
// (/root/reference/common/include/nmf.hpp:17-92), so that its callers -- the nmf command line tool
// (nmf/src/main.cpp:218-233), smallk::Nmf (smallk/src/smallk.cpp:604-619), the dense NMF tests
// (tests/src/test_dense_nmf.cpp:302-317) -- compile against this file unchanged.  The bodies are in
// smallk_amd/csrc/facade.cpp, on top of the C ABI declared in include/smallk_amd.h.
#pragma once

// Return codes of every entry point of the seam (SMK_* in smallk_amd.h carry the same values).
enum Result { OK = 0, NOTINITIALIZED = -1, INITIALIZED = -2, BAD_PARAM = -3, FAILURE = -4, SIZE_TOO_LARGE = -5,
              FLATCLUST_FAILURE = -6 };

// Solvers.  NOTE: smallk::Algorithm (smallk.hpp) numbers them differently.
//   MU    multiplicative updating                       HALS  hierarchical alternating least squares
//   RANK2 closed-form rank-2 solves, k is forced to 2   BPP   NNLS by block principal pivoting
enum NmfAlgorithm { MU, HALS, RANK2, BPP };

// Stopping rules: ratio of projected-gradient norms (to iteration 1) / relative change of ||W||_F.
enum NmfProgressAlgorithm { PG_RATIO, DELTA_FNORM };

struct NmfStats
{
    unsigned long long elapsed_us;   // wall clock of the solve
    int iteration_count;             // index of the iteration that stopped the loop (max_iter if none did)
    NmfStats() : elapsed_us(0u), iteration_count(0) {}
};

struct NmfOptions
{
    double tol;                                   // stopping tolerance, in (0, 1)
    NmfAlgorithm algorithm;
    NmfProgressAlgorithm prog_est_algorithm;
    int height, width, k;                         // A is height x width, W height x k, H k x width
    int min_iter, max_iter, tolcount;             // the rule must hold `tolcount` times in a row
    int max_threads;                              // accepted; host side only
    bool verbose, normalize;                      // normalize: unit-norm columns of W, H rescaled
};

// Lifecycle.  NmfInitialize selects the GPU and creates the HIP stream (the reference starts
// Elemental/MPI here).
void NmfInitialize(int argc, char* argv[]);
Result NmfIsInitialized();
void NmfFinalize();

// Option validation (k > 0, k <= width, 0 < tol < 1, positive iteration counts, RANK2 => k == 2).
bool IsValid(const NmfOptions& opts, bool validate_matrix = true);

// Dense A (fp64 column-major host buffer, leading dimension ldim_A).  W and H are in/out: initial
// guess in, factors out.  Throws std::logic_error when ldim_W < height or ldim_H < k (nmf.cpp:213-219).
Result Nmf(const NmfOptions& options, double* buf_A, int ldim_A, double* buf_W, int ldim_W, double* buf_H, int ldim_H,
           NmfStats& stats);

// Sparse A in CSC with 32-bit indices (col_offsets has width + 1 entries).  Same solvers; the two
// products with A become gathers over the stored entries.
Result NmfSparse(const NmfOptions& options, const unsigned int height, const unsigned int width, const unsigned int nz,
                 const unsigned int* col_offsets, const unsigned int* row_indices, const double* data, double* buf_W,
                 int ldim_W, double* buf_H, int ldim_H, NmfStats& stats);

// MI355X extension (new names only): how a dense A is held in HBM by Nmf(): 0 = fp32 (default), 1 = bf16.
void NmfSetDeviceStorage(int storage);
int NmfGetDeviceStorage();
