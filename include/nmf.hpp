// include/nmf.hpp -- the inner (POD) seam of the reference NMF library, served by the
// MI355X solver.  Declarations mirror /root/reference/common/include/nmf.hpp:17-92 so that
// existing callers (nmf CLI nmf/src/main.cpp:218-233, smallk::Nmf smallk/src/smallk.cpp:604-619,
// tests/src/test_dense_nmf.cpp:302-317) compile unchanged; the bodies live in
// smallk_amd/csrc/facade.cpp on top of the C ABI (include/smallk_amd.h).
#pragma once

enum Result
{
    OK                =  0,
    NOTINITIALIZED    = -1,
    INITIALIZED       = -2,
    BAD_PARAM         = -3,
    FAILURE           = -4,
    SIZE_TOO_LARGE    = -5,
    FLATCLUST_FAILURE = -6
};

enum NmfAlgorithm
{
    MU,     // multiplicative updating (Lee & Seung)
    HALS,   // hierarchical alternating least squares (Cichocki & Pan)
    RANK2,  // rank-2 specialisation (Kuang & Park): closed-form 2x2 solves, k is forced to 2
    BPP     // block principal pivoting (Kim & Park)
};

enum NmfProgressAlgorithm
{
    PG_RATIO,     // ratio of projected-gradient norms
    DELTA_FNORM   // relative change of ||W||_F
};

struct NmfStats
{
    NmfStats() : elapsed_us(0u), iteration_count(0) {}
    unsigned long long elapsed_us;
    int iteration_count;
};

struct NmfOptions
{
    double tol;
    NmfAlgorithm algorithm;
    NmfProgressAlgorithm prog_est_algorithm;
    int height;
    int width;
    int k;
    int min_iter;
    int max_iter;
    int tolcount;
    int max_threads;
    bool verbose;
    bool normalize;
};

// Select the GPU and create the HIP stream (stands in for Elemental/MPI initialisation).
void NmfInitialize(int argc, char* argv[]);
Result NmfIsInitialized();
void NmfFinalize();

bool IsValid(const NmfOptions& opts, bool validate_matrix = true);

// Dense NMF.  Host buffers are fp64 column-major; W and H are in/out.
// Throws std::logic_error when a leading dimension is too small (reference: nmf.cpp:213-219).
Result Nmf(const NmfOptions& options,
           double* buf_A, int ldim_A,
           double* buf_W, int ldim_W,
           double* buf_H, int ldim_H,
           NmfStats& stats);

// Sparse input (CSC, 32-bit indices): same algorithms, the products with A become gathers.
Result NmfSparse(const NmfOptions& options,
                 const unsigned int height,
                 const unsigned int width,
                 const unsigned int nz,
                 const unsigned int* col_offsets,
                 const unsigned int* row_indices,
                 const double* data,
                 double* buf_W, int ldim_W,
                 double* buf_H, int ldim_H,
                 NmfStats& stats);

// ---- MI355X extension (not in the reference): how A is held in HBM for Nmf() ----------
// 0 = fp32 (default), 1 = bf16.  The host API stays fp64 either way.
void NmfSetDeviceStorage(int storage);
int NmfGetDeviceStorage();
