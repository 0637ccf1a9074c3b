// include/smallk.hpp -- public libsmallk API (v1.6.2 surface) served by the MI355X solver.
//
// The declarations reproduce the interface of /root/reference/smallk/include/smallk.hpp:29-336
// (same names, argument meaning, defaults and exception behaviour) so that a program written
// against libsmallk -- e.g. examples/smallk_example.cpp or pysmallk's cdef extern block
// (pysmallk/interface/smallk_lib.pyx:42-88) -- links against libsmallk_amd.so instead.
// NMF (MU / HALS / BPP / RANK2, dense or sparse A) and the clustering entry points (LoadDictionary,
// HierNmf2, HierNmf2WithFlat) all run on the GPU through include/smallk_amd.h; SMK_NUM_GPUS=N shards a
// dense Nmf() over N devices of the node.
#pragma once

#include <string>
#include <vector>

#define SMALLK_MAJOR_VERSION 1
#define SMALLK_MINOR_VERSION 6
#define SMALLK_PATCH_LEVEL   2

namespace smallk
{
    enum Algorithm
    {
        MU,     // multiplicative updating
        BPP,    // block principal pivoting
        HALS,   // hierarchical alternating least squares
        RANK2   // rank-2 specialisation
    };

    enum OutputFormat
    {
        XML,
        JSON
    };

    // -- lifecycle ---------------------------------------------------------------------------
    void Initialize(int& argc, char**& argv);   // must precede every other call
    bool IsInitialized();
    void Finalize();

    // -- version -----------------------------------------------------------------------------
    unsigned int GetMajorVersion();
    unsigned int GetMinorVersion();
    unsigned int GetPatchLevel();
    std::string GetVersionString();

    // -- parameters (defaults: precision 6, tol 0.005, maxiter 5000, miniter 5) ----------------
    unsigned int GetOutputPrecision();
    void SetOutputPrecision(const unsigned int num_digits = 6);
    double GetNmfTolerance();
    void SetNmfTolerance(const double tol = 0.005);          // logic_error unless 0 < tol < 1
    unsigned int GetMaxIter();
    void SetMaxIter(const unsigned int max_iterations = 5000);
    unsigned int GetMinIter();
    void SetMinIter(const unsigned int min_iterations = 5);
    unsigned int GetMaxThreads();
    void SetMaxThreads(const unsigned int max_threads);
    void Reset();                                            // restore defaults, drop the matrix
    void SeedRNG(const int seed);

    // -- input matrix ------------------------------------------------------------------------
    void LoadMatrix(const std::string& filepath);            // .csv dense, .mtx MatrixMarket sparse
    void LoadMatrix(const double* buffer,                    // dense, column-major
                    const unsigned int ldim,
                    const unsigned int height,
                    const unsigned int width);
    void LoadMatrix(const unsigned int height,               // sparse CSC
                    const unsigned int width,
                    const unsigned int nz,
                    const std::vector<double>& data,
                    const std::vector<unsigned int>& row_indices,
                    const std::vector<unsigned int>& col_offsets);
    bool IsMatrixLoaded();

    std::string GetOutputDir();
    void SetOutputDir(const std::string& outdir);            // logic_error if it does not exist

    // -- factorisation: A ~ W H; writes w.csv / h.csv into the output dir --------------------
    void Nmf(const unsigned int k,
             const Algorithm algorithm = BPP,
             const std::string& initfile_w = std::string(""),
             const std::string& initfile_h = std::string(""));

    // Factors of the last Nmf() call; valid until the next Nmf / Reset / LoadMatrix.
    const double* LockedBufferW(unsigned int& ldim, unsigned int& height, unsigned int& width);
    const double* LockedBufferH(unsigned int& ldim, unsigned int& height, unsigned int& width);

    // -- clustering: HierNMF2 (+ optional flat clustering of the leaves); writes tree_N.{xml,json},
    //    assignments_N.csv (and assignments_flat_N.csv, assignments_fuzzy_N.csv, clusters_N.*) --------
    void LoadDictionary(const std::string& filepath);
    void LoadDictionary(const std::vector<std::string>& terms);
    unsigned int GetMaxTerms();
    void SetMaxTerms(const unsigned int max_terms = 5);
    OutputFormat GetOutputFormat();
    void SetOutputFormat(const OutputFormat format = JSON);
    double GetHierNmf2Tolerance();
    void SetHierNmf2Tolerance(const double tol = 0.0001);
    void HierNmf2(const unsigned int num_clusters);
    void HierNmf2WithFlat(const unsigned int num_clusters);

    // -- MI355X extensions (new names only; nothing above changes) ----------------------------
    enum DeviceStorage { DEVICE_F32 = 0, DEVICE_BF16 = 1 };
    void SetDeviceStorage(const DeviceStorage storage);      // how A is held in HBM (default F32)
    DeviceStorage GetDeviceStorage();
    unsigned int GetIterationCount();                        // NmfStats.iteration_count of the last Nmf()
    unsigned long long GetElapsedMicroseconds();             // NmfStats.elapsed_us of the last Nmf()

} // namespace smallk
