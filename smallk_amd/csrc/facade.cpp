// smallk_amd/csrc/facade.cpp -- C++ host layer above the C ABI:
//   * the inner seam  ::Nmf(NmfOptions, ...)           (reference common/src/nmf.cpp:173-229)
//   * the public API  namespace smallk                 (reference smallk/src/smallk.cpp:81-672)
//   * CSV reader/writer used by LoadMatrix / init files / w.csv, h.csv
//                                                       (reference common/include/delimited_file.hpp:49-135)
// Same names, argument meaning, clamping and exception types as the reference; the numeric
// work is delegated to the GPU solver through include/smallk_amd.h.
#include "../../include/nmf.hpp"
#include "../../include/smallk.hpp"
#include "../../include/smallk_amd.h"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <thread>

// =============================================================================================
// CSV (delimited) files
// =============================================================================================
namespace smallk_amd_io {

// Row-major text, `precision` digits, scientific notation, ',' between values, '\n' per row:
// byte-identical to WriteDelimitedFile (delimited_file.hpp:49-76).
bool WriteCsv(const double* buffer, unsigned int ldim, unsigned int height, unsigned int width,
              const std::string& filename, unsigned int precision)
{
    std::ofstream out(filename);
    if (!out) return false;
    out << std::scientific;
    out.precision(precision);
    for (unsigned int r = 0; r != height; ++r) {
        for (unsigned int c = 0; c + 1 < width; ++c) out << buffer[(size_t)c * ldim + r] << ',';
        out << buffer[(size_t)(width - 1) * ldim + r] << std::endl;
    }
    out.close();
    return true;
}

static bool is_comment(const std::string& line) { return !line.empty() && (line[0] == '#' || line[0] == '%'); }

// Reads a dense CSV into a column-major buffer (ldim = height).  Leading blank/comment lines are
// skipped; width = delimiters in the first data line + 1; height = number of remaining lines
// (LoadDelimitedFile, delimited_file.hpp:79-135; GetDimensions, delimited_file.cpp:72-98).
bool LoadCsv(std::vector<double>& buffer, unsigned int& height, unsigned int& width, const std::string& filename)
{
    std::ifstream in(filename);
    if (!in) return false;
    std::string line;
    std::vector<std::string> rows;
    bool started = false;
    while (std::getline(in, line)) {
        if (!started) {
            if (line.empty() || is_comment(line)) continue;
            started = true;
        }
        // the reference stops counting at a final line without '\n' only if it is empty
        rows.push_back(line);
    }
    while (!rows.empty() && rows.back().empty()) rows.pop_back();
    if (rows.empty()) return false;
    width = 1 + (unsigned int)std::count(rows[0].begin(), rows[0].end(), ',');
    height = (unsigned int)rows.size();
    buffer.assign((size_t)height * width, 0.0);
    for (unsigned int r = 0; r < height; ++r) {
        std::istringstream data(rows[r]);
        char dummy;
        for (unsigned int c = 0; c != width; ++c) {
            double v = 0.0;
            data >> v;
            data >> dummy;
            buffer[(size_t)c * height + r] = v;
        }
    }
    return true;
}

// MatrixMarket coordinate -> CSC (LoadMatrixMarketFile, sparse_matrix_io.hpp:118-262):
// banner "%%MatrixMarket matrix coordinate {real|integer|pattern} {general|symmetric|skew-symmetric}",
// comment lines, "rows cols nnz", then 1-based "row col [value]" lines; symmetric / skew files mirror
// off-diagonal entries; entries are bucketed by column in file order (duplicates kept, as Compress() does).
bool LoadMatrixMarket(const std::string& filename, unsigned int& height, unsigned int& width,
                      std::vector<unsigned int>& col_offsets, std::vector<unsigned int>& row_indices,
                      std::vector<double>& data)
{
    std::ifstream in(filename);
    if (!in) return false;
    std::string line;
    if (!std::getline(in, line)) return false;
    std::string lower = line;
    std::transform(lower.begin(), lower.end(), lower.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    if (lower.compare(0, 14, "%%matrixmarket") != 0) return false;
    std::istringstream banner(lower);
    std::string tag, obj, fmt, field, symm;
    banner >> tag >> obj >> fmt >> field >> symm;
    if (obj != "matrix") return false;
    if (fmt != "coordinate") { std::cerr << "Only sparse MatrixMarket files are supported." << std::endl; return false; }
    const bool is_pattern = field == "pattern";
    if (field != "real" && field != "integer" && !is_pattern) {
        std::cerr << "Only real, integer, and pattern MatrixMarket formats are supported." << std::endl;
        return false;
    }
    const bool is_symmetric = symm == "symmetric", is_skew = symm == "skew-symmetric";
    if (symm != "general" && !is_symmetric && !is_skew) {
        std::cerr << "Only general, symmetric, and skew-symmetric MatrixMarket formats are supported." << std::endl;
        return false;
    }
    // skip comments
    unsigned int nnz_file = 0;
    for (;;) {
        if (!std::getline(in, line)) return false;
        if (line.empty() || line[0] == '%') continue;
        std::istringstream sz(line);
        if (!(sz >> height >> width >> nnz_file)) { std::cerr << "could not read matrix coordinate information" << std::endl; return false; }
        break;
    }
    std::vector<unsigned int> r, c;
    std::vector<double> v;
    r.reserve(nnz_file); c.reserve(nnz_file); v.reserve(nnz_file);
    unsigned int line_count = 0;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        ++line_count;
        std::istringstream d(line);
        long long row = 0, col = 0;
        double val = 1.0;
        d >> row >> col;
        if (!is_pattern) d >> val;
        if (row <= 0 || col <= 0 || row > (long long)height || col > (long long)width) {
            std::cerr << "\nError reading file " << filename << "\nLine " << line << " contains an invalid index." << std::endl;
            return false;
        }
        r.push_back((unsigned)(row - 1)); c.push_back((unsigned)(col - 1)); v.push_back(val);
        if (row != col && (is_symmetric || is_skew)) {
            r.push_back((unsigned)(col - 1)); c.push_back((unsigned)(row - 1)); v.push_back(is_skew ? -val : val);
        }
    }
    if (line_count != nnz_file) {
        std::cerr << "\nError reading file " << filename << "\nFound " << line_count << " nonzero entries, expected "
                  << nnz_file << std::endl;
        return false;
    }
    if (v.empty()) return false;
    // stable counting sort by column
    col_offsets.assign((size_t)width + 1, 0u);
    for (unsigned cc : c) col_offsets[(size_t)cc + 1] += 1;
    for (unsigned j = 0; j < width; ++j) col_offsets[(size_t)j + 1] += col_offsets[j];
    row_indices.resize(v.size());
    data.resize(v.size());
    std::vector<unsigned> fill(col_offsets.begin(), col_offsets.end() - 1);
    for (size_t i = 0; i < v.size(); ++i) {
        const unsigned q = fill[c[i]]++;
        row_indices[q] = r[i];
        data[q] = v[i];
    }
    return true;
}

}  // namespace smallk_amd_io

extern "C" int smk_load_matrix_market(const char* filename, unsigned* height, unsigned* width, unsigned* nnz,
                                      unsigned* col_offsets, unsigned* row_indices, double* data)
{
    std::vector<unsigned> co, ri;
    std::vector<double> d;
    unsigned h = 0, w = 0;
    if (!smallk_amd_io::LoadMatrixMarket(filename ? filename : "", h, w, co, ri, d)) return 0;
    *height = h; *width = w; *nnz = (unsigned)d.size();
    if (col_offsets) std::memcpy(col_offsets, co.data(), co.size() * sizeof(unsigned));
    if (row_indices) std::memcpy(row_indices, ri.data(), ri.size() * sizeof(unsigned));
    if (data) std::memcpy(data, d.data(), d.size() * sizeof(double));
    return 1;
}

extern "C" int smk_write_csv(const double* buf, unsigned ldim, unsigned height, unsigned width,
                             const char* filename, unsigned precision)
{
    return smallk_amd_io::WriteCsv(buf, ldim, height, width, filename, precision) ? 1 : 0;
}

extern "C" int smk_load_csv(const char* filename, double* out, unsigned long cap, unsigned* height, unsigned* width)
{
    std::vector<double> v;
    unsigned h = 0, w = 0;
    if (!smallk_amd_io::LoadCsv(v, h, w, filename)) return 0;
    *height = h;
    *width = w;
    if ((unsigned long)h * w > cap) return -1;
    std::memcpy(out, v.data(), sizeof(double) * (size_t)h * w);
    return 1;
}

// =============================================================================================
// inner seam: nmf.hpp
// =============================================================================================
static int g_nmf_storage = SMK_STORE_F32;

void NmfSetDeviceStorage(int storage) { g_nmf_storage = (storage == SMK_STORE_BF16) ? SMK_STORE_BF16 : SMK_STORE_F32; }
int NmfGetDeviceStorage() { return g_nmf_storage; }

void NmfInitialize(int /*argc*/, char* /*argv*/[])
{
    if (smk_initialize(-1) != SMK_OK) throw std::runtime_error(std::string("NmfInitialize: ") + smk_last_error());
}

Result NmfIsInitialized() { return smk_is_initialized() == SMK_INITIALIZED ? Result::INITIALIZED : Result::NOTINITIALIZED; }

void NmfFinalize() { smk_finalize(); }

static smk_options to_c(const NmfOptions& o)
{
    smk_options c;
    c.tol = o.tol;
    c.algorithm = (int)o.algorithm;                 // same numbering as NmfAlgorithm
    c.prog_est_algorithm = (int)o.prog_est_algorithm;
    c.height = o.height; c.width = o.width; c.k = o.k;
    c.min_iter = o.min_iter; c.max_iter = o.max_iter; c.tolcount = o.tolcount;
    c.max_threads = o.max_threads;
    c.verbose = o.verbose ? 1 : 0;
    c.normalize = o.normalize ? 1 : 0;
    return c;
}

bool IsValid(const NmfOptions& opts, bool validate_matrix)
{
    smk_options c = to_c(opts);
    return smk_is_valid(&c, validate_matrix ? 1 : 0) != 0;
}

static Result to_result(int rc)
{
    switch (rc) {
        case SMK_OK: return Result::OK;
        case SMK_NOTINITIALIZED: return Result::NOTINITIALIZED;
        case SMK_INITIALIZED: return Result::INITIALIZED;
        case SMK_BAD_PARAM: return Result::BAD_PARAM;
        case SMK_SIZE_TOO_LARGE: return Result::SIZE_TOO_LARGE;
        case SMK_FAILURE: return Result::FAILURE;
        default:
            // device errors / unsupported algorithm have no Result code in the reference: surface loudly
            throw std::runtime_error(std::string("smallk_amd device path: ") + smk_last_error());
    }
}

// Column shards of a dense factorisation: SMK_NUM_GPUS=N runs Nmf() / smallk::Nmf() / the nmf tool on N
// GPUs of this node (one host thread per device, RCCL collectives issued from C); SMK_SHARDS_ON_ONE_GPU=1 keeps the N
// shards on the current device with the in-process stand-in for RCCL (boxes with fewer GPUs, tests).
static void requested_shards(int* shards, int* stub)
{
    *shards = 1;
    *stub = 0;
    const char* e = getenv("SMK_NUM_GPUS");
    if (!e || atoi(e) <= 1) return;
    *shards = atoi(e) > 16 ? 16 : atoi(e);
    const char* s1 = getenv("SMK_SHARDS_ON_ONE_GPU");
    *stub = (s1 && atoi(s1) != 0) ? 1 : 0;
}

Result Nmf(const NmfOptions& options, double* buf_a, int ldim_a, double* buf_w, int ldim_w, double* buf_h,
           int ldim_h, NmfStats& stats)
{
    if (smk_is_initialized() != SMK_INITIALIZED) {
        std::cerr << "nmflib error: nmf_initialize() must be called prior to any factorization routine\n" << std::endl;
        return Result::NOTINITIALIZED;
    }
    if (!IsValid(options)) return Result::BAD_PARAM;
    // leading-dimension violations throw in the reference (nmf.cpp:213-219)
    if (ldim_w < options.height) throw std::logic_error("nmflib error: leading dimension of W return buffer too small");
    if (ldim_h < options.k) throw std::logic_error("nmflib error: leading dimension of H return buffer too small");
    if (options.algorithm == NmfAlgorithm::RANK2 && options.k != 2) throw std::runtime_error("rank2 algorithm requires k == 2");
    smk_options c = to_c(options);
    smk_stats st{0, 0};
    int shards = 1, stub = 0;
    requested_shards(&shards, &stub);
    int rc = shards > 1 ? smk_nmf_dense_sharded(&c, buf_a, ldim_a, buf_w, ldim_w, buf_h, ldim_h, &st, g_nmf_storage, shards, nullptr, stub)
                        : smk_nmf_dense(&c, buf_a, ldim_a, buf_w, ldim_w, buf_h, ldim_h, &st, g_nmf_storage);
    stats.elapsed_us = st.elapsed_us;
    stats.iteration_count = st.iteration_count;
    return to_result(rc);
}

Result NmfSparse(const NmfOptions& options, const unsigned int height, const unsigned int width, const unsigned int nz,
                 const unsigned int* col_offsets, const unsigned int* row_indices, const double* data, double* buf_w,
                 int ldim_w, double* buf_h, int ldim_h, NmfStats& stats)
{
    if (smk_is_initialized() != SMK_INITIALIZED) {
        std::cerr << "nmflib error: nmf_initialize() must be called prior to any factorization routine\n" << std::endl;
        return Result::NOTINITIALIZED;
    }
    if (!IsValid(options)) return Result::BAD_PARAM;
    if (ldim_w < options.height) throw std::logic_error("nmflib error: leading dimension of W return buffer too small");
    if (ldim_h < options.k) throw std::logic_error("nmflib error: leading dimension of H return buffer too small");
    smk_options c = to_c(options);
    smk_stats st{0, 0};
    int rc = smk_nmf_sparse(&c, height, width, nz, col_offsets, row_indices, data, buf_w, ldim_w, buf_h, ldim_h, &st);
    stats.elapsed_us = st.elapsed_us;
    stats.iteration_count = st.iteration_count;
    return to_result(rc);
}

// =============================================================================================
// outer seam: namespace smallk  (global state, not thread safe -- as the reference, smallk.cpp:46-67)
// =============================================================================================
namespace smallk {

static bool matrix_loaded = false, is_sparse = false;
static std::vector<double> buf_a, buf_w, buf_h;
static std::vector<double> sp_data;
static std::vector<unsigned int> sp_rows, sp_cols;   // CSC: row indices, column offsets (width+1)
static unsigned int m = 0u, n = 0u, k = 0u;
static unsigned int ldim_a = 0u, ldim_w = 0u, ldim_h = 0u;
static double nmf_tolerance = 0.005;
static double hier_nmf2_tolerance = 0.0001;
static unsigned int max_iter = 5000, min_iter = 5, max_threads = 1, maxterms = 5, outprecision = 6;
static OutputFormat clustfile_format = OutputFormat::JSON;
static std::string outdir, matrix_filepath;
static uint64_t rng_seed = 0;
static uint64_t rng_draws = 0;            // so that successive RandomMatrix calls differ
static DeviceStorage device_storage = DEVICE_F32;
static NmfStats last_stats;
// The loaded matrix is uploaded to HBM once and stays there across Nmf / HierNmf2 calls
// (smallk_example.cpp factors the same matrix seven times); dropped by Reset / LoadMatrix / Finalize
// and rebuilt when the device storage type changes.
static smk_matrix* resident = nullptr;
static int resident_storage = -1;
static void drop_resident()
{
    if (resident) smk_matrix_destroy(resident);
    resident = nullptr;
    resident_storage = -1;
}
static smk_matrix* ensure_resident()
{
    const int want = is_sparse ? SMK_STORE_F32 : (int)device_storage;
    if (resident && resident_storage == want) return resident;
    drop_resident();
    int rc;
    if (is_sparse) {
        rc = smk_matrix_create_sparse(&resident, m, n, 0, n, (int64_t)sp_data.size(), &sp_cols[0], &sp_rows[0], &sp_data[0]);
    } else {
        rc = smk_matrix_create(&resident, m, n, 0, n, want);
        if (rc == SMK_OK) rc = smk_matrix_upload_f64(resident, &buf_a[0], ldim_a);
    }
    if (rc != SMK_OK) {
        drop_resident();
        throw std::runtime_error(std::string("smallk error: could not place the matrix on the device: ") + smk_last_error());
    }
    resident_storage = want;
    return resident;
}
static bool dict_loaded = false;
static std::string dict_filepath;
static std::vector<std::string> dictionary;

static const std::string DEFAULT_FILENAME_W("w.csv");
static const std::string DEFAULT_FILENAME_H("h.csv");

static unsigned int hw_threads()
{
    unsigned int t = std::thread::hardware_concurrency();
    return t == 0 ? 2 : t;
}

// smallk.cpp:81-111
void Reset()
{
    min_iter = 5;
    max_iter = 5000;
    nmf_tolerance = 0.005;
    hier_nmf2_tolerance = 0.0001;
    max_threads = hw_threads();
    maxterms = 5;
    outprecision = 6;
    clustfile_format = OutputFormat::JSON;
    outdir = std::string("");
    matrix_loaded = false;
    drop_resident();
    dict_loaded = false;
    is_sparse = false;
    matrix_filepath.clear();
    dict_filepath.clear();
    buf_a.clear(); buf_w.clear(); buf_h.clear();
    sp_data.clear(); sp_rows.clear(); sp_cols.clear();
    m = n = k = ldim_a = ldim_w = ldim_h = 0u;
}

// smallk.cpp:114-119 (time seeded RNG; GPU init replaces EL::Initialize)
void Initialize(int& /*argc*/, char**& /*argv*/)
{
    Reset();
    rng_seed = (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count();
    // SMALLK_SEED pins the seed for callers that cannot be changed to call SeedRNG() (the reference's own example
    // programs draw their initial factors right after Initialize)
    if (const char* e = getenv("SMALLK_SEED")) rng_seed = (uint64_t)strtoull(e, nullptr, 10);
    rng_draws = 0;
    if (smk_initialize(-1) != SMK_OK) throw std::runtime_error(std::string("smallk error (Initialize): ") + smk_last_error());
}

bool IsInitialized() { return smk_is_initialized() == SMK_INITIALIZED; }
void Finalize()
{
    drop_resident();
    smk_finalize();
}

unsigned int GetMajorVersion() { return SMALLK_MAJOR_VERSION; }
unsigned int GetMinorVersion() { return SMALLK_MINOR_VERSION; }
unsigned int GetPatchLevel() { return SMALLK_PATCH_LEVEL; }
std::string GetVersionString()
{
    std::ostringstream v;
    v << GetMajorVersion() << "." << GetMinorVersion() << "." << GetPatchLevel();
    return v.str();
}

void SeedRNG(const int seed) { rng_seed = (uint64_t)(int64_t)seed; rng_draws = 0; }

// uniform [0,1) init (RandomMatrix, center .5 radius .5: matrix_generator.hpp:61-82).  The
// reference's stream depends on its thread count; here it is the counter-based generator.
static void RandomMatrix(double* buf, unsigned int ld, unsigned int h, unsigned int w)
{
    smk_uniform_fill_host(buf, ld, h, w, 0, 0, h, rng_seed + 0x9E37u * (++rng_draws), 0);
}

static bool has_ext(const std::string& path, const char* ext)
{
    size_t dot = path.find_last_of('.');
    if (dot == std::string::npos) return false;
    std::string e = path.substr(dot + 1);
    std::transform(e.begin(), e.end(), e.begin(), [](unsigned char ch) { return (char)std::toupper(ch); });
    return e == ext;
}

// smallk.cpp:163-201
void LoadMatrix(const std::string& filepath)
{
    if (filepath.empty()) throw std::runtime_error("smallk error (LoadMatrix): matrix filename is invalid.");
    std::cout << "Loading matrix..." << std::endl;
    matrix_loaded = false;
    drop_resident();
    if (has_ext(filepath, "MTX")) {       // IsSparse(filepath): MatrixMarket -> sparse (smallk.cpp:172-186)
        if (!smallk_amd_io::LoadMatrixMarket(filepath, m, n, sp_cols, sp_rows, sp_data)) {
            matrix_filepath.clear();
            throw std::runtime_error("smallk error (LoadMatrix): load failed for file " + filepath);
        }
        is_sparse = true;
        matrix_loaded = true;
        matrix_filepath = filepath;
        return;
    }
    bool ok = smallk_amd_io::LoadCsv(buf_a, m, n, filepath);
    if (!ok || buf_a.size() < (size_t)m * n) {
        matrix_filepath.clear();
        throw std::runtime_error("smallk error (LoadMatrix): load failed for file " + filepath);
    }
    ldim_a = m;
    is_sparse = false;
    matrix_loaded = true;
    matrix_filepath = filepath;
}

// Dense column-major buffer (documented semantics, smallk.hpp:176-188).  The reference's copy
// loop (smallk.cpp:249-255) swaps height and width and is only right for square input; this
// implements what the header documents.
void LoadMatrix(const double* buffer, const unsigned int ldim, const unsigned int height, const unsigned int width)
{
    std::cout << "Loading dense matrix..." << std::endl;
    matrix_loaded = false;
    drop_resident();
    if (0 == height) throw std::runtime_error("smallk error (LoadSparseMatrixFromBuffer): invalid height input.");
    if (0 == width) throw std::runtime_error("smallk error (LoadSparseMatrixFromBuffer): invalid width input.");
    if (!buffer) throw std::runtime_error("smallk error (LoadSparseMatrixFromBuffer): empty data pointer.");
    if (ldim < height) throw std::runtime_error("smallk error (LoadMatrix): leading dimension smaller than height.");
    buf_a.resize((size_t)height * width);
    ldim_a = height;
    for (unsigned int c = 0; c != width; ++c)
        for (unsigned int r = 0; r != height; ++r) buf_a[(size_t)c * height + r] = buffer[(size_t)c * ldim + r];
    m = height;
    n = width;
    is_sparse = false;
    matrix_loaded = true;
    matrix_filepath = "NA";
}

// sparse CSC from memory (smallk.cpp:268-340)
void LoadMatrix(const unsigned int height, const unsigned int width, const unsigned int nz,
                const std::vector<double>& data, const std::vector<unsigned int>& row_indices,
                const std::vector<unsigned int>& col_offsets)
{
    std::cout << "Loading sparse matrix..." << std::endl;
    matrix_loaded = false;
    drop_resident();
    if (row_indices.size() != data.size())
        throw std::runtime_error("smallk error (LoadSparseMatrixFromBuffer): invalid input vectors.");
    if (0 == height) throw std::runtime_error("smallk error (LoadSparseMatrixFromBuffer): invalid height input.");
    if (0 == width) throw std::runtime_error("smallk error (LoadSparseMatrixFromBuffer): invalid width input.");
    if ((uint64_t)data.size() > (uint64_t)height * width)
        throw std::runtime_error("smallk error (LoadSparseMatrixFromBuffer): more nonzeros than matrix elements.");
    if (col_offsets.size() < (size_t)width + 1 || data.size() < nz)
        throw std::runtime_error("smallk error (LoadSparseMatrixFromBuffer): invalid input vectors.");
    sp_data.assign(data.begin(), data.begin() + nz);
    sp_rows.assign(row_indices.begin(), row_indices.begin() + nz);
    sp_cols.assign(col_offsets.begin(), col_offsets.begin() + width + 1);
    m = height;
    n = width;
    is_sparse = true;
    matrix_loaded = true;
    matrix_filepath = "NA";
}

bool IsMatrixLoaded() { return matrix_loaded; }

std::string GetOutputDir() { return outdir; }

static std::string ensure_trailing_sep(const std::string& s)
{
    if (s.empty()) return std::string("");
    return s[s.size() - 1] == '/' ? s : s + '/';
}

// smallk.cpp:348-380
void SetOutputDir(const std::string& output_dir)
{
    std::ostringstream msg;
    msg << "smallk error (SetOutputDir): ";
    std::string full_path;
    if (!output_dir.empty() && ('/' != output_dir[0])) {
        char* cur = getcwd(nullptr, 0);
        if (!cur) { msg << "could not determine current directory."; throw std::runtime_error(msg.str()); }
        full_path = ensure_trailing_sep(std::string(cur));
        free(cur);
    }
    full_path += output_dir;
    struct stat st;
    if (!(0 == stat(full_path.c_str(), &st) && (S_IFDIR == (st.st_mode & S_IFDIR)))) {
        msg << "the directory \"" << full_path << "\" does not exist.";
        throw std::logic_error(msg.str());
    }
    outdir = ensure_trailing_sep(full_path);
}

double GetNmfTolerance() { return nmf_tolerance; }
double GetHierNmf2Tolerance() { return hier_nmf2_tolerance; }
unsigned int GetMaxIter() { return max_iter; }
unsigned int GetMinIter() { return min_iter; }
unsigned int GetMaxThreads() { return max_threads; }
unsigned int GetMaxTerms() { return maxterms; }
unsigned int GetOutputPrecision() { return outprecision; }
OutputFormat GetOutputFormat() { return clustfile_format; }

void SetNmfTolerance(const double tol)
{
    if ((tol <= 0.0) || (tol >= 1.0)) throw std::logic_error("smallk error (SetNmfTolerance): require 0.0 < tol < 1.0");
    nmf_tolerance = tol;
}
void SetHierNmf2Tolerance(const double tol)
{
    if ((tol <= 0.0) || (tol >= 1.0)) throw std::logic_error("smallk error (SetHierNmf2Tolerance): require 0.0 < tol < 1.0");
    hier_nmf2_tolerance = tol;
}
void SetMaxIter(const unsigned int v) { max_iter = v == 0 ? 1 : v; }
void SetMinIter(const unsigned int v) { min_iter = v == 0 ? 1 : v; }
void SetMaxThreads(const unsigned int mt)
{
    max_threads = std::min(mt, hw_threads());
    if (0 == max_threads) max_threads = 1;
}
void SetMaxTerms(const unsigned int v) { maxterms = v == 0 ? 1 : v; }
void SetOutputPrecision(const unsigned int num_digits)
{
    outprecision = num_digits;
    if (0 == outprecision) outprecision = 1;
    if (outprecision > (unsigned)std::numeric_limits<double>::max_digits10) outprecision = std::numeric_limits<double>::max_digits10;
}
void SetOutputFormat(const OutputFormat format) { clustfile_format = format; }

void SetDeviceStorage(const DeviceStorage s) { device_storage = (s == DEVICE_BF16) ? DEVICE_BF16 : DEVICE_F32; }
DeviceStorage GetDeviceStorage() { return device_storage; }
unsigned int GetIterationCount() { return (unsigned int)last_stats.iteration_count; }
unsigned long long GetElapsedMicroseconds() { return last_stats.elapsed_us; }

static std::string elapsed_string(unsigned long long us)
{   // utils.cpp:172-218
    const unsigned long long S = 1000000ull, M = 60 * S, HR = 60 * M;
    unsigned long long hrs = 0, min = 0, sec = 0, e = us;
    if (e >= HR) { hrs = e / HR; e -= hrs * HR; }
    if (e >= M) { min = e / M; e -= min * M; }
    if (e >= S) { sec = e / S; e -= sec * S; }
    double seconds = (double)sec + e / 1.0e6;
    char buf[256];
    if (hrs > 0) snprintf(buf, sizeof(buf), "%d hours %d min %.3f sec.", (int)hrs, (int)min, seconds);
    else if (min > 0) snprintf(buf, sizeof(buf), "%d min %.3f sec.", (int)min, seconds);
    else snprintf(buf, sizeof(buf), "%.3f sec.", seconds);
    return std::string(buf);
}

static void print_opts(const NmfOptions& o)
{   // smallk.cpp:869-917
    using std::cout; using std::endl;
    cout << "\n                parameters: \n" << endl;
    cout << "\t         algorithm: ";
    switch (o.algorithm) {
        case NmfAlgorithm::MU: cout << "Multiplicative Updating"; break;
        case NmfAlgorithm::HALS: cout << "HALS"; break;
        case NmfAlgorithm::RANK2: cout << "Rank 2"; break;
        case NmfAlgorithm::BPP: cout << "Nonnegative Least Squares with Block Principal Pivoting"; break;
    }
    cout << endl;
    cout << "\tstopping criterion: "
         << (o.prog_est_algorithm == NmfProgressAlgorithm::PG_RATIO ? "Ratio of Projected Gradients" : "Relative Change in the F-norm of W")
         << endl;
    cout << "\t            height: " << o.height << endl;
    cout << "\t             width: " << o.width << endl;
    cout << "\t                 k: " << o.k << endl;
    cout << "\t           miniter: " << o.min_iter << endl;
    cout << "\t           maxiter: " << o.max_iter << endl;
    cout << "\t               tol: " << o.tol << endl;
    cout << "\t        matrixfile: " << matrix_filepath << endl;
    cout << "\t        maxthreads: " << o.max_threads << endl;
    cout << endl;
}

// ---- smallk::Nmf (smallk.cpp:471-650) in four steps: rank + algorithm, initial factors, device run, result files ----

// public algorithm id -> the inner seam's (the two enums are numbered differently, smallk.cpp:497-513)
static NmfAlgorithm inner_algorithm(const Algorithm algorithm)
{
    switch (algorithm) {
        case Algorithm::MU: return NmfAlgorithm::MU;
        case Algorithm::HALS: return NmfAlgorithm::HALS;
        case Algorithm::RANK2: return NmfAlgorithm::RANK2;
        case Algorithm::BPP: return NmfAlgorithm::BPP;
    }
    throw std::logic_error("smallk error (NMF): unknown NMF algorithm.");
}

// One initial factor, rows x cols with leading dimension rows: drawn from the generator when `file` is empty,
// else read from a CSV file.  The file goes into a scratch vector first, so a file of the wrong shape leaves
// `dst` (and the sizes LockedBufferW/H report) consistent.
static void initial_factor(const char* name, const std::string& file, unsigned rows, unsigned cols, std::vector<double>& dst)
{
    std::cout << "Initializing matrix " << name << "..." << std::endl;
    dst.resize((size_t)rows * cols);
    if (file.empty()) {
        RandomMatrix(dst.data(), rows, rows, cols);
        return;
    }
    std::vector<double> loaded;
    unsigned fh = 0, fw = 0;
    if (!smallk_amd_io::LoadCsv(loaded, fh, fw, file))
        throw std::runtime_error("smallk error (Nmf): load failed for file \"" + file + "\"");
    if (fh != rows || fw != cols) {
        std::cerr << "\tdimensions of matrix " << name << " are " << fh << " x " << fw << std::endl;
        std::cerr << "\texpected " << rows << " x " << cols << std::endl;
        throw std::logic_error(std::string("smallk error (Nmf): non-conformant matrix ") + name + ".");
    }
    std::copy(loaded.begin(), loaded.begin() + (size_t)rows * cols, dst.begin());
}

// the inner seam's checks (::Nmf, nmf.cpp:173-229) without its per-call upload: A stays resident in HBM;
// SMK_NUM_GPUS > 1 shards a dense matrix over that many devices instead
static Result run_on_device(const NmfOptions& opts, NmfStats& stats)
{
    smk_options c = to_c(opts);
    if (smk_is_initialized() != SMK_INITIALIZED) {
        std::cerr << "nmflib error: nmf_initialize() must be called prior to any factorization routine\n" << std::endl;
        return Result::NOTINITIALIZED;
    }
    if (!smk_is_valid(&c, 1)) return Result::BAD_PARAM;
    smk_stats st{0, 0};
    int rc, shards = 1, stub = 0;
    requested_shards(&shards, &stub);
    if (!is_sparse && shards > 1) {
        rc = smk_nmf_dense_sharded(&c, &buf_a[0], ldim_a, &buf_w[0], ldim_w, &buf_h[0], ldim_h, &st, (int)device_storage,
                                   shards, nullptr, stub);
    } else {
        smk_solver* sv = nullptr;
        rc = smk_solver_create(&sv, &c, ensure_resident());
        if (rc == SMK_OK) rc = smk_solver_set_factors(sv, &buf_w[0], ldim_w, &buf_h[0], ldim_h);
        if (rc == SMK_OK) {
            rc = smk_solver_run(sv, &st);
            // like the reference, W/H hold the last iterate even when the solver reports failure
            if (rc == SMK_OK || rc == SMK_FAILURE) (void)smk_solver_get_factors(sv, 0, &buf_w[0], ldim_w, &buf_h[0], ldim_h);
        }
        smk_solver_destroy(sv);
    }
    stats.elapsed_us = st.elapsed_us;
    stats.iteration_count = st.iteration_count;
    return to_result(rc);
}

void Nmf(const unsigned int kval, const Algorithm algorithm, const std::string& csv_file_w, const std::string& csv_file_h)
{
    if (!matrix_loaded) throw std::logic_error("smallk error (NMF): no matrix has been loaded.");
    if (max_iter < min_iter) throw std::logic_error("smallk error (NMF): min_iterations exceeds max_iterations.");
    if (0 == kval) throw std::logic_error("smallk error (NMF): k must be greater than 0.");
    const NmfAlgorithm alg = inner_algorithm(algorithm);
    const unsigned int rank = (NmfAlgorithm::RANK2 == alg) ? 2u : kval;
    const uint64_t int_max = (uint64_t)std::numeric_limits<int>::max();
    if ((uint64_t)m * kval > int_max) throw std::logic_error("smallk error (Nmf): mxk matrix W is too large.");
    if ((uint64_t)kval * n > int_max) throw std::logic_error("smallk error (Nmf): kxn matrix H is too large.");

    // W first, then H: the generator's draw order is part of the interface (clust_hier_util.hpp:196-203 relies on it too)
    std::vector<double> w0, h0;
    initial_factor("W", csv_file_w, m, rank, w0);
    initial_factor("H", csv_file_h, rank, n, h0);
    k = rank;
    ldim_w = m;
    ldim_h = k;
    buf_w.swap(w0);
    buf_h.swap(h0);

    NmfOptions opts;
    opts.algorithm = alg;
    // MU -> DELTA_FNORM, everything else PG_RATIO (smallk.cpp:581-584)
    opts.prog_est_algorithm = (NmfAlgorithm::MU == alg) ? NmfProgressAlgorithm::DELTA_FNORM : NmfProgressAlgorithm::PG_RATIO;
    opts.tol = nmf_tolerance;
    opts.height = m;
    opts.width = n;
    opts.k = k;
    opts.min_iter = min_iter;
    opts.max_iter = max_iter;
    opts.tolcount = 1;
    opts.max_threads = max_threads;
    opts.verbose = true;
    opts.normalize = true;
    print_opts(opts);

    NmfStats stats;
    const Result result = run_on_device(opts, stats);
    last_stats = stats;
    std::cout << "Elapsed wall clock time: " << elapsed_string(stats.elapsed_us) << std::endl << std::endl;
    if (Result::OK != result) throw std::runtime_error("smallk error (Nmf): NMF solver failure.");

    std::cout << "Writing output files..." << std::endl;
    const struct { const char* what; const std::string file; const double* buf; unsigned ld, rows, cols; } outputs[] = {
        {"W", outdir + DEFAULT_FILENAME_W, &buf_w[0], ldim_w, m, k},
        {"H", outdir + DEFAULT_FILENAME_H, &buf_h[0], ldim_h, k, n}};
    for (const auto& o : outputs)
        if (!smallk_amd_io::WriteCsv(o.buf, o.ld, o.rows, o.cols, o.file, outprecision))
            throw std::runtime_error(std::string("smallk error (Nmf): could not write ") + o.what + " result.");
}

const double* LockedBufferW(unsigned int& ldim, unsigned int& height, unsigned int& width)
{
    ldim = m; height = m; width = k;
    return buf_w.empty() ? nullptr : &buf_w[0];
}

const double* LockedBufferH(unsigned int& ldim, unsigned int& height, unsigned int& width)
{
    ldim = k; height = k; width = n;
    return buf_h.empty() ? nullptr : &buf_h[0];
}

// ---- clustering ---------------------------------------------------------------------------

// smallk.cpp:675-691; LoadStringsFromFile (common/src/utils.cpp:220-239): one term per complete
// line, an unterminated last line is dropped
void LoadDictionary(const std::string& filepath)
{
    std::cout << "Loading dictionary..." << std::endl;
    dictionary.clear();
    dict_loaded = false;
    std::ifstream in(filepath);
    if (!in) throw std::runtime_error("smallk error (LoadDictionary): load failed for file " + filepath);
    std::string line;
    while (in) {
        std::getline(in, line, '\n');
        if (in.eof()) break;
        dictionary.push_back(line);
    }
    dict_filepath = filepath;
    dict_loaded = true;
}

// smallk.cpp:694-707
void LoadDictionary(const std::vector<std::string>& terms)
{
    std::cout << "Loading dictionary..." << std::endl;
    dictionary.assign(terms.begin(), terms.end());
    dict_filepath.clear();
    dict_loaded = true;
}

// smallk.cpp:738-856 (HierNmf2Internal): RANK2 + PG_RATIO, unbalanced 0.1, trial_allowance 3;
// writes assignments_<N>.csv and tree_<N>.{xml,json} into the output directory.
static void hier_nmf2_internal(const bool generate_flat, const unsigned int num_clusters)
{
    if (!matrix_loaded) throw std::logic_error("smallk error (HierNmf2): no matrix has been loaded.");
    if (!dict_loaded) throw std::logic_error("smallk error (HierNmf2): no dictionary has been loaded.");
    if (0 == num_clusters) throw std::logic_error("smallk error (HierNmf2): num_clusters must be greater than 0.");
    if (2ull * m > (uint64_t)std::numeric_limits<int>::max())
        throw std::logic_error("smallk error (HierNmf2): matrix height too large.");
    if (2ull * n > (uint64_t)std::numeric_limits<int>::max())
        throw std::logic_error("smallk error (HierNmf2): matrix width too large.");
    if (dictionary.size() < m)
        throw std::logic_error("smallk error (HierNmf2): dictionary has fewer terms than the matrix has rows.");

    smk_clust_options co;
    co.nmf.tol = hier_nmf2_tolerance;
    co.nmf.algorithm = SMK_ALG_RANK2;
    co.nmf.prog_est_algorithm = SMK_PROG_PG_RATIO;
    co.nmf.height = (int)m;
    co.nmf.width = (int)n;
    co.nmf.k = 2;
    co.nmf.min_iter = (int)min_iter;
    co.nmf.max_iter = (int)max_iter;
    co.nmf.tolcount = 1;
    co.nmf.max_threads = (int)max_threads;
    co.nmf.verbose = 0;
    co.nmf.normalize = 1;
    co.maxterms = (int)maxterms;
    co.unbalanced = 0.1;
    co.trial_allowance = 3;
    co.num_clusters = (int)num_clusters;
    co.verbose = 1;
    co.flat = generate_flat ? 1 : 0;

    const std::string output_dir = ensure_trailing_sep(outdir);
    const bool xml = (OutputFormat::XML == clustfile_format);
    std::ostringstream assign_name, tree_name;
    assign_name << output_dir << "assignments_" << num_clusters << ".csv";
    tree_name << output_dir << "tree_" << num_clusters << (xml ? ".xml" : ".json");

    using std::cout; using std::endl;
    cout << "\n\t        parameters: \n" << endl;
    cout << "\t            height: " << co.nmf.height << endl;
    cout << "\t             width: " << co.nmf.width << endl;
    cout << "\t        matrixfile: " << matrix_filepath << endl;
    cout << "\t          dictfile: " << dict_filepath << endl;
    cout << "\t               tol: " << co.nmf.tol << endl;
    cout << "\t           miniter: " << co.nmf.min_iter << endl;
    cout << "\t           maxiter: " << co.nmf.max_iter << endl;
    cout << "\t          maxterms: " << co.maxterms << endl;
    cout << "\t        maxthreads: " << co.nmf.max_threads << endl;

    smk_tree* tree = nullptr;
    smk_clust_stats stats = {0, 0};
    const auto t0 = std::chrono::high_resolution_clock::now();
    int rc;
    rc = smk_clust_resident(&co, ensure_resident(), rng_seed, &rng_draws, nullptr, &tree, &stats);
    const uint64_t us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
                            std::chrono::high_resolution_clock::now() - t0).count();
    cout << "\nElapsed wall clock time: " << elapsed_string(us) << endl;
    if (rc != SMK_OK) {
        smk_tree_destroy(tree);
        const std::string why = smk_last_error();
        throw std::runtime_error(why.empty() ? std::string("smallk error (HierNMF2): HierNMF2 fatal error.")
                                             : "smallk error (HierNMF2): " + why);
    }
    cout << (stats.nmf_count - stats.max_count) << "/" << stats.nmf_count << " factorizations converged." << endl << endl;
    cout << "Writing output files..." << endl;
    if (smk_tree_write_assignments(tree, assign_name.str().c_str()) != SMK_OK)
        std::cerr << "\terror writing assignments file" << endl;
    std::vector<const char*> terms(dictionary.size());
    for (size_t i = 0; i < dictionary.size(); ++i) terms[i] = dictionary[i].c_str();
    if (smk_tree_write(tree, tree_name.str().c_str(), xml ? 0 : 1, terms.data(), (int64_t)terms.size()) != SMK_OK)
        std::cerr << "\terror writing hierarchical results file" << endl;
    if (generate_flat) {
        // RunHierNmf2 (run_hier_nmf2.hpp:57-66) + FlatClustWriteResults(outdir, ...) (flat_clust_output.cpp:144-173)
        const unsigned int kc = num_clusters;
        k = kc; ldim_w = m; ldim_h = kc;
        buf_w.assign((size_t)m * kc, 0.0);
        buf_h.assign((size_t)kc * n, 0.0);
        std::vector<float> probabilities((size_t)kc * n);
        std::vector<unsigned int> assignments_flat(n);
        std::vector<int> term_indices((size_t)maxterms * kc, 0);
        int frc = smk_tree_flat_factors(tree, &buf_w[0], m, &buf_h[0], kc);
        if (frc == SMK_OK) frc = smk_compute_fuzzy_assignments(&buf_h[0], kc, kc, n, probabilities.data());
        if (frc == SMK_OK) frc = smk_compute_assignments(&buf_h[0], kc, kc, n, assignments_flat.data());
        if (frc == SMK_OK) frc = smk_top_terms((int)maxterms, &buf_w[0], m, m, kc, term_indices.data());
        if (frc == SMK_OK) {
            std::ostringstream fa, ff, fr;
            fa << output_dir << "assignments_flat_" << kc << ".csv";
            ff << output_dir << "assignments_fuzzy_" << kc << ".csv";
            fr << output_dir << "clusters_" << kc << (xml ? ".xml" : ".json");
            frc = smk_flatclust_write_results(fa.str().c_str(), ff.str().c_str(), fr.str().c_str(), assignments_flat.data(),
                                              n, probabilities.data(), terms.data(), (int64_t)terms.size(),
                                              term_indices.data(), (int64_t)term_indices.size(), xml ? 0 : 1, maxterms,
                                              n, kc);
        }
        if (frc != SMK_OK) {
            smk_tree_destroy(tree);
            throw std::runtime_error(std::string("smallk error (HierNMF2): ") + smk_last_error());
        }
    }
    smk_tree_destroy(tree);
}

void HierNmf2(const unsigned int num_clusters) { hier_nmf2_internal(false, num_clusters); }
void HierNmf2WithFlat(const unsigned int num_clusters) { hier_nmf2_internal(true, num_clusters); }

}  // namespace smallk

// =============================================================================================
// extern "C" handles onto namespace smallk for bindings that cannot speak C++ (ctypes, cgo, JNI).
// One function per entry of the reference's Cython extern block
// (pysmallk/interface/smallk_lib.pyx:42-88).  C++ exceptions become a status code
// (1 = std::logic_error, 2 = std::runtime_error / other) plus smk_api_last_exception().
// =============================================================================================
static std::string g_api_exc;

template <typename F>
static int api_guard(F&& f)
{
    try {
        f();
        g_api_exc.clear();
        return 0;
    } catch (const std::logic_error& e) {
        g_api_exc = e.what();
        return 1;
    } catch (const std::exception& e) {
        g_api_exc = e.what();
        return 2;
    } catch (...) {
        g_api_exc = "unknown exception";
        return 2;
    }
}

extern "C" {

const char* smk_api_last_exception(void) { return g_api_exc.c_str(); }

int smk_api_initialize(void)
{
    static int argc = 0;
    static char** argv = nullptr;
    return api_guard([&] { smallk::Initialize(argc, argv); });
}
int smk_api_is_initialized(void) { return smallk::IsInitialized() ? 1 : 0; }
void smk_api_finalize(void) { smallk::Finalize(); }
void smk_api_reset(void) { smallk::Reset(); }
void smk_api_seed_rng(int seed) { smallk::SeedRNG(seed); }
unsigned smk_api_get_major_version(void) { return smallk::GetMajorVersion(); }
unsigned smk_api_get_minor_version(void) { return smallk::GetMinorVersion(); }
unsigned smk_api_get_patch_level(void) { return smallk::GetPatchLevel(); }

int smk_api_load_matrix_file(const char* path) { return api_guard([&] { smallk::LoadMatrix(std::string(path ? path : "")); }); }
int smk_api_load_matrix_dense(const double* buf, unsigned ldim, unsigned height, unsigned width)
{
    return api_guard([&] { smallk::LoadMatrix(buf, ldim, height, width); });
}
int smk_api_load_matrix_sparse(unsigned height, unsigned width, unsigned nz, const double* data,
                               const unsigned* row_indices, const unsigned* col_offsets)
{
    return api_guard([&] {
        std::vector<double> d(data, data + nz);
        std::vector<unsigned> ri(row_indices, row_indices + nz), co(col_offsets, col_offsets + width + 1);
        smallk::LoadMatrix(height, width, nz, d, ri, co);
    });
}
int smk_api_is_matrix_loaded(void) { return smallk::IsMatrixLoaded() ? 1 : 0; }

int smk_api_set_output_dir(const char* dir) { return api_guard([&] { smallk::SetOutputDir(std::string(dir ? dir : "")); }); }
const char* smk_api_get_output_dir(void)
{
    static std::string s;
    s = smallk::GetOutputDir();
    return s.c_str();
}
void smk_api_set_output_precision(unsigned d) { smallk::SetOutputPrecision(d); }
unsigned smk_api_get_output_precision(void) { return smallk::GetOutputPrecision(); }
int smk_api_set_nmf_tolerance(double tol) { return api_guard([&] { smallk::SetNmfTolerance(tol); }); }
double smk_api_get_nmf_tolerance(void) { return smallk::GetNmfTolerance(); }
void smk_api_set_max_iter(unsigned v) { smallk::SetMaxIter(v); }
unsigned smk_api_get_max_iter(void) { return smallk::GetMaxIter(); }
void smk_api_set_min_iter(unsigned v) { smallk::SetMinIter(v); }
unsigned smk_api_get_min_iter(void) { return smallk::GetMinIter(); }
void smk_api_set_max_threads(unsigned v) { smallk::SetMaxThreads(v); }
unsigned smk_api_get_max_threads(void) { return smallk::GetMaxThreads(); }
void smk_api_set_max_terms(unsigned v) { smallk::SetMaxTerms(v); }
unsigned smk_api_get_max_terms(void) { return smallk::GetMaxTerms(); }
void smk_api_set_output_format(int f) { smallk::SetOutputFormat(f == 0 ? smallk::XML : smallk::JSON); }
int smk_api_get_output_format(void) { return (int)smallk::GetOutputFormat(); }
int smk_api_set_hiernmf2_tolerance(double tol) { return api_guard([&] { smallk::SetHierNmf2Tolerance(tol); }); }
double smk_api_get_hiernmf2_tolerance(void) { return smallk::GetHierNmf2Tolerance(); }
void smk_api_set_device_storage(int s) { smallk::SetDeviceStorage(s == 1 ? smallk::DEVICE_BF16 : smallk::DEVICE_F32); }
int smk_api_get_device_storage(void) { return (int)smallk::GetDeviceStorage(); }
unsigned smk_api_get_iteration_count(void) { return smallk::GetIterationCount(); }

/* algorithm uses smallk::Algorithm numbering: MU=0, BPP=1, HALS=2, RANK2=3 (smallk.hpp:34-40) */
int smk_api_nmf(unsigned k, int algorithm, const char* initfile_w, const char* initfile_h)
{
    return api_guard([&] {
        smallk::Nmf(k, (smallk::Algorithm)algorithm, std::string(initfile_w ? initfile_w : ""),
                    std::string(initfile_h ? initfile_h : ""));
    });
}
const double* smk_api_locked_buffer_w(unsigned* ldim, unsigned* height, unsigned* width)
{
    return smallk::LockedBufferW(*ldim, *height, *width);
}
const double* smk_api_locked_buffer_h(unsigned* ldim, unsigned* height, unsigned* width)
{
    return smallk::LockedBufferH(*ldim, *height, *width);
}
int smk_api_hiernmf2(unsigned num_clusters) { return api_guard([&] { smallk::HierNmf2(num_clusters); }); }
int smk_api_hiernmf2_with_flat(unsigned num_clusters) { return api_guard([&] { smallk::HierNmf2WithFlat(num_clusters); }); }
int smk_api_load_dictionary_file(const char* path) { return api_guard([&] { smallk::LoadDictionary(std::string(path ? path : "")); }); }
int smk_api_load_dictionary(const char* const* terms, unsigned count)
{
    return api_guard([&] {
        std::vector<std::string> v;
        for (unsigned i = 0; i < count; ++i) v.push_back(terms && terms[i] ? terms[i] : "");
        smallk::LoadDictionary(v);
    });
}

}  // extern "C"
