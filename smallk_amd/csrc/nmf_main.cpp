// smallk_amd/csrc/nmf_main.cpp -- the `nmf` command line tool on the MI355X solver.
// Same 17 flags, defaults and flow as the reference CLI (nmf/src/command_line.cpp:34-54,172-354;
// nmf/src/main.cpp:41-255): load A (.mtx -> sparse, .csv -> dense), initialise W and H from files or
// the RNG, call Nmf()/NmfSparse() at the inner seam (not smallk::), write the factors as CSV.
// Extensions: --storage f32|bf16 (how a dense A is held in HBM), --seed N (reproducible random init).
#include <getopt.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nmf.hpp"
#include "../../include/smallk_amd.h"

namespace {

struct CommandLineOptions {
    NmfOptions nmf_opts;
    std::string infile_A, infile_W, infile_H, outfile_W = "w.csv", outfile_H = "h.csv";
    int output_precision = 6;
    bool show_help = false;
    int storage = SMK_STORE_F32;
    long long seed = -1;
};

option longopts[] = {
    {"matrixfile", required_argument, nullptr, 'a'}, {"k", required_argument, nullptr, 'b'},
    {"algorithm", required_argument, nullptr, 'c'},  {"stopping", required_argument, nullptr, 'd'},
    {"tol", required_argument, nullptr, 'e'},        {"tolcount", required_argument, nullptr, 'f'},
    {"infile_W", required_argument, nullptr, 'g'},   {"infile_H", required_argument, nullptr, 'h'},
    {"outfile_W", required_argument, nullptr, 'i'},  {"outfile_H", required_argument, nullptr, 'j'},
    {"miniter", required_argument, nullptr, 'k'},    {"maxiter", required_argument, nullptr, 'l'},
    {"outprecision", required_argument, nullptr, 'm'}, {"maxthreads", required_argument, nullptr, 'n'},
    {"normalize", required_argument, nullptr, 'o'},  {"verbose", required_argument, nullptr, 'p'},
    {"help", no_argument, nullptr, 'q'},             {"storage", required_argument, nullptr, 'r'},
    {"seed", required_argument, nullptr, 's'},       {nullptr, 0, nullptr, 0}};

void ShowHelp(const std::string& prog)
{
    std::cout << "\nUsage: " << prog << "\n"
              << "        --matrixfile <filename>  Filename of the matrix to be factored.\n"
              << "                                 Either CSV format for dense or MatrixMarket format for sparse.\n"
              << "        --k <integer value>      The common dimension for factors W and H.\n"
              << "        [--algorithm  BPP]       NMF algorithm: MU, HALS, RANK2, BPP\n"
              << "        [--stopping  PG_RATIO]   Stopping criterion: PG_RATIO, DELTA\n"
              << "        [--tol  0.005]           Tolerance for the selected stopping criterion.\n"
              << "        [--tolcount  1]          Tolerance count; declare convergence after this many\n"
              << "                                 iterations with metric < tolerance; default is to\n"
              << "                                 declare convergence on the first such iteration.\n"
              << "        [--infile_W  (empty)]    Dense mxk matrix to initialize W; CSV file.\n"
              << "        [--infile_H  (empty)]    Dense kxn matrix to initialize H; CSV file.\n"
              << "        [--outfile_W  w.csv]     Filename for the W matrix result.\n"
              << "        [--outfile_H  h.csv]     Filename for the H matrix result.\n"
              << "        [--miniter  5]           Minimum number of iterations to perform.\n"
              << "        [--maxiter  5000]        Maximum number of iterations to perform.\n"
              << "        [--outprecision  6]      Write results with this many digits after the decimal point.\n"
              << "        [--maxthreads    N]      Upper limit to thread count (host side only).\n"
              << "        [--normalize  1]         Whether to normalize W and scale H.\n"
              << "        [--verbose  1]           Whether to print updates to the screen.\n"
              << "        [--storage  f32]         MI355X: hold a dense A in HBM as f32 or bf16.\n"
              << "        [--seed  (time)]         MI355X: seed of the random initializer.\n"
              << std::endl;
}

bool ParseCommandLine(int argc, char* argv[], CommandLineOptions& o)
{
    o.nmf_opts.algorithm = NmfAlgorithm::BPP;
    o.nmf_opts.height = o.nmf_opts.width = o.nmf_opts.k = 0;
    o.nmf_opts.min_iter = 5;
    o.nmf_opts.max_iter = 5000;
    o.nmf_opts.verbose = true;
    o.nmf_opts.normalize = true;
    o.nmf_opts.tol = 0.005;
    o.nmf_opts.tolcount = 1;
    o.nmf_opts.prog_est_algorithm = NmfProgressAlgorithm::PG_RATIO;
    int user_max_threads = -1, c, index;
    auto upper = [](std::string s) { std::transform(s.begin(), s.end(), s.begin(), ::toupper); return s; };
    while (-1 != (c = getopt_long(argc, argv, ":a:b:c:d:e:f:g:h:i:j:k:l:m:n:o:p:qr:s:", longopts, &index))) {
        std::string tmp;
        switch (c) {
            case 'a': o.infile_A = optarg; break;
            case 'b': o.nmf_opts.k = atoi(optarg); break;
            case 'c':
                tmp = upper(optarg);
                if (tmp == "MU") o.nmf_opts.algorithm = NmfAlgorithm::MU;
                else if (tmp == "HALS") o.nmf_opts.algorithm = NmfAlgorithm::HALS;
                else if (tmp == "RANK2") o.nmf_opts.algorithm = NmfAlgorithm::RANK2;
                else if (tmp == "BPP") o.nmf_opts.algorithm = NmfAlgorithm::BPP;
                else { std::cerr << "Invalid value specified for command-line argument: " << tmp << std::endl; return false; }
                break;
            case 'd':
                tmp = upper(optarg);
                if (tmp == "PG_RATIO") o.nmf_opts.prog_est_algorithm = NmfProgressAlgorithm::PG_RATIO;
                else if (tmp == "DELTA") o.nmf_opts.prog_est_algorithm = NmfProgressAlgorithm::DELTA_FNORM;
                else { std::cerr << "Invalid value specified for command-line argument: " << tmp << std::endl; return false; }
                break;
            case 'e': o.nmf_opts.tol = atof(optarg); break;
            case 'f': o.nmf_opts.tolcount = atoi(optarg); break;
            case 'g': o.infile_W = optarg; break;
            case 'h': o.infile_H = optarg; break;
            case 'i': o.outfile_W = optarg; break;
            case 'j': o.outfile_H = optarg; break;
            case 'k': o.nmf_opts.min_iter = atoi(optarg); break;
            case 'l': o.nmf_opts.max_iter = atoi(optarg); break;
            case 'm': {
                int p = atoi(optarg);
                if (p <= 0) p = std::numeric_limits<float>::max_digits10;
                else if (p >= std::numeric_limits<double>::max_digits10) p = std::numeric_limits<double>::max_digits10;
                o.output_precision = p;
                break;
            }
            case 'n': user_max_threads = atoi(optarg); break;
            case 'o': o.nmf_opts.normalize = (0 != atoi(optarg)); break;
            case 'p': o.nmf_opts.verbose = (0 != atoi(optarg)); break;
            case 'q': o.show_help = true; break;
            case 'r': o.storage = (upper(optarg) == "BF16") ? SMK_STORE_BF16 : SMK_STORE_F32; break;
            case 's': o.seed = atoll(optarg); break;
            case ':': std::cerr << "missing argument for option " << argv[optind - 1] << std::endl; return false;
            default: std::cerr << "invalid option: " << argv[optind - 1] << std::endl; return false;
        }
    }
    if (1 == argc) o.show_help = true;
    if (o.show_help) return false;
    int hw = (int)std::thread::hardware_concurrency();
    if (hw <= 0) hw = 2;
    if (user_max_threads <= 0) user_max_threads = hw;
    o.nmf_opts.max_threads = std::min(user_max_threads, hw);
    if (o.infile_A.empty()) { std::cerr << "required command line argument --matrixfile not found" << std::endl; return false; }
    if (0 == o.nmf_opts.k && NmfAlgorithm::RANK2 != o.nmf_opts.algorithm) {
        std::cerr << "required command line argument --k not found" << std::endl;
        return false;
    }
    if (NmfAlgorithm::RANK2 == o.nmf_opts.algorithm && 2 != o.nmf_opts.k) {
        std::cerr << "warning: forcing k=2 for RANK2 algorithm" << std::endl;
        o.nmf_opts.k = 2;
    }
    return true;
}

bool has_ext(const std::string& path, const char* ext)
{
    size_t dot = path.find_last_of('.');
    if (dot == std::string::npos) return false;
    std::string e = path.substr(dot + 1);
    std::transform(e.begin(), e.end(), e.begin(), ::toupper);
    return e == ext;
}

bool load_csv(const std::string& path, std::vector<double>& buf, unsigned& h, unsigned& w)
{
    // two passes through the C ABI helper: sizes, then data
    double dummy;
    int rc = smk_load_csv(path.c_str(), &dummy, 0, &h, &w);
    if (rc == 0) return false;
    buf.assign((size_t)h * w, 0.0);
    return smk_load_csv(path.c_str(), buf.data(), (unsigned long)buf.size(), &h, &w) == 1;
}

}  // namespace

int main(int argc, char* argv[])
{
    CommandLineOptions opts;
    if (!ParseCommandLine(argc, argv, opts)) { ShowHelp(argv[0]); return opts.show_help ? 0 : -1; }
    if (!IsValid(opts.nmf_opts, false)) { ShowHelp(argv[0]); return -1; }

    NmfInitialize(argc, argv);
    NmfSetDeviceStorage(opts.storage);

    std::vector<double> buf_a, buf_w, buf_h, sp_data;
    std::vector<unsigned> sp_rows, sp_cols;
    unsigned m = 0, n = 0, nnz = 0;
    bool sparse = false;
    std::cout << "Loading matrix..." << std::endl;
    if (has_ext(opts.infile_A, "MTX")) {
        if (smk_load_matrix_market(opts.infile_A.c_str(), &m, &n, &nnz, nullptr, nullptr, nullptr) != 1) {
            std::cerr << "\nload failed for file " << opts.infile_A << std::endl;
            NmfFinalize();
            return -1;
        }
        sp_cols.resize((size_t)n + 1); sp_rows.resize(nnz); sp_data.resize(nnz);
        smk_load_matrix_market(opts.infile_A.c_str(), &m, &n, &nnz, sp_cols.data(), sp_rows.data(), sp_data.data());
        sparse = true;
    } else if (has_ext(opts.infile_A, "CSV")) {
        if (!load_csv(opts.infile_A, buf_a, m, n)) {
            std::cerr << "\nload failed for file " << opts.infile_A << std::endl;
            NmfFinalize();
            return -1;
        }
    } else {
        std::cerr << "\nInvalid matrix file: " << opts.infile_A << std::endl;
        NmfFinalize();
        return -1;
    }
    opts.nmf_opts.height = (int)m;
    opts.nmf_opts.width = (int)n;
    const unsigned k = (unsigned)opts.nmf_opts.k;
    if (!IsValid(opts.nmf_opts, true)) { NmfFinalize(); return -1; }

    // W and H: files or uniform [0,1) (RANDOM_MATRIX center 0.5 radius 0.5, nmf/src/main.cpp:37-38)
    unsigned long long seed = opts.seed >= 0 ? (unsigned long long)opts.seed
                                             : (unsigned long long)std::chrono::high_resolution_clock::now().time_since_epoch().count();
    unsigned hw_ = m, ww_ = k, hh_ = k, wh_ = n;
    if (opts.nmf_opts.verbose) std::cout << "Initializing matrix W..." << std::endl;
    if (opts.infile_W.empty()) {
        buf_w.resize((size_t)m * k);
        smk_uniform_fill_host(buf_w.data(), m, m, k, 0, 0, m, seed, 0);
    } else if (!load_csv(opts.infile_W, buf_w, hw_, ww_) || hw_ != m || ww_ != k) {
        std::cerr << "\tdimensions of matrix W are " << hw_ << " x " << ww_ << "\n\texpected " << m << " x " << k << std::endl;
        NmfFinalize();
        return -1;
    }
    if (opts.nmf_opts.verbose) std::cout << "Initializing matrix H..." << std::endl;
    if (opts.infile_H.empty()) {
        buf_h.resize((size_t)k * n);
        smk_uniform_fill_host(buf_h.data(), k, k, n, 0, 0, k, seed + 1, 0);
    } else if (!load_csv(opts.infile_H, buf_h, hh_, wh_) || hh_ != k || wh_ != n) {
        std::cerr << "\tdimensions of matrix H are " << hh_ << " x " << wh_ << "\n\texpected " << k << " x " << n << std::endl;
        NmfFinalize();
        return -1;
    }

    NmfStats stats;
    Result result;
    try {
        if (sparse)
            result = NmfSparse(opts.nmf_opts, m, n, nnz, sp_cols.data(), sp_rows.data(), sp_data.data(), buf_w.data(),
                               (int)m, buf_h.data(), (int)k, stats);
        else
            result = Nmf(opts.nmf_opts, buf_a.data(), (int)m, buf_w.data(), (int)m, buf_h.data(), (int)k, stats);
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        NmfFinalize();
        return -1;
    }
    if (opts.nmf_opts.verbose)
        std::cout << "Elapsed wall clock time: " << stats.elapsed_us / 1000.0 << " ms. (" << stats.iteration_count
                  << " iterations)" << std::endl;
    if (Result::OK != result) {
        std::cerr << "NMF solver failure (Result " << (int)result << ")" << std::endl;
        NmfFinalize();
        return -1;
    }
    if (opts.nmf_opts.verbose) std::cout << "Writing output files..." << std::endl;
    if (!smk_write_csv(buf_w.data(), m, m, k, opts.outfile_W.c_str(), (unsigned)opts.output_precision))
        std::cerr << "\terror writing output file " << opts.outfile_W << std::endl;
    if (!smk_write_csv(buf_h.data(), k, k, n, opts.outfile_H.c_str(), (unsigned)opts.output_precision))
        std::cerr << "\terror writing output file " << opts.outfile_H << std::endl;
    NmfFinalize();
    return 0;
}
