// smallk_amd/csrc/flatclust_main.cpp -- the `flatclust` command line tool on the MI355X solvers.
// Same flags, defaults and flow as the reference CLI (flatclust/src/command_line.cpp:36-57,150-442;
// flatclust/src/main.cpp:43-294): load the dictionary and A, initialise W and H (files or RNG), run
// FlatClust / FlatClustSparse (HALS, RANK2 or BPP), then write assignments_N.csv,
// assignments_fuzzy_N.csv and clusters_N.{xml,json}.  Extensions: --storage f32|bf16, --seed N.
#include <getopt.h>

#include <chrono>
#include <cstdlib>
#include <limits>
#include <sstream>

#include "cli_common.h"

namespace {

struct Options {
    smk_options nmf;
    int maxterms = 5, num_clusters = 0, verbose = 1;
    std::string infile_A, infile_W, infile_H, dictfile, outdir, clustfile, assignfile, fuzzyfile;
    bool show_help = false;
    int format = 0;
    int storage = SMK_STORE_F32;
    long long seed = -1;
};

option longopts[] = {{"matrixfile", required_argument, nullptr, 'a'}, {"dictfile", required_argument, nullptr, 'b'},
                     {"clusters", required_argument, nullptr, 'c'},   {"tol", required_argument, nullptr, 'd'},
                     {"outdir", required_argument, nullptr, 'e'},     {"miniter", required_argument, nullptr, 'f'},
                     {"maxiter", required_argument, nullptr, 'g'},    {"help", no_argument, nullptr, 'h'},
                     {"algorithm", required_argument, nullptr, 'i'},  {"verbose", required_argument, nullptr, 'k'},
                     {"maxthreads", required_argument, nullptr, 'l'}, {"maxterms", required_argument, nullptr, 'm'},
                     {"infile_W", required_argument, nullptr, 'n'},   {"infile_H", required_argument, nullptr, 'o'},
                     {"clustfile", required_argument, nullptr, 'q'},  {"assignfile", required_argument, nullptr, 'r'},
                     {"format", required_argument, nullptr, 's'},     {"fuzzyfile", required_argument, nullptr, 't'},
                     {"storage", required_argument, nullptr, 'u'},    {"seed", required_argument, nullptr, 'v'},
                     {nullptr, 0, nullptr, 0}};

void ShowHelp(const std::string& prog)
{
    std::cout << "\nUsage: " << prog << "\n"
              << "        --matrixfile <filename>      Filename of the matrix to be factored.\n"
              << "                                     Either CSV format for dense or MatrixMarket format for sparse.\n"
              << "        --dictfile <filename>        The name of the dictionary file.\n"
              << "        --clusters <integer>         The number of clusters to generate.\n"
              << "        [--algorithm  BPP]           The NMF algorithm to use: \n"
              << "                                         HALS:  hierarchical alternating least squares\n"
              << "                                         RANK2: rank2 with optimal active set selection\n"
              << "                                                (for two clusters only)\n"
              << "                                         BPP:   block principal pivoting\n"
              << "        [--infile_W  (empty)]        Dense matrix to initialize W, CSV file.\n"
              << "                                     The matrix has m rows and 'clusters' columns.\n"
              << "                                     If unspecified, W will be randomly initialized.\n"
              << "        [--infile_H  (empty)]        Dense matrix to initialize H, CSV file. \n"
              << "                                     The matrix has 'clusters' rows and n columns.\n"
              << "                                     If unspecified, H will be randomly initialized. \n"
              << "        [--tol  0.0001]              Tolerance value for the progress metric. \n"
              << "        [--outdir  (empty)]          Output directory.  If unspecified, results will be \n"
              << "                                     written to the current directory.\n"
              << "        [--miniter  5]               Minimum number of iterations to perform.\n"
              << "        [--maxiter  5000]            Maximum number of  iterations to perform. \n"
              << "        [--maxterms  5]              Number of terms per node. \n"
              << "        [--maxthreads    N]          Upper limit to thread count (host side only). \n"
              << "        [--verbose  1]               Whether to print updates to the screen.\n"
              << "                                         1 == yes, 0 == no\n"
              << "        [--format  XML]              Format of the output file containing the tree.\n"
              << "                                         XML: XML format\n"
              << "                                         JSON: JavaScript Object Notation\n"
              << "        [--clustfile clusters_N.ext] Name of the output XML file containing the tree.\n"
              << "                                     N is the number of clusters for this run.\n"
              << "                                     The string 'ext' depends on the desired format.\n"
              << "                                     This filename is relative to the outdir.\n"
              << "        [--assignfile assignments_N.csv]  Name of the file containing final assignments.\n"
              << "                                          N is the number of clusters for this run.\n"
              << "                                          This filename is relative to the outdir.\n"
              << "        [--fuzzyfile assignments_fuzzy_N.csv] Name of fuzzy assignment file.\n"
              << "                                              N is the number of clusters for this run.\n"
              << "                                              This filename is relative to the outdir.\n"
              << "        [--storage  f32]             MI355X: hold a dense matrix in HBM as f32 or bf16.\n"
              << "        [--seed  (time)]             MI355X: seed of the random initializers.\n"
              << std::endl;
}

bool ParseCommandLine(int argc, char* argv[], Options& o)
{
    o.nmf.height = o.nmf.width = o.nmf.k = 0;
    o.nmf.min_iter = 5;
    o.nmf.max_iter = 5000;
    o.nmf.tol = 0.0001;
    o.nmf.tolcount = 1;
    o.nmf.verbose = 1;
    o.nmf.normalize = 1;
    o.nmf.algorithm = SMK_ALG_BPP;
    o.nmf.prog_est_algorithm = SMK_PROG_PG_RATIO;
    int user_max_threads = -1, c, index;
    while (-1 != (c = getopt_long(argc, argv, ":a:b:c:d:e:f:g:hi:k:l:m:n:o:q:r:s:t:u:v:", longopts, &index))) {
        std::string tmp;
        switch (c) {
            case 'a': o.infile_A = optarg; break;
            case 'b': o.dictfile = optarg; break;
            case 'c': o.num_clusters = atoi(optarg); o.nmf.k = atoi(optarg); break;
            case 'd': o.nmf.tol = atof(optarg); break;
            case 'e': o.outdir = optarg; break;
            case 'f': o.nmf.min_iter = atoi(optarg); break;
            case 'g': o.nmf.max_iter = atoi(optarg); break;
            case 'h': o.show_help = true; break;
            case 'i':
                tmp = cli::upper(optarg);
                if (tmp == "HALS") o.nmf.algorithm = SMK_ALG_HALS;
                else if (tmp == "RANK2") o.nmf.algorithm = SMK_ALG_RANK2;
                else if (tmp == "BPP") o.nmf.algorithm = SMK_ALG_BPP;
                else { std::cerr << "Invalid value specified for command-line argument " << tmp << std::endl; return false; }
                break;
            case 'k': o.verbose = (0 != atoi(optarg)); break;
            case 'l': user_max_threads = atoi(optarg); break;
            case 'm': o.maxterms = atoi(optarg); break;
            case 'n': o.infile_W = optarg; break;
            case 'o': o.infile_H = optarg; break;
            case 'q': o.clustfile = optarg; break;
            case 'r': o.assignfile = optarg; break;
            case 't': o.fuzzyfile = optarg; break;
            case 's':
                tmp = cli::upper(optarg);
                if (tmp == "XML") o.format = 0;
                else if (tmp == "JSON") o.format = 1;
                else { std::cerr << "Invalid value specified for command-line argument " << tmp << std::endl; return false; }
                break;
            case 'u': o.storage = (cli::upper(optarg) == "BF16") ? SMK_STORE_BF16 : SMK_STORE_F32; break;
            case 'v': o.seed = atoll(optarg); break;
            case ':': std::cerr << "missing argument for option " << argv[optind - 1] << std::endl; return false;
            default: std::cerr << "invalid option: " << argv[optind - 1] << std::endl; return false;
        }
    }
    if (1 == argc) o.show_help = true;
    if (o.show_help) return false;
    const int hw = cli::hw_threads();
    if (user_max_threads <= 0) user_max_threads = hw;
    o.nmf.max_threads = std::min(user_max_threads, hw);
    if (!o.verbose) o.nmf.verbose = 0;
    if (o.infile_A.empty()) { std::cerr << "required command line argument --matrixfile not found" << std::endl; return false; }
    if (o.dictfile.empty()) { std::cerr << "required command line argument --dictfile not found" << std::endl; return false; }
    if (0 == o.num_clusters) { std::cerr << "required command line argument --clusters not found" << std::endl; return false; }
    const std::string od = cli::ensure_trailing_sep(o.outdir);
    std::ostringstream a, f, r;
    a << "assignments_" << o.num_clusters << ".csv";
    f << "assignments_fuzzy_" << o.num_clusters << ".csv";
    r << "clusters_" << o.num_clusters << (o.format ? ".json" : ".xml");
    o.assignfile = od + (o.assignfile.empty() ? a.str() : o.assignfile);
    o.fuzzyfile = od + (o.fuzzyfile.empty() ? f.str() : o.fuzzyfile);
    o.clustfile = od + (o.clustfile.empty() ? r.str() : o.clustfile);
    if (o.nmf.algorithm == SMK_ALG_RANK2 && o.num_clusters != 2) {
        if (o.verbose) std::cout << "\nwarning: forcing clusters=2 for RANK2 algorithm" << std::endl;
        o.num_clusters = 2;
        o.nmf.k = 2;
    }
    return true;
}

bool IsValid(const Options& o)
{
    using std::cerr; using std::endl;
    if (!o.outdir.empty() && !cli::directory_exists(o.outdir)) {
        cerr << "the specified output directory \"" << o.outdir << "\" does not exist" << endl;
        return false;
    }
    if (o.num_clusters <= 0) { cerr << "value for --clusters must be a positive integer" << endl; return false; }
    if (o.nmf.tol <= 0.0 || o.nmf.tol >= 1.0) { cerr << "tolerance must be in the interval (0.0, 1.0)" << endl; return false; }
    if (o.nmf.min_iter <= 0) { cerr << "miniter must be a positive integer" << endl; return false; }
    if (o.nmf.max_iter <= 0) { cerr << "maxiter must be a positive integer" << endl; return false; }
    if (o.maxterms <= 0) { cerr << "maxterms must be a positive integer" << endl; return false; }
    return true;
}

void PrintOpts(const Options& o)
{
    using std::cout; using std::endl;
    cout << "\n     Command line options: \n" << endl;
    cout << "\t            height: " << o.nmf.height << endl;
    cout << "\t             width: " << o.nmf.width << endl;
    cout << "\t        matrixfile: " << o.infile_A << endl;
    cout << "\t          infile_W: " << o.infile_W << endl;
    cout << "\t          infile_H: " << o.infile_H << endl;
    cout << "\t          dictfile: " << o.dictfile << endl;
    cout << "\t        assignfile: " << o.assignfile << endl;
    cout << "\t         fuzzyfile: " << o.fuzzyfile << endl;
    cout << "\t            format: " << (o.format ? "JSON" : "XML") << endl;
    cout << "\t         clustfile: " << o.clustfile << endl;
    cout << "\t         algorithm: "
         << (o.nmf.algorithm == SMK_ALG_HALS ? "HALS" : o.nmf.algorithm == SMK_ALG_RANK2 ? "Rank 2"
                                                       : "Nonnegative Least Squares with Block Principal Pivoting")
         << endl;
    cout << "\t          clusters: " << o.num_clusters << endl;
    cout << "\t               tol: " << o.nmf.tol << endl;
    cout << "\t            outdir: " << o.outdir << endl;
    cout << "\t           miniter: " << o.nmf.min_iter << endl;
    cout << "\t           maxiter: " << o.nmf.max_iter << endl;
    cout << "\t          maxterms: " << o.maxterms << endl;
    cout << "\t        maxthreads: " << o.nmf.max_threads << endl;
    cout << "\t           verbose: " << o.verbose << endl;
    cout << endl;
}

}  // namespace

int main(int argc, char* argv[])
{
    Options opts;
    if (!ParseCommandLine(argc, argv, opts)) {
        if (opts.show_help) { ShowHelp(argv[0]); return 0; }
        return -1;
    }
    if (!IsValid(opts)) return -1;
    if (smk_initialize(-1) != SMK_OK) { std::cerr << smk_last_error() << std::endl; return -1; }

    if (opts.verbose) std::cout << "loading dictionary..." << std::endl;
    std::vector<std::string> dictionary;
    if (!cli::load_strings(opts.dictfile, dictionary)) {
        std::cerr << "\ncould not load dictionary file " << opts.dictfile << std::endl;
        smk_finalize();
        return -1;
    }
    if (opts.verbose) std::cout << "loading matrix..." << std::endl;
    cli::InputMatrix A;
    const int lrc = cli::load_matrix(opts.infile_A, A);
    if (lrc != 0) {
        std::cerr << (lrc == -2 ? "\nunsupported file type: " : "\nload failed for file ") << opts.infile_A << std::endl;
        smk_finalize();
        return -1;
    }
    const unsigned m = A.m, n = A.n, k = (unsigned)opts.nmf.k;
    opts.nmf.height = (int)m;
    opts.nmf.width = (int)n;
    const unsigned long long lim = (unsigned long long)std::numeric_limits<int>::max();
    if ((unsigned long long)m * k > lim) { std::cerr << "W matrix size too large" << std::endl; smk_finalize(); return -1; }
    if ((unsigned long long)n * k > lim) { std::cerr << "H matrix size too large" << std::endl; smk_finalize(); return -1; }

    const uint64_t seed = opts.seed >= 0 ? (uint64_t)opts.seed
                                         : (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count();
    std::vector<double> buf_w, buf_h;
    unsigned hw_ = m, ww_ = k, hh_ = k, wh_ = n;
    if (opts.verbose) std::cout << "Initializing matrix W..." << std::endl;
    if (opts.infile_W.empty()) {
        buf_w.resize((size_t)m * k);
        smk_uniform_fill_host(buf_w.data(), m, m, k, 0, 0, m, seed, 0);
    } else if (!cli::load_csv(opts.infile_W, buf_w, hw_, ww_)) {
        std::cerr << "\nload failed for file " << opts.infile_W << std::endl;
        smk_finalize();
        return -1;
    }
    if (hw_ != m || ww_ != k) {
        std::cerr << "\tdimensions of matrix W are " << hw_ << " x " << ww_ << "\n\texpected " << m << " x " << k << std::endl;
        smk_finalize();
        return -1;
    }
    if (opts.verbose) std::cout << "Initializing matrix H..." << std::endl;
    if (opts.infile_H.empty()) {
        buf_h.resize((size_t)k * n);
        smk_uniform_fill_host(buf_h.data(), k, k, n, 0, 0, k, seed + 1, 0);
    } else if (!cli::load_csv(opts.infile_H, buf_h, hh_, wh_)) {
        std::cerr << "\nload failed for file " << opts.infile_H << std::endl;
        smk_finalize();
        return -1;
    }
    if (hh_ != k || wh_ != n) {
        std::cerr << "\tdimensions of matrix H are " << hh_ << " x " << wh_ << "\n\texpected " << k << " x " << n << std::endl;
        smk_finalize();
        return -1;
    }
    if (opts.verbose) PrintOpts(opts);

    smk_stats stats = {0, 0};
    int result;
    if (A.sparse)
        result = smk_flatclust_sparse(&opts.nmf, m, n, A.nnz, A.cols.data(), A.rows.data(), A.data.data(), buf_w.data(), m,
                                      buf_h.data(), k, &stats);
    else
        result = smk_flatclust_dense(&opts.nmf, A.dense.data(), m, buf_w.data(), m, buf_h.data(), k, &stats, opts.storage);
    if (result != SMK_OK) {
        std::cerr << "\nNMF solver failure." << std::endl;
        if (smk_last_error()[0]) std::cerr << smk_last_error() << std::endl;
    } else {
        const auto t0 = std::chrono::high_resolution_clock::now();
        std::vector<float> probabilities((size_t)k * n);
        std::vector<unsigned> assignments(n);
        std::vector<int> term_indices((size_t)opts.maxterms * k, 0);
        std::vector<const char*> terms(dictionary.size());
        for (size_t i = 0; i < dictionary.size(); ++i) terms[i] = dictionary[i].c_str();
        int rc = smk_compute_fuzzy_assignments(buf_h.data(), k, k, n, probabilities.data());
        if (rc == SMK_OK) rc = smk_compute_assignments(buf_h.data(), k, k, n, assignments.data());
        if (rc == SMK_OK) rc = smk_top_terms(opts.maxterms, buf_w.data(), m, m, k, term_indices.data());
        if (rc == SMK_OK)
            rc = smk_flatclust_write_results(opts.assignfile.c_str(), opts.fuzzyfile.c_str(), opts.clustfile.c_str(),
                                             assignments.data(), n, probabilities.data(), terms.data(), (int64_t)terms.size(),
                                             term_indices.data(), (int64_t)term_indices.size(), opts.format,
                                             (unsigned)opts.maxterms, n, (unsigned)opts.num_clusters);
        if (rc != SMK_OK) std::cerr << "error writing results: " << smk_last_error() << std::endl;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count() +
                          stats.elapsed_us / 1000.0;
        if (opts.verbose) std::cout << "Elapsed wall clock time: " << cli::elapsed_ms_string(ms) << "\n" << std::endl;
    }
    smk_finalize();
    return 0;
}
