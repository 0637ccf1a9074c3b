// smallk_amd/csrc/hierclust_main.cpp -- the `hierclust` command line tool on the MI355X HierNMF2.
// Same flags, defaults and flow as the reference CLI (hierclust/src/command_line.cpp:37-58,148-357;
// hierclust/src/main.cpp:45-264): load the dictionary and A (.mtx sparse / .csv dense), run
// Clust / ClustSparse, write assignments_N.csv and tree_N.{xml,json} (+ the flat-clustering files
// with --flat 1).  Extensions: --storage f32|bf16, --seed N.
#include <getopt.h>

#include <chrono>
#include <cstdlib>
#include <limits>
#include <sstream>

#include "cli_common.h"

namespace {

struct Options {
    smk_clust_options c;
    std::string infile_A, dictfile, outdir, treefile, assignfile, initdir;
    bool show_help = false;
    int format = 0;             // 0 XML (default), 1 JSON
    int storage = SMK_STORE_F32;
    long long seed = -1;
};

option longopts[] = {{"matrixfile", required_argument, nullptr, 'a'}, {"dictfile", required_argument, nullptr, 'b'},
                     {"clusters", required_argument, nullptr, 'c'},   {"tol", required_argument, nullptr, 'd'},
                     {"outdir", required_argument, nullptr, 'e'},     {"miniter", required_argument, nullptr, 'f'},
                     {"maxiter", required_argument, nullptr, 'g'},    {"help", no_argument, nullptr, 'h'},
                     {"trial_allowance", required_argument, nullptr, 'i'}, {"unbalanced", required_argument, nullptr, 'j'},
                     {"verbose", required_argument, nullptr, 'k'},    {"maxthreads", required_argument, nullptr, 'l'},
                     {"maxterms", required_argument, nullptr, 'm'},   {"initdir", required_argument, nullptr, 'n'},
                     {"treefile", required_argument, nullptr, 'q'},   {"assignfile", required_argument, nullptr, 'r'},
                     {"flat", required_argument, nullptr, 's'},       {"format", required_argument, nullptr, 't'},
                     {"storage", required_argument, nullptr, 'u'},    {"seed", required_argument, nullptr, 'v'},
                     {nullptr, 0, nullptr, 0}};

void ShowHelp(const std::string& prog)
{
    std::cout << "\nUsage: " << prog << "\n"
              << "        --matrixfile <filename>     Filename of the matrix to be factored.\n"
              << "                                    Either CSV format for dense or MatrixMarket format for sparse.\n"
              << "        --dictfile <filename>       The name of the dictionary file.\n"
              << "        --clusters <integer>        The number of clusters to generate.\n"
              << "        [--initdir  (empty)]        Directory of initializers for all Rank2 factorizations.\n"
              << "                                    If unspecified, random init will be used. \n"
              << "        [--tol  0.0001]             Tolerance value for each factorization. \n"
              << "        [--outdir  (empty)]         Output directory.  If unspecified, results will be \n"
              << "                                    written to the current directory.\n"
              << "        [--miniter  5]              Minimum number of iterations to perform.\n"
              << "        [--maxiter  5000]           Maximum number of  iterations to perform. \n"
              << "        [--maxterms  5]             Number of terms per node. \n"
              << "        [--maxthreads    N]         Upper limit to thread count (host side only). \n"
              << "        [--unbalanced  0.1]         Threshold for determining leaf node imbalance. \n"
              << "        [--trial_allowance  3]      Number of split attempts. \n"
              << "        [--flat  0]                 Whether to generate a flat clustering result. \n"
              << "                                        1 == yes, 0 == no\n"
              << "        [--verbose  1]              Whether to print updates to the screen.\n"
              << "                                        1 == yes, 0 == no\n"
              << "        [--format  XML]             Format of the output file containing the tree.\n"
              << "                                        XML: XML format\n"
              << "                                        JSON: JavaScript Object Notation\n"
              << "        [--treefile  tree_N.ext]    Name of the output file containing the tree.\n"
              << "                                    N is the number of clusters for this run.\n"
              << "                                    The string 'ext' depends on the desired format.\n"
              << "                                    This filename is relative to the outdir.\n"
              << "        [--assignfile assignments_N.csv]  Name of the file containing final assignments.\n"
              << "                                          N is the number of clusters for this run.\n"
              << "                                          This filename is relative to the outdir.\n"
              << "        [--storage  f32]            MI355X: hold a dense matrix in HBM as f32 or bf16.\n"
              << "        [--seed  (time)]            MI355X: seed of the random initializers.\n"
              << std::endl;
}

bool ParseCommandLine(int argc, char* argv[], Options& o)
{
    o.c.nmf.height = o.c.nmf.width = o.c.nmf.k = 0;
    o.c.nmf.min_iter = 5;
    o.c.nmf.max_iter = 5000;
    o.c.nmf.tol = 0.0001;
    o.c.nmf.tolcount = 1;
    o.c.nmf.verbose = 0;                  // nmf is silent
    o.c.nmf.normalize = 0;                // rank2 normalizes on each iter
    o.c.nmf.algorithm = SMK_ALG_RANK2;
    o.c.nmf.prog_est_algorithm = SMK_PROG_PG_RATIO;
    o.c.maxterms = 5;
    o.c.trial_allowance = 3;
    o.c.unbalanced = 0.1;
    o.c.num_clusters = 0;
    o.c.verbose = 1;
    o.c.flat = 0;
    int user_max_threads = -1, c, index;
    while (-1 != (c = getopt_long(argc, argv, ":a:b:c:d:e:f:g:hi:j:k:l:m:n:q:r:s:t:u:v:", longopts, &index))) {
        std::string tmp;
        switch (c) {
            case 'a': o.infile_A = optarg; break;
            case 'b': o.dictfile = optarg; break;
            case 'c': o.c.num_clusters = atoi(optarg); break;
            case 'd': o.c.nmf.tol = atof(optarg); break;
            case 'e': o.outdir = optarg; break;
            case 'f': o.c.nmf.min_iter = atoi(optarg); break;
            case 'g': o.c.nmf.max_iter = atoi(optarg); break;
            case 'h': o.show_help = true; break;
            case 'i': o.c.trial_allowance = atoi(optarg); break;
            case 'j': o.c.unbalanced = atof(optarg); break;
            case 'k': o.c.verbose = (0 != atoi(optarg)); break;
            case 'l': user_max_threads = atoi(optarg); break;
            case 'm': o.c.maxterms = atoi(optarg); break;
            case 'n': o.initdir = optarg; break;
            case 'q': o.treefile = optarg; break;
            case 'r': o.assignfile = optarg; break;
            case 's': o.c.flat = (0 != atoi(optarg)); break;
            case 't':
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
    o.c.nmf.max_threads = std::min(user_max_threads, hw);
    if (o.infile_A.empty()) { std::cerr << "required command line argument --matrixfile not found" << std::endl; return false; }
    if (o.dictfile.empty()) { std::cerr << "required command line argument --dictfile not found" << std::endl; return false; }
    if (0 == o.c.num_clusters) { std::cerr << "required command line argument --clusters not found" << std::endl; return false; }
    if (!o.initdir.empty()) o.initdir = cli::ensure_trailing_sep(o.initdir);
    const std::string output_dir = cli::ensure_trailing_sep(o.outdir);
    std::ostringstream a, t;
    a << "assignments_" << o.c.num_clusters << ".csv";
    t << "tree_" << o.c.num_clusters << (o.format ? ".json" : ".xml");
    o.assignfile = output_dir + (o.assignfile.empty() ? a.str() : o.assignfile);
    o.treefile = output_dir + (o.treefile.empty() ? t.str() : o.treefile);
    return true;
}

bool IsValid(const Options& o)
{
    if (!o.outdir.empty() && !cli::directory_exists(o.outdir)) {
        std::cerr << "the specified output directory \"" << o.outdir << "\" does not exist" << std::endl;
        return false;
    }
    if (!o.initdir.empty() && !cli::directory_exists(o.initdir)) {
        std::cerr << "the specified init directory \"" << o.initdir << "\" does not exist" << std::endl;
        return false;
    }
    return smk_clust_is_valid(&o.c, 0) != 0;
}

void PrintOpts(const Options& o)
{
    using std::cout; using std::endl;
    cout << "\n     Command line options: \n" << endl;
    cout << "\t            height: " << o.c.nmf.height << endl;
    cout << "\t             width: " << o.c.nmf.width << endl;
    cout << "\t        matrixfile: " << o.infile_A << endl;
    cout << "\t           initdir: " << o.initdir << endl;
    cout << "\t          dictfile: " << o.dictfile << endl;
    cout << "\t        assignfile: " << o.assignfile << endl;
    cout << "\t            format: " << (o.format ? "JSON" : "XML") << endl;
    cout << "\t          treefile: " << o.treefile << endl;
    cout << "\t          clusters: " << o.c.num_clusters << endl;
    cout << "\t               tol: " << o.c.nmf.tol << endl;
    cout << "\t            outdir: " << o.outdir << endl;
    cout << "\t           miniter: " << o.c.nmf.min_iter << endl;
    cout << "\t           maxiter: " << o.c.nmf.max_iter << endl;
    cout << "\t          maxterms: " << o.c.maxterms << endl;
    cout << "\t        maxthreads: " << o.c.nmf.max_threads << endl;
    cout << "\t        unbalanced: " << o.c.unbalanced << endl;
    cout << "\t   trial_allowance: " << o.c.trial_allowance << endl;
    cout << "\t              flat: " << o.c.flat << endl;
    cout << "\t           verbose: " << o.c.verbose << endl;
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

    if (opts.c.verbose) std::cout << "loading dictionary..." << std::endl;
    std::vector<std::string> dictionary;
    if (!cli::load_strings(opts.dictfile, dictionary)) {
        std::cerr << "\ncould not load dictionary file " << opts.dictfile << std::endl;
        smk_finalize();
        return -1;
    }
    if (opts.c.verbose) std::cout << "loading matrix..." << std::endl;
    cli::InputMatrix A;
    const int lrc = cli::load_matrix(opts.infile_A, A);
    if (lrc != 0) {
        std::cerr << (lrc == -2 ? "\nunsupported file type: " : "\nload failed for file ") << opts.infile_A << std::endl;
        smk_finalize();
        return -1;
    }
    const unsigned m = A.m, n = A.n, num_clusters = (unsigned)opts.c.num_clusters;
    if (2ull * m > (unsigned long long)std::numeric_limits<int>::max()) { std::cerr << "W matrix size too large" << std::endl; smk_finalize(); return -1; }
    if (2ull * n > (unsigned long long)std::numeric_limits<int>::max()) { std::cerr << "H matrix size too large" << std::endl; smk_finalize(); return -1; }
    opts.c.nmf.height = (int)m;
    opts.c.nmf.width = (int)n;
    opts.c.nmf.k = 2;
    if (opts.c.verbose) PrintOpts(opts);

    const uint64_t seed = opts.seed >= 0 ? (uint64_t)opts.seed
                                         : (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count();
    uint64_t draws = 0;
    smk_tree* tree = nullptr;
    smk_clust_stats stats = {0, 0};
    const auto t0 = std::chrono::high_resolution_clock::now();
    int result;
    if (A.sparse)
        result = smk_clust_sparse(&opts.c, A.nnz, A.cols.data(), A.rows.data(), A.data.data(), seed, &draws,
                                  opts.initdir.empty() ? nullptr : opts.initdir.c_str(), &tree, &stats);
    else
        result = smk_clust_dense(&opts.c, A.dense.data(), m, opts.storage, seed, &draws,
                                 opts.initdir.empty() ? nullptr : opts.initdir.c_str(), &tree, &stats);

    std::vector<double> buf_w, buf_h;
    std::vector<float> probabilities;
    std::vector<unsigned> assignments_flat;
    std::vector<int> term_indices((size_t)opts.c.maxterms * num_clusters, 0);
    bool have_flat = false;
    if (opts.c.flat && result == SMK_OK) {
        buf_w.assign((size_t)m * num_clusters, 0.0);
        buf_h.assign((size_t)num_clusters * n, 0.0);
        probabilities.assign((size_t)num_clusters * n, 0.f);
        assignments_flat.assign(n, 0u);
        have_flat = smk_tree_flat_factors(tree, buf_w.data(), m, buf_h.data(), num_clusters) == SMK_OK &&
                    smk_compute_fuzzy_assignments(buf_h.data(), num_clusters, num_clusters, n, probabilities.data()) == SMK_OK &&
                    smk_compute_assignments(buf_h.data(), num_clusters, num_clusters, n, assignments_flat.data()) == SMK_OK &&
                    smk_top_terms(opts.c.maxterms, buf_w.data(), m, m, num_clusters, term_indices.data()) == SMK_OK;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    std::cout << "\nElapsed wall clock time: " << cli::elapsed_ms_string(ms) << std::endl;
    std::cout << (stats.nmf_count - stats.max_count) << "/" << stats.nmf_count << " factorizations converged.\n" << std::endl;

    int exit_code = 0;
    if (!tree) {
        std::cerr << "\nHierarchical clustering fatal error." << std::endl;
        if (smk_last_error()[0]) std::cerr << smk_last_error() << std::endl;
        exit_code = (result == SMK_FAILURE) ? 0 : -1;      // the reference returns 0 after a solver failure
    } else {
        if (opts.c.verbose) std::cout << "Writing output files..." << std::endl;
        if (smk_tree_write_assignments(tree, opts.assignfile.c_str()) != SMK_OK)
            std::cerr << "\terror writing assignments file" << std::endl;
        std::vector<const char*> terms(dictionary.size());
        for (size_t i = 0; i < dictionary.size(); ++i) terms[i] = dictionary[i].c_str();
        if (smk_tree_write(tree, opts.treefile.c_str(), opts.format, terms.data(), (int64_t)terms.size()) != SMK_OK)
            std::cerr << "\terror writing factorization file" << std::endl;
        if (have_flat) {
            const std::string od = cli::ensure_trailing_sep(opts.outdir);
            std::ostringstream fa, ff, fr;
            fa << od << "assignments_flat_" << num_clusters << ".csv";
            ff << od << "assignments_fuzzy_" << num_clusters << ".csv";
            fr << od << "clusters_" << num_clusters << (opts.format ? ".json" : ".xml");
            if (smk_flatclust_write_results(fa.str().c_str(), ff.str().c_str(), fr.str().c_str(), assignments_flat.data(), n,
                                            probabilities.data(), terms.data(), (int64_t)terms.size(), term_indices.data(),
                                            (int64_t)term_indices.size(), opts.format, (unsigned)opts.c.maxterms, n,
                                            num_clusters) != SMK_OK)
                std::cerr << "\terror writing flat clustering results: " << smk_last_error() << std::endl;
        }
        smk_tree_destroy(tree);
    }
    smk_finalize();
    return exit_code;
}
