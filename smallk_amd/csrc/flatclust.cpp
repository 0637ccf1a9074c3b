// smallk_amd/csrc/flatclust.cpp -- flat clustering: FlatClust / FlatClustSparse and the result files.
//
// Reference: flatclust/src/flat_clust.cpp:118-264 (the same NmfSolve<> as Nmf(), restricted to
// HALS / RANK2 / BPP), common/include/assignments.hpp:32-113 (argmax and fuzzy assignments),
// common/include/terms.hpp:62-108 (top terms per column of W), common/src/assignments.cpp and
// common/src/flat_clust_output.cpp + flatclust_{json,xml}_writer.cpp (the three output files).
// The factorisation runs on the GPU through smk_nmf_dense / smk_nmf_sparse; what is left here is
// O((m + n) k) host post-processing of the returned factors and text output.
#include "common.h"
#include "../../include/smallk_amd.h"

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <map>
#include <numeric>
#include <string>
#include <vector>

using smk::set_error;

namespace {

int flat_precheck(const smk_options* o, int64_t ldW, int64_t ldH)
{
    if (!o) return SMK_BAD_PARAM;
    if (o->algorithm != SMK_ALG_HALS && o->algorithm != SMK_ALG_RANK2 && o->algorithm != SMK_ALG_BPP) {
        set_error("unknown NMF algorithm");          // RunFlatClust throws runtime_error (flat_clust.cpp:75-79)
        return SMK_BAD_PARAM;
    }
    if (o->algorithm == SMK_ALG_RANK2 && o->k != 2) { set_error("rank2 algorithm requires k == 2"); return SMK_BAD_PARAM; }
    if (ldW < o->height) { set_error("nmflib error: leading dimension of W return buffer too small"); return SMK_BAD_PARAM; }
    if (ldH < o->k) { set_error("nmflib error: leading dimension of H return buffer too small"); return SMK_BAD_PARAM; }
    return SMK_OK;
}

}  // namespace

extern "C" {

int smk_flatclust_dense(const smk_options* o, const double* A, int64_t ldA, double* W, int64_t ldW, double* H,
                        int64_t ldH, smk_stats* stats, int storage)
{
    if (smk_is_initialized() != SMK_INITIALIZED) {
        set_error("flatclust error: smk_initialize() must be called prior to any factorization routine");
        return SMK_NOTINITIALIZED;
    }
    if (!o || !smk_is_valid(o, 1)) return SMK_BAD_PARAM;
    const int rc = flat_precheck(o, ldW, ldH);
    if (rc != SMK_OK) return rc;
    return smk_nmf_dense(o, A, ldA, W, ldW, H, ldH, stats, storage);
}

int smk_flatclust_sparse(const smk_options* o, unsigned height, unsigned width, unsigned nz,
                         const unsigned* col_offsets, const unsigned* row_indices, const double* data, double* W,
                         int64_t ldW, double* H, int64_t ldH, smk_stats* stats)
{
    if (smk_is_initialized() != SMK_INITIALIZED) {
        set_error("flatclust error: smk_initialize() must be called prior to any factorization routine");
        return SMK_NOTINITIALIZED;
    }
    if (!o || !smk_is_valid(o, 1)) return SMK_BAD_PARAM;
    const int rc = flat_precheck(o, ldW, ldH);
    if (rc != SMK_OK) return rc;
    return smk_nmf_sparse(o, height, width, nz, col_offsets, row_indices, data, W, ldW, H, ldH, stats);
}

// ComputeAssignments, assignments.hpp:72-113: label of document c = row of the largest entry of
// column c of H (first one on ties)
int smk_compute_assignments(const double* H, unsigned ldH, unsigned k, unsigned n, unsigned* out)
{
    if (!H || !out || ldH < k || k == 0) return SMK_BAD_PARAM;
    if (k > n) { set_error("ComputeAssignments: dimensions of matrix H are invalid"); return SMK_BAD_PARAM; }
    for (unsigned c = 0; c < n; ++c) {
        const double* col = H + (size_t)c * ldH;
        unsigned best = 0;
        double mx = col[0];
        for (unsigned r = 1; r < k; ++r)
            if (col[r] > mx) { mx = col[r]; best = r; }
        out[c] = best;
    }
    return SMK_OK;
}

// ComputeFuzzyAssignments, assignments.hpp:32-69: column c of H scaled to sum 1, as float, stored at
// c*ldH + r
int smk_compute_fuzzy_assignments(const double* H, unsigned ldH, unsigned k, unsigned n, float* probabilities)
{
    if (!H || !probabilities || ldH < k || k == 0) return SMK_BAD_PARAM;
    for (unsigned c = 0; c < n; ++c) {
        const size_t off = (size_t)c * ldH;
        double sum = 0.0;
        for (unsigned r = 0; r < k; ++r) sum += H[off + r];
        const double inv = 1.0 / sum;
        for (unsigned r = 0; r < k; ++r) probabilities[off + r] = (float)(H[off + r] * inv);
    }
    return SMK_OK;
}

// TopTerms (buffer form), terms.hpp:62-108: for each column of W the row indices of its largest
// entries, maxterms slots per column (only min(maxterms, height) are filled).  Column c is read at
// c*height like the reference (ldim is accepted and, as there, must equal height).  Ties: lower
// index first.
int smk_top_terms(int maxterms, const double* W, unsigned ldim, unsigned height, unsigned width, int* term_indices)
{
    if (!W || !term_indices || maxterms <= 0) return SMK_BAD_PARAM;
    if (height < width) { set_error("TopTerms: height of W buffer must be >= width"); return SMK_BAD_PARAM; }
    (void)ldim;
    const unsigned cnt = std::min<unsigned>((unsigned)maxterms, height);
    std::vector<unsigned> order(height);
    for (unsigned c = 0; c < width; ++c) {
        const double* d = W + (size_t)c * height;
        std::iota(order.begin(), order.end(), 0u);
        std::partial_sort(order.begin(), order.begin() + cnt, order.end(),
                          [d](unsigned a, unsigned b) { return d[a] > d[b] || (d[a] == d[b] && a < b); });
        for (unsigned q = 0; q < cnt; ++q) term_indices[(size_t)c * maxterms + q] = (int)order[q];
    }
    return SMK_OK;
}

// WriteAssignmentsFile, common/src/assignments.cpp:23-41
int smk_write_assignments_file(const unsigned* labels, unsigned n, const char* path)
{
    if (!path || (n && !labels)) return 0;
    std::ofstream f(path);
    if (!f) return 0;
    if (n > 0) f << labels[0];
    for (unsigned i = 1; i < n; ++i) f << ',' << labels[i];
    f << std::endl;
    return 1;
}

// WriteFuzzyAssignmentsFile, common/src/assignments.cpp:44-70: one line per document, %.3e
int smk_write_fuzzy_assignments_file(const float* probabilities, unsigned k, unsigned n, const char* path)
{
    if (!path || !probabilities || k == 0) return 0;
    std::ofstream f(path);
    if (!f) return 0;
    for (unsigned c = 0; c < n; ++c) {
        const size_t off = (size_t)c * k;
        f << std::scientific << std::setprecision(3) << probabilities[off];
        for (unsigned r = 1; r < k; ++r) f << ',' << std::scientific << std::setprecision(3) << probabilities[off + r];
        f << std::endl;
    }
    return 1;
}

// FlatClustWriteResults, common/src/flat_clust_output.cpp:56-141 with the node writers of
// flatclust_json_writer.cpp / flatclust_xml_writer.cpp.  format 0 = XML, 1 = JSON.
int smk_flatclust_write_results(const char* assignfile, const char* fuzzyfile, const char* resultfile,
                                const unsigned* assignments, unsigned num_assignments, const float* probabilities,
                                const char* const* dictionary, int64_t dictionary_size, const int* term_indices,
                                int64_t num_term_indices, int format, unsigned maxterms, unsigned num_docs,
                                unsigned num_clusters)
{
    if (!assignfile || !fuzzyfile || !resultfile || !assignments || !probabilities || !term_indices ||
        (format != 0 && format != 1))
        return SMK_BAD_PARAM;
    if (num_term_indices < (int64_t)num_clusters * maxterms) {
        set_error("FlatClustWriteResults: term count is invalid");
        return SMK_BAD_PARAM;
    }
    std::map<int, int> doc_counts;
    for (unsigned i = 0; i < num_assignments; ++i) doc_counts[(int)assignments[i]] += 1;
    for (unsigned i = 0; i < num_clusters; ++i)
        if (doc_counts.count((int)i))
            for (unsigned q = 0; q < maxterms; ++q) {
                const int idx = term_indices[(size_t)i * maxterms + q];
                if (!dictionary || idx < 0 || idx >= dictionary_size) {
                    set_error("FlatClustWriteResults: dictionary too small");
                    return SMK_BAD_PARAM;
                }
            }
    if (doc_counts.size() != num_clusters)
        printf("Warning: only %zu clusters received an assignment.\n\n", doc_counts.size());
    if (!smk_write_assignments_file(assignments, num_assignments, assignfile))
        fprintf(stderr, "\terror writing flat assignments file\n");
    if (!smk_write_fuzzy_assignments_file(probabilities, num_clusters, num_docs, fuzzyfile))
        fprintf(stderr, "\terror writing fuzzy assignments file\n");
    std::ofstream f(resultfile);
    if (!f) {
        fprintf(stderr, "FlatClustWriteResults: could not open output file %s\n", resultfile);
        return SMK_FAILURE;
    }
    const std::string S4("    "), S8 = S4 + S4, S12 = S8 + S4, S16 = S12 + S4;
    const bool json = (format == 1);
    if (json) f << "{\n" << S4 << "\"doc_count\": " << num_docs << ",\n" << S4 << "\"nodes\": [\n";
    else f << "<?xml version=\"1.0\"?>\n<DataSet id=\"" << num_docs << "\">\n";
    for (unsigned i = 0; i < num_clusters; ++i) {
        const auto it = doc_counts.find((int)i);
        const int count = (it == doc_counts.end()) ? 0 : it->second;
        if (json) {
            if (i) f << ",\n";
            f << S8 << "{\n" << S12 << "\"id\": " << i << ",\n" << S12 << "\"doc_count\": " << count << ",\n";
            if (it != doc_counts.end() && maxterms > 0) {
                f << S12 << "\"top_terms\": [\n";
                for (unsigned q = 0; q < maxterms; ++q)
                    f << S16 << "\"" << dictionary[term_indices[(size_t)i * maxterms + q]] << "\""
                      << (q + 1 < maxterms ? ",\n" : "\n");
                f << S12 << "]\n";
            }
            f << S8 << "}";
        } else {
            f << S4 << "<node id=\"" << i << "\">\n" << S8 << "<doc_count>" << count << "</doc_count>\n";
            if (it != doc_counts.end()) {
                f << S8 << "<top_terms>\n";
                for (unsigned q = 0; q < maxterms; ++q)
                    f << S12 << "<term name=\"" << dictionary[term_indices[(size_t)i * maxterms + q]] << "\"/>\n";
                f << S8 << "</top_terms>\n";
            }
            f << S4 << "</node>\n";
        }
    }
    if (json) f << "\n" << S4 << "]\n}\n";
    else f << "</DataSet>\n";
    return f.good() ? SMK_OK : SMK_FAILURE;
}

}  // extern "C"
