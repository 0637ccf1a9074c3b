// smallk_amd/csrc/hierclust.cpp -- HierNMF2 (rank-2 hierarchical clustering) on top of the
// resident-matrix / solver C ABI.
//
// What the reference does (hierclust/include/clust_hier_generic.hpp:67-196): factor A with k = 2,
// split the documents by the larger H row, then repeatedly split the leaf with the highest
// priority score; each candidate split is a rank-2 NMF of the node's column subset, with an
// outlier-dropping retry loop (TrialSplit :203-327) and a modified-NDCG priority
// (compute_priority, clust_hier_util.hpp:105-173).
//
// Here A is uploaded once and stays in HBM; a node's submatrix is assembled on the device
// (smk_matrix_gather_cols: HBM -> HBM column gather for dense A, CSC cut by prefix sums for sparse A,
// sparse_subset.hip) and every factorisation is the RANK2 schedule of solver.cpp.  The tree search
// itself stays on the host like the reference's; for large vocabularies its argsorts run on the GPU
// (sort.hip).
#include "common.h"
#include "../../include/smallk_amd.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <fstream>
#include <functional>
#include <limits>
#include <numeric>
#include <sstream>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

using smk::i64;
using smk::set_error;

namespace {

const unsigned NONE = SMK_TREE_NONE;
const i64 DEVICE_SORT_MIN = 131072;      // below this the PCIe round trip outweighs the host sort

struct Node {
    double priority = 0.0;
    unsigned parent = NONE, left = NONE, right = NONE;
    bool valid = false, is_left = false;
    std::vector<double> topic;
    std::vector<int> terms;
    std::vector<unsigned> docs;
};

}  // namespace

struct smk_tree {
    std::vector<Node> nodes;
    std::vector<char> is_leaf;
    unsigned active = 0, index0 = 0, index1 = 0;
    i64 term_count = 0, doc_count = 0, leaf_doc_count = 0;
    int maxterms = 0;
    std::vector<unsigned> assignments, outliers;
    int flat_k = 0;                         // ClustFlat result (opts.flat): W m x k, H k x n
    std::vector<double> flatW, flatH;
};

namespace {

// ---- Tree<T>, hierclust/include/tree.hpp ----------------------------------------------------------
void tree_init(smk_tree& t, unsigned node_count, i64 m, i64 n)      // Init :120-144
{
    t.doc_count = n;
    t.term_count = m;
    t.nodes.assign(node_count, Node());         // topic vectors are created when a node is opened (tree_partition): a node the search
                                                // never reaches keeps an empty one, read as zeros (smk_tree_node_topic)
    t.is_leaf.assign(node_count, 0);
    t.active = 0;
}

void tree_min_max(const smk_tree& t, double& mn, double& mx, unsigned& mx_index)   // :147-170
{
    mn = std::numeric_limits<double>::max();
    mx = std::numeric_limits<double>::lowest();
    for (unsigned q = 0; q < t.is_leaf.size(); ++q) {
        if (!t.is_leaf[q]) continue;
        const double p = t.nodes[q].priority;
        if (p > 0.0 && p < mn) mn = p;
        if (p > mx) { mx = p; mx_index = q; }
    }
}

void tree_open_children(smk_tree& t, unsigned parent)
{
    const unsigned idx[2] = {t.index0, t.index1};
    for (int s = 0; s < 2; ++s) {
        Node& nd = t.nodes[idx[s]];
        nd.parent = parent; nd.left = NONE; nd.right = NONE; nd.valid = true; nd.is_left = (s == 0);
        t.is_leaf[idx[s]] = 1;
    }
}

// documents go left when H(0,c) > H(1,c); topic vectors are the two columns of W  (:173-211, :214-266)
void tree_partition(smk_tree& t, const std::vector<unsigned>* source, const double* W, i64 m, const double* H, i64 w)
{
    for (i64 c = 0; c < w; ++c) {
        const unsigned doc = source ? (*source)[(size_t)c] : (unsigned)c;
        t.nodes[H[2 * c] > H[2 * c + 1] ? t.index0 : t.index1].docs.push_back(doc);
    }
    t.nodes[t.index0].topic.assign(W, W + m);
    t.nodes[t.index1].topic.assign(W + m, W + 2 * m);
}

void tree_split_root(smk_tree& t, const double* W, i64 m, const double* H, i64 w)
{
    t.index0 = 0; t.index1 = 1;
    tree_open_children(t, NONE);
    t.active += 2;
    tree_partition(t, nullptr, W, m, H, w);
}

void tree_split(smk_tree& t, unsigned node, const double* W, i64 m, const double* H, i64 w)
{
    t.index0 = t.active; t.index1 = t.active + 1;
    t.active += 2;
    t.nodes[node].left = t.index0; t.nodes[node].right = t.index1;
    t.is_leaf[node] = 0;
    tree_open_children(t, node);
    const std::vector<unsigned> src = t.nodes[node].docs;
    tree_partition(t, &src, W, m, H, w);
}

void tree_top_terms(smk_tree& t, int maxterms)     // ComputeTopTerms :282-299 + TopTerms, terms.hpp:25-58
{
    t.maxterms = maxterms;
    // only the first `cnt` positions of the (value desc, index asc) order are needed: ONE pass over the topic vector with the
    // current top `cnt` in a small sorted buffer (a term enters only if it beats the buffer's last entry; ties keep the lower
    // index, which the scan order gives for free).  The full-length index array + partial_sort of rounds 1-3 cost ~8 ms per node
    // at a million terms; the nodes are independent and go to a few host threads.
    const i64 mterms = t.term_count;
    auto one = [&](Node& nd) {
        const double* d = nd.topic.data();
        nd.terms.assign((size_t)maxterms, 0);
        const size_t cnt = std::min<size_t>((size_t)maxterms, (size_t)mterms);
        std::vector<int> top;
        top.reserve(cnt + 1);
        for (i64 i = 0; i < mterms; ++i) {
            if (top.size() == cnt && !(d[i] > d[top.back()])) continue;
            size_t pos = top.size();
            while (pos > 0 && d[i] > d[top[pos - 1]]) --pos;          // equal values stay in front (lower index first)
            top.insert(top.begin() + (std::ptrdiff_t)pos, (int)i);
            if (top.size() > cnt) top.pop_back();
        }
        std::copy(top.begin(), top.end(), nd.terms.begin());
    };
    std::vector<Node*> todo;
    for (Node& nd : t.nodes)
        if (nd.valid && (i64)nd.topic.size() == mterms) todo.push_back(&nd);
    if (mterms < 65536 || todo.size() < 2) { for (Node* nd : todo) one(*nd); return; }
    const unsigned nth = (unsigned)std::min<size_t>(4, todo.size());
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nth; ++w)
        th.emplace_back([&, w] { for (size_t q = w; q < todo.size(); q += nth) one(*todo[q]); });
    for (std::thread& x : th) x.join();
}

void tree_assignments(smk_tree& t)                 // ComputeAssignments :302-338
{
    t.outliers.clear();
    t.assignments.assign((size_t)t.doc_count, NONE);
    t.leaf_doc_count = 0;
    for (unsigned q = 0; q < t.nodes.size(); ++q) {
        if (!t.is_leaf[q]) continue;
        t.leaf_doc_count += (i64)t.nodes[q].docs.size();
        for (unsigned d : t.nodes[q].docs) t.assignments[d] = q;
    }
    for (unsigned q = 0; q < t.assignments.size(); ++q)
        if (t.assignments[q] == NONE) t.outliers.push_back(q);
}

// ---- priority score, clust_hier_util.hpp:25-173 -----------------------------------------------------
// desc_ordered(): indices by decreasing value, ties by increasing index.  Sorting (value, index)
// pairs gives the same permutation as the reference's indirect comparator without the random
// access per comparison.
std::vector<int> desc_ordered(const double* v, size_t n)
{
    std::vector<std::pair<double, int>> p(n);
    for (size_t i = 0; i < n; ++i) p[i] = std::make_pair(v[i], (int)i);
    std::sort(p.begin(), p.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
        return a.first > b.first || (a.first == b.first && a.second < b.second);
    });
    std::vector<int> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = p[i].second;
    return idx;
}

// ordered() of a permutation (no ties) is its inverse
std::vector<int> inverse_permutation(const std::vector<int>& perm)
{
    std::vector<int> inv(perm.size());
    for (size_t i = 0; i < perm.size(); ++i) inv[(size_t)perm[i]] = (int)i;
    return inv;
}

// numerator of NDCG_part (clust_hier_util.hpp:50-99); the denominator (ideal score) is the same for
// both children and computed once by the caller
double ndcg_cum(const std::vector<int>& seq, const std::vector<int>& test, const std::vector<double>& weight_part)
{
    double cum = 0.0;
    for (size_t i = 0; i < test.size(); ++i) {
        double s = weight_part[(size_t)seq[(size_t)test[i]]];
        if (i > 0) s /= std::log2((double)(i + 1));
        cum += s;
    }
    return cum;
}

double priority_score(const double* wp, const double* wc, i64 n)
{
    static const bool timing = [] { const char* e = getenv("SMK_CLUST_TIMING"); return e && atoi(e) > 1; }();
    const auto T0 = std::chrono::high_resolution_clock::now();
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "[priority] %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - T0).count());
    };
    i64 n_part = 0;
    for (i64 i = 0; i < n; ++i) n_part += (wp[i] != 0.0);
    if (n_part <= 1) return -3.0;
    // Large vocabularies: the three argsorts (and the weight sort below) run as stable radix sorts on
    // the otherwise idle GPU (sort.hip); same permutation as the host comparator.  Without an
    // initialised device (host-only callers of smk_clust_priority) or for small n: host sorts, the
    // three independent ones side by side.
    const bool on_device = n >= DEVICE_SORT_MIN && smk_is_initialized() == SMK_INITIALIZED;
    // the whole score on the device (sort.hip): three doubles come back.  SMK_PRIORITY_HOST=1 keeps the host arithmetic
    // around the device sorts (bit-identical to the reference's sequential sums; the device sums differ by rounding).
    static const bool host_arith = [] { const char* e = getenv("SMK_PRIORITY_HOST"); return e && atoi(e) != 0; }();
    if (on_device && !host_arith) {
        double sc = 0.0;
        if (smk::device_priority_score(wp, wc, n, n_part, &sc, nullptr) == 0) { lap("device score done"); return sc; }
    }
    std::vector<int> idx_parent, idx_c1, idx_c2;
    bool sorted = false;
    if (on_device) {
        idx_parent.resize((size_t)n); idx_c1.resize((size_t)n); idx_c2.resize((size_t)n);
        const double* keys[3] = {wp, wc, wc + n};
        int* idx[3] = {idx_parent.data(), idx_c1.data(), idx_c2.data()};
        double* none[3] = {nullptr, nullptr, nullptr};
        sorted = smk::device_sort_desc(keys, idx, none, 3, n, nullptr) == 0;
    }
    if (!sorted && n >= 65536) {
        std::thread t1([&] { idx_c1 = desc_ordered(wc, (size_t)n); });
        std::thread t2([&] { idx_c2 = desc_ordered(wc + n, (size_t)n); });
        idx_parent = desc_ordered(wp, (size_t)n);
        t1.join();
        t2.join();
    } else if (!sorted) {
        idx_parent = desc_ordered(wp, (size_t)n);
        idx_c1 = desc_ordered(wc, (size_t)n);
        idx_c2 = desc_ordered(wc + n, (size_t)n);
    }
    lap("argsorts done");
    std::vector<double> weight((size_t)n), weight_part((size_t)n, 0.0);
    for (i64 j = 0; j < n; ++j) weight[(size_t)j] = std::log((double)(n - j));
    for (i64 i = 0; i < n; ++i)
        if (wp[idx_parent[(size_t)i]] == 0.0) {
            for (i64 j = i; j < n; ++j) weight[(size_t)j] = 1.0;
            break;
        }
    for (i64 j = 0; j < n_part; ++j) weight_part[(size_t)j] = std::log((double)(n_part - j));
    const std::vector<int> pos1 = inverse_permutation(idx_c1), pos2 = inverse_permutation(idx_c2);
    for (i64 i = 0; i < n; ++i) {
        const int t = idx_parent[(size_t)i];
        const int max_pos = std::max(pos1[(size_t)t], pos2[(size_t)t]);
        double discount = std::log((double)(n - max_pos));
        if (discount == 0.0) discount = std::log(2.0);
        weight[(size_t)i] /= discount;
        weight_part[(size_t)i] /= discount;
    }
    lap("weights done");
    const std::vector<int> seq = inverse_permutation(idx_parent);
    const double c1 = ndcg_cum(seq, idx_c1, weight_part), c2 = ndcg_cum(seq, idx_c2, weight_part);
    lap("ndcg done");
    bool weight_sorted = false;
    if (on_device) {
        const double* keys[1] = {weight.data()};
        int* idx[1] = {nullptr};
        double* out[1] = {weight.data()};
        weight_sorted = smk::device_sort_desc(keys, idx, out, 1, n, nullptr) == 0;
    }
    if (!weight_sorted) std::sort(weight.begin(), weight.end(), std::greater<double>());
    lap("weight sort done");
    double ideal_cum = 0.0;
    for (i64 i = 0; i < n; ++i) ideal_cum += (i > 0) ? weight[(size_t)i] / std::log2((double)(i + 1)) : weight[(size_t)i];
    lap("ideal done");
    return (c1 / ideal_cum) * (c2 / ideal_cum);
}

std::vector<unsigned> set_diff(const std::vector<unsigned>& a, const std::vector<unsigned>& b)   // setdiff.hpp:23-46
{
    std::vector<unsigned> out;
    size_t i = 0;
    for (unsigned x : b) {
        while (a[i] < x) out.push_back(a[i++]);
        ++i;
    }
    out.insert(out.end(), a.begin() + (std::ptrdiff_t)i, a.end());
    return out;
}

// ---- the search -------------------------------------------------------------------------------------
struct Stopwatch {       // SMK_CLUST_TIMING=1: where the wall time of a run goes
    double* acc;
    std::chrono::high_resolution_clock::time_point t0;
    explicit Stopwatch(double* a) : acc(a), t0(std::chrono::high_resolution_clock::now()) {}
    ~Stopwatch() { *acc += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count(); }
};

struct Run {
    double t_subset = 0, t_factor = 0, t_priority = 0, t_init = 0;
    double t_free = 0, t_tree = 0, t_scatter = 0, t_terms = 0, t_alloc = 0;      // inside "host bookkeeping": device frees, tree edits, scatter of W, top terms
    long iterations = 0, lanes_tried = 0, lanes_accepted = 0;        // speculative steps on devices 3 .. 8 (clust_hier)
    double t_round_max = 0, t_tasks = 0;    // sum over rounds of the longest trial split / sum of the trial splits kept: the multi-device projection
    const smk_clust_options* o = nullptr;
    smk_matrix* full = nullptr;
    i64 m = 0, n = 0;
    uint64_t seed = 0, draws = 0;
    std::string initdir;
    int init_counter = 1;
    smk_clust_stats stats = {0, 0};
    std::vector<unsigned> new_to_old;
    std::vector<double> Winit, Hinit;       // full-size initialisers from files
    std::vector<double> Ws, Hs;             // factors of the node being split (actual_split), reused
};

int load_init_file(const std::string& path, std::vector<double>& buf, unsigned h, unsigned w)
{
    buf.resize((size_t)h * w);
    unsigned fh = 0, fw = 0;
    if (!smk_load_csv(path.c_str(), buf.data(), (unsigned long)buf.size(), &fh, &fw) || fh != h || fw != w) {
        set_error("Load failed for file " + path);
        return SMK_FAILURE;
    }
    return SMK_OK;
}

// one node factorisation with up to three sets of initialisers (clust_hier_generic.hpp:97-121, :421-453)
int factor_node(Run& r, const smk_matrix* a, i64 h, i64 w, const unsigned* rows, const unsigned* cols,
                std::vector<double>& W, std::vector<double>& H, const char* what)
{
    // results only (smk_solver_get_factors overwrites every entry): no zero fill -- a fresh 14 MB vector per node cost ~3 ms of
    // page faults and memset at a million terms; callers that keep the buffers (actual_split) reuse them from node to node
    if (W.size() < (size_t)h * 2) W.resize((size_t)h * 2);
    if (H.size() < (size_t)w * 2) H.resize((size_t)w * 2);
    smk_options so = r.o->nmf;
    so.height = (int)h; so.width = (int)w; so.k = 2;
    so.algorithm = SMK_ALG_RANK2;
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (!r.initdir.empty()) {
            std::ostringstream fw, fh;
            fw << r.initdir << "Winit_" << r.init_counter << ".csv";
            fh << r.initdir << "Hinit_" << r.init_counter << ".csv";
            int rc = load_init_file(fw.str(), r.Winit, (unsigned)r.m, 2);
            if (rc == SMK_OK) rc = load_init_file(fh.str(), r.Hinit, 2, (unsigned)r.n);
            if (rc != SMK_OK) return rc;
            ++r.init_counter;
            for (i64 i = 0; i < h; ++i) {
                const i64 src = rows ? rows[i] : i;
                W[(size_t)i] = r.Winit[(size_t)src];
                W[(size_t)(h + i)] = r.Winit[(size_t)(r.m + src)];
            }
            for (i64 c = 0; c < w; ++c) {
                const i64 src = cols ? cols[c] : c;
                H[(size_t)(2 * c)] = r.Hinit[(size_t)(2 * src)];
                H[(size_t)(2 * c + 1)] = r.Hinit[(size_t)(2 * src + 1)];
            }
        }
        // random initialisers: the two counter-based uniform matrices of this attempt (W then H, as the reference draws them) are
        // generated on the device by the solver itself -- the host copies were 24 ms of fills + uploads per C5-shaped run
        const bool seeded = r.initdir.empty();
        uint64_t seed_w = 0, seed_h = 0;
        if (seeded) { seed_w = r.seed + 0x9E37u * (++r.draws); seed_h = r.seed + 0x9E37u * (++r.draws); }
        Stopwatch sw(&r.t_factor);
        smk_solver* s = nullptr;
        smk_stats st = {0, 0};
        int rc = smk_solver_create(&s, &so, a);
        if (rc == SMK_OK) rc = seeded ? smk_solver_set_factors_uniform(s, seed_w, seed_h) : smk_solver_set_factors(s, W.data(), h, H.data(), 2);
        if (rc == SMK_OK) {
            rc = smk_solver_run(s, &st);
            if (rc == SMK_OK) rc = smk_solver_get_factors(s, 0, W.data(), h, H.data(), 2);
        }
        smk_solver_destroy(s);
        r.iterations += st.iteration_count;
        {
            static const bool timing = [] { const char* e = getenv("SMK_CLUST_TIMING"); return e && atoi(e) != 0; }();
            if (timing)
                fprintf(stderr, "[smk_clust] %s %ld x %ld nnz %ld: %d iterations in %.1f ms (%.1f us each)\n", what, (long)h, (long)w,
                        (long)smk_matrix_nnz(a), st.iteration_count, st.elapsed_us * 1e-3,
                        st.iteration_count ? (double)st.elapsed_us / st.iteration_count : 0.0);
        }
        if (rc == SMK_OK) {
            r.stats.nmf_count += 1;
            if (st.iteration_count == so.max_iter) r.stats.max_count += 1;
            return SMK_OK;
        }
        if (rc != SMK_FAILURE) return rc;
        // a NaN projected gradient is an exception in the reference (projected_gradient.hpp:94-121 throws through
        // NmfSolve), not a solver `false`: no retry
        if (strstr(smk_last_error(), "ProjectedGradientNorm: NaN")) return SMK_FAILURE;
        printf("\n%s factorization failed, retrying with new initializers...\n", what);
    }
    set_error(std::string("HierNMF2: ") + (rows || cols ? "node" : "root node") +
              " factorization failed after three attempts");
    return SMK_FAILURE;
}

// ActualSplit, clust_hier_generic.hpp:383-499.  W is m x 2 (rows of the compacted problem scattered
// back), H is 2 x |subset|.
int actual_split(Run& r, const std::vector<unsigned>& subset, const double* w_parent, std::vector<double>& W,
                 std::vector<double>& H, std::vector<unsigned>& labels, double* priority)
{
    const i64 m = r.m;
    { Stopwatch sw(&r.t_alloc); W.assign((size_t)m * 2, 0.0); H.assign(subset.size() * 2, 0.0); }
    if (subset.size() <= 3) {
        labels.assign(subset.size(), 1u);
        *priority = -1.0;
        return SMK_OK;
    }
    smk_matrix* sub = nullptr;
    int64_t nh = 0;
    r.new_to_old.resize((size_t)m);
    int rc;
    {
        Stopwatch sw(&r.t_subset);
        rc = smk_matrix_gather_cols(r.full, subset.data(), (int64_t)subset.size(), &sub, r.new_to_old.data(), &nh);
    }
    if (rc != SMK_OK) return rc;
    std::vector<double>&Ws = r.Ws, &Hs = r.Hs;              // reused from node to node (factor_node fills the first nh x 2 / 2 x |subset| entries)
    rc = factor_node(r, sub, nh, (i64)subset.size(), r.new_to_old.data(), subset.data(), Ws, Hs, "Node");
    { Stopwatch sw(&r.t_free); smk_matrix_destroy(sub); }
    if (rc != SMK_OK) return rc;
    bool has0 = false, has1 = false;
    {
    Stopwatch sw_sc(&r.t_scatter);
    labels.clear();
    for (size_t c = 0; c < subset.size(); ++c) {
        if (Hs[2 * c] > Hs[2 * c + 1]) { labels.push_back(0u); has0 = true; }
        else { labels.push_back(1u); has1 = true; }
    }
    for (i64 i = 0; i < nh; ++i) {
        W[(size_t)r.new_to_old[(size_t)i]] = Ws[(size_t)i];
        W[(size_t)(m + r.new_to_old[(size_t)i])] = Ws[(size_t)(nh + i)];
    }
    H.assign(Hs.begin(), Hs.begin() + (std::ptrdiff_t)(2 * subset.size()));       // Hs is a reused buffer: only this node's part
    }
    Stopwatch sw(&r.t_priority);
    *priority = (has0 && has1) ? priority_score(w_parent, W.data(), m) : -1.0;
    return SMK_OK;
}

// TrialSplit, clust_hier_generic.hpp:203-327.  `subset` is the node's document list and is edited
// in place when outliers are dropped.
int trial_split(Run& r, std::vector<unsigned>& subset, double min_priority, const double* w_parent,
                std::vector<double>& W, std::vector<double>& H, double* priority_out, bool* used_min_priority = nullptr)
{
    if (used_min_priority) *used_min_priority = false;
    const smk_clust_options& o = *r.o;
    const std::vector<unsigned> backup(subset);
    std::vector<unsigned> small, labels, labels_small;
    std::vector<double> Wtmp, Htmp;
    int trial = 0;
    double pr = -2.0;
    while (trial < o.trial_allowance) {
        int rc = actual_split(r, subset, w_parent, W, H, labels, &pr);
        if (rc != SMK_OK) return rc;
        if (pr < 0.0) break;
        int counts[2] = {0, 0};
        for (unsigned l : labels) counts[l] += 1;
        const int smallest = std::min(counts[0], counts[1]);
        if (!((double)smallest < o.unbalanced * (double)labels.size())) break;
        const unsigned lab = (smallest == counts[0]) ? 0u : 1u;
        small.clear();
        for (size_t q = 0; q < labels.size(); ++q)
            if (labels[q] == lab) small.push_back(subset[q]);
        double pr_small = 0.0;
        rc = actual_split(r, small, W.data() + (size_t)lab * r.m, Wtmp, Htmp, labels_small, &pr_small);
        if (rc != SMK_OK) return rc;
        if (used_min_priority) *used_min_priority = true;       // the only place the outcome depends on the other leaves' priorities
        if (!(pr_small < min_priority)) break;
        trial += 1;
        if (trial < o.trial_allowance) {
            printf("dropping %zu items ...\n", small.size());
            subset = set_diff(subset, small);
        }
    }
    if (trial == o.trial_allowance) {
        if (o.verbose) printf("recycling %zu items ...\n", small.size());
        subset = backup;
        W.assign((size_t)r.m * 2, 0.0);
        H.assign(subset.size() * 2, 0.0);
        pr = -2.0;
    }
    *priority_out = pr;
    return SMK_OK;
}

// ---- two devices (SMK_CLUST_DEVICES=2): the two TrialSplits of a step are independent (clust_hier_generic.hpp:383-517
// runs them one after the other) -- the second one runs on a worker thread that owns a second device context and a copy
// of A there.  The sequential run hands the second child the initialiser draws (or init-file counter) that follow the
// first child's; the worker assumes the first child takes its nominal share (one factorisation, or none for a node of
// <= 3 documents) and the step is repeated for the second child on the main device if that turns out wrong (a retry or
// an outlier trial consumed more) -- so the tree is the one-device tree, draw for draw.
// MEASUREMENT HOOK (SMK_CLUST_SERIALIZE=1, contexts sharing one GPU): the trial splits of a round run one after the other, so
// each one's wall time is what it would be on a device of its own and "sum over rounds of the longest" projects the run on
// SMK_CLUST_DEVICES real devices.  The results are the same either way.
std::mutex g_serialize_mu;
inline bool serialize_trials()
{
    static const bool on = [] { const char* e = getenv("SMK_CLUST_SERIALIZE"); return e && atoi(e) != 0; }();
    return on;
}

struct SplitTask {
    std::vector<unsigned>* docs = nullptr;
    double min_priority = 0.0;
    const double* w_parent = nullptr;
    std::vector<double>*W = nullptr, *H = nullptr;
    double priority = 0.0;
    bool used_min_priority = false;
    double seconds = 0.0;                 // wall time of this trial split (projection of the multi-device critical path)
    uint64_t draws0 = 0, draws1 = 0;
    int counter0 = 0, counter1 = 0;
    int rc = SMK_OK;
    std::string err;
    smk_clust_stats stats = {0, 0};
    long iterations = 0;
};

struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    SplitTask* task = nullptr;
    bool quit = false, done = false, ready = false;
    int device = 0, setup_rc = SMK_OK;
    const smk_matrix* src = nullptr;
    Run run;                              // the worker's own scratch, timers and copy of A

    void body()
    {
        setup_rc = smk_thread_context_begin(device);
        const bool have_ctx = setup_rc == SMK_OK;
        smk_matrix* copy = nullptr;
        if (setup_rc == SMK_OK) setup_rc = smk_matrix_clone(src, &copy);
        run.full = copy;
        { std::lock_guard<std::mutex> lk(mu); ready = true; }
        cv.notify_all();
        for (;;) {
            SplitTask* t = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || task != nullptr; });
                if (quit) break;
                t = task;
            }
            run.draws = t->draws0;
            run.init_counter = t->counter0;
            run.stats = smk_clust_stats{0, 0};              // per task: only an ACCEPTED speculation counts in the run's totals
            run.iterations = 0;
            {
                std::unique_lock<std::mutex> one;
                if (serialize_trials()) one = std::unique_lock<std::mutex>(g_serialize_mu);
                Stopwatch sw(&t->seconds);
                t->rc = (setup_rc == SMK_OK) ? trial_split(run, *t->docs, t->min_priority, t->w_parent, *t->W, *t->H, &t->priority, &t->used_min_priority)
                                             : setup_rc;
            }
            if (t->rc != SMK_OK) t->err = smk_last_error();
            t->draws1 = run.draws;
            t->counter1 = run.init_counter;
            t->stats = run.stats;
            t->iterations = run.iterations;
            { std::lock_guard<std::mutex> lk(mu); task = nullptr; done = true; }
            cv.notify_all();
        }
        smk::device_priority_release();
        smk_matrix_destroy(copy);
        if (have_ctx) smk_thread_context_end();
    }
    void submit(SplitTask* t)
    {
        { std::lock_guard<std::mutex> lk(mu); done = false; task = t; }
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

// the workers of this run: SMK_CLUST_DEVICES - 1 of them (none: one device, or they could not be set up), worker j on device
// cur + 1 + j (all on the current device with SMK_SHARDS_ON_ONE_GPU=1: tests)
std::vector<Worker*> start_workers(const Run& r)
{
    std::vector<Worker*> out;
    const char* e = getenv("SMK_CLUST_DEVICES");
    if (!e || atoi(e) < 2) return out;
    const int want = std::min(atoi(e), 8);
    const int cur = smk_current_device(), ndev = smk_device_count();
    if (cur < 0 || ndev < 1) return out;
    const char* one = getenv("SMK_SHARDS_ON_ONE_GPU");
    const bool same = one && atoi(one) != 0;
    const int devices = same ? want : std::min(want, ndev);
    for (int j = 0; j + 1 < devices; ++j) {
        Worker* w = new Worker;
        w->device = same ? cur : (cur + 1 + j) % ndev;
        w->src = r.full;
        w->run.o = r.o; w->run.m = r.m; w->run.n = r.n; w->run.seed = r.seed; w->run.initdir = r.initdir;
        w->th = std::thread([w] { w->body(); });
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->ready; });
        }
        if (w->setup_rc != SMK_OK) { w->stop(); delete w; break; }
        out.push_back(w);
    }
    return out;
}

// ClustHier, clust_hier_generic.hpp:67-196
int clust_hier(Run& r, smk_tree& tree)
{
    const smk_clust_options& o = *r.o;
    const i64 m = r.m, n = r.n;
    const unsigned num_clusters = (unsigned)o.num_clusters;
    const unsigned node_count = 2 * (num_clusters - 1);
    tree_init(tree, node_count, m, n);

    std::vector<double> W0, H0;
    int rc = factor_node(r, r.full, m, n, nullptr, nullptr, W0, H0, "Root node");
    if (rc != SMK_OK) return rc;

    struct WorkerGuard {
        std::vector<Worker*> ws;
        Run& r;
        ~WorkerGuard()
        {
            for (Worker* w : ws) {
                w->stop();
                r.t_subset += w->run.t_subset; r.t_factor += w->run.t_factor; r.t_priority += w->run.t_priority; r.t_init += w->run.t_init;
                delete w;
            }
        }
    } guard{start_workers(r), r};
    const std::vector<Worker*>& workers = guard.ws;
    Worker* worker = workers.empty() ? nullptr : workers[0];
    std::vector<std::vector<double>> Wbuf(node_count), Hbuf(node_count);
    double min_priority = 0.0, max_priority = 0.0;
    unsigned split_index = 0;

    // ---- more than two devices: speculative steps (round 4) ----------------------------------------------------------------
    // The reference's loop is sequential: split the leaf of highest priority, trial-split its two children, repeat.  Two devices
    // take the two children of a step.  Further devices work AHEAD: while the children of step i are being factored, lane d
    // (d = 1, 2, 3: devices 2d and 2d + 1) factors the children that step i + d will create IF it splits the d-th best of the
    // leaves that exist now -- which it does unless a child created in between comes out with a higher priority.  A lane is
    // accepted only if everything it assumed turns out true, in the reference's own order: the leaf it split is the arg-max when
    // its step comes; the initialiser draws it used are the ones the sequential run would have handed it (every earlier trial
    // took its nominal share); and the smallest positive leaf priority it was given -- which matters only inside an outlier
    // trial -- is the one the sequential run would have passed, or was never consulted.  A lane that fails any test is dropped
    // with all later lanes and the loop goes on from there, so the tree is the one-device tree draw for draw and bit for bit.
    struct Lane {
        unsigned node = 0;
        std::vector<unsigned> docs[2];
        std::vector<double> W[2], H[2];
        SplitTask task[2];
    };
    const size_t max_lanes = workers.size() >= 3 ? (workers.size() - 1) / 2 : 0;
    auto nominal = [&](size_t ndocs, uint64_t& draws, int& counter) {      // what one trial split takes when nothing is retried
        if (ndocs > 3) { if (r.initdir.empty()) draws += 2; else counter += 1; }
    };

    unsigned i = 0;
    while (i + 1 < num_clusters) {
        if (i == 0) {
            min_priority = INFINITY;
            tree_split_root(tree, W0.data(), m, H0.data(), n);
        } else {
            tree_min_max(tree, min_priority, max_priority, split_index);
            if (max_priority < 0.0) {
                printf("\nHierNMF2: no further factorization possible.\n\n");
                break;
            }
            Stopwatch sw(&r.t_tree);
            tree_split(tree, split_index, Wbuf[split_index].data(), m, Hbuf[split_index].data(),
                       (i64)(Hbuf[split_index].size() / 2));
            std::vector<double>().swap(Wbuf[split_index]);
            std::vector<double>().swap(Hbuf[split_index]);
        }
        const unsigned idx[2] = {tree.index0, tree.index1};
        // a child's own topic vector is the "parent" vector of its trial split; trial_split writes Wbuf / Hbuf / docs, never a
        // topic vector, and tree.nodes is not resized during the search: pointers, not copies (2 x 8 MB per step at 1 M terms)
        const double *w_parent0 = tree.nodes[idx[0]].topic.data(), *w_parent1 = tree.nodes[idx[1]].topic.data();
        uint64_t cursor_draws = r.draws;
        int cursor_counter = r.init_counter;
        nominal(tree.nodes[idx[0]].docs.size(), cursor_draws, cursor_counter);
        SplitTask spec;
        std::vector<unsigned> docs1_backup;
        bool speculated = false;
        if (worker) {
            // second child on the other device, assuming the first one takes its nominal share of the initialisers
            docs1_backup = tree.nodes[idx[1]].docs;
            spec.docs = &tree.nodes[idx[1]].docs; spec.min_priority = min_priority; spec.w_parent = w_parent1;
            spec.W = &Wbuf[idx[1]]; spec.H = &Hbuf[idx[1]];
            spec.draws0 = cursor_draws;
            spec.counter0 = cursor_counter;
            worker->submit(&spec);
            speculated = true;
        }
        nominal(tree.nodes[idx[1]].docs.size(), cursor_draws, cursor_counter);

        // lanes: the steps after this one, on the leaves that exist now, best first
        std::vector<Lane> lanes;
        if (max_lanes > 0 && i > 0) {
            std::vector<unsigned> cand;
            for (unsigned q = 0; q < tree.is_leaf.size(); ++q)
                if (tree.is_leaf[q] && q != idx[0] && q != idx[1] && !(tree.nodes[q].priority < 0.0)) cand.push_back(q);
            std::stable_sort(cand.begin(), cand.end(), [&](unsigned a, unsigned b) { return tree.nodes[a].priority > tree.nodes[b].priority; });
            const size_t steps_left = (size_t)(num_clusters - 1) - (size_t)(i + 1);
            const size_t nl = std::min(std::min(max_lanes, cand.size()), steps_left);
            lanes.resize(nl);
            for (size_t d = 0; d < nl; ++d) {
                Lane& ln = lanes[d];
                ln.node = cand[d];
                const std::vector<unsigned>& src = tree.nodes[ln.node].docs;
                const std::vector<double>& Hn = Hbuf[ln.node];
                for (size_t c = 0; c < src.size(); ++c) ln.docs[Hn[2 * c] > Hn[2 * c + 1] ? 0 : 1].push_back(src[c]);      // tree_partition's rule
                // the smallest positive priority among the leaves the sequential run would see when this lane's step starts,
                // as far as it is known now: the old leaves that have not been split by then (this lane's own leaf included)
                double mn = std::numeric_limits<double>::max();
                for (size_t c2 = d; c2 < cand.size(); ++c2) { const double p = tree.nodes[cand[c2]].priority; if (p > 0.0 && p < mn) mn = p; }
                for (int sidx = 0; sidx < 2; ++sidx) {
                    SplitTask& t = ln.task[sidx];
                    t.docs = &ln.docs[sidx]; t.min_priority = mn; t.w_parent = Wbuf[ln.node].data() + (size_t)sidx * m;
                    t.W = &ln.W[sidx]; t.H = &ln.H[sidx];
                    t.draws0 = cursor_draws; t.counter0 = cursor_counter;
                    nominal(ln.docs[sidx].size(), cursor_draws, cursor_counter);
                    workers[1 + 2 * d + (size_t)sidx]->submit(&t);
                }
            }
        }

        double pr0 = 0.0, sec0 = 0.0;
        {
            std::unique_lock<std::mutex> one;
            if (serialize_trials()) one = std::unique_lock<std::mutex>(g_serialize_mu);
            Stopwatch sw(&sec0);
            rc = trial_split(r, tree.nodes[idx[0]].docs, min_priority, w_parent0, Wbuf[idx[0]], Hbuf[idx[0]], &pr0);
        }
        if (speculated) worker->wait();
        for (size_t d = 0; d < lanes.size(); ++d) { workers[1 + 2 * d]->wait(); workers[2 + 2 * d]->wait(); }
        if (rc != SMK_OK) return rc;
        {
            // what this round costs when every task has a device of its own (the trial splits are measured one by one, whether
            // they overlapped here or not), against the work of the tasks that end up in the tree
            double longest = std::max(sec0, speculated ? spec.seconds : 0.0);
            for (const Lane& ln : lanes) longest = std::max(longest, std::max(ln.task[0].seconds, ln.task[1].seconds));
            r.t_round_max += longest;
            r.t_tasks += sec0;
        }
        tree.nodes[idx[0]].priority = pr0;
        if (speculated && spec.draws0 == r.draws && spec.counter0 == r.init_counter) {
            if (spec.rc != SMK_OK) { set_error(spec.err); return spec.rc; }
            tree.nodes[idx[1]].priority = spec.priority;
            r.draws = spec.draws1;
            r.init_counter = spec.counter1;
            r.stats.nmf_count += spec.stats.nmf_count; r.stats.max_count += spec.stats.max_count;
            r.iterations += spec.iterations;
            r.t_tasks += spec.seconds;
        } else {
            if (speculated) tree.nodes[idx[1]].docs = docs1_backup;      // the first child took more initialisers: this one again, in order
            double pr1 = 0.0, sec1 = 0.0;
            { Stopwatch sw(&sec1); rc = trial_split(r, tree.nodes[idx[1]].docs, min_priority, w_parent1, Wbuf[idx[1]], Hbuf[idx[1]], &pr1); }
            r.t_round_max += sec1;
            r.t_tasks += sec1;
            if (rc != SMK_OK) return rc;
            tree.nodes[idx[1]].priority = pr1;
        }
        if (o.verbose) { printf("[%u] ", i + 1); fflush(stdout); }
        i += 1;

        // ---- accept the lanes whose assumptions held, in order ----
        for (size_t d = 0; d < lanes.size() && i + 1 < num_clusters; ++d) {
            Lane& ln = lanes[d];
            double mn = 0.0, mx = 0.0;
            unsigned arg = 0;
            tree_min_max(tree, mn, mx, arg);
            if (mx < 0.0 || arg != ln.node) break;                                  // a newer child ranks higher (or the search is over)
            SplitTask &t0 = ln.task[0], &t1 = ln.task[1];
            if (t0.draws0 != r.draws || t0.counter0 != r.init_counter) break;     // an earlier trial took more initialisers than its share
            if (t1.draws0 != t0.draws1 || t1.counter0 != t0.counter1) break;
            if ((t0.used_min_priority && t0.min_priority != mn) || (t1.used_min_priority && t1.min_priority != mn)) break;
            if (t0.rc != SMK_OK) { set_error(t0.err); return t0.rc; }               // what the sequential run would have hit, in its order
            if (t1.rc != SMK_OK) { set_error(t1.err); return t1.rc; }
            {
                Stopwatch sw(&r.t_tree);
                tree_split(tree, ln.node, Wbuf[ln.node].data(), m, Hbuf[ln.node].data(), (i64)(Hbuf[ln.node].size() / 2));
                std::vector<double>().swap(Wbuf[ln.node]);
                std::vector<double>().swap(Hbuf[ln.node]);
            }
            const unsigned cidx[2] = {tree.index0, tree.index1};
            for (int sidx = 0; sidx < 2; ++sidx) {
                SplitTask& t = ln.task[sidx];
                tree.nodes[cidx[sidx]].docs.swap(ln.docs[sidx]);                    // as trial_split left them (outliers dropped)
                Wbuf[cidx[sidx]].swap(ln.W[sidx]);
                Hbuf[cidx[sidx]].swap(ln.H[sidx]);
                tree.nodes[cidx[sidx]].priority = t.priority;
                r.stats.nmf_count += t.stats.nmf_count; r.stats.max_count += t.stats.max_count;
                r.iterations += t.iterations;
                r.t_tasks += t.seconds;
            }
            r.draws = t1.draws1;
            r.init_counter = t1.counter1;
            r.lanes_accepted += 1;
            if (o.verbose) { printf("[%u] ", i + 1); fflush(stdout); }
            i += 1;
        }
        r.lanes_tried += (long)lanes.size();
    }
    {
        Stopwatch sw(&r.t_terms);
        tree_top_terms(tree, o.maxterms);
        tree_assignments(tree);
    }
    printf("\n");
    return SMK_OK;
}

// ClustFlat, hierclust/include/clust_flat_generic.hpp:33-74: W = the leaf topic vectors
// (Tree::FlatclustInitW, tree.hpp:341-385), H random, then NnlsHals with W fixed.
int clust_flat(Run& r, smk_tree& t)
{
    const smk_clust_options& o = *r.o;
    const i64 m = r.m, n = r.n;
    const int k = o.num_clusters;
    int leaves = 0;
    for (char l : t.is_leaf) leaves += (l != 0);
    if (leaves != k) {
        fprintf(stderr, "Insufficient number of leaf nodes for flat clustering.\n");
        set_error("Insufficient number of leaf nodes for flat clustering.");
        return SMK_FLATCLUST_FAILURE;
    }
    if (k > 1024) { set_error("flat clustering: more than 1024 clusters is not built on the device path"); return SMK_UNSUPPORTED; }
    std::vector<double> W((size_t)m * k), H((size_t)k * n);
    int c = 0;
    for (size_t q = 0; q < t.nodes.size(); ++q)
        if (t.is_leaf[q]) std::copy(t.nodes[q].topic.begin(), t.nodes[q].topic.end(), W.begin() + (size_t)(c++) * m);
    smk_options so = o.nmf;
    so.height = (int)m; so.width = (int)n; so.k = k;
    so.algorithm = SMK_ALG_HALS;
    so.prog_est_algorithm = SMK_PROG_PG_RATIO;
    int rc = SMK_FAILURE;
    for (int attempt = 0; attempt < 3; ++attempt) {
        smk_uniform_fill_host(H.data(), k, k, n, 0, 0, k, r.seed + 0x9E37u * (++r.draws), 0);
        smk_solver* s = nullptr;
        int its = 0;
        rc = smk_solver_create(&s, &so, r.full);
        if (rc == SMK_OK) rc = smk_solver_set_factors(s, W.data(), m, H.data(), k);
        if (rc == SMK_OK) rc = smk_solver_nnls_hals(s, o.nmf.tol, o.verbose, o.nmf.max_iter, &its);
        if (rc == SMK_OK) rc = smk_solver_get_factors(s, 0, W.data(), m, H.data(), k);
        smk_solver_destroy(s);
        if (rc != SMK_FAILURE) break;
    }
    if (rc == SMK_FAILURE) {
        printf("Flatclust NNLS solver failed after 3 attempts.\n");
        fprintf(stderr, "Flat clustering failed.\n");
        return SMK_FLATCLUST_FAILURE;
    }
    if (rc != SMK_OK) return rc;
    t.flat_k = k;
    t.flatW.swap(W);
    t.flatH.swap(H);
    return SMK_OK;
}

int check_sizes(const smk_clust_options* o)        // clust.cpp:116-131
{
    const uint64_t lim = (uint64_t)std::numeric_limits<int>::max();
    if (2ull * (uint64_t)o->nmf.height > lim) { set_error("W matrix size too large"); return SMK_SIZE_TOO_LARGE; }
    if (2ull * (uint64_t)o->nmf.width > lim) { set_error("H matrix size too large"); return SMK_SIZE_TOO_LARGE; }
    return SMK_OK;
}

int run_clust(const smk_clust_options* opts, smk_matrix* full, uint64_t seed, uint64_t* draws, const char* initdir,
              smk_tree** tree_out, smk_clust_stats* stats)
{
    // the flat step factors with k = number of clusters: refuse before the tree search, not after it
    if (opts->flat && opts->num_clusters > 1024) {
        set_error("flat clustering: more than 1024 clusters is not built on the device path");
        return SMK_UNSUPPORTED;
    }
    Run r;
    r.o = opts; r.full = full; r.m = opts->nmf.height; r.n = opts->nmf.width;
    r.seed = seed; r.draws = draws ? *draws : 0;
    if (initdir) r.initdir = initdir;
    smk_tree* t = new smk_tree;
    double t_search = 0.0, t_flat = 0.0;
    int rc;
    { Stopwatch sw(&t_search); rc = clust_hier(r, *t); }
    if (rc == SMK_OK && opts->flat) { Stopwatch sw(&t_flat); rc = clust_flat(r, *t); }
    if (draws) *draws = r.draws;
    if (stats) *stats = r.stats;
    smk::device_priority_release();        // workspace of the device-side priority score (kept between the calls of a run)
    if (const char* e = getenv("SMK_CLUST_TIMING"))
        if (atoi(e))
            fprintf(stderr, "[smk_clust] subset %.3fs  factor %.3fs (%ld RANK2 iterations)  priority %.3fs  init %.3fs  |  tree search %.3fs in all "
                    "(host bookkeeping %.3fs: device frees %.3f, tree edits %.3f, labels + scatter of W %.3f, top terms + assignments %.3f, zero-filled W buffers %.3f)  flat step %.3fs  speculative steps accepted %ld of %ld; trial splits kept %.3fs, sum over rounds of the longest one %.3fs (= their time on "
                    "SMK_CLUST_DEVICES real devices)\n",
                    r.t_subset, r.t_factor, r.iterations, r.t_priority, r.t_init, t_search,
                    t_search - r.t_subset - r.t_factor - r.t_priority - r.t_init, r.t_free, r.t_tree, r.t_scatter, r.t_terms, r.t_alloc, t_flat, r.lanes_accepted, r.lanes_tried, r.t_tasks, r.t_round_max);
    // a failed flat step still returns the tree (RunClust, clust.cpp:53-61: the caller writes it)
    if (rc != SMK_OK && rc != SMK_FLATCLUST_FAILURE) { delete t; return rc; }
    *tree_out = t;
    return rc;
}

int precheck(const smk_clust_options* opts, smk_tree** tree)
{
    if (!tree) return SMK_BAD_PARAM;
    *tree = nullptr;
    if (smk_is_initialized() != SMK_INITIALIZED) {
        set_error("clustlib error: smk_initialize() must be called prior to any clustering routine");
        return SMK_NOTINITIALIZED;
    }
    if (!opts || !smk_clust_is_valid(opts, 1)) return SMK_BAD_PARAM;
    return check_sizes(opts);
}

}  // namespace

extern "C" {

// IsValid(ClustOptions), hierclust/src/clust_options.cpp:16-110
int smk_clust_is_valid(const smk_clust_options* o, int validate_matrix)
{
    if (!o) return 0;
    const char* msg = nullptr;
    if (validate_matrix) {
        if (o->nmf.height <= 0) msg = "error: matrix height must be a positive integer";
        else if (o->nmf.width <= 0) msg = "error: matrix width must be a positive integer";
        else if (o->nmf.k <= 0) msg = "error: cluster count must be a positive integer";
        else if (o->nmf.k > o->nmf.width) msg = "error: k value cannot exceed the matrix width";
    }
    if (!msg) {
        if (o->num_clusters <= 1) msg = "error: number of clusters must be >= 2";
        else if (o->nmf.tol <= 0.0 || o->nmf.tol >= 1.0) msg = "error: tolerance must be in the interval (0.0, 1.0)";
        else if (o->nmf.min_iter <= 0) msg = "error: miniter must be a positive integer";
        else if (o->nmf.max_iter <= 0) msg = "error: maxiter must be a positive integer";
        else if (o->maxterms <= 0) msg = "error: maxterms must be a positive integer";
        else if (o->trial_allowance < 0) msg = "error: trial_allowance for hierarchical clustering is negative";
        else if (o->unbalanced < 0.0 || o->unbalanced >= 1.0) msg = "error: the unbalanced value should be in the interval [0, 1)";
        else if (o->nmf.prog_est_algorithm != SMK_PROG_PG_RATIO && o->nmf.prog_est_algorithm != SMK_PROG_DELTA_FNORM)
            msg = "error: unknown stopping criterion ";
    }
    if (msg) { fprintf(stderr, "%s\n", msg); set_error(msg); return 0; }
    return 1;
}

int smk_clust_dense(const smk_clust_options* opts, const double* A, int64_t ldA, int storage, uint64_t seed,
                    uint64_t* draws, const char* initdir, smk_tree** tree, smk_clust_stats* stats)
{
    int rc = precheck(opts, tree);
    if (rc != SMK_OK) return rc;
    if (!A || ldA < opts->nmf.height) { set_error("invalid leading dimension for input matrix"); return SMK_BAD_PARAM; }
    smk_matrix* a = nullptr;
    rc = smk_matrix_create(&a, opts->nmf.height, opts->nmf.width, 0, opts->nmf.width, storage);
    if (rc == SMK_OK) rc = smk_matrix_upload_f64(a, A, ldA);
    if (rc == SMK_OK) rc = run_clust(opts, a, seed, draws, initdir, tree, stats);
    smk_matrix_destroy(a);
    return rc;
}

int smk_clust_sparse(const smk_clust_options* opts, int64_t nnz, const unsigned* col_offsets,
                     const unsigned* row_indices, const double* data, uint64_t seed, uint64_t* draws,
                     const char* initdir, smk_tree** tree, smk_clust_stats* stats)
{
    int rc = precheck(opts, tree);
    if (rc != SMK_OK) return rc;
    smk_matrix* a = nullptr;
    double t_create = 0.0;
    {
        Stopwatch sw(&t_create);
        rc = smk_matrix_create_sparse(&a, opts->nmf.height, opts->nmf.width, 0, opts->nmf.width, nnz, col_offsets,
                                      row_indices, data);
    }
    if (const char* e = getenv("SMK_CLUST_TIMING"))
        if (atoi(e)) fprintf(stderr, "[smk_clust] matrix to the device (CSC upload + transpose): %.3fs\n", t_create);
    if (rc == SMK_OK) rc = run_clust(opts, a, seed, draws, initdir, tree, stats);
    smk_matrix_destroy(a);
    return rc;
}

// the same on a matrix that already lives in HBM (dense or sparse; not modified, not owned)
int smk_clust_resident(const smk_clust_options* opts, const smk_matrix* a, uint64_t seed, uint64_t* draws,
                       const char* initdir, smk_tree** tree, smk_clust_stats* stats)
{
    int rc = precheck(opts, tree);
    if (rc != SMK_OK) return rc;
    if (!a) return SMK_BAD_PARAM;
    return run_clust(opts, const_cast<smk_matrix*>(a), seed, draws, initdir, tree, stats);
}

void smk_tree_destroy(smk_tree* t) { delete t; }
int smk_tree_node_count(const smk_tree* t) { return t ? (int)t->nodes.size() : 0; }
int64_t smk_tree_term_count(const smk_tree* t) { return t ? t->term_count : 0; }
int64_t smk_tree_doc_count(const smk_tree* t) { return t ? t->doc_count : 0; }

int smk_tree_get_node(const smk_tree* t, int q, smk_tree_node* out)
{
    if (!t || !out || q < 0 || q >= (int)t->nodes.size()) return SMK_BAD_PARAM;
    const Node& nd = t->nodes[(size_t)q];
    out->priority = nd.priority;
    out->parent = nd.parent; out->left_child = nd.left; out->right_child = nd.right;
    out->is_valid = nd.valid; out->is_left_child = nd.is_left; out->is_leaf = t->is_leaf[(size_t)q];
    out->doc_count = (int64_t)nd.docs.size();
    return SMK_OK;
}

int smk_tree_node_docs(const smk_tree* t, int q, unsigned* out)
{
    if (!t || !out || q < 0 || q >= (int)t->nodes.size()) return SMK_BAD_PARAM;
    std::copy(t->nodes[(size_t)q].docs.begin(), t->nodes[(size_t)q].docs.end(), out);
    return SMK_OK;
}

int smk_tree_node_topic(const smk_tree* t, int q, double* out)
{
    if (!t || !out || q < 0 || q >= (int)t->nodes.size()) return SMK_BAD_PARAM;
    const std::vector<double>& tv = t->nodes[(size_t)q].topic;
    if (tv.empty()) std::fill(out, out + t->term_count, 0.0);             // a node the search never opened
    else std::copy(tv.begin(), tv.end(), out);
    return SMK_OK;
}

int smk_tree_node_terms(const smk_tree* t, int q, int* out)
{
    if (!t || !out || q < 0 || q >= (int)t->nodes.size()) return 0;
    std::copy(t->nodes[(size_t)q].terms.begin(), t->nodes[(size_t)q].terms.end(), out);
    return (int)t->nodes[(size_t)q].terms.size();
}

int64_t smk_tree_assignments(const smk_tree* t, unsigned* out)
{
    if (!t) return 0;
    if (out) std::copy(t->assignments.begin(), t->assignments.end(), out);
    return (int64_t)t->assignments.size();
}

int64_t smk_tree_outliers(const smk_tree* t, unsigned* out)
{
    if (!t) return 0;
    if (out) std::copy(t->outliers.begin(), t->outliers.end(), out);
    return (int64_t)t->outliers.size();
}

// Tree::WriteAssignments, tree.hpp:388-423: labels on one line ("-1" for outliers), a blank line,
// then the outlier document indices.
int smk_tree_write_assignments(const smk_tree* t, const char* path)
{
    if (!t || !path || t->assignments.empty()) return SMK_BAD_PARAM;
    std::ofstream f(path);
    if (!f) {
        fprintf(stderr, "Tree::WriteAssignments: could not open output file %s\n", path);
        return SMK_FAILURE;
    }
    f << t->assignments[0];
    for (size_t q = 1; q < t->assignments.size(); ++q) {
        f << ",";
        if (t->assignments[q] == NONE) f << -1; else f << t->assignments[q];
    }
    f << "\n\n";
    if (!t->outliers.empty()) {
        f << t->outliers[0];
        for (size_t q = 1; q < t->outliers.size(); ++q) f << ',' << t->outliers[q];
        f << "\n";
    }
    return f.good() ? SMK_OK : SMK_FAILURE;
}

// Tree::WriteTree (tree.hpp:426-465) through the XML / JSON node writers
// (hierclust/src/hierclust_xml_writer.cpp, hierclust_json_writer.cpp).  Ids print as signed ints,
// so an absent parent/child is -1.
int smk_tree_write(const smk_tree* t, const char* path, int format, const char* const* dict, int64_t dict_size)
{
    if (!t || !path || (format != 0 && format != 1)) return SMK_BAD_PARAM;
    for (const Node& nd : t->nodes)
        for (int idx : nd.terms)
            if (idx < 0 || idx >= dict_size || !dict) { set_error("Tree::Write: dictionary too small"); return SMK_BAD_PARAM; }
    std::ofstream f(path);
    if (!f) {
        fprintf(stderr, "Tree::Write: could not open output file %s\n", path);
        return SMK_FAILURE;
    }
    const std::string S4("    "), S8 = S4 + S4, S12 = S8 + S4, S16 = S12 + S4;
    const bool json = (format == 1);
    if (json) f << "{\n" << S4 << "\"doc_count\": " << t->leaf_doc_count << ",\n" << S4 << "\"nodes\": [\n";
    else f << "<?xml version=\"1.0\"?>\n<DataSet id=\"" << t->leaf_doc_count << "\">\n";
    for (size_t q = 0; q < t->nodes.size(); ++q) {
        const Node& nd = t->nodes[q];
        const int parent = (int)nd.parent, left = (int)nd.left, right = (int)nd.right;
        if (json) {
            if (q) f << ",\n";
            f << S8 << "{\n" << S12 << "\"id\": " << q << ",\n";
            f << S12 << "\"parent_id\": " << parent << ",\n";
            f << S12 << "\"left_child\": " << (nd.is_left ? "true" : "false") << ",\n";
            f << S12 << "\"left_child_id\": " << left << ",\n";
            f << S12 << "\"right_child_id\": " << right << ",\n";
            f << S12 << "\"doc_count\": " << nd.docs.size() << ",\n";
            if (!nd.terms.empty()) {
                f << S12 << "\"top_terms\": [\n";
                for (size_t i = 0; i < nd.terms.size(); ++i)
                    f << S16 << "\"" << dict[nd.terms[i]] << "\"" << (i + 1 < nd.terms.size() ? ",\n" : "\n");
                f << S12 << "]\n";
            }
            f << S8 << "}";
        } else {
            f << S4 << "<node id=\"" << q << "\">\n";
            f << S8 << "<parent_id>" << parent << "</parent_id>\n";
            f << S8 << "<left_child>" << (nd.is_left ? "true" : "false") << "</left_child>\n";
            f << S8 << "<left_child_id>" << left << "</left_child_id>\n";
            f << S8 << "<right_child_id>" << right << "</right_child_id>\n";
            f << S8 << "<doc_count>" << nd.docs.size() << "</doc_count>\n";
            f << S8 << "<top_terms>\n";
            for (int idx : nd.terms) f << S12 << "<term name=\"" << dict[idx] << "\"/>\n";
            f << S8 << "</top_terms>\n";
            f << S4 << "</node>\n";
        }
    }
    if (json) f << "\n" << S4 << "]\n}\n";
    else f << "</DataSet>\n";
    return f.good() ? SMK_OK : SMK_FAILURE;
}

// factors of the flat clustering that followed the tree search (opts.flat): W m x k, H k x n
int smk_tree_flat_factors(const smk_tree* t, double* W, int64_t ldW, double* H, int64_t ldH)
{
    if (!t || !W || !H) return SMK_BAD_PARAM;
    if (t->flat_k == 0) { set_error("no flat clustering result in this tree"); return SMK_BAD_PARAM; }
    const i64 m = t->term_count, n = t->doc_count, k = t->flat_k;
    if (ldW < m || ldH < k) return SMK_BAD_PARAM;
    for (i64 c = 0; c < k; ++c) std::copy(t->flatW.begin() + c * m, t->flatW.begin() + (c + 1) * m, W + c * ldW);
    for (i64 c = 0; c < n; ++c) std::copy(t->flatH.begin() + c * k, t->flatH.begin() + (c + 1) * k, H + c * ldH);
    return SMK_OK;
}

double smk_clust_priority(const double* w_parent, const double* w_child, int64_t n)
{
    if (!w_parent || !w_child || n <= 0) return -3.0;
    return priority_score(w_parent, w_child, n);
}

}  // extern "C"
