// oracle/ref_flat_glue.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" handles onto the reference's own flat-clustering output code, compiled in place from
// /root/reference by `make -C oracle ref`:
//   common/src/flat_clust_output.cpp (FlatClustWriteResults :56-141), common/src/assignments.cpp,
//   common/src/flatclust_json_writer.cpp, flatclust_xml_writer.cpp, utils.cpp, constants.cpp,
//   common/include/assignments.hpp (ComputeAssignments / ComputeFuzzyAssignments templates).
#include <string>
#include <vector>
#include "assignments.hpp"
#include "file_format.hpp"
#include "flat_clust_output.hpp"

extern "C" int ref_flat_write_results(const char* assignfile, const char* fuzzyfile, const char* resultfile,
                                      const unsigned* assignments, unsigned num_assignments, const float* probs,
                                      unsigned num_probs, const char* const* dictionary, int dict_size,
                                      const int* term_indices, int num_terms, int json, unsigned maxterms,
                                      unsigned num_docs, unsigned num_clusters)
{
    std::vector<unsigned int> a(assignments, assignments + num_assignments);
    std::vector<float> p(probs, probs + num_probs);
    std::vector<std::string> d(dictionary, dictionary + dict_size);
    std::vector<int> t(term_indices, term_indices + num_terms);
    FlatClustWriteResults(std::string(assignfile), std::string(fuzzyfile), std::string(resultfile), a, p, d, t,
                          json ? FileFormat::JSON : FileFormat::XML, maxterms, num_docs, num_clusters);
    return 1;
}

extern "C" void ref_compute_assignments(const double* H, unsigned ldH, unsigned k, unsigned n, unsigned* out)
{
    std::vector<unsigned int> a;
    ComputeAssignments(a, H, ldH, k, n);
    for (unsigned i = 0; i < n; ++i) out[i] = a[i];
}

extern "C" void ref_compute_fuzzy(const double* H, unsigned ldH, unsigned k, unsigned n, float* out)
{
    std::vector<float> p;
    ComputeFuzzyAssignments(p, H, ldH, k, n);
    for (size_t i = 0; i < (size_t)k * n; ++i) out[i] = p[i];
}
