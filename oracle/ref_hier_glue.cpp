// oracle/ref_hier_glue.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" handles onto the reference's own HierNMF2 file writers and SetDiff, compiled in place
// from /root/reference by `make -C oracle ref`:
//   hierclust/src/hierclust_json_writer.cpp, hierclust_xml_writer.cpp, hierclust_writer_factory.cpp
//   hierclust/include/setdiff.hpp:23-46
// Tree<T> itself (hierclust/include/tree.hpp) includes dense_matrix.hpp -> Elemental and cannot be
// built here, so ref_write_tree() replays its node loop (Tree::WriteTree / WriteNodes, tree.hpp:426-465)
// over plain arrays and calls the reference writer objects for every byte that reaches the file.
#include <fstream>
#include <string>
#include <vector>
#include "hierclust_writer_factory.hpp"
#include "setdiff.hpp"

extern "C" int ref_write_tree(const char* path, int json, int leaf_doc_count, int node_count,
                              const unsigned* parent, const int* is_left, const unsigned* left,
                              const unsigned* right, const int* doc_count, const int* term_offsets,
                              const int* terms, const char* const* dictionary, int dict_size)
{
    std::vector<std::string> dict(dictionary, dictionary + dict_size);
    IHierclustWriter* w = CreateHierclustWriter(json ? FileFormat::JSON : FileFormat::XML);
    std::ofstream out(path);
    if (!out) return 0;
    w->WriteHeader(out, leaf_doc_count);
    for (int q = 0; q < node_count; ++q) {
        std::vector<int> t(terms + term_offsets[q], terms + term_offsets[q + 1]);
        w->WriteNodeBegin(out, q);
        w->WriteParentId(out, parent[q]);
        w->WriteLeftChild(out, is_left[q] != 0, left[q]);
        w->WriteRightChild(out, right[q]);
        w->WriteDocCount(out, doc_count[q]);
        w->WriteTopTerms(out, t, dict);
        w->WriteNodeEnd(out);
    }
    w->WriteFooter(out);
    out.close();
    delete w;
    return 1;
}

// result capacity >= na
extern "C" int ref_setdiff(const unsigned* a, int na, const unsigned* b, int nb, unsigned* out)
{
    std::vector<unsigned> A(a, a + na), B(b, b + nb);
    std::vector<unsigned> r = SetDiff(A, B);
    for (size_t i = 0; i < r.size(); ++i) out[i] = r[i];
    return (int)r.size();
}
