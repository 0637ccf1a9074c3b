// oracle/ref_csv_glue.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" handles onto the reference's own CSV templates
// (common/include/delimited_file.hpp:49-76 WriteDelimitedFile,
//  :79-135 LoadDelimitedFile), compiled in place from /root/reference by
// `make -C oracle ref`.  Used by tests/test_csv_io.py to byte-compare the
// product's w.csv/h.csv writer and reader with the reference's.
#include <vector>
#include <string>
#include <cstring>
#include "delimited_file.hpp"

extern "C" int ref_write_csv(const double* buf, unsigned ldim, unsigned height, unsigned width,
                             const char* filename, unsigned precision)
{
    return WriteDelimitedFile<double>(buf, ldim, height, width, std::string(filename), precision) ? 1 : 0;
}

// Loads into a caller buffer of capacity `cap` doubles (column-major, ldim = height).
extern "C" int ref_load_csv(const char* filename, double* out, unsigned long cap,
                            unsigned* height, unsigned* width)
{
    std::vector<double> v;
    unsigned h = 0, w = 0;
    if (!LoadDelimitedFile<double>(v, h, w, std::string(filename))) return 0;
    *height = h; *width = w;
    if ((unsigned long)h * w > cap) return -1;
    std::memcpy(out, v.data(), sizeof(double) * (size_t)h * w);
    return 1;
}
