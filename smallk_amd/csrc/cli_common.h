// smallk_amd/csrc/cli_common.h -- helpers shared by the hierclust / flatclust command line tools.
#pragma once
#include <sys/stat.h>

#include <algorithm>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/smallk_amd.h"

namespace cli {

inline std::string upper(std::string s) { std::transform(s.begin(), s.end(), s.begin(), ::toupper); return s; }

inline bool has_ext(const std::string& path, const char* ext)
{
    const size_t dot = path.find_last_of('.');
    return dot != std::string::npos && upper(path.substr(dot + 1)) == ext;
}

inline std::string ensure_trailing_sep(const std::string& s)     // EnsureTrailingPathSep, common/src/utils.cpp
{
    if (s.empty() || s.back() == '/') return s;
    return s + "/";
}

inline bool directory_exists(const std::string& d)
{
    struct stat st;
    return stat(d.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

inline int hw_threads()
{
    const int hw = (int)std::thread::hardware_concurrency();
    return hw <= 0 ? 2 : hw;
}

// LoadStringsFromFile, common/src/utils.cpp:220-239
inline bool load_strings(const std::string& path, std::vector<std::string>& out)
{
    std::ifstream in(path);
    if (!in) return false;
    std::string line;
    while (in) {
        std::getline(in, line, '\n');
        if (in.eof()) break;
        out.push_back(line);
    }
    return true;
}

inline bool load_csv(const std::string& path, std::vector<double>& buf, unsigned& h, unsigned& w)
{
    double dummy;
    if (smk_load_csv(path.c_str(), &dummy, 0, &h, &w) == 0) return false;
    buf.assign((size_t)h * w, 0.0);
    return smk_load_csv(path.c_str(), buf.data(), (unsigned long)buf.size(), &h, &w) == 1;
}

struct InputMatrix {
    bool sparse = false;
    unsigned m = 0, n = 0, nnz = 0;
    std::vector<double> dense, data;
    std::vector<unsigned> rows, cols;
};

// .mtx -> sparse CSC, .csv -> dense column-major (IsSparse / IsDense, common/src/file_loader.cpp)
inline int load_matrix(const std::string& path, InputMatrix& a)
{
    if (has_ext(path, "MTX")) {
        if (smk_load_matrix_market(path.c_str(), &a.m, &a.n, &a.nnz, nullptr, nullptr, nullptr) != 1) return -1;
        a.cols.resize((size_t)a.n + 1); a.rows.resize(a.nnz); a.data.resize(a.nnz);
        if (smk_load_matrix_market(path.c_str(), &a.m, &a.n, &a.nnz, a.cols.data(), a.rows.data(), a.data.data()) != 1)
            return -1;
        a.sparse = true;
        return 0;
    }
    if (has_ext(path, "CSV")) return load_csv(path, a.dense, a.m, a.n) && a.dense.size() >= (size_t)a.m * a.n ? 0 : -1;
    return -2;      // unsupported file type
}

inline std::string elapsed_ms_string(double ms)
{
    char buf[64];
    if (ms < 1000.0) snprintf(buf, sizeof buf, "%g ms.", ms);
    else snprintf(buf, sizeof buf, "%g s.", ms * 0.001);
    return buf;
}

}  // namespace cli
