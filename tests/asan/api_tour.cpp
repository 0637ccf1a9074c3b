// tests/asan/api_tour.cpp -- TEST INFRASTRUCTURE ONLY.  A walk over namespace smallk (include/smallk.hpp) and the
// inner seam (include/nmf.hpp), built with -fsanitize=address,undefined against the stub device layer:
//     api_tour <datadir>     datadir holds a.csv, a.mtx, w_init.csv, h_init.csv, w_bad.csv, dictionary.txt
// Every error path is entered as well (the exceptions must come from consistent state, not from stray memory).
#include "smallk.hpp"
#include "nmf.hpp"

#include <cstdio>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

static int failures = 0;
#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); ++failures; } } while (0)
template <typename E, typename F> static void expect_throw(F f, const char* what)
{
    try { f(); std::fprintf(stderr, "no exception: %s\n", what); ++failures; }
    catch (const E&) {}
    catch (...) { std::fprintf(stderr, "wrong exception type: %s\n", what); ++failures; }
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string dir = std::string(argv[1]) + "/";
    // ---- before Initialize ----
    {
        NmfOptions o{};
        o.tol = 0.005; o.algorithm = NmfAlgorithm::MU; o.prog_est_algorithm = NmfProgressAlgorithm::DELTA_FNORM;
        o.height = 4; o.width = 3; o.k = 2; o.min_iter = 1; o.max_iter = 2; o.tolcount = 1; o.max_threads = 1; o.verbose = false; o.normalize = true;
        std::vector<double> A(12, 1.0), W(8, 1.0), H(6, 1.0);
        NmfStats st;
        CHECK(Nmf(o, A.data(), 4, W.data(), 4, H.data(), 2, st) == Result::NOTINITIALIZED);
    }
    smallk::Initialize(argc, argv);
    CHECK(smallk::IsInitialized());
    CHECK(smallk::GetVersionString() == "1.6.2");
    // setters / getters / clamps (smallk.cpp:391-468)
    smallk::SetMaxIter(0);       CHECK(smallk::GetMaxIter() == 1);
    smallk::SetMinIter(0);       CHECK(smallk::GetMinIter() == 1);
    smallk::SetOutputPrecision(99); CHECK(smallk::GetOutputPrecision() <= 17);
    expect_throw<std::logic_error>([] { smallk::SetNmfTolerance(1.5); }, "tolerance out of range");
    expect_throw<std::logic_error>([] { smallk::SetHierNmf2Tolerance(0.0); }, "hier tolerance out of range");
    expect_throw<std::logic_error>([] { smallk::SetOutputDir("/nonexistent/dir/"); }, "output dir");
    smallk::Reset();
    smallk::SetOutputDir(dir);
    smallk::SeedRNG(7);
    expect_throw<std::logic_error>([] { smallk::Nmf(3, smallk::Algorithm::BPP); }, "no matrix");
    expect_throw<std::runtime_error>([&] { smallk::LoadMatrix(dir + "missing.csv"); }, "missing file");
    CHECK(!smallk::IsMatrixLoaded());

    // ---- dense CSV ----
    smallk::LoadMatrix(dir + "a.csv");
    CHECK(smallk::IsMatrixLoaded());
    smallk::SetMinIter(2);
    smallk::SetMaxIter(6);
    for (smallk::Algorithm alg : {smallk::Algorithm::MU, smallk::Algorithm::HALS, smallk::Algorithm::BPP, smallk::Algorithm::RANK2}) {
        smallk::Nmf(3, alg);
        unsigned ld = 0, h = 0, w = 0;
        const double* pw = smallk::LockedBufferW(ld, h, w);
        CHECK(pw && ld == h && w == (alg == smallk::Algorithm::RANK2 ? 2u : 3u));
        double s = 0.0;
        for (unsigned i = 0; i < h * w; ++i) s += pw[i];          // touch every entry the API says exists
        const double* ph = smallk::LockedBufferH(ld, h, w);
        for (unsigned i = 0; i < h * w; ++i) s += ph[i];
        CHECK(s == s);
    }
    smallk::Nmf(3, smallk::Algorithm::BPP, dir + "w_init.csv", dir + "h_init.csv");
    expect_throw<std::logic_error>([&] { smallk::Nmf(3, smallk::Algorithm::BPP, dir + "w_bad.csv", dir + "h_init.csv"); }, "non-conformant W");
    {   // after the failed call the buffers still describe the previous factorisation: read them fully
        unsigned ld = 0, h = 0, w = 0;
        const double* pw = smallk::LockedBufferW(ld, h, w);
        double s = 0.0;
        for (unsigned i = 0; i < h * w; ++i) s += pw[i];
        const double* ph = smallk::LockedBufferH(ld, h, w);
        for (unsigned i = 0; i < h * w; ++i) s += ph[i];
        CHECK(s == s);
    }
    expect_throw<std::runtime_error>([&] { smallk::Nmf(3, smallk::Algorithm::BPP, dir + "missing.csv", ""); }, "missing init file");
    expect_throw<std::logic_error>([] { smallk::Nmf(0, smallk::Algorithm::MU); }, "k == 0");

    // ---- host buffers: dense (documented column-major h x w) and CSC ----
    {
        const unsigned h = 9, w = 5;
        std::vector<double> buf((size_t)(h + 2) * w);
        for (size_t i = 0; i < buf.size(); ++i) buf[i] = 0.1 + (double)(i % 7);
        smallk::LoadMatrix(buf.data(), h + 2, h, w);
        smallk::Nmf(2, smallk::Algorithm::HALS);
        std::vector<unsigned> co = {0, 2, 3, 3, 5, 6}, ri = {0, 3, 8, 1, 2, 4};
        std::vector<double> va = {1, 2, 3, 4, 5, 6};
        smallk::LoadMatrix(h, w, 6, va, ri, co);
        smallk::Nmf(2, smallk::Algorithm::MU);
    }

    // ---- sparse MatrixMarket + clustering ----
    smallk::LoadMatrix(dir + "a.mtx");
    smallk::Nmf(4, smallk::Algorithm::BPP);
    expect_throw<std::runtime_error>([&] { smallk::LoadDictionary(dir + "missing.txt"); }, "missing dictionary");
    smallk::LoadDictionary(dir + "dictionary.txt");
    smallk::SetMaxTerms(3);
    for (smallk::OutputFormat f : {smallk::OutputFormat::XML, smallk::OutputFormat::JSON}) {
        smallk::SetOutputFormat(f);
        smallk::HierNmf2(4);
    }
    try { smallk::HierNmf2WithFlat(3); } catch (const std::runtime_error&) {}    // may legitimately stop early
    expect_throw<std::exception>([] { smallk::HierNmf2(1); }, "one cluster");
    smallk::Reset();
    CHECK(!smallk::IsMatrixLoaded());
    smallk::Finalize();
    CHECK(!smallk::IsInitialized());
    std::printf("api_tour: %d failures\n", failures);
    return failures ? 1 : 0;
}
