// mkl_spgemm_g4s.cpp — the reference benchmark's command line (mm/src/mkl_spgemm.cpp:5-87) on the device library:
//     ./mkl_spgemm_g4s [mat1 [mat2 [threads]]]
// argv as the reference reads it: no argument → can_24 · can_24 (:8-9); one → A = B = mat1 (:10-13); two or more → mat1 · mat2 (:14-17);
// a third argument (mm/README.md:9 passes a thread count) is accepted and ignored, as the reference ignores it (:61 hard-codes 14).
// A bare name is routed like the reference does (:18-37): *ER* → <dir>/ER/<name>.mtx, *G500* → <dir>/G500/<name>.mtx, otherwise
// <dir>/suite_sparse/<name>/<name>.mtx, with <dir> = $G4S_MATRIX_DIR or ../matrix; an argument that names an existing file is used as is
// (the reference ships no matrices). Then: load → CSR, make the shapes conformable (:47-57), total_flop = compute_flop(A,B) (:63),
// one warm-up mkl(A,B,C,timing) (:67), mean of 10 more (:72-81), Timings::print(2·flop) (:82) — the same bytes on stdout.
// Extensions after the reference's arguments: --dump prints C as "C row col value" lines; $G4S_BENCH_ITERS overrides the 10.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include "g4s/mtx.hpp"

static std::string route(const std::string &mat)
{
    if (std::ifstream(mat).good()) return mat;
    const char *e = std::getenv("G4S_MATRIX_DIR");
    const std::string dir = e ? e : "../matrix";
    if (mat.find("ER") != std::string::npos) return dir + "/ER/" + mat + ".mtx";
    if (mat.find("G500") != std::string::npos) return dir + "/G500/" + mat + ".mtx";
    return dir + "/suite_sparse/" + mat + "/" + mat + ".mtx";
}

int main(int argc, char **argv)
{
    bool dump = false;
    int nargs = 1;                                                  // the reference's argc: arguments before the first extension flag
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--dump")) dump = true;
        else if (!dump) nargs = i + 1;
    }
    std::string mat1 = "can_24", mat2 = "can_24";
    if (nargs == 2) { mat1 = argv[1]; mat2 = argv[1]; }
    if (nargs >= 3) { mat1 = argv[1]; mat2 = argv[2]; }
    std::printf("从文件 %s 读取矩阵A:\n", nargs >= 2 ? argv[1] : "(null)");   // :38 (glibc prints "(null)" for the reference's NULL argv[1])
    try {
        g4s::CSR<int32_t, double> A = g4s::read_matrix_market(route(mat1)), B, C;
        if (mat1 == mat2) B = A;
        else {
            B = g4s::read_matrix_market(route(mat2));
            if (A.cols < B.rows) B = g4s::leading_submatrix(B, A.cols, B.cols);          // :50-53
            else if (A.cols > B.rows) A = g4s::leading_submatrix(A, A.rows, B.rows);     // :54-57
        }
        const long long total_flop = g4s::get_flop(A, B);
        g4s::Timings timing, benchtiming;
        g4s::mkl(A, B, C, timing);
        C.make_empty();
        const char *it = std::getenv("G4S_BENCH_ITERS");
        const int iter = it ? std::atoi(it) : 10;
        for (int i = 0; i < iter; ++i) {
            g4s::mkl(A, B, C, timing);
            benchtiming += timing;
            if (i < iter - 1) C.make_empty();
        }
        benchtiming /= iter;
        benchtiming.print((double)(total_flop * 2));
        if (dump)
            for (int32_t r = 0; r < C.rows; ++r)
                for (int32_t k = C.rowptr[r]; k < C.rowptr[r + 1]; ++k) std::printf("C %d %d %.17g\n", r, C.colids[k], C.values[k]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
