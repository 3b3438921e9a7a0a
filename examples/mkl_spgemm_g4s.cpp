// mkl_spgemm_g4s.cpp — the reference benchmark's command line on the device library:
//     ./mkl_spgemm_g4s matA.mtx [matB.mtx] [runs]
// (mm/src/mkl_spgemm.cpp:5-87: load .mtx → CSR, make the shapes conformable, 1 warm-up + mean of 10 runs of mkl(A,B,C,timing),
// print the stage table and GFLOPS = 2·flop/t). The reference routes bare matrix names into ../matrix/{ER,G500,suite_sparse}
// (:18-37) — directories it does not ship — so this driver takes file paths. `--dump` prints C as "row col value" lines.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "g4s/mtx.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s matA.mtx [matB.mtx] [runs] [--dump]\n", argv[0]); return 2; }
    bool dump = false;
    int runs = 10;
    std::string fa = argv[1], fb = argv[1];
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--dump")) dump = true;
        else if (std::strspn(argv[i], "0123456789") == std::strlen(argv[i])) runs = std::atoi(argv[i]);
        else fb = argv[i];
    }
    try {
        g4s::CSR<int32_t, double> A = g4s::read_matrix_market(fa), B = g4s::read_matrix_market(fb), C;
        if (A.cols != B.rows) {                         // mkl_spgemm.cpp:47-57: cut both to the common inner dimension
            const int32_t k = A.cols < B.rows ? A.cols : B.rows;
            A = g4s::leading_submatrix(A, A.rows, k);
            B = g4s::leading_submatrix(B, k, B.cols);
        }
        std::printf("A: %d x %d nnz %d   B: %d x %d nnz %d\n", A.rows, A.cols, A.nnz, B.rows, B.cols, B.nnz);
        const long long flop = g4s::get_flop(A, B);
        g4s::Timings timing, bench;
        g4s::mkl(A, B, C, timing);
        for (int i = 0; i < runs; ++i) { g4s::mkl(A, B, C, timing); bench += timing; }
        if (runs > 0) bench /= runs;
        std::printf("C: %d x %d nnz %d   flop %lld   compression %.3f\n", C.rows, C.cols, C.nnz, flop, C.nnz ? (double)flop / C.nnz : 0.0);
        bench.print(2.0 * (double)flop);
        if (dump)
            for (int32_t r = 0; r < C.rows; ++r)
                for (int32_t k = C.rowptr[r]; k < C.rowptr[r + 1]; ++k) std::printf("C %d %d %.17g\n", r, C.colids[k], C.values[k]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
