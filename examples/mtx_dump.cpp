// mtx_dump.cpp — prints the CSR a .mtx file reads into ("rows cols nnz", then rowptr, then "col value" lines): used by the CPU
// tests to compare g4s::read_matrix_market with the oracle's restatement of CSR::construct (no GPU needed).
#include <cstdio>
#include "g4s/mtx.hpp"
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    try {
        auto A = g4s::read_matrix_market(argv[1]);
        std::printf("%d %d %d\n", A.rows, A.cols, A.nnz);
        for (int32_t r = 0; r <= A.rows; ++r) std::printf("%d\n", A.rowptr[r]);
        for (int32_t k = 0; k < A.nnz; ++k) std::printf("%d %.17g\n", A.colids[k], A.values[k]);
    } catch (const std::exception &e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
