// spgemm_driver.cpp — the shipped benchmark's protocol (mm/src/mkl_spgemm.cpp:60-85: 1 warm-up + mean of 10 runs of
// mkl(A,B,C,timing), stage table, GFLOPS = 2·flop/t) on the device library, plus the OptMatmul use of GraphProcess
// (deepmd/source/op/opt_matmul.cc:43-61). Input: a small synthetic banded matrix (the reference's .mtx inputs are not shipped).
// Build: g++ -std=c++17 -O2 -Iinclude examples/spgemm_driver.cpp -Lg4s_amd/lib -lg4s_hip -Wl,-rpath,$PWD/g4s_amd/lib -o spgemm_driver
#include <cmath>
#include <cstdio>
#include <vector>
#include "g4s/csr.hpp"
#include "g4s/graph.hpp"

int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 20000, hb = 4;
    std::vector<int32_t> rp(n + 1), ci;
    std::vector<double> va;
    for (int i = 0; i < n; ++i) {
        rp[i] = (int32_t)ci.size();
        for (int c = std::max(0, i - hb); c <= std::min(n - 1, i + hb); ++c) { ci.push_back(c); va.push_back(1.0 / (1 + std::abs(i - c))); }
    }
    rp[n] = (int32_t)ci.size();
    g4s::CSR<int32_t, double> A(rp.data(), ci.data(), va.data(), n, n, rp[n]), C;
    const long long flop = g4s::get_flop(A, A);
    g4s::Timings timing, bench;
    g4s::mkl(A, A, C, timing);                               // warm-up (mkl_spgemm.cpp:67)
    for (int i = 0; i < 10; ++i) { g4s::mkl(A, A, C, timing); bench += timing; }
    bench /= 10;
    std::printf("A: %d x %d nnz %d; C nnz %d; flop %lld\n", A.rows, A.cols, A.nnz, C.nnz, flop);
    bench.print(2.0 * (double)flop);
    // check one row against the closed form of a banded product: row i of A·A spans columns i-2hb .. i+2hb
    const int i = n / 2;
    bool ok = C.rowptr[i + 1] - C.rowptr[i] == 4 * hb + 1 && C.colids[C.rowptr[i]] == i - 2 * hb;
    // y = A·1 through the mv-shaped call
    std::vector<double> x(n, 1.0), y(n, -1.0);
    g4s::SpMV(A, x.data(), y.data());
    double want = 0; for (int c = -hb; c <= hb; ++c) want += 1.0 / (1 + std::abs(c));
    ok = ok && std::fabs(y[i] - want) < 1e-12;
    // OptMatmul through GraphProcess: result[M×K] = xx[M×N]·w[N×K]
    const int M = 100, N = 20, K = 30;
    std::vector<double> xx(M * N), w(N * K), res(M * K, 0.0);
    for (int k = 0; k < M * N; ++k) xx[k] = (k % 7) - 3;
    for (int k = 0; k < N * K; ++k) w[k] = (k % 5) - 2;
    std::vector<const double *> rows(M);
    for (int e = 0; e < M; ++e) rows[e] = xx.data() + (size_t)e * N;
    Graph graph{M, K, rows.data(), w.data(), nullptr};
    g4s_pattern_desc pat{};
    pat.kind = G4S_PATTERN_DENSE_ROW_TIMES_MATRIX; pat.inner = N;
    g4s::GraphProcess(&graph, res.data(), [](int, int, Graph *, double *) {}, [](int, Graph *, double *) {}, pat);
    double ref = 0; for (int k = 0; k < N; ++k) ref += xx[5 * N + k] * w[k * K + 7];
    ok = ok && res[5 * K + 7] == ref;
    std::printf("%s\n", ok ? "CHECK OK" : "CHECK FAILED");
    return ok ? 0 : 1;
}
