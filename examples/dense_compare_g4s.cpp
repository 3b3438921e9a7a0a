// dense_compare_g4s.cpp — the reference's dense comparison programs on the device library:
//   mm/src/cblas_dxxmm.c:57-111,199-206   dsymm, dtrmm, dgemm: 10 repetitions each, mean wall time, one line per routine
//   mv/mv.c:6-27,69-94                    dsymv, dtrmv, dspmv ("sspmv" in the source), dgemv: one shot each
// Usage: dense_compare_g4s <dim>. The reference fills A from a .mtx file (into otherwise uninitialised memory) and B with rand(); here both
// are seeded so that runs repeat. The printed lines keep the reference's wording ("… 运行时间：%f 毫秒").
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "g4s.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int dim = argc > 1 ? std::atoi(argv[1]) : 1000;
    if (dim <= 0) { std::fprintf(stderr, "usage: %s dim\n", argv[0]); return 2; }
    const size_t nn = (size_t)dim * dim;
    std::vector<double> A(nn), B(nn), C(nn), x(dim), y(dim), AP((size_t)dim * (dim + 1) / 2);
    std::srand(1);
    for (auto &v : A) v = std::rand() / (double)RAND_MAX - 0.5;
    for (auto &v : B) v = std::rand() / (double)RAND_MAX - 0.5;
    for (auto &v : x) v = 1.0;                                      // mv.c:64-66
    for (auto &v : AP) v = std::rand() / (double)RAND_MAX - 0.5;
    // device-resident operands: the timed region is the routine, as the reference times the BLAS call alone
    void *dA = nullptr, *dB = nullptr, *dC = nullptr, *dx = nullptr, *dy = nullptr, *dAP = nullptr;
    if (g4s_dev_alloc(&dA, nn * 8) || g4s_dev_alloc(&dB, nn * 8) || g4s_dev_alloc(&dC, nn * 8) || g4s_dev_alloc(&dx, dim * 8) || g4s_dev_alloc(&dy, dim * 8) ||
        g4s_dev_alloc(&dAP, AP.size() * 8)) { std::fprintf(stderr, "error: %s\n", g4s_last_error()); return 1; }
    g4s_memcpy_h2d(dA, A.data(), nn * 8); g4s_memcpy_h2d(dB, B.data(), nn * 8); g4s_memcpy_h2d(dx, x.data(), dim * 8); g4s_memcpy_h2d(dAP, AP.data(), AP.size() * 8);
    struct { const char *name; int kind; } mm[] = {{"cblas_dsymm", G4S_DENSE_DSYMM}, {"cblas_dtrmm", G4S_DENSE_DTRMM}, {"cblas_dgemm", G4S_DENSE_DGEMM}};
    for (auto &r : mm) {
        if (g4s_dense_mm(r.kind, dim, (const double *)dA, (double *)dB, (double *)dC, G4S_DEVICE_POINTERS)) { std::fprintf(stderr, "error: %s\n", g4s_last_error()); return 1; }
        g4s_memcpy_h2d(dB, B.data(), nn * 8);                       // dtrmm works in place: restore B outside the timed region
        double total = 0.0;
        for (int it = 0; it < 10; ++it) {                           // cblas_dxxmm.c:66-75
            const double t0 = now_ms();
            g4s_dense_mm(r.kind, dim, (const double *)dA, (double *)dB, (double *)dC, G4S_DEVICE_POINTERS);
            total += now_ms() - t0;
            if (r.kind == G4S_DENSE_DTRMM) g4s_memcpy_h2d(dB, B.data(), nn * 8);
        }
        std::printf("%s 运行时间：%f 毫秒\n", r.name, total / 10);
        std::printf("    (%.1f GFLOPS fp64, dim %d)\n", 2.0 * dim * (double)dim * dim / (total / 10) / 1e6, dim);
    }
    struct { const char *name; int kind; const void *a; } mv[] = {{"matrix_multiply_dsymv", G4S_DENSE_DSYMV, dA}, {"matrix_multiply_dtrmv", G4S_DENSE_DTRMV, dA},
                                                                 {"matrix_multiply_sspmv", G4S_DENSE_DSPMV, dAP}, {"matrix_multiply_dgemv", G4S_DENSE_DGEMV, dA}};
    for (auto &r : mv) {
        g4s_dense_mv(r.kind, dim, (const double *)r.a, (double *)dx, (double *)dy, G4S_DEVICE_POINTERS);      // page-in
        g4s_memcpy_h2d(dx, x.data(), dim * 8);
        const double t0 = now_ms();
        if (g4s_dense_mv(r.kind, dim, (const double *)r.a, (double *)dx, (double *)dy, G4S_DEVICE_POINTERS)) { std::fprintf(stderr, "error: %s\n", g4s_last_error()); return 1; }
        const double t = now_ms() - t0;
        g4s_memcpy_h2d(dx, x.data(), dim * 8);
        std::printf("%s 运行时间：%f毫秒\n", r.name, t);               // mv.c:73 (no space before 毫秒 there)
    }
    g4s_memcpy_d2h(y.data(), dy, dim * 8);
    double want = 0.0;                                              // dgemv ran last: y = A·1 = row sums
    for (int j = 0; j < dim; ++j) want += A[(size_t)j * dim];
    std::printf("%s\n", std::abs(y[0] - want) <= 1e-9 * dim ? "CHECK OK" : "CHECK FAILED");
    for (void *p : {dA, dB, dC, dx, dy, dAP}) g4s_dev_free(p);
    return 0;
}
