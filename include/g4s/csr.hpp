// g4s/csr.hpp — header-only C++ host side over the C-ABI (include/g4s.h), keeping the reference's spelling:
//   CSR<IT,NT>                         mm/inc/CSR.h:22-113   (rows, cols, nnz, rowptr, colids, values, zerobased; owns its arrays)
//   HashSpGEMM<vectorProbing,sortOutput>(a,b,c,multop,addop)   mm/inc/hash_mult.h:1028-1057, short forms :1103-1113
//   mkl(a,b,c,timing)                  mm/inc/mkl_mult.h:113-124 (the call the shipped benchmark times)
//   Timings                            mm/inc/Timings.h:4-23, mm/src/Timings.cpp:36-65
//   SpMV(a,x,y,alpha,beta)             the CSR mat-vec this build defines for mv/ 
// Only IT = int32_t, NT = double exist in the reference (mm/inc/define.h:14-15) and on the device. Arrays handed back by the
// library are allocated with g4s_malloc and released with g4s_free (the my_malloc/my_free pairing of mm/inc/utility.h:126-153).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include "../g4s.h"

namespace g4s {

inline void check(g4s_status st, const char *what)
{
    if (st != G4S_OK) throw std::runtime_error(std::string(what) + ": " + g4s_last_error());
}

struct Timings : g4s_timings {
    Timings() { reset(); }
    void reset() { create = spmm = convert = order = export_csr = destroy = total = 0.0; }
    void operator+=(const Timings &b)
    {
        create += b.create; spmm += b.spmm; convert += b.convert; order += b.order;
        export_csr += b.export_csr; destroy += b.destroy; total += b.total;
    }
    void operator/=(double x)
    {
        create /= x; spmm /= x; convert /= x; order /= x; export_csr /= x; destroy /= x; total /= x;
    }
    // Timings::print, byte for byte (mm/src/Timings.cpp:36-60). The reference keeps seconds and prints 1000·t; g4s_timings holds
    // milliseconds, so t below is field/1000. Called with total_flop = 2·flop (mm/src/mkl_spgemm.cpp:82). Percentages are of `total`,
    // the wall time around the whole call, not of the sum of the stages; perf lines divide flop/1e9 by seconds (inf for a zero stage,
    // as in the reference).
    bool measure_separate = true, measure_total = true;   // mm/inc/Timings.h:6-7
    void print(double total_flop) const
    {
        const double total_flop_G = total_flop / 1000000000;
        std::printf("total flop %lf\n", total_flop);
        const double c = create / 1000, s = spmm / 1000, v = convert / 1000, o = order / 1000, e = export_csr / 1000, d = destroy / 1000, t = total / 1000;
        const double sum_total = c + s + v + o + e + d;
        if (measure_separate) {
            std::printf("time(ms):\n");
            std::printf("    create             %8.3lfms %6.2lf%%\n", 1000 * c, c / t * 100);
            std::printf("    spmm               %8.3lfms %6.2lf%%\n", 1000 * s, s / t * 100);
            std::printf("    convert            %8.3lfms %6.2lf%%\n", 1000 * v, v / t * 100);
            std::printf("    order              %8.3lfms %6.2lf%%\n", 1000 * o, o / t * 100);
            std::printf("    export_csr         %8.3lfms %6.2lf%%\n", 1000 * e, e / t * 100);
            std::printf("    destroy            %8.3lfms %6.2lf%%\n", 1000 * d, d / t * 100);
            std::printf("    sum_total          %8.3lfms %6.2lf%%\n", 1000 * sum_total, sum_total / t * 100);
            std::printf("perf(Gflops):\n");
            std::printf("    create             %6.2lf\n", total_flop_G / c);
            std::printf("    spmm               %6.2lf\n", total_flop_G / s);
            std::printf("    convert            %6.2lf\n", total_flop_G / v);
            std::printf("    order              %6.2lf\n", total_flop_G / o);
            std::printf("    export_csr         %6.2lf\n", total_flop_G / e);
            std::printf("    destroy            %6.2lf\n", total_flop_G / d);
            std::printf("    total              %6.2lf\n", total_flop_G / t);
        }
    }
    void reg_print(double total_flop) const                // mm/src/Timings.cpp:62-65
    {
        const double total_flop_G = total_flop / 1000000000;
        std::printf("%le\n", total_flop_G / (total / 1000));
    }
};

template <class IT = int32_t, class NT = double>
class CSR {
    static_assert(std::is_same<IT, int32_t>::value && std::is_same<NT, double>::value, "the reference and the device use int32 / fp64 only");
public:
    CSR() : rows(0), cols(0), nnz(0), rowptr(nullptr), colids(nullptr), values(nullptr), zerobased(true) {}
    // raw-array constructor: copies, like mm/inc/CSR.h:102-113
    CSR(const IT *rp, const IT *ci, const NT *va, IT M, IT N, IT nz) : rows(M), cols(N), nnz(nz), zerobased(true)
    {
        rowptr = (IT *)g4s_malloc(sizeof(IT) * ((size_t)M + 1));
        colids = (IT *)g4s_malloc(sizeof(IT) * (size_t)nz);
        values = (NT *)g4s_malloc(sizeof(NT) * (size_t)nz);
        std::memcpy(rowptr, rp, sizeof(IT) * ((size_t)M + 1));
        std::memcpy(colids, ci, sizeof(IT) * (size_t)nz);
        std::memcpy(values, va, sizeof(NT) * (size_t)nz);
    }
    CSR(const CSR &o) : CSR(o.rowptr, o.colids, o.values, o.rows, o.cols, o.nnz) {}
    CSR &operator=(const CSR &o) { if (this != &o) { CSR t(o); swap(t); } return *this; }
    CSR(CSR &&o) noexcept : CSR() { swap(o); }
    CSR &operator=(CSR &&o) noexcept { swap(o); return *this; }
    ~CSR() { make_empty(); }
    void make_empty()                                   // mm/inc/CSR.h:50-62
    {
        g4s_free(rowptr); g4s_free(colids); g4s_free(values);
        rowptr = colids = nullptr; values = nullptr; rows = cols = nnz = 0;
    }
    void swap(CSR &o) noexcept
    {
        std::swap(rows, o.rows); std::swap(cols, o.cols); std::swap(nnz, o.nnz); std::swap(rowptr, o.rowptr);
        std::swap(colids, o.colids); std::swap(values, o.values); std::swap(zerobased, o.zerobased);
    }
    IT rows, cols, nnz;
    IT *rowptr, *colids;
    NT *values;
    bool zerobased;
};

// C = A·B on the device, result adopted by `c`. Only the arithmetic semiring can run on the GPU; any other functor is refused.
template <bool vectorProbing = false, bool sortOutput = true, typename IT, typename NT, typename Mul, typename Add>
void HashSpGEMM(const CSR<IT, NT> &a, const CSR<IT, NT> &b, CSR<IT, NT> &c, Mul, Add, Timings *timing = nullptr)
{
    static_assert(std::is_same<Mul, std::multiplies<NT>>::value && std::is_same<Add, std::plus<NT>>::value,
                  "device SpGEMM implements multiplies/plus only (the functors every call site of the reference passes)");
    c.make_empty();
    int64_t cnnz = 0;
    check(g4s_spgemm_csr_i32_f64(a.rowptr, a.colids, a.values, b.rowptr, b.colids, b.values, &c.rowptr, &c.colids, &c.values,
                                 a.rows, a.cols, b.cols, &cnnz, timing, sortOutput ? G4S_SORT_OUTPUT : 0u),
          "HashSpGEMM");
    c.rows = a.rows; c.cols = b.cols; c.nnz = (IT)cnnz; c.zerobased = true;
}
template <typename IT, typename NT>
void HashSpGEMM(const CSR<IT, NT> &a, const CSR<IT, NT> &b, CSR<IT, NT> &c) { HashSpGEMM<false, true>(a, b, c, std::multiplies<NT>(), std::plus<NT>()); }

// The wrapper the shipped benchmark calls: mkl(A,B,C,timing) (mm/inc/mkl_mult.h:113-124 ← mm/src/mkl_spgemm.cpp:67,74).
template <typename IT, typename NT>
void mkl(const CSR<IT, NT> &a, const CSR<IT, NT> &b, CSR<IT, NT> &c, Timings &timing) { HashSpGEMM<false, true>(a, b, c, std::multiplies<NT>(), std::plus<NT>(), &timing); }

template <typename IT, typename NT>
long long get_flop(const CSR<IT, NT> &a, const CSR<IT, NT> &b)   // mm/inc/hash_mult.h:46-62
{
    int64_t flop = 0;
    check(g4s_spgemm_flop(a.rows, a.rowptr, a.colids, b.rowptr, &flop, nullptr, G4S_HOST_POINTERS), "get_flop");
    return flop;
}

template <typename IT, typename NT>
void SpMV(const CSR<IT, NT> &a, const NT *x, NT *y, NT alpha = 1.0, NT beta = 0.0)
{
    check(g4s_spmv_csr_i32_f64(a.rows, a.cols, a.rowptr, a.colids, a.values, x, y, alpha, beta, G4S_HOST_POINTERS), "SpMV");
}

} // namespace g4s
