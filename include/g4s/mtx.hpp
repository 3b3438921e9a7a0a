// g4s/mtx.hpp — MatrixMarket coordinate reader with the semantics of CSR<IT,NT>::construct (mm/inc/CSR.h:485-669; banner rules
// :441-478), written for this project (host-side I/O; the GPU library never parses files):
//   * banner "%%MatrixMarket matrix coordinate <real|integer|pattern|complex> <general|symmetric|skew-symmetric>"; "array" storage,
//     "vector" objects and "hermitian" symmetry are rejected, as in the reference;
//   * pattern entries get the value 1; complex entries keep their real part; indices are 1-based in the file, 0-based in memory;
//   * symmetric / skew-symmetric files are expanded: every off-diagonal (i,j,v) also yields (j,i,v) resp. (j,i,−v);
//   * entries are ordered by (row, column) — the reference sorts the fused key cols·i + j; duplicates are kept, not merged.
#pragma once
#include <algorithm>
#include <cstdint>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "csr.hpp"

namespace g4s {

inline CSR<int32_t, double> read_matrix_market(const std::string &filename)
{
    std::ifstream in(filename);
    if (!in) throw std::runtime_error("unable to open file \"" + filename + "\" for reading");
    std::string line, tag, object, storage, field, symmetry, extra;
    std::getline(in, line);
    {
        std::istringstream b(line);
        b >> tag >> object >> storage >> field >> symmetry;
        if (!b || (b >> extra) || tag != "%%MatrixMarket" || object != "matrix") throw std::runtime_error("invalid MatrixMarket banner");
    }
    if (storage != "coordinate") throw std::runtime_error("not impl storage type " + storage);
    const bool pattern = field == "pattern", cplx = field == "complex";
    if (!pattern && !cplx && field != "real" && field != "integer") throw std::runtime_error("invalid MatrixMarket data type [" + field + "]");
    int sym = 0;
    if (symmetry == "symmetric") sym = 1;
    else if (symmetry == "skew-symmetric") sym = 2;
    else if (symmetry != "general") throw std::runtime_error("not impl matrix type: " + symmetry);
    do {
        if (!std::getline(in, line)) throw std::runtime_error("invalid MatrixMarket coordinate format");
    } while (!line.empty() && line[0] == '%');
    long long rows = 0, cols = 0, declared = 0;
    {
        std::istringstream sz(line);
        if (!(sz >> rows >> cols >> declared) || (sz >> extra)) throw std::runtime_error("invalid MatrixMarket coordinate format");
    }
    struct Entry { long long key; double v; long long seq; };
    std::vector<Entry> e;
    e.reserve((size_t)declared * (sym ? 2 : 1));
    for (long long k = 0; k < declared; ++k) {
        long long i, j;
        double v = 1.0, im;
        if (!(in >> i >> j)) throw std::runtime_error("read nnz not equal to declared nnz " + std::to_string(k));
        if (cplx) { if (!(in >> v >> im)) throw std::runtime_error("read nnz not equal to declared nnz " + std::to_string(k)); }
        else if (!pattern) { if (!(in >> v)) throw std::runtime_error("read nnz not equal to declared nnz " + std::to_string(k)); }
        --i; --j;
        e.push_back({cols * i + j, v, (long long)e.size()});
        if (sym && i != j) e.push_back({cols * j + i, sym == 2 ? -v : v, (long long)e.size()});
    }
    std::sort(e.begin(), e.end(), [](const Entry &a, const Entry &b) { return a.key != b.key ? a.key < b.key : a.seq < b.seq; });
    std::vector<int32_t> rp((size_t)rows + 1, 0), ci(e.size());
    std::vector<double> va(e.size());
    for (size_t k = 0; k < e.size(); ++k) {
        rp[(size_t)(e[k].key / cols) + 1]++;
        ci[k] = (int32_t)(e[k].key % cols);
        va[k] = e[k].v;
    }
    for (long long r = 0; r < rows; ++r) rp[r + 1] += rp[r];
    return CSR<int32_t, double>(rp.data(), ci.data(), va.data(), (int32_t)rows, (int32_t)cols, (int32_t)e.size());
}

// Leading M×N sub-matrix — the shape fix-up the driver applies so that A·B is conformable (mm/inc/CSR.h:691-733, used at
// mm/src/mkl_spgemm.cpp:47-57).
inline CSR<int32_t, double> leading_submatrix(const CSR<int32_t, double> &a, int32_t M, int32_t N)
{
    std::vector<int32_t> rp((size_t)M + 1, 0), ci;
    std::vector<double> va;
    for (int32_t r = 0; r < M && r < a.rows; ++r) {
        for (int32_t k = a.rowptr[r]; k < a.rowptr[r + 1]; ++k)
            if (a.colids[k] < N) { ci.push_back(a.colids[k]); va.push_back(a.values[k]); }
        rp[r + 1] = (int32_t)ci.size();
    }
    for (int32_t r = a.rows; r < M; ++r) rp[r + 1] = (int32_t)ci.size();
    return CSR<int32_t, double>(rp.data(), ci.data(), va.data(), M, N, (int32_t)ci.size());
}

} // namespace g4s
