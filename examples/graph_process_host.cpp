// graph_process_host.cpp — the reference's four-argument GraphProcess spelling (deepmd/source/op/graph.h:21) against include/g4s/graph.hpp: the graph of the
// OptMatmul call site (deepmd/source/op/opt_matmul.cc:43-61: numNodes = M, degree = K, edgeWeight = row pointers into xx, states = w) built the way that call
// site builds it, the gather written here. Reads M N K and the two operands from stdin (binary doubles after the text header line), writes result[M·K] to stdout
// as binary doubles; "parallel" as argv[1] declares the gather race-free (it is: every (vertex, neighbour) writes its own slot).
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>
#include "g4s/graph.hpp"

int main(int argc, char **argv)
{
    int M = 0, N = 0, K = 0;
    if (scanf("%d %d %d\n", &M, &N, &K) != 3 || M < 0 || N < 0 || K < 0) return 2;
    std::vector<double> xx((size_t)M * N), w((size_t)N * K), res((size_t)M * K, -1.0);
    if (fread(xx.data(), sizeof(double), xx.size(), stdin) != xx.size() || fread(w.data(), sizeof(double), w.size(), stdin) != w.size()) return 2;
    if (argc > 1 && !strcmp(argv[1], "parallel")) g4s_set_host_callback_policy(G4S_HOST_CALLBACKS_PARALLEL);
    if (argc > 1 && !strcmp(argv[1], "refuse")) g4s_set_host_callback_policy(G4S_HOST_CALLBACKS_REFUSE);
    // "scoped_race_free": the declaration as a scope around the untouched call (g4s::ScopedRaceFree) instead of the process-wide policy;
    // "scoped_pattern": g4s::ScopedPattern names the call's pattern — the same untouched call line then runs the fp64 MFMA kernel (needs a GPU)
    const bool scoped_rf = argc > 1 && !strcmp(argv[1], "scoped_race_free"), scoped_pat = argc > 1 && !strcmp(argv[1], "scoped_pattern");
    g4s_pattern_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.kind = G4S_PATTERN_DENSE_ROW_TIMES_MATRIX;
    desc.inner = N;
    std::unique_ptr<g4s::ScopedRaceFree> rf(scoped_rf ? new g4s::ScopedRaceFree() : nullptr);
    std::unique_ptr<g4s::ScopedPattern> pat(scoped_pat ? new g4s::ScopedPattern(desc) : nullptr);
    struct Graph graph;
    graph.states = w.data();
    graph.numNodes = M;
    graph.degree = K;
    graph.temp = nullptr;
    std::vector<const double *> rows((size_t)M);
    for (int i = 0; i < M; i++) rows[i] = xx.data() + (size_t)i * N;
    graph.edgeWeight = rows.data();
    const int inner = N;
    try {
        GraphProcess(&graph, res.data(),
            [&](int e, int a, struct Graph *g, double *out) {
                const int col = getNeighbors(g, e);
                double s = 0.0;
                for (int k = 0; k < inner; k++) s += g->edgeWeight[e][k] * g->states[k * col + a];
                out[e * col + a] = s;
            },
            [&](int, struct Graph *, double *) {});
    } catch (const std::exception &ex) {
        fprintf(stderr, "%s\n", ex.what());
        return 3;
    }
    fwrite(res.data(), sizeof(double), res.size(), stdout);
    return 0;
}
