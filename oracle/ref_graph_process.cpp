// ref_graph_process.cpp — C entry points around the REFERENCE's own GraphProcess, compiled in place from
// /root/reference/deepmd/source/op/graph.h (never copied into this repo). Output: oracle/_ref/libref_graph.so.
// TEST INFRASTRUCTURE: used only to pin oracle_spmm_dense / the pattern oracles against the reference's driver loop.
// graph.h omits <omp.h> (its includer provides it in the reference build); the system header is included first.
#include <omp.h>
#include "graph.h" // -I/root/reference/deepmd/source/op

typedef void (*fun_gather)(int, int, const double **, const double *, double *);
typedef void (*fun_apply)(int, const double **, const double *, double *);

// GraphProcess driven by C callbacks of the CitcomS shape (citcoms/lib/global_defs.h:48-49).
// Only race-free gathers may be passed: the reference hard-codes 8 OpenMP threads (graph.h:23).
extern "C" __attribute__((visibility("default")))
void ref_graph_process_cb(int numNodes, int degree, const double **edgeWeight, const double *states, double *temp,
                          double *result, fun_gather gather, fun_apply apply)
{
    Graph graph;
    graph.numNodes = numNodes;
    graph.degree = degree;
    graph.edgeWeight = edgeWeight;
    graph.states = states;
    graph.temp = temp;
    GraphProcess(&graph, result,
        [&](int vi, int nb, struct Graph *g, double *res) { gather(vi, nb, g->edgeWeight, g->states, res); },
        [&](int vi, struct Graph *g, double *res) { apply(vi, g->edgeWeight, g->states, res); });
}

// The OptMatmul use of GraphProcess (deepmd/source/op/opt_matmul.cc:43-61 builds exactly this graph: numNodes=M,
// degree=K, edgeWeight = row pointers into xx, states = w): result[M×K] = xx[M×N]·w[N×K] through the reference driver.
extern "C" __attribute__((visibility("default")))
void ref_graph_process_dense(int M, int N, int K, const double *xx, const double *w, double *result)
{
    Graph graph;
    graph.numNodes = M;
    graph.degree = K;
    graph.states = w;
    graph.temp = nullptr;
    const double **rows = new const double *[M > 0 ? M : 1];
    for (int i = 0; i < M; ++i) rows[i] = xx + (size_t)i * N;
    graph.edgeWeight = rows;
    GraphProcess(&graph, result,
        [&](int e, int a, struct Graph *g, double *res) {
            const int col = getNeighbors(g, e);
            double s = 0.0;
            for (int k = 0; k < N; ++k) s += g->edgeWeight[e][k] * g->states[k * col + a];
            res[e * col + a] = s;
        },
        [&](int, struct Graph *, double *) {});
    delete[] rows;
}
