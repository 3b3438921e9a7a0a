// g4s/graph.hpp — the vertex-centric graph programming interface of deepmd/source/op/graph.h:5-32, over the C-ABI.
//   struct Graph { numNodes, degree, edgeWeight, states, temp }      graph.h:5-11
//   GraphProcess(graph, result, gather, apply)                       graph.h:21-32
// The reference runs the callbacks on 8 OpenMP threads; a GPU cannot call host code, so GraphProcess takes the descriptor of
// the pattern the callbacks implement (one of the three that exist in the reference, include/g4s.h) and runs that pattern's
// kernel. The callbacks stay in the signature so call sites keep their shape; they are not invoked.
#pragma once
#include <functional>
#include <stdexcept>
#include <string>
#include "../g4s.h"

struct Graph {
    int numNodes;
    int degree;
    const double **edgeWeight;
    const double *states;
    double *temp;
};
inline int getNumNodes(struct Graph *graph) { return graph->numNodes; }
inline int getNeighbors(struct Graph *graph, int) { return graph->degree; }

namespace g4s {
namespace detail {
inline void key_gather(int, int, const double **, const double *, double *) {}
inline void key_apply(int, const double **, const double *, double *) {}
} // namespace detail

inline void GraphProcess(struct Graph *graph, double *result,
                         std::function<void(int, int, struct Graph *, double *)> /*gather*/,
                         std::function<void(int, struct Graph *, double *)> /*apply*/,
                         const g4s_pattern_desc &pattern, double *seconds = nullptr)
{
    // std::function objects have no identity to key a registry on: register the descriptor under a private key pair per call.
    if (g4s_register_pattern(&detail::key_gather, &detail::key_apply, &pattern) != G4S_OK)
        throw std::runtime_error(std::string("GraphProcess: ") + g4s_last_error());
    const g4s_status st = g4s_spmm_dense((uint32_t)graph->numNodes, (uint32_t)graph->degree, graph->edgeWeight, graph->states, graph->temp,
                                         result, &detail::key_gather, &detail::key_apply, seconds, 8);
    g4s_unregister_pattern(&detail::key_gather, &detail::key_apply);
    if (st != G4S_OK) throw std::runtime_error(std::string("GraphProcess: ") + g4s_last_error());
}
} // namespace g4s
