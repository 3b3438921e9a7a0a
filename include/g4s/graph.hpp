// g4s/graph.hpp — the vertex-centric graph programming interface of deepmd/source/op/graph.h:5-32, over the C-ABI.
//   struct Graph { numNodes, degree, edgeWeight, states, temp }      graph.h:5-11
//   GraphProcess(graph, result, gather, apply)                       graph.h:21-32
// Two spellings:
//   * GraphProcess(graph, result, gather, apply) — the reference's own, global, four arguments (graph.h:21): a call site such as
//     deepmd/source/op/opt_matmul.cc:51 compiles against this header unchanged. std::function callbacks are host code the device cannot run and
//     carry no identity a pattern could be registered under, so this form gives them the interface's own semantics: the reference's driver loop
//     (for every vertex: gather for each neighbour slot, then apply) on the host, through g4s_spmm_dense's path for unregistered pairs — one thread
//     in ascending vertex order by default, 8 threads (the reference's hard-coded count, graph.h:23) with vertices handed out one at a time when the
//     caller has declared its gathers race-free, or refusal: g4s_set_host_callback_policy (include/g4s.h).
//   * g4s::GraphProcess(graph, result, gather, apply, pattern[, seconds]) — the device form: the descriptor says which of the three patterns the
//     callbacks implement and that pattern's kernel runs; the callbacks stay in the signature so the call keeps its shape, they are not invoked.
//   * g4s::ScopedPattern p(desc); in front of an UNTOUCHED four-argument call (round 5): the descriptor sits in a thread-local slot that the four-argument
//     form consults, so the call line stays byte-identical to the reference's (opt_matmul.cc:51) and still runs the pattern's kernel. Scopes nest; the
//     innermost wins. g4s::ScopedRaceFree is the same kind of scope for the host loop: the four-argument calls inside it run the reference's 8 threads
//     (graph.h:23) — the caller's declaration that ITS gathers do not race, without touching the process-wide policy.
#pragma once
#include <functional>
#include <mutex>
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
// The C-ABI's callbacks are plain function pointers without a user-data argument: the std::function pair of the call in progress sits behind one
// process-wide slot (calls from several threads take turns; the library's worker threads read the same slot).
struct HostCall {
    struct Graph *graph;
    const std::function<void(int, int, struct Graph *, double *)> *gather;
    const std::function<void(int, struct Graph *, double *)> *apply;
};
inline HostCall *&host_call() { static HostCall *p = nullptr; return p; }
inline std::mutex &host_call_mutex() { static std::mutex m; return m; }
inline void tramp_gather(int vi, int nb, const double **, const double *, double *result) { HostCall *c = host_call(); (*c->gather)(vi, nb, c->graph, result); }
inline void tramp_apply(int vi, const double **, const double *, double *result) { HostCall *c = host_call(); if (*c->apply) (*c->apply)(vi, c->graph, result); }
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

// The descriptor of the four-argument GraphProcess calls made by this thread while the object lives (see the header of this file).
class ScopedPattern {
  public:
    explicit ScopedPattern(const g4s_pattern_desc &desc, double *seconds = nullptr) : desc_(desc), seconds_(seconds), prev_(current()) { current() = this; }
    ~ScopedPattern() { current() = prev_; }
    ScopedPattern(const ScopedPattern &) = delete;
    ScopedPattern &operator=(const ScopedPattern &) = delete;
    const g4s_pattern_desc &desc() const { return desc_; }
    double *seconds() const { return seconds_; }
    static ScopedPattern *&current() { static thread_local ScopedPattern *p = nullptr; return p; }
  private:
    g4s_pattern_desc desc_;
    double *seconds_;
    ScopedPattern *prev_;
};
// "The gathers of the four-argument calls in this scope do not race": their host loop runs with the reference's thread count (graph.h:23).
class ScopedRaceFree {
  public:
    ScopedRaceFree() { g4s_set_host_callback_policy_thread(G4S_HOST_CALLBACKS_PARALLEL, &prev_); }
    ~ScopedRaceFree() { g4s_set_host_callback_policy_thread(prev_, nullptr); }
    ScopedRaceFree(const ScopedRaceFree &) = delete;
    ScopedRaceFree &operator=(const ScopedRaceFree &) = delete;
  private:
    int32_t prev_ = -1;
};
} // namespace g4s

// The reference's spelling (deepmd/source/op/graph.h:21-32), global like there.
inline void GraphProcess(struct Graph *graph, double *result, std::function<void(int, int, struct Graph *, double *)> gather,
                         std::function<void(int, struct Graph *, double *)> apply)
{
    if (const g4s::ScopedPattern *sp = g4s::ScopedPattern::current()) {   // a scope in front of this call names its pattern: the device form
        g4s::GraphProcess(graph, result, gather, apply, sp->desc(), sp->seconds());
        return;
    }
    std::lock_guard<std::mutex> turn(g4s::detail::host_call_mutex());
    g4s::detail::HostCall call{graph, &gather, &apply};
    g4s::detail::host_call() = &call;
    const g4s_status st = g4s_spmm_dense((uint32_t)getNumNodes(graph), (uint32_t)graph->degree, graph->edgeWeight, graph->states, graph->temp, result,
                                         &g4s::detail::tramp_gather, &g4s::detail::tramp_apply, nullptr, 8 /* graph.h:23 */);
    g4s::detail::host_call() = nullptr;
    if (st != G4S_OK) throw std::runtime_error(std::string("GraphProcess: ") + g4s_last_error());
}
