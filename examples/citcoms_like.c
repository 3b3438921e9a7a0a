/* citcoms_like.c — a C translation unit that uses the library exactly the way CitcomS does:
 *   - declares `extern void spmm_dense(...)` with the reference's prototype and binds it to a function pointer
 *     (citcoms/bin/Citcom.c:45-48,93);
 *   - owns host callbacks gather()/apply() with the reference's signatures (citcoms/lib/Element_calculations.c:453-473) that
 *     read process globals — they are never called by the GPU path, only used as the registration key;
 *   - calls E->spmm_dense(nel, ends, elt_k, u, Au, Au, gather, apply, &time, 1) on row-pointer element matrices with the unused
 *     slot 0 (citcoms/lib/Drive_solvers.c:52-55; call site Element_calculations.c:500).
 * It checks the result against a plain C evaluation of the same gather loop (this file's own reference, not the oracle).
 * Build: gcc -std=c99 -O2 -Iinclude examples/citcoms_like.c -Lg4s_amd/lib -lg4s_hip -lm -o citcoms_like */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "g4s.h"

extern void spmm_dense(uint32_t numNodes, uint32_t degree, const double **edgeWeight, const double *vertexStates, double *temp,
                       double *result, fun_gather gather, fun_apply apply, double *time, int threadNum);

struct Env {
    void (*spmm_dense)(uint32_t, uint32_t, const double **, const double *, double *, double *, fun_gather, fun_apply, double *, int);
    int nel, nno, neq;
    int *ien;  /* [nel][8], 0-based */
    int *id;   /* [nno][3] */
};
static struct Env *tempE; /* the callbacks' process global (Element_calculations.c:449) */

static void gather(int e, int a, const double **elt_k, const double *u, double *Au)
{
    const int n = 24;
    const int node = tempE->ien[e * 8 + a];
    for (int i = 0; i < 3; ++i) {
        const int aa = tempE->id[node * 3 + i];
        for (int b = 0; b < 8; ++b) {
            const int ii = (3 * a + i) * n + 3 * b, nb = tempE->ien[e * 8 + b];
            Au[aa] += elt_k[e + 1][ii] * u[tempE->id[nb * 3]] + elt_k[e + 1][ii + 1] * u[tempE->id[nb * 3 + 1]] +
                      elt_k[e + 1][ii + 2] * u[tempE->id[nb * 3 + 2]];
        }
    }
}
static void apply(int e, const double **elt_k, const double *u, double *Au) { (void)e; (void)elt_k; (void)u; (void)Au; }

int main(void)
{
    const int ex = 6, ey = 5, ez = 4, nx = ex + 1, ny = ey + 1, nz = ez + 1;
    struct Env E;
    E.spmm_dense = spmm_dense; /* Citcom.c:93 */
    E.nel = ex * ey * ez; E.nno = nx * ny * nz; E.neq = 3 * E.nno;
    E.ien = malloc(sizeof(int) * E.nel * 8); E.id = malloc(sizeof(int) * E.nno * 3);
    int e = 0;
    for (int k = 0; k < ez; ++k) for (int j = 0; j < ey; ++j) for (int i = 0; i < ex; ++i, ++e) {
        const int n0 = (k * ny + j) * nx + i;
        const int nodes[8] = {n0, n0 + 1, n0 + 1 + nx, n0 + nx, n0 + nx * ny, n0 + 1 + nx * ny, n0 + 1 + nx + nx * ny, n0 + nx + nx * ny};
        for (int a = 0; a < 8; ++a) E.ien[e * 8 + a] = nodes[a];
    }
    for (int n = 0; n < E.nno; ++n) for (int d = 0; d < 3; ++d) E.id[n * 3 + d] = 3 * n + d;
    double **elt_k = malloc(sizeof(double *) * (E.nel + 1)); /* slot 0 unused, as in Drive_solvers.c:52-55 */
    elt_k[0] = NULL;
    unsigned s = 12345u;
    for (e = 1; e <= E.nel; ++e) {
        elt_k[e] = malloc(sizeof(double) * 576);
        for (int q = 0; q < 576; ++q) { s = s * 1664525u + 1013904223u; elt_k[e][q] = (double)(s >> 8) / 16777216.0 - 0.5; }
    }
    double *u = malloc(sizeof(double) * (E.neq + 1)), *Au = calloc(E.neq + 1, sizeof(double)), *ref = calloc(E.neq + 1, sizeof(double));
    for (int q = 0; q <= E.neq; ++q) { s = s * 1664525u + 1013904223u; u[q] = (double)(s >> 8) / 16777216.0 - 0.5; }
    tempE = &E;

    g4s_pattern_desc d = {0};
    d.kind = G4S_PATTERN_ELEMENT_BLOCK_MATVEC;
    d.num_elems = E.nel; d.nodes_per_elem = 8; d.dof = 3; d.ien = E.ien; d.id = E.id; d.nno = E.nno; d.neq = E.neq;
    d.edge_weight_base = 1; d.static_weights = 1;
    if (g4s_register_pattern(gather, apply, &d) != G4S_OK) { fprintf(stderr, "register: %s\n", g4s_last_error()); return 1; }

    double time = -1.0;
    E.spmm_dense((uint32_t)E.nel, 8, (const double **)elt_k, u, Au, Au, gather, apply, &time, 1); /* Element_calculations.c:500 */

    for (e = 0; e < E.nel; ++e) for (int a = 0; a < 8; ++a) gather(e, a, (const double **)elt_k, u, ref); /* the reference's host loop */
    double maxerr = 0.0, maxabs = 0.0;
    for (int q = 0; q < E.neq; ++q) { maxerr = fmax(maxerr, fabs(Au[q] - ref[q])); maxabs = fmax(maxabs, fabs(ref[q])); }
    printf("nel %d neq %d  device seconds %.3g  max|err| %.3e  max|Au| %.3e\n", E.nel, E.neq, time, maxerr, maxabs);
    const int ok = maxerr <= 1e-12 * maxabs && time >= 0.0;
    printf("%s\n", ok ? "CHECK OK" : "CHECK FAILED");
    return ok ? 0 : 1;
}
