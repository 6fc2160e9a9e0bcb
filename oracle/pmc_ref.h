/*
 * oracle/pmc_ref.h -- CPU ORACLE for the max-clique inlier selection that sits between the matcher and the ORORA solver
 * ("PMC max-clique prune", SURVEY.md 3.4 / App. B.3 "TIMs ... over the max-clique inliers", B.5).  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  Upstream ORORA (url-kaist/outlier-robust-radar-odometry: an EMPTY submodule in the reference checkout,
 * /root/reference/.gitmodules:1-3, README.md:19,26-29) takes this stage from TEASER++ (Yang et al., T-RO 2020, "max clique
 * inlier selection"), which calls the PMC library (Rossi et al., "Parallel Maximum Clique Algorithms", 2015).  Neither
 * source is in /root/reference; this file restates the published construction:
 *
 *   consistency graph   vertices = the K matches (src_i -> dst_i); an edge i ~ j (i != j) iff the two matches preserve their
 *                       mutual distance up to the TIM noise bound:  | ||src_i - src_j|| - ||dst_i - dst_j|| | < tau
 *                       (TEASER++: tau = 2 x the point noise bound = rsx_orora_params.tim_noise_bound).
 *                       Evaluated WITHOUT square roots, in fp64 on the float inputs, operation for operation:
 *                           A = dax*dax + day*day      (da = src_j - src_i, exact in fp64)
 *                           B = dbx*dbx + dby*dby
 *                           s = (A + B) - tau*tau
 *                           edge  <=>  s < 0  ||  s*s < 4*(A*B)          [ (sqrt A - sqrt B)^2 < tau^2 ]
 *                       (symmetric in i, j bit for bit; NaN coordinates give no edge).
 *   k-core numbers      core(v) = the largest c such that v lies in a subgraph of minimum degree c (unique; PMC's pruning
 *                       bound: a clique of size s needs s vertices of core >= s - 1, so omega <= max core + 1).
 *   greedy clique       PMC's heuristic with a FIXED tie rule (PMC itself runs it under OpenMP and is not deterministic):
 *                       order = vertices by (core descending, index ascending).  Seeds in that order, at most
 *                       PMCREF_MAX_SEEDS of them; a vertex that belongs to the clique in hand is not a seed (its walk would
 *                       find that clique again: on the bench's data further seeds of that kind never improved anything,
 *                       seeds outside it improve 1 pair in 4 by a few vertices):
 *                           stop when core(seed) + 1 <= |best|  (no later vertex can be in a larger clique), or when
 *                           |best| = max core + 1 (proven maximum);
 *                           P = N(seed) restricted to vertices of core >= |best|;  C = {seed};
 *                           walk `order` from the top: a vertex still in P joins C and P <- P & N(it);
 *                           the seed is abandoned as soon as |C| + |P| <= |best|;
 *                           C replaces best when it is strictly larger.
 *   output              membership flags; the selected matches keep their original order.
 *
 * What is NOT restated: PMC's exact branch and bound.  pmcref_exact_size() below is an independent exact solver (Tomita-style
 * colouring bound) used by the tests and the bench to REPORT how often the greedy clique is the maximum one; the product's
 * info word says when optimality is proven by the core bound.
 */
#ifndef PMC_REF_H
#define PMC_REF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PMCREF_MAX_K 2048   /* matches per pair the stage prunes; larger pairs pass through unpruned (info.flags) */
#ifdef PMCREF_MAX_SEEDS_OVERRIDE
#define PMCREF_MAX_SEEDS PMCREF_MAX_SEEDS_OVERRIDE
#else
#define PMCREF_MAX_SEEDS 2
#endif

typedef struct {
  int32_t size;      /* matches selected */
  int32_t max_core;  /* largest core number of the consistency graph (omega <= max_core + 1) */
  int32_t seeds;     /* greedy seeds started */
  int32_t flags;     /* bit 0: proven maximum (size == max_core + 1); bit 1: passed through unpruned (K < 2 or K > MAX_K) */
} pmcref_info;
#define PMCREF_PROVEN 1
#define PMCREF_PASSTHROUGH 2

/* the K x K edge predicate as a byte matrix (adj[i*k + j] = 1 iff i ~ j): for the tests */
void pmcref_adjacency(const float *src_xy, const float *dst_xy, int32_t k, double tau, uint8_t *adj);
/* core numbers of a byte adjacency matrix */
void pmcref_core_numbers(const uint8_t *adj, int32_t k, int32_t *core);
/* the whole stage for one pair: member[k] = 1 for selected matches */
void pmcref_select(const float *src_xy, const float *dst_xy, int32_t k, double tau, uint8_t *member, pmcref_info *info);
/* batch: pair i owns matches [offsets[i], offsets[i+1]); member is concatenated like the matches */
void pmcref_select_batch(const float *src_xy, const float *dst_xy, const int64_t *offsets, int32_t n_pairs, double tau,
                         uint8_t *member, pmcref_info *info, int nthreads);
/* exact clique number of a byte adjacency matrix by branch and bound (greedy-colouring bound), starting from the lower
 * bound `lb` (a known clique size, 0 if none); -1 when more than max_nodes search nodes were expanded */
int32_t pmcref_exact_size(const uint8_t *adj, int32_t k, int32_t lb, int64_t max_nodes);

#ifdef __cplusplus
}
#endif
#endif
