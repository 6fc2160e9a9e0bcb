/*
 * oracle/pmc_ref.c -- CPU ORACLE for the max-clique inlier selection before the ORORA solver.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED: restates the published construction (see pmc_ref.h); the reference's ORORA submodule, TEASER++ and the
 * PMC library are all absent from /root/reference (.gitmodules:1-3, README.md:19,26-29).
 */
#include "pmc_ref.h"

#include <stdlib.h>
#include <string.h>

/* the edge predicate of pmc_ref.h, operation for operation (compiled with -ffp-contract=off) */
static int edge(const float *s, const float *d, int32_t i, int32_t j, double tau2) {
  const double dax = (double)s[2 * j] - (double)s[2 * i], day = (double)s[2 * j + 1] - (double)s[2 * i + 1];
  const double dbx = (double)d[2 * j] - (double)d[2 * i], dby = (double)d[2 * j + 1] - (double)d[2 * i + 1];
  const double A = dax * dax + day * day;
  const double B = dbx * dbx + dby * dby;
  const double sm = (A + B) - tau2;
  return (sm < 0.0) || (sm * sm < 4.0 * (A * B));
}

void pmcref_adjacency(const float *src_xy, const float *dst_xy, int32_t k, double tau, uint8_t *adj) {
  const double tau2 = tau * tau;
  for (int32_t i = 0; i < k; i++)
    for (int32_t j = 0; j < k; j++) adj[(size_t)i * k + j] = (i != j) && edge(src_xy, dst_xy, i, j, tau2);
}

/* ---- bitset rows: W 64-bit words per row ---- */
typedef uint64_t word;
#define WBITS 64
static inline int tst(const word *r, int32_t v) { return (int)((r[v / WBITS] >> (v % WBITS)) & 1u); }
static inline void set1(word *r, int32_t v) { r[v / WBITS] |= (word)1 << (v % WBITS); }

static void build_rows(const float *s, const float *d, int32_t k, double tau, int32_t W, word *rows, int32_t *deg) {
  const double tau2 = tau * tau;
  memset(rows, 0, sizeof(word) * (size_t)k * W);
  memset(deg, 0, sizeof(int32_t) * (size_t)k);
  for (int32_t i = 0; i < k; i++)
    for (int32_t j = i + 1; j < k; j++) /* the predicate is symmetric bit for bit (squares of negated differences) */
      if (edge(s, d, i, j, tau2)) {
        set1(rows + (size_t)i * W, j);
        set1(rows + (size_t)j * W, i);
        deg[i]++;
        deg[j]++;
      }
}

/* Batagelj-Zaversnik bucket peeling, O(K + E): the core numbers are unique, the order of removal does not matter */
static void cores_from_rows(const word *rows, int32_t k, int32_t W, const int32_t *deg0, int32_t *core) {
  int32_t *deg = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)k), *vert = deg + k, *pos = vert + k;
  int32_t *bin = (int32_t *)calloc((size_t)k + 1, sizeof(int32_t));
  memcpy(deg, deg0, sizeof(int32_t) * (size_t)k);
  for (int32_t v = 0; v < k; v++) bin[deg[v]]++;
  for (int32_t d = 0, start = 0; d <= k; d++) {
    const int32_t n = bin[d];
    bin[d] = start;
    start += n;
  }
  for (int32_t v = 0; v < k; v++) {
    pos[v] = bin[deg[v]]++;
    vert[pos[v]] = v;
  }
  for (int32_t d = k; d > 0; d--) bin[d] = bin[d - 1];
  bin[0] = 0;
  for (int32_t i = 0; i < k; i++) {
    const int32_t v = vert[i];
    core[v] = deg[v];
    const word *r = rows + (size_t)v * W;
    for (int32_t w = 0; w < W; w++) {
      word m = r[w];
      while (m) {
        const int32_t u = w * WBITS + __builtin_ctzll(m);
        m &= m - 1;
        if (deg[u] > deg[v]) { /* move u to the front of its bin, then into the bin below */
          const int32_t du = deg[u], pu = pos[u], pw = bin[du], x = vert[pw];
          if (u != x) {
            pos[u] = pw;
            vert[pu] = x;
            pos[x] = pu;
            vert[pw] = u;
          }
          bin[du]++;
          deg[u]--;
        }
      }
    }
  }
  free(deg);
  free(bin);
}

void pmcref_core_numbers(const uint8_t *adj, int32_t k, int32_t *core) {
  if (k <= 0) return;
  const int32_t W = (k + WBITS - 1) / WBITS;
  word *rows = (word *)calloc((size_t)k * W, sizeof(word));
  int32_t *deg = (int32_t *)calloc((size_t)k, sizeof(int32_t));
  for (int32_t i = 0; i < k; i++)
    for (int32_t j = 0; j < k; j++)
      if (adj[(size_t)i * k + j]) {
        set1(rows + (size_t)i * W, j);
        deg[i]++;
      }
  cores_from_rows(rows, k, W, deg, core);
  free(rows);
  free(deg);
}

static int32_t popcnt_row(const word *r, int32_t W) {
  int32_t n = 0;
  for (int32_t w = 0; w < W; w++) n += __builtin_popcountll(r[w]);
  return n;
}

void pmcref_select(const float *src_xy, const float *dst_xy, int32_t k, double tau, uint8_t *member, pmcref_info *info) {
  pmcref_info inf = {0, 0, 0, 0};
  if (k < 2 || k > PMCREF_MAX_K) { /* nothing to prune with / too large for the stage: every match passes */
    for (int32_t i = 0; i < k; i++) member[i] = 1;
    inf.size = k > 0 ? k : 0;
    inf.flags = PMCREF_PASSTHROUGH;
    if (info) *info = inf;
    return;
  }
  const int32_t W = (k + WBITS - 1) / WBITS;
  word *rows = (word *)malloc(sizeof(word) * (size_t)k * W);
  int32_t *deg = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
  int32_t *core = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
  word *P = (word *)malloc(sizeof(word) * 3 * (size_t)W), *C = P + W, *best = C + W;
  build_rows(src_xy, dst_xy, k, tau, W, rows, deg);
  cores_from_rows(rows, k, W, deg, core);
  int32_t max_core = 0;
  for (int32_t v = 0; v < k; v++)
    if (core[v] > max_core) max_core = core[v];
  /* order: core descending, index ascending (counting sort, stable) */
  {
    int32_t t = 0;
    for (int32_t c = max_core; c >= 0; c--)
      for (int32_t v = 0; v < k; v++)
        if (core[v] == c) order[t++] = v;
  }
  memset(best, 0, sizeof(word) * (size_t)W);
  int32_t best_n = 0, seeds = 0;
  for (int32_t t = 0; t < k && seeds < PMCREF_MAX_SEEDS; t++) {
    const int32_t v = order[t];
    if (core[v] + 1 <= best_n || best_n == max_core + 1) break;
    if (tst(best, v)) continue; /* a member of the clique in hand: its walk would find (a part of) that clique again */
    seeds++;
    const word *rv = rows + (size_t)v * W;
    memset(P, 0, sizeof(word) * (size_t)W);
    memset(C, 0, sizeof(word) * (size_t)W);
    for (int32_t u = 0; u < k; u++)
      if (tst(rv, u) && core[u] >= best_n) set1(P, u);
    set1(C, v);
    int32_t n = 1, np = popcnt_row(P, W);
    int abandoned = n + np <= best_n;
    for (int32_t s = 0; s < k && np > 0 && !abandoned; s++) {
      const int32_t u = order[s];
      if (!tst(P, u)) continue;
      set1(C, u);
      n++;
      const word *ru = rows + (size_t)u * W;
      for (int32_t w = 0; w < W; w++) P[w] &= ru[w];
      np = popcnt_row(P, W);
      if (n + np <= best_n) abandoned = 1;
    }
    if (!abandoned && n > best_n) {
      best_n = n;
      memcpy(best, C, sizeof(word) * (size_t)W);
    }
  }
  for (int32_t i = 0; i < k; i++) member[i] = (uint8_t)tst(best, i);
  inf.size = best_n;
  inf.max_core = max_core;
  inf.seeds = seeds;
  inf.flags = best_n == max_core + 1 ? PMCREF_PROVEN : 0;
  if (info) *info = inf;
  free(rows);
  free(deg);
  free(core);
  free(order);
  free(P);
}

void pmcref_select_batch(const float *src_xy, const float *dst_xy, const int64_t *offsets, int32_t n_pairs, double tau,
                         uint8_t *member, pmcref_info *info, int nthreads) {
#pragma omp parallel for schedule(dynamic, 2) num_threads(nthreads > 0 ? nthreads : 1)
  for (int32_t i = 0; i < n_pairs; i++) {
    const int64_t o = offsets[i];
    pmcref_select(src_xy + 2 * o, dst_xy + 2 * o, (int32_t)(offsets[i + 1] - o), tau, member + o, info ? info + i : NULL);
  }
}

/* ---- independent exact solver: branch and bound with a greedy sequential colouring bound (Tomita & Seki's MCQ) ---- */
typedef struct {
  const uint8_t *adj;
  int32_t k, best;
  int64_t nodes, max_nodes;
} exact_state;

static void expand(exact_state *S, int32_t *R, int32_t nr, int32_t size) {
  if (S->nodes < 0) return;
  if (++S->nodes > S->max_nodes) {
    S->nodes = -1;
    return;
  }
  /* colour R greedily in its given order; visit in non-increasing colour */
  int32_t *col = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(nr > 0 ? nr : 1)), *ord = col + nr;
  {
    int32_t *cls = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nr > 0 ? nr : 1));
    int32_t placed = 0, c = 0;
    uint8_t *done = (uint8_t *)calloc((size_t)(nr > 0 ? nr : 1), 1);
    while (placed < nr) {
      c++;
      int32_t nc = 0;
      for (int32_t i = 0; i < nr; i++) {
        if (done[i]) continue;
        int ok = 1;
        for (int32_t j = 0; j < nc && ok; j++) ok = !S->adj[(size_t)R[i] * S->k + cls[j]];
        if (ok) {
          cls[nc++] = R[i];
          done[i] = 1;
          ord[placed] = R[i];
          col[placed] = c;
          placed++;
        }
      }
    }
    free(cls);
    free(done);
  }
  int32_t *next = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nr > 0 ? nr : 1));
  for (int32_t i = nr - 1; i >= 0 && S->nodes >= 0; i--) {
    if (size + col[i] <= S->best) break;
    const int32_t v = ord[i];
    int32_t nn = 0;
    for (int32_t j = 0; j < i; j++)
      if (S->adj[(size_t)v * S->k + ord[j]]) next[nn++] = ord[j];
    if (nn == 0) {
      if (size + 1 > S->best) S->best = size + 1;
    } else {
      expand(S, next, nn, size + 1);
    }
  }
  free(next);
  free(col);
}

int32_t pmcref_exact_size(const uint8_t *adj, int32_t k, int32_t lb, int64_t max_nodes) {
  if (k <= 0) return 0;
  exact_state S = {adj, k, lb, 0, max_nodes};
  /* initial order: degree descending (index ascending) */
  int32_t *R = (int32_t *)malloc(sizeof(int32_t) * (size_t)k), *deg = (int32_t *)calloc((size_t)k, sizeof(int32_t));
  for (int32_t i = 0; i < k; i++)
    for (int32_t j = 0; j < k; j++) deg[i] += adj[(size_t)i * k + j];
  for (int32_t i = 0; i < k; i++) R[i] = i;
  for (int32_t i = 1; i < k; i++) { /* insertion sort, stable */
    const int32_t v = R[i];
    int32_t j = i;
    while (j > 0 && deg[R[j - 1]] < deg[v]) {
      R[j] = R[j - 1];
      j--;
    }
    R[j] = v;
  }
  expand(&S, R, k, 0);
  free(R);
  free(deg);
  if (S.best < 1) S.best = 1;
  return S.nodes < 0 ? -1 : S.best;
}
