"""oracle/pmc_np.py -- a second, independent CPU restatement of the max-clique inlier selection before the ORORA solver, in
numpy.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the stage's sources (ORORA's submodule, TEASER++, the PMC library) are absent from /root/reference
(.gitmodules:1-3; README.md:19,26-29).  Written from the definitions in oracle/pmc_ref.h by a DIFFERENT computational route
than oracle/pmc_ref.c, so that agreement between the two is evidence that both follow the stated definitions:
  * the consistency graph as one vectorised K x K evaluation (pmc_ref.c: pair by pair over the upper triangle);
  * core numbers straight from the definition -- for c = 0, 1, 2, ... strip vertices of degree < c until none is left; a
    vertex's core number is the last c whose stripped graph still holds it (pmc_ref.c: Batagelj-Zaversnik buckets);
  * the greedy clique on Python sets (pmc_ref.c: bitsets).
Modelling choices SHARED with pmc_ref.c (agreement does not validate them): `CHOICES`."""
import numpy as np

MAX_K = 2048
MAX_SEEDS = 2
PROVEN, PASSTHROUGH = 1, 2

CHOICES = (
    "edge i ~ j iff | ||src_i - src_j|| - ||dst_i - dst_j|| | < tau with tau = the TIM noise bound (2 x point noise), evaluated "
    "without square roots in fp64 as s = (A + B) - tau^2, edge <=> s < 0 or s^2 < 4 (A B); greedy clique instead of PMC's exact "
    "branch and bound, seeds and candidates in (core descending, index ascending) order, candidates of a seed restricted to "
    "core >= |best|, a seed abandoned when |C| + |P| <= |best|, at most 2 seeds, members of the clique in hand are not seeds; pairs of fewer than 2 or more than 2048 "
    "matches pass through unpruned"
)


def adjacency(src, dst, tau):
    s = np.asarray(src, dtype=np.float32).astype(np.float64)
    d = np.asarray(dst, dtype=np.float32).astype(np.float64)
    da = s[None, :, :] - s[:, None, :]
    db = d[None, :, :] - d[:, None, :]
    A = da[..., 0] * da[..., 0] + da[..., 1] * da[..., 1]
    B = db[..., 0] * db[..., 0] + db[..., 1] * db[..., 1]
    sm = (A + B) - tau * tau
    with np.errstate(invalid="ignore", over="ignore"):
        adj = (sm < 0.0) | (sm * sm < 4.0 * (A * B))
    np.fill_diagonal(adj, False)
    return adj


def core_numbers(adj):
    k = len(adj)
    core = np.zeros(k, dtype=np.int64)
    alive = np.ones(k, dtype=bool)
    c = 0
    while alive.any():
        c += 1
        while True:  # strip everything of degree < c inside what is left
            deg = (adj[:, alive] & alive[:, None]).sum(axis=1)
            drop = alive & (deg < c)
            if not drop.any():
                break
            alive &= ~drop
        core[alive] = c
    return core


def select(src, dst, tau):
    """-> (member bool[k], info dict)"""
    k = len(src)
    if k < 2 or k > MAX_K:
        return np.ones(k, dtype=bool), {"size": k, "max_core": 0, "seeds": 0, "flags": PASSTHROUGH}
    adj = adjacency(src, dst, tau)
    core = core_numbers(adj)
    max_core = int(core.max())
    order = sorted(range(k), key=lambda v: (-int(core[v]), v))
    nbrs = [set(np.flatnonzero(adj[v]).tolist()) for v in range(k)]
    best, seeds = [], 0
    for v in order:
        if seeds >= MAX_SEEDS or core[v] + 1 <= len(best) or len(best) == max_core + 1:
            break
        if v in best:
            continue
        seeds += 1
        P = {u for u in nbrs[v] if core[u] >= len(best)}
        C = [v]
        abandoned = len(C) + len(P) <= len(best)
        for u in order:
            if abandoned or not P:
                break
            if u in P:
                C.append(u)
                P &= nbrs[u]
                abandoned = len(C) + len(P) <= len(best)
        if not abandoned and len(C) > len(best):
            best = C
    member = np.zeros(k, dtype=bool)
    member[best] = True
    return member, {"size": len(best), "max_core": max_core, "seeds": seeds, "flags": PROVEN if len(best) == max_core + 1 else 0}
