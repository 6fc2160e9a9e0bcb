"""Offline model: how many rows would the window kernel fetch past an XCD's L2 if the workgroups of an XCD were ordered by
their query's first short-list entry?  Takes the short lists of the bench batch (rsx_sc_window_previews), plays the row
accesses of the 8 XCDs through an LRU of `cap` rows each (4 MB L2 / 2432 B ~ 1700 rows, shared with everything else: 1000), in
(a) the launch order of today (workgroup b = query b, XCD b % 8) and (b) queries sorted by first slot, XCD x = a contiguous
eighth.  Usage: python tools/exp_window_locality.py"""
import os
import sys
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lru_misses(seq_per_xcd, cap):
    miss = tot = 0
    for seq in seq_per_xcd:
        c = OrderedDict()
        for r in seq:
            tot += 1
            if r in c:
                c.move_to_end(r)
            else:
                miss += 1
                c[r] = 1
                if len(c) > cap:
                    c.popitem(last=False)
    return miss, tot


def main():
    from navtech_radar_slam_amd import scancontext as sc
    d = np.load("/tmp/ab_host_data.npz")
    db, q = d["db"], d["q"]
    g = sc.SCManager(capacity_hint=len(db) + 8)
    g.add_descriptors_f32(db[:9970])
    slots, pv, ks, sm, cnt = g.window_previews(q, k=10)
    nq = len(q)
    used = [slots[i][(ks[i] != -2) & (slots[i] >= 0)] for i in range(nq)]   # records the kernel computed (pass 1 + admitted pass 2)
    print("records per query", np.mean([len(u) for u in used]))
    # 32 workgroups resident per XCD at a time: interleave their accesses group by group (32 rows per wave step)
    def play(order_per_xcd):
        seqs = []
        for qs in order_per_xcd:
            seq = []
            for w0 in range(0, len(qs), 32):          # a "round" of resident workgroups
                grp = [used[i] for i in qs[w0:w0 + 32]]
                m = max(len(u) for u in grp)
                for p in range(0, m, 32):             # all of them walk their lists 32 entries at a time
                    for u in grp:
                        seq.extend(u[p:p + 32].tolist())
            seqs.append(seq)
        return seqs
    today = [list(range(x, nq, 8)) for x in range(8)]
    first = np.array([u[0] if len(u) else -1 for u in used])
    order = np.argsort(first, kind="stable")
    chunk = -(-nq // 8)
    srt = [order[x * chunk:(x + 1) * chunk].tolist() for x in range(8)]
    for cap in (500, 1000, 1700):
        m0, t0 = lru_misses(play(today), cap)
        m1, t1 = lru_misses(play(srt), cap)
        print(f"L2 rows {cap}: today miss {m0 / t0:.3f}  sorted by first candidate miss {m1 / t1:.3f}")


if __name__ == "__main__":
    main()
