"""oracle/cen2019_np.py -- two further CPU restatements of cen2019 keypoint extraction in numpy.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (the ORORA submodule that holds the upstream features.cpp is an empty directory in /root/reference:
.gitmodules:1-3, README.md:29).  Nothing here can pin the C oracle to the reference; what it gives is a cross-check:

  extract_sequential   written from SURVEY.md Appendix B.2 alone, step by step (sort, greedy marking with a region
                       budget, per-azimuth run extraction), independently of oracle/cen2019_ref.c.  Disagreements with
                       the C oracle are either bugs or unstated choices; the choices it shares with the C oracle are
                       listed in `CHOICES` below.
  extract_parallel     the SORT-FREE reformulation the HIP kernels (csrc/cen2019.hip) implement, so that the
                       equivalence  greedy marking in global intensity order  ==  per-run minima + one selection
                       is checked on the CPU, without a GPU, against both of the above.

Why the reformulation is exact.  Order the pixels by key = (h descending, pixel index ascending) and let
neg(p) = (s(p) < 0).  Visiting a candidate p marks p and the maximal runs of neg pixels adjacent to it on either side
(B.2 step 4).  A pixel with s >= 0 is therefore only ever marked by ITSELF; a maximal neg run N is marked as a whole,
by the first visited pixel among its "touchers" = {the pixel just left of N, the pixel just right of N, the pixels of
N}, i.e. by the toucher with the smallest key, minkey(N).  Hence, with no sequential state at all:
    mark key  MK(p) = key(p) if s(p) >= 0 else minkey(run of p)
    a visited candidate p opens a NEW region  <=>  key(p) == minkey(N) for every neg run N adjacent to / containing p
    (a neg candidate whose run is already marked is skipped; a non-neg one that touches a marked run is "already")
The region budget stops the walk right after the candidate that opens region number max_points: K* = the max_points-th
smallest key among the region-opening candidates (or +inf).  A pixel ends up marked iff MK(p) <= K* and MK(p) is a
candidate (h > mean_h).  Non-candidates can be given keys too: they rank after every candidate, so they never change
a candidate's minimum.  Everything after that (runs, adjacency, arg-max) is per azimuth as in B.2 step 5."""
import numpy as np

CHOICES = (
    "fft = float32(byte) / 255; mean(fft) from the integer byte sum in double; g = |fft(r+1) - fft(r-1)| with "
    "reflect-101 borders; mean(h) from a 2^40 fixed-point sum (order independent); equal h sorted by pixel index; "
    "boundary predicate of the region growth: s < 0 continues (SURVEY B.2 step 4 flags it as suspicious); a run that "
    "reaches the last range bin is never closed and yields nothing; first maximum of h inside a run"
)


def _images(img, col_offset, cols):
    img = np.asarray(img, dtype=np.uint8)
    b = img[:, col_offset:col_offset + cols]
    rows = b.shape[0]
    n = rows * cols
    fft = b.astype(np.float32) / np.float32(255.0)
    if cols > 1:
        rp = np.minimum(np.arange(cols) + 1, cols - 1)
        rp[cols - 1] = cols - 2
        rm = np.maximum(np.arange(cols) - 1, 0)
        rm[0] = 1
        g = np.abs(fft[:, rp] - fft[:, rm]).astype(np.float32)
    else:
        g = np.zeros_like(fft)
    maxg = np.float32(g.max()) if g.size else np.float32(0)
    mean = np.float32(float(b.astype(np.uint64).sum()) / 255.0 / float(n))
    gn = (g / maxg).astype(np.float32) if maxg > 0 else np.zeros_like(g)
    s = (fft - mean).astype(np.float32)
    h = (s * (np.float32(1.0) - gn)).astype(np.float32)
    fix = int(np.rint(h.astype(np.float64) * 1099511627776.0).astype(np.int64).sum())
    mean_h = np.float32(float(fix) / 1099511627776.0 / float(n))
    return s, h, mean_h


def _keypoints_from_marks(mark, h, min_range):
    """B.2 step 5 on a boolean mark image."""
    rows, cols = mark.shape
    out = []
    lo = max(0, int(min_range))
    for a in range(rows):
        m = mark[a]
        nb = mark[(a - 1) % rows] | mark[(a + 1) % rows]
        r = lo
        while r < cols:
            if not m[r]:
                r += 1
                continue
            e = r
            while e + 1 < cols and m[e + 1]:
                e += 1
            if e + 1 < cols and nb[r:e + 1].any():          # closed by an unmarked pixel, neighbour azimuth marked
                out.append((a, r + int(np.argmax(h[a, r:e + 1]))))
            r = e + 1
    return np.asarray(out, dtype=np.int32).reshape(-1, 2)


def extract_sequential(img, col_offset=11, cols=None, max_points=10000, min_range=58):
    """SURVEY B.2, literally."""
    if cols is None:
        cols = img.shape[1] - col_offset
    s, h, mean_h = _images(img, col_offset, cols)
    rows = s.shape[0]
    flat_h = h.reshape(-1)
    cand = np.nonzero(flat_h > mean_h)[0]
    order = cand[np.lexsort((cand, -flat_h[cand].astype(np.float64)))]   # h descending, then index
    mark = np.zeros((rows, cols), dtype=bool)
    neg = s < 0
    regions, visited = 0, 0
    for idx in order:
        if regions >= max_points:
            break
        visited += 1
        a, r = divmod(int(idx), cols)
        if mark[a, r]:
            continue
        lo = r
        while lo - 1 >= 0 and neg[a, lo - 1]:
            lo -= 1
        hi = r
        while hi + 1 < cols and neg[a, hi + 1]:
            hi += 1
        already = mark[a, lo:hi + 1].any()
        mark[a, lo:hi + 1] = True
        if not already:
            regions += 1
    return _keypoints_from_marks(mark, h, min_range), {"ncand": len(cand), "jstar": visited, "mean_h": float(mean_h)}


def _ord32(f):
    u = f.view(np.uint32).astype(np.uint64)
    return np.where(u & 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000)


def mark_keys(s, h):
    """Per pixel: key (h descending, pixel index ascending as one uint64), mark key MK, and whether the pixel, when
    visited, opens a new region."""
    rows, cols = s.shape
    hz = np.where(h == 0, np.float32(0.0), h)                        # -0.0 and +0.0 compare equal in the sort
    key = (((~_ord32(hz)) & 0xFFFFFFFF) << np.uint64(32)) | np.arange(rows * cols, dtype=np.uint64).reshape(rows, cols)
    neg = s < 0
    mk = key.copy()
    opens = np.ones((rows, cols), dtype=bool)
    inf = np.uint64(0xFFFFFFFFFFFFFFFF)
    for a in range(rows):
        n = neg[a]
        if not n.any():
            continue
        d = np.diff(np.concatenate([[0], n.astype(np.int8), [0]]))
        starts, ends = np.nonzero(d == 1)[0], np.nonzero(d == -1)[0] - 1     # maximal neg runs [start, end]
        k = key[a]
        run_min = np.minimum.reduceat(np.where(n, k, inf), starts)            # ends are implied: non-neg pixels are +inf
        left = np.where(starts > 0, k[np.maximum(starts - 1, 0)], inf)
        right = np.where(ends < cols - 1, k[np.minimum(ends + 1, cols - 1)], inf)
        run_min = np.minimum(run_min, np.minimum(left, right))
        run_of = np.cumsum(d[:-1] == 1) - 1                                   # run index of every neg pixel
        mk[a, n] = run_min[run_of[n]]
        op = np.ones(cols, dtype=bool)
        op[n] = k[n] == mk[a, n]
        pos = ~n
        lrun = np.zeros(cols, dtype=bool)
        lrun[1:] = n[:-1]
        rrun = np.zeros(cols, dtype=bool)
        rrun[:-1] = n[1:]
        mk_l = np.concatenate([[inf], mk[a, :-1]])
        mk_r = np.concatenate([mk[a, 1:], [inf]])
        op[pos] = ((~lrun[pos]) | (mk_l[pos] == k[pos])) & ((~rrun[pos]) | (mk_r[pos] == k[pos]))
        opens[a] = op
    return key, mk, opens


def extract_parallel(img, col_offset=11, cols=None, max_points=10000, min_range=58):
    """The sort-free form (module docstring): per-run minima, one selection, per-azimuth extraction."""
    if cols is None:
        cols = img.shape[1] - col_offset
    s, h, mean_h = _images(img, col_offset, cols)
    key, mk, opens = mark_keys(s, h)
    mh = np.float32(0.0) if mean_h == 0 else mean_h
    k_mean = ((~_ord32(np.array([mh], dtype=np.float32))) & 0xFFFFFFFF)[0] << np.uint64(32)   # candidates: key < k_mean
    openers = np.sort(key[opens & (key < k_mean)])
    if max_points <= 0:
        mark = np.zeros(key.shape, dtype=bool)
        kstar = None
    else:
        kstar = openers[max_points - 1] if len(openers) >= max_points else np.uint64(0xFFFFFFFFFFFFFFFF)
        mark = (mk <= kstar) & (mk < k_mean)
    jstar = 0 if kstar is None else int(((key <= kstar) & (key < k_mean)).sum())
    return _keypoints_from_marks(mark, h, min_range), {"ncand": int((key < k_mean).sum()), "jstar": jstar, "mean_h": float(mean_h)}
