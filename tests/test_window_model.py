"""CPU check of the arithmetic behind the window kernel (csrc/sc_window.hip): a numpy emulation of its two GEMMs -- fp16
operands, products exact in fp32, fp32 accumulation -- against the oracle's pair function, on the data families the GPU tests
use.  What it pins (without a GPU):
  * the alignment rule: with the sector keys scaled by a power of two and split into fp16 hi + lo, the correlation KC[k] is
    within kWinAlignEps * sqrt(E_q E_e) of the real value, so "exactly one shift within 2 eps of the maximum" implies that shift
    is fastAlignUsingVkey's answer (SC.cpp:93-113) -- and the two cases the kernel refuses to judge (keys >= 4e6, key norms
    more than 1e6 apart) are exactly where the reference's fp64 search stops following the real argmin;
  * the preview margin: 1 - S_k / n_eff(k) from fp16 unit columns is within WINDOW_MARGIN of the oracle's d_k;
  * the union-of-windows preview is a lower bound of the pair distance whatever shift the reference picks among the admissible
    ones.
The kernel itself is compared pair by pair with the oracle in tests/test_gpu_sc_window.py."""
import numpy as np
import pytest

from navtech_radar_slam_amd import synth

EPS_ALIGN = 3e-5      # sc_window.hip kWinAlignEps
MARGIN = 1.25e-3      # rsx.h RSX_SC_WINDOW_MARGIN
NS, NR = 60, 20


def split_key(v):
    """sc_window.hip split_key: scale so that max |x| is in [2^9, 2^10), hi = fp16(x), lo = fp16(x - hi)"""
    mx = np.abs(v).max()
    e = 0 if mx == 0 else 10 - int(np.frexp(mx)[1])
    x = np.ldexp(v, e)
    hi = x.astype(np.float32).astype(np.float16)
    lo = (x - hi.astype(np.float64)).astype(np.float32).astype(np.float16)
    return hi, lo, float(np.sqrt((x * x).sum()) * (1 + 1e-6)), float(np.sqrt((v * v).sum()))


def kc_fp16(vq, ve):
    """KC[k] = sum_j q[(j + k) % 60] e[j] the way the kernel accumulates it: hi*hi + hi*lo + lo*hi, fp32 accumulator"""
    qh, ql, nq, uq = split_key(vq)
    eh, el, ne, ue = split_key(ve)
    out = np.zeros(NS, dtype=np.float32)
    for k in range(NS):
        acc = np.float32(0.0)
        for a, b in ((qh, eh), (qh, el), (ql, eh)):
            prod = np.roll(a, -k).astype(np.float32) * b.astype(np.float32)     # exact: 11-bit x 11-bit significands
            for s in range(0, NS, 16):                                          # one MFMA per 16 k: fp32 adds
                acc = np.float32(acc + prod[s:s + 16].sum(dtype=np.float32))
        out[k] = acc
    return out, nq, ne, uq, ue


def admissible(vq, ve):
    """the shifts the kernel keeps (None: it refuses to judge the pair)"""
    kc, nq, ne, uq, ue = kc_fp16(vq, ve)
    if not (np.isfinite(kc).all() and uq < 4e6 and ue < 4e6 and min(uq, ue) > 0 and min(uq, ue) >= 1e-6 * max(uq, ue)):
        return None
    return np.nonzero(kc >= kc.max() - 2 * EPS_ALIGN * nq * ne)[0]


def key_families(rng):
    yield "binary", (rng.integers(0, 21, NS) * 0.1), (rng.integers(0, 21, NS) * 0.1)
    yield "continuous", rng.uniform(0, 3, NS), rng.uniform(0, 3, NS)
    v = rng.uniform(0, 3, NS)
    yield "rotated copy", v, np.roll(v, int(rng.integers(0, NS)))
    yield "six decades", rng.uniform(0, 1, NS) * 10.0 ** rng.uniform(-3, 3, NS), rng.uniform(0, 1, NS) * 10.0 ** rng.uniform(-3, 3, NS)
    yield "mixed signs", rng.normal(size=NS), rng.normal(size=NS)
    yield "tiny vs normal", rng.uniform(0, 1, NS) * 1e-30, rng.uniform(0, 1, NS)
    yield "huge", rng.uniform(0, 1, NS) * 1e30, rng.uniform(0, 1, NS) * 1e30
    yield "sparse", np.where(rng.uniform(size=NS) < 0.1, 2.0, 0.0), np.where(rng.uniform(size=NS) < 0.1, 2.0, 0.0)


def test_key_correlation_error_is_inside_the_alignment_budget():
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(60):
        for name, vq, ve in key_families(rng):
            kc, nq, ne, uq, ue = kc_fp16(vq, ve)
            # the same quantity in fp64, in the kernel's scaled units
            eq = 0 if np.abs(vq).max() == 0 else 10 - int(np.frexp(np.abs(vq).max())[1])
            ee = 0 if np.abs(ve).max() == 0 else 10 - int(np.frexp(np.abs(ve).max())[1])
            xq, xe = np.ldexp(vq, eq), np.ldexp(ve, ee)
            exact = np.array([(np.roll(xq, -k) * xe).sum() for k in range(NS)])
            if nq * ne > 0:
                worst = max(worst, np.abs(kc - exact).max() / (nq * ne))
    assert worst < EPS_ALIGN / 4, worst      # the budget is a rigorous worst case; fp32 sums of 192 terms sit far below it


def test_unique_admissible_shift_is_the_reference_alignment(oracle):
    rng = np.random.default_rng(1)
    judged = unique = 0
    for _ in range(150):
        for name, vq, ve in key_families(rng):
            adm = admissible(vq, ve)
            want = oracle.fast_align(vq.astype(np.float64), ve.astype(np.float64))
            if adm is None:
                continue
            judged += 1
            assert want in adm, (name, want, adm)              # the reference's choice is always among the admissible shifts
            if len(adm) == 1:
                unique += 1
                assert adm[0] == want, (name, adm, want)
    assert judged > 600 and unique > 0.5 * judged, (judged, unique)


def test_refused_pairs_are_where_the_reference_leaves_the_real_argmin(oracle):
    """keys of 1e30: every distance is >= the 1e7 the reference's search starts from, it answers 0 (SC.cpp:100-106);
    a 1e-30 key against an O(1) key: all 60 distances are the same double, it answers the first one"""
    rng = np.random.default_rng(2)
    vq, ve = rng.uniform(0.5, 1, NS) * 1e30, rng.uniform(0.5, 1, NS) * 1e30
    assert admissible(vq, ve) is None and oracle.fast_align(vq, ve) == 0
    vq, ve = rng.uniform(0.5, 1, NS), np.roll(rng.uniform(0.5, 1, NS), 7) * 1e-30
    assert admissible(vq, ve) is None and oracle.fast_align(vq, ve) == 0


def unit_columns_fp16(d):
    """sc_filter.hip normalise_column: x / ||column|| * 2^15 -> fp16 (an empty column stays 0); d is [60][20]"""
    n = np.sqrt((d.astype(np.float64) ** 2).sum(axis=1))
    u = np.where(n[:, None] > 0, d / np.where(n == 0, 1, n)[:, None], 0.0) * 32768.0
    return u.astype(np.float32).astype(np.float16), n > 0


@pytest.mark.parametrize("binary", [True, False])
def test_preview_margin_and_union_lower_bound(oracle, binary):
    rng = np.random.default_rng(3 + binary)
    descs = synth.random_descriptors(40 + binary, 48, binary=binary)
    if not binary:
        descs[:8] = (descs[:8].reshape(8, NS, NR) * 10.0 ** rng.uniform(-3, 3, (8, NS, NR))).reshape(8, 1200).astype(np.float32)
    descs[9][20 * 5:20 * 9] = 0
    worst = 0.0
    nkept = nmask = 0
    for _ in range(120):
        i, j = rng.integers(0, len(descs), 2)
        q = descs[i] if rng.uniform() < 0.7 else synth.rotate_descriptor(descs[j], int(rng.integers(0, NS)))
        e = descs[j]
        qh, qm = unit_columns_fp16(q.reshape(NS, NR))
        eh, em = unit_columns_fp16(e.reshape(NS, NR))
        q64, e64 = q.astype(np.float64), e.astype(np.float64)
        pv = np.full(NS, np.inf)
        for k in range(NS):   # S_k = sum_j <q column (j + k) % 60, e column j>, fp32 accumulation of exact products
            ne = int((np.roll(qm, -k) & em).sum())
            if ne:
                S = (np.roll(qh, -k, axis=0).astype(np.float32) * eh.astype(np.float32)).sum(dtype=np.float32)
                pv[k] = 1.0 - float(S) / 1073741824.0 / ne
        dist, shift = oracle.distance(q64, e64)
        ks = oracle.fast_align(oracle.sectorkey(q64), oracle.sectorkey(e64))
        win = [(ks + o) % NS for o in range(-3, 4)]
        if dist >= 1e7:
            assert np.isinf(pv[win]).all()
            continue
        worst = max(worst, abs(pv[win].min() - dist))
        # the shift mask (sc_window.hip): a window shift whose preview is more than two margins above the best one is
        # strictly worse than the minimum, so the exact evaluation may leave it out -- the reference's final shift never is
        kept = [k for k in win if pv[k] <= pv[win].min() + 2 * MARGIN]
        assert shift in kept, (shift, kept)
        for k in win:
            if k not in kept and np.isfinite(pv[k]):
                assert oracle.dist_direct(oracle.circshift(e64, k), q64) > dist
        nkept += len(kept)
        nmask += 1
        # whatever set of alignments the kernel keeps, as long as it contains the reference's: the union preview is a lower bound
        others = rng.integers(0, NS, 3).tolist() + [ks]
        union = sorted({(a + o) % NS for a in others for o in range(-3, 4)})
        assert pv[union].min() - MARGIN <= dist
    assert worst <= MARGIN / 2, worst
    assert nmask > 50 and nkept < 4 * nmask   # the mask does cut: well under 4 of 7 shifts survive on average
