"""csrc/sc_plan.h -- the integer logic that cuts a query batch into host-entry pieces, filter batches and the XCD-aware work
split of the two-wave filter kernel -- compiled with the host compiler and checked over a sweep of sizes: every query / every
(tile-block, query tile) unit is covered exactly once, the shapes are the documented ones."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include "sc_plan.h"
using namespace rsx::sc::plan;
int main() {
  std::printf("{\"pieces\": [");
  const int nqs[] = {1, 63, 64, 1000, 4095, 4096, 4097, 5555, 8192, 10000, 65536, 1000000};
  bool first = true;
  for (int nq : nqs) for (int f : {0, 256, 1024}) for (int g : {5, 10, 25, 40}) {
    int32_t sz[8];
    const int n = host_pieces(nq, f, g, 8, sz);
    std::printf("%s[%d, %d, %d, [", first ? "" : ", ", nq, f, g);
    first = false;
    for (int i = 0; i < n; i++) std::printf("%s%d", i ? ", " : "", sz[i]);
    std::printf("]]");
  }
  std::printf("], \"batches\": [");
  first = true;
  for (long long n : {1ll, 970ll, 9970ll, 99970ll, 1000000ll, 40000000ll}) for (long long nq : {1ll, 64ll, 100ll, 8192ll, 100000ll}) {
    std::printf("%s[%lld, %lld, %lld]", first ? "" : ", ", n, nq, (long long)filter_batch(n, nq));
    first = false;
  }
  std::printf("], \"xcd\": [");
  first = true;
  for (long long nq : {64ll, 511ll, 512ll, 1024ll, 3200ll, 4608ll, 8192ll, 8189ll, 100000ll}) for (long long n : {1250ll, 9970ll, 99970ll}) for (int cu : {256, 304, 250}) {
    const long long nqt = (nq + 3) / 4, ntb = ((n + 31) / 32 + 3) / 4;
    const XcdSplit s = xcd_split(nqt, ntb, cu);
    std::printf("%s[%lld, %lld, %d, %d, %d, %d, %u]", first ? "" : ", ", nqt, ntb, cu, s.on ? 1 : 0, s.nqt_x, s.len, s.grid);
    first = false;
  }
  std::printf("]}\n");
  return 0;
}
'''


@pytest.fixture(scope="module")
def plans(tmp_path_factory):
    d = tmp_path_factory.mktemp("plan")
    src = d / "drv.cpp"
    src.write_text(DRIVER)
    exe = d / "drv"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "navtech-radar-slam_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    return json.loads(subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout)


def test_host_pieces_cover_the_batch(plans):
    for nq, first, growth, sizes in plans["pieces"]:
        assert sum(sizes) == nq and all(s > 0 for s in sizes) and len(sizes) <= 8, (nq, first, growth, sizes)
        if first <= 0 or nq < 4 * first:
            assert sizes == [nq]
        else:
            assert all(s % 64 == 0 for s in sizes[:-1]) and sizes[0] == (first + 63) // 64 * 64
            if growth >= 10 and len(sizes) < 8:
                assert all(b >= a for a, b in zip(sizes[:-2], sizes[1:-1]))      # pieces do not shrink (the last takes the rest)
    # the shipped plan on the bench batch
    assert [s for nq, f, g, s in plans["pieces"] if (nq, f, g) == (8192, 1024, 25)] == [[1024, 2560, 4608]]


def test_filter_batches_are_even_and_bounded(plans):
    for n, nq, qb in plans["batches"]:
        ld = (n + 31) // 32 * 32
        assert 1 <= qb <= nq or (nq < 64 and qb == nq)
        if qb < nq:
            assert qb % 64 == 0 and (qb * ld <= 2 ** 29 or qb == 64)             # <= 1 GiB of fp16 bounds (64 queries at least)
            nb = -(-nq // qb)
            assert nq - (nb - 1) * qb > qb // 2 or nb == 1 or nb >= 8, (n, nq, qb)   # no stub of a last batch among a few
    assert [qb for n, nq, qb in plans["batches"] if (n, nq) == (99970, 8192)] == [4096]


def test_xcd_split_covers_every_unit_once(plans):
    used_somewhere = False
    for nqt, ntb, cu, on, nqt_x, ln, grid in plans["xcd"]:
        if not on:
            continue
        used_somewhere = True
        if ntb * nqt > 2_000_000:
            continue                                                              # (the sweep below is pure Python)
        assert cu % 8 == 0 and nqt_x == -(-nqt // 8) and ln >= 1 and grid % 8 == 0
        seen = set()
        for b in range(grid):                                                     # the kernel's own arithmetic (sc_spec.hip)
            x, j = b & 7, b >> 3
            sub, utb = divmod(j, ntb)
            qx1 = min((x + 1) * nqt_x, nqt)
            u0 = x * nqt_x + sub * ln
            u1 = min(u0 + ln, qx1)
            for u in range(u0, max(u0, u1)):
                assert (utb, u) not in seen
                seen.add((utb, u))
        assert len(seen) == ntb * nqt, (nqt, ntb, cu)
    assert used_somewhere
    # the bench launch: 2048 query tiles x 78 tile-blocks on 256 CUs -> two sub-ranges of 128 tiles per XCD
    assert [r[3:] for r in plans["xcd"] if r[:3] == [2048, 78, 256]][0] == [1, 256, 128, 8 * 2 * 78]
    # a 1024-query piece keeps the contiguous split (whole rounds would cost 17 %)
    assert [r[3] for r in plans["xcd"] if r[:3] == [256, 78, 256]] == [0]
