"""staged bring-up of csrc/pmc.hip: one stage per process (tools/gpu_pmc_debug.sh runs each under a short timeout)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from navtech_radar_slam_amd import _rsx, orora, synth  # noqa: E402

stage = sys.argv[1]
t0 = time.time()


def say(*a):
    print(f"[{stage} +{time.time() - t0:.1f}s]", *a, flush=True)


reg = orora.Orora()
say("handle")
if stage == "plain":
    src, dst, off, truth = synth.orora_pairs(1, 8, k_range=(50, 100))
    r = reg.register_batch(src, dst, off)
    say("plain solver ok", r["status"].tolist())
elif stage.startswith("pmc"):
    n, lo, hi = {"pmc1": (1, 5, 5), "pmc2": (1, 60, 60), "pmc3": (1, 300, 300), "pmc4": (4, 100, 400), "pmc5": (64, 300, 1500), "pmc6": (600, 300, 1500)}[stage]
    src, dst, off, truth = synth.orora_pairs(2, n, k_range=(lo, hi))
    say("data", np.diff(off)[:8])
    m, info = reg.max_clique_batch(src, dst, off)
    say("selection ok", info[:4])
    from oracle import pyoracle as po
    wm, winfo = po.pmc_select_batch(src, dst, off, 1.5, nthreads=8)
    say("identical to oracle:", bool(np.array_equal(m, wm)), bool(np.array_equal(info, winfo)))
    if not np.array_equal(info, winfo):
        bad = np.flatnonzero(info != winfo)[:4]
        say("first differences", [(int(i), info[i], winfo[i]) for i in bad])
elif stage == "solver_pmc":
    src, dst, off, truth = synth.orora_pairs(3, 16, k_range=(100, 600))
    p = orora.default_params()
    p.flags |= _rsx.ORORA_PMC
    r = reg.register_batch(src, dst, off, p)
    say("solver behind the selection ok", r["status"].tolist(), float(np.abs(r["x"] - truth[:, 0]).max()))
    say(reg.last_pmc_info(16)[:4])
