"""ORORA scan pairs per second with and without the max-clique inlier selection (csrc/pmc.hip), the selection alone, and its
agreement with the oracle on the bench's 3 500 pairs."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navtech_radar_slam_amd import _rsx, orora, synth  # noqa: E402

n_pairs = int(os.environ.get("PAIRS", 3500))
src, dst, off, truth = synth.orora_pairs(777, n_pairs)
reg = orora.Orora()
reg.reserve(int(off[-1]))
d_src, d_dst, d_off = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), torch.from_numpy(off).cuda()
d_res = torch.zeros((n_pairs, 5), dtype=torch.float64, device="cuda")
d_mem = torch.zeros(int(off[-1]), dtype=torch.uint8, device="cuda")
d_info = torch.zeros((n_pairs, 4), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
L = _rsx.lib()


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


p_on = orora.default_params()
p_on.flags |= _rsx.ORORA_PMC
t_plain = timed(lambda: reg.register_batch_device(d_src.data_ptr(), d_dst.data_ptr(), d_off.data_ptr(), n_pairs, d_res.data_ptr(), stream=st))
t_pmc = timed(lambda: reg.register_batch_device(d_src.data_ptr(), d_dst.data_ptr(), d_off.data_ptr(), n_pairs, d_res.data_ptr(), p_on, stream=st))
res = d_res.cpu().numpy().view(_rsx.ORORA_RESULT_DTYPE).reshape(n_pairs)
t_sel = timed(lambda: _rsx.check(L.rsx_orora_max_clique_batch_device(reg._h, d_src.data_ptr(), d_dst.data_ptr(), d_off.data_ptr(), n_pairs, None,
                                                                        d_mem.data_ptr(), d_info.data_ptr(), st)))
info = d_info.cpu().numpy()
print(f"pairs {n_pairs}  matches {int(off[-1])}  solver alone {t_plain * 1e3:.3f} ms ({n_pairs / t_plain:.0f} pairs/s)  "
      f"selection + solver {t_pmc * 1e3:.3f} ms ({n_pairs / t_pmc:.0f} pairs/s)  selection alone {t_sel * 1e3:.3f} ms")
print(f"selected fraction {info[:, 0].sum() / off[-1]:.3f}  proven maximum {np.mean(info[:, 3] & 1):.3f}  seeds/pair {info[:, 2].mean():.2f}")
print("error vs truth with the selection: x %.4f m  yaw %.2e rad" % (np.abs(res["x"] - truth[:, 0]).max(), np.abs(res["yaw"] - truth[:, 2]).max()))
if not os.environ.get("NO_ORACLE"):
    from oracle import pyoracle as po
    n_chk = min(n_pairs, 400)
    wm, winfo = po.pmc_select_batch(src[:off[n_chk]], dst[:off[n_chk]], off[:n_chk + 1], 1.5, nthreads=os.cpu_count() or 1)
    mem = d_mem.cpu().numpy()
    print("selection identical to the oracle on the first", n_chk, "pairs:", bool(np.array_equal(mem[:off[n_chk]], wm)),
          bool(np.array_equal(info[:n_chk, 0], winfo["size"])))
