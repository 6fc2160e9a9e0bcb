"""oracle/odometry_chain.py -- the file-based odometry loop on the CPU, chained from the oracle's own stages.  TEST
INFRASTRUCTURE ONLY (tests, smoke and bench.py's checker / cpu_baseline leg).

PARITY UNPINNED: the upstream `odometry.cpp` lives in the reference's ORORA submodule, an empty directory in
/root/reference (.gitmodules:1-3; README.md:26-29,54-60 describe the entry: file-based polar images in, odometry out).
Per scan, as SURVEY.md section 3.4 / Appendix B records the published pipeline:
    cen2019 keypoints (oracle/cen2019_ref.c) -> metres in the sensor frame -> Cartesian image + ORB-style descriptors
    (oracle/frontend_ref.c) -> knnMatch(2) + ratio in both directions, kept when they agree -> ORORA (oracle/orora_ref.c)
    with src = this scan's points, dst = the previous scan's points: p_previous = R(yaw) p_this + (x, y)
and the accumulated pose is the composition of those motions.  Between matcher and solver sits the max-clique inlier
selection ("PMC max-clique prune"; oracle/pmc_ref.c), on by default since round 6 (`pmc=False`: the chain of
rounds 3-5, every cross-checked match to the solver)."""
import numpy as np

from . import pyoracle as po


def compose(p, rel):
    c, s = np.cos(p[2]), np.sin(p[2])
    return np.array([p[0] + c * rel[0] - s * rel[1], p[1] + s * rel[0] + c * rel[1], p[2] + rel[2]])


def run(images, azimuths, resolution=0.0595, col_offset=11, max_points=10000, min_range=58, ratio=0.8, max_keypoints=16384,
        orora_params=None, W=964, cart_res=0.2592, pmc=True):
    """images: (n, rows, row_stride) uint8; azimuths (rows,) or (n, rows).  -> list of per-scan dicts
    {n_keypoints, n_matches, result (ORORA_RESULT_DTYPE record or None for the first scan), xy, pose (accumulated)}."""
    images = np.asarray(images)
    az = np.asarray(azimuths, dtype=np.float32)
    n, rows, stride = images.shape
    fe = po.FrontendRef(rows=rows, cols=stride - col_offset, W=W, cart_res=cart_res)
    out, prev, pose = [], None, np.zeros(3)
    for i in range(n):
        azi = az[i] if az.ndim == 2 else az
        tg = po.cen2019_extract(images[i], col_offset=col_offset, max_points=max_points, min_range=min_range)
        nk = len(tg)
        tg = tg[:max_keypoints]
        xy = po.cen2019_to_cartesian(tg, azi, resolution)
        fe.cartesian(images[i], azi, resolution, col_offset=col_offset)   # scan i through ITS OWN azimuth grid
        desc, valid = fe.describe(xy)
        rec = {"n_keypoints": nk, "n_matches": 0, "result": None, "xy": xy}
        if prev is not None:
            fwd, _, _ = fe.match(prev[1], prev[2], desc, valid, ratio=ratio)
            bwd, _, _ = fe.match(desc, valid, prev[1], prev[2], ratio=ratio)
            ii = np.nonzero(fwd >= 0)[0]
            ii = ii[bwd[fwd[ii]] == ii]
            src, dst = xy[fwd[ii]], prev[0][ii]
            rec["n_matches"] = len(ii)
            if pmc:
                tau = (orora_params or po.orora_default_params()).tim_noise_bound
                member, info = po.pmc_select_batch(src, dst, np.array([0, len(ii)], dtype=np.int64), tau)
                src, dst = src[member.astype(bool)], dst[member.astype(bool)]
                rec["n_selected"], rec["pmc_info"] = len(src), info[0]
            r = po.orora_register_batch(src, dst, np.array([0, len(src)], dtype=np.int64), params=orora_params)[0]
            rec["result"] = r
            if r["status"] == 0:
                pose = compose(pose, (r["x"], r["y"], r["yaw"]))
        rec["pose"] = pose.copy()
        out.append(rec)
        prev = (xy, desc, valid)
    return out
