"""oracle/orora_np.py -- a second, independent CPU restatement of ORORA's solver (GNC-TLS rotation + A-COTE translation) in
numpy.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the solver's sources are the reference's empty ORORA submodule (.gitmodules:1-3; README.md:19,26-27).
This file is written from SURVEY.md Appendix B.3 / B.4 (the published algorithm: ORORA, Lim et al. ICRA 2023; TEASER++'s
GNC-TLS and adaptive-voting estimators) by a DIFFERENT computational route than oracle/orora_ref.c, so that agreement
between the two is evidence that both follow the spec rather than each other:
  * rotation by an explicit 2 x 2 SVD of the weighted cross-covariance, R = V diag(1, det(V U^T)) U^T (B.3), not the
    atan2 closed form;
  * the scalar TLS estimator by explicit enumeration: for every consensus set of the sorted interval endpoints the
    weighted mean and the cost are recomputed from the set's members, not from running sums.
Modelling choices that the absent upstream source would pin and that are SHARED with orora_ref.c (listed so that a reader
knows agreement does not validate them): `CHOICES`."""
import numpy as np

CHOICES = (
    "TIMs on the ring of K consecutive matches (flag: complete graph); first-iteration mu = 1 / (2 max r^2 / c^2 - 1) and stop "
    "if mu <= 0; weights updated AFTER the cost of the iteration is taken with the old weights; mu *= 1.4; stop when "
    "|cost - previous cost| < 1e-6; A-COTE bound of a match = sum over its two points (dst, R src) of the radial bound and "
    "range x tangential bound projected on the axis (|cos|, |sin|); scalar TLS cost normalised per point: sum_C w (x - x^)^2 + "
    "#outliers with w = beta^-2 (flag: TEASER++'s mixed-unit form); ties between equal endpoint values: lower endpoints "
    "(+id) after upper endpoints (-id) of smaller index, i.e. ordered by (value, signed id)"
)


def _rotation_svd(a, b, w):
    """argmin_R sum w ||b - R a||^2 by SVD of H = sum w a b^T (B.3)."""
    H = (a * w[:, None]).T @ b
    U, _, Vt = np.linalg.svd(H)
    V = Vt.T
    d = np.linalg.det(V @ U.T)
    return V @ np.diag([1.0, d]) @ U.T


def gnc_rotation(src, dst, c, gnc_factor=1.4, cost_threshold=1e-6, max_iterations=100, complete=False):
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    k = len(src)
    if complete:
        i, j = np.triu_indices(k, 1)
    else:
        i, j = np.arange(k), (np.arange(k) + 1) % k
    a, b = src[j] - src[i], dst[j] - dst[i]
    w = np.ones(len(a))
    c2 = c * c if c * c >= 1e-16 else 1e-2
    mu, prev, R = 1.0, np.inf, np.eye(2)
    it = 0
    while it < max_iterations:
        H = (a * w[:, None]).T @ b
        R = _rotation_svd(a, b, w) if np.linalg.norm(H) > 0 else np.eye(2)
        r2 = ((b - a @ R.T) ** 2).sum(axis=1)
        if it == 0:
            mu = 1.0 / (2.0 * r2.max() / c2 - 1.0)
            if mu <= 0:
                it = 1
                break
        th1, th2 = (mu + 1.0) / mu * c2, mu / (mu + 1.0) * c2
        cost = float((w * r2).sum())
        with np.errstate(divide="ignore", invalid="ignore"):
            mid = np.sqrt(c2 * mu * (mu + 1.0) / r2) - mu
        w = np.where(r2 >= th1, 0.0, np.where(r2 <= th2, 1.0, mid))
        diff = abs(cost - prev)
        mu *= gnc_factor
        prev = cost
        it += 1
        if diff < cost_threshold:
            break
    return R, it, int((w >= 0.5).sum())


def scalar_tls(x, beta, teaser_cost=False):
    """Consensus maximisation by interval stabbing, every consensus set evaluated from scratch (B.4)."""
    x, beta = np.asarray(x, dtype=np.float64), np.asarray(beta, dtype=np.float64)
    n = len(x)
    ends = [(x[i] - beta[i], i + 1) for i in range(n)] + [(x[i] + beta[i], -(i + 1)) for i in range(n)]
    ends.sort()
    inside = np.zeros(n, dtype=bool)
    best = (np.inf, 0.0)
    w_all = 1.0 / (beta * beta)
    for v, sid in ends:
        inside[abs(sid) - 1] = sid > 0
        if not inside.any():
            continue
        w, xs = w_all[inside], x[inside]
        xh = float((w * xs).sum() / w.sum())
        if teaser_cost:
            cost = float(((xs - xh) ** 2).sum() + beta[~inside].sum())
        else:
            cost = float((w * (xs - xh) ** 2).sum() + (n - inside.sum()))
        if cost < best[0]:
            best = (cost, xh)
    return best[1]


def _bounds(p, s_r, s_t):
    rho = np.hypot(p[:, 0], p[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        c = np.where(rho > 0, np.abs(p[:, 0]) / rho, 1.0)
        s = np.where(rho > 0, np.abs(p[:, 1]) / rho, 0.0)
    return c * s_r + s * rho * s_t, s * s_r + c * rho * s_t


def register(src, dst, tim_noise_bound=1.5, noise_bound_radial=0.3536, noise_bound_tangential=1.8 * np.pi / 180.0, gnc_factor=1.4,
             cost_threshold=1e-6, max_iterations=100, complete=False, teaser_cost=False):
    """-> dict(x, y, yaw, iterations, rot_inliers, trans_inliers) with dst ~= R(yaw) src + (x, y)."""
    src32, dst32 = np.asarray(src, dtype=np.float32), np.asarray(dst, dtype=np.float32)
    src, dst = src32.astype(np.float64), dst32.astype(np.float64)
    R, it, rot_in = gnc_rotation(src, dst, tim_noise_bound, gnc_factor, cost_threshold, max_iterations, complete)
    rs = src @ R.T
    v = dst - rs
    bx1, by1 = _bounds(dst, noise_bound_radial, noise_bound_tangential)
    bx2, by2 = _bounds(rs, noise_bound_radial, noise_bound_tangential)
    bx, by = bx1 + bx2, by1 + by2
    tx, ty = scalar_tls(v[:, 0], bx, teaser_cost), scalar_tls(v[:, 1], by, teaser_cost)
    inl = int(((np.abs(v[:, 0] - tx) <= bx) & (np.abs(v[:, 1] - ty) <= by)).sum())
    return {"x": tx, "y": ty, "yaw": float(np.arctan2(R[1, 0], R[0, 0])), "iterations": it, "rot_inliers": rot_in, "trans_inliers": inl}
