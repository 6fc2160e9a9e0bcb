"""Synthetic MulRan-shape inputs for tests and bench.py.

No MulRan data exists offline, so every workload is generated: radar feature clouds in the
sensor frame (what /orora/cloud_local carries), keyframe sequences with planted revisits,
matched point pairs for ORORA, and 400 x 3360 polar power images.  Pure numpy; seeds fixed by
the callers (1234 DB / 4321 queries / 777 ORORA).
"""
import numpy as np

NUM_RING, NUM_SECTOR, MAX_RADIUS = 20, 60, 80.0


def radar_cloud(rng, n_points=None, binary_z=True, guard=0.0):
    """One feature cloud as (n, 4) float32 x,y,z,intensity (pcl::PointXYZI payload).

    ranges ~ mixture(uniform[2,80] background, wall/cluster structures); a few points beyond
    80 m (ignored by the descriptor, Scancontext.cpp:175).  binary_z: z = 0 (2-D radar
    features => every occupied bin = LIDAR_HEIGHT); else z ~ U(-1, 4).  guard > 0 pushes
    points away from ring/sector bin edges by that fraction of a bin (parity runs that must not
    depend on the last ulp of atan/sqrt).
    """
    if n_points is None:
        n_points = int(rng.integers(500, 4001))
    n_bg = n_points // 3
    n_st = n_points - n_bg
    r = rng.uniform(2.0, 88.0, n_bg)
    th = rng.uniform(0.0, 2 * np.pi, n_bg)
    # structures: a handful of wall segments (line pieces) and blobs
    n_seg = int(rng.integers(4, 12))
    seg_id = rng.integers(0, n_seg, n_st)
    c_r = rng.uniform(5.0, 75.0, n_seg)
    c_t = rng.uniform(0.0, 2 * np.pi, n_seg)
    cx, cy = c_r * np.cos(c_t), c_r * np.sin(c_t)
    ang = rng.uniform(0.0, np.pi, n_seg)
    half = rng.uniform(1.0, 15.0, n_seg)
    t = rng.uniform(-1.0, 1.0, n_st) * half[seg_id]
    sx = cx[seg_id] + t * np.cos(ang[seg_id]) + rng.normal(0, 0.15, n_st)
    sy = cy[seg_id] + t * np.sin(ang[seg_id]) + rng.normal(0, 0.15, n_st)
    x = np.concatenate([r * np.cos(th), sx])
    y = np.concatenate([r * np.sin(th), sy])
    if guard > 0.0:
        x, y = _apply_guard(x, y, guard)
    if binary_z:
        z = np.zeros_like(x)
    else:
        z = rng.uniform(-1.0, 4.0, x.shape[0])
    inten = rng.uniform(0.0, 1.0, x.shape[0])
    return np.stack([x, y, z, inten], axis=1).astype(np.float32)


def _apply_guard(x, y, guard):
    """Move points so that range/angle sit at least `guard` bins away from any bin edge."""
    r = np.hypot(x, y)
    th = np.mod(np.arctan2(y, x), 2 * np.pi)
    rb = r / (MAX_RADIUS / NUM_RING)
    fr = rb - np.floor(rb)
    rb = np.floor(rb) + np.clip(fr, guard, 1 - guard)
    tb = th / (2 * np.pi / NUM_SECTOR)
    ft = tb - np.floor(tb)
    tb = np.floor(tb) + np.clip(ft, guard, 1 - guard)
    r = rb * (MAX_RADIUS / NUM_RING)
    th = tb * (2 * np.pi / NUM_SECTOR)
    return r * np.cos(th), r * np.sin(th)


def revisit(rng, cloud, yaw, jitter=0.05, drop=0.1, guard=0.0):
    """A later observation of the same place: rotate by yaw, jitter, drop/add some points."""
    keep = rng.uniform(size=cloud.shape[0]) > drop
    c = cloud[keep].astype(np.float64)
    cs, sn = np.cos(yaw), np.sin(yaw)
    x = cs * c[:, 0] - sn * c[:, 1] + rng.normal(0, jitter, c.shape[0])
    y = sn * c[:, 0] + cs * c[:, 1] + rng.normal(0, jitter, c.shape[0])
    if guard > 0.0:
        x, y = _apply_guard(x, y, guard)
    out = np.stack([x, y, c[:, 2], c[:, 3]], axis=1)
    n_new = max(1, int(0.05 * cloud.shape[0]))
    extra = radar_cloud(rng, n_new, binary_z=bool(np.all(cloud[:, 2] == 0)), guard=guard)
    return np.concatenate([out, extra], axis=0).astype(np.float32)


def keyframe_clouds(seed, n, binary_z=True, loop_frac=0.05, guard=0.0, min_gap=50,
                    n_points=None):
    """n keyframe clouds; ~loop_frac of them are planted revisits of an earlier keyframe.

    Returns (clouds, truth) with truth[i] = (earlier index, yaw sector shift) or None.
    """
    rng = np.random.default_rng(seed)
    clouds, truth = [], []
    for i in range(n):
        if i > min_gap and rng.uniform() < loop_frac:
            j = int(rng.integers(0, i - min_gap))
            while truth[j] is not None:  # revisit originals only
                j = int(rng.integers(0, i - min_gap))
            ks = int(rng.integers(0, NUM_SECTOR))
            yaw = ks * (2 * np.pi / NUM_SECTOR)
            clouds.append(revisit(rng, clouds[j], yaw, guard=guard))
            truth.append((j, ks))
        else:
            clouds.append(radar_cloud(rng, n_points, binary_z=binary_z, guard=guard))
            truth.append(None)
    return clouds, truth


def random_descriptors(seed, n, binary=True, fill=0.25):
    """Fast descriptor-level generator for very large DBs (n x 1200 float32, sector-major
    [s*20+r]); values are fp32 by construction.  Used where pushing 10^5 clouds through the
    build path would dominate a test's run time."""
    rng = np.random.default_rng(seed)
    occ = rng.uniform(size=(n, NUM_SECTOR * NUM_RING)) < fill
    # empty sectors happen in real scans: blank a random arc in some entries
    arc = rng.integers(0, NUM_SECTOR, n)
    width = rng.integers(0, 8, n)
    cols = np.arange(NUM_SECTOR)[None, :]
    blank = ((cols - arc[:, None]) % NUM_SECTOR) < width[:, None]
    occ &= ~np.repeat(blank, NUM_RING, axis=1)
    if binary:
        d = np.where(occ, np.float32(2.0), np.float32(0.0))
    else:
        d = np.where(occ, rng.uniform(1.0, 6.0, occ.shape).astype(np.float32), np.float32(0.0))
    return np.ascontiguousarray(d, dtype=np.float32)


def rotate_descriptor(desc_f32, k):
    """Column-rotate right by k sectors: new[(s+k)%60] = old[s] (Scancontext.cpp:39-59)."""
    d = desc_f32.reshape(NUM_SECTOR, NUM_RING)
    return np.roll(d, k, axis=0).reshape(-1).copy()


# ---------------------------------------------------------------------------------------------
# ORORA: matched feature pairs between consecutive scans (seed 777)
# ---------------------------------------------------------------------------------------------
def orora_pairs(seed, n_pairs, k_range=(300, 1500), outlier_range=(0.2, 0.6), max_range=150.0):
    """n_pairs scan pairs.  Each: K matches, a fraction of them outliers, inliers perturbed by
    anisotropic polar noise (radial sigma 0.06 m, tangential sigma = range * 0.2 deg).

    Returns src (M,2) f32, dst (M,2) f32, offsets (n_pairs+1,) i64, truth (n_pairs,3) f64 with
    dst = R(yaw) src + (x, y).
    """
    rng = np.random.default_rng(seed)
    ks = rng.integers(k_range[0], k_range[1] + 1, n_pairs)
    offsets = np.zeros(n_pairs + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(ks)
    m = int(offsets[-1])
    src = np.empty((m, 2), dtype=np.float32)
    dst = np.empty((m, 2), dtype=np.float32)
    truth = np.empty((n_pairs, 3), dtype=np.float64)
    for i in range(n_pairs):
        k = int(ks[i])
        r = rng.uniform(4.0, max_range, k)
        th = rng.uniform(0.0, 2 * np.pi, k)
        s = np.stack([r * np.cos(th), r * np.sin(th)], axis=1)
        yaw = rng.uniform(-0.2, 0.2)
        t = rng.uniform(-2.5, 2.5, 2)
        c, sn = np.cos(yaw), np.sin(yaw)
        d = s @ np.array([[c, sn], [-sn, c]]) + t
        # anisotropic noise in the polar frame of the destination point
        rd = np.hypot(d[:, 0], d[:, 1])
        ud = d / rd[:, None]
        td = np.stack([-ud[:, 1], ud[:, 0]], axis=1)
        d = d + ud * rng.normal(0, 0.06, k)[:, None] + td * (rd * np.deg2rad(0.2) * rng.normal(0, 1.0, k))[:, None]
        n_out = int(rng.uniform(*outlier_range) * k)
        out_idx = rng.choice(k, n_out, replace=False)
        ro = rng.uniform(4.0, max_range, n_out)
        to = rng.uniform(0.0, 2 * np.pi, n_out)
        d[out_idx] = np.stack([ro * np.cos(to), ro * np.sin(to)], axis=1)
        src[offsets[i]:offsets[i + 1]] = s
        dst[offsets[i]:offsets[i + 1]] = d
        truth[i] = (t[0], t[1], yaw)
    return src, dst, offsets, truth


# ---------------------------------------------------------------------------------------------
# polar radar images in MulRan "polar_oxford_form" (SURVEY.md B.1): one row per azimuth,
# bytes 0-7 int64 timestamp, 8-9 uint16 encoder count (azimuth = count * 2 pi / 5600), 10 valid
# flag, 11.. power bytes (3360 bins for the Navtech CIR204-H, ~0.0595 m per bin)
# ---------------------------------------------------------------------------------------------
OXFORD_META = 11
RADAR_RESOLUTION = 0.0595


def polar_image(seed, rows=400, cols=3360, n_targets=1200, shift_rows=0, t0=1_560_000_000_000_000_000,
                noise_seed=None):
    """Synthetic scan: speckle noise floor that decays with range + bright extended targets.
    shift_rows rolls the scene in azimuth (a pure sensor rotation); noise_seed draws a different
    speckle realisation over the same targets (a later scan from the same pose).  Returns (img uint8
    [rows, 11+cols], azimuths float32 [rows] in rad, target centres (n,2) as (a, r))."""
    rng = np.random.default_rng(seed)
    r_idx = np.arange(cols, dtype=np.float32)[None, :]
    floor = 18.0 + 30.0 * np.exp(-r_idx / 900.0)
    nrng = rng if noise_seed is None else np.random.default_rng(noise_seed)
    noise = nrng.gamma(2.0, floor / 2.0, size=(rows, cols)).astype(np.float32)
    if noise_seed is not None:
        rng.gamma(2.0, floor / 2.0, size=(rows, cols))  # keep the target stream aligned with noise_seed=None
    power = noise
    ta = rng.integers(0, rows, n_targets)
    tr = rng.integers(70, cols - 40, n_targets)
    amp = rng.uniform(90.0, 220.0, n_targets)
    wa = rng.integers(1, 4, n_targets)
    wr = rng.integers(2, 9, n_targets)
    for a0, r0, am, da, dr in zip(ta, tr, amp, wa, wr):
        aa = (np.arange(-da, da + 1) + a0) % rows
        rr = np.arange(max(0, r0 - dr), min(cols, r0 + dr + 1))
        prof = np.exp(-0.5 * ((rr - r0) / (0.5 * dr + 0.5)) ** 2)[None, :] * \
            np.exp(-0.5 * (np.arange(-da, da + 1) / (0.5 * da + 0.5)) ** 2)[:, None]
        power[np.ix_(aa, rr)] += am * prof
    power = np.roll(power, shift_rows, axis=0)
    img = np.zeros((rows, OXFORD_META + cols), dtype=np.uint8)
    img[:, OXFORD_META:] = np.clip(power, 0, 255).astype(np.uint8)
    counts = (np.arange(rows) * (5600 // rows)).astype(np.uint16)
    ts = (t0 + np.arange(rows, dtype=np.int64) * 625_000).astype("<i8")
    img[:, 0:8] = ts.view(np.uint8).reshape(rows, 8)
    img[:, 8:10] = counts.astype("<u2").view(np.uint8).reshape(rows, 2)
    img[:, 10] = 255
    az = (counts.astype(np.float64) * 2 * np.pi / 5600.0).astype(np.float32)
    centres = np.stack([(ta + shift_rows) % rows, tr], axis=1)
    return img, az, centres


# ---------------------------------------------------------------------------------------------
# Trajectory-shaped keyframe sequences (SURVEY.md 8d config 2 / "value distributions"): a vehicle
# driving a street grid through one fixed world of wall segments and poles, one radar feature cloud per
# keyframe.  Consecutive keyframes overlap almost completely, streets are driven several times in both
# directions (real revisits), and the descriptors come out of the descriptor-BUILD path (the caller
# pushes these clouds through rsx_sc_add_points), not out of a descriptor-level generator.
# ---------------------------------------------------------------------------------------------
class World:
    """Fixed landmark map: wall segments along the streets of a `blocks` x `blocks` grid (block edge
    `block_m`), poles and clutter.  Every landmark has a fixed height z in [-1, 4) (used by the
    continuous-z family; radar features themselves are z = 0)."""

    def __init__(self, seed, blocks=14, block_m=110.0, density=0.09):
        from scipy.spatial import cKDTree
        rng = np.random.default_rng(seed)
        self.blocks, self.block_m = blocks, block_m
        size = blocks * block_m
        pts = []
        # walls: both sides of every street, set back 7-16 m, pieces of 8-45 m with gaps
        for axis in (0, 1):
            for line in range(blocks + 1):
                for side in (-1.0, 1.0):
                    t = rng.uniform(0.0, 20.0)
                    while t < size:
                        ln = rng.uniform(8.0, 45.0)
                        back = rng.uniform(7.0, 16.0)
                        n = max(2, int(ln / rng.uniform(0.5, 1.4)))
                        u = t + np.sort(rng.uniform(0.0, ln, n))
                        v = line * block_m + side * back + rng.normal(0.0, 0.12, n)
                        pts.append(np.stack([u, v] if axis == 0 else [v, u], axis=1))
                        t += ln + rng.uniform(3.0, 30.0)
        walls = np.concatenate(pts)
        n_clutter = int(density * size * size) - len(walls)
        clutter = rng.uniform(-60.0, size + 60.0, (max(n_clutter, 0), 2))
        self.xy = np.concatenate([walls, clutter])
        self.z = rng.uniform(-1.0, 4.0, len(self.xy))
        self.tree = cKDTree(self.xy)

    def drive(self, rng, n, step_m=2.0):
        """n poses (x, y, heading) of a random drive along the street grid, `step_m` apart."""
        b, L = self.blocks, self.block_m
        node = np.array([int(rng.integers(0, b + 1)), int(rng.integers(0, b + 1))])
        dirs = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]])
        d = int(rng.integers(0, 4))
        poses = []
        while len(poses) < n:
            # at an intersection: mostly straight on, sometimes turn, (almost) never back
            options = [(d, 0.55), ((d + 1) % 4, 0.21), ((d + 3) % 4, 0.21), ((d + 2) % 4, 0.03)]
            options = [(o, w) for o, w in options if np.all(node + dirs[o] >= 0) and np.all(node + dirs[o] <= b)]
            w = np.array([w for _, w in options])
            d = options[int(rng.choice(len(options), p=w / w.sum()))][0]
            lane = rng.normal(0.0, 0.4)                     # lateral offset of this pass
            k = int(round(L / step_m))
            s = (np.arange(k) + rng.uniform(0.0, 1.0)) * step_m
            base = node * L
            xy = base[None, :] + s[:, None] * dirs[d][None, :] + lane * np.array([-dirs[d][1], dirs[d][0]])[None, :]
            hd = np.arctan2(dirs[d][1], dirs[d][0]) + rng.normal(0.0, 0.02, k)
            poses.append(np.concatenate([xy, hd[:, None]], axis=1))
            node = node + dirs[d]
        return np.concatenate(poses)[:n]

    def observe(self, rng, poses, binary_z=True, p_detect=0.7, sigma=0.08, clutter_frac=0.05, r_max=84.0):
        """Feature clouds in the sensor frame of each pose -> (points (M,4) f32, offsets (n+1,) i64)."""
        hits = self.tree.query_ball_point(poses[:, :2], r_max, return_sorted=False)
        out, off = [], [0]
        for p, h in zip(poses, hits):
            h = np.asarray(h, dtype=np.int64)
            h = h[rng.uniform(size=len(h)) < p_detect]
            dxy = self.xy[h] - p[None, :2] + rng.normal(0.0, sigma, (len(h), 2))
            c, s = np.cos(-p[2]), np.sin(-p[2])
            x = c * dxy[:, 0] - s * dxy[:, 1]
            y = s * dxy[:, 0] + c * dxy[:, 1]
            z = np.zeros(len(h)) if binary_z else self.z[h]
            nc = int(clutter_frac * len(h)) + 1
            rr, tt = rng.uniform(2.0, 88.0, nc), rng.uniform(0.0, 2 * np.pi, nc)
            x = np.concatenate([x, rr * np.cos(tt)])
            y = np.concatenate([y, rr * np.sin(tt)])
            z = np.concatenate([z, np.zeros(nc) if binary_z else rng.uniform(-1.0, 4.0, nc)])
            out.append(np.stack([x, y, z, np.ones_like(x)], axis=1).astype(np.float32))
            off.append(off[-1] + len(x))
        return np.concatenate(out), np.asarray(off, dtype=np.int64)


def trajectory_keyframes(seed_db, n_db, seed_q, n_q, binary_z=True, revisit_frac=0.5, min_gap=64):
    """A drive of n_db keyframes + n_q query scans.  revisit_frac of the queries are new observations of a
    place the drive passed (near keyframe src >= 0, arbitrary heading, up to 1 m off the driven line);
    the rest are places in the same world the drive never saw (src = -1, no loop expected).
    -> db_points, db_offsets, q_points, q_offsets, q_src (int64[n_q])."""
    world = World(seed_db)
    rng = np.random.default_rng(seed_db + 1)
    poses = world.drive(rng, n_db)
    db_pts, db_off = world.observe(rng, poses, binary_z=binary_z)
    rq = np.random.default_rng(seed_q)
    src = np.full(n_q, -1, dtype=np.int64)
    n_rev = int(revisit_frac * n_q)
    src[:n_rev] = rq.integers(0, max(1, n_db - min_gap), n_rev)
    rq.shuffle(src)
    qposes = np.empty((n_q, 3))
    rev = src >= 0
    qposes[rev, :2] = poses[src[rev], :2] + rq.normal(0.0, 0.5, (int(rev.sum()), 2))
    size = world.blocks * world.block_m
    # novel places: mid-block positions (the drive stays on the streets)
    nb = int((~rev).sum())
    cell = rq.integers(0, world.blocks, (nb, 2))
    qposes[~rev, :2] = (cell + rq.uniform(0.3, 0.7, (nb, 2))) * world.block_m
    qposes[:, 2] = rq.uniform(0.0, 2 * np.pi, n_q)
    q_pts, q_off = world.observe(rq, qposes, binary_z=binary_z)
    del size
    return db_pts, db_off, q_pts, q_off, src


# ---------------------------------------------------------------------------------------------
# A MOVING sensor: one fixed world of point reflectors seen from a sequence of known poses
# (SURVEY 8d config 1 / BASELINE configs[0] and [2]: scan pairs with KNOWN motion, end to end).
# Frame convention = the one the hot path uses (cen2019 step 5, rsx.h rsx_frontend_describe): a keypoint
# at azimuth phi and range rho sits at (rho cos phi, rho sin phi) in the sensor frame.  A pose (x, y, yaw)
# places the sensor in the world: p_world = R(yaw) p_sensor + (x, y).
# ---------------------------------------------------------------------------------------------
def relative_pose(p_from, p_to):
    """Pose of sensor `p_to` expressed in the frame of sensor `p_from` (both (x, y, yaw) in the world):
    p_from_frame = R(yaw_rel) p_to_frame + (x_rel, y_rel) -- what ORORA estimates with src = the scan taken
    at p_to and dst = the scan taken at p_from."""
    c, s = np.cos(p_from[2]), np.sin(p_from[2])
    dx, dy = p_to[0] - p_from[0], p_to[1] - p_from[1]
    yaw = (p_to[2] - p_from[2] + np.pi) % (2 * np.pi) - np.pi
    return np.array([c * dx + s * dy, -s * dx + c * dy, yaw])


def compose_pose(p, rel):
    """p o rel: the world pose of a sensor whose pose in the frame of `p` is `rel`."""
    c, s = np.cos(p[2]), np.sin(p[2])
    return np.array([p[0] + c * rel[0] - s * rel[1], p[1] + s * rel[0] + c * rel[1], p[2] + rel[2]])


def polar_sequence(seed, n_scans, rows=400, cols=3360, n_buildings=260, n_poles=500, world_radius=215.0,
                   speed=(0.4, 2.6), yaw_rate=0.06, t0=1_560_000_000_000_000_000, period_ns=250_000_000):
    """n_scans MulRan-shape polar images of ONE world seen from a moving sensor.

    World: rectangular buildings (walls = chains of reflectors ~0.9 m apart at irregular spacing, each with its own fixed strength, so a
    wall has a texture that survives from scan to scan) and isolated poles, none within 6 m of the driven line.
    The sensor drives a smooth random curve (per scan: forward step within `speed` m with a slow random walk, a
    lateral slip of a few cm, yaw increment ~N(0, yaw_rate) rad with memory).  Every reflector within radar range is
    drawn as a Gaussian blob at its CONTINUOUS (azimuth row, range bin) position -- 1.8 deg beam (sigma ~0.85 rows),
    1.2-3 range bins -- over an independent speckle floor per scan, so keypoints are unbiased observations of the
    reflectors up to the polar quantisation.  No occlusion.  -> (images [n, rows, 11+cols] u8, azimuths f32[rows],
    poses f64[n, 3] world poses with poses[0] = 0, stamps int64[n])."""
    rng = np.random.default_rng(seed)
    poses = np.zeros((n_scans, 3))
    v = rng.uniform(*speed)
    w = 0.0
    for i in range(1, n_scans):
        v = float(np.clip(v + rng.normal(0.0, 0.25), speed[0], speed[1]))
        w = 0.7 * w + rng.normal(0.0, yaw_rate)
        poses[i] = compose_pose(poses[i - 1], (v, rng.normal(0.0, 0.03), w))
    centre = poses[:, :2].mean(axis=0)
    rad = world_radius + np.abs(poses[:, :2] - centre).max()

    def disc(n):
        rr, tt = rad * np.sqrt(rng.uniform(0.0, 1.0, n)), rng.uniform(0.0, 2 * np.pi, n)
        return centre[None, :] + np.stack([rr * np.cos(tt), rr * np.sin(tt)], axis=1)

    pts = [disc(n_poles)]
    amps = [rng.uniform(150.0, 240.0, n_poles)]
    for c0 in disc(n_buildings):
        wx, wy, th = rng.uniform(8.0, 40.0), rng.uniform(8.0, 40.0), rng.uniform(0.0, np.pi)
        per = []
        for (ax, ay, bx, by) in ((-wx, -wy, wx, -wy), (wx, -wy, wx, wy), (wx, wy, -wx, wy), (-wx, wy, -wx, -wy)):
            n = max(2, int(np.hypot(bx - ax, by - ay) / 2 / 0.9))
            t = np.sort(rng.uniform(0.0, 1.0, n))      # irregular spacing: a periodic wall lets matches slide along it
            per.append(np.stack([ax + t * (bx - ax), ay + t * (by - ay)], axis=1) / 2)
        per = np.concatenate(per) + rng.normal(0.0, 0.05, (sum(len(q) for q in per), 2))
        cs, sn = np.cos(th), np.sin(th)
        pts.append(c0[None, :] + per @ np.array([[cs, sn], [-sn, cs]]))
        amps.append(rng.uniform(60.0, 200.0, len(per)))
    world = np.concatenate(pts)
    amp = np.concatenate(amps)
    dmin = np.min(np.hypot(world[:, None, 0] - poses[None, :, 0], world[:, None, 1] - poses[None, :, 1]), axis=1)
    keep = dmin > 6.0
    world, amp = world[keep], amp[keep]
    sig_r = rng.uniform(1.2, 3.0, len(world))
    sig_a = rng.uniform(0.75, 1.0, len(world))
    r_idx = np.arange(cols, dtype=np.float32)[None, :]
    floor = 18.0 + 30.0 * np.exp(-r_idx / 900.0)
    counts = (np.arange(rows) * (5600 // rows)).astype(np.uint16)
    az = (counts.astype(np.float64) * 2 * np.pi / 5600.0).astype(np.float32)
    az_step = 2 * np.pi / rows
    images = np.zeros((n_scans, rows, OXFORD_META + cols), dtype=np.uint8)
    stamps = t0 + np.arange(n_scans, dtype=np.int64) * period_ns
    da, dr = np.arange(-3, 4), np.arange(-9, 10)
    for i in range(n_scans):
        power = rng.gamma(2.0, floor / 2.0, size=(rows, cols)).astype(np.float64)
        c, s = np.cos(poses[i, 2]), np.sin(poses[i, 2])
        d = world - poses[i, None, :2]
        px, py = c * d[:, 0] + s * d[:, 1], -s * d[:, 0] + c * d[:, 1]
        rho = np.hypot(px, py)
        phi = np.mod(np.arctan2(py, px), 2 * np.pi)
        rc = rho / RADAR_RESOLUTION - 0.5            # bin r covers [r res, (r+1) res): centre (r + 0.5) res
        ac = phi / az_step
        vis = (rc > 62.0) & (rc < cols - 12.0)
        rc, ac, am, sr, sa = rc[vis], ac[vis], amp[vis], sig_r[vis], sig_a[vis]
        a0, r0 = np.round(ac).astype(np.int64), np.round(rc).astype(np.int64)
        aa = a0[:, None] + da[None, :]                                    # [K, 7]
        rb = r0[:, None] + dr[None, :]                                    # [K, 19]
        pa = np.exp(-0.5 * ((aa - ac[:, None]) / sa[:, None]) ** 2)
        pr = np.exp(-0.5 * ((rb - rc[:, None]) / sr[:, None]) ** 2) * am[:, None]
        flat = ((aa % rows)[:, :, None] * cols + rb[:, None, :]).reshape(-1)
        power += np.bincount(flat, weights=(pa[:, :, None] * pr[:, None, :]).reshape(-1), minlength=rows * cols).reshape(rows, cols)
        images[i, :, OXFORD_META:] = np.clip(power, 0, 255).astype(np.uint8)
        ts = (stamps[i] + np.arange(rows, dtype=np.int64) * 625_000).astype("<i8")
        images[i, :, 0:8] = ts.view(np.uint8).reshape(rows, 8)
        images[i, :, 8:10] = counts.astype("<u2").view(np.uint8).reshape(rows, 2)
        images[i, :, 10] = 255
    return images, az, poses, stamps
