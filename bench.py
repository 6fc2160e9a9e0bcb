#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ScanContext + ORORA hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
With --gpus N > 1 and no torch.distributed environment the script starts its own N ranks (one per
GPU, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`);
launched under torch.distributed.run it uses the ranks it is given.  It fails if fewer than N GPUs
are visible and reports the number of ranks that actually took part (`n_gpus`, `rccl_ranks`).

Metric (BASELINE.json): ScanContext loop-queries/sec against an N-scan keyframe DB.  A "step" is one
batch of Q = 8192 exhaustive queries (distanceBtnScanContext against EVERY eligible DB entry, top-10)
against a 10 000-keyframe DB resident in HBM -- the configuration the north-star target is quoted on
(>= 10k queries/s vs a 10k-scan DB on one MI355X).

Data ("value distributions"): the DB is a synthetic DRIVE -- 10 000 keyframes 2 m
apart on the street grid of one fixed world of walls and clutter, every feature cloud pushed through the
descriptor-BUILD path (rsx_sc_add_points) -- so consecutive keyframes overlap, streets are revisited in
both directions and the descriptors are as dense and as similar to each other as a real sequence's.
Queries: 8192 new scans, half of them revisits of a driven place (arbitrary heading), half places the
drive never saw.  Nothing is planted at descriptor level; correctness of what was timed is checked
against the CPU oracle on >= 256 queries of the timed batch (`oracle_checked_queries`).

Besides `value` (that workload through the default path: fp16 MFMA lower-bound filter -> exact fp64
re-scoring of the survivors) the line carries
  data_dependence   the same batch shape on (a) the round-1 descriptor-level random DB with planted rotated
                    copies, (b) the drive with continuous landmark heights (non-binary descriptors, oracle-checked) and
                    (c) the exact-all path (filter off: every pair scored in fp64) = the data-independent floor, each with
                    queries/s and exact evaluations per query
  host_entry        the same batch through the synchronous host-buffer call (PCIe upload of the queries and download of
                    the records inside the time)
  layouts           (N > 1) every query-groups x DB-shards layout of this world, same DB and batch, identical results;
  layout_emulation  (N = 1) what one rank computes per step in every layout of 2 / 4 / 8 GPUs, emulated on this GPU
  scale_100k        8192 queries vs a 100 000-keyframe DB (the size where DB shards pay), same sharding
  roofline          the dominant kernel (spectral MFMA filter): algorithmic flops / hipEvent time, vs the dense fp16 peak
  cpu_baseline      the CPU oracle (== the reference's Scancontext.cpp, tests/test_oracle_pin.py) on this box's cores
  latency_q1_n1k_us BASELINE configs[1]; orora / cen2019 / icp / frontend: the other parts of the path
  odometry_e2e      BASELINE configs[0] and [2]: the file-based odometry entry end to end on a moving sensor with known poses
                    (scans/s resident / from host memory / from PNG files, PNG decode separately, checked against the oracle chain)
With --gpus G the DB is sharded block-cyclically over G ranks (same DB, same batch => strong scaling); a
query batch is two stages with one RCCL all-gather of per-rank top-k lists each (sharded.py).

RSX_BENCH_LOCAL_BACKEND=module:function (tests only) replaces the GPU shard by a CPU stand-in and RCCL by
gloo so that the launcher / timing / reduction logic can be driven at world 2 without GPUs
(tests/test_bench_launcher.py); such a run is marked "dry_run": true and measures nothing.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak (~2.5 PF)
ALG_BYTES_PER_PAIR = 4800  # one 20x60 fp32 descriptor per (query, DB entry) pair
# the filter's algorithmic work: the 60-shift circular cross-correlation of two column-normalised
# 20x60 images = 60 shifts x 1200 multiply-adds per (query, DB entry) pair
ALG_FLOP_PER_PAIR = 2 * 60 * 1200
# the spectral form of the filter computes the same 60 values with a Z15 DFT: per pair
# stage 1 (8 frequencies x 4 Z4-shifts x K=80 complex, Hermitian half) 9280 MAC + stage 2 (4 x 15 x 16) 960 MAC
# + the exact n_eff correlation of the column masks (60 x 60) 3600 MAC
SPEC_FLOP_PER_PAIR = 2 * (9280 + 960 + 3600)  # (the mask correlation is skipped for queries without empty columns)


# ----------------------------------------------------------------------------------------------
# launcher
# ----------------------------------------------------------------------------------------------

def usable_cores():
    """Host cores this process may actually use: the visible CPUs, cut by the affinity mask and by the cgroup's CPU quota (the GPU
    boxes of this pool show 256 CPUs under a quota of 16: 256 threads there are throttled, not 16 times faster than 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, dry_run):
    """Re-execute this script as n ranks under torch.distributed.run (one process per GPU)."""
    if not dry_run:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit(f"bench.py: --gpus {n} but only {have} GPU(s) visible; refusing to measure {have} GPU(s) {n} times")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------
def random_db_and_queries(n_db, n_q, seed_db=1234, seed_q=4321):
    """Round-1 workload: descriptor-level generator (iid binary descriptors with blank arcs), queries =
    rotated, slightly corrupted copies of DB entries (planted loops with a known answer)."""
    from navtech_radar_slam_amd import synth
    descs = synth.random_descriptors(seed_db, n_db, binary=True)
    rng = np.random.default_rng(seed_q)
    src = rng.integers(0, max(1, n_db - 64), n_q)
    rot = rng.integers(0, 60, n_q)
    q = np.stack([synth.rotate_descriptor(descs[s], int(r)) for s, r in zip(src, rot)])
    drop = rng.integers(0, 1200, (n_q, 24))
    np.put_along_axis(q, drop, 0.0, axis=1)
    return descs, q, src, rot


class Workload:
    """One DB (sharded over the ranks) + one resident query batch + the timed step."""

    def __init__(self, ctx, name, k, capacity, filter_mode=0, query_groups=1, filter_shards=False):
        from navtech_radar_slam_amd import sharded
        self.ctx, self.name, self.k = ctx, name, k
        if filter_shards:  # replicated DB, the filter cut over the ranks by slot range (one all-to-all of bound rows)
            self.ssc = sharded.FilterShardedScanContext(device=ctx.local_rank, capacity_hint=capacity + 8, filter_mode=filter_mode)
        elif ctx.stub:
            self.ssc = sharded.ShardedScanContext(local_backend=lambda sr, sw: ctx.stub(sr, sw), query_groups=query_groups)
        else:
            self.ssc = sharded.ShardedScanContext(device=ctx.local_rank, capacity_hint=capacity * query_groups // ctx.world + 8,
                                                  filter_mode=filter_mode, query_groups=query_groups)
        self.mgr = self.ssc.backend
        self.hits = None

    def add_clouds(self, pts, off):
        for i in range(len(off) - 1):
            self.mgr.makeAndSaveScancontextAndKeys(pts[off[i]:off[i + 1]])

    def add_descriptors(self, descs):
        if self.ctx.stub:
            self.ssc.add_descriptors_f32(descs)
            return
        import torch
        d = torch.from_numpy(descs).cuda()
        self.ssc.add_descriptors_device(d.data_ptr(), len(descs), stream=self.ctx.stream)
        torch.cuda.synchronize()

    def set_queries(self, q_f32, n_elig):
        self.nq, self.n_elig = len(q_f32), n_elig
        self.q_host = q_f32
        if not self.ctx.stub:
            import torch
            self.d_q = torch.from_numpy(q_f32).cuda()

    def step(self):
        if self.ctx.stub:
            self.hits = self.ssc.query(self.q_host, k=self.k, n_eligible=self.n_elig)
        else:
            self.hits = self.ssc.query_device(self.d_q.data_ptr(), self.nq, self.k, n_eligible=self.n_elig, stream=self.ctx.stream)

    def results(self):
        from navtech_radar_slam_amd import scancontext
        if self.ctx.stub:
            return np.ascontiguousarray(self.hits)
        return self.hits.cpu().numpy().view(scancontext.HIT_DTYPE).reshape(self.nq, self.k)

    def local_pairs(self, n_elig):
        """(query, entry) pairs ONE launch of this rank's filter scores: its slice of the batch x its shard's eligible entries"""
        lo, hi, _ = self.ssc._slice(self.nq)
        return (hi - lo) * len(range(self.ssc.shard_rank, n_elig, self.ssc.shard_world))

    def timed(self, steps, warmup, profile=False, settle_steps=0):
        """W untimed steps, then exactly `steps` steps between barrier + synchronize on both sides.
        -> (max-over-ranks seconds, per-rank seconds, (launches, kernel ms), (exact window evaluations, candidates
        that went through alignment + preview)).  settle_steps: that many untimed steps BEFORE the W warm-up steps (a fixed
        number: every rank must run the same sequence of collectives) -- the headline leg follows 18 000 tiny build-path
        kernels, and three 3-ms steps do not always bring the clocks up (one full run of round 3 timed its first leg at 4.1 ms
        per step and every later leg of the same shape at 3.1)"""
        ctx = self.ctx
        for _ in range(0 if ctx.stub else settle_steps):
            self.step()
        for _ in range(warmup):
            self.step()
        self.ssc.time_exchanges = ctx.world > 1 and not ctx.stub   # two events per collective (sharded._ExchangeTimer)
        self.ssc.exchange_ms()
        ctx.barrier()
        if profile and not ctx.stub:
            self.mgr.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        ctx.barrier()
        dt = time.perf_counter() - t0
        self.exchange_ms_per_step = self.ssc.exchange_ms() / max(1, steps)   # this rank: inside collectives, waiting for peers included
        self.ssc.time_exchanges = False
        prof, resc = (0, 0.0), (0, 0)
        if profile and not ctx.stub:
            prof = self.mgr.profile_read()
            self.resc3 = self.mgr.profile_read_rescoring()
            resc = (self.resc3[1], self.resc3[0])
            self.mgr.profile_enable(False)
        per_rank = ctx.all_gather_float(dt)
        return max(per_rank), per_rank, prof, resc

    def close(self):
        self.ssc.close()  # the shard and the sub-communicators of its layout


class Ctx:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        spec = os.environ.get("RSX_BENCH_LOCAL_BACKEND")
        self.stub = None
        if spec:
            mod, fn = spec.split(":")
            self.stub = getattr(importlib.import_module(mod), fn)
        if args.gpus != self.world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        self.stream = 0
        if not self.stub:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a GPU (librsx has no CPU fallback)")
            if torch.cuda.device_count() <= self.local_rank:
                raise SystemExit(f"bench.py: rank {self.rank} has no GPU {self.local_rank} ({torch.cuda.device_count()} visible)")
            torch.cuda.set_device(self.local_rank)
        if self.world > 1 or args.force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            if self.stub:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local_rank))
        if not self.stub:
            # an explicit (non-null) torch stream: the C-ABI launches on it, torch.distributed collectives and
            # torch.cuda.Event see the same stream
            self.tstream = torch.cuda.Stream()
            torch.cuda.set_stream(self.tstream)
            self.stream = self.tstream.cuda_stream

    @property
    def distributed(self):
        return self.dist.is_initialized()

    def barrier(self):
        if not self.stub:
            self.torch.cuda.synchronize()
        if self.distributed:
            self.dist.barrier()
        if not self.stub:
            self.torch.cuda.synchronize()

    def all_gather_float(self, x):
        if not self.distributed:
            return [float(x)]
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.stub else "cuda")
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def shutdown(self):
        if self.distributed:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# CPU baseline + oracle check (rank 0, after timing)
# ----------------------------------------------------------------------------------------------
def cpu_baseline_and_check(db_pts, db_off, q_descs, n_elig, k, gpu_hits, min_checked=256):
    """The CPU baseline beside the GPU number, and the checker of what was timed.
    (1) `kind: "reference"`: the reference's OWN pair function -- oracle/_ref/libref_sc_sse2.so = Scancontext.cpp:116-148
        compiled unmodified (oracle/ref_sc.cpp), its heap-allocating circshift / col() temporaries and all -- over every
        eligible entry for a bounded sample of the timed batch, OpenMP over (query, entry block) on every host core.
    (2) the oracle port (oracle/sc_ref.c, bit-identical to (1): tests/test_oracle_pin.py), which is ~10x faster per pair
        because it does not allocate, is timed the same way (`port_value`) and its exhaustive top-k records are the
        checker: the first `nq` queries of the timed batch, GPU records vs oracle records."""
    from oracle import pyoracle as po
    cores = usable_cores()
    m = po.Manager()
    for i in range(len(db_off) - 1):          # the oracle builds its own descriptors from the same clouds
        m.add_points(db_pts[db_off[i]:db_off[i + 1]])
    n = len(m)
    t0 = time.perf_counter()
    m.exhaustive(q_descs[0].astype(np.float64), n_eligible=n_elig, k=k, nthreads=1)
    t1 = time.perf_counter() - t0
    one_thread_qps = 1.0 / t1
    # ~8 s of CPU work per leg spread over all cores, at least min_checked queries for the check
    nq = int(max(min_checked, min(4 * cores, (8.0 * cores) / max(t1, 1e-6))))
    nq = min(nq, len(q_descs))
    qs = q_descs[:nq].astype(np.float64)
    t0 = time.perf_counter()
    want = m.exhaustive_batch(qs, n_eligible=n_elig, k=k, nthreads=cores)
    dt = time.perf_counter() - t0
    got = gpu_hits[:nq]
    same = [bool(np.array_equal(got[i], want[i])) for i in range(nq)]
    top1_same = int(np.sum((got["index"][:, 0] == want["index"][:, 0]) & (got["shift"][:, 0] == want["shift"][:, 0])))
    port = {"value": nq / dt, "queries": nq, "one_thread_value": one_thread_qps}
    base = None
    try:
        ref = po.RefSC()
        db = np.stack([m.descriptor(i) for i in range(min(n, n_elig))]).astype(np.float64)
        t0 = time.perf_counter()
        ref.distances_batch(qs[:1], db, nthreads=1)
        r1 = time.perf_counter() - t0
        nr = int(max(2, min(nq, (8.0 * cores) / max(r1, 1e-6))))
        t0 = time.perf_counter()
        rd, rs = ref.distances_batch(qs[:nr], db, nthreads=cores)
        rdt = time.perf_counter() - t0
        # the reference's top-1 of every sampled query = the oracle's (the whole distance rows are bit-identical: test_oracle_pin)
        hit = rd < 1e7
        ref_top1 = np.where(hit.any(axis=1), np.argmin(np.where(hit, rd, np.inf), axis=1), 0)
        agree = int(np.sum((ref_top1 == want["index"][:nr, 0]) | ~hit.any(axis=1)))
        base = {"value": nr / rdt, "unit": "queries/s", "cores": cores, "kind": "reference",
                "sample": f"{nr} queries of the timed batch x all {len(db)} eligible keyframes through the reference's own "
                          f"distanceBtnScanContext (oracle/_ref/libref_sc_sse2.so = Scancontext.cpp:116-148 compiled unmodified, "
                          f"{ref.build_info()}), OpenMP over (query, 64-entry block) on {cores} threads; 1 thread: {1.0 / r1:.2f} queries/s",
                "one_thread_value": 1.0 / r1, "top1_equal_to_oracle": agree, "queries": nr, "visible_cpus": os.cpu_count(),
                "cores_note": "threads = the cores this process may use (affinity mask and cgroup CPU quota, usable_cores())",
                "port_value": port["value"], "port_one_thread_value": one_thread_qps,
                "port_note": f"oracle/sc_ref.c (the non-allocating restatement, bit-identical results): {nq} exhaustive top-{k} "
                             f"queries, OpenMP over queries"}
    except (OSError, FileNotFoundError) as e:
        base = {"value": port["value"], "unit": "queries/s", "cores": cores, "kind": "port",
                "sample": f"{nq} exhaustive top-{k} queries (the first {nq} of the timed batch) vs the same {n}-keyframe DB, OpenMP "
                          f"over queries on {cores} threads (oracle/sc_ref.c; oracle/_ref is not built here: {e}); "
                          f"1 thread: {one_thread_qps:.2f} queries/s",
                "one_thread_value": one_thread_qps}
    check = {"oracle_checked_queries": nq, "oracle_identical_queries": int(np.sum(same)), "oracle_top1_identical": top1_same,
             "first_mismatch": None if all(same) else int(same.index(False))}
    return base, check


def orora_leg(device, skip_cpu):
    """Second half of the metric: ORORA scan-pairs/sec (BASELINE configs[2]: one KAIST03-length
    sequence, 3500 consecutive pairs, 300-1500 matches each, batched registration on one GPU)."""
    import torch
    from navtech_radar_slam_amd import orora, synth
    n_pairs = 3500
    src, dst, off, truth = synth.orora_pairs(777, n_pairs)
    reg = orora.Orora(device=device)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    d_off = torch.from_numpy(off).cuda()
    d_res = torch.zeros((n_pairs, 5), dtype=torch.float64, device="cuda")  # 40-byte rsx_orora_result
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        reg.register_batch_device(d_src.data_ptr(), d_dst.data_ptr(), d_off.data_ptr(), n_pairs, d_res.data_ptr(), stream=stream)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res = d_res.cpu().numpy().view(orora.ORORA_RESULT_DTYPE).reshape(n_pairs)
    err = max(np.abs(res["x"] - truth[:, 0]).max(), np.abs(res["y"] - truth[:, 1]).max())
    leg = {"pairs_per_sec": n_pairs / dt, "ms_per_batch": dt * 1e3, "n_pairs": n_pairs,
           "matches_per_pair": "300-1500", "outliers": "20-60%", "max_abs_translation_error_m": float(err),
           "max_abs_yaw_error_rad": float(np.abs(res["yaw"] - truth[:, 2]).max()), "dtype": "f64",
           "note": "latency/VALU-bound per pair (K x 16 B input); no HBM roofline applies (SURVEY 8d)"}
    # the stage in front of the solver in the upstream pipeline (round 6): max-clique inlier selection on the distance-consistency
    # graph of the matches (csrc/pmc.hip, RSX_ORORA_PMC; on by default in the odometry pipeline).  The same 3 500 pairs with it:
    # selection + solver in one call, the selection alone, what it keeps, and its records against the oracle
    from navtech_radar_slam_amd import _rsx as _r
    p_on = orora.default_params()
    p_on.flags |= _r.ORORA_PMC
    reg.reserve(int(off[-1]))
    d_mem = torch.zeros(int(off[-1]), dtype=torch.uint8, device="cuda")
    d_info = torch.zeros((n_pairs, 4), dtype=torch.int32, device="cuda")

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    dt_on = timed(lambda: reg.register_batch_device(d_src.data_ptr(), d_dst.data_ptr(), d_off.data_ptr(), n_pairs, d_res.data_ptr(), p_on, stream=stream))
    res_on = d_res.cpu().numpy().view(orora.ORORA_RESULT_DTYPE).reshape(n_pairs)
    dt_sel = timed(lambda: _r.check(_r.lib().rsx_orora_max_clique_batch_device(reg._h, d_src.data_ptr(), d_dst.data_ptr(), d_off.data_ptr(), n_pairs,
                                                                              None, d_mem.data_ptr(), d_info.data_ptr(), stream)))
    info = d_info.cpu().numpy()
    mem = d_mem.cpu().numpy()
    leg["with_max_clique_selection"] = {
        "pairs_per_sec": n_pairs / dt_on, "ms_per_batch": dt_on * 1e3, "selection_alone_ms_per_batch": dt_sel * 1e3,
        "selected_fraction_of_matches": float(info[:, 0].sum() / off[-1]), "pairs_proven_maximum_by_core_bound": float(np.mean(info[:, 3] & 1)),
        "seeds_per_pair": float(info[:, 2].mean()),
        "max_abs_translation_error_m": float(max(np.abs(res_on["x"] - truth[:, 0]).max(), np.abs(res_on["y"] - truth[:, 1]).max())),
        "max_abs_yaw_error_rad": float(np.abs(res_on["yaw"] - truth[:, 2]).max()), "dtype": "f64 predicate / u32 bitsets",
        "note": "consistency graph i ~ j <=> | |src_i - src_j| - |dst_i - dst_j| | < tim_noise_bound (fp64, K^2 predicates per pair), "
                "exact core numbers, greedy clique in (core, index) order, a vertex set = one 32-bit register per lane of a wavefront; "
                "fp64-VALU + latency bound, no HBM roofline applies"}
    if not skip_cpu:
        from oracle import pyoracle as po
        cores = usable_cores()
        t0 = time.perf_counter()
        want = po.orora_register_batch(src, dst, off, nthreads=cores)
        cdt = time.perf_counter() - t0
        leg["cpu_baseline"] = {"value": n_pairs / cdt, "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": f"the same {n_pairs} pairs, OpenMP over pairs (oracle/orora_ref.c)"}
        leg["max_abs_pose_diff_vs_oracle"] = float(max(np.abs(res[f] - want[f]).max() for f in ("x", "y", "yaw")))
        n_chk = n_pairs   # the selection is integer work: every pair of the batch is compared
        t0 = time.perf_counter()
        wm, winfo = po.pmc_select_batch(src[:off[n_chk]], dst[:off[n_chk]], off[:n_chk + 1], p_on.tim_noise_bound, nthreads=cores)
        sdt = time.perf_counter() - t0
        n_reg = 400
        s2, d2, o2 = po.pmc_compact(src[:off[n_reg]], dst[:off[n_reg]], off[:n_reg + 1], wm[:off[n_reg]])
        want_on = po.orora_register_batch(s2, d2, o2, nthreads=cores)
        sel = leg["with_max_clique_selection"]
        sel["oracle_checked_pairs"] = n_chk
        sel["selection_identical_to_oracle"] = bool(np.array_equal(mem[:off[n_chk]], wm) and all(np.array_equal(info[:n_chk, j], winfo[f]) for j, f in enumerate(("size", "max_core", "seeds", "flags"))))
        sel["solver_checked_pairs"] = n_reg
        sel["max_abs_pose_diff_vs_oracle"] = float(max(np.abs(res_on[f][:n_reg] - want_on[f]).max() for f in ("x", "y", "yaw")))
        sel["cpu_baseline"] = {"value": n_chk / sdt, "unit": "pairs/s (selection alone)", "cores": cores, "kind": "port",
                               "sample": f"the same {n_chk} pairs, OpenMP over pairs (oracle/pmc_ref.c)"}
    reg.close()
    return leg


def committed_profile(kernel):
    """What the COMMITTED rocprofv3 passes (tools/prof.sh runs this same workload; counters cannot be collected
    from inside the bench) say about the dominant kernel: HBM-side bytes per launch (FETCH_SIZE doubled: gfx950
    reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md; both counters are KB) and its average
    duration in the kernel trace.  -> dict or None"""
    import glob
    import re
    pat, tag = ("r*_sc_filter_v*_rocprofv3.txt", "FilterArgs") if kernel == "sc_filter_kernel" else ("r*_sc_spec_v*_rocprofv3.txt", "SpecArgs")
    def order(f):  # (round, version) numerically: r03_..._v10 comes after r03_..._v9
        m = re.search(r"r(\d+)_.*_v(\d+)_rocprofv3", os.path.basename(f))
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)), key=order)
    if not files:
        return None
    fetch = write = avg_us = None
    for line in open(files[-1]):
        if kernel not in line and tag not in line:
            continue
        m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\s*\d+\s+avg_per_dispatch=\s*([0-9.]+)", line)
        if m:
            if m.group(1) == "FETCH_SIZE":
                fetch = float(m.group(2))
            else:
                write = float(m.group(2))
        m = re.search(r"avg_us=\s*([0-9.]+)", line)
        if m and avg_us is None:
            avg_us = float(m.group(1))
    out = {"source": os.path.relpath(files[-1], ROOT), "kernel_trace_avg_launch_ms": None if avg_us is None else avg_us / 1e3}
    out["hbm_bytes_per_launch"] = None if fetch is None or write is None else (2.0 * fetch + write) * 1024.0
    return out


def cen2019_leg(device):
    """Third part of the path: cen2019 keypoint extraction on MulRan-shape polar scans (400 azimuths x
    3360 range bins, 11 metadata bytes per row).  Three figures: the synchronous single-scan host entry (1.35 MB PCIe upload
    and keypoint download inside), the batched host entry, and the batched device entry (images resident in HBM, nothing
    returns to the host).  Roofline: SURVEY 8d prices a scan at 1.344 MB of compulsory bytes; the batched chain reads the image
    twice (stats, openers; the runs pass works on 2 + 2 bytes per 8 pixels of records and the bytes of a row's ~14 candidate
    threads: profiles/r06_cen2019_wave_rocprofv3.txt) and is VALU-issue-bound (per-row scans: 88-99 % of the SIMD cycles), so
    the HBM fraction is reported for the record, not as the bound."""
    import ctypes as C
    import torch
    from navtech_radar_slam_amd import _rsx, cen2019, synth
    imgs = [synth.polar_image(100 + i)[0] for i in range(4)]
    ex = cen2019.Cen2019(rows=400, cols=3360, device=device)
    n = 0
    for i in range(3):
        n = len(ex.extract(imgs[i % 4]))
    reps = 30
    t0 = time.perf_counter()
    for i in range(reps):
        n = len(ex.extract(imgs[i % 4]))
    dt = (time.perf_counter() - t0) / reps
    # the same call with the image in page-locked memory (rsx_host_alloc_pinned): the upload is one asynchronous DMA
    with _rsx.PinnedArray(imgs[0].shape, np.uint8) as pin:
        pin.a[:] = imgs[0]
        for _ in range(3):
            ex.extract(pin.a)
        t0 = time.perf_counter()
        for _ in range(reps):
            n_pin = len(ex.extract(pin.a))
        dt_pin = (time.perf_counter() - t0) / reps
        same_pin = n_pin == len(ex.extract(imgs[0]))
    batch = 64
    stack = np.stack([imgs[i % 4] for i in range(batch)])
    ex.extract_batch(stack)
    t0 = time.perf_counter()
    for _ in range(3):
        ex.extract_batch(stack)
    dt_b = (time.perf_counter() - t0) / 3 / batch
    d = torch.from_numpy(stack).cuda()
    tg = torch.zeros((batch, 20000, 2), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(batch, dtype=torch.int32, device="cuda")
    p = _rsx.Cen2019Params(10000, 58)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _rsx.check(_rsx.lib().rsx_cen2019_extract_batch_device(ex._h, d.data_ptr(), batch, stack.strides[0], stack.shape[2], 11, C.byref(p),
                                                              None, 0, 0.0595, tg.data_ptr(), None, 20000, cnt.data_ptr(), st))
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    dt_d = (time.perf_counter() - t0) / 10 / batch
    ex.close()
    alg = 400 * 3360
    return {"scans_per_sec": 1.0 / dt, "ms_per_scan": dt * 1e3, "image": "400x3360 u8 (+11 B/row metadata)",
            "keypoints_last_scan": int(n), "dtype": "u8/f32/u64 keys", "includes": "H2D image + D2H keypoints (host-buffer entry)",
            "pinned_image": {"scans_per_sec": 1.0 / dt_pin, "ms_per_scan": dt_pin * 1e3, "same_keypoint_count": bool(same_pin)},
            "batched_host_scans_per_sec": 1.0 / dt_b, "batched_device_scans_per_sec": 1.0 / dt_d, "batch": batch,
            "launches_per_scan_or_batch": 10, "image_reads": {"batched": 2, "single_scan": 3},
            "intermediate_bytes_per_pixel": {"batched": 0.625, "single_scan": 0.375}, "algorithmic_bytes_per_scan": alg,
            "hbm_algorithmic_GBps_batched_device": alg / dt_d / 1e9, "hbm_frac_batched_device": alg / dt_d / 1e9 / HBM_PEAK_GBS,
            "note": "no sort, no host sync -- the greedy region marking in closed form; round 5: per-run maxima as plain v_max_f64 "
                    "scans of (segment | ord(h)) keys, marks as an OR over run flags, 2 + 1 bytes per 8 pixels between the passes; "
                    "10 launches per call whatever the batch (one memset + 9 kernels).  Round 6: in a batch the row kernels run one "
                    "WAVEFRONT per azimuth (chunks of 512 bins, scan totals in SGPRs, no barrier per row); the runs pass reads no image "
                    "-- per thread 2 + 2 bytes of records, the ~14 threads of a row that can hold a hit compacted into one pass; "
                    "single scans keep the workgroup-per-azimuth forms.  VALU-issue-bound, not HBM-bound"}


def allpairs_leg(device, n=100000, k=10):
    """BASELINE configs[4] on ONE GPU: every keyframe of a 100 000-keyframe DB against the keyframes at least 30 older
    than itself (rsx_sc_query_self_device: the filter's work items are the triangle of visible (tile-block, query tile)
    pairs), top-10.  Planted revisits must come back as top-1.  The 8-GPU form of the same job is tools/bench_allpairs.py."""
    import torch
    from navtech_radar_slam_amd import scancontext, synth
    descs = synth.random_descriptors(777, n, binary=True)
    rng = np.random.default_rng(1)
    loops = rng.integers(n // 2, n, 200)
    rots = rng.integers(0, 60, 200)
    for i, r in zip(loops, rots):                 # keyframe i repeats keyframe i - n/2, rotated
        descs[i] = synth.rotate_descriptor(descs[i - n // 2], int(r))
    g = scancontext.SCManager(device=device, capacity_hint=n)
    g.add_descriptors_f32(descs)
    out = torch.zeros((n, k, 2), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    g.query_self_device(0, 4096, k, out.data_ptr(), exclude_recent=30, stream=st)   # warm-up (workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.query_self_device(0, n, k, out.data_ptr(), exclude_recent=30, stream=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = out.cpu().numpy().view(scancontext.HIT_DTYPE).reshape(n, k)
    uniq = np.unique(loops, return_index=True)[1]            # a keyframe planted twice keeps its last rotation
    ok = bool(all(res["index"][i, 0] == i - n // 2 for i in loops[uniq]))
    g.close()
    pairs = n * (n - 31) / 2
    return {"seconds": dt, "queries_per_sec": n / dt, "eligible_pairs_per_sec": pairs / dt, "keyframes": n, "topk": k,
            "planted_revisits_recovered": ok, "workload": f"scancontext_allpairs_self_top{k}_db{n}_exclude30_random",
            "note": "BASELINE configs[4] (synthetic 100k-scan DB, all-queries distance matrix) on one MI355X; the matrix is "
                    "never materialised"}


def slam_stream_leg(device, db_pts, db_off, n_kf=3000, every=4):
    """BASELINE configs[3] in its one-GPU form: streaming SLAM emulation.  The trajectory keyframes arrive one by one
    (descriptor build on the GPU); at every 4th keyframe the detector runs in the reference's candidate mode
    (ring-key 3-NN + 3 pair distances, frozen tree prefix) AND in exhaustive mode.  Reports keyframes/s including both
    queries; results are compared with the oracle on the same stream."""
    from navtech_radar_slam_amd import scancontext
    from oracle import pyoracle as po
    n_kf = min(n_kf, len(db_off) - 1)
    cand = scancontext.SCManager(device=device, sc_dist_thres=0.45, capacity_hint=n_kf + 8)
    exh = scancontext.SCManager(device=device, sc_dist_thres=0.45, capacity_hint=n_kf + 8)
    got = []
    t0 = time.perf_counter()
    for i in range(n_kf):
        c = db_pts[db_off[i]:db_off[i + 1]]
        cand.makeAndSaveScancontextAndKeys(c)
        exh.makeAndSaveScancontextAndKeys(c)
        if i % every == every - 1:
            got.append((i, cand.detectLoopClosureID(full=True), exh.detectLoopClosureID(mode=scancontext.MODE_EXHAUSTIVE, full=True)))
    dt = time.perf_counter() - t0
    cand.close()
    exh.close()
    # the oracle on the same stream (candidate mode; exhaustive = scan of the same frozen prefix)
    o = po.Manager(dist_thres=0.45)
    cores = usable_cores()
    j = 0
    same_c = same_e = 0
    counter, tree = 0, 0
    for i in range(n_kf):
        o.add_points(db_pts[db_off[i]:db_off[i + 1]])
        if i % every == every - 1:
            want_c = o.detect_loop_closure()
            n = i + 1
            want_e = (-1, 0.0, 1e7, 0)
            if n >= 31:
                if counter % 30 == 0:
                    tree = n - 30
                counter += 1
                h = o.exhaustive(o.descriptor(i), n_eligible=tree, k=1, nthreads=cores)[0]
                if h["dist"] < 1e7:
                    yaw = float(np.float32(np.float64(np.float32(h["shift"] * 6.0)) * np.pi / 180.0))
                    want_e = (int(h["index"]) if h["dist"] < 0.45 else -1, yaw, float(h["dist"]), int(h["index"]))
            same_c += got[j][1] == want_c
            same_e += got[j][2] == want_e
            j += 1
    loops_c = sum(1 for g_ in got if g_[1][0] >= 0)
    loops_e = sum(1 for g_ in got if g_[2][0] >= 0)
    # the same stream from C++ (host/sc_shim_demo --stream: the SCManager shim, two managers, the clouds read into memory
    # first): what the loop costs without the Python harness around every call
    cpp = None
    exe = os.path.join(ROOT, "navtech-radar-slam_amd", "host", "sc_shim_demo")
    if os.path.exists(exe):
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".clouds", delete=False) as tf:
            path = tf.name
            tf.write(np.int32(n_kf).tobytes())
            for i in range(n_kf):
                c = np.ascontiguousarray(db_pts[db_off[i]:db_off[i + 1], :4], dtype=np.float32)
                tf.write(np.int32(len(c)).tobytes())
                tf.write(c.tobytes())
        try:
            r = subprocess.run([exe, "--stream", path, "--every", str(every)], capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("STREAM")]
            dets = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("DET")]
            if r.returncode == 0 and line:
                kv = dict(x.split("=") for x in line[0].split()[1:])
                same = len(dets) == len(got) and all(int(d[1]) == g_[0] and int(d[2]) == g_[1][0] and int(d[4]) == g_[2][0] and
                                                     abs(float(d[3]) - g_[1][1]) < 1e-6 and abs(float(d[5]) - g_[2][1]) < 1e-6 for d, g_ in zip(dets, got))
                cpp = {"keyframes_per_sec": float(kv["keyframes_per_sec"]), "seconds": float(kv["seconds"]), "detections": int(kv["detections"]),
                       "detections_identical_to_python_harness": bool(same),
                       "note": "host/sc_shim_demo --stream: the C++ SCManager shim (Scancontext.h), log lines off"}
            else:
                cpp = {"error": (r.stderr or r.stdout)[-300:]}
        finally:
            os.unlink(path)
    return {"keyframes": n_kf, "keyframes_per_sec": n_kf / dt, "seconds": dt, "cpp_host": cpp, "queries": len(got), "loops_candidate_mode": loops_c,
            "loops_exhaustive_mode": loops_e, "candidate_mode_identical_to_oracle": int(same_c), "exhaustive_mode_identical_to_oracle": int(same_e),
            "workload": f"streaming_slam_emulation_{n_kf}_trajectory_keyframes_detect_every_{every}",
            "note": "BASELINE configs[3] on one MI355X (two handles fed in parallel: candidate and exhaustive detector); the "
                    "sharded form is rsx_scs_* / sharded.py"}


def frontend_leg(device):
    """the ORORA front end between the cen2019 keypoints and the solver -- polar -> Cartesian remap,
    ORB-style descriptors, brute-force Hamming knnMatch(2) + ratio -- per scan, host buffers in and out."""
    from navtech_radar_slam_amd import cen2019, frontend, synth
    img0, az, _ = synth.polar_image(100, noise_seed=1)
    img1, _, _ = synth.polar_image(100, noise_seed=2)
    ex = cen2019.Cen2019(rows=400, cols=3360, device=device)
    fe = frontend.Frontend(400, 3360, device=device)

    def keypoints(img):
        t = ex.extract(img)
        rr = (t[:, 1] + 0.5) * synth.RADAR_RESOLUTION
        return np.stack([rr * np.cos(az[t[:, 0]]), rr * np.sin(az[t[:, 0]])], axis=1).astype(np.float32)

    xy0, xy1 = keypoints(img0), keypoints(img1)
    fe.cartesian(img0, az, synth.RADAR_RESOLUTION, want_image=False)
    d0, v0 = fe.describe(xy0)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        fe.cartesian(img1, az, synth.RADAR_RESOLUTION, want_image=False)
        d1, v1 = fe.describe(xy1)
        idx, _, _ = fe.match(d0, v0, d1, v1)
    dt = (time.perf_counter() - t0) / reps
    ex.close()
    fe.close()
    return {"ms_per_scan": dt * 1e3, "keypoints": [int(len(xy0)), int(len(xy1))], "ratio_test_matches": int((idx >= 0).sum()),
            "cartesian_image": "964x964 f32 @ 0.2592 m", "descriptor_bits": 256, "dtype": "u8/f32/u32 popcount",
            "hamming_pairs_per_scan": int(len(xy0)) * int(len(xy1)),
            "note": "remap + 7x7 blur + describe + knnMatch(2) incl. the PCIe copies of the synchronous host-buffer entry; "
                    "a radar delivers 4 scans/s"}


def odometry_e2e_leg(device, skip_cpu, n_unique=24, n_scans=528):
    """BASELINE configs[0] + configs[2]: the file-based odometry entry end to end on a MOVING sensor with known poses --
    polar scan -> cen2019 keypoints -> Cartesian image + ORB-style descriptors -> knnMatch(2) + ratio + cross check ->
    ORORA -> relative motion -- through the device-resident windowed pipeline (rsx_odometry_push: every stage once per
    window of 64 scans, all pairs of a window in ONE ORORA batch).  Sequence: `n_unique` synthetic MulRan-shape scans
    of one world along a drive (synth.polar_sequence), driven back and forth to `n_scans` scans (every consecutive pair
    is a real motion pair with a known answer).  Reported: scans/s with the images resident in HBM, scans/s from host
    memory (PCIe upload inside), the C++ entry on PNG files (decode threads + pipeline), PNG inflate cost separately."""
    import shutil
    import tempfile
    import torch
    from navtech_radar_slam_amd import odometry, synth
    imgs, az, poses, stamps = synth.polar_sequence(11, n_unique)
    order, i, step = [], 0, 1
    while len(order) < n_scans:                                   # 0 1 .. n-1 n-2 .. 1 0 1 ..
        order.append(i)
        if not 0 <= i + step < n_unique:
            step = -step
        i += step
    order = np.asarray(order)
    seq = np.ascontiguousarray(imgs[order])
    od = odometry.Odometry(400, 3360, device=device)
    od.push(seq[:200], az)                                         # warm-up: workspaces of both extraction lanes and all three sets, the Cartesian maps
    od.reset()
    t0 = time.perf_counter()
    res = od.push(seq, az)
    dt_host = time.perf_counter() - t0
    d = torch.from_numpy(seq).cuda()
    torch.cuda.synchronize()
    od.reset()
    t0 = time.perf_counter()
    res_dev = od.push(seq, az, device_ptr=d.data_ptr())
    dt_dev = time.perf_counter() - t0
    del d
    worst_t = worst_y = 0.0
    for i in range(1, n_scans):
        truth = synth.relative_pose(poses[order[i - 1]], poses[order[i]])
        worst_t = max(worst_t, float(np.hypot(res["x"][i] - truth[0], res["y"][i] - truth[1])))
        worst_y = max(worst_y, abs(float(res["yaw"][i] - truth[2])))
    leg = {"scans": n_scans, "unique_scans": n_unique, "window": 64,
           "scans_per_sec_resident": n_scans / dt_dev, "scans_per_sec_host_images": n_scans / dt_host,
           "ms_per_scan_resident": dt_dev / n_scans * 1e3, "ms_per_scan_host_images": dt_host / n_scans * 1e3,
           "identical_resident_vs_host": bool(res.tobytes() == res_dev.tobytes()),
           "pairs_registered": int((res["status"] == 0).sum()), "matches_per_pair_min_mean": [int(res["n_matches"][1:].min()), float(res["n_matches"][1:].mean())],
           "keypoints_per_scan_mean": float(res["n_keypoints"].mean()),
           "worst_pair_error_vs_truth": {"translation_m": worst_t, "yaw_rad": worst_y},
           "image": "400x3360 u8 (+11 B/row metadata), 1.35 MB per scan", "dtype": "u8/f32/u32 popcount/f64",
           "max_clique_selection": "on (rsx_odometry_default_params: RSX_ORORA_PMC)",
           "note": "resident: images already in HBM, 48 B per scan come back; host_images: pageable host array, the PCIe upload of "
                   "every window is inside the time; PNG decode excluded from both (reported under file_entry)"}
    if not skip_cpu:
        from oracle import odometry_chain
        t0 = time.perf_counter()
        chain = odometry_chain.run(imgs[:8], az, resolution=synth.RADAR_RESOLUTION)
        cdt = time.perf_counter() - t0
        diff = max(max(abs(float(res[f][i]) - float(chain[i]["result"][f])) for f in ("x", "y", "yaw")) for i in range(1, 8))
        same_counts = all(int(res["n_keypoints"][i]) == chain[i]["n_keypoints"] and int(res["n_matches"][i]) == chain[i]["n_matches"] for i in range(8))
        leg["cpu_baseline"] = {"value": 8 / cdt, "unit": "scans/s", "cores": 1, "kind": "port",
                               "sample": "the first 8 scans through oracle/odometry_chain.py (cen2019_ref -> frontend_ref -> orora_ref), 1 thread"}
        leg["oracle_checked_scans"] = 8
        leg["max_abs_pose_diff_vs_oracle"] = diff
        leg["counts_identical_to_oracle"] = bool(same_counts)
    # the C++ entry on PNG files: decode pool + pipeline
    exe = os.path.join(ROOT, "navtech-radar-slam_amd", "host", "odometry")
    tmp = tempfile.mkdtemp(prefix="rsx_odo_")
    try:
        from PIL import Image
        dd = os.path.join(tmp, "polar_oxford_form")
        os.makedirs(dd)
        uniq = []
        for i in range(n_unique):
            pth = os.path.join(tmp, f"u{i}.png")
            Image.fromarray(imgs[i], mode="L").save(pth)
            uniq.append(pth)
        for j, i in enumerate(order):
            os.link(uniq[i], os.path.join(dd, f"{1560000000000000000 + j * 250000000}.png"))
        r = subprocess.run([exe, f"seq_dir:={tmp}", f"device:={device}", "--timing", "--out", os.path.join(tmp, "poses.txt")],
                           capture_output=True, text=True, timeout=600)
        tl = [ln for ln in r.stderr.splitlines() if ln.startswith("timing:")]
        if r.returncode == 0 and tl:
            kv = dict(x.split("=") for x in tl[0].split()[1:])
            last = open(os.path.join(tmp, "poses.txt")).read().strip().splitlines()[-1].split()
            acc = np.zeros(3)
            for i in range(1, n_scans):
                if res["status"][i] == 0:
                    acc = synth.compose_pose(acc, (res["x"][i], res["y"][i], res["yaw"][i]))
            leg["file_entry"] = {"scans": int(kv["scans"]), "decode_threads": int(kv["decode_threads"]),
                                 "png_decode_ms_per_scan_per_thread": float(kv["decode_ms_per_scan_per_thread"]),
                                 "pipeline_scans_per_sec": float(kv["pipeline_scans_per_s"]), "total_scans_per_sec": float(kv["total_scans_per_s"]),
                                 "decode_wait_s": float(kv["decode_wait_s"]),
                                 "final_pose_equals_python_harness": bool(np.allclose([float(v) for v in last[1:4]], acc, atol=1e-4)),
                                 "note": "host/odometry seq_dir:=<dir> --timing: PNG inflate on a thread pool into pinned windows "
                                         "(double buffered) while the GPU runs the previous window; total = wall clock incl. decode"}
        else:
            leg["file_entry"] = {"error": (r.stderr or r.stdout)[-400:]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    od.close()
    return leg


def icp_leg(device):
    """the ICP loop verification that follows a ScanContext candidate
    (PGO.cpp:357-403): one keyframe scan against a +-25-keyframe submap, host buffers in."""
    from navtech_radar_slam_amd import icp
    rng = np.random.default_rng(5)
    walls = []
    for _ in range(40):
        a, b = rng.uniform(-80, 80, 2), rng.uniform(-80, 80, 2)
        t = rng.uniform(0, 1, 1500)[:, None]
        walls.append(np.c_[a + t * (b - a), rng.uniform(0, 3, 1500)])
    tgt = np.concatenate(walls).astype(np.float32)                      # 60 000-point submap
    yaw = 0.04
    R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    sub = tgt[rng.choice(len(tgt), 1500, replace=False)] + rng.normal(0, 0.02, (1500, 3))
    src = ((sub - np.array([0.8, -0.5, 0.0])) @ R).astype(np.float32)   # tgt = R src + t
    ic = icp.Icp(device=device)
    res = ic.align(src, tgt)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        res = ic.align(src, tgt)
    dt = (time.perf_counter() - t0) / reps
    ic.close()
    return {"ms_per_align": dt * 1e3, "source_points": len(src), "target_points": len(tgt), "iterations": res["iterations"],
            "converged": res["converged"], "fitness": res["fitness"], "accepted": bool(res["converged"] and res["fitness"] <= 0.3),
            "dtype": "f32 (fp64 moment sums)",
            "note": "one persistent launch per alignment (grid barrier per iteration); brute-force nearest neighbour, %.1e distance "
                    "evaluations per iteration; host-buffer entry (uploads both clouds, one read-back)" % (len(src) * len(tgt))}


def q1_latency_leg(device, q_descs):
    """ONE exhaustive query against N keyframes, N = 1 k / 10 k / 100 k: the regime of the live 1 Hz detector
    (PGO.cpp:561,577) and the only one where north_star's HBM roofline applies -- a single query cannot re-use anything, so
    every eligible entry is streamed out of HBM once.  Default path since round 5: ONE launch (sc_q1_kernel, csrc/sc_q1.hip)
    that streams the fp16 image + key image of every eligible entry (2672 B), previews every pair on the matrix cores and
    scores the few survivors exactly.  Time per call from the device's point of view (back-to-back calls on one stream over
    a pool of 16 different queries, query and result resident in HBM) and as one synchronous host call.  `hbm_frac` =
    SURVEY 8d's algorithmic bytes (N x 4800 B, one fp32 descriptor per pair) / time / 8 TB/s; `hbm_frac_read` = the bytes the
    kernel actually requests per entry (2672 B).  `floor_us` = the same call against a 32-keyframe DB: what one launch of
    this path costs before the first byte of a database is streamed (host call, kernel start, the query's images, the
    hand-off to the last workgroup, one exact evaluation).  The two older paths are timed beside it with the mode forced
    (exact-all: sc_pair2_kernel, every entry in fp64; filter: the batched chain of six launches) and must return the same
    records, as must 2..8 queries per call."""
    import torch
    from navtech_radar_slam_amd import scancontext, synth
    out = {}
    st = torch.cuda.current_stream().cuda_stream
    pool = np.ascontiguousarray(q_descs[:16])
    d_q = torch.from_numpy(pool).cuda()

    def stream_us(h, k, nq=1, reps=100):
        o = torch.zeros((16, k, 2), dtype=torch.float64, device="cuda")
        for i in range(8):
            h.query_device(d_q[i % (17 - nq):].data_ptr(), nq, k, o[i % (17 - nq):].data_ptr(), n_eligible=n_elig, stream=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            q0 = i % (17 - nq)
            h.query_device(d_q[q0:].data_ptr(), nq, k, o[q0:].data_ptr(), n_eligible=n_elig, stream=st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    def all_records(h, k, nq):
        o = torch.zeros((16, k, 2), dtype=torch.float64, device="cuda")
        for q0 in range(0, 16, nq):
            h.query_device(d_q[q0:].data_ptr(), min(nq, 16 - q0), k, o[q0:].data_ptr(), n_eligible=n_elig, stream=st)
        torch.cuda.synchronize()
        return o

    for n in (32, 1000, 10000, 100000, 200000, 400000):
        descs = synth.random_descriptors(77, n, binary=True)
        h = scancontext.SCManager(device=device, capacity_hint=n + 8)
        h.add_descriptors_f32(descs)
        n_elig = n - 30 if n > 100 else n
        dev_us = stream_us(h, 1)
        default_kernel = h.profiled_kernel_name()
        if n == 32:
            out["floor_us"] = dev_us
            out["floor_kernel"] = default_kernel
            h.close()
            continue
        if n > 100000:
            # past the 256 MiB Infinity Cache (the stream is 2672 B per entry: 534 MB at 200 k, 1.07 GB at 400 k): back-to-back
            # calls re-stream the same database, and at 100 k (267 MB) that could be served on-die -- these two sizes cannot
            hx = scancontext.SCManager(device=device, capacity_hint=n + 8, filter_mode=1)
            hx.add_descriptors_f32(descs)
            same = bool(torch.equal(all_records(hx, 1, 4)[:4], all_records(h, 1, 1)[:4]))
            hx.close()
            out[f"n{n}"] = {"us_per_query_stream": dev_us, "queries_per_sec_stream": 1e6 / dev_us, "default_path_kernel": default_kernel,
                            "first_4_queries_identical_to_exact_all": same,
                            "algorithmic_bytes": n_elig * ALG_BYTES_PER_PAIR, "bytes_requested": n_elig * 2672,
                            "hbm_frac": n_elig * ALG_BYTES_PER_PAIR / (dev_us * 1e-6) / (HBM_PEAK_GBS * 1e9),
                            "hbm_frac_read": n_elig * 2672 / (dev_us * 1e-6) / (HBM_PEAK_GBS * 1e9),
                            "beyond_infinity_cache": True}
            h.close()
            del descs
            continue
        dev_us_k10 = stream_us(h, 10)
        # the host-buffer entry as a C caller pays for it: the C-ABI call alone, argument marshalling outside the loop (the
        # Python wrapper's array conversions are 4-5 us a call)
        from navtech_radar_slam_amd.scancontext import HIT_DTYPE
        hout = np.zeros((1, 1), dtype=HIT_DTYPE)
        qptrs = [int(pool[i:i + 1].ctypes.data) for i in range(16)]
        optr = int(hout.ctypes.data)
        host_us = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            for i in range(20):
                st_ = h._L.rsx_sc_query(h._h, qptrs[i % 16], 1, 1, n_elig, optr)
            host_us = min(host_us, (time.perf_counter() - t0) / 20 * 1e6)
        assert st_ == 0
        want = all_records(h, 1, 1)
        same_q = all(bool(torch.equal(all_records(h, 1, nq), want)) for nq in (2, 3, 5, 8))
        forced = {}
        for name, mode in (("exact_all", 1), ("filter", 2)):
            hf = scancontext.SCManager(device=device, capacity_hint=n + 8, filter_mode=mode)
            hf.add_descriptors_f32(descs)
            forced[name] = (stream_us(hf, 1, reps=50 if name == "filter" or n <= 10000 else 10), bool(torch.equal(all_records(hf, 1, 1), want)))
            hf.close()
        out[f"n{n}"] = {"us_per_query_stream": dev_us, "us_per_query_host_call": host_us, "queries_per_sec_stream": 1e6 / dev_us,
                        "us_per_query_stream_top10": dev_us_k10,
                        "default_path_kernel": default_kernel,
                        "us_per_query_stream_exact_all": forced["exact_all"][0], "us_per_query_stream_filter_forced": forced["filter"][0],
                        "forced_paths_identical": forced["exact_all"][1] and forced["filter"][1],
                        "records_identical_2_to_8_queries_per_call": same_q,
                        "algorithmic_bytes": n_elig * ALG_BYTES_PER_PAIR, "bytes_requested": n_elig * 2672,
                        "hbm_frac": n_elig * ALG_BYTES_PER_PAIR / (dev_us * 1e-6) / (HBM_PEAK_GBS * 1e9),
                        "hbm_frac_read": n_elig * 2672 / (dev_us * 1e-6) / (HBM_PEAK_GBS * 1e9),
                        "hbm_frac_exact_all": n_elig * ALG_BYTES_PER_PAIR / (forced["exact_all"][0] * 1e-6) / (HBM_PEAK_GBS * 1e9)}
        h.close()
    out["note"] = ("one query, top-1, a pool of 16 queries in turn; us_per_query_stream = the default path (default_path_kernel: one "
                   "launch), back-to-back device-resident calls; host_call = the C-ABI call rsx_sc_query from host buffers (the query staged through pinned memory and copied up, the record written into pinned host memory by the kernel, one synchronise; best of 4 x 20 calls); "
                   "hbm_frac = N x 4800 B / time / 8 TB/s on the default path, hbm_frac_read = the 2672 B per entry it requests; "
                   "floor_us = the same call against 32 keyframes")
    return out


def loop_verify_leg(device):
    """SURVEY 8(f) rank 2 as the survey wrote it: doICPVirtualRelative (PGO.cpp:355-406) on keyframe clouds resident in HBM --
    submap assembly around the loop keyframe (+-25 keyframes through the root pose, PGO.cpp:329-352), VoxelGrid 0.4 m of both
    clouds, ICP, the fitness gate -- everything inside the time, one call per loop candidate.  A street driven twice
    (the second pass closes the loops); the verdict of the first candidate is compared with the oracle chain."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_loopverify import street_drive
    from navtech_radar_slam_amd import loopverify
    from oracle import pyoracle as po
    clouds, pose6 = street_drive(seed=3, n=120, step=2.0, revisit_at=70)
    kf = loopverify.KeyframeStore(device=device)
    t0 = time.perf_counter()
    for c in clouds:
        kf.add(c)
    t_add = (time.perf_counter() - t0) / len(clouds)
    pairs = [(i, 70 + i) for i in range(5, 45, 2)]            # the second pass over keyframes 5..43
    res = kf.verify(*pairs[0], pose6[pairs[0][0]])
    t0 = time.perf_counter()
    acc = 0
    its = 0
    for lo, cu in pairs:
        r = kf.verify(lo, cu, pose6[lo])
        acc += int(r["accepted"])
        its += r["iterations"]
    dt = (time.perf_counter() - t0) / len(pairs)
    want = po.loop_verify(clouds, pairs[0][0], pairs[0][1], pose6[pairs[0][0]], sum_order=po.ICP_SUM_TREE)
    # the independent restatement (float sums in ascending index order: the stand-in for PCL's own loops, oracle/icp_ref.c) as a
    # reported gate beside the tree-order one, which follows the device's summation order
    want_seq = po.loop_verify(clouds, pairs[0][0], pairs[0][1], pose6[pairs[0][0]], sum_order=po.ICP_SUM_SEQUENTIAL_FLOAT)
    t0 = time.perf_counter()
    m = kf.build_map(pose6, skip=2, leaf=0.4)
    t_map = time.perf_counter() - t0
    kf.close()
    return {"ms_per_verification": dt * 1e3, "verifications": len(pairs), "accepted": acc, "mean_icp_iterations": its / len(pairs),
            "source_points_after_voxelgrid": int(res["n_source"]), "target_points_after_voxelgrid": int(res["n_target"]),
            "keyframe_points": int(np.mean([len(c) for c in clouds])), "ms_per_keyframe_add": t_add * 1e3,
            "first_verdict_equals_oracle": bool(res["accepted"] == want["accepted"] and res["n_source"] == want["n_source"] and
                                                res["n_target"] == want["n_target"] and abs(res["fitness"] - want["fitness"]) < 1e-4 * max(1.0, want["fitness"]) and
                                                res["iterations"] == want["iterations"]),
            "first_verdict_vs_sequential_float_oracle": {"same_verdict": bool(res["accepted"] == want_seq["accepted"]),
                                                         "fitness_rel_diff": float(abs(res["fitness"] - want_seq["fitness"]) / max(1e-30, abs(want_seq["fitness"]))),
                                                         "iterations": [int(res["iterations"]), int(want_seq["iterations"])],
                                                         "note": "PCL's own source is absent: parity with PCL at the 0.3 fitness gate is NOT established; "
                                                                 "this is the oracle's second, order-independent reading"},
            "map": {"points": int(len(m)), "keyframes": len(clouds), "ms": t_map * 1e3},
            "dtype": "f32 (fp64 moment sums)",
            "launches_per_verification": 2,
            "note": "rsx_loop_verify: keyframe clouds in HBM; per candidate TWO launches -- both submaps transformed + VoxelGrid-"
                    "filtered by one cooperative kernel, the persistent ICP kernel (brute-force nearest neighbour, one grid barrier "
                    "per iteration) + gate -- and one read-back; the cloud sizes stay on the device in between"}


def layout_emulation_leg(device, db_descs, q_descs, n_elig, k, res=None, reps=5):
    """What ONE rank computes between the collectives in every layout of 2 / 4 / 8 GPUs, emulated on this one GPU.
    Q x S (query groups x DB shards): a handle holding shard 0 of S (the DB descriptors handed over, the handle keeps its
    residue class) and the first nq / Q queries; S > 1: stage 1 + stage 2 of the two-stage protocol (the stage-1 list of
    the shard itself as the bound: a looser bound than the merged one, so stage 2 is not under-estimated); S = 1: the
    single-stage query.  `Gf` (filter shards over a replicated DB, sharded.FilterShardedScanContext): the range filter of
    rank 0 for ALL queries (1 / G of the slots) + the short list / window / re-scoring of the first nq / G queries with the
    column blocks an all-to-all would have delivered (computed beforehand, untimed); its records are checked against the
    unsharded result `res`.  The exchanges are NOT in these numbers -- Q x S: latency-bound all-gathers of 16-byte
    records (~0.1 ms each, S > 1 only) and the final all-gather of the slices; Gf: one all-to-all whose bytes per rank are
    listed (`filter_shard_exchange`) -- the real curve is the driver's SCALE_r*.json."""
    import torch
    from navtech_radar_slam_amd import scancontext, sharded
    nq = len(q_descs)
    d_q = torch.from_numpy(q_descs).cuda()
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    cache = {}

    def timed(run):
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    for g in (1, 2, 4, 8):
        row = {}
        for q in [d for d in range(1, g + 1) if g % d == 0]:
            s_w = g // q
            nq_r = -(-nq // q)
            if (s_w, nq_r) not in cache:
                h = scancontext.SCManager(device=device, shard_rank=0, shard_world=s_w, capacity_hint=len(db_descs) // s_w + 8)
                h.add_descriptors_f32(db_descs)
                a = torch.zeros((nq_r, k, 2), dtype=torch.float64, device="cuda")
                b = torch.zeros((nq_r, k, 2), dtype=torch.float64, device="cuda")

                def run():
                    if s_w == 1:
                        h.query_device(d_q.data_ptr(), nq_r, k, a.data_ptr(), n_eligible=n_elig, stream=st)
                    else:
                        h.query_stage1_device(d_q.data_ptr(), nq_r, k, a.data_ptr(), n_eligible=n_elig, stream=st)
                        h.query_stage2_device(nq_r, k, a.data_ptr(), b.data_ptr(), stream=st)
                cache[(s_w, nq_r)] = timed(run)
                h.close()
            row[f"{q}x{s_w}"] = cache[(s_w, nq_r)]
        out[str(g)] = row
    # filter shards over the replicated DB
    h = scancontext.SCManager(device=device, capacity_hint=len(db_descs) + 8)
    h.add_descriptors_f32(db_descs)
    exchange, identical = {}, True
    for g in (2, 4, 8):
        lay = sharded.FilterShardedScanContext.__new__(sharded.FilterShardedScanContext)
        lay.world = g
        ld_r, rng = lay.ranges(n_elig)
        chunk = -(-nq // g)
        send = torch.zeros((g * chunk, ld_r), dtype=torch.float16, device="cuda")
        recv = torch.zeros((g, chunk, ld_r), dtype=torch.float16, device="cuda")
        for r, (first, cnt) in enumerate(rng):  # what rank 0 receives: rows [0, chunk) of every rank's matrix
            h.filter_range_device(d_q.data_ptr(), nq, first, cnt, send.data_ptr(), ld_r, stream=st)
            torch.cuda.synchronize()  # st is the handle's own stream, not ordered with torch's copy
            recv[r].copy_(send[:chunk])
            torch.cuda.synchronize()
        mine = torch.zeros((chunk, k, 2), dtype=torch.float64, device="cuda")

        def run():
            h.filter_range_device(d_q.data_ptr(), nq, rng[0][0], rng[0][1], send.data_ptr(), ld_r, stream=st)
            h.query_bounds_device(d_q.data_ptr(), min(chunk, nq), k, mine.data_ptr(), recv.data_ptr(), g, ld_r, chunk * ld_r,
                                  n_eligible=n_elig, stream=st)
        out[str(g)][f"{g}f"] = timed(run)
        if res is not None:
            got = mine.cpu().numpy().view(scancontext.HIT_DTYPE).reshape(chunk, k)
            identical = identical and bool(np.array_equal(got[:min(chunk, nq)], res[:min(chunk, nq)]))
        exchange[f"{g}f"] = {"all_to_all_bytes_sent_per_rank": int((g - 1) * chunk * ld_r * 2),
                             "ms_at_50GBps_per_peer_link": (chunk * ld_r * 2) / 50e9 * 1e3}
        del send, recv
    h.close()
    base = out["1"]["1x1"]
    return {"per_rank_ms_per_step": out, "best_layout": {g: min(r, key=r.get) for g, r in out.items()},
            "compute_speedup_of_best_layout": {g: base / min(r.values()) for g, r in out.items()},
            "compute_speedup_db_shards_only": {g: base / r[f"1x{g}"] for g, r in out.items()},
            "compute_speedup_filter_shards": {g: base / r[f"{g}f"] for g, r in out.items() if f"{g}f" in r},
            "filter_shard_exchange": exchange, "filter_shard_records_identical": identical if res is not None else None,
            "layout_key": "QxS = query groups x DB shards (two-stage protocol inside a group); Gf = filter shards over a "
                          "replicated DB (one all-to-all of bound rows)",
            "note": "per-rank compute between the collectives, emulated on ONE GPU (not a multi-GPU measurement); exchanges excluded. "
                    "ms_at_50GBps_per_peer_link: every peer pair moves its block over its own xGMI link concurrently"}


def host_entry_leg(mgr, q_descs, n_elig, k, resident_ms, reps=20):
    """the same batch through the synchronous HOST-buffer entry (rsx_sc_query): host queries in
    (nq x 4800 B over PCIe), host records out (nq x k x 16 B), upload and download inside the time.  The call cuts the batch
    into pieces and uploads piece c + 1 while piece c is filtered (sc_api.cpp host_pieces); timed from pageable memory (what a
    caller with a plain malloc'd buffer gets) and from page-locked memory (rsx_host_alloc_pinned)."""
    from importlib import import_module
    _rsx = import_module("navtech-radar-slam_amd._rsx")

    def timed(q, out):
        for _ in range(15):  # the leg follows host-side set-up with the GPU idle: let the clocks come back up (cf. settle_steps)
            mgr.query(q, k=k, n_eligible=n_elig, out=out)
        t0 = time.perf_counter()
        for _ in range(reps):
            mgr.query(q, k=k, n_eligible=n_elig, out=out)
        return (time.perf_counter() - t0) / reps

    nq = len(q_descs)
    dt_pageable = timed(q_descs, None)
    ref = mgr.query(q_descs, k=k, n_eligible=n_elig)
    with _rsx.PinnedArray((nq, 1200), np.float32) as pq, _rsx.PinnedArray((nq, k), ref.dtype) as po:
        pq.a[:] = q_descs.reshape(nq, 1200)
        dt_pinned = timed(pq.a, po.a)
        same = bool(np.array_equal(po.a, ref))
    return {"ms_per_step": dt_pageable * 1e3, "queries_per_sec": nq / dt_pageable, "memory": "pageable (what a caller with a plain buffer gets)",
            "pinned": {"ms_per_step": dt_pinned * 1e3, "queries_per_sec": nq / dt_pinned, "memory": "rsx_host_alloc_pinned"},
            "vs_resident": resident_ms / (dt_pageable * 1e3), "pinned_vs_resident": resident_ms / (dt_pinned * 1e3),
            "pinned_identical_to_pageable": same,
            "h2d_bytes": int(q_descs.nbytes), "d2h_bytes": nq * k * 16,
            "note": "rsx_sc_query: H2D of the queries in pieces, each FILTERED while the next one goes up, short lists / window previews / "
                    "re-scoring once over the whole batch, + D2H of the records; one synchronous call.  vs_resident = the headline's "
                    "ms_per_step (queries already in HBM) / this"}


def roofline_of(wl, launches, kern_ms, n_elig):
    ctx = wl.ctx
    local_pairs = wl.local_pairs(n_elig)  # pairs one launch of this rank scores
    alg_bytes = local_pairs * ALG_BYTES_PER_PAIR + wl.nq * 4800 + wl.nq * wl.k * 16
    avg_kern_s = (kern_ms / max(launches, 1)) * 1e-3
    kernel = wl.mgr.profiled_kernel_name()
    if kernel in ("sc_filter_kernel", "sc_spec_filter_kernel", "sc_spec2_filter_kernel"):
        spectral = kernel != "sc_filter_kernel"
        # queries whose 60 columns are all non-empty skip the n_eff mask correlation (3600 MAC per pair): n_eff is then
        # the entry's column count at every shift -- count only what was executed
        lo_q, hi_q, _ = wl.ssc._slice(wl.nq)
        full = float(np.mean((wl.q_host[lo_q:hi_q].reshape(hi_q - lo_q, 60, 20) != 0).any(axis=2).all(axis=1)))
        spec_flop = 2 * (9280 + 960 + 3600 * (1.0 - full))
        alg_flop = local_pairs * (spec_flop if spectral else ALG_FLOP_PER_PAIR)
        achieved = alg_flop / avg_kern_s / 1e12 if avg_kern_s > 0 else 0.0
        prof = committed_profile(kernel) if ctx.world == 1 and (wl.nq, n_elig) == (8192, 9970) else None
        alg_bytes = local_pairs * ALG_BYTES_PER_PAIR + (hi_q - lo_q) * 4800 + (hi_q - lo_q) * wl.k * 16
        return kernel, {
            "bound": "mfma", "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / MFMA_F16_PEAK_TFLOPS, "kernel": kernel, "launches": launches,
            "avg_launch_ms": kern_ms / max(launches, 1), "avg_launch_ms_source": "hipEvents recorded by librsx around every launch on its stream, this run",
            "traffic": prof["hbm_bytes_per_launch"] if prof else None, "traffic_unit": "bytes per launch",
            "traffic_source": (prof["source"] + " (committed rocprofv3 PMC passes of this workload; not measured in this run)") if prof else None,
            "committed_profile_avg_launch_ms": prof["kernel_trace_avg_launch_ms"] if prof else None,
            # the same algorithmic flops over the committed rocprofv3 --kernel-trace average (another box, under the profiler)
            "frac_at_committed_profile": (alg_flop / (prof["kernel_trace_avg_launch_ms"] * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS) if prof else None,
            "algorithmic_flop_per_launch": alg_flop,
            "algorithmic_flop_per_pair": spec_flop if spectral else ALG_FLOP_PER_PAIR,
            "queries_with_60_nonempty_columns": full,
            "direct_form_equivalent_tflops": local_pairs * ALG_FLOP_PER_PAIR / avg_kern_s / 1e12 if avg_kern_s > 0 else 0.0,
            "algorithmic_bytes_per_launch": alg_bytes,
            "note": ("dominant kernel = spectral fp16 MFMA lower-bound filter: the 60-shift circular cross-correlation of "
                     "a pair via a Z15 DFT + direct Z4 correlation (27.7 kflop per pair incl. the exact n_eff mask "
                     "correlation on the fp8 matrix cores) instead of 144 kflop per pair in the direct K = 1200 form; "
                     "`achieved` counts the spectral algorithm's own flops against the dense fp16 peak. "
                     if spectral else
                     "dominant kernel = fp16 MFMA lower-bound filter (144 kflop per (query, entry) pair: 60-shift circular "
                     "cross-correlation, K = 1200). ") +
                    "DB tiles are register-resident across the query batch, so the SURVEY 8d byte figure "
                    "(algorithmic_bytes_per_launch = 4800 B per pair) is not HBM traffic and no HBM fraction is quoted; "
                    "exact fp64 re-scoring of the survivors is inside value/ms_per_step"}
    hbm_alg = alg_bytes / avg_kern_s / 1e9 if avg_kern_s > 0 else 0.0
    return kernel, {"bound": "hbm", "achieved": hbm_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm_alg / HBM_PEAK_GBS, "traffic": None, "kernel": kernel, "launches": launches,
                    "avg_launch_ms": kern_ms / max(launches, 1), "algorithmic_bytes_per_launch": alg_bytes,
                    "note": "algorithmic bytes = 4800 B per (query, entry) pair; batched queries re-use DB tiles from "
                            "L2/Infinity Cache, so this is an algorithmic-throughput figure; the kernel itself is "
                            "fp64-VALU-bound (see DESIGN.md)"}


class LegGuard:
    """N > 1 only.  A secondary leg (another layout, the random / continuous-z / exact-all / 100 k workloads) must never cost the
    headline: RCCL beyond world 1 has never run on this code (no multi-GPU box in five rounds), and a wedged collective would sit
    there until the driver's own limit.  Each leg runs under a deadline (RSX_BENCH_LEG_TIMEOUT seconds, default 240).
      * the leg raises: recorded under `secondary_failures`, the run goes on (if only some ranks raised, the next collective
        wedges and the deadline of the next leg catches it);
      * the deadline passes: a rank stuck inside a collective cannot be interrupted, so every rank leaves through os._exit --
        rank 0 first writes the ONE JSON line with the headline it already holds and the failure recorded.  Exit status 0: the
        headline is a complete, valid measurement; what is missing says so in the line."""

    def __init__(self, ctx, out, failures, json_fd):
        self.ctx, self.out, self.failures, self.json_fd = ctx, out, failures, json_fd
        self.secondary = []
        self.timeout = float(os.environ.get("RSX_BENCH_LEG_TIMEOUT", "240"))
        self.active = ctx.world > 1
        self._lock = threading.Lock()

    def _expire(self, name):
        with self._lock:
            msg = f"{name}: no result after {self.timeout:.0f} s (wedged collective?) -- the run was cut here"
            sys.stderr.write(f"bench.py rank {self.ctx.rank}: {msg}\n")
            sys.stderr.flush()
            if self.ctx.rank == 0:
                self.secondary.append(msg)
                line = dict(self.out)
                line["secondary_failures"] = list(self.secondary)
                line["failures"] = list(self.failures)
                line["cut_short"] = True
                os.write(self.json_fd, (json.dumps(line) + "\n").encode())
            else:
                time.sleep(1.0)  # rank 0 writes first: the launcher tears the others down when one rank leaves
            os._exit(0)

    def fault(self, name):
        """test hook (tests/test_bench_launcher.py, stub backend only): RSX_BENCH_TEST_FAULT = raise:<leg> | sleep:<leg>"""
        spec = os.environ.get("RSX_BENCH_TEST_FAULT", "")
        if self.ctx.stub and spec.endswith(":" + name):
            if spec.startswith("raise:"):
                raise RuntimeError("injected fault")
            if spec.startswith("sleep:"):
                time.sleep(3600)

    def run(self, name, fn):
        if not self.active:
            return fn()
        timer = threading.Timer(self.timeout, self._expire, args=(name,))
        timer.daemon = True
        timer.start()
        try:
            self.fault(name)
            return fn()
        except Exception as e:  # noqa: BLE001
            self.secondary.append(f"{name}: {type(e).__name__}: {e}"[:300])
            return None
        finally:
            timer.cancel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--db", type=int, default=10000, help="keyframes in the (global) DB")
    ap.add_argument("--queries", type=int, default=8192, help="queries per step")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-main", action="store_true", help="time only the headline workload (profiling runs)")
    ap.add_argument("--data", choices=["trajectory", "random"], default="trajectory",
                    help="headline DB: descriptors built from a synthetic drive (default) or the round-1 random descriptors")
    ap.add_argument("--filter-kind", choices=["auto", "direct", "spectral", "spectral2"], default="auto",
                    help="form of the MFMA lower-bound filter (auto = the library default); A/B runs")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even at world 1")
    ap.add_argument("--all-layouts", action="store_true",
                    help="N > 1: time the mixed query-groups x DB-shards layouts too (default: the two pure ones and the headline)")
    ap.add_argument("--query-groups", type=int, default=1,
                    help="headline layout = query groups x DB shards (sharded.py); 1 (default) = pure DB shards, what north_star names "
                         "(the DB sharded over the GPUs, RCCL all-gather of the top-k); 0 = auto (as many query groups as the batch feeds)")
    args = ap.parse_args()

    dry = bool(os.environ.get("RSX_BENCH_LOCAL_BACKEND"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, dry))

    # the contract is ONE JSON line on stdout: libraries that chat on stdout (RCCL prints a version banner there)
    # are sent to stderr for the whole run; the line itself goes to the original descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.filter_kind != "auto" and not dry:
        from navtech_radar_slam_amd import _rsx as _r
        _r.default_filter_kind = {"direct": _r.KIND_DIRECT, "spectral": _r.KIND_SPECTRAL, "spectral2": _r.KIND_SPECTRAL2}[args.filter_kind]
    ctx = Ctx(args)
    from navtech_radar_slam_amd import scancontext, synth
    rank, world = ctx.rank, ctx.world
    n_db, nq, k = args.db, args.queries, args.topk
    n_elig = n_db - 30  # NUM_EXCLUDE_RECENT (SC.h:92): the newest 30 keyframes are never candidates
    out = {}

    # ---- headline workload ------------------------------------------------------------------
    t_gen = time.perf_counter()
    db_pts = db_off = None
    from navtech_radar_slam_amd.sharded import auto_layout
    qgroups = args.query_groups if args.query_groups > 0 else auto_layout(world, nq)
    auto_qg = auto_layout(world, nq)
    if world % qgroups:
        raise SystemExit(f"bench.py: --query-groups {qgroups} does not divide --gpus {world}")
    db_descs = None
    if args.data == "trajectory" and not ctx.stub:
        db_pts, db_off, q_pts, q_off, q_src = synth.trajectory_keyframes(1234, n_db, 4321, nq, binary_z=True)
        # descriptor-BUILD path, keyframe by keyframe, into a full replica on every rank; the (sharded) workloads of every
        # layout are then filled from its descriptors (a shard keeps its residue class of whatever it is handed)
        replica = scancontext.SCManager(device=ctx.local_rank, capacity_hint=n_db + 8)
        for i in range(n_db):
            replica.makeAndSaveScancontextAndKeys(db_pts[db_off[i]:db_off[i + 1]])
        db_descs = replica.export_descriptors_f32(0, n_db)
        replica.close()
        main_wl = Workload(ctx, "trajectory", k, n_db, query_groups=qgroups)
        main_wl.add_descriptors(db_descs)
        qb = scancontext.SCManager(device=ctx.local_rank, capacity_hint=nq + 8)
        for i in range(nq):
            qb.makeAndSaveScancontextAndKeys(q_pts[q_off[i]:q_off[i + 1]])
        q_descs = qb.export_descriptors_f32(0, nq)
        qb.close()
        del q_pts
        data_note = (f"DB = {n_db} keyframes of a synthetic drive (2 m apart, street grid, revisits in both directions), radar "
                     f"feature clouds (z = 0, ~1200 points) through the descriptor-build path; queries = {nq} new scans, "
                     f"{int((q_src >= 0).sum())} revisits of driven places + {int((q_src < 0).sum())} places never seen")
    else:
        descs, q_descs, r_src, r_rot = random_db_and_queries(n_db, nq)
        main_wl = Workload(ctx, "random", k, n_db, query_groups=qgroups)
        main_wl.add_descriptors(descs)
        if not ctx.stub:
            db_descs = descs
        data_note = "descriptor-level random binary DB, queries = rotated corrupted copies (planted loops)"
    main_wl.set_queries(q_descs, n_elig)
    gen_s = time.perf_counter() - t_gen

    dt, per_rank, (launches, kern_ms), (evals, cands) = main_wl.timed(args.steps, args.warmup, profile=True, settle_steps=96)
    res = main_wl.results()
    failures = []
    if args.data != "trajectory" or ctx.stub:
        ok = r_src < n_elig
        planted_ok = bool(np.all(res["index"][ok, 0] == r_src[ok]) and np.all(res["shift"][ok, 0] == r_rot[ok]))
        if not planted_ok:
            failures.append("planted loops not recovered as top-1")
    else:
        planted_ok = None

    if rank == 0:
        qps = nq * args.steps / dt
        out = {
            "metric": "sc_loop_queries_per_sec_vs_10k_scan_db", "value": qps, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f16 filter + f64 exact",
            "data": "synthetic",
            "config": {"workload": f"scancontext_exhaustive_top{k}_q{nq}_db{n_db}_{main_wl.name}", "db_keyframes": n_db,
                       "queries_per_step": nq, "topk": k, "rings_x_sectors": "20x60",
                       "parallelism": f"query_groups{qgroups}_x_db_shards{world // qgroups}" if world > 1 else "single_gpu",
                       "layout": main_wl.ssc.layout,
                       "value_is": "device-resident entry (rsx_sc_query_device: queries already in HBM, records left in HBM), "
                                   "as the bench contract asks; the PCIe-inclusive host-buffer entry is `host_entry` in this line",
                       "pairs_per_sec": qps * n_elig, "data_note": data_note, "data_generation_s": gen_s},
            "rccl_ranks": ctx.dist.get_world_size() if ctx.distributed else 1,
            "backend": (ctx.dist.get_backend() if ctx.distributed else "none"),
            "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
        }
        if ctx.stub:
            out["dry_run"] = True
            out["planted_loops_recovered"] = planted_ok
        else:
            kernel, roof = roofline_of(main_wl, launches, kern_ms, n_elig)
            out["dtype"] = "f16 filter + f64 exact" if "filter" in kernel else "f64"
            out["roofline"] = roof
            out["exact_evals_per_query"] = evals / max(1, nq * args.steps)
            out["previewed_candidates_per_query"] = cands / max(1, nq * args.steps)
            r3 = getattr(main_wl, "resc3", None)
            if r3:  # who served them: the matrix-core window kernel (sc_window.hip) or the per-wavefront VALU alignment + preview
                out["window_previews_per_query"] = r3[3] / max(1, nq * args.steps)
                out["valu_previews_per_query"] = r3[4] / max(1, nq * args.steps)
                out["exact_window_shifts_per_evaluation"] = r3[5] / max(1, r3[1])  # <= 7: shifts the previews could not exclude
            if planted_ok is not None:
                out["planted_loops_recovered"] = planted_ok

    # every leg below is secondary: at N > 1 each runs under LegGuard (its own try + deadline), so that neither an exception
    # nor a wedged collective in one of them can cost the headline already in `out`
    guard = LegGuard(ctx, out, failures, json_fd)

    # ---- every layout of this world (query groups x DB shards), same DB, same batch ------------------
    if world > 1 and not args.only_main:
        fill = db_descs if db_descs is not None else descs
        lay = {}
        st_l = max(3, args.steps // 4)

        def time_layout(qg=None, filter_shards=False):
            wl = Workload(ctx, main_wl.name, k, n_db, filter_shards=True) if filter_shards else Workload(ctx, main_wl.name, k, n_db, query_groups=qg)
            wl.add_descriptors(fill)
            wl.set_queries(q_descs, n_elig)
            dtl, prl, _, _ = wl.timed(st_l, 1)
            same = bool(np.array_equal(wl.results(), res))
            if not same:
                failures.append(f"layout {wl.ssc.layout} disagrees with layout {main_wl.ssc.layout}")
            lay[wl.ssc.layout] = {"ms_per_step": dtl / st_l * 1e3, "per_rank_ms_per_step": [t / st_l * 1e3 for t in prl],
                                  "exchange_ms_per_step_per_rank": ctx.all_gather_float(wl.exchange_ms_per_step), "identical_to_headline": same}
            if not filter_shards:
                lay[wl.ssc.layout]["auto_layout"] = qg == auto_qg
            wl.close()
            return True

        for qg in [d for d in range(1, world + 1) if world % d == 0]:
            if qg not in (1, world, qgroups, auto_qg) and not args.all_layouts:
                continue   # mixed layouts need torch.distributed subgroups: opt-in (--all-layouts); gloo-tested, never run on RCCL
            name = f"{qg}x{world // qg}"
            if qg == qgroups:
                lay[main_wl.ssc.layout] = {"ms_per_step": dt / args.steps * 1e3, "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
                                           "exchange_ms_per_step_per_rank": guard.run(f"layout {name} (exchange times)", lambda: ctx.all_gather_float(main_wl.exchange_ms_per_step)),
                                           "headline": True}
                continue
            if guard.run(f"layout {name}", lambda qg=qg: time_layout(qg)) is None:
                lay[name] = {"error": guard.secondary[-1]}
        if not ctx.stub:
            # filter shards over a replicated DB: has run in two processes on one GPU over gloo (tests/test_gpu_sc_layouts.py),
            # never on RCCL with more than one rank
            if guard.run(f"layout {world}f", lambda: time_layout(filter_shards=True)) is None:
                lay[f"{world}f"] = {"error": guard.secondary[-1]}
        if rank == 0:
            out["layouts"] = lay
            timed_l = {n: v["ms_per_step"] for n, v in lay.items() if "ms_per_step" in v}
            out["best_layout"] = min(timed_l, key=timed_l.get)
            out["best_layout_queries_per_sec"] = nq / (timed_l[out["best_layout"]] * 1e-3)
            out["layouts_key"] = ("QxS = query groups x DB shards; Gf = filter shards over a replicated DB (one all-to-all of bound rows); "
                                  "identical results in every layout.  `value` is the headline layout (default 1xN: the DB sharded over "
                                  "the GPUs, RCCL all-gather of the top-k, as north_star names it); best_layout is printed beside it; "
                                  "exchange_ms_per_step_per_rank = device time inside the collectives (waiting for the slowest peer included)")

    # ---- data dependence: the random DB and the exact-all floor, same batch shape --------------
    if not ctx.stub and not args.only_main:
        dd = {}

        def leg_random_db():
            descs, rq, r_src, r_rot = random_db_and_queries(n_db, nq)
            wl = Workload(ctx, "random", k, n_db, query_groups=qgroups)
            wl.add_descriptors(descs)
            wl.set_queries(rq, n_elig)
            dtr, _, _, (ev, cd) = wl.timed(max(3, args.steps // 2), 2, profile=True)
            r = wl.results()
            ok = r_src < n_elig
            rp = bool(np.all(r["index"][ok, 0] == r_src[ok]) and np.all(r["shift"][ok, 0] == r_rot[ok]))
            if not rp:
                failures.append("random DB: planted loops not recovered as top-1")
            steps_r = max(3, args.steps // 2)
            dd["random_db"] = {"queries_per_sec": nq * steps_r / dtr, "ms_per_step": dtr / steps_r * 1e3,
                               "exact_evals_per_query": ev / (nq * steps_r), "previewed_candidates_per_query": cd / (nq * steps_r),
                               "planted_loops_recovered": rp,
                               "workload": f"scancontext_exhaustive_top{k}_q{nq}_db{n_db}_random (round-1 headline data)"}
            wl.close()
            return True

        # the continuous-z family of SURVEY 8d "value distributions": the same drive with every landmark at its own height
        # (z ~ U(-1, 4): non-binary descriptors, no exact ties), DB and queries through the build path, oracle-checked
        def leg_continuous_z():
            c_pts, c_off, cq_pts, cq_off, _ = synth.trajectory_keyframes(1234, n_db, 4321, nq, binary_z=False)
            cb = scancontext.SCManager(device=ctx.local_rank, capacity_hint=n_db + nq + 8)
            for i in range(n_db):
                cb.makeAndSaveScancontextAndKeys(c_pts[c_off[i]:c_off[i + 1]])
            for i in range(nq):
                cb.makeAndSaveScancontextAndKeys(cq_pts[cq_off[i]:cq_off[i + 1]])
            c_all = cb.export_descriptors_f32(0, n_db + nq)
            cb.close()
            wl = Workload(ctx, "trajectory_continuous_z", k, n_db, query_groups=qgroups)
            wl.add_descriptors(c_all[:n_db])
            wl.set_queries(np.ascontiguousarray(c_all[n_db:]), n_elig)
            st_c = max(3, args.steps // 2)
            dtc, _, _, (evc, cdc) = wl.timed(st_c, 2, profile=True)
            rc = wl.results()
            dd["trajectory_continuous_z"] = {"queries_per_sec": nq * st_c / dtc, "ms_per_step": dtc / st_c * 1e3,
                                             "exact_evals_per_query": evc / (nq * st_c), "previewed_candidates_per_query": cdc / (nq * st_c),
                                             "workload": f"scancontext_exhaustive_top{k}_q{nq}_db{n_db}_trajectory_z_uniform(-1,4)"}
            if rank == 0 and not args.no_cpu_baseline:
                from oracle import pyoracle as po
                om = po.Manager()
                for i in range(n_db):
                    om.add_points(c_pts[c_off[i]:c_off[i + 1]])
                n_chk = 128
                want = om.exhaustive_batch(c_all[n_db:n_db + n_chk].astype(np.float64), n_eligible=n_elig, k=k, nthreads=usable_cores())
                ident = int(sum(bool(np.array_equal(rc[i], want[i])) for i in range(n_chk)))
                dd["trajectory_continuous_z"]["oracle_checked_queries"] = n_chk
                dd["trajectory_continuous_z"]["oracle_identical_queries"] = ident
                if ident != n_chk:
                    failures.append("continuous-z trajectory: GPU top-k differs from the oracle")
            wl.close()
            return True

        # exact-all: filter off, every (query, entry) pair through the fp64 pair kernel
        def leg_exact_all():
            wl = Workload(ctx, "exact_all", k, n_db, filter_mode=1, query_groups=qgroups)
            wl.add_descriptors(db_descs)
            wl.set_queries(q_descs, n_elig)
            dte, _, _, _ = wl.timed(3, 1)
            same = bool(np.array_equal(wl.results(), res))
            if not same:
                failures.append("filtered path and exact-all path disagree")
            dd["exact_all_floor"] = {"queries_per_sec": nq * 3 / dte, "ms_per_step": dte / 3 * 1e3,
                                     "exact_evals_per_query": float(len(range(wl.ssc.shard_rank, n_elig, wl.ssc.shard_world))),
                                     "identical_to_filtered_path": same,
                                     "note": "filter_mode = 1: every eligible pair scored by the exact fp64 kernel -- what the path "
                                             "costs when the data lets the filter prune nothing"}
            wl.close()
            return True

        # 100k-keyframe DB: where DB shards pay
        def leg_100k():
            n100 = 100000
            d100, q100, s100, r100 = random_db_and_queries(n100, nq, seed_db=2234, seed_q=5321)
            wl = Workload(ctx, "random100k", k, n100, query_groups=qgroups)
            wl.add_descriptors(d100)
            wl.set_queries(q100, n100 - 30)
            st100 = max(3, args.steps // 4)
            dt100, pr100, _, (ev100, cd100) = wl.timed(st100, 2, profile=True)
            r = wl.results()
            ok = s100 < n100 - 30
            p100 = bool(np.all(r["index"][ok, 0] == s100[ok]) and np.all(r["shift"][ok, 0] == r100[ok]))
            if not p100:
                failures.append("100k DB: planted loops not recovered as top-1")
            wl.close()
            if world == 1 and not ctx.stub:
                # BASELINE configs[4]: the per-layout figures for the DB size where sharding is supposed to pay
                emu100 = layout_emulation_leg(ctx.local_rank, d100, q100, n100 - 30, k, res=r, reps=3)
                if emu100["filter_shard_records_identical"] is False:
                    failures.append("layout emulation (100k): the filter-shard layout's records differ from the unsharded ones")
                out["layout_emulation_100k"] = emu100
            if rank == 0:
                out["scale_100k"] = {"value": nq * st100 / dt100, "unit": "queries/s", "ms_per_step": dt100 / st100 * 1e3, "steps": st100,
                                     "workload": f"scancontext_exhaustive_top{k}_q{nq}_db{n100}_random", "n_gpus": world, "scaling": "strong",
                                     "layout": f"{qgroups}x{world // qgroups}",
                                     "per_rank_ms_per_step": [t / st100 * 1e3 for t in pr100],
                                     "exact_evals_per_query": ev100 / (nq * st100), "previewed_candidates_per_query": cd100 / (nq * st100),
                                     "planted_loops_recovered": p100}
            return True

        if args.data == "trajectory":
            guard.run("data_dependence random_db", leg_random_db)
            guard.run("data_dependence trajectory_continuous_z", leg_continuous_z)
        guard.run("data_dependence exact_all_floor", leg_exact_all)
        if rank == 0:
            out["data_dependence"] = dd   # (filled in place by the legs above; scale_100k below is its own key)
        guard.run("scale_100k", leg_100k)

    if rank == 0 and not ctx.stub:
        if not args.only_main and world == 1:
            out["host_entry"] = host_entry_leg(main_wl.mgr, q_descs, n_elig, k, out["ms_per_step"])
            if not out["host_entry"]["pinned_identical_to_pageable"]:
                failures.append("host-buffer entry: pinned and pageable buffers give different records")
            if not np.array_equal(main_wl.mgr.query(q_descs[:64], k=k, n_eligible=n_elig), res[:64]):
                failures.append("host-buffer entry disagrees with the device entry")
            out["layout_emulation"] = layout_emulation_leg(ctx.local_rank, db_descs, q_descs, n_elig, k, res=res)
            if out["layout_emulation"]["filter_shard_records_identical"] is False:
                failures.append("layout emulation: the filter-shard layout's records differ from the unsharded ones")
        if not args.only_main:
            # BASELINE configs[1]: 1 query vs 1k-keyframe DB (latency of the synchronous host call)
            small = scancontext.SCManager(device=ctx.local_rank)
            small.add_descriptors_f32(main_wl.mgr.export_descriptors_f32(0, min(1000, main_wl.mgr.local_size)))
            for _ in range(5):
                small.query(q_descs[:1], k=1, n_eligible=970)
            t0 = time.perf_counter()
            for _ in range(50):
                small.query(q_descs[:1], k=1, n_eligible=970)
            out["latency_q1_n1k_us"] = (time.perf_counter() - t0) / 50 * 1e6
            small.close()
            out["latency_q1"] = q1_latency_leg(ctx.local_rank, q_descs)
            for nm in ("n1000", "n10000", "n100000"):
                lq = out["latency_q1"][nm]
                if not (lq["forced_paths_identical"] and lq["records_identical_2_to_8_queries_per_call"]):
                    failures.append(f"latency_q1 {nm}: the single-query path, the exact-all path and the filter chain disagree")
            for nm in ("n200000", "n400000"):
                if not out["latency_q1"][nm]["first_4_queries_identical_to_exact_all"]:
                    failures.append(f"latency_q1 {nm}: the single-query path and the exact-all path disagree")
            out["orora"] = orora_leg(ctx.local_rank, args.no_cpu_baseline)
            sel = out["orora"]["with_max_clique_selection"]
            if sel.get("selection_identical_to_oracle") is False or sel.get("max_abs_pose_diff_vs_oracle", 0.0) > 1e-4:
                failures.append("orora: the max-clique selection (or the solver behind it) differs from the oracle")
            out["cen2019"] = cen2019_leg(ctx.local_rank)
            out["icp"] = icp_leg(ctx.local_rank)
            out["loop_verify"] = loop_verify_leg(ctx.local_rank)
            out["frontend"] = frontend_leg(ctx.local_rank)
            out["odometry_e2e"] = odometry_e2e_leg(ctx.local_rank, args.no_cpu_baseline)
            oe = out["odometry_e2e"]
            if not oe["identical_resident_vs_host"] or oe.get("counts_identical_to_oracle") is False or oe.get("max_abs_pose_diff_vs_oracle", 0.0) > 1e-4:
                failures.append("odometry_e2e: pipeline differs from the oracle chain / between its two entries")
            if oe["worst_pair_error_vs_truth"]["translation_m"] > 0.3 or oe["worst_pair_error_vs_truth"]["yaw_rad"] > 1.5e-2:
                failures.append("odometry_e2e: a relative pose is off the known motion")
            out["allpairs_100k"] = allpairs_leg(ctx.local_rank)
            if not out["allpairs_100k"]["planted_revisits_recovered"]:
                failures.append("all-pairs: planted revisits not recovered")
            if db_pts is not None and not args.no_cpu_baseline:
                out["slam_stream"] = slam_stream_leg(ctx.local_rank, db_pts, db_off)
                ss = out["slam_stream"]
                if ss["candidate_mode_identical_to_oracle"] != ss["queries"] or ss["exhaustive_mode_identical_to_oracle"] != ss["queries"]:
                    failures.append("streaming SLAM emulation differs from the oracle")
                if ss.get("cpp_host") and ss["cpp_host"].get("detections_identical_to_python_harness") is False:
                    failures.append("streaming SLAM emulation: the C++ host's detections differ from the Python harness's")
        if not args.no_cpu_baseline:
            if db_pts is None:
                raise SystemExit("cpu_baseline needs --data trajectory (the oracle rebuilds the DB from the clouds)")
            base, chk = cpu_baseline_and_check(db_pts, db_off, q_descs, n_elig, k, res)
            out["cpu_baseline"] = base
            out.update(chk)
            if chk["oracle_identical_queries"] != chk["oracle_checked_queries"]:
                failures.append(f"GPU top-{k} differs from the oracle on query {chk['first_mismatch']}")
    if rank == 0:
        out["failures"] = failures
        if guard.secondary:
            out["secondary_failures"] = guard.secondary
        # BASELINE.json:metric has two halves ("SC loop-queries/sec ... + ORORA scan-pairs/sec"): the second one (and the
        # figures of the stages that feed it) ride inside `config`, the object every record of this line keeps whole
        sec = {}
        if "orora" in out:
            sec["orora_pairs_per_sec"] = out["orora"]["pairs_per_sec"]
            sec["orora_max_abs_pose_diff_vs_oracle"] = out["orora"].get("max_abs_pose_diff_vs_oracle")
            sec["orora_pairs_per_sec_with_max_clique_selection"] = out["orora"]["with_max_clique_selection"]["pairs_per_sec"]
        if "cen2019" in out:
            sec["cen2019_single_scan_ms_pinned"] = out["cen2019"]["pinned_image"]["ms_per_scan"]
            sec["cen2019_batched_device_scans_per_sec"] = out["cen2019"]["batched_device_scans_per_sec"]
        if "odometry_e2e" in out:
            sec["odometry_scans_per_sec_resident"] = out["odometry_e2e"].get("scans_per_sec_resident")
            fe_ = out["odometry_e2e"].get("file_entry")
            if isinstance(fe_, dict) and "total_scans_per_sec" in fe_:
                sec["odometry_png_files_scans_per_sec"] = fe_["total_scans_per_sec"]   # host/odometry incl. the PNG inflate
        if "latency_q1" in out:
            sec["single_query_us"] = {nm: v.get("us_per_query_stream") for nm, v in out["latency_q1"].items() if isinstance(v, dict) and "us_per_query_stream" in v}
        if "host_entry" in out:
            sec["host_buffer_queries_per_sec"] = out["host_entry"].get("queries_per_sec")
        if sec:
            out["config"]["secondary"] = sec
    main_wl.close()
    ctx.shutdown()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if failures:
        sys.exit(3)


if __name__ == "__main__":
    main()
