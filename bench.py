#!/usr/bin/env python3
"""bench.py -- EKF steps/s (predict + project/Jacobian + S_i + RANSAC + LI update + rescue + HI update) at
N=500 inverse-depth landmarks (n=3013), 200 RANSAC hypotheses, fp32 covariance path: BASELINE.json configs[2],
the configuration the headline metric is quoted on.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one whole filter step on one synthetic frame (3pre_amd/synth.py, deterministic seeds).  State
and covariance are resident in HBM before the timed region; per step only the frame's measurements (<= 400
pixels), the odometry increment and the hypothesis draws cross PCIe, as they would from the matcher.
N > 1: the EKF state does not shard (DESIGN.md "multi-GPU"), so each rank filters its own independent
sequence (weak scaling, no data-path collective); value = steps of all ranks / max-over-ranks time.
The sharded RANSAC scorer with its RCCL all-reduce is measured as a separate leg ("ransac_shard").

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK = {"f32": 157.3, "f64": 78.6, "bf16": 2500.0, "i8": 5000.0}       # dense MFMA TFLOP/s, MI355X_MICROARCH.md "Chip-level parameters" / "Matrix cores"


COLL_DEV = "cuda"        # device of the small tensors the bench all-reduces (rehearsal over gloo: "cpu")


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seq, threshold, budget_s=12.0):
    """The reference's algorithm on the host cores, timed over a bounded sample of the SAME sequence (checker code, used here only as a
    baseline).  Three legs, as BASELINE.md section 3 asks: the C restatement (oracle/pre3_oracle.c: explicit inv(S), K*S*K' as two
    products, sparse-H row products, sequential per-hypothesis loop; scalar C loops -- the image has no host BLAS to link -- with
    OpenMP over independent output rows) at ONE thread and at ALL cores, and the numpy twin (interpreter + OpenBLAS), the closest
    thing to MATLAB's own execution model.  `value` is the fastest of the three."""
    import oracle as orc
    twin = importlib.import_module("oracle.np_twin")
    orc.build()
    lib = orc.lib()
    N = seq["N"]
    types, off, _ = orc.landmark_table(np.zeros(N, int))
    n_hyp = seq["steps"][0]["hyp"].shape[0]

    def run(step_fn, budget):
        x, P = seq["x0"], seq["P0"]
        t_tot, done = 0.0, 0
        for s in seq["steps"]:
            t0 = time.perf_counter()
            o = step_fn(types, off, seq["cam"], x, P, s["u"], s["meas_idx"], s["z"], s["hyp"], threshold, early_exit=False)
            t_tot += time.perf_counter() - t0
            done += 1
            x, P = o["x_kk"], o["P_kk"]
            if t_tot > budget:
                break
        return done / t_tot, done

    ncores = os.cpu_count() or 1
    lib.orc_set_threads(1)
    c1, n1 = run(orc.step, budget_s)
    lib.orc_set_threads(0)
    call, nall = run(orc.step, budget_s)
    threads_all = int(lib.orc_get_max_threads())
    tw, ntw = run(twin.step, budget_s * 0.7)
    try:
        import threadpoolctl
        blas_threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = ncores
    # where ONE CPU step spends its time (numpy twin, the fastest leg): dense algebra of the two updates against the interpreted per-landmark
    # and per-hypothesis loops -- so that the GPU / CPU ratio can be read for what it is
    split = {}
    try:
        s0 = seq["steps"][0]
        cam = seq["cam"]
        mi = np.asarray(s0["meas_idx"], np.int64)
        tt = time.perf_counter(); x1, P1 = twin.predict(seq["x0"], seq["P0"], s0["u"]); split["predict"] = time.perf_counter() - tt
        tt = time.perf_counter()
        h_, has_ = twin.project(types, off, x1, cam); Hc_, Hl_ = twin.jacobian(types, off, x1, cam, h_, has_); twin.innovation(types, off, P1, Hc_, Hl_, has_)
        split["project_jacobian_S_i (per-landmark loops)"] = time.perf_counter() - tt
        zf = np.zeros((N, 2)); zf[mi] = s0["z"]
        tt = time.perf_counter(); rr = twin.ransac(types, off, x1, P1, Hc_, Hl_, zf, h_, mi, mi, cam, s0["hyp"], threshold, False); split["ransac (per-hypothesis loop)"] = time.perf_counter() - tt
        li_ = np.zeros(N, np.int32); li_[mi] = rr["li_mask"]
        tt = time.perf_counter(); x2, P2 = twin.update_landmarks(types, off, np.nonzero(li_)[0], x1, P1, Hc_, Hl_, zf, h_); split["LI update (dense: S, inv, K S K')"] = time.perf_counter() - tt
        tt = time.perf_counter()
        h2, has2 = twin.project(types, off, x2, cam, h_, has_); Hc2, Hl2 = twin.jacobian(types, off, x2, cam, h2, has2)
        ic_ = np.zeros(N, np.int32); ic_[mi] = 1
        hi_ = twin.rescue(types, off, P2, Hc2, Hl2, h2, zf, ic_, li_)
        split["rescue (per-landmark loops)"] = time.perf_counter() - tt
        tt = time.perf_counter(); twin.update_landmarks(types, off, np.nonzero(hi_)[0], x2, P2, Hc2, Hl2, zf, h2); split["HI update (dense)"] = time.perf_counter() - tt
        tot = sum(split.values())
        split = {k: round(1e3 * v, 2) for k, v in split.items()}
        split["dense_share"] = round((split["LI update (dense: S, inv, K S K')"] + split["HI update (dense)"]) / (1e3 * tot), 3)
    except Exception as e:                                     # (a baseline detail must never take the line down)
        split = {"error": repr(e)}
    legs = {"c_restatement_1_thread": {"value": c1, "steps": n1, "cores": 1},
            "c_restatement_all_cores": {"value": call, "steps": nall, "cores": threads_all},
            "numpy_twin_openblas": {"value": tw, "steps": ntw, "cores": int(blas_threads)}}
    best = max(legs, key=lambda k: legs[k]["value"])
    return {"value": legs[best]["value"], "unit": "steps/s", "cores": legs[best]["cores"], "kind": "port", "fastest_leg": best,
            "cpu_model": _cpu_model(), "host_cores": ncores, "legs": legs, "blas_threads": int(blas_threads),
            "one_step_split_ms": split,
            "sample": "first steps of the same N=%d, %d-hypothesis sequence (all hypotheses evaluated, fp64), ~%d s per leg: C restatement of "
                      "the reference at 1 thread and at all cores (scalar loops, OpenMP over rows; no host BLAS in the image), and the numpy "
                      "twin (Python loops + OpenBLAS); MATLAB itself is unavailable" % (N, n_hyp, int(budget_s))}


def all_ranks_ok(dist, ok):
    """True iff `ok` on every rank (one tiny all-reduce): a leg whose set-up failed somewhere is skipped everywhere, so that no
    rank waits in a collective the others never enter."""
    if dist is None:
        return ok
    import torch
    t = torch.tensor([1 if ok else 0], device=COLL_DEV, dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


COLLECTIVE_BY = {True: "%s enqueued by libpre3 on its own stream (pre3_comm_create: RCCL bound at run time), one host wait per round",
                 False: "torch.distributed between host synchronisations (3pre_amd/dist.py; the %s of the RCCL build is not used in this run)"}


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT (from C) when the first communicator of a process comes up: fd 1 goes to stderr meanwhile, so
    that the contract's ONE JSON line stays the only thing on stdout"""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def comm_leg(dist, local_rank, holder):
    """libpre3's own RCCL communicator, as a leg of its own: binding the library is local (set-up), creating the communicator is collective"""
    cm = importlib.import_module("3pre_amd.comm")
    with _StdoutToStderr():
        cm.unique_id()                                  # binds librccl.so in this process; raises if it cannot
    yield None
    with _StdoutToStderr():
        holder["comm"] = cm.Comm.from_torch_distributed(local_rank)
    yield holder["comm"].info()


def matcher_shard_leg(pre3, dist, rank, world, local_rank=0, K=4096, reps=100, comm=None):
    """BASELINE.json configs[3] over the ranks, device resident: the queries are replicated and every rank keeps its slice of the
    database packed in HBM (matcher.MatchShard); a match = int8-MFMA distance kernel on the slice, ONE all-gather of the per-query
    partials as device tensors (RCCL over xGMI), merge + ratio test + compaction on the device; only the match list crosses PCIe.
    Whole-job descriptor pairs/s (the collective and the merge are inside the timed region)."""
    import torch
    pd = importlib.import_module("3pre_amd.dist")
    mt = importlib.import_module("3pre_amd.matcher")
    rng = np.random.default_rng(5000)                   # same data on every rank
    L1 = np.minimum(np.round(np.abs(rng.standard_normal((K, 128))) * 40), 255).astype(np.uint8)
    L2 = np.clip(L1[rng.permutation(K)].astype(int) + rng.integers(-2, 3, (K, 128)), 0, 255).astype(np.uint8)
    L1c, L2c = np.asfortranarray(L1.T), np.asfortranarray(L2.T)      # 128 x K, column-major like the reference
    lo, hi = pd.shard_range(K, rank, world)
    sh = mt.MatchShard(L1c, L2c[:, lo:hi], lo, device=local_rank)
    # with libpre3's own communicator the whole match (kernel, ncclAllGather, merge) is enqueued on the shard's stream and waited for once
    one_match = (lambda: sh.match(1.5)) if comm is not None else (lambda: pd.siftmatch_sharded_resident(sh, 1.5))
    if comm is not None:
        sh.set_comm(comm)
    yield None                                          # set-up done: nothing collective has been issued yet (run_leg agrees on it)
    try:
        for _ in range(3):
            m = one_match()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            m = one_match()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=COLL_DEV, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
    finally:
        sh.close()
    yield {"workload": "configs[3]: 4096x4096x128 uint8, database columns sharded over %d GPU(s), operands resident in HBM, partials "
                        "all-gathered as device tensors, merge on the device" % world,
            "ms_per_match": 1e3 * el / reps, "pairs_per_s": reps * K * K / el, "matches": int(np.asarray(m).shape[1]), "n_gpus": world, "scaling": "strong",
            "collective": COLLECTIVE_BY[comm is not None] % "ncclAllGather"}


def ransac_shard_leg(pre3, synth, dist, rank, world, local_rank, N=2000, n_hyp=1000, reps=10, comm=None):
    """BASELINE.json configs[4]: N=2000 landmarks (n=12013), 1000 hypotheses sharded over the ranks with an RCCL
    all-reduce of supports + inlier masks (3pre_amd/dist.ransac_sharded); state replicated.  Whole-job
    hypotheses/s at this number of GPUs (the all-reduce and the replay are inside the timed region)."""
    import torch
    pd = importlib.import_module("3pre_amd.dist")
    seq = synth.make_sequence(N, 1, n_hyp)              # same seed on every rank: identical replicas
    s = seq["steps"][0]
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", device=local_rank, max_hyp=n_hyp)
    try:
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.ekf_prediction(s["u"])
        f.search_IC_matches()
        f.set_measurements(s["meas_idx"], s["z"])
        split = {}
        if comm is not None:                            # pre3_ransac_sharded: scoring, ncclAllReduce, selection on the filter's stream, one wait
            f.set_comm(comm)
            one_round = lambda timing=None: f.ransac_sharded_stream(s["hyp"], 1.0, early_exit=False, fetch=False)
        else:
            one_round = lambda timing=None: pd.ransac_sharded(f, s["hyp"], 1.0, early_exit=False, timing=timing, fetch=False)
        yield None                                      # set-up done: nobody enters the collectives unless everybody can (run_leg agrees on it)

        for _ in range(2):
            out = one_round()
        f.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = one_round(split)
        f.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
            t = torch.tensor([el], device=COLL_DEV, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        yield {"workload": "configs[4]: N=%d (n=%d), %d hypotheses (k=3), m=%d measured, f32; per round: H*P and H*P*H' gathers, "
                            "sharded scoring, all-reduce, replay" % (N, seq["n"], n_hyp, len(s["meas_idx"])),
                "value": reps * n_hyp / el, "unit": "hypotheses/s", "n_gpus": world, "ms_per_round": 1e3 * el / reps,
                "max_support": int(out["max_support"]), "scaling": "strong", "collective": COLLECTIVE_BY[comm is not None] % "ncclAllReduce",
                "ms_split_rank0": {k_: 1e3 * v / reps for k_, v in split.items()} if split else None}
    finally:
        f.close()


def matcher_leg(pre3, reps=20):
    """BASELINE.json configs[3]: 4096 x 4096 x 128 uint8 descriptors, int8-MFMA distance kernel with the best/second-best scan fused
    (one launch), inputs resident in HBM (HIP events around back-to-back launches: ~2.6 us of each is an empty launch)."""
    import ctypes as C
    rng = np.random.default_rng(5000)
    K = 4096
    L1 = np.minimum(np.round(np.abs(rng.standard_normal((K, 128))) * 40), 255).astype(np.uint8)
    L2 = np.clip(L1[rng.permutation(K)].astype(int) + rng.integers(-2, 3, (K, 128)), 0, 255).astype(np.uint8)
    lib = pre3._lib.lib
    h = lib.pre3_match_bench_create(0, 128, K, L1.ctypes.data_as(C.c_void_p), K, L2.ctypes.data_as(C.c_void_p))
    if not h:
        return None
    ms = C.c_double(0)
    rc = lib.pre3_match_bench_run(C.c_void_p(h), reps, C.byref(ms))
    lib.pre3_match_bench_destroy(C.c_void_p(h))
    if rc != 0:
        return None
    tops = 2.0 * K * K * 128 / (ms.value * 1e-3) / 1e12
    # The class matching_sift_based.m:104-118 actually passes is DOUBLE holding siftdescriptor.c:125-141's unit-norm real values (normalise,
    # clip at 0.2, renormalise): `double_unit_norm` is that distribution (bf16 distance GEMM ranks, guard-band candidates re-evaluated exactly;
    # pre3_match.hip "float / double classes").  Beside it: doubles that hold integers (Lowe-format files: int8 route) and a real-valued set
    # at the uint8 scale.
    fl = {}
    real1 = np.abs(rng.standard_normal((K, 128))) * 40
    real2 = real1[rng.permutation(K)] + rng.uniform(-2, 2, (K, 128))

    def sift_like(X):
        X = np.abs(X); X = X / np.linalg.norm(X, axis=1, keepdims=True)
        X = np.minimum(X, 0.2)
        return X / np.linalg.norm(X, axis=1, keepdims=True)
    unit1 = sift_like(rng.standard_normal((K, 128)))
    unit2 = sift_like(unit1[rng.permutation(K)] + 0.02 * np.abs(rng.standard_normal((K, 128))))
    for name, A, B in (("double_unit_norm", unit1, unit2), ("double_integer_valued", L1.astype(np.float64), L2.astype(np.float64)),
                       ("double_real_valued", real1, real2), ("float_real_valued", real1.astype(np.float32), real2.astype(np.float32))):
        hf = lib.pre3_match_bench_create_cls(0, 0 if A.dtype == np.float64 else 1, 128, K, A.ctypes.data_as(C.c_void_p), K, B.ctypes.data_as(C.c_void_p))
        if not hf:
            continue
        msf, info = C.c_double(0), (C.c_int32 * 3)()
        ok = lib.pre3_match_bench_run(C.c_void_p(hf), reps, C.byref(msf)) == 0 and lib.pre3_match_bench_info(C.c_void_p(hf), info) == 0
        lib.pre3_match_bench_destroy(C.c_void_p(hf))
        if ok:
            fl[name] = {"us_per_match": 1e3 * msf.value, "route": {0: "exact kernels", 1: "int8 MFMA", 2: "bf16 MFMA rank + exact re-evaluation"}[info[0]],
                        "candidates_per_query": info[2] / K, "queries_scanned_in_full": info[1]}
    return {"workload": "configs[3]: 4096x4096x128 uint8, 1 GPU", "ms_per_match": ms.value, "pairs_per_s": K * K / (ms.value * 1e-3),
            "int8_mfma_TOPS": tops, "float_classes": fl,
            "roofline": {"kernel": "k_match_i8_q (v_mfma_i32_16x16x64_i8, query-per-lane scan fused, one launch)", "bound": "mfma", "unit": "TOP/s",
                         "achieved": tops, "peak": PEAK["i8"], "frac": tops / PEAK["i8"], "traffic": None,
                         "algorithmic": "2 * K1 * K2 * 128 integer ops per match; the kernel is bound by its scan's vector instructions and by "
                                        "streaming the database through L2, not by the matrix pipe (DESIGN.md section 4, K11)"}}


def fp64_n200_leg(pre3, synth, steps=40, warm=4):
    """BASELINE.json configs[1] / BASELINE.md section 3: N=200 landmarks (n=1213), fp64, predict + update only (no RANSAC: every measured
    landmark is updated, r = 2 * 0.8 * 200 = 320 rows, SURVEY 8(d)'s example): the K9 launch (P <- P - W'W on v_mfma_f64_16x16x4_f64) timed with
    HIP events on the library's stream, its fraction of the f64 MFMA peak, and the max abs error of one predict + update against the C oracle
    (oracle/pre3_oracle.c: explicit inv(S), K*S*K' -- checker code, used here as the checker)."""
    import oracle as orc
    orc.build()
    N = 200
    seq = synth.make_sequence(N, steps + warm + 1, 8, outlier_frac=0.0)
    types, off, n = orc.landmark_table(np.zeros(N, int))
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f64", max_hyp=8, std_z=1.0)
    try:
        f.set_x_p_k_k(seq["x0"], seq["P0"])

        def one(s):
            f.step_all(s["u"], s["meas_idx"], s["z"])       # mono_slam.m's 'PURE_EKF' branch as one call (round 5; the four calls gave the same bits, 15 us slower)
        # timed first, checked afterwards: the C oracle's OpenMP pool keeps spinning on the host cores for a while after a call, and this
        # leg is host-driven (four calls per update): with the check in front the timed loop ran at 450 instead of 5700 updates/s
        for s in seq["steps"][:1 + warm]:
            one(s)
        f.sync()
        f.kernel_timing(8)      # one K9 launch in eight between HIP events, as the headline does: an event's record is a barrier packet with a completion signal, ~6 us of stream time each
        f.timer_start()
        for s in seq["steps"][1 + warm:1 + warm + steps]:
            one(s)
        ev = f.timer_stop()
        kt = f.kernel_timing_read()
        f.kernel_timing(False)
        # parity of one predict + update against the C oracle, from the same state
        s = seq["steps"][0]
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        one(s)
        xg, Pg = f.get_x_k_k(), f.get_p_k_k()
        x1, P1 = orc.predict(seq["x0"], seq["P0"], s["u"])
        h, has_h = orc.project(types, off, x1, seq["cam"])
        Hc, Hl = orc.jacobian(types, off, x1, seq["cam"], h, has_h)
        zfull = np.zeros((N, 2))
        zfull[s["meas_idx"]] = s["z"]
        xr, Pr = orc.update_landmarks(types, off, np.asarray(s["meas_idx"], np.int32), x1, P1, Hc, Hl, zfull, h)
        err_x, err_P = float(np.abs(xg - xr).max()), float(np.abs(Pg - Pr).max())
    finally:
        f.close()
    r = 2 * int(np.mean([len(s_["meas_idx"]) for s_ in seq["steps"][1 + warm:1 + warm + steps]]))
    t_s = kt["total_ms"] * 1e-3
    ach = kt["flops"] / t_s / 1e12 if t_s > 0 else 0.0
    return {"workload": "configs[1]: N=200 inverse-depth landmarks (n=%d), fp64, predict + update of all %d measured landmarks (r = %d rows), no RANSAC" % (n, r // 2, r),
            "updates_per_s": steps / (ev * 1e-3), "ms_per_predict_update": ev / steps,
            "k9": {"kernel": "k_downdate_1t<double> (v_mfma_f64_16x16x4_f64)", "launches": kt["launches"], "avg_launch_us": 1e3 * kt["total_ms"] / max(kt["launches"], 1),
                   "achieved": ach, "unit": "TFLOP/s", "peak": PEAK["f64"], "frac": ach / PEAK["f64"], "algorithmic": "SYRK n(n+1)r per launch"},
            "max_abs_err_vs_c_oracle": {"x": err_x, "P": err_P, "P_scale": float(np.abs(Pr).max()),
                                        "how": "one predict + update from the same state: GPU (Cholesky solve, symmetric down-date) vs oracle/pre3_oracle.c (explicit inv(S), K*S*K')"}}


def n2000_step_leg(pre3, synth, N=2000, n_hyp=1000, steps=6, warm=2):
    """BASELINE.json configs[4]'s state size on ONE GPU, the whole step ("the scaling axis is state dimension n"): N=2000
    landmarks (n=12013, P = 579 MB in fp32), 1000 hypotheses, ~1600 measured, LI update of ~2550 rows = 40 panels.  Steps/s, and the K9 launches
    of the LI updates priced like `roofline` (HIP events on the library's stream).  At 40 panels the factorisation takes the launch-per-panel form
    (the persistent launch was extended beyond its LDS window in round 4 and measured at 40 panels: no faster than the per-panel form, pre3_cholp.h), so the down-date is a launch of its own here."""
    seq = synth.make_sequence(N, steps + warm, n_hyp)
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
    try:
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.defer_hi_update(True)
        for s in seq["steps"][:warm]:
            f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
        f.sync()
        f.kernel_timing(1)
        f.timer_start()
        t0 = time.perf_counter()
        st = [f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False) for s in seq["steps"][warm:warm + steps]]
        ev = f.timer_stop()
        el = time.perf_counter() - t0
        kt = f.kernel_timing_read()
        f.kernel_timing(False)
        persistent = f.chol_persist()
    finally:
        f.close()
    n = seq["n"]
    t_s = kt["total_ms"] * 1e-3
    fused = kt["launches"] > 0 and kt["fused"] == kt["launches"]
    ach = (kt["flops"] + (kt["fact_flops"] if fused else 0.0)) / t_s / 1e12 if t_s > 0 else 0.0
    return {"workload": "configs[4] size on one GPU: N=%d (n=%d), %d hypotheses (k=3, all evaluated), %d measured, f32 covariance path, full 1PRE step" % (N, n, n_hyp, len(seq["steps"][0]["meas_idx"])),
            "value": steps / el, "unit": "steps/s", "ms_per_step": 1e3 * el / steps, "hip_event_ms_per_step": ev / steps,
            "mean_li_rows": 2 * float(np.mean([s_["n_li"] for s_ in st])), "mean_hi_rows": 2 * float(np.mean([s_["n_hi"] for s_ in st])),
            "factorisation": "one persistent launch with the down-date inside (k_cholp)" if fused else "launch per 64-row panel (k_chol_step; the persistent form is used up to 16 panels)",
            "persistent_form_available": bool(persistent),
            "k9": {"launches": kt["launches"], "avg_launch_us": 1e3 * kt["total_ms"] / max(kt["launches"], 1), "achieved": 6.0 * ach, "unit": "TFLOP/s", "peak": PEAK["bf16"],
                   "frac": 6.0 * ach / PEAK["bf16"], "f32_equivalent_ratio": ach / PEAK["f32"], "algorithmic": "SYRK n(n+1)r per LI launch, x6 executed (three-way bf16 split)"}}


def frame_leg(pre3, synth, N=500, K2=600, frames=40, warm=4, n_hyp=200):
    """One reference FRAME, not just the inner step (mono_slam.m:113-264): map_management (one delete_a_feature + one add_features_inverse_depth,
    map_management.m:27-79), ekf_prediction, search_IC_matches + matching_sift_based on the frame's SIFT set (K2 = 600 keypoints, SIFT_extract_save.m:68-89:
    descriptors are unit-norm doubles), RANSAC, LI update, rescue, HI update.  Frames/s with the scan crossing PCIe inside the timed region, and the
    per-stage split from a second pass with a stream synchronisation behind every stage."""
    rng = np.random.default_rng(7000)
    seq = synth.make_sequence(N, frames + warm, n_hyp, **{k: v for k, v in synth.HEADLINE.items() if k == "motion_noise"})
    thr = synth.HEADLINE["threshold"]

    def sift_like(X):
        X = np.abs(X); X = X / np.linalg.norm(X, axis=0, keepdims=True)
        X = np.minimum(X, 0.2)
        return X / np.linalg.norm(X, axis=0, keepdims=True)
    bank = sift_like(rng.standard_normal((128, N)))
    scans = []
    for s in seq["steps"]:                                   # the frame's SIFT set: the measured landmarks' descriptors (perturbed) at their pixels + distractors
        m = len(s["meas_idx"])
        d = sift_like(bank[:, s["meas_idx"]] + 0.01 * rng.standard_normal((128, m)))
        nd = max(0, K2 - m)
        d = np.concatenate([d, sift_like(rng.standard_normal((128, nd)))], 1)
        pos = np.concatenate([s["z"].T, np.stack([rng.uniform(1, 175, nd), rng.uniform(1, 143, nd)])], 1)
        pos = np.concatenate([pos, np.full((1, m + nd), 2.0), np.zeros((1, m + nd))], 0)
        perm = rng.permutation(m + nd)
        scans.append((np.asfortranarray(d[:, perm]), np.asfortranarray(pos[:, perm])))
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, max_landmarks=N + 2, std_z=thr)
    try:
        f.set_x_p_k_k(seq["x0"], seq["P0"])
        f.set_descriptors(bank)
        f.defer_hi_update(True)
        new_desc = sift_like(rng.standard_normal((128, frames + warm + 1)))

        def draws(m, u):                                     # select_random_match.m:47-51: 3 distinct measurements per hypothesis (vectorised; duplicates redrawn)
            # u: the frame's uniform variates, drawn while the device runs the IC search -- only their scaling by the match count waits for it
            if m <= 3:
                return synth.draw_hypotheses(rng, m, n_hyp)
            h = (u * m).astype(np.int32)
            bad = (h[:, 0] == h[:, 1]) | (h[:, 0] == h[:, 2]) | (h[:, 1] == h[:, 2])
            while bad.any():
                h[bad] = rng.integers(0, m, (int(bad.sum()), 3))
                bad = (h[:, 0] == h[:, 1]) | (h[:, 0] == h[:, 2]) | (h[:, 1] == h[:, 2])
            return h

        stage = {k: [] for k in ("map_management", "prediction", "scan_upload", "ic_search", "ransac_updates")}
        hyp_u = [None]

        def frame(k, split):
            t = [time.perf_counter()]

            def mark(name):
                if split:
                    f.sync()
                    t.append(time.perf_counter())
                    stage[name].append(t[-1] - t[-2])
            s = seq["steps"][k]
            # map_management.m:27-79: the landmark added last frame goes (frame 0: the last one of the map), a new one comes in
            f.map_management([f.N - 1], np.array([[rng.uniform(30, 140), rng.uniform(30, 110)]]), 1.0, 0.5)      # (pre3_map_management: one pass over P)
            f.set_descriptors(new_desc[:, k:k + 1], first=f.N - 1)
            mark("map_management")
            f.ekf_prediction(s["u"])
            mark("prediction")
            f.load_scan(*scans[k])
            mark("scan_upload")
            hyp_u[0] = rng.random((n_hyp, 3))                  # (while the device still pulls the scan)
            ic = f.matching_sift_based(1.5, strict_reference=True)
            mark("ic_search")
            if split and os.environ.get("PRE3_FRAME_DEBUG"):
                print("frame %d: ic_search %.0f us, ranked %s, matches %d" % (k, 1e6 * (t[-1] - t[-2]), f.ic_search_was_ranked(), len(ic["meas_idx"])), file=sys.stderr, flush=True)
            m = len(ic["meas_idx"])
            f.step_predicted(draws(m, hyp_u[0]), threshold=thr, early_exit=False)      # RANSAC, LI update, rescue, HI update: pre3_step's launches on the installed measurements
            mark("ransac_updates")
            return m
        for k in range(warm):
            frame(k, False)
        f.sync()
        t0 = time.perf_counter()
        ms = [frame(k, False) for k in range(warm, warm + frames // 2)]
        f.sync()
        el = time.perf_counter() - t0
        for k in range(warm + frames // 2, warm + frames):
            frame(k, True)
        ic_route = {2: "fused (exact 32 x 32 tiles riding in the projection launch + one gate workgroup: two launches)", 1: "ranked (bf16 matrix cores + exact tail)",
                    0: "exact 64 x 64 tiled kernel"}.get(f.ic_search_route(), "?")
    finally:
        f.close()
    return {"workload": "one mono_slam.m frame at N=%d (n=%d): map_management (1 delete + 1 add), prediction, IC search on a %d-keypoint SIFT set (unit-norm doubles, "
                        "uploaded per frame), RANSAC (%d hypotheses), LI update, rescue, HI update; f32 covariance path; threshold / motion noise as the headline" % (N, seq["n"], K2, n_hyp),
            "frames_per_s": len(ms) / el, "ms_per_frame": 1e3 * el / len(ms), "mean_ic_matches": float(np.mean(ms)), "ic_search_route": ic_route,
            "stage_us_synchronised": {k_: 1e6 * float(np.median(v)) for k_, v in stage.items()},
            "stage_us_max": {k_: 1e6 * float(np.max(v)) for k_, v in stage.items()},
            "note": "frames_per_s: no synchronisation inside a frame except what the calls themselves need (the IC search returns its match list, the RANSAC "
                    "statistics and the row counts are polled); stage_us: a second pass with a stream synchronisation behind every stage (median over its frames; stage_us_max: the slowest one -- a frame in a few thousand launches takes tens of ms in the runtime), so its sum exceeds ms_per_frame"}


def vo_leg(pre3, pnum=500, n_hyp=700, reps=50):
    """the VO front end's 4-point RANSAC (vodometry_dr_ye.m:171-236), 700 hypotheses over pnum matched 3-D points,
    inputs resident (pre3_vo_bench: the dist / score / final kernels, HIP events)."""
    vo = importlib.import_module("3pre_amd.vo")
    rng = np.random.default_rng(6000)
    a = rng.normal(0, 0.1, 3); th = np.linalg.norm(a); kx = a / th
    Kx = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    p2 = np.stack([rng.uniform(-1.5, 1.5, pnum), rng.uniform(-1, 1, pnum), rng.uniform(0.6, 5, pnum)])
    p1 = R @ p2 + rng.normal(0, 0.05, 3)[:, None] + rng.normal(0, 0.002, (3, pnum))
    bad = rng.choice(pnum, int(0.3 * pnum), replace=False)
    p1[:, bad] += rng.normal(0, 0.5, (3, len(bad)))
    match = np.stack([np.arange(1, pnum + 1), rng.permutation(pnum) + 1])
    draws = vo.draw_hypotheses(match, n_hyp, rng)
    ms = vo.vo_bench(p1, p2, draws, reps=reps)
    out = vo.vo_ransac(p1, p2, draws)
    return {"workload": "VO 4-point 3D-3D RANSAC: %d matches, %d hypotheses, 30 %% outliers" % (pnum, n_hyp), "us_per_ransac": 1e3 * ms,
            "hypotheses_per_s": n_hyp / (ms * 1e-3), "n_support": out["n_support"], "sta": out["sta"]}


def self_launch(args):
    """`python bench.py --gpus N` with no launcher environment: start the N ranks ourselves, as a child `torch.distributed.run`, BEFORE this
    process touches the GPU (a process that has initialised HIP must not exec), relay its output and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    import tempfile
    fd, side = tempfile.mkstemp(prefix="pre3_bench_", suffix=".json")
    os.close(fd)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PRE3_BENCH_SIDEFILE=side)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    seen = False
    for line in proc.stdout:                            # pass the ranks' output through; remember whether the line came
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.lstrip().startswith("{") and '"metric"' in line:
            seen = True
    rc = proc.wait()
    try:
        if not seen and os.path.getsize(side) > 0:
            # rank 0 measured the headline but never got to print (e.g. aborted inside a later collective): print what it left
            with open(side) as fh:
                line = fh.readline().strip()
            try:
                rec = json.loads(line)
                rec["communicator"] = "rank 0 ended (exit code %d) before printing; this is the headline it had measured before the auxiliary legs" % rc
                line = json.dumps(rec)
            except ValueError:
                pass
            print(line)
            sys.stdout.flush()
            rc = rc or 3
    finally:
        try:
            os.unlink(side)
        except OSError:
            pass
    return rc


def legs_agree(dist, err, name):
    """Collective-safe error hand-off after a leg's collectives: if the leg failed on ANY rank, every rank learns it here (one MIN
    all-reduce) and no rank walks into further collectives with a rank missing.  Returns True (every rank fine), False (a rank failed:
    the caller skips the remaining legs) or None (the exchange itself failed: the communicator is unusable, nothing collective may follow).
    The headline was measured before the legs and is still printed in every case -- an auxiliary leg must not cost the line."""
    if dist is None:
        return err is None                      # one process: the leg's error is recorded in the line, nothing can be left waiting
    try:
        ok = all_ranks_ok(dist, err is None)
    except Exception as e:                      # pragma: no cover  (a collective that timed out takes the communicator with it)
        sys.stderr.write("bench.py: the agreement exchange after leg %r failed (%r): no further collectives\n" % (name, e))
        sys.stderr.flush()
        return None
    if not ok:
        sys.stderr.write("bench.py: leg %r failed on a rank (%r) -- the remaining legs are skipped\n" % (name, err))
        sys.stderr.flush()
    return ok


def run_leg(dist, name, gen_fn, inject, hang=False):
    """A sharded leg is a generator: everything up to its first `yield` is set-up (no collective), the rest is the measured run (its own
    collectives) and yields the result.  Every rank takes the same path through here and issues the same collectives in the same order:
    set-up -> ONE agreement -> run -> ONE agreement.  A rank whose set-up fails (or is made to, `inject`) therefore cannot be left one
    collective apart from the others.  Returns (result | None, error | None, agreed) with agreed = True / False / None (communicator broken)."""
    gen, err, res = None, None, None
    try:
        if inject:
            raise RuntimeError("injected failure (PRE3_BENCH_FAIL_LEG): rehearses the error hand-off")
        gen = gen_fn()
        next(gen)
    except Exception as e:                              # pragma: no cover
        err = e
    agreed = legs_agree(dist, err, name + " (set-up)")
    if agreed:
        try:
            if hang:                                    # PRE3_BENCH_HANG_LEG: stands in for a collective whose partner never arrives (rehearses the watchdog)
                time.sleep(10 ** 6)
            res = next(gen)
        except Exception as e:                          # pragma: no cover
            err = e
        agreed = legs_agree(dist, err, name)
    if gen is not None:
        try:
            gen.close()                                 # runs the generator's `finally` (closes filters / shards)
        except Exception:                               # pragma: no cover
            pass
    return res, err, agreed


def check_step(pre3, f, seq, s, thr, dtype):
    """One more step behind the timed region, replayed on the numpy twin from the SAME starting state (the filter's own x_k_k, p_k_k):
    inlier sets and RANSAC statistics must be identical, state and covariance within the fp tolerance of the dtype."""
    twin = importlib.import_module("oracle.np_twin")
    import oracle as orc
    N = seq["N"]
    types, off, _ = orc.landmark_table(np.zeros(N, int))
    f.defer_hi_update(False)
    x0, P0 = f.get_x_k_k(), f.get_p_k_k()
    st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=thr, early_exit=False)
    li, hi = f.get_flags()
    xg, Pg = f.get_x_k_k(), f.get_p_k_k()
    ref = twin.step(types, off, seq["cam"], x0, P0, s["u"], s["meas_idx"], s["z"], s["hyp"], thr, early_exit=False)
    scale = float(np.abs(ref["P_kk"]).max())
    eP, ex = float(np.abs(Pg - ref["P_kk"]).max() / scale), float(np.abs(xg - ref["x_kk"]).max())
    tolP, tolx = (3e-4, 2e-5) if dtype == "f32" else (1e-11, 1e-10)
    li_ok, hi_ok = bool(np.array_equal(li, ref["li"])), bool(np.array_equal(hi, ref["hi"]))
    r = ref["ransac"]
    st_ok = (st["best"], st["max_support"]) == (r["best"], r["max_support"])
    return {"checked": bool(li_ok and hi_ok and st_ok and eP < tolP and ex < tolx), "li_set_equal": li_ok, "hi_set_equal": hi_ok,
            "ransac_winner_equal": bool(st_ok), "n_li": int(st["n_li"]), "n_hi": int(st["n_hi"]), "P_rel_err": eP, "x_abs_err": ex,
            "tolerance": {"P_rel": tolP, "x_abs": tolx},
            "how": "the step after the timed region, GPU vs oracle/np_twin.py from the filter's own state (LI/HI sets and the RANSAC winner exact)"}


def main():
    # The interpreter's cyclic garbage collector is off while the legs run (as `timeit` does for what it times) and collected between them: a
    # generation-2 pass over this process' objects takes ~40 ms -- 200 steps' worth -- and falls deterministically inside one timed loop or
    # another (tools/hiccup.py: every call above 1 ms disappears with gc.disable()).  It is the caller's interpreter, not the library.
    gc.collect()
    gc.disable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--landmarks", type=int, default=500)
    ap.add_argument("--hyp", type=int, default=200)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--kt-every", type=int, default=8, help="bracket one K9 launch in N with HIP events (an event pair costs ~11 us of stream time); 1 = every launch")
    ap.add_argument("--no-pend-hi", action="store_true", help="headline leg: the HI update's down-date as its own launch behind the update (PRE3_OPT_PEND_HI = 0, the library's default)")
    ap.add_argument("--sync-hi", action="store_true", help="complete every step's HI update inside the step call (default: deferred to the next call)")
    ap.add_argument("--k9-f32", action="store_true", help="fp32 path: K9 on the f32 MFMA instead of the three-way bf16 split (PRE3_OPT_K9_BF16X3 = 0)")
    ap.add_argument("--threshold", type=float, default=None, help="RANSAC threshold in pixels; default: the reference's own constant, 1.0 "
                    "(ransac_hypotheses.m:33 with mono_slam.m:78's sigma_image_noise = 1)")
    ap.add_argument("--motion-noise", type=float, default=None, help="how far the synthetic truth leaves the odometry, in units of the process noise the "
                    "filter assumes (3pre_amd/synth.py HEADLINE: 2.5 -- the 3-point hypothesis states are then off by a pixel or so and the chi2 rescue "
                    "has work in every step: ~36 HI rows in steps 5..25, ~31 over 60 steps, LI ~550 rows; 0.5 = the sequences of rounds 1-3)")
    ap.add_argument("--legacy-steps", type=int, default=60, help="steps of each of the two legs on the rounds-1-3 sequence (motion noise 0.5): `thr05` "
                    "(threshold 0.5 px, round 3's headline; it runs the headline's --steps / --warmup window) and `no_hi` (threshold 1.0, rounds 1-2); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of host work per CPU-baseline leg (three legs)")
    ap.add_argument("--no-check", action="store_true", help="skip the parity check of one step behind the timed region")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the sharded-RANSAC, matcher, fp64 and frame legs")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args))                 # nothing in this process has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PRE3_BENCH_REHEARSAL=gloo: every rank on cuda:0 with the gloo backend -- rehearses the whole multi-rank flow (self-launch, the
    # sharded legs' host-staged branches, the agreement exchanges, max-over-ranks timing) on a one-GPU box; the numbers mean nothing
    global COLL_DEV
    rehearsal = os.environ.get("PRE3_BENCH_REHEARSAL") == "gloo"
    if rehearsal:
        local_rank, COLL_DEV = 0, "cpu"
        os.environ["PRE3_CHOL_FORM"] = "0"          # several processes on one device: their persistent launches could interleave
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); refusing to print a line that would claim the wrong n_gpus\n" % (args.gpus, world))
        sys.exit(2)
    import torch
    dist = None
    if world > 1 or launched:                       # launched by torch.distributed.run (any N)
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on STDOUT when its communicator comes up: keep the contract's "ONE JSON line" by sending fd 1 to
        # stderr while the process group is created and its first collective runs
        sys.stdout.flush()
        saved_out = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=90))
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=90))
            dist.barrier()
            if not rehearsal:
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_out, 1)
            os.close(saved_out)
    pre3 = importlib.import_module("3pre_amd")
    synth = importlib.import_module("3pre_amd.synth")
    if pre3.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)

    N, K, W = args.landmarks, args.steps, args.warmup
    thr = synth.HEADLINE["threshold"] if args.threshold is None else args.threshold
    motion = synth.HEADLINE["motion_noise"] if args.motion_noise is None else args.motion_noise
    K2 = max(0, args.legacy_steps)

    def timed_steps(f, steps, th):
        """the contract's timed region: barrier + synchronize on both sides, max over ranks; K9 / fused-launch brackets collected meanwhile"""
        f.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        f.kernel_timing(max(1, args.kt_every))
        f.timer_start()
        t0 = time.perf_counter()
        st = [f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=th, early_exit=False) for s in steps]
        ev = f.timer_stop()                      # HIP events on the stream the kernels run on (synchronises; flushes a deferred HI update)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
            t = torch.tensor([el], device=COLL_DEV, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        k = f.kernel_timing_read()
        f.kernel_timing(False)
        return el, st, k, ev

    seq = synth.make_sequence(N, K + W + 1, args.hyp, seed=None if rank == 0 else 10_000 * rank + N, motion_noise=motion)     # + 1: the checked step
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype=args.dtype, device=local_rank, max_hyp=args.hyp, std_z=thr)
    b3 = f.k9_bf16x3(False if args.k9_f32 else None) if args.dtype == "f32" else False
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    pend_hi = False
    if not args.sync_hi and not args.no_pend_hi and args.dtype == "f32":
        pend_hi = f.pend_hi(True)                  # PRE3_OPT_PEND_HI: the HI update's down-date of P is taken along by the next step's launches (P swept once per
                                                   # step; the same arithmetic to fp32 rounding -- check_step below compares what was timed with the twin)
    if not args.sync_hi and os.environ.get("BENCH_WARM_DEFERRED", "1") != "0":
        f.defer_hi_update(True)                    # (round 6: the warm-up steps run in the timed steps' own form, so that the timed region does not hold the first launch of
                                                   #  the pending form's kernels; BENCH_WARM_DEFERRED=0: rounds 1-5's order, options switched on behind the warm-up)
    for s in seq["steps"][:W]:
        f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=thr, early_exit=False)
    if not args.sync_hi:
        f.defer_hi_update(True)                    # PRE3_OPT_DEFER_HI: the HI update of step k is completed by the call of step k+1 (same results;
                                                   # the caller's time between steps overlaps the rescue stage); the final timer_stop() flushes the last one
    elapsed, stats, kt, ev_ms = timed_steps(f, seq["steps"][W:W + K], thr)
    # parity of what was timed: the NEXT step of the same sequence, GPU vs the numpy twin (outside the timed region)
    chk = None
    if rank == 0 and not args.no_check:
        try:
            chk = check_step(pre3, f, seq, seq["steps"][W + K], thr, args.dtype)
        except Exception as e:                                  # pragma: no cover
            chk = {"checked": False, "check_error": repr(e)[:300]}
    f.close()

    def k9_rate(ktx):
        """the bracketed launches of a leg, priced like `roofline`"""
        if not (ktx["total_ms"] > 0 and ktx["launches"] > 0):
            return None
        six = 6.0 if (args.dtype == "f32" and b3) else 1.0
        fused = ktx["fused"] == ktx["launches"]
        work = ktx["flops"] + (ktx["fact_flops"] if fused else 0.0)
        ach = work / (ktx["total_ms"] * 1e-3) / 1e12
        pk = PEAK["bf16"] if six > 1 else PEAK[args.dtype]
        return {"kernel": "k_cholp (factorisation + solve + down-date in one launch)" if fused else "k_downdate_b3 / k_downdate_1t", "achieved": six * ach,
                "unit": "TFLOP/s", "peak": pk, "frac": six * ach / pk, "launches": ktx["launches"], "avg_launch_us": 1e3 * ktx["total_ms"] / ktx["launches"],
                "f32_equivalent_ratio": ach / PEAK["f32"] if args.dtype == "f32" else None}

    # Two legs on the sequence of rounds 1-3 (motion noise 0.5), a fresh filter: `thr05` = round 3's headline (threshold 0.5 px: ~260 LI inliers, the
    # rest of the true inliers come back through the rescue) over the same --steps / --warmup window, then `no_hi` = the same filter continued at the
    # reference's 1.0 px (rounds 1-2: the rescue finds next to nothing) -- so that the rounds stay comparable.
    thr05, no_hi, k9_alone = None, None, None
    if K2:
        seq2 = synth.make_sequence(N, W + K + 3 + K2, args.hyp, seed=None if rank == 0 else 10_000 * rank + N)
        f2 = pre3.EkfFilter(seq2["cam"], np.zeros(N, np.int32), dtype=args.dtype, device=local_rank, max_hyp=args.hyp, std_z=0.5)
        if args.dtype == "f32":
            f2.k9_bf16x3(False if args.k9_f32 else None)
        f2.set_x_p_k_k(seq2["x0"], seq2["P0"])
        for s in seq2["steps"][:W]:
            f2.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=0.5, early_exit=False)
        if not args.sync_hi:
            f2.defer_hi_update(True)
        el5, st5, kt5, _ = timed_steps(f2, seq2["steps"][W:W + K], 0.5)
        thr05 = {"value": world * K / el5, "unit": "steps/s", "steps": K, "warmup": W, "ms_per_step": 1e3 * el5 / K, "threshold_px": 0.5, "motion_noise": 0.5,
                 "mean_li_rows": 2 * float(np.mean([s_["n_li"] for s_ in st5])), "mean_hi_rows": 2 * float(np.mean([s_["n_hi"] for s_ in st5])),
                 "dominant_launch": k9_rate(kt5),
                 "note": "round 3's headline workload: threshold 0.5 px on the motion-noise-0.5 sequence, same --steps / --warmup window"}
        s2 = seq2["steps"][W + K:]
        for s in s2[:3]:
            f2.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
        el2, st2, kt2, _ = timed_steps(f2, s2[3:3 + K2], 1.0)
        no_hi = {"value": world * K2 / el2, "unit": "steps/s", "steps": K2, "ms_per_step": 1e3 * el2 / K2, "threshold_px": 1.0, "motion_noise": 0.5,
                 "mean_li_rows": 2 * float(np.mean([s_["n_li"] for s_ in st2])), "mean_hi_rows": 2 * float(np.mean([s_["n_hi"] for s_ in st2])),
                 "dominant_launch": k9_rate(kt2),
                 "note": "the same filter continued with the reference's own RANSAC threshold on the motion-noise-0.5 sequence: the headline "
                         "workload of rounds 1 and 2 (the rescue finds next to nothing)"}
        # the stand-alone K9 launch (k_downdate_b3 behind a finished factorisation: what rounds 1-3 priced as `roofline`) at the headline's row
        # count, on this leg's filter, which is done (the synthetic W of pre3_bench_downdate overwrites its P): the kernel-quality figure that the
        # fused launch's duration can no longer show
        try:
            r_k9 = int(round(2 * float(np.mean([s_["n_li"] for s_ in stats]))))
            ms_k9 = f2.bench_downdate(r_k9, 30)
            n_ = seq["n"]
            six = 6.0 if (args.dtype == "f32" and b3) else 1.0
            tf = n_ * (n_ + 1.0) * r_k9 / (ms_k9 * 1e-3) / 1e12
            k9_alone = {"kernel": "k_downdate_b3 as a launch of its own (pre3_bench_downdate: back-to-back launches, synthetic W, planes already split)", "rows": r_k9,
                        "avg_launch_us": 1e3 * ms_k9, "achieved": six * tf, "unit": "TFLOP/s", "peak": PEAK["bf16"] if six > 1 else PEAK[args.dtype],
                        "frac": six * tf / (PEAK["bf16"] if six > 1 else PEAK[args.dtype]), "f32_equivalent_ratio": tf / PEAK["f32"] if args.dtype == "f32" else None}
        except Exception as e:                                      # pragma: no cover
            k9_alone = {"error": repr(e)[:200]}
        f2.close()

    out = None
    if rank == 0:
        n = seq["n"]
        n_li = float(np.mean([s["n_li"] for s in stats]))
        n_hi = float(np.mean([s["n_hi"] for s in stats]))
        t_s = kt["total_ms"] * 1e-3
        achieved = kt["flops"] / t_s / 1e12 if t_s > 0 else 0.0                      # the SYRK count n(n+1)r alone
        fused = kt["launches"] > 0 and kt["fused"] == kt["launches"]                  # every bracketed launch was k_cholp with the down-date inside
        mean_r = (kt["flops"] / max(kt["launches"], 1)) / (n * (n + 1.0))
        traffic, traffic_src = None, None
        for tag, key in (("r6", "pmc_cholp"), ("r5", "pmc_cholp"), ("r4", "pmc_cholp"), ("r3", "pmc_k9")) if fused else (("r3", "pmc_k9"), ("r2", "pmc_k9"), ("r1", "pmc_k9")):
            try:     # HBM bytes per LI launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs)
                with open(os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, key))) as fh:
                    pj = json.load(fh)
                traffic = pj["hbm_bytes_per_li_launch"]["fetch_doubled"]
                traffic_src = ("profiles/%s_%s.json (WRITE_SIZE + 2 x FETCH_SIZE of the LI launches, mean r = %s: gfx950 reports half the bytes of 16 B/lane "
                               "reads, MI355X_MICROARCH.md; raw sum %.1f MB)" % (tag, key, pj.get("mean_rows", "?"), pj["hbm_bytes_per_li_launch"]["raw"] / 1e6))
                if not (fused and key == "pmc_k9"):
                    break
                traffic_src += " -- measured on the stand-alone K9 launch of round 3; no PMC pass of the fused launch is committed yet"
                break
            except Exception:
                pass
        if fused:
            # The dominant launch is now k_cholp: update.m:32-38 of an LI update in ONE launch -- factorisation of S (a latency chain on one
            # workgroup), blocked triangular solve, x-update, and P - W'W accumulated panel by panel by consumer workgroups on the other CUs.
            # Every product of the launch except the 64-column diagonal chains runs as six bf16 products with f32 accumulation.
            work = kt["flops"] + kt["fact_flops"]
            algo = ("per launch: SYRK n(n+1)r (the graded down-date, update.m:37) + r^3/3 + n r^2 (factorisation of S and W = L^-1 [HP | nu], update.m:32-33), "
                    "r = rows of that update; averaged over the launches of the timed region bracketed with HIP events on the library's stream "
                    "(one LI update in --kt-every); executed = 6 x that (three-way bf16 split); SURVEY 8(d)'s un-halved 2n^2r convention doubles the SYRK part")
            roofline = dict(kernel="k_cholp (persistent factorisation + solve + x-update with the K9 down-date inside: crit / row / strip workgroups and, on the CUs "
                                   "they leave idle, consumer workgroups holding 64x64 tiles of P as accumulators; six bf16 MFMA products per f32 product)",
                            bound="mfma", dtype="bf16 operands (3-way split of f32), f32 accumulate", unit="TFLOP/s", achieved=6.0 * work / t_s / 1e12, peak=PEAK["bf16"],
                            frac=6.0 * work / t_s / 1e12 / PEAK["bf16"], launches=kt["launches"], avg_launch_us=1e3 * kt["total_ms"] / kt["launches"], mean_rows=mean_r,
                            traffic=traffic, traffic_source=traffic_src, algorithmic=algo,
                            f32_equivalent={"achieved": work / t_s / 1e12, "f32_mfma_peak": PEAK["f32"], "ratio": work / t_s / 1e12 / PEAK["f32"],
                                            "note": "the launch priced as the f32 work it replaces (SYRK + factorisation + solve) / its whole duration"},
                            downdate_only={"syrk_TFLOPs_f32_equivalent": achieved, "ratio_of_f32_mfma_peak": achieved / PEAK["f32"],
                                           "executed_bf16_frac_of_peak": 6.0 * achieved / PEAK["bf16"],
                                           "note": "the SYRK count alone over the SAME duration (the launch also holds the factorisation's 60-us dependent chain): "
                                                   "what north_star's 'P-update >= 60 % of the fp32 MFMA roofline' would read if the whole launch were charged to K9; "
                                                   "the consumers' own matrix-pipe occupancy is in profiles/r6_pmc_cholp.json"})
        else:
            algo = ("symmetric rank-r down-date (SYRK): n(n+1)r flop per launch, r = rows of that update, averaged over the K9 launches of "
                    "the timed region that were bracketed with HIP events: one in --kt-every of the launches with >= 128 rows, i.e. the LI "
                    "updates; the HI updates' launches (r <= 64) are HBM-bound read-modify-writes of P and are not priced against the "
                    "matrix roofline; SURVEY 8(d)'s un-halved convention 2n^2r gives twice this figure")
            common = {"unit": "TFLOP/s", "traffic": traffic, "traffic_source": traffic_src, "launches": kt["launches"],
                      "avg_launch_us": 1e3 * kt["total_ms"] / max(kt["launches"], 1), "mean_rows": mean_r, "algorithmic": algo,
                      "survey_2n2r_equivalent": 2.0 * achieved * n / (n + 1.0)}
            if b3:
                # the launch executes six bf16 products per f32 product (three-way split of W): priced against the
                # bf16 dense peak on the flops it executes; the f32-equivalent rate is reported next to it
                roofline = dict(kernel="k_downdate_b3 (K9: P <- P - W'W; W split exactly into three bf16 planes, six bf16 MFMA products with "
                                       "f32 accumulation per f32 product; 128x128 + 64x64 tiles; the x-update and rescue-projection riders "
                                       "share the launch)",
                                bound="mfma", dtype="bf16 operands (3-way split of f32), f32 accumulate", achieved=6.0 * achieved, peak=PEAK["bf16"],
                                frac=6.0 * achieved / PEAK["bf16"],
                                f32_equivalent={"achieved": achieved, "f32_mfma_peak": PEAK["f32"], "ratio": achieved / PEAK["f32"],
                                                "note": "the same launches priced as the f32 SYRK they replace: n(n+1)r flop / time; BASELINE.json's "
                                                        "north star asks for the P-update at >= 60 % of the fp32 MFMA roofline -- this ratio is that figure"}, **common)
            else:
                roofline = dict(kernel="k_downdate_1t / k_downdate (K9: P <- P - W'W on the %s MFMA; the x-update and rescue-projection riders "
                                       "share the one-tile launch)" % args.dtype,
                                bound="mfma", dtype=args.dtype, achieved=achieved, peak=PEAK[args.dtype], frac=achieved / PEAK[args.dtype], **common)
        ms_step = 1e3 * elapsed / K
        # the step's ideal = its algorithmic flops at the MFMA peak of the dtype, no latency: F_upd(r) = 2*19*n*r + 2*19*r^2 + r^3/3
        # + 2*n*r^2 + 2*n^2*r + 2*n*r for the LI and the HI update (r = measured mean rows)
        def f_upd(r):
            return 2 * 19 * n * r + 2 * 19 * r * r + r ** 3 / 3.0 + 2 * n * r * r + 2.0 * n * n * r + 2 * n * r
        ideal_us = (f_upd(2 * n_li) + f_upd(2 * n_hi)) / (PEAK[args.dtype] * 1e12) * 1e6
        out = {
            "metric": "EKF steps/sec (predict+RANSAC+update) at N=500 landmarks; P-update %MFMA peak",
            "value": world * K / elapsed, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[2]: N=%d inverse-depth landmarks (n=%d), %d RANSAC hypotheses (k=3, all evaluated), "
                                   "%s covariance path, full 1PRE step" % (N, n, args.hyp, args.dtype),
                       "measured_per_step": int(np.mean([len(s["meas_idx"]) for s in seq["steps"][W:W + K]])),
                       "mean_li_rows": 2 * n_li, "mean_hi_rows": 2 * n_hi, "ransac_threshold_px": thr, "motion_noise": motion,
                       "sequence": "3pre_amd/synth.py HEADLINE: SURVEY 8(d)'s measurement spec (0.8 N measured, 20 % gross outliers, threshold 1.0 px); the truth "
                                   "leaves the odometry by motion_noise x the filter's process noise, which is what gives rescue_hi_inliers.m work in every step",
                       "parallelism": "replicas x%d" % world,
                       "options": {"PRE3_OPT_DEFER_HI": 0 if args.sync_hi else 1, "PRE3_OPT_PEND_HI": 1 if pend_hi else 0},
                       "hip_event_ms_per_step": ev_ms / K},
            "roofline": roofline,
            "k9_standalone": k9_alone,
            "thr05": thr05,
            "no_hi": no_hi,
            "step_ideal": {"ideal_us": ideal_us, "achieved_us": 1e3 * ms_step, "step_ideal_frac": ideal_us / (1e3 * ms_step),
                           "note": "SURVEY 8(d): algorithmic flops of the LI + HI updates at the %s MFMA peak, no latency" % args.dtype},
        }
    if rank == 0 and chk is not None:
        out.update(chk)
    # secondary legs: sharded RANSAC and sharded matcher at every N, kernel-only matcher and VO RANSAC at N=1.  A leg that fails on any
    # rank is recorded and ends the legs (legs_agree): no rank may be left inside a collective, and the headline is printed regardless.
    comm_broken = False
    side = os.environ.get("PRE3_BENCH_SIDEFILE")
    if rank == 0 and side:
        # the headline alone, before any auxiliary collective: the self-launch parent (which never touches the GPU) prints this line if
        # this process should die inside a collective (an RCCL watchdog abort leaves no chance to print)
        try:
            with open(side, "w") as fh:
                fh.write(json.dumps(out) + "\n")
        except OSError:                                         # pragma: no cover
            pass
    # A collective of an auxiliary leg whose partner never arrives does not return (libpre3's own communicator has no time-out; torch's
    # aborts the process after its 90 s): under an external launcher nobody would print the headline then.  A watchdog thread does: if the
    # legs are not through after PRE3_BENCH_LEG_TIMEOUT seconds (default 150), rank 0 prints the headline it already has -- with the
    # legs marked as timed out -- and every rank leaves with exit code 3.
    import threading
    printed = threading.Lock()
    def _watchdog():
        if not printed.acquire(blocking=False):
            return                                              # the main thread is printing / has printed
        if rank == 0:
            o = dict(out) if out is not None else {}
            o["legs"] = "timed out inside an auxiliary leg's collective (exit code 3); the headline was measured before the legs"
            sys.stdout.write(json.dumps(o) + "\n")
            sys.stdout.flush()
        sys.stderr.write("bench.py: rank %d: the auxiliary legs did not finish in time -- leaving\n" % rank)
        sys.stderr.flush()
        os._exit(3)
    wd = None
    if not args.no_extra_legs and (dist is not None or os.environ.get("PRE3_BENCH_COMM", "1") != "0"):
        wd = threading.Timer(float(os.environ.get("PRE3_BENCH_LEG_TIMEOUT", "150")) + (0.0 if rank == 0 else 5.0), _watchdog)
        wd.daemon = True
        wd.start()
    holder = {}
    if not args.no_extra_legs:
        legs = [("ransac_shard", lambda: ransac_shard_leg(pre3, synth, dist, rank, world, local_rank, comm=holder.get("comm"))),
                ("matcher_shard", lambda: matcher_shard_leg(pre3, dist, rank, world, local_rank, comm=holder.get("comm")))]
        if not rehearsal and os.environ.get("PRE3_BENCH_COMM", "1") != "0":     # (gloo rehearsal: several ranks share one GPU, which RCCL refuses)
            legs.insert(0, ("rccl", lambda: comm_leg(dist, local_rank, holder)))
        for name, fn in legs:
            leg, err, agreed = run_leg(dist, name, fn, os.environ.get("PRE3_BENCH_FAIL_LEG") == name and rank == world - 1,
                                       hang=os.environ.get("PRE3_BENCH_HANG_LEG") == name and rank == world - 1)
            if rank == 0:
                out[name] = leg if (err is None and agreed) else {"error": repr(err)[:300] if err is not None else "failed on another rank"}
            if agreed is None:
                comm_broken = True
            if not agreed and name == "rccl" and agreed is not None:
                holder.pop("comm", None)                        # no communicator of libpre3's own: the legs go through torch.distributed
                continue
            if not agreed:
                break                                           # no further collectives after a failed leg
        if holder.get("comm") is not None and not comm_broken:
            holder.pop("comm").close()
        if world == 1:
            for name, fn in (("matcher", lambda: matcher_leg(pre3)), ("vo_ransac", lambda: vo_leg(pre3)), ("fp64_n200", lambda: fp64_n200_leg(pre3, synth)),
                             ("frame", lambda: frame_leg(pre3, synth)), ("n2000_step", lambda: n2000_step_leg(pre3, synth))):
                try:
                    gc.collect()                                 # (between legs: see main)
                    out[name] = fn()
                except Exception as e:                          # pragma: no cover
                    out[name] = {"error": repr(e)[:300]}
    # the line goes out BEFORE the final barrier and the teardown: a collective whose partner is gone does not raise on RCCL, it aborts
    if not printed.acquire(blocking=False):
        time.sleep(30)                                          # the watchdog is printing and leaving: nothing more to do here
        os._exit(3)
    if wd is not None:
        wd.cancel()
    # the CPU baseline LAST (no collective in it: the watchdog is off): its thread pools (256 OpenMP threads of the C restatement, OpenBLAS under
    # the numpy twin) keep spinning on the host cores for a while after each call and slowed every host-driven GPU leg that used to follow them
    # (fp64_n200: 425 instead of 5700 updates/s)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(seq, thr, args.cpu_budget)
    if rank == 0:
        if comm_broken:
            out["communicator"] = "broken during an auxiliary leg (exit code 3); the headline was measured before the legs"
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None and not comm_broken:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:                                  # pragma: no cover
            sys.stderr.write("bench.py: final barrier failed (%r)\n" % (e,))
            comm_broken = True
    if comm_broken:
        sys.stderr.flush()
        os._exit(3)                                             # non-zero: a collective broke (and skip the teardown of a communicator that no longer answers)


if __name__ == "__main__":
    main()
