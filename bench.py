#!/usr/bin/env python
"""bench.py -- SfT Gauss-Newton/LM iterations per second on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment the script
spawns its N ranks itself (torch.distributed.run, one rank per GPU, RCCL) and refuses to run when the node has fewer
than N devices; the driver's own `python -m torch.distributed.run ... bench.py --gpus N` launch works the same way.

One *step* = one pass of the hot path over one batch: every rank solves `--batch` independent single-frame SfT problems
of BASELINE.json configs[1] (500-node template 20x25, 1000 synthetic ORB matches, 640x480 camera) from their uploaded
initial state to LM termination, device-resident (inputs are in HBM before the timed region starts).  The path shards
over independent problems with no data-path collective -> weak scaling.

Everything that is timed runs in the PRODUCT library (libdefslam_hip.so); the kernel duration comes from two HIP events
this script records on the library's launch stream (dsh_stream).  Only the isolated-assembly leg uses the lab build.

Rank 0 prints ONE JSON line; `value` = LM iterations of all ranks / max-over-ranks wall time.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # FP64 vector == matrix peak
PROFILE_ROUND = "r06"


def kernel_source_hash() -> str:
    """Identifies the device code a PMC traffic figure was measured on (profiles/<round>/traffic.json stores it)."""
    import re
    h = hashlib.sha256()
    for name in ("sft_kernels.hip", "sft_wide.h", "sft_wave.h", "sft_batch.h", "tile_chol.h", "sft_problem.h"):
        with open(os.path.join(ROOT, "defslam_amd", "csrc", name), "r", encoding="utf-8") as f:
            src = f.read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)      # comments and layout do not change the device code
        src = re.sub(r"//[^\n]*", "", src)
        h.update(re.sub(r"\s+", " ", src).encode())
    return h.hexdigest()[:16]


def cpu_baseline(tmpl, m, budget_s, gpu_frames=None):
    """The C oracle (a restatement of the reference's g2o path: dense (6+3n)^2 Eigen-style pivoted LDLT, 1 thread) on a bounded sample of
    the same workload.  Timing as BASELINE.md section 3 defines it: the -O3 -march=native build (oracle/_build/libdefslam_oracle_fast.so,
    loaded by nothing else), one C2 problem, 1 warm-up, median of >= 5 runs.  Separately -- never timed as the baseline -- the parity build
    solves ids 0.. of the batch the GPU has just been timed on and the two results are compared (`parity_check`)."""
    import oracle
    from defslam_amd import synth
    tc = oracle.template_build(tmpl.xyz0, tmpl.facets)
    regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)

    def solve(fr, fast, mode=0):
        return oracle.sft_solve(tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz, *regs, max_iters=50, ldlt_mode=mode, fast=fast)

    # ---- timing: one problem (id 0), warm-up + median
    fr0 = synth.make_frame(tmpl, m, 0)
    r = solve(fr0, True)
    ts = []
    t_all = time.perf_counter()
    while len(ts) < 5 or (time.perf_counter() - t_all < budget_s and len(ts) < 20):
        t0 = time.perf_counter()
        r = solve(fr0, True)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    iters, trials, D = r.iters, r.trials, int(r.dims[0])
    cpu = host_cpu()
    # ---- the same restatement on many cores at once (the problems of a batch are independent: one thread per problem, ctypes releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(32, cpu["cores"] or 1))
    frames = [synth.make_frame(tmpl, m, 100 + p) for p in range(T)]
    t1 = time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        it_par = sum(ex.map(lambda fr: solve(fr, True).iters, frames))
    dt_par = time.perf_counter() - t1
    many = {"value": it_par / dt_par, "unit": "iters/s", "cores": T, "problems": T, "seconds": dt_par,
            "what": "one problem per thread, each solved once, all at the same time (fast build)"}
    out = {"value": iters / med, "unit": "iters/s", "cores": 1, "kind": "port", "runs": len(ts), "warmups": 1, "flags": oracle.FAST_FLAGS,
           "seconds_per_solve_median": med, "seconds_per_solve_min": float(min(ts)), "many_cores": many, "host_cpu": cpu["model"],
           "host_cores": cpu["cores"], "host_threads": cpu["threads"],
           "sample": f"problem id 0 of the same workload solved {len(ts)} times after one warm-up, median: {iters} LM iterations, {trials} dense LDLT trials, "
                     f"D={D}; oracle/sft_oracle.c ldlt_mode=0 (Eigen-style pivoted dense LDLT), 1 thread, timing build {oracle.FAST_FLAGS} "
                     f"(reference binary not buildable: Eigen/OpenCV absent)",
           "lm_trials_per_s": trials / med, "frames_per_s": 1.0 / med}
    # ---- the reference's own default problem size (e2e.reference_default): the same frames, the same restatement, one thread
    try:
        rt, rm, rn, rseq = reference_default_frames()
        rtc = oracle.template_build(rt.xyz0, rt.facets)

        def rsolve(fr):
            return oracle.sft_solve(rtc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz, *regs, max_iters=50, ldlt_mode=0, fast=True)
        rf0 = synth.make_frame(rt, rm, 0)
        rr = rsolve(rf0)
        rts = []
        for _ in range(9):
            t0 = time.perf_counter()
            rr = rsolve(rf0)
            rts.append(time.perf_counter() - t0)
        rmed = float(np.median(rts))
        T, x = np.eye(4, dtype=np.float32), rt.xyz0.copy()
        rtot, rit = 0.0, 0
        for k in range(rn):
            fr = synth.make_sequence_frame(rt, rm, k, rn, rseq, init_xyz=x, init_Tcw=T)
            t0 = time.perf_counter()
            ro = rsolve(fr)
            rtot += time.perf_counter() - t0
            rit += ro.iters
            T, x = ro.Tcw, ro.xyz
        out["reference_default"] = {"single_frame": {"ms_per_frame_median": 1e3 * rmed, "frames_per_s": 1.0 / rmed, "iters": int(rr.iters), "trials": int(rr.trials), "runs": len(rts)},
                                    "seq100": {"frames": rn, "frames_per_s": rn / rtot, "ms_per_frame_mean": 1e3 * rtot / rn, "iters_per_frame": rit / rn},
                                    "cores": 1, "kind": "port", "dim": int(rr.dims[0]),
                                    "what": "oracle/sft_oracle.c (dense pivoted LDLT, timing build), one thread, the frames of e2e.reference_default (warm start through its own results)"}
    except Exception as e:  # noqa: BLE001
        out["reference_default"] = {"error": f"{type(e).__name__}: {e}"}
    parity = None
    if gpu_frames:
        # ---- parity of the benched launch: the oracle's PARITY build (-O2 -ffp-contract=off) on the same ids
        ids, it_eq, tr_eq, acc_eq, outl_eq = [], True, True, True, True
        verr = perr = 0.0
        for pid, f in gpu_frames:
            ro = solve(synth.make_frame(tmpl, m, pid), False, mode=1)
            ids.append(int(pid))
            it_eq = it_eq and f.iters == ro.iters
            tr_eq = tr_eq and f.trials == ro.trials
            same_len = f.trace.shape[0] == ro.trace.shape[0]
            acc_eq = acc_eq and same_len and bool(np.array_equal(f.trace[:, 2], ro.trace[:, 2])) and bool(np.array_equal(f.trace[:, 6], ro.trace[:, 6]))
            outl_eq = outl_eq and bool(np.array_equal(f.mvbOutlier, np.asarray(ro.outlier, bool)))
            verr = max(verr, float(np.abs(f.nodes_xyz - ro.xyz).max() / np.abs(ro.xyz).max()))
            perr = max(perr, float(np.abs(f.pose7 - ro.pose7).max()))
        parity = {"ids": ids, "iters_equal": it_eq, "trials_equal": tr_eq, "accept_reject_equal": acc_eq, "outliers_equal": outl_eq,
                  "max_rel_vertex_err": verr, "max_pose_err": perr, "tolerance": {"vertex_rel": 1e-7, "pose": 1e-8, "north_star": 1e-4},
                  "ok": bool(it_eq and tr_eq and acc_eq and outl_eq and verr <= 1e-7 and perr <= 1e-8),
                  "what": "results of the TIMED launch (downloaded behind the timed region) against oracle.sft_solve (parity build, ldlt_mode=1) on the same ids"}
    return out, parity


def host_cpu():
    """Model name and core / thread counts of the box the CPU baseline ran on (from lscpu)."""
    info = {"model": "unknown", "cores": os.cpu_count() or 0, "threads": os.cpu_count() or 0}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {}
        for ln in txt.splitlines():
            if ":" in ln:
                k, v = ln.split(":", 1)
                kv[k.strip()] = v.strip()
        info["model"] = kv.get("Model name", info["model"])
        threads = int(kv.get("CPU(s)", info["threads"]))
        tpc = int(kv.get("Thread(s) per core", "1") or 1)
        info["threads"] = threads
        info["cores"] = threads // max(tpc, 1)
    except Exception:  # noqa: BLE001
        pass
    return info


def flush_c_stdio():
    """RCCL prints a version banner through C stdio when a communicator is created; flushed here it lands BEFORE the JSON line
    (the last line of stdout is the result), not at interpreter exit."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n: int, dry: bool = False) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks here.  Fails loudly when the node has fewer devices
    (unless --dry-ranks asked for a rehearsal of the N-rank path on the devices that exist)."""
    import torch
    have = torch.cuda.device_count()
    if have < 1:
        print("bench.py: no GPU visible", file=sys.stderr)
        return 2
    if have < n and not dry:
        print(f"bench.py: --gpus {n} requested but this node exposes {have} GPU(s); refusing to report an {n}-GPU number from fewer devices", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def rank_plan(gpus, dry_ranks, all_on_device, backend, env, device_count, shared_camera="auto"):
    """Which device this rank computes on and which collective backend carries the barrier -- from the launcher's environment alone, so that
    the rule is testable without a GPU.  Under the driver's `torch.distributed.run --nproc-per-node N bench.py --gpus N` every rank takes
    device LOCAL_RANK (one process per GPU, RCCL); a rank whose device does not exist, or a WORLD_SIZE that contradicts --gpus, is an error."""
    rank, world, local_rank = int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), int(env.get("LOCAL_RANK", "0"))
    if world != gpus:
        return {"error": f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to mislabel the run"}
    ranks_per_device, device = 1, local_rank
    if all_on_device >= 0:
        device, ranks_per_device = all_on_device, world
    elif dry_ranks > 0:
        ndev = max(device_count, 1)
        ranks_per_device = -(-world // ndev)
        device = local_rank % ndev
        if ndev < world and backend == "nccl":
            backend = "gloo"      # RCCL: "duplicate GPU detected" for two ranks of one communicator on one device
    elif device_count <= local_rank:
        return {"error": f"rank {rank} needs GPU {local_rank} but the node exposes {device_count}"}
    # the optional joint-problem leg is a collective of the library's own RCCL communicator: on by default with ONE rank only -- with several, a
    # collective that has never run on the node must not be able to cost the replica line (asked for explicitly it runs under a watchdog)
    want_shared = shared_camera == "on" or (shared_camera == "auto" and world == 1)
    return {"rank": rank, "world": world, "device": device, "ranks_per_device": ranks_per_device, "backend": backend,
            "parallelism": f"{world} x independent problems (no collective)", "shared_camera": bool(want_shared and backend == "nccl")}


def e2e_legs(ctx, tmpl, m, frames, regs):
    """End-to-end frame timing through the one-shot ABI call dsh_sft_solve: pack + upload + run + download + classification
    (SURVEY 8d "frame"), host wall clock around the C call with the C structs prepared beforehand."""
    from defslam_amd import sft, synth
    out = {}
    # (1) one C2 frame, repeated
    call = ctx.prepare_solve(frames[0], *regs, 1, 50)
    call()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    f0 = call.frame
    out["single_frame"] = {"ms_per_frame_median": float(np.median(ts)), "ms_per_frame_min": float(ts.min()), "frames_e2e_per_s": float(1e3 / np.median(ts)),
                           "iters": int(f0.iters), "trials": int(f0.trials), "runs": 20,
                           "what": "dsh_sft_solve wall clock: host packing, one H2D copy, the persistent kernel, one D2H copy of the result slab, device-side classification"}
    # (2) SEQ100: a 100-frame sequence with smooth deformation, every frame warm-started from the previous result
    n_frames = synth.SEQ100["n_frames"]
    T, x = np.eye(4, dtype=np.float32), tmpl.xyz0.copy()
    tot = 0.0
    iters = trials = 0
    errs = []
    for k in range(n_frames):
        fr = synth.make_sequence_frame(tmpl, m, k, n_frames, synth.SEQ100["seq_id"], init_xyz=x, init_Tcw=T)
        call = ctx.prepare_solve(sft.frame_from_synth(fr), *regs, 1, 50)
        t0 = time.perf_counter()
        call()
        tot += time.perf_counter() - t0
        f = call.frame
        iters += f.iters
        trials += f.trials
        errs.append(float(np.abs(f.nodes_xyz - fr.gt_xyz).max()))
        T, x = f.Tcw, f.nodes_xyz
    out["seq100"] = {"frames": n_frames, "frames_e2e_per_s": n_frames / tot, "ms_per_frame_mean": 1e3 * tot / n_frames, "iters_per_frame": iters / n_frames,
                     "trials_per_frame": trials / n_frames, "iters_per_s": iters / tot, "max_vertex_error_vs_gt_last_frame": errs[-1],
                     "what": "synthetic 100-frame sequence (smooth bend + camera loop), warm start frame to frame with the float32 pose round trip; "
                             "sum of dsh_sft_solve wall clocks (frame synthesis excluded)"}
    # (3) SEQMAP: tracking AND mapping interleaved (BASELINE configs[2] substitute, whole): 40 tracked frames on a 168-node template, every 10th
    # frame a keyframe whose mapping work (Schwarp initialisation + search + fit, normals, Shape-from-Normals, registration, new template +
    # embedding) runs between two frames; the frame after it is solved on the new template with RegTemp = 0 (DefTracking.cc:109-115)
    from defslam_amd import seqmap
    def seq_leg(cfg, what):
        seq = synth.make_interleaved_sequence(**cfg)
        legs = {}
        for route in ("device", "host"):
            seqmap.run(ctx, seq, route=route)                   # warm-up (graph cache, scratch)
            st = seqmap.run(ctx, seq, route=route)
            tot = st["t_track"] + st["t_map"]
            legs[route] = {"frames_e2e_per_s": st["frames"] / tot, "ms_tracking_per_frame": 1e3 * st["t_track"] / st["frames"],
                           "ms_mapping_per_keyframe": 1e3 * st["t_map"] / max(st["keyframes"], 1), "iters_per_frame": st["iters"] / st["frames"],
                           "min_inlier_fraction": float(min(st["inliers"])), "frames": st["frames"], "keyframes": st["keyframes"], "templates": st["templates"],
                           "switch_frames": len(st["switch_frames"]), "solves": st["frames"] + len(st["switch_frames"]), "db_records": st.get("db_records")}
        r = dict(legs["device"])
        r["route"] = "device: DiffProp records resident in HBM (dsh_schwarp_fit_batch_store -> dsh_normals_estimate_db -> dsh_sfn_estimate_db)"
        r["host_record_route"] = legs["host"]
        r["template_nodes"] = int(cfg["mesh"][0] * cfg["mesh"][1])
        r["what"] = what
        return r

    out["seq_mapping"] = seq_leg(dict(synth.SEQMAP),
                                 "synth.SEQMAP: wall clock inside the C-ABI calls of defslam_amd/seqmap.py (tracking every frame, the whole mapping chain every "
                                 "10th frame, template switch on the next one -- that frame is solved twice like DefTracking.cc:109-123 + :244-247: RegTemp = 0 first, then the "
                                 "regular solve without the observations the first one flagged); synthetic data generation excluded; tests/test_seqmap_gpu.py checks every "
                                 "stage of this loop against its oracle and the two record routes against each other (bit-identical)")
    c2 = dict(synth.SEQMAP)
    c2["mesh"] = (20, 25)
    out["seq_mapping_c2_template"] = seq_leg(c2, "the same sequence on the 500-node template of BASELINE configs[1] (20 x 25 grid)")
    out["reference_default"] = reference_default_leg(ctx, regs)
    return out


def reference_default_frames():
    """The problem the reference's one performance claim is about ("real-time ... i7", README.md:4,30): the hard-coded 10 x 10 = 100-node template
    (Modules/Template/TriangularMesh.cc:63-64), 1200 ORB features per frame (scripts/stereo0_template.yaml) of which 450 are matched to the
    template here, the Mandala regularisers: D = 306.  One frame from rest, and a 100-frame sequence (smooth bend + camera loop)."""
    from defslam_amd import synth
    rows, cols, m = synth.CONFIGS["REF"]
    tmpl = synth.make_grid_template(rows, cols)
    return tmpl, m, 100, 3


def reference_default_leg(ctx, regs):
    """GPU side of `e2e.reference_default`: dsh_sft_solve wall clock (host buffers in and out), one frame repeated and the warm-started sequence;
    the CPU restatement on the same frames is timed by cpu_baseline() and attached as `cpu_restatement`."""
    from defslam_amd import sft, synth
    tmpl, m, n_frames, seq_id = reference_default_frames()
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    call = ctx.prepare_solve(sft.frame_from_synth(synth.make_frame(tmpl, m, 0)), *regs, 1, 50)
    call()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    f0 = call.frame
    T, x = np.eye(4, dtype=np.float32), tmpl.xyz0.copy()
    tot, iters, trials, last = 0.0, 0, 0, None
    for k in range(n_frames):
        fr = synth.make_sequence_frame(tmpl, m, k, n_frames, seq_id, init_xyz=x, init_Tcw=T)
        call = ctx.prepare_solve(sft.frame_from_synth(fr), *regs, 1, 50)
        t0 = time.perf_counter()
        call()
        tot += time.perf_counter() - t0
        f = call.frame
        iters += f.iters
        trials += f.trials
        last = float(np.abs(f.nodes_xyz - fr.gt_xyz).max())
        T, x = f.Tcw, f.nodes_xyz
    return {"template_nodes": tmpl.n, "matches": m, "n_frame_keypoints": synth.N_FRAME_KEYPOINTS, "dim": int(f0.dim), "half_bandwidth": int(f0.half_bandwidth),
            "single_frame": {"ms_per_frame_median": 1e3 * med, "frames_per_s": 1.0 / med, "iters": int(f0.iters), "trials": int(f0.trials), "iters_per_s": f0.iters / med, "runs": 30},
            "seq100": {"frames": n_frames, "frames_per_s": n_frames / tot, "ms_per_frame_mean": 1e3 * tot / n_frames, "iters_per_frame": iters / n_frames,
                       "trials_per_frame": trials / n_frames, "max_vertex_error_vs_gt_last_frame": last},
            "what": "the reference's own default problem size (10 x 10 template, TriangularMesh.cc:63-64; 1200 features per frame, scripts/stereo0_template.yaml; "
                    "450 matches; Mandala regularisers): dsh_sft_solve wall clock, host buffers in and out -- one frame from rest repeated (median of 30), and a "
                    "100-frame synthetic sequence warm-started frame to frame (sum of the call times; frame synthesis excluded).  One problem on one GPU: "
                    "the latency mode (speculative damping lanes), where a 306 x 306 dense LDLT on a CPU core is cheap -- the least flattering comparison"}


def shared_camera_leg(ctx, tmpl, m, regs, rank, world, dist, torch):
    """One JOINT Shape-from-Template problem over all ranks: every rank holds one C2-sized patch (its own 500-node template and 1000
    matches), all patches are seen by one camera; per damping trial the ranks all-reduce their 6x6 Schur complement of the camera
    (dsh_sft_shared_solve: RCCL, 32 doubles, three times per trial).  Wall clock of the collective call, max over ranks."""
    from defslam_amd import sft, synth
    gt = synth.sequence_gt_pose(7, 100)             # the same ground-truth camera for every patch
    fr = synth.make_frame(tmpl, m, 5000 + rank, gt_pose=gt)
    uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
    if rank == 0:
        uid = torch.frombuffer(bytearray(sft.comm_unique_id()), dtype=torch.uint8).cuda()
    if dist is not None:
        dist.broadcast(uid, 0)
    comm = sft.Comm(ctx, world, rank, bytes(uid.cpu().numpy().tobytes()))
    flush_c_stdio()
    try:
        ts = []
        f = None
        for _ in range(4):
            f = sft.frame_from_synth(fr)
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sft.SharedCameraPoseOptimization(ctx, comm, f, *regs)
            ts.append(time.perf_counter() - t0)
        t = torch.tensor([min(ts[1:])], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finally:
        comm.close()
    return {"ranks": world, "patches": world, "nodes_total": world * tmpl.n, "matches_total": world * m, "iters": int(f.iters), "trials": int(f.trials),
            "ms_per_joint_frame": 1e3 * dt, "joint_iters_per_s": f.iters / dt, "collectives_per_trial": 3, "doubles_per_collective": 32,
            "what": "optional mode (SURVEY 8e): one joint problem, one patch per rank, shared camera; host-sequenced phase kernels with an RCCL "
                    "all-reduce of the camera block between them -- latency-bound by design, the default for independent problems is the replica batch above"}


def connected_leg(local_rank, tmpl, fr, regs):
    """ONE connected template cut across two ranks (dsh_sft_connected_solve_group: two contexts on this GPU, the two all-reduces per damping
    trial done by a summation kernel; between two GPUs the same driver runs them over RCCL): wall clock of the call, best of three."""
    from defslam_amd import sft
    ctxs = [sft.Context(local_rank), sft.Context(local_rank)]
    try:
        for c in ctxs:
            c.template_build(tmpl.xyz0, tmpl.facets)
        ts = []
        fs = None
        for _ in range(3):
            fs = [sft.frame_from_synth(fr), sft.frame_from_synth(fr)]
            t0 = time.perf_counter()
            sft.ConnectedPoseOptimizationGroup(ctxs[0], ctxs[1], fs, *regs)
            ts.append(time.perf_counter() - t0)
        _, counts = ctxs[0].problem_info(0)
        cut = sft.two_sided_cut(int(counts[5]) - 6, int(counts[6]))
        dt = min(ts)
        return {"ranks": 2, "devices": 1, "iters": int(fs[0].iters), "trials": int(fs[0].trials), "ms_per_frame": 1e3 * dt, "iters_per_s": fs[0].iters / dt,
                "separator_scalars": cut[1], "part0_scalars": cut[0], "part1_scalars": cut[2] - cut[3],
                "doubles_per_trial": {"schur_allreduce": (cut[1] // 16) ** 2 * 256 + 8 * cut[1] + 64, "update_allreduce": int(counts[5])},
                "what": "optional mode (SURVEY 8e, connected mesh): one problem, the band cut at a separator of one bandwidth, one part per rank, host-sequenced "
                        "phase kernels; here both ranks share this GPU and the all-reduce is a kernel -- latency-bound by design"}
    finally:
        for c in ctxs:
            c.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="independent problems per GPU per step (default: 16384 for C2 = 64 per CU, 42 GB of HBM: problems need 7-35 damping "
                                                         "trials and a launch ends with its slowest one, so a deep batch amortises the tail; 16 for C5 = BASELINE configs[4])")
    ap.add_argument("--config", default="C2", choices=["smoke", "C2", "C5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip latency / end-to-end / isolated-assembly legs (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank); gloo only to exercise the multi-rank path on a single GPU")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1, help="testing aid: every rank uses this device index instead of LOCAL_RANK")
    ap.add_argument("--dry-ranks", type=int, default=0, help="rehearsal of the N-rank launch on however many GPUs exist: N ranks, rank r on device r mod #GPUs, the real "
                                                             "nccl (RCCL) backend when every rank has its own GPU and gloo otherwise (RCCL refuses two ranks of one communicator "
                                                             "on one device); a per-rank memory guard shrinks the batch to what the shared device holds; the line says dry_ranks")
    ap.add_argument("--shared-camera", default="auto", choices=["auto", "on", "off"], help="the optional joint-problem leg (library-owned RCCL communicator): auto = on with one "
                                                                                            "rank, off with several (a collective that has not run on this node yet must not cost the replica line)")
    args = ap.parse_args()
    if args.dry_ranks > 0:
        args.gpus = args.dry_ranks
    if args.batch <= 0:
        args.batch = {"C2": 16384, "C5": 16, "smoke": 512}[args.config]
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, dry=args.dry_ranks > 0))

    import torch
    from defslam_amd import sft, synth

    plan = rank_plan(args.gpus, args.dry_ranks, args.all_ranks_on_device, args.dist_backend, os.environ, torch.cuda.device_count(), args.shared_camera)
    if "error" in plan:
        print("bench.py: " + plan["error"], file=sys.stderr)
        sys.exit(2)
    rank, world, local_rank, ranks_per_device = plan["rank"], plan["world"], plan["device"], plan["ranks_per_device"]
    args.dist_backend = plan["backend"]
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)
        flush_c_stdio()
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    # per-rank memory guard: the batch arena is one hipMalloc (C2: 2.6 MB per problem = 42 GB at 16384); ranks that share a device share its HBM
    batch_asked = args.batch
    if torch.cuda.is_available():
        free_b, _ = torch.cuda.mem_get_info(local_rank)
        per_problem = {"C2": 2.7e6, "C5": 60e6, "smoke": 0.5e6}[args.config]
        fit = int(0.8 * free_b / ranks_per_device / per_problem)
        if fit < args.batch:
            args.batch = max(1, fit)
    rows, cols, m = synth.CONFIGS[args.config]
    regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    tmpl = synth.make_grid_template(rows, cols)
    ctx = sft.Context(local_rank)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    # problems are sharded over ranks by id: rank r owns ids r*B .. r*B+B-1 (no data-path collective)
    frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, rank * args.batch + p)) for p in range(args.batch)]
    ctx.batch_upload(frames, *regs, 1, 50)

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.batch_run()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = ctx.batch_run_timed(args.steps)     # K launches bracketed by this script's HIP events on the library's launch stream
    barrier()
    wall = time.perf_counter() - t0
    iters, trials = ctx.batch_counts()              # per step (every step restarts from the uploaded state)
    parity_ids = [i for i in (0, 1, args.batch // 2, args.batch - 1) if 0 <= i < args.batch]
    parity_ids = sorted(set(parity_ids)) if (rank == 0 and world == 1 and not args.no_cpu_baseline and args.config != "C5") else []
    gpu_sample = []
    if parity_ids:                                  # the results of the last timed launch, before any other leg re-uploads
        ctx.batch_download(only=parity_ids)         # writes the results into the Frame objects (the reference mutates its frame in place) ...
        gpu_sample = [(rank * args.batch + i, frames[i]) for i in parity_ids]
        for i in parity_ids:                        # ... so the legs below get fresh inputs for these ids
            frames[i] = sft.frame_from_synth(synth.make_frame(tmpl, m, rank * args.batch + i))
    infos = [ctx.problem_info(b) for b in range(args.batch)]
    alg_bytes = sum(i[0] for i in infos)
    _, counts = infos[0]

    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    wall_t = torch.tensor([wall], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([iters, trials, args.batch], dtype=torch.float64, device=red_dev)
    devs = torch.zeros(max(world, 1), dtype=torch.float64, device=red_dev)
    devs[rank] = 1.0 + local_rank                   # which device every rank computed on
    rank_ms = torch.zeros(max(world, 1), dtype=torch.float64, device=red_dev)
    rank_ms[rank] = 1e3 * wall / args.steps         # every rank's own wall clock per step
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(devs, op=dist.ReduceOp.SUM)
        dist.all_reduce(rank_ms, op=dist.ReduceOp.SUM)
    wall = float(wall_t.item())
    g_iters, g_trials, g_problems = (float(v) for v in tot.tolist())
    n_gpus = len(set(int(v) for v in devs.tolist()))   # distinct devices behind the ranks the collective saw

    # ---- optional mode of the north star, measured next to the replicas: ONE joint problem, one patch per rank, shared camera,
    # RCCL all-reduce of the camera block (collective: every rank takes part)
    shared = None
    shared_hung = False
    if plan["shared_camera"] and not args.no_extra_legs and args.config == "C2":
        # The leg is a collective of the library's own RCCL communicator: with several ranks it runs under a watchdog, so that a rank
        # that fails (or a communicator that cannot be formed on this node) costs this optional object, never the replica line above.
        box = {}

        def run_leg():
            try:
                torch.cuda.set_device(local_rank)   # the current device is per THREAD (default 0): without this every rank's tensors land on GPU 0
                box["r"] = shared_camera_leg(ctx, tmpl, m, regs, rank, world, dist, torch)
            except Exception as e:  # noqa: BLE001
                box["r"] = {"error": f"{type(e).__name__}: {e}"}

        if world > 1:
            import threading
            th = threading.Thread(target=run_leg, daemon=True)
            th.start()
            th.join(timeout=120.0)
            if th.is_alive():
                box["r"] = {"error": "timeout: the shared-camera collective did not finish within 120 s on this rank"}
                shared_hung = True
            # every rank learns whether ANY rank hung -- through the rendezvous store (TCP), not through a collective that may sit
            # behind the hung one -- so that all ranks take the same exit below
            try:
                store = dist.distributed_c10d._get_default_store()
                store.set(f"bench_shared_hung_{rank}", "1" if shared_hung else "0")
                for r in range(world):
                    key = f"bench_shared_hung_{r}"
                    store.wait([key], __import__("datetime").timedelta(seconds=150))
                    if store.get(key) == b"1":
                        shared_hung = True
            except Exception:  # noqa: BLE001  (a rank that never reports is a hung rank)
                shared_hung = True
        else:
            run_leg()
        shared = box.get("r")
        if not shared_hung:
            ctx.batch_upload(frames, *regs, 1, 50)      # the legs below expect the replica batch
            ctx.batch_run()
            ctx.synchronize()

    if rank == 0:
        # HBM traffic per launch: rocprofv3 PMC passes of exactly this configuration, carried with their provenance and dropped when the
        # device code has changed since (profiles/<round>/traffic.json; tools/profile_bench.sh regenerates it)
        traffic = traffic_asm = None
        traffic_phase = {}
        traffic_source = "none: no PMC pass recorded for this configuration"
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic.json")))
            key = f"{args.config}_B{args.batch}"
            src_hash = kernel_source_hash()
            if tj.get("kernel_source_hash") != src_hash:
                traffic_source = (f"stale: profiles/{PROFILE_ROUND}/traffic.json was measured on device code {tj.get('kernel_source_hash')}, this build is {src_hash}")
            else:
                if key in tj:
                    traffic = tj[key]["bytes_per_launch"]
                if key + "_assembly" in tj:
                    traffic_asm = tj[key + "_assembly"]["bytes_per_launch"]
                # throughput shape: bytes per STEP of each phase kernel (all of its launches of a step summed)
                traffic_phase = {ph: tj[f"{key}_{ph}"]["bytes_per_step"] for ph in ("factor", "lin", "trial") if f"{key}_{ph}" in tj}
                traffic_source = f"profiles/{PROFILE_ROUND}/traffic.json (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes; device code {src_hash}, commit {tj.get('commit')})"
        except Exception as e:  # noqa: BLE001
            traffic_source = f"none: {type(e).__name__}"
        ms_per_step = 1e3 * wall / args.steps
        num_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        value = g_iters * args.steps / wall
        kern_ms = kernel_ms / args.steps            # avg duration of the persistent kernel (rank 0)
        # The one persistent kernel is ~70% banded-arrowhead Cholesky (FP64 MFMA) and ~15% assembly (HBM-bound in principle,
        # SURVEY 8d "state both fractions").  roofline = the dominant part: algorithmic solve flops per launch
        # (SURVEY 8d block-banded convention D*beta^2, with the half-bandwidth this ordering actually has, plus the 7 border
        # rows and the two triangular solves; padding and zero tiles the MFMAs also execute are NOT counted) over the kernel time.
        Dn, kd = int(counts[5]) - 6, int(counts[6])
        flops_trial = Dn * kd * kd + 2 * 7 * Dn * kd + 4 * Dn * kd + 4 * 7 * Dn
        flops_per_launch = flops_trial * (trials / 1.0)
        achieved_tf = flops_per_launch / (kern_ms * 1e-3) / 1e12
        # assembly: algorithmic bytes per launch (SURVEY 8d per-iteration figure x iterations the launch executes)
        bytes_per_launch = (alg_bytes / args.batch) * iters
        nTt = ((Dn + 31) // 32) * 2                      # 16-row tile rows of the padded node block
        # per damping trial: H read once (register-window solver: compact 3x3 blocks; wider bands: tiles), L written + read, border rows, Linv
        h_read = 72 * (int(counts[1]) + int(counts[8])) if kd <= 128 else nTt * (-(-kd // 16) + 1) * 2048
        stream_trial = h_read + 2 * nTt * 8 * 2048 + 3 * 7 * 16 * nTt * 8 + 2 * nTt * 2048
        stream_bytes = stream_trial * trials
        hbm_gbs = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "SfT GN iters/sec (500-node mesh, 1k matches)", "value": value, "unit": "iters/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: single-frame SfT, {rows * cols}-node template ({rows}x{cols}), {m} matches, 640x480",
                       "problems_per_gpu": args.batch, "parallelism": plan["parallelism"],
                       "max_lm_iters": 50, "regularisers": list(regs), "wavefronts_per_problem": int(counts[7]), "ranks": world,
                       "ranks_per_device": ranks_per_device, "dist_backend": args.dist_backend if world > 1 else None,
                       "dry_ranks": args.dry_ranks > 0, "problems_per_gpu_asked": batch_asked},
            "rank_ms_per_step": [float(v) for v in rank_ms.tolist()],
            "frames_per_s": g_problems * args.steps / wall,
            "lm_trials_per_s": g_trials * args.steps / wall,
            "iters_per_frame": g_iters / g_problems,
            "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / FP64_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": ("all launches of a step (sftb_lin / sftb_factor / sftb_trial kernels)" if int(counts[7]) == 1 else
                                    ("sft_lm_kernel<4>" if int(counts[7]) == 4 else
                                     # eight wavefronts per problem: the persistent kernel above half a problem per CU, the latency mode below
                                     ("sft_spec_kernel<8> (latency mode: every launch of the step)" if 2 * args.batch <= num_cus else "sft_lm_kernel<8>"))), "kernel_ms": kern_ms,
                         "timing": "HIP events recorded by bench.py on dsh_stream() around the K steps of the product library",
                         "algorithmic_flops_per_launch": flops_per_launch, "flops_per_lm_trial": flops_trial, "dim": Dn + 6, "half_bandwidth": kd,
                         "hbm_assembly": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                                          "algorithmic_bytes_per_launch": bytes_per_launch,
                                          "measured_traffic_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None},
                         # what THIS algorithm has to stream per launch: assembly bytes + per damping trial the H tiles read once,
                         # L written once by the factorisation and read once by the back substitution
                         "hbm_solver_stream": {"achieved": (bytes_per_launch + stream_bytes) / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": (bytes_per_launch + stream_bytes) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "bytes_per_trial": stream_trial, "achievable_stream_GBps_this_box": 5400.0},
                         "note": "one persistent kernel = residuals + Jacobian assembly + banded-arrowhead Cholesky (FP64 MFMA) + LM control; "
                                 "frac = algorithmic solve flops / FP64 peak over the WHOLE kernel time; hbm_assembly = SURVEY 8d assembly bytes over the same time"},
        }
        if shared is not None:
            out["shared_camera"] = shared
        curve = None
        if not args.no_extra_legs and not shared_hung and int(counts[7]) == 1 and world == 1:
            # The throughput claim at the batch sizes a multi-template / multi-keyframe user would actually have: the PRODUCT library on prefixes of
            # the same batch, HIP events on its stream (3 steps after a warm-up step); the rounds per step are added from the lab build below
            curve = {"what": "product library, prefixes of the benched batch, HIP events on dsh_stream(), 3 timed steps after one warm-up; rounds_per_step = "
                             "rounds of LIN / FACTOR / TRIAL launches one step takes (from the lab build's per-launch events on the same batches)", "points": []}
            for Bc in [b for b in (1, 8, 32, 128, 256, 512, 1024, 2048, 4096) if b < args.batch] + [args.batch]:
                ctx.batch_upload(frames[:Bc], *regs, 1, 50)
                ctx.batch_run()
                ctx.synchronize()
                nrep = 5 if Bc <= 512 else 3
                ms_c = ctx.batch_run_timed(nrep) / nrep
                it_c, tr_c = ctx.batch_counts()
                _, cc = ctx.problem_info(0)
                # launch shape (include/defslam_hip.h): latency mode with K speculative lanes below half a problem per CU, then the throughput shape
                # (tail kernel alone up to two problems per CU, rounds + tail above)
                shape = (f"latency mode, {min(4, num_cus // Bc)} lanes per problem" if 2 * Bc <= num_cus else
                         ("tail kernel alone" if int(cc[7]) == 1 and Bc <= 2 * num_cus else ("rounds + tail kernel" if int(cc[7]) == 1 else f"persistent kernel, {int(cc[7])} wavefronts per problem")))
                curve["points"].append({"problems": Bc, "iters_per_s": it_c / (ms_c * 1e-3), "ms_per_step": ms_c, "us_per_problem": 1e3 * ms_c / Bc, "iters": int(it_c), "trials": int(tr_c),
                                        "wavefronts_per_problem": int(cc[7]), "shape": shape, "rounds_per_step": None})
            full = curve["points"][-1]["iters_per_s"]
            for pt in curve["points"]:
                pt["fraction_of_full_batch_rate"] = pt["iters_per_s"] / full
            rates = [pt["iters_per_s"] for pt in curve["points"]]
            curve["monotone"] = bool(all(b >= a for a, b in zip(rates, rates[1:])))
            out["batch_curve"] = curve
        if not args.no_extra_legs and not shared_hung:   # (a context stuck in a hung collective cannot run the other legs)
            # single-problem latency leg (the >=200 iters/s target of BASELINE.json is for ONE problem on one GPU)
            ctx.batch_upload(frames[:1], *regs, 1, 50)
            ctx.batch_run()
            ctx.synchronize()
            ms1 = ctx.batch_run_timed(5) / 5
            it1, tr1 = ctx.batch_counts()
            out["latency"] = {"single_problem_iters_per_s": it1 / (ms1 * 1e-3), "ms_per_frame": ms1, "iters": it1, "trials": tr1,
                              "what": "device timeline of the latency mode, inputs resident (HIP events on the library's stream): the speculative-lane launches "
                                      "INCLUDING the host's read-backs of the done flags between groups of launches; the end-to-end frame is in `e2e`"}
            if args.config == "C2":
                out["e2e"] = e2e_legs(ctx, tmpl, m, frames, regs)
            if world == 1:
                try:
                    out["connected_mesh"] = connected_leg(local_rank, tmpl, synth.make_frame(tmpl, m, 0), regs)
                except Exception as e:  # noqa: BLE001
                    out["connected_mesh"] = {"error": f"{type(e).__name__}: {e}"}
            # the Jacobian assembly on its own: launches that do one linearisation + normal-equation assembly per problem (measurement
            # kernel of the LAB build, same device functions as the product kernel), rank 0's GPU
            ctx.close()
            lab = sft.Context(local_rank, lab=True)
            lab.template_build(tmpl.xyz0, tmpl.facets)
            if curve is not None:   # rounds per step of the smaller batches of the curve
                for pt in curve["points"][:-1]:
                    if pt["wavefronts_per_problem"] != 1 or pt["problems"] <= 2 * num_cus:
                        continue
                    lab.batch_upload(frames[:pt["problems"]], *regs, 1, 50)
                    lab.batch_run()
                    lab.synchronize()
                    pt["rounds_per_step"] = lab.rounds_timed()[1]
            lab.batch_upload(frames, *regs, 1, 50)
            lab.batch_run()
            lab.synchronize()
            if int(counts[7]) == 1:
                # Throughput shape = rounds of phase kernels: the step's time by kernel, from per-launch HIP events of the lab build (the same
                # device code; the product library has no timing entry points).  The dominant kernel is the one-wavefront factorisation
                # (FP64 roofline); the Jacobian assembly is its own kernel now, so its HBM roofline is a production number.
                ph, n_rounds = lab.rounds_timed()
                if curve is not None:
                    curve["points"][-1]["rounds_per_step"] = n_rounds
                rf = out["roofline"]
                # the damping trials the FACTOR launches factored (the last problems of a step are run to their end by sftb_tail_kernel: not counted here)
                n_fact = ph.pop("factorisations_in_rounds")
                n_lin = ph.pop("linearisations_in_rounds")
                flops_factor = flops_trial * float(n_fact)
                tf = flops_factor / (ph["factor"] * 1e-3) / 1e12
                rf.update({"kernel": "sftb_factor_kernel", "kernel_ms": ph["factor"], "achieved": tf, "frac": tf / FP64_PEAK_TFLOPS,
                           "traffic": traffic_phase.get("factor"), "traffic_GBps": (traffic_phase["factor"] / (ph["factor"] * 1e-3) / 1e9) if "factor" in traffic_phase else None,
                           "traffic_what": "HBM bytes of the kernel's launches of one step (PMC)",
                           "timing": "HIP events in front of and behind every launch of one step (dsh_lab_sft_rounds_timed, libdefslam_hip_lab.so: the device code of "
                                     "the timed product run); kernel_ms = the sum over the step's sftb_factor_kernel launches",
                           "phases_ms": ph, "rounds_per_step": n_rounds, "phases_sum_over_step": sum(ph.values()) / ms_per_step,
                           "algorithmic_flops_per_launch": flops_factor, "factorisations_in_rounds": int(n_fact), "damping_trials_per_step": int(trials),
                           "note": "a step = rounds of LIN (residuals + Jacobian assembly), FACTOR (banded-arrowhead Cholesky + back substitution, one wavefront per "
                                   "problem, FP64 MFMA) and TRIAL (update, chi2, LM control) launches, then ONE launch of the tail kernel for the last problems (two per CU "
                                   "and fewer: phases_ms.tail); frac = algorithmic solve flops / FP64 peak over the FACTOR "
                                   "kernel's own time; hbm_assembly = SURVEY 8d assembly bytes over the LIN kernel's own time"})
                bytes_lin = (alg_bytes / args.batch) * float(n_lin)     # the linearisations the LIN launches performed (the last problems': tail kernel)
                lin_gbs = bytes_lin / (ph["lin"] * 1e-3) / 1e9
                rf["hbm_assembly"] = {"kernel": "sftb_lin_kernel", "kernel_ms": ph["lin"], "achieved": lin_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": lin_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_lin, "linearisations_in_rounds": int(n_lin), "linearisations_per_step": int(iters),
                                      "traffic": traffic_phase.get("lin"), "traffic_over_algorithmic": (traffic_phase["lin"] / bytes_lin) if "lin" in traffic_phase else None,
                                      "what": "every linearisation of a step (residuals + records + normal equations of the problems that start an iteration), "
                                              "algorithmic bytes over the kernel's own time"}
                out["hbm_assembly"] = {k: rf["hbm_assembly"][k] for k in ("kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "linearisations_in_rounds")}   # (the production number at the top level of the line as well)
                fs = (stream_trial * float(n_fact)) / (ph["factor"] * 1e-3) / 1e9
                rf["hbm_solver_stream"] = {"kernel": "sftb_factor_kernel", "achieved": fs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fs / HBM_PEAK_GBS,
                                           "bytes_per_trial": stream_trial, "achievable_stream_GBps_this_box": 5400.0,
                                           "what": "per damping trial the compact H blocks read once, L written once and read once by the deferred back substitution"}
            asm_ms = lab.batch_assemble_timed(5) / 5
            lab.close()
            out["roofline"]["hbm_assembly_isolated"] = {
                "achieved": alg_bytes / (asm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (asm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "ms_per_pass": asm_ms, "algorithmic_bytes_per_pass": alg_bytes, "traffic": traffic_asm, "traffic_source": traffic_source,
                "measured_traffic_GBps": (traffic_asm / (asm_ms * 1e-3) / 1e9) if traffic_asm else None,
                "library": "libdefslam_hip_lab.so (dsh_lab_sft_assemble_timed -> sft_assembly_kernel)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], parity = cpu_baseline(tmpl, m, args.cpu_seconds, gpu_sample)
            if parity is not None:
                out["parity_check"] = parity
            ref_cpu = out["cpu_baseline"].pop("reference_default", None)
            if ref_cpu is not None and "reference_default" in out.get("e2e", {}):
                rd = out["e2e"]["reference_default"]
                rd["cpu_restatement"] = ref_cpu
                if "single_frame" in ref_cpu:
                    rd["gpu_over_cpu"] = {"single_frame": rd["single_frame"]["frames_per_s"] / ref_cpu["single_frame"]["frames_per_s"],
                                          "seq100": rd["seq100"]["frames_per_s"] / ref_cpu["seq100"]["frames_per_s"]}
        flush_c_stdio()
        print(json.dumps(out), flush=True)
    if shared_hung:
        os._exit(0)   # the line is out; a thread is still inside the hung collective, a clean shutdown would wait for it
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
