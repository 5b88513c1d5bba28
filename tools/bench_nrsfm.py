#!/usr/bin/env python
"""Mapping-side (NRSfM) measurements: BBS evaluation, per-map-point normal solve, Schwarzian warp fit.

The C ABI of these calls hands over HOST buffers (like the reference's call sites do), so the wall-clock rates printed
here include the PCIe copies; run it under `rocprofv3 --kernel-trace --stats` (tools/profile_nrsfm.sh) for the kernel
durations the roofline figures in profiles/README.md are computed from.  One JSON line per measurement.
CPU legs: oracle/_ref/libbbs_ref.so is the reference's own bbs.cc ("reference"), the others are the C restatement ("port")."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=1 << 22)
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--matches", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    from defslam_amd import sft, nrsfm, synth
    ctx = sft.Context(0)
    rng = np.random.default_rng(5)
    out = []

    # ---- BBS::eval (Thirdparty/BBS/bbs.cc:243-390): warp evaluation at key points, 13 x 15 control grid, valdim 2 ----
    b = nrsfm.Bbs(-0.7, 0.7, 13, -0.55, 0.55, 15, 2)
    ctrl = rng.normal(size=(2, 13 * 15))
    for n in (args.matches, args.sites):
        u = rng.uniform(-0.69, 0.69, n)
        v = rng.uniform(-0.54, 0.54, n)
        dt = timeit(lambda: nrsfm.bbs_eval(ctx, b, ctrl, u, v, 0, 0), args.reps)
        rec = {"metric": "BBS eval sites/s (13x15 grid, valdim 2, host buffers in/out)", "value": n / dt, "unit": "sites/s", "sites": n, "ms_per_call": 1e3 * dt,
               "algorithmic_bytes_per_site": 16 + 16 + 1}
        if not args.no_cpu:
            import oracle
            bb = (b.umin, b.umax, b.nptsu, b.vmin, b.vmax, b.nptsv, b.valdim)
            m = min(n, 1 << 20)
            dtc = timeit(lambda: oracle.ref_bbs_eval(bb, ctrl, u[:m], v[:m], 0, 0), 2)
            rec["cpu_baseline"] = {"value": m / dtc, "unit": "sites/s", "cores": 1, "kind": "reference", "sample": f"{m} sites, oracle/_ref/libbbs_ref.so (reference bbs.cc, OpenMP disabled)"}
        out.append(rec)

    # ---- NormalEstimator::ObtainK1K2 (Modules/Mapping/NormalEstimator.cc:60-260): map points x views ----
    for P in (2000, args.points):
        sc = synth.make_normals_scene(n_points=P, n_views=4, seed=11)
        a = (sc["rec_ptr"], sc["recs"], sc["rec_is_ref"], sc["rec_first_normal"], sc["rec_has_first_normal"], sc["x0"], sc["has_x0"], sc["ref_uv"])
        dt = timeit(lambda: nrsfm.ObtainK1K2(ctx, *a), args.reps)
        R = sc["recs"].shape[0]
        rec = {"metric": "NRSfM normal solve map points/s (<=4 views per point, host buffers in/out)", "value": P / dt, "unit": "points/s", "points": P, "records": int(R),
               "ms_per_call": 1e3 * dt, "algorithmic_bytes_per_record": 18 * 4 + 1 + 8 + 1 + 12 + 1}
        if not args.no_cpu:
            import oracle
            m = min(P, 20000)
            sub = synth.make_normals_scene(n_points=m, n_views=4, seed=11) if m != P else sc
            a2 = (sub["rec_ptr"], sub["recs"], sub["rec_is_ref"], sub["rec_first_normal"], sub["rec_has_first_normal"], sub["x0"], sub["has_x0"], sub["ref_uv"])
            dtc = timeit(lambda: oracle.normals(*a2), 1)
            rec["cpu_baseline"] = {"value": m / dtc, "unit": "points/s", "cores": 1, "kind": "port", "sample": f"{m} points, oracle/nrsfm_oracle.c (Ceres-style LM restated; parity unpinned)"}
        out.append(rec)

        # the same records held in HBM (dsh_diffdb): the call ships key points in, normals out
        owner = np.repeat(np.arange(P, dtype=np.int32), np.diff(sc["rec_ptr"]))
        db = nrsfm.DiffDatabase(ctx, R)
        db.append(sc["recs"], owner)
        ids = np.arange(P, dtype=np.int32)
        for per_record in (False, True):
            dtd = timeit(lambda: nrsfm.ObtainK1K2Database(ctx, db, ids, sc["x0"], sc["has_x0"], sc["ref_uv"], per_record=per_record), args.reps)
            out.append({"metric": "NRSfM normal solve map points/s (records resident in HBM: dsh_normals_estimate_db" + (", per-record normals returned)" if per_record else ")"),
                        "value": P / dtd, "unit": "points/s", "points": P, "records": int(R), "ms_per_call": 1e3 * dtd})
        db.close()

    # ---- SchwarpDatabase::calculateSchwarps (Modules/Mapping/SchwarpDatabase.cc:246-340): one keyframe pair ----
    wp = synth.make_warp_problem(n_matches=args.matches, seed=3)
    wb = nrsfm.Bbs(*wp["bbs"])
    fit = lambda: nrsfm.calculateSchwarps(ctx, wb, wp["kp1"], wp["kp2"], wp["invsig"], wp["fy"], wp["fx"], 1e-2, wp["fx"], wp["fy"], wp["x0"], 3)
    dt = timeit(fit, args.reps)
    rec = {"metric": "Schwarp fit keyframe pairs/s (13x15 grid, 3 LM iterations)", "value": 1.0 / dt, "unit": "fits/s", "matches": args.matches, "ms_per_call": 1e3 * dt}
    if not args.no_cpu:
        import oracle
        dtc = timeit(lambda: oracle.schwarp_fit(wp["bbs"], wp["kp1"], wp["kp2"], wp["invsig"], wp["fy"], wp["fx"], 1e-2, wp["fx"], wp["fy"], wp["x0"], 3), 1)
        rec["cpu_baseline"] = {"value": 1.0 / dtc, "unit": "fits/s", "cores": 1, "kind": "port", "sample": "same problem, oracle/schwarp_oracle.c (dense normal equations, 1 thread; parity unpinned)"}
    out.append(rec)
    # ---- the same fit for B keyframe pairs per call (SchwarpDatabase::add: one warp per anchor keyframe) ----
    for B in (8, 64):
        probs = []
        for b in range(B):
            q = synth.make_warp_problem(n_matches=args.matches, seed=100 + b)
            probs.append(dict(bbs=nrsfm.Bbs(*q["bbs"]), kp1=q["kp1"], kp2=q["kp2"], invsig=q["invsig"], fx_slot=q["fy"], fy_slot=q["fx"], lam=1e-2, fx=q["fx"], fy=q["fy"], x0=q["x0"]))
        dtb = timeit(lambda: nrsfm.calculateSchwarpsBatch(ctx, probs, 3), max(2, args.reps // 4))
        out.append({"metric": "Schwarp fit keyframe pairs/s, batched (13x15 grid, 3 LM iterations, host buffers in/out)", "value": B / dtb, "unit": "fits/s", "pairs_per_call": B,
                    "matches": args.matches, "ms_per_call": 1e3 * dtb})
    # the same with Warp::initialize inside the call (dsh_schwarp_problem.init_lambda) against one dsh_warp_initialize per pair before it
    dt_init = timeit(lambda: nrsfm.WarpInitialize(ctx, wb, wp["kp1"], wp["kp2"], 1e-2), args.reps)
    for B in (8, 64):
        probs = []
        for b in range(B):
            q = synth.make_warp_problem(n_matches=args.matches, seed=100 + b)
            probs.append(dict(bbs=nrsfm.Bbs(*q["bbs"]), kp1=q["kp1"], kp2=q["kp2"], invsig=q["invsig"], fx_slot=q["fy"], fy_slot=q["fx"], lam=1e-2, fx=q["fx"], fy=q["fy"], init_lam=1e-2))
        dtb = timeit(lambda: nrsfm.calculateSchwarpsBatch(ctx, probs, 3), max(2, args.reps // 4))
        out.append({"metric": "Schwarp initialise + fit keyframe pairs/s, batched with Warp::initialize inside the call", "value": B / dtb, "unit": "pairs/s",
                    "pairs_per_call": B, "matches": args.matches, "ms_per_call": 1e3 * dtb, "ms_per_single_dsh_warp_initialize": 1e3 * dt_init})
    for r in out:
        print(json.dumps(r))
    ctx.close()


if __name__ == "__main__":
    main()
