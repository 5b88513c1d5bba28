"""GPU tests of the device-resident mapping chain (dsh_diffdb_*, dsh_schwarp_fit_batch_store, dsh_normals_estimate_db): the DiffProp
records stay in HBM between SchwarpDatabase::calculateSchwarps (SchwarpDatabase.cc:299-345) and NormalEstimator::ObtainK1K2
(NormalEstimator.cc:50-110).  The database path must reproduce, bit for bit, what the host-side records give through
dsh_normals_estimate, and that agrees with the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NKEYS = ["rec_ptr", "recs", "rec_is_ref", "rec_first_normal", "rec_has_first_normal", "x0", "has_x0", "ref_uv"]


def _pairs(n_pairs, n_pool, seed):
    """Keyframe pairs over one pool of map points (fixed key points in the reference keyframe, a rigid plane seen from n_pairs other
    cameras): pair b sees a random subset of the pool (some matches belong to no stored point: id -1, the reference's `reference
    keyframe is another one` case); a point's matches appear in several pairs."""
    from defslam_amd import nrsfm, synth
    rng = np.random.default_rng(seed)
    pool = np.stack([rng.uniform(-0.5, 0.5, n_pool), rng.uniform(-0.4, 0.4, n_pool)], 1)
    probs = []
    for b in range(n_pairs):
        P = int(rng.integers(40, 320))
        sub = rng.choice(n_pool, size=P, replace=False).astype(np.int32)
        motion = (rng.normal(scale=0.05, size=3), rng.normal(scale=0.08, size=3))
        pr = synth.make_warp_problem(P, seed=100 * seed + b, outliers=0.04 if b % 2 else 0.0, kp1=pool[sub], motion=motion)
        pid = sub.copy()
        pid[rng.uniform(size=P) < 0.15] = -1
        idx2 = rng.permutation(4 * P)[:P].astype(np.int32)
        probs.append(dict(bbs=nrsfm.Bbs(*pr["bbs"]), kp1=pr["kp1"], kp2=pr["kp2"], invsig=pr["invsig"], fx_slot=pr["fy"], fy_slot=pr["fx"], lam=1e-2,
                          fx=pr["fx"], fy=pr["fy"], x0=pr["x0"], max_iters=3, point_id=pid, idx2=idx2, tag=1000 + b))
    return probs, pool.astype(np.float32)


def _host_grouping(probs, res, ids):
    """What the host map of the reference holds after the fits: per requested point its records in insertion order."""
    per = {}
    for q, r in zip(probs, res):
        for i in range(q["point_id"].shape[0]):
            if q["point_id"][i] >= 0 and not r[2][i]:
                per.setdefault(int(q["point_id"][i]), []).append((r[1][i], q["tag"], int(q["idx2"][i])))
    recs, tags, idx2, owner, ptr = [], [], [], [], [0]
    for k, p in enumerate(ids):
        for rec, t, j in per.get(int(p), []):
            recs.append(rec); tags.append(t); idx2.append(j); owner.append(k)
        ptr.append(len(recs))
    R = len(recs)
    return (np.array(ptr, np.int32), np.array(recs, np.float32).reshape(R, 18), np.ones(R, np.uint8), np.zeros((R, 2), np.float32), np.zeros(R, np.uint8),
            np.array(tags, np.int32), np.array(idx2, np.int32), np.array(owner, np.int32))


def _check(d, g, tags, idx2, owner):
    for k in ["k1k2", "cov", "status", "normal_ref", "iters"]:
        np.testing.assert_array_equal(getattr(d, k), getattr(g, k), err_msg=k)
    np.testing.assert_array_equal(d.rec_point, owner)
    np.testing.assert_array_equal(d.rec_tag, tags)
    np.testing.assert_array_equal(d.rec_idx2, idx2)
    np.testing.assert_array_equal(d.normal_rec.view(np.uint32), g.normal_rec.view(np.uint32))
    np.testing.assert_array_equal(d.rec_written, g.rec_written)


@pytest.mark.parametrize("n_pairs,n_pool,seed", [(6, 500, 1), (1, 400, 2), (17, 900, 3)])
def test_database_chain_equals_host_records_and_the_oracle(gpu_ctx, oracle_mod, n_pairs, n_pool, seed):
    from defslam_amd import nrsfm
    rng = np.random.default_rng(seed)
    probs, pool = _pairs(n_pairs, n_pool, seed)
    host = nrsfm.calculateSchwarpsBatch(gpu_ctx, probs)
    db = nrsfm.DiffDatabase(gpu_ctx, 20000)
    stored = nrsfm.calculateSchwarpsBatch(gpu_ctx, probs, db=db)
    for h, s in zip(host, stored):                                  # storing does not change what the fit returns
        np.testing.assert_array_equal(h[0], s[0])
        np.testing.assert_array_equal(h[1].view(np.uint32), s[1].view(np.uint32))
        np.testing.assert_array_equal(h[2], s[2])
        np.testing.assert_array_equal(h[3], s[3])
    n_kept = sum(int(((q["point_id"] >= 0) & ~r[2]).sum()) for q, r in zip(probs, host))
    assert len(db) == n_kept > 0
    # request: a scrambled subset of the pool and ids nobody stored (beyond the largest stored id too)
    ids = rng.permutation(n_pool)[: n_pool // 2].astype(np.int32)
    ids = np.concatenate([ids[:7], [n_pool + 5, 10 * n_pool], ids[7:]]).astype(np.int32)
    P = ids.shape[0]
    x0 = rng.normal(scale=0.2, size=(P, 2)).astype(np.float32)
    has_x0 = (rng.uniform(size=P) < 0.5).astype(np.uint8)
    ref_uv = np.where((ids < n_pool)[:, None], pool[np.minimum(ids, n_pool - 1)], 0.0).astype(np.float32)
    ptr, recs, is_ref, fn, hfn, tags, idx2, owner = _host_grouping(probs, host, ids)
    assert (n_pairs == 1 or (np.diff(ptr) >= 2).any()) and (np.diff(ptr) == 0).any()
    g = nrsfm.ObtainK1K2(gpu_ctx, ptr, recs, is_ref, fn, hfn, x0, has_x0, ref_uv)
    d = nrsfm.ObtainK1K2Database(gpu_ctx, db, ids, x0, has_x0, ref_uv)
    _check(d, g, tags, idx2, owner)
    # the oracle on the same records
    o = oracle_mod.normals(ptr, recs, is_ref, fn, hfn, x0, has_x0, ref_uv)
    np.testing.assert_array_equal(d.status, o["status"])
    np.testing.assert_array_equal(d.rec_written, o["rec_written"])
    assert (d.iters == o["iters"]).mean() >= 0.99                   # same accept/reject sequence (a long run on a flat minimum may end a step apart)
    solved = o["status"] == 0
    assert solved.any()
    # points whose few records disagree (an outlier match that survived the fit) have a flat minimum: the last bits of the two
    # implementations' iterates are amplified there.  1e-9 for (at least) 99 % of the points, 1e-4 for every one.
    err = np.abs(d.k1k2[solved] - o["k1k2"][solved]).max(1) / np.maximum(np.abs(o["k1k2"][solved]).max(1), 1e-3)
    assert (err < 1e-9).mean() >= 0.99 and err.max() < 1e-4, (float((err < 1e-9).mean()), float(err.max()))
    tight = np.zeros(P, bool)
    tight[np.flatnonzero(solved)[err < 1e-9]] = True                  # the normals follow k1k2: float32 resolution where k1k2 agrees, 1e-4 elsewhere
    np.testing.assert_allclose(d.normal_ref[tight], o["normal_ref"][tight], rtol=0, atol=1e-6)
    np.testing.assert_allclose(d.normal_ref[solved], o["normal_ref"][solved], rtol=0, atol=2e-4)
    wr = o["rec_written"].astype(bool)
    np.testing.assert_allclose(d.normal_rec[wr & tight[owner]], o["normal_rec"][wr & tight[owner]], rtol=0, atol=1e-6)
    np.testing.assert_allclose(d.normal_rec[wr], o["normal_rec"][wr], rtol=0, atol=2e-4)
    # the same database filled two other ways: without copying a record to the host, and from host records
    db2 = nrsfm.DiffDatabase(gpu_ctx, 20000)
    half = max(1, n_pairs // 2)
    quiet = nrsfm.calculateSchwarpsBatch(gpu_ctx, probs[:half], db=db2, want_records=False)     # two calls: appending continues where the first ended
    assert all(not r[1].any() for r in quiet) and all((r[2] == h[2]).all() for r, h in zip(quiet, host))
    if half < n_pairs:
        nrsfm.calculateSchwarpsBatch(gpu_ctx, probs[half:], db=db2, want_records=False)
    assert len(db2) == n_kept
    _check(nrsfm.ObtainK1K2Database(gpu_ctx, db2, ids, x0, has_x0, ref_uv), g, tags, idx2, owner)
    db3 = nrsfm.DiffDatabase(gpu_ctx, n_kept)
    for q, r in zip(probs, host):
        k = (q["point_id"] >= 0) & ~r[2]
        db3.append(r[1][k], q["point_id"][k], np.full(int(k.sum()), q["tag"], np.int32), q["idx2"][k])
    assert len(db3) == n_kept
    _check(nrsfm.ObtainK1K2Database(gpu_ctx, db3, ids, x0, has_x0, ref_uv), g, tags, idx2, owner)
    # Shape from Normals with the normals picked on the device out of that solve: points (reference keyframe) and records (second keyframe)
    okp = np.flatnonzero(d.status == 0)[:150]
    okr = np.flatnonzero(d.rec_written)[:150]
    sel = np.r_[okp, -1 - okr].astype(np.int32)
    nrm = np.r_[d.normal_ref[okp], d.normal_rec[okr]]
    uu, vv = rng.uniform(-0.5, 0.5, sel.shape[0]), rng.uniform(-0.4, 0.4, sel.shape[0])
    b1 = nrsfm.Bbs(-0.62, 0.62, 13, -0.52, 0.52, 15, 1)
    h = nrsfm.ShapeFromNormals(gpu_ctx, b1, uu, vv, nrm, 1e-3, 1.3, uu, vv)
    nrsfm.ObtainK1K2Database(gpu_ctx, db, ids, x0, has_x0, ref_uv, per_record=False)      # the solve the handle remembers
    s = nrsfm.ShapeFromNormalsDatabase(gpu_ctx, b1, db, sel, uu, vv, 1e-3, 1.3, uu, vv)
    assert h[0] and s[0]
    np.testing.assert_array_equal(s[1], h[1])
    np.testing.assert_array_equal(s[3].view(np.uint32), h[3].view(np.uint32))
    from defslam_amd.sft import DshError
    for bad in (P, -1 - d.normal_rec.shape[0]):
        with pytest.raises(DshError, match="outside the last normal solve"):
            nrsfm.ShapeFromNormalsDatabase(gpu_ctx, b1, db, np.array([bad], np.int32), uu[:1], vv[:1], 1e-3, 1.3, uu, vv)
    # without the per-record outputs
    dq = nrsfm.ObtainK1K2Database(gpu_ctx, db, ids, x0, has_x0, ref_uv, per_record=False)
    np.testing.assert_array_equal(dq.k1k2, g.k1k2)
    assert dq.normal_rec.shape[0] == 0
    # a cleared database holds nothing: every point is skipped like a point without records
    db.clear()
    assert len(db) == 0
    e = nrsfm.ObtainK1K2Database(gpu_ctx, db, ids, x0, has_x0, ref_uv)
    ge = nrsfm.ObtainK1K2(gpu_ctx, np.zeros(P + 1, np.int32), np.zeros((0, 18), np.float32), np.zeros(0, np.uint8), np.zeros((0, 2), np.float32),
                          np.zeros(0, np.uint8), x0, has_x0, ref_uv)
    np.testing.assert_array_equal(e.status, ge.status)
    assert e.normal_rec.shape[0] == 0
    for x in (db, db2, db3):
        x.close()


def test_database_capacity_and_argument_errors(gpu_ctx):
    from defslam_amd import nrsfm
    from defslam_amd.sft import DshError
    probs, _ = _pairs(2, 300, 9)
    # a database grows on demand like the reference's map (WarpDatabase.h:61): a store that overflows the initial capacity keeps every
    # record, in the same order as a database that was large from the start
    small = nrsfm.DiffDatabase(gpu_ctx, 20)
    big = nrsfm.DiffDatabase(gpu_ctx, 4096)
    rs = nrsfm.calculateSchwarpsBatch(gpu_ctx, probs, db=small)
    rb = nrsfm.calculateSchwarpsBatch(gpu_ctx, probs, db=big)
    assert len(small) == len(big) > 20
    for a, b in zip(rs, rb):
        np.testing.assert_array_equal(a[2], b[2])      # drop flags
    ids = np.unique(np.concatenate([q["point_id"] for q in probs]))
    ids = ids[ids >= 0].astype(np.int32)
    z2, z1 = np.zeros((ids.size, 2)), np.zeros(ids.size)
    ns, nb = (nrsfm.ObtainK1K2Database(gpu_ctx, d, ids, z2, z1, z2) for d in (small, big))
    np.testing.assert_array_equal(ns.k1k2, nb.k1k2)
    np.testing.assert_array_equal(ns.rec_point, nb.rec_point)
    np.testing.assert_array_equal(ns.rec_idx2, nb.rec_idx2)
    big.close()
    small.append(np.zeros((30, 18), np.float32), np.zeros(30, np.int32))      # a host append grows it as well
    small.clear()
    small.append(np.zeros((5, 18), np.float32), np.arange(5, dtype=np.int32))
    assert len(small) == 5
    r = nrsfm.ObtainK1K2Database(gpu_ctx, small, np.zeros(0, np.int32), np.zeros((0, 2)), np.zeros(0), np.zeros((0, 2)))
    assert r.k1k2.shape == (0, 2)
    with pytest.raises(DshError):
        nrsfm.DiffDatabase(gpu_ctx, 0)
    # a database belongs to the context that created it
    from defslam_amd import sft
    other = sft.Context(0)
    try:
        with pytest.raises(DshError, match="bad argument"):
            nrsfm.ObtainK1K2Database(other, small, np.arange(3, dtype=np.int32), np.zeros((3, 2)), np.zeros(3), np.zeros((3, 2)))
        with pytest.raises(DshError, match="another context"):
            nrsfm.calculateSchwarpsBatch(other, probs, db=small)
    finally:
        other.close()
    small.append(np.zeros((0, 18), np.float32), np.zeros(0, np.int32))      # nothing to append is fine
    assert len(small) == 5
    small.close()


def test_database_outlives_its_context_and_can_be_destroyed_in_either_order():
    """dsh_destroy detaches the databases of the context: they stay valid objects (calls on them return an error instead of touching freed
    memory) and dsh_diffdb_destroy works after the context is gone (Context.close() before DiffDatabase.__del__)."""
    from defslam_amd import nrsfm, sft
    from defslam_amd.sft import DshError
    ctx = sft.Context(0)
    db = nrsfm.DiffDatabase(ctx, 64)
    db.append(np.zeros((5, 18), np.float32), np.arange(5, dtype=np.int32))
    db2 = nrsfm.DiffDatabase(ctx, 64)
    db2.close()                      # database first
    lib = ctx._L
    ctx.close()                      # then the context, with `db` still alive
    assert int(lib.dsh_diffdb_count(db._h)) == 5
    assert lib.dsh_diffdb_append(db._h, 0, None, None, None, None) != 0      # detached: refused, nothing dereferenced
    assert lib.dsh_diffdb_destroy(db._h) == 0
    db._h = None
