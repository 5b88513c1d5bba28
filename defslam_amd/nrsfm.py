"""Host-side mirror of the mapping-side (NRSfM) entry points, on top of the C ABI.

Reference interfaces being mirrored:
  * BBS::eval / BBS::coloc / BBS::coloc_deriv   (Thirdparty/BBS/bbs.h:52-66)
  * defSLAM::NormalEstimator::ObtainK1K2()      (Modules/Mapping/NormalEstimator.h:46-53)
The WarpDatabase / MapPoint / KeyFrame objects are replaced by flat record arrays (INTEGRATION.md shows the
shim that fills them from `WarpDatabase::getDiffDatabase()`).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .sft import Context, _ptr


@dataclass
class Bbs:
    """BBS::bbs_t."""
    umin: float
    umax: float
    nptsu: int
    vmin: float
    vmax: float
    nptsv: int
    valdim: int

    def c(self) -> _lib.BbsC:
        return _lib.BbsC(self.umin, self.umax, self.nptsu, self.vmin, self.vmax, self.nptsv, self.valdim)


def bbs_eval(ctx: Context, bbs: Bbs, ctrlpts: np.ndarray, u: np.ndarray, v: np.ndarray, du: int = 0, dv: int = 0):
    """BBS::eval: returns (val[n, valdim], outside[n])."""
    ctrl = np.ascontiguousarray(ctrlpts, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    n = u.shape[0]
    val = np.zeros((n, bbs.valdim))
    outside = np.zeros(n, np.uint8)
    b = bbs.c()
    ctx._check(ctx._L.dsh_bbs_eval(ctx._h, C.byref(b), _ptr(ctrl, C.c_double), _ptr(u, C.c_double), _ptr(v, C.c_double), n, du, dv,
                                   _ptr(val, C.c_double), _ptr(outside, C.c_uint8)), "dsh_bbs_eval")
    return val, outside.astype(bool)


def bbs_coloc(ctx: Context, bbs: Bbs, u: np.ndarray, v: np.ndarray, du: int = 0, dv: int = 0):
    """Row view of BBS::coloc / coloc_deriv: (cols[n,16], w[n,16], n_outside)."""
    u = np.ascontiguousarray(u, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    n = u.shape[0]
    cols = np.zeros((n, 16), np.int32)
    w = np.zeros((n, 16))
    cnt = C.c_int32(0)
    b = bbs.c()
    ctx._check(ctx._L.dsh_bbs_coloc(ctx._h, C.byref(b), _ptr(u, C.c_double), _ptr(v, C.c_double), n, du, dv, _ptr(cols, C.c_int32),
                                    _ptr(w, C.c_double), C.byref(cnt)), "dsh_bbs_coloc")
    return cols, w, cnt.value


@dataclass
class NormalsResult:
    k1k2: np.ndarray
    cov: np.ndarray
    status: np.ndarray
    normal_ref: np.ndarray
    normal_rec: np.ndarray
    rec_written: np.ndarray
    iters: np.ndarray


def ObtainK1K2(ctx: Context, rec_ptr, recs, rec_is_ref, rec_first_normal, rec_has_first_normal, x0, has_x0, ref_uv) -> NormalsResult:
    """NormalEstimator::ObtainK1K2 over the points with new observations. `recs` is (R, 18) float32 in DIFFPROP_FIELDS order."""
    rec_ptr = np.ascontiguousarray(rec_ptr, np.int32)
    P = rec_ptr.shape[0] - 1
    recs = np.ascontiguousarray(recs, np.float32).reshape(-1, 18)
    R = recs.shape[0]
    is_ref = np.ascontiguousarray(rec_is_ref, np.uint8)
    fn = np.ascontiguousarray(rec_first_normal, np.float32).reshape(-1, 2)
    hfn = np.ascontiguousarray(rec_has_first_normal, np.uint8)
    x0 = np.ascontiguousarray(x0, np.float32).reshape(-1, 2)
    hx0 = np.ascontiguousarray(has_x0, np.uint8)
    uv = np.ascontiguousarray(ref_uv, np.float32).reshape(-1, 2)
    out = NormalsResult(np.zeros((P, 2)), np.zeros((P, 2, 2)), np.zeros(P, np.int32), np.zeros((P, 3), np.float32),
                        np.zeros((R, 3), np.float32), np.zeros(R, np.uint8), np.zeros(P, np.int32))
    ctx._check(ctx._L.dsh_normals_estimate(ctx._h, P, _ptr(rec_ptr, C.c_int32), _ptr(recs, C.c_float), _ptr(is_ref, C.c_uint8), _ptr(fn, C.c_float),
                                           _ptr(hfn, C.c_uint8), _ptr(x0, C.c_float), _ptr(hx0, C.c_uint8), _ptr(uv, C.c_float),
                                           _ptr(out.k1k2, C.c_double), _ptr(out.cov, C.c_double), _ptr(out.status, C.c_int32),
                                           _ptr(out.normal_ref, C.c_float), _ptr(out.normal_rec, C.c_float), _ptr(out.rec_written, C.c_uint8),
                                           _ptr(out.iters, C.c_int32)), "dsh_normals_estimate")
    return out


class DiffDatabase:
    """defSLAM::WarpDatabase's mapPointsDB_ (WarpDatabase.h:61) held in HBM: dsh_diffdb."""

    def __init__(self, ctx: Context, capacity: int):
        self._ctx, self._h = ctx, None
        h = C.c_void_p()
        ctx._check(ctx._L.dsh_diffdb_create(ctx._h, int(capacity), C.byref(h)), "dsh_diffdb_create")
        self._h = h

    def close(self):
        if self._h:
            self._ctx._L.dsh_diffdb_destroy(self._h)
            self._h = None

    __del__ = close

    def clear(self):
        self._ctx._check(self._ctx._L.dsh_diffdb_clear(self._h), "dsh_diffdb_clear")

    def __len__(self):
        return int(self._ctx._L.dsh_diffdb_count(self._h))

    def append(self, recs, point_id, tag=None, idx2=None):
        recs = np.ascontiguousarray(recs, np.float32).reshape(-1, 18)
        pid = np.ascontiguousarray(point_id, np.int32)
        tag = np.ascontiguousarray(tag, np.int32) if tag is not None else None
        idx2 = np.ascontiguousarray(idx2, np.int32) if idx2 is not None else None
        assert pid.shape[0] == recs.shape[0]
        self._ctx._check(self._ctx._L.dsh_diffdb_append(self._h, recs.shape[0], _ptr(recs, C.c_float), _ptr(pid, C.c_int32), _ptr(tag, C.c_int32),
                                                        _ptr(idx2, C.c_int32)), "dsh_diffdb_append")


@dataclass
class NormalsDbResult:
    k1k2: np.ndarray
    cov: np.ndarray
    status: np.ndarray
    normal_ref: np.ndarray
    iters: np.ndarray
    rec_point: np.ndarray
    rec_tag: np.ndarray
    rec_idx2: np.ndarray
    normal_rec: np.ndarray
    rec_written: np.ndarray


def ObtainK1K2Database(ctx: Context, db: DiffDatabase, point_ids, x0, has_x0, ref_uv, per_record=True) -> NormalsDbResult:
    """NormalEstimator::ObtainK1K2 over the device-resident database (dsh_normals_estimate_db): the records never visit the host."""
    ids = np.ascontiguousarray(point_ids, np.int32)
    P = ids.shape[0]
    x0 = np.ascontiguousarray(x0, np.float32).reshape(-1, 2)
    hx0 = np.ascontiguousarray(has_x0, np.uint8)
    uv = np.ascontiguousarray(ref_uv, np.float32).reshape(-1, 2)
    cap = len(db) if per_record else 0
    k, cov, st, nref, it = np.zeros((P, 2)), np.zeros((P, 2, 2)), np.zeros(P, np.int32), np.zeros((P, 3), np.float32), np.zeros(P, np.int32)
    rp, rt, ri = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    nrec, wr = np.zeros((cap, 3), np.float32), np.zeros(cap, np.uint8)
    n = C.c_int32(0)
    pr = per_record and cap > 0
    ctx._check(ctx._L.dsh_normals_estimate_db(ctx._h, db._h, P, _ptr(ids, C.c_int32), _ptr(x0, C.c_float), _ptr(hx0, C.c_uint8), _ptr(uv, C.c_float),
                                              _ptr(k, C.c_double), _ptr(cov, C.c_double), _ptr(st, C.c_int32), _ptr(nref, C.c_float), _ptr(it, C.c_int32),
                                              cap, C.byref(n), _ptr(rp, C.c_int32) if pr else None, _ptr(rt, C.c_int32) if pr else None,
                                              _ptr(ri, C.c_int32) if pr else None, _ptr(nrec, C.c_float) if pr else None, _ptr(wr, C.c_uint8) if pr else None),
               "dsh_normals_estimate_db")
    R = n.value if pr else 0
    return NormalsDbResult(k, cov, st, nref, it, rp[:R], rt[:R], ri[:R], nrec[:R], wr[:R])


def schwarp_eval(ctx: Context, bbs: Bbs, kp1, kp2, invsig, fx_slot, fy_slot, lam, x, want_jacobian=True):
    """Warps::Warp::Evaluate + Warps::Schwarzian::Evaluate once: (residuals[2P+4N], J[(2P+4N), 2N] or None)."""
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    invsig = np.ascontiguousarray(invsig, np.float32)
    x = np.ascontiguousarray(x, np.float64)
    P, N = kp1.shape[0], bbs.nptsu * bbs.nptsv
    r = np.zeros(2 * P + 4 * N)
    J = np.zeros((2 * P + 4 * N, 2 * N)) if want_jacobian else None
    b = bbs.c()
    ctx._check(ctx._L.dsh_schwarp_eval(ctx._h, C.byref(b), P, _ptr(kp1, C.c_float), _ptr(kp2, C.c_float), _ptr(invsig, C.c_float), float(fx_slot),
                                       float(fy_slot), float(lam), _ptr(x, C.c_double), _ptr(r, C.c_double), _ptr(J, C.c_double)), "dsh_schwarp_eval")
    return r, J


def calculateSchwarps(ctx: Context, bbs: Bbs, kp1, kp2, invsig, fx_slot, fy_slot, lam, fx, fy, x0, max_iters=3):
    """SchwarpDatabase::calculateSchwarps: returns (x, diffprops[P,18] float32, drop[P], info, costs)."""
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    invsig = np.ascontiguousarray(invsig, np.float32)
    x = np.array(x0, np.float64, copy=True)
    P = kp1.shape[0]
    diff = np.zeros((P, 18), np.float32)
    drop = np.zeros(P, np.uint8)
    info = np.zeros(2, np.int32)
    costs = np.zeros(2)
    b = bbs.c()
    ctx._check(ctx._L.dsh_schwarp_fit(ctx._h, C.byref(b), P, _ptr(kp1, C.c_float), _ptr(kp2, C.c_float), _ptr(invsig, C.c_float), float(fx_slot),
                                      float(fy_slot), float(lam), float(fx), float(fy), int(max_iters), _ptr(x, C.c_double), _ptr(diff, C.c_float),
                                      _ptr(drop, C.c_uint8), _ptr(info, C.c_int32), _ptr(costs, C.c_double)), "dsh_schwarp_fit")
    return x, diff, drop.astype(bool), info, costs


def calculateSchwarpsBatch(ctx: Context, problems, max_iters=3, db=None, want_records=True):
    """B fits in one call (dsh_schwarp_fit_batch).  With db (a DiffDatabase) the kept records also go into the device-resident database
    (dsh_schwarp_fit_batch_store): every problem then carries point_id[P] (map point of each match, < 0 = not stored), optionally idx2[P]
    and tag; want_records=False leaves the records on the device (the diffprops of the result stay zero).  problems: dicts with bbs (Bbs), kp1, kp2, invsig, fx_slot, fy_slot, lam, fx, fy and
    either x0 (start value) or init_lam (the fit starts from Warp::initialize with that bending weight, computed inside the call).
    Returns a list of (x, diffprops, drop, info, costs) like calculateSchwarps; with init_lam the tuple ends with init_ok."""
    B = len(problems)
    arr = (_lib.SchwarpProblemC * B)()
    # inputs and outputs of all fits live in pooled arrays (one address computation per pool, views per fit)
    Ps = np.array([np.shape(q["kp1"])[0] for q in problems], np.int64)
    off = np.concatenate([[0], np.cumsum(Ps)])
    n2s = np.array([2 * q["bbs"].nptsu * q["bbs"].nptsv for q in problems], np.int64)
    xoff = np.concatenate([[0], np.cumsum(n2s)])
    kp1 = np.ascontiguousarray(np.concatenate([np.asarray(q["kp1"], np.float32).reshape(-1, 2) for q in problems]))
    kp2 = np.ascontiguousarray(np.concatenate([np.asarray(q["kp2"], np.float32).reshape(-1, 2) for q in problems]))
    isg = np.ascontiguousarray(np.concatenate([np.asarray(q["invsig"], np.float32).reshape(-1) for q in problems]))
    assert kp1.shape[0] == kp2.shape[0] == isg.shape[0] == off[-1]
    x = np.zeros(int(xoff[-1]))
    store = db is not None
    diff = (np.empty if (want_records or not store) else np.zeros)((int(off[-1]), 18), np.float32)     # the call writes every record it is handed a buffer for
    drop = np.empty(int(off[-1]), np.uint8)
    a_kp1, a_kp2, a_isg, a_x, a_diff, a_drop = (v.ctypes.data for v in (kp1, kp2, isg, x, diff, drop))
    if store:
        pid = np.ascontiguousarray(np.concatenate([np.asarray(q["point_id"], np.int32).reshape(-1) for q in problems]))
        has_idx2 = [q.get("idx2") is not None for q in problems]
        idx2 = np.ascontiguousarray(np.concatenate([np.asarray(q["idx2"], np.int32).reshape(-1) if h else np.arange(P, dtype=np.int32)
                                                    for q, h, P in zip(problems, has_idx2, Ps)]))
        assert pid.shape[0] == idx2.shape[0] == off[-1]
        a_pid, a_idx2 = pid.ctypes.data, idx2.ctypes.data
        st = (_lib.SchwarpStoreC * B)()
    for b, q in enumerate(problems):
        o, P, xo = int(off[b]), int(Ps[b]), int(xoff[b])
        init_lam = float(q.get("init_lam", 0.0))
        if init_lam <= 0.0:
            x[xo:xo + int(n2s[b])] = q["x0"]
        a = arr[b]
        a.bbs = q["bbs"].c()
        a.P = P
        a.kp1, a.kp2, a.invsig = a_kp1 + 8 * o, a_kp2 + 8 * o, a_isg + 4 * o
        a.fx_slot, a.fy_slot, a.lam, a.fx, a.fy = float(q["fx_slot"]), float(q["fy_slot"]), float(q["lam"]), float(q["fx"]), float(q["fy"])
        a.max_iters = int(q.get("max_iters", max_iters))
        a.x, a.drop = a_x + 8 * xo, a_drop + o
        a.diff = a_diff + 72 * o if (want_records or not store) else None
        a.init_lambda = init_lam
        if store:
            st[b].point_id, st[b].idx2, st[b].tag = a_pid + 4 * o, a_idx2 + 4 * o, int(q.get("tag", b))
    if store:
        ctx._check(ctx._L.dsh_schwarp_fit_batch_store(ctx._h, B, arr, st, db._h), "dsh_schwarp_fit_batch_store")
    else:
        ctx._check(ctx._L.dsh_schwarp_fit_batch(ctx._h, B, arr), "dsh_schwarp_fit_batch")
    dropb = drop.view(np.bool_)
    out = []
    for b in range(B):
        o, e, xo = int(off[b]), int(off[b + 1]), int(xoff[b])
        a = arr[b]
        t = (x[xo:xo + int(n2s[b])], diff[o:e], dropb[o:e], np.array(a.info[:], np.int32), np.array(a.costs[:]))
        out.append(t + (bool(a.init_ok),) if a.init_lambda > 0.0 else t)
    return out


def bbs_bending(bbs: Bbs, lam: float) -> np.ndarray:
    """BBS::BendingEigen as a dense symmetric N x N matrix (host side of the library)."""
    L = _lib.load()
    N = bbs.nptsu * bbs.nptsv
    out = np.zeros((N, N))
    b = bbs.c()
    rc = L.dsh_bbs_bending(C.byref(b), float(lam), _ptr(out, C.c_double))
    if rc != 0:
        raise ValueError("dsh_bbs_bending: bad argument")
    return out


def ShapeFromNormals(ctx: Context, bbs: Bbs, u, v, normals, bending_weight: float, mean_depth: float, u_all, v_all):
    """ShapeFromNormals(refKf, bendingWeight).estimate(): returns (ok, ctrl_raw[N], ctrl[N], pts[n_all, 3] float32)."""
    u = np.ascontiguousarray(u, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    ua = np.ascontiguousarray(u_all, np.float64)
    va = np.ascontiguousarray(v_all, np.float64)
    N = bbs.nptsu * bbs.nptsv
    raw = np.zeros(N)
    ctrl = np.zeros(N)
    pts = np.zeros((ua.shape[0], 3), np.float32)
    ok = C.c_int32(0)
    b = bbs.c()
    ctx._check(ctx._L.dsh_sfn_estimate(ctx._h, C.byref(b), u.shape[0], _ptr(u, C.c_double), _ptr(v, C.c_double), _ptr(nrm, C.c_float), float(bending_weight),
                                       float(mean_depth), ua.shape[0], _ptr(ua, C.c_double), _ptr(va, C.c_double), _ptr(raw, C.c_double), _ptr(ctrl, C.c_double),
                                       _ptr(pts, C.c_float), C.byref(ok)), "dsh_sfn_estimate")
    return bool(ok.value), raw, ctrl, pts


def ShapeFromNormalsDatabase(ctx: Context, bbs: Bbs, db: DiffDatabase, sel, u, v, bending_weight: float, mean_depth: float, u_all, v_all):
    """ShapeFromNormals with the normals picked on the device from the last ObtainK1K2Database of db (dsh_sfn_estimate_db): sel >= 0 = index of
    a requested point (normal in its reference keyframe), sel < 0 = record -1 - sel (normal propagated to the record's second keyframe)."""
    sel = np.ascontiguousarray(sel, np.int32)
    u = np.ascontiguousarray(u, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    assert sel.shape[0] == u.shape[0] == v.shape[0]
    ua = np.ascontiguousarray(u_all, np.float64)
    va = np.ascontiguousarray(v_all, np.float64)
    N = bbs.nptsu * bbs.nptsv
    raw, ctrl, pts, ok = np.zeros(N), np.zeros(N), np.zeros((ua.shape[0], 3), np.float32), C.c_int32(0)
    b = bbs.c()
    ctx._check(ctx._L.dsh_sfn_estimate_db(ctx._h, C.byref(b), db._h, u.shape[0], _ptr(sel, C.c_int32), _ptr(u, C.c_double), _ptr(v, C.c_double), float(bending_weight),
                                          float(mean_depth), ua.shape[0], _ptr(ua, C.c_double), _ptr(va, C.c_double), _ptr(raw, C.c_double), _ptr(ctrl, C.c_double),
                                          _ptr(pts, C.c_float), C.byref(ok)), "dsh_sfn_estimate_db")
    return bool(ok.value), raw, ctrl, pts


def WarpInitialize(ctx: Context, bbs: Bbs, kp1, kp2, lam: float):
    """Warps::Warp::initialize: returns (ok, x[2N])."""
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    x = np.zeros(2 * bbs.nptsu * bbs.nptsv)
    ok = C.c_int32(0)
    b = bbs.c()
    ctx._check(ctx._L.dsh_warp_initialize(ctx._h, C.byref(b), kp1.shape[0], _ptr(kp1, C.c_float), _ptr(kp2, C.c_float), float(lam), _ptr(x, C.c_double), C.byref(ok)),
               "dsh_warp_initialize")
    return bool(ok.value), x


def CalculateInitialSchwarp(ctx: Context, bbs: Bbs, kp1, kp2, invsig, fx, fy, lam: float):
    """DefORBmatcher::CalculateInitialSchwarp (DefORBmatcher.cc:111-187): Warp::initialize, NaN control points of the first
    2 NCu NCu entries -> 0, then the residuals the reference reads from ceres::Problem::Evaluate -- loss-corrected by HuberLoss(5.77)
    (Corrector with rho'' <= 0: the one 2P-residual block times sqrt(rho'(|r|^2))) -- and its test residuals[2i]^2 + residuals[2i+1]^2 > 20.
    Returns (x[2N], outlier[P] bool, corrected residuals[2P])."""
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    P = kp1.shape[0]
    _, x = WarpInitialize(ctx, bbs, kp1, kp2, lam)
    nz = min(x.size, 2 * bbs.nptsu * bbs.nptsu)
    x[:nz][np.isnan(x[:nz])] = 0.0
    res, _ = schwarp_eval(ctx, bbs, kp1, kp2, invsig, fx, fy, 0.0, x, want_jacobian=False)
    r = res[:2 * P].copy()
    s = 0.0
    for v in r:                       # index-order sum, like the shim
        s += v * v
    if s > 5.77 * 5.77:
        r *= np.sqrt(5.77 / np.sqrt(s))
    i = np.arange(P)
    err = r[2 * i] ** 2 + r[2 * i + 1] ** 2 if P else np.zeros(0)
    return x, err > 20, r


def searchBySchwarp(ctx: Context, bbs: Bbs, x, kp1, desc1, cam2, bounds2, kp2, desc2, has_mp2, radius: float = 2.0, th_low: int = 50, grid=(64, 48)):
    """DefORBmatcher::searchBySchwarp: returns match[Q] (index into keyframe 2 or -1)."""
    x = np.ascontiguousarray(x, np.float64)
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    cam2 = np.ascontiguousarray(cam2, np.float32)
    b2 = np.ascontiguousarray(bounds2, np.float32)
    mp2 = np.ascontiguousarray(has_mp2, np.uint8)
    match = np.full(kp1.shape[0], -1, np.int32)
    nm = C.c_int32(0)
    b = bbs.c()
    ctx._check(ctx._L.dsh_search_by_schwarp(ctx._h, C.byref(b), _ptr(x, C.c_double), kp1.shape[0], _ptr(kp1, C.c_float), _ptr(d1, C.c_uint8), _ptr(cam2, C.c_float),
                                            _ptr(b2, C.c_float), int(grid[0]), int(grid[1]), kp2.shape[0], _ptr(kp2, C.c_float), _ptr(d2, C.c_uint8), _ptr(mp2, C.c_uint8),
                                            float(radius), int(th_low), _ptr(match, C.c_int32), C.byref(nm)), "dsh_search_by_schwarp")
    assert nm.value == int((match >= 0).sum())
    return match
