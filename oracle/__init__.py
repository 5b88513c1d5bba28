"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  Nothing under defslam_amd/ imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
TRACE_STRIDE = 8


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "_build", "libdefslam_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "libdefslam_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.sft_oracle_solve.restype = C.c_int
        _LIB.tmpl_oracle_build.restype = C.c_int
    return _LIB


_LIB_FAST = None
FAST_FLAGS = "-O3 -march=native (FMA contraction allowed)"


def lib_fast() -> C.CDLL:
    """The timing-only build of sft_oracle.c (oracle/Makefile: libdefslam_oracle_fast.so).  bench.py's cpu_baseline is its only user."""
    global _LIB_FAST
    if _LIB_FAST is None:
        so = os.path.join(_HERE, "_build", "libdefslam_oracle_fast.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "_build/libdefslam_oracle_fast.so"], stdout=subprocess.DEVNULL)
        _LIB_FAST = C.CDLL(so)
        _LIB_FAST.sft_oracle_solve.restype = C.c_int
    return _LIB_FAST


def _p(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(t))


@dataclass
class TemplateConsts:
    n: int
    xyz0: np.ndarray
    facets: np.ndarray          # sorted (F,3)
    E: int
    edge_nodes: np.ndarray
    edge_L0: np.ndarray
    nbr_ptr: np.ndarray
    nbr_idx: np.ndarray
    nbr_w: np.ndarray
    inc_ptr: np.ndarray
    inc_edge: np.ndarray
    boundary: np.ndarray
    k0: np.ndarray
    lap0: np.ndarray
    median_L: float


def template_build(xyz0: np.ndarray, facets: np.ndarray) -> TemplateConsts:
    L = lib()
    xyz0 = np.ascontiguousarray(xyz0, dtype=np.float64)
    facets = np.ascontiguousarray(facets, dtype=np.int32)
    n, F = xyz0.shape[0], facets.shape[0]
    edge_nodes = np.zeros((3 * F, 2), np.int32)
    edge_L0 = np.zeros(3 * F)
    nbr_ptr = np.zeros(n + 1, np.int32)
    nbr_idx = np.zeros(6 * F, np.int32)
    nbr_w = np.zeros(6 * F)
    inc_ptr = np.zeros(n + 1, np.int32)
    inc_edge = np.zeros(6 * F, np.int32)
    boundary = np.zeros(n, np.uint8)
    k0 = np.zeros(n)
    fs = np.zeros((F, 3), np.int32)
    lap0 = np.zeros((n, 3))
    med = C.c_double(0)
    E = L.tmpl_oracle_build(n, _p(xyz0, C.c_double), F, _p(facets, C.c_int32), _p(edge_nodes, C.c_int32), _p(edge_L0, C.c_double),
                            _p(nbr_ptr, C.c_int32), _p(nbr_idx, C.c_int32), _p(nbr_w, C.c_double), _p(inc_ptr, C.c_int32),
                            _p(inc_edge, C.c_int32), _p(boundary, C.c_uint8), _p(k0, C.c_double), _p(fs, C.c_int32),
                            _p(lap0, C.c_double), C.byref(med))
    nn = nbr_ptr[n]
    return TemplateConsts(n, xyz0, fs, E, edge_nodes[:E].copy(), edge_L0[:E].copy(), nbr_ptr, nbr_idx[:nn].copy(), nbr_w[:nn].copy(),
                          inc_ptr, inc_edge[:inc_ptr[n]].copy(), boundary, k0, lap0, med.value)


def template_embed(tc: TemplateConsts, pts):
    """TriangularMesh::calculateFeaturesCoordinates (TriangularMesh.cc:133-236) of float32 points: (facet id or -1, facet nodes, float32
    barycentrics)."""
    L = lib()
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    n_pts = pts.shape[0]
    fid = np.zeros(n_pts, np.int32)
    b = np.zeros((n_pts, 3), np.float32)
    xyz0 = np.ascontiguousarray(tc.xyz0, np.float64)
    fs = np.ascontiguousarray(tc.facets, np.int32)
    L.tmpl_oracle_embed(tc.n, _p(xyz0, C.c_double), int(fs.shape[0]), _p(fs, C.c_int32), n_pts, _p(pts, C.c_float), _p(fid, C.c_int32), _p(b, C.c_float))
    nodes = np.where((fid >= 0)[:, None], fs[np.maximum(fid, 0)], 0).astype(np.int32)
    return fid, nodes, b


@dataclass
class SftResult:
    ret: int
    Tcw: np.ndarray
    pose7: np.ndarray
    xyz: np.ndarray
    chi2_obs: np.ndarray
    outlier: np.ndarray
    rep_error: float
    iters: int
    trials: int
    trace: np.ndarray
    dims: np.ndarray


def sft_solve(tc: TemplateConsts, Tcw, K, n_frame, obs_nodes, obs_bary, obs_uv, obs_invsig2, xyz,
              reg_lap, reg_inex, reg_temp, layers=1, max_iters=50, ldlt_mode=0, fast=False) -> SftResult:
    L = lib_fast() if fast else lib()   # fast: the timing build (never a parity reference)
    M = int(obs_nodes.shape[0])
    Tcw = np.ascontiguousarray(Tcw, dtype=np.float32)
    K = np.ascontiguousarray(K, dtype=np.float64)
    obs_nodes = np.ascontiguousarray(obs_nodes, dtype=np.int32)
    obs_bary = np.ascontiguousarray(obs_bary, dtype=np.float64)
    obs_uv = np.ascontiguousarray(obs_uv, dtype=np.float64)
    obs_invsig2 = np.ascontiguousarray(obs_invsig2, dtype=np.float64)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    Tout = np.zeros((4, 4), np.float32)
    pose7 = np.zeros(7)
    xyz_out = np.zeros_like(xyz)
    chi2 = np.zeros(M)
    outl = np.zeros(M, np.uint8)
    rep = C.c_double(0)
    it = C.c_int32(0)
    tr = C.c_int32(0)
    trace = np.zeros((max(max_iters, 1), TRACE_STRIDE))
    dims = np.zeros(6, np.int32)
    D = C.c_double
    ret = L.sft_oracle_solve(
        tc.n, _p(tc.xyz0, D), _p(tc.boundary, C.c_uint8), _p(tc.nbr_ptr, C.c_int32), _p(tc.nbr_idx, C.c_int32), _p(tc.nbr_w, D),
        _p(tc.k0, D), tc.E, _p(tc.edge_nodes, C.c_int32), _p(tc.edge_L0, D), _p(tc.inc_ptr, C.c_int32), _p(tc.inc_edge, C.c_int32),
        D(tc.median_L), _p(Tcw, C.c_float), _p(K, D), int(n_frame), M, _p(obs_nodes, C.c_int32), _p(obs_bary, D), _p(obs_uv, D),
        _p(obs_invsig2, D), _p(xyz, D), D(reg_lap), D(reg_inex), D(reg_temp), int(layers), int(max_iters), int(ldlt_mode),
        _p(Tout, C.c_float), _p(pose7, D), _p(xyz_out, D), _p(chi2, D), _p(outl, C.c_uint8), C.byref(rep), C.byref(it), C.byref(tr),
        _p(trace, D), _p(dims, C.c_int32))
    return SftResult(ret, Tout, pose7, xyz_out, chi2, outl, rep.value, it.value, tr.value, trace[:it.value].copy(), dims)


def sft_system(tc: TemplateConsts, Tcw, K, n_frame, obs_nodes, obs_bary, obs_uv, obs_invsig2, xyz, reg_lap, reg_inex, reg_temp, layers=1):
    """One 'residuals + Jacobians + normal equations' pass at the given state: (H dense, b, robust chi2)."""
    L = lib()
    L.sft_oracle_system.restype = C.c_int
    M = int(obs_nodes.shape[0])
    Tcw = np.ascontiguousarray(Tcw, dtype=np.float32)
    K = np.ascontiguousarray(K, dtype=np.float64)
    obs_nodes = np.ascontiguousarray(obs_nodes, dtype=np.int32)
    obs_bary = np.ascontiguousarray(obs_bary, dtype=np.float64)
    obs_uv = np.ascontiguousarray(obs_uv, dtype=np.float64)
    obs_invsig2 = np.ascontiguousarray(obs_invsig2, dtype=np.float64)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    D = C.c_double

    def call(Dexp, H, b, chi):
        return L.sft_oracle_system(
            tc.n, _p(tc.xyz0, D), _p(tc.boundary, C.c_uint8), _p(tc.nbr_ptr, C.c_int32), _p(tc.nbr_idx, C.c_int32), _p(tc.nbr_w, D),
            _p(tc.k0, D), tc.E, _p(tc.edge_nodes, C.c_int32), _p(tc.edge_L0, D), _p(tc.inc_ptr, C.c_int32), _p(tc.inc_edge, C.c_int32),
            D(tc.median_L), _p(Tcw, C.c_float), _p(K, D), int(n_frame), M, _p(obs_nodes, C.c_int32), _p(obs_bary, D), _p(obs_uv, D),
            _p(obs_invsig2, D), _p(xyz, D), D(reg_lap), D(reg_inex), D(reg_temp), int(layers), int(Dexp),
            _p(H, D) if H is not None else None, _p(b, D) if b is not None else None, C.byref(chi) if chi is not None else None)

    dim = call(0, None, None, None)
    H = np.zeros((dim, dim), order="F")
    b = np.zeros(dim)
    chi = C.c_double(0)
    call(dim, H, b, chi)
    return H, b, chi.value


# ---- B-spline oracle + the reference's own bbs.cc (oracle/_ref/libbbs_ref.so) -------------------------------
class _BbsT(C.Structure):  # BBS::bbs_t, Thirdparty/BBS/bbs.h:41-50
    _fields_ = [("umin", C.c_double), ("umax", C.c_double), ("nptsu", C.c_int), ("vmin", C.c_double), ("vmax", C.c_double),
                ("nptsv", C.c_int), ("valdim", C.c_int)]


def bbs_eval(bbs, ctrl, u, v, du=0, dv=0):
    """oracle/bbs_oracle.c; bbs = (umin, umax, nptsu, vmin, vmax, nptsv, valdim)."""
    L = lib()
    umin, umax, nptsu, vmin, vmax, nptsv, valdim = bbs
    ctrl = np.ascontiguousarray(ctrl, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    n = u.shape[0]
    val = np.zeros((n, valdim))
    st = np.zeros(n, np.uint8)
    D = C.c_double
    L.bbs_oracle_eval(D(umin), D(umax), int(nptsu), D(vmin), D(vmax), int(nptsv), int(valdim), _p(ctrl, D), _p(u, D), _p(v, D), n, int(du), int(dv),
                      _p(val, D), _p(st, C.c_uint8))
    return val, st.astype(bool)


def bbs_coloc(bbs, u, v, du=0, dv=0):
    L = lib()
    L.bbs_oracle_coloc.restype = C.c_int
    umin, umax, nptsu, vmin, vmax, nptsv, _ = bbs
    u = np.ascontiguousarray(u, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    n = u.shape[0]
    cols = np.zeros((n, 16), np.int32)
    w = np.zeros((n, 16))
    D = C.c_double
    ret = L.bbs_oracle_coloc(D(umin), D(umax), int(nptsu), D(vmin), D(vmax), int(nptsv), _p(u, D), _p(v, D), n, int(du), int(dv),
                             _p(cols, C.c_int32), _p(w, D))
    return cols, w, ret


def bbs_basis(order, t):
    L = lib()
    b = (C.c_double * 4)()
    L.bbs_oracle_basis(int(order), C.c_double(t), b)
    return np.array(b[:])


_REF = None


def ref_bbs_lib():
    """The reference's bbs.cc compiled as-is (oracle/Makefile `ref`); None when it has not been built."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libbbs_ref.so")
        if not os.path.exists(so):
            return None
        _REF = C.CDLL(so)
    return _REF


def ref_bbs_eval(bbs, ctrl, u, v, du=0, dv=0):
    R = ref_bbs_lib()
    b = _BbsT(*bbs)
    ctrl = np.ascontiguousarray(ctrl, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    n = u.shape[0]
    val = np.zeros((n, bbs[6]))
    D = C.c_double
    R._ZN3BBS4evalEPNS_6_bbs_tEPdS2_S2_iS2_ii(C.byref(b), _p(ctrl, D), _p(u, D), _p(v, D), n, _p(val, D), int(du), int(dv))
    return val


def ref_bbs_basis(order, t):
    R = ref_bbs_lib()
    b = (C.c_double * 4)()
    fn = [R._ZN3BBS10eval_basisEdPd, R._ZN3BBS12eval_basis_dEdPd, R._ZN3BBS13eval_basis_ddEdPd][order]
    fn(C.c_double(t), b)
    return np.array(b[:])


def ref_bbs_coloc_dense(bbs, u, v, du=0, dv=0):
    """Dense (nsites x ncols) colocation matrix from the reference's CSC output; returns (A, ret_code)."""
    R = ref_bbs_lib()
    b = _BbsT(*bbs)
    u = np.ascontiguousarray(u, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    n = u.shape[0]
    ncol = bbs[2] * bbs[5]
    pr = np.zeros(16 * n)
    ir = np.zeros(16 * n, np.uint64)
    jc = np.zeros(ncol + 1, np.uint64)
    D = C.c_double
    f = R._ZN3BBS11coloc_derivEPNS_6_bbs_tEPdS2_iiiS2_PmS3_
    f.restype = C.c_int
    ret = f(C.byref(b), _p(u, D), _p(v, D), n, int(du), int(dv), _p(pr, D), _p(ir, C.c_uint64), _p(jc, C.c_uint64))
    A = np.zeros((n, ncol))
    if ret == 0:
        for j in range(ncol):
            for q in range(int(jc[j]), int(jc[j + 1])):
                A[int(ir[q]), j] = pr[q]
    return A, ret


# ---- normals oracle ----------------------------------------------------------------------------------
def normals(rec_ptr, recs, rec_is_ref, rec_first_n, rec_has_first_n, x0, has_x0, ref_uv):
    L = lib()
    rec_ptr = np.ascontiguousarray(rec_ptr, np.int32)
    P = rec_ptr.shape[0] - 1
    recs = np.ascontiguousarray(recs, np.float32).reshape(-1, 18)
    R = recs.shape[0]
    is_ref = np.ascontiguousarray(rec_is_ref, np.uint8)
    fn = np.ascontiguousarray(rec_first_n, np.float32).reshape(-1, 2)
    hfn = np.ascontiguousarray(rec_has_first_n, np.uint8)
    x0 = np.ascontiguousarray(x0, np.float32).reshape(-1, 2)
    hx0 = np.ascontiguousarray(has_x0, np.uint8)
    uv = np.ascontiguousarray(ref_uv, np.float32).reshape(-1, 2)
    out = dict(k1k2=np.zeros((P, 2)), cov=np.zeros((P, 2, 2)), status=np.zeros(P, np.int32), normal_ref=np.zeros((P, 3), np.float32),
               normal_rec=np.zeros((R, 3), np.float32), rec_written=np.zeros(R, np.uint8), iters=np.zeros(P, np.int32), term=np.zeros(P, np.int32))
    L.nrsfm_oracle_normals(P, _p(rec_ptr, C.c_int32), _p(recs, C.c_float), _p(is_ref, C.c_uint8), _p(fn, C.c_float), _p(hfn, C.c_uint8),
                           _p(x0, C.c_float), _p(hx0, C.c_uint8), _p(uv, C.c_float), _p(out["k1k2"], C.c_double), _p(out["cov"], C.c_double),
                           _p(out["status"], C.c_int32), _p(out["normal_ref"], C.c_float), _p(out["normal_rec"], C.c_float),
                           _p(out["rec_written"], C.c_uint8), _p(out["iters"], C.c_int32), _p(out["term"], C.c_int32))
    return out


def record_coeffs(rec18):
    L = lib()
    rec = np.ascontiguousarray(rec18, np.float32)
    q1 = np.zeros(10)
    q2 = np.zeros(10)
    L.nrsfm_oracle_record_coeffs(_p(rec, C.c_float), _p(q1, C.c_double), _p(q2, C.c_double))
    return q1, q2


def poly_eval(q1, q2, x):
    L = lib()
    q1 = np.ascontiguousarray(q1, np.float64)
    q2 = np.ascontiguousarray(q2, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    e = np.zeros(2)
    J = np.zeros(4)
    L.nrsfm_oracle_poly_eval(_p(q1, C.c_double), _p(q2, C.c_double), _p(x, C.c_double), _p(e, C.c_double), _p(J, C.c_double))
    return e, J.reshape(2, 2)


# ---- Schwarp oracle ----------------------------------------------------------------------------------
def schwarp_eval(bbs, kp1, kp2, invsig, fxs, fys, lam, x, want_jacobian=True):
    L = lib()
    umin, umax, nu, vmin, vmax, nv, _ = bbs
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    invsig = np.ascontiguousarray(invsig, np.float32)
    x = np.ascontiguousarray(x, np.float64)
    P, N = kp1.shape[0], nu * nv
    r = np.zeros(2 * P + 4 * N)
    J = np.zeros((2 * P + 4 * N, 2 * N)) if want_jacobian else None
    D = C.c_double
    L.schwarp_oracle_eval(D(umin), D(umax), int(nu), D(vmin), D(vmax), int(nv), P, _p(kp1, C.c_float), _p(kp2, C.c_float), _p(invsig, C.c_float),
                          D(fxs), D(fys), D(lam), _p(x, D), _p(r, D), _p(J, D))
    return r, J


def schwarp_eval_initial(bbs, kp1, kp2, invsig, fx, fy, x):
    """The residuals DefORBmatcher::CalculateInitialSchwarp reads (DefORBmatcher.cc:155-175): ceres::Problem::Evaluate of the Warp block
    under HuberLoss(5.77) with the default apply_loss_function = true, i.e. loss-corrected.  Returns (residuals[2P], cost)."""
    L = lib()
    umin, umax, nu, vmin, vmax, nv, _ = bbs
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    invsig = np.ascontiguousarray(invsig, np.float32)
    x = np.ascontiguousarray(x, np.float64)
    P = kp1.shape[0]
    r = np.zeros(2 * P)
    D = C.c_double
    L.schwarp_oracle_eval_initial.restype = D
    cost = L.schwarp_oracle_eval_initial(D(umin), D(umax), int(nu), D(vmin), D(vmax), int(nv), P, _p(kp1, C.c_float), _p(kp2, C.c_float), _p(invsig, C.c_float),
                                         D(fx), D(fy), _p(x, D), _p(r, D))
    return r, float(cost)


def schwarp_fit(bbs, kp1, kp2, invsig, fxs, fys, lam, fx, fy, x0, max_iters=3):
    L = lib()
    umin, umax, nu, vmin, vmax, nv, _ = bbs
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    invsig = np.ascontiguousarray(invsig, np.float32)
    x = np.array(x0, np.float64, copy=True)
    P = kp1.shape[0]
    diff = np.zeros((P, 18), np.float32)
    drop = np.zeros(P, np.uint8)
    info = np.zeros(2, np.int32)
    costs = np.zeros(2)
    D = C.c_double
    L.schwarp_oracle_fit(D(umin), D(umax), int(nu), D(vmin), D(vmax), int(nv), P, _p(kp1, C.c_float), _p(kp2, C.c_float), _p(invsig, C.c_float),
                         D(fxs), D(fys), D(lam), C.c_float(fx), C.c_float(fy), int(max_iters), _p(x, D), _p(diff, C.c_float), _p(drop, C.c_uint8),
                         _p(info, C.c_int32), _p(costs, D))
    return x, diff, drop.astype(bool), info, costs


# ---- Shape-from-Normals oracle (sfn_oracle.c) ----------------------------------------------------------
def sfn_bending(bbs, lam):
    """Dense N x N bending matrix (restated BBS bending energy)."""
    L = lib()
    umin, umax, nu, vmin, vmax, nv, _ = bbs
    N = nu * nv
    Bm = np.zeros((N, N))
    D = C.c_double
    L.sfn_oracle_bending(D(umin), D(umax), int(nu), D(vmin), D(vmax), int(nv), D(lam), _p(Bm, D))
    return Bm


def ref_bbs_bending(bbs, lam):
    """The reference's own bending_ur (oracle/_ref/libbbs_ref.so): dense symmetric matrix from its CSC upper-right part."""
    R = ref_bbs_lib()
    b = _BbsT(*bbs)
    nu, nv = bbs[2], bbs[5]
    N = nu * nv
    pr = np.zeros(N * 32)
    ir = np.zeros(N * 32, np.uint64)
    jc = np.zeros(N + 8, np.uint64)
    D = C.c_double
    f = R._ZN3BBS10bending_urEPNS_6_bbs_tEdPdPmS3_
    f.restype = None
    f(C.byref(b), D(lam), _p(pr, D), _p(ir, C.c_uint64), _p(jc, C.c_uint64))
    Bm = np.zeros((N, N))
    for j in range(N):
        for q in range(int(jc[j]), int(jc[j + 1])):
            i = int(ir[q])
            Bm[i, j] = pr[q]
            Bm[j, i] = pr[q]
    return Bm


def sfn_rows(bbs, u, v, normals):
    L = lib()
    umin, umax, nu, vmin, vmax, nv, _ = bbs
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64)
    nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    n = u.shape[0]
    M = np.zeros((2 * n, nu * nv))
    D = C.c_double
    L.sfn_oracle_rows(D(umin), D(umax), int(nu), D(vmin), D(vmax), int(nv), n, _p(u, D), _p(v, D), _p(nrm, C.c_float), _p(M, D))
    return M


def sfn_estimate(bbs, u, v, normals, bending_weight, mean_depth, u_all, v_all):
    """ShapeFromNormals::estimate: returns (ok, ctrl_raw[N], ctrl[N], pts[n_all, 3] float32)."""
    L = lib()
    umin, umax, nu, vmin, vmax, nv, _ = bbs
    u = np.ascontiguousarray(u, np.float64); v = np.ascontiguousarray(v, np.float64)
    ua = np.ascontiguousarray(u_all, np.float64); va = np.ascontiguousarray(v_all, np.float64)
    nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    N = nu * nv
    raw = np.zeros(N); ctrl = np.zeros(N)
    pts = np.zeros((ua.shape[0], 3), np.float32)
    D = C.c_double
    L.sfn_oracle_estimate.restype = C.c_int
    ok = L.sfn_oracle_estimate(D(umin), D(umax), int(nu), D(vmin), D(vmax), int(nv), u.shape[0], _p(u, D), _p(v, D), _p(nrm, C.c_float), D(bending_weight),
                               D(mean_depth), ua.shape[0], _p(ua, D), _p(va, D), _p(raw, D), _p(ctrl, D), _p(pts, C.c_float))
    return bool(ok), raw, ctrl, pts


def warp_initialize(bbs, kp1, kp2, lam):
    """Warps::Warp::initialize: returns (ok, x[2N])."""
    L = lib()
    umin, umax, nu, vmin, vmax, nv, _ = bbs
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    x = np.zeros(2 * nu * nv)
    D = C.c_double
    L.warp_oracle_initialize.restype = C.c_int
    ok = L.warp_oracle_initialize(D(umin), D(umax), int(nu), D(vmin), D(vmax), int(nv), kp1.shape[0], _p(kp1, C.c_float), _p(kp2, C.c_float), D(lam), _p(x, D))
    return bool(ok), x


# ---- warp-guided match search oracle (match_oracle.c) -------------------------------------------------
def search_by_schwarp(bbs, x, kp1, desc1, cam2, bounds2, kp2, desc2, has_mp2, radius=2.0, th_low=50, grid=(64, 48)):
    """DefORBmatcher::searchBySchwarp: returns match[Q] (index in keyframe 2 or -1)."""
    L = lib()
    umin, umax, nu, vmin, vmax, nv, _ = bbs
    x = np.ascontiguousarray(x, np.float64)
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    cam2 = np.ascontiguousarray(cam2, np.float32)
    b2 = np.ascontiguousarray(bounds2, np.float32)
    mp2 = np.ascontiguousarray(has_mp2, np.uint8)
    match = np.full(kp1.shape[0], -1, np.int32)
    D = C.c_double
    L.match_oracle_search_by_schwarp.restype = C.c_int
    n = L.match_oracle_search_by_schwarp(D(umin), D(umax), int(nu), D(vmin), D(vmax), int(nv), _p(x, D), kp1.shape[0], _p(kp1, C.c_float), _p(d1, C.c_uint8),
                                         _p(cam2, C.c_float), _p(b2, C.c_float), int(grid[0]), int(grid[1]), kp2.shape[0], _p(kp2, C.c_float), _p(d2, C.c_uint8),
                                         _p(mp2, C.c_uint8), C.c_float(radius), int(th_low), _p(match, C.c_int32))
    assert n == int((match >= 0).sum())
    return match


# ---- surface registration oracle (horn_oracle.c) -------------------------------------------------------
def sim3_exp(u):
    """g2o Sim3(update): returns [qx qy qz qw tx ty tz s]."""
    L = lib()
    u = np.ascontiguousarray(u, np.float64)
    out = np.zeros(8)
    L.horn_oracle_sim3_exp(_p(u, C.c_double), _p(out, C.c_double))
    return out


def horn_system(pts1, pts2, sim3, huber=0.01):
    """Normal equations of one OptimizeHorn linearisation: (H 7x7, b 7, robust chi2)."""
    L = lib()
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 3)
    p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 3)
    s = np.ascontiguousarray(sim3, np.float64)
    H = np.zeros((7, 7)); b = np.zeros(7); chi = C.c_double()
    L.horn_oracle_system(p1.shape[0], _p(p1, C.c_float), _p(p2, C.c_float), _p(s, C.c_double), C.c_double(huber), _p(H, C.c_double), _p(b, C.c_double),
                         C.byref(chi))
    return H.T.copy(), b, chi.value


def scale_min_median(mono, stereo, u):
    """GroundTruthTools::scaleMinMedian with the rand() stream given as `u`: dict(scale, consumed, status, medians)."""
    L = lib()
    m = np.ascontiguousarray(mono, np.float32).reshape(-1, 3)
    s = np.ascontiguousarray(stereo, np.float32).reshape(-1, 3)
    u = np.ascontiguousarray(u, np.float64)
    consumed = C.c_int32(); status = C.c_int32()
    med = np.zeros(m.shape[0], np.float32)
    L.horn_oracle_scale_min_median.restype = C.c_float
    sc = L.horn_oracle_scale_min_median(m.shape[0], _p(m, C.c_float), _p(s, C.c_float), _p(u, C.c_double), u.shape[0], C.byref(consumed), C.byref(status),
                                        _p(med, C.c_float))
    return dict(scale=float(sc), consumed=consumed.value, status=status.value, medians=med)


def optimize_horn(pts1, pts2, sim3_init, chi, huber=0.01, device_sum_order=False):
    """Optimizer::OptimizeHorn: dict(ok, sim3, chi2, count, iters[2], trials[2]).  device_sum_order: add the edges' terms
    in the fixed tree of the device kernel instead of edge order (same numbers to rounding; used to compare trajectories)."""
    L = lib()
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 3)
    p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 3)
    s = np.array(sim3_init, np.float64).copy()
    chi2 = C.c_double(); count = C.c_int32()
    iters = np.zeros(2, np.int32); trials = np.zeros(2, np.int32)
    L.horn_oracle_optimize.restype = C.c_int
    ok = L.horn_oracle_optimize(p1.shape[0], _p(p1, C.c_float), _p(p2, C.c_float), _p(s, C.c_double), C.c_double(chi), C.c_double(huber), int(bool(device_sum_order)),
                                C.byref(chi2), C.byref(count), _p(iters, C.c_int32), _p(trials, C.c_int32))
    return dict(ok=bool(ok), sim3=s, chi2=chi2.value, count=count.value, iters=iters, trials=trials)


def horn_compose(sim3, Twc):
    """Tail of SurfaceRegistration::registerSurfaces: (s22, new Tcw float32 4x4)."""
    L = lib()
    s = np.ascontiguousarray(sim3, np.float64)
    T = np.ascontiguousarray(Twc, np.float32).reshape(16)
    s22 = C.c_double(); out = np.zeros(16, np.float32)
    L.horn_oracle_compose(_p(s, C.c_double), _p(T, C.c_float), C.byref(s22), _p(out, C.c_float))
    return s22.value, out.reshape(4, 4)
