"""Host-side mirror of the reference's SfT operator interface, on top of the C ABI.

Reference interface being mirrored (Modules/Tracking/DefOptimizer.h:51-53):

    int defSLAM::Optimizer::DefPoseOptimization(Frame* pFrame, Map* mMap, double RegLap = 5000,
                                                double RegInex = 5000, double RegTemp = 0,
                                                uint NeighboursLayers = 1);

`DefPoseOptimization(ctx, frame, ...)` below has the same argument meaning, the same
return value (inliers) and the same in-place side effects (frame pose, outlier flags,
repError, node positions, map-point positions).  The ORB_SLAM2 Frame/Map objects are
replaced by the flat `Frame` / `Context.template` views that a C++ shim would fill
(INTEGRATION.md).  All compute happens in libdefslam_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib


class DshError(RuntimeError):
    pass


def _ptr(a: Optional[np.ndarray], ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


@dataclass
class Frame:
    """The fields of ORB_SLAM2::Frame / DefMap the optimiser reads and writes."""
    Tcw: np.ndarray                 # (4,4) float32, pFrame->mTcw (in/out)
    K: np.ndarray                   # fx, fy, cx, cy
    N: int                          # pFrame->N
    obs_nodes: np.ndarray           # (M,3) int32  facet nodes (ascending) of each matched map point
    obs_bary: np.ndarray            # (M,3) float64 DefMapPoint::b1,b2,b3
    obs_uv: np.ndarray              # (M,2) float64 pFrame->mvKeysUn[i].pt
    obs_invsig2: np.ndarray         # (M,)  float64 pFrame->mvInvLevelSigma2[octave]
    nodes_xyz: np.ndarray           # (n,3) float64 Node::x,y,z (in/out)
    mvbOutlier: np.ndarray = field(default=None)   # (M,) bool, written
    repError: float = 0.0           # written
    mappoints: np.ndarray = field(default=None)    # (M,3) float32, written (RecalculatePosition)
    # diagnostics (not part of the reference interface)
    pose7: np.ndarray = field(default=None)
    chi2_obs: np.ndarray = field(default=None)
    iters: int = 0
    trials: int = 0
    dim: int = 0
    half_bandwidth: int = 0
    status: int = 0
    trace: np.ndarray = field(default=None)


class Context:
    """One GPU context (dsh_ctx): owns the template and the device buffers."""

    def __init__(self, device: int = 0, lab: bool = False):
        """lab=True binds libdefslam_hip_lab.so (the product ABI + include/defslam_hip_debug.h) instead of the product library."""
        self.lab = bool(lab)
        self._L = _lib.load_lab() if lab else _lib.load()
        h = C.c_void_p()
        rc = self._L.dsh_create(C.byref(h), device)
        if rc != _lib.DSH_OK:
            raise DshError(f"dsh_create failed with status {rc} (no gfx950 device visible?)")
        self._h = h
        self.n = 0
        self._keep = None

    def close(self):
        if getattr(self, "_events", None) is not None:
            self._events.close()
            self._events = None
        if getattr(self, "_h", None):
            self._L.dsh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != _lib.DSH_OK:
            raise DshError(f"{what}: status {rc}: {self._L.dsh_last_error(self._h).decode()}")

    # ---- template ---------------------------------------------------------------------------
    def template_build(self, xyz0: np.ndarray, facets: np.ndarray):
        xyz0 = np.ascontiguousarray(xyz0, np.float64)
        facets = np.ascontiguousarray(facets, np.int32)
        self._check(self._L.dsh_template_build(self._h, xyz0.shape[0], _ptr(xyz0, C.c_double), facets.shape[0], _ptr(facets, C.c_int32)),
                    "dsh_template_build")
        self.n = xyz0.shape[0]

    def template_set(self, xyz0, boundary, nbr_ptr, nbr_idx, nbr_w, k0, edge_nodes, edge_L0, median_L):
        xyz0 = np.ascontiguousarray(xyz0, np.float64)
        boundary = np.ascontiguousarray(boundary, np.uint8)
        nbr_ptr = np.ascontiguousarray(nbr_ptr, np.int32)
        nbr_idx = np.ascontiguousarray(nbr_idx, np.int32)
        nbr_w = np.ascontiguousarray(nbr_w, np.float64)
        k0 = np.ascontiguousarray(k0, np.float64)
        edge_nodes = np.ascontiguousarray(edge_nodes, np.int32)
        edge_L0 = np.ascontiguousarray(edge_L0, np.float64)
        self._check(self._L.dsh_template_set(self._h, xyz0.shape[0], _ptr(xyz0, C.c_double), _ptr(boundary, C.c_uint8), _ptr(nbr_ptr, C.c_int32),
                                             _ptr(nbr_idx, C.c_int32), _ptr(nbr_w, C.c_double), _ptr(k0, C.c_double), edge_L0.shape[0],
                                             _ptr(edge_nodes, C.c_int32), _ptr(edge_L0, C.c_double), float(median_L)), "dsh_template_set")
        self.n = xyz0.shape[0]

    def template_get(self) -> dict:
        n, E, nnz = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.dsh_template_dims(self._h, C.byref(n), C.byref(E), C.byref(nnz)), "dsh_template_dims")
        out = dict(boundary=np.zeros(n.value, np.uint8), nbr_ptr=np.zeros(n.value + 1, np.int32), nbr_idx=np.zeros(nnz.value, np.int32),
                   nbr_w=np.zeros(nnz.value), k0=np.zeros(n.value), edge_nodes=np.zeros((E.value, 2), np.int32), edge_L0=np.zeros(E.value))
        med = C.c_double()
        self._check(self._L.dsh_template_get(self._h, _ptr(out["boundary"], C.c_uint8), _ptr(out["nbr_ptr"], C.c_int32), _ptr(out["nbr_idx"], C.c_int32),
                                             _ptr(out["nbr_w"], C.c_double), _ptr(out["k0"], C.c_double), _ptr(out["edge_nodes"], C.c_int32),
                                             _ptr(out["edge_L0"], C.c_double), C.byref(med)), "dsh_template_get")
        out["median_L"] = med.value
        return out

    def template_embed(self, pts: np.ndarray):
        pts = np.ascontiguousarray(pts, np.float32)
        P = pts.shape[0]
        fid = np.zeros(P, np.int32)
        nodes = np.zeros((P, 3), np.int32)
        bary = np.zeros((P, 3), np.float32)
        self._check(self._L.dsh_template_embed(self._h, P, _ptr(pts, C.c_float), _ptr(fid, C.c_int32), _ptr(nodes, C.c_int32), _ptr(bary, C.c_float)),
                    "dsh_template_embed")
        return fid, nodes, bary

    def template_embed_device(self, pts: np.ndarray):
        """Same as template_embed, on the GPU (one wavefront per point)."""
        pts = np.ascontiguousarray(pts, np.float32)
        P = pts.shape[0]
        fid = np.zeros(P, np.int32)
        nodes = np.zeros((P, 3), np.int32)
        bary = np.zeros((P, 3), np.float32)
        self._check(self._L.dsh_template_embed_device(self._h, P, _ptr(pts, C.c_float), _ptr(fid, C.c_int32), _ptr(nodes, C.c_int32), _ptr(bary, C.c_float)),
                    "dsh_template_embed_device")
        return fid, nodes, bary

    # ---- batched SfT ------------------------------------------------------------------------
    def _frame_c(self, f: Frame, reg_lap, reg_inex, reg_temp, layers, max_iters, keep: list) -> _lib.SftFrameC:
        Tcw = np.ascontiguousarray(f.Tcw, np.float32)
        nodes = np.ascontiguousarray(f.obs_nodes, np.int32)
        bary = np.ascontiguousarray(f.obs_bary, np.float64)
        uv = np.ascontiguousarray(f.obs_uv, np.float64)
        isg = np.ascontiguousarray(f.obs_invsig2, np.float64)
        xyz = np.ascontiguousarray(f.nodes_xyz, np.float64)
        keep += [Tcw, nodes, bary, uv, isg, xyz]
        fc = _lib.SftFrameC()
        fc.Tcw = _ptr(Tcw, C.c_float)
        for i in range(4):
            fc.K[i] = float(f.K[i])
        fc.n_frame = int(f.N)
        fc.M = int(nodes.shape[0])
        fc.obs_nodes = _ptr(nodes, C.c_int32)
        fc.obs_bary = _ptr(bary, C.c_double)
        fc.obs_uv = _ptr(uv, C.c_double)
        fc.obs_invsig2 = _ptr(isg, C.c_double)
        fc.xyz = _ptr(xyz, C.c_double)
        fc.reg_lap, fc.reg_inex, fc.reg_temp = float(reg_lap), float(reg_inex), float(reg_temp)
        fc.neighbour_layers = int(layers)
        fc.max_iters = int(max_iters)
        return fc

    def batch_upload(self, frames: Sequence[Frame], RegLap=5000.0, RegInex=5000.0, RegTemp=0.0, NeighboursLayers=1, max_iters=50):
        keep: list = []
        arr = (_lib.SftFrameC * len(frames))()
        for i, f in enumerate(frames):
            arr[i] = self._frame_c(f, RegLap, RegInex, RegTemp, NeighboursLayers, max_iters, keep)
        self._check(self._L.dsh_sft_batch_upload(self._h, len(frames), arr), "dsh_sft_batch_upload")
        self._frames = list(frames)
        self._max_iters = max_iters

    def batch_run(self):
        self._check(self._L.dsh_sft_batch_run(self._h), "dsh_sft_batch_run")

    def batch_run_timed(self, launches: int = 1) -> float:
        """`launches` back-to-back runs bracketed by two HIP events recorded on the context's stream (the caller's own
        events through dsh_stream(): the product ABI has no timing entry point); returns milliseconds."""
        if getattr(self, "_events", None) is None:
            self._events = _lib.HipEvents()
        st = self.stream()
        self._events.start(st)
        for _ in range(int(launches)):
            self.batch_run()
        return self._events.stop_ms(st)

    # ---- lab build only (include/defslam_hip_debug.h) ---------------------------------------
    def _need_lab(self, what: str):
        if not self.lab:
            raise DshError(f"{what} is a lab entry point (include/defslam_hip_debug.h): create the context with Context(device, lab=True)")

    def set_option(self, name: str, value: int):
        self._need_lab("dsh_lab_set_option")
        self._check(self._L.dsh_lab_set_option(self._h, name.encode(), int(value)), "dsh_lab_set_option")

    def solver_info(self, b: int = 0) -> dict:
        """How problem b of the uploaded batch is solved (lab build): two-sided factorisation and its cut, lanes, tile mode."""
        self._need_lab("dsh_lab_sft_solver_info")
        o = (C.c_int32 * 8)()
        self._check(self._L.dsh_lab_sft_solver_info(self._h, int(b), o), "dsh_lab_sft_solver_info")
        return dict(split=int(o[0]), c0=int(o[1]), s=int(o[2]), n1p=int(o[3]), pad=int(o[4]), lanes=int(o[5]), tile_mode=int(o[6]), waves=int(o[7]))

    def lab_run_timed(self, launches: int = 1) -> float:
        self._need_lab("dsh_lab_sft_run_timed")
        ms = C.c_double()
        self._check(self._L.dsh_lab_sft_run_timed(self._h, int(launches), C.byref(ms)), "dsh_lab_sft_run_timed")
        return ms.value

    def batch_assemble_timed(self, launches: int = 1) -> float:
        """`launches` launches of one linearisation + normal-equation assembly per problem (measurement aid); milliseconds."""
        self._need_lab("dsh_lab_sft_assemble_timed")
        ms = C.c_double()
        self._check(self._L.dsh_lab_sft_assemble_timed(self._h, int(launches), C.byref(ms)), "dsh_lab_sft_assemble_timed")
        return ms.value

    def wave_check(self, rel: float = 1.0, launches: int = 1, only: int = 0):
        """A/B of the one-wavefront factorisation against the four-wavefront solver on the normal equations of every uploaded problem at
        its initial state (dsh_lab_sft_wave_check): (x_ref, x_new, ok[B, 2], ms[2]); x_*: list of per-problem solutions (Dnp + 6)."""
        self._need_lab("dsh_lab_sft_wave_check")
        B = len(self._frames)
        dims = []
        for b in range(B):
            _, counts = self.problem_info(b)
            Dn = int(counts[5]) - 6
            dims.append(((Dn + 31) // 32) * 32 + 6)
        tot = int(sum(dims))
        xr, xn = np.zeros(tot), np.zeros(tot)
        ok = np.zeros((B, 2), np.int32)
        ms = np.zeros(2)
        self._check(self._L.dsh_lab_sft_wave_check(self._h, C.c_double(rel), int(launches), int(only), _ptr(xr, C.c_double), _ptr(xn, C.c_double), _ptr(ok, C.c_int32),
                                                   _ptr(ms, C.c_double)), "dsh_lab_sft_wave_check")
        offs = np.concatenate([[0], np.cumsum(dims)])
        return [xr[offs[b]:offs[b + 1]] for b in range(B)], [xn[offs[b]:offs[b + 1]] for b in range(B)], ok, ms

    def rounds_timed(self):
        """One run of the uploaded batch as rounds of phase kernels with events around every launch (lab): (dict of total ms per phase, rounds)."""
        self._need_lab("dsh_lab_sft_rounds_timed")
        ms = np.zeros(7)
        r = np.zeros(1, np.int32)
        self._check(self._L.dsh_lab_sft_rounds_timed(self._h, _ptr(ms, C.c_double), _ptr(r, C.c_int32)), "dsh_lab_sft_rounds_timed")
        return dict(init=float(ms[0]), lin=float(ms[1]), factor=float(ms[2]), trial=float(ms[3]), tail=float(ms[4]), factorisations_in_rounds=int(ms[5]), linearisations_in_rounds=int(ms[6])), int(r[0])

    def dump(self, b: int, what: int, n: int):
        self._need_lab("dsh_lab_sft_dump")
        out = np.zeros(int(n))
        self._check(self._L.dsh_lab_sft_dump(self._h, int(b), int(what), int(n), _ptr(out, C.c_double)), "dsh_lab_sft_dump")
        return out

    def phase_ms(self, b: int = 0):
        self._need_lab("dsh_lab_sft_phase_ms")
        out = np.zeros(8)
        self._check(self._L.dsh_lab_sft_phase_ms(self._h, b, _ptr(out, C.c_double)), "dsh_lab_sft_phase_ms")
        return dict(trsm=out[0], residuals=out[1], assembly=out[2], copy=out[3], panel=out[4], update=out[5], backsub=out[6], control=out[7])

    def step_trace(self, b: int = 0):
        self._need_lab("dsh_lab_sft_step_trace")
        out = np.zeros(64)
        self._check(self._L.dsh_lab_sft_step_trace(self._h, b, _ptr(out, C.c_double)), "dsh_lab_sft_step_trace")
        return out.reshape(8, 8)

    def synchronize(self):
        self._check(self._L.dsh_synchronize(self._h), "dsh_synchronize")

    def stream(self) -> int:
        return int(self._L.dsh_stream(self._h) or 0)

    def batch_counts(self):
        it, tr = C.c_int64(), C.c_int64()
        self._check(self._L.dsh_sft_batch_counts(self._h, C.byref(it), C.byref(tr)), "dsh_sft_batch_counts")
        return it.value, tr.value

    def problem_info(self, b: int):
        nbytes = C.c_int64()
        counts = np.zeros(9, np.int32)
        self._check(self._L.dsh_sft_batch_problem_info(self._h, b, C.byref(nbytes), _ptr(counts, C.c_int32)), "dsh_sft_batch_problem_info")
        return nbytes.value, counts

    def batch_download(self, only=None) -> List[int]:
        """Write results back into the uploaded Frame objects (the reference's in-place mutations,
        DefOptimizer.cc:515-576) and return the per-frame inlier counts.  `only`: ids whose arrays are wanted (the C call skips
        output pointers that are null; counters and statistics come back for every problem)."""
        frames = self._frames
        res = (_lib.SftResultC * len(frames))()
        keep = []
        want = None if only is None else set(int(i) for i in only)
        for i, f in enumerate(frames):
            if want is not None and i not in want:
                keep.append(None)
                continue
            M, n = f.obs_nodes.shape[0], f.nodes_xyz.shape[0]
            bufs = dict(Tcw=np.zeros((4, 4), np.float32), pose7=np.zeros(7), xyz=np.zeros((n, 3)), chi2=np.zeros(M), outl=np.zeros(M, np.uint8),
                        mp=np.zeros((M, 3), np.float32), trace=np.zeros((max(self._max_iters, 1), _lib.DSH_TRACE_STRIDE)))
            keep.append(bufs)
            r = res[i]
            r.Tcw = _ptr(bufs["Tcw"], C.c_float)
            r.pose7 = _ptr(bufs["pose7"], C.c_double)
            r.xyz = _ptr(bufs["xyz"], C.c_double)
            r.chi2_obs = _ptr(bufs["chi2"], C.c_double)
            r.outlier = _ptr(bufs["outl"], C.c_uint8)
            r.mappoint_xyz = _ptr(bufs["mp"], C.c_float)
            r.trace = _ptr(bufs["trace"], C.c_double)
        self._check(self._L.dsh_sft_batch_download(self._h, len(frames), res), "dsh_sft_batch_download")
        out = []
        for i, f in enumerate(frames):
            b, r = keep[i], res[i]
            if b is None:
                f.iters, f.trials, f.dim, f.half_bandwidth, f.status = r.iters, r.trials, r.dim, r.half_bandwidth, r.status
                out.append(int(r.inliers))
                continue
            f.Tcw = b["Tcw"]
            f.pose7 = b["pose7"]
            f.nodes_xyz = b["xyz"]
            f.chi2_obs = b["chi2"]
            f.mvbOutlier = b["outl"].astype(bool)
            f.mappoints = b["mp"]
            f.repError = float(np.float32(r.rep_error))   # pFrame->repError is a float (Frame.h:213)
            f.rep_error_f64 = r.rep_error
            f.iters, f.trials, f.dim, f.half_bandwidth, f.status = r.iters, r.trials, r.dim, r.half_bandwidth, r.status
            f.trace = b["trace"][:r.iters].copy()
            out.append(int(r.inliers))
        return out

    def prepare_solve(self, f: Frame, RegLap=5000.0, RegInex=5000.0, RegTemp=0.0, NeighboursLayers=1, max_iters=50):
        """The one-shot ABI call dsh_sft_solve (pack + upload + run + download) with its C structs and output buffers built
        beforehand: the returned callable makes exactly one C call and then points the Frame's result fields at the
        buffers -- what bench.py times as the end-to-end frame."""
        keep: list = []
        fc = self._frame_c(f, RegLap, RegInex, RegTemp, NeighboursLayers, max_iters, keep)
        M, n = f.obs_nodes.shape[0], f.nodes_xyz.shape[0]
        b = dict(Tcw=np.zeros((4, 4), np.float32), pose7=np.zeros(7), xyz=np.zeros((n, 3)), chi2=np.zeros(M), outl=np.zeros(M, np.uint8),
                 mp=np.zeros((M, 3), np.float32), trace=np.zeros((max(max_iters, 1), _lib.DSH_TRACE_STRIDE)))
        r = _lib.SftResultC()
        r.Tcw = _ptr(b["Tcw"], C.c_float)
        r.pose7 = _ptr(b["pose7"], C.c_double)
        r.xyz = _ptr(b["xyz"], C.c_double)
        r.chi2_obs = _ptr(b["chi2"], C.c_double)
        r.outlier = _ptr(b["outl"], C.c_uint8)
        r.mappoint_xyz = _ptr(b["mp"], C.c_float)
        r.trace = _ptr(b["trace"], C.c_double)
        ctx = self

        class _Call:
            frame = f

            def __call__(self_inner) -> int:
                rc = ctx._L.dsh_sft_solve(ctx._h, C.byref(fc), C.byref(r))
                if rc != _lib.DSH_OK:
                    ctx._check(rc, "dsh_sft_solve")
                f.Tcw, f.pose7, f.nodes_xyz, f.chi2_obs, f.mappoints = b["Tcw"], b["pose7"], b["xyz"], b["chi2"], b["mp"]
                f.mvbOutlier = b["outl"].view(np.bool_)
                f.repError = float(np.float32(r.rep_error))
                f.rep_error_f64 = r.rep_error
                f.iters, f.trials, f.dim, f.half_bandwidth, f.status = r.iters, r.trials, r.dim, r.half_bandwidth, r.status
                f.trace = b["trace"][:r.iters]
                return int(r.inliers)

        call = _Call()
        call._keep = (keep, b, fc, r)
        return call

    def debug_system(self, b: int, D: int):
        self._need_lab("dsh_lab_sft_system")
        H = np.zeros((D, D), order="F")
        bv = np.zeros(D)
        chi = C.c_double()
        self._check(self._L.dsh_lab_sft_system(self._h, b, D, _ptr(H, C.c_double), _ptr(bv, C.c_double), C.byref(chi)), "dsh_lab_sft_system")
        return H, bv, chi.value


def DefPoseOptimization(ctx: Context, pFrame: Frame, RegLap: float = 5000, RegInex: float = 5000, RegTemp: float = 0,
                        NeighboursLayers: int = 1, max_iters: int = 50) -> int:
    """Shape-from-template with camera motion estimation for one frame (DefOptimizer.cc:251-578)."""
    ctx.batch_upload([pFrame], RegLap, RegInex, RegTemp, NeighboursLayers, max_iters)
    ctx.batch_run()
    return ctx.batch_download()[0]


def DefPoseOptimizationBatch(ctx: Context, frames: Sequence[Frame], RegLap: float = 5000, RegInex: float = 5000, RegTemp: float = 0,
                             NeighboursLayers: int = 1, max_iters: int = 50) -> List[int]:
    """Independent problems (different frames / keyframes against the same template) in one launch."""
    ctx.batch_upload(frames, RegLap, RegInex, RegTemp, NeighboursLayers, max_iters)
    ctx.batch_run()
    return ctx.batch_download()


# ---- shared-camera mode across GPUs (include/defslam_hip.h: dsh_comm_*, dsh_sft_shared_solve*) -------------------------
def _result_buffers(f: Frame, max_iters: int):
    M, n = f.obs_nodes.shape[0], f.nodes_xyz.shape[0]
    b = dict(Tcw=np.zeros((4, 4), np.float32), pose7=np.zeros(7), xyz=np.zeros((n, 3)), chi2=np.zeros(M), outl=np.zeros(M, np.uint8),
             mp=np.zeros((M, 3), np.float32), trace=np.zeros((max(max_iters, 1), _lib.DSH_TRACE_STRIDE)))
    r = _lib.SftResultC()
    r.Tcw = _ptr(b["Tcw"], C.c_float)
    r.pose7 = _ptr(b["pose7"], C.c_double)
    r.xyz = _ptr(b["xyz"], C.c_double)
    r.chi2_obs = _ptr(b["chi2"], C.c_double)
    r.outlier = _ptr(b["outl"], C.c_uint8)
    r.mappoint_xyz = _ptr(b["mp"], C.c_float)
    r.trace = _ptr(b["trace"], C.c_double)
    return b, r


def _write_back(f: Frame, b: dict, r) -> int:
    f.Tcw, f.pose7, f.nodes_xyz, f.chi2_obs, f.mappoints = b["Tcw"], b["pose7"], b["xyz"], b["chi2"], b["mp"]
    f.mvbOutlier = b["outl"].astype(bool)
    f.repError = float(np.float32(r.rep_error))
    f.rep_error_f64 = r.rep_error
    f.iters, f.trials, f.dim, f.half_bandwidth, f.status = r.iters, r.trials, r.dim, r.half_bandwidth, r.status
    f.trace = b["trace"][:r.iters].copy()
    return int(r.inliers)


def comm_unique_id() -> bytes:
    """ncclGetUniqueId: call on one rank and hand the bytes to every rank."""
    buf = C.create_string_buffer(_lib.DSH_COMM_ID_BYTES)
    if _lib.load().dsh_comm_unique_id(buf) != _lib.DSH_OK:
        raise DshError("dsh_comm_unique_id failed (RCCL not available?)")
    return buf.raw


class Comm:
    """An RCCL communicator on a context's GPU (collective over the ranks: one process per GPU)."""

    def __init__(self, ctx: Context, nranks: int, rank: int, unique_id: bytes):
        self._ctx = ctx
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), _lib.DSH_COMM_ID_BYTES)
        ctx._check(ctx._L.dsh_comm_create(ctx._h, int(nranks), int(rank), buf, C.byref(h)), "dsh_comm_create")
        self._h = h
        self.nranks, self.rank = int(nranks), int(rank)

    def close(self):
        if getattr(self, "_h", None):
            self._ctx._L.dsh_comm_destroy(self._h)
            self._h = None


def SharedCameraPoseOptimization(ctx: Context, comm: Comm, pFrame: Frame, RegLap: float = 5000, RegInex: float = 5000, RegTemp: float = 0,
                                 NeighboursLayers: int = 1, max_iters: int = 50) -> int:
    """Collective over the communicator's ranks: every rank passes its own patch (template in `ctx`, observations in `pFrame`), all
    patches are seen by one camera whose pose is estimated jointly (one all-reduce of the camera block per damping trial)."""
    keep: list = []
    fc = ctx._frame_c(pFrame, RegLap, RegInex, RegTemp, NeighboursLayers, max_iters, keep)
    b, r = _result_buffers(pFrame, max_iters)
    ctx._check(ctx._L.dsh_sft_shared_solve(ctx._h, comm._h, C.byref(fc), C.byref(r)), "dsh_sft_shared_solve")
    return _write_back(pFrame, b, r)


def SharedCameraPoseOptimizationGroup(ctxs: Sequence[Context], frames: Sequence[Frame], RegLap: float = 5000, RegInex: float = 5000, RegTemp: float = 0,
                                      NeighboursLayers: int = 1, max_iters: int = 50) -> List[int]:
    """The shared-camera protocol inside one process over len(ctxs) contexts (the all-reduce is a summation kernel)."""
    G = len(ctxs)
    keep: list = []
    fcs = (_lib.SftFrameC * G)()
    res = (_lib.SftResultC * G)()
    bufs = []
    for g in range(G):
        fcs[g] = ctxs[g]._frame_c(frames[g], RegLap, RegInex, RegTemp, NeighboursLayers, max_iters, keep)
        b, r = _result_buffers(frames[g], max_iters)
        bufs.append(b)
        res[g] = r
    hs = (C.c_void_p * G)(*[c._h for c in ctxs])
    ctxs[0]._check(ctxs[0]._L.dsh_sft_shared_solve_group(G, hs, fcs, res), "dsh_sft_shared_solve_group")
    return [_write_back(frames[g], bufs[g], res[g]) for g in range(G)]


def ConnectedPoseOptimization(ctx: Context, comm: Comm, pFrame: Frame, RegLap: float = 5000, RegInex: float = 5000, RegTemp: float = 0,
                              NeighboursLayers: int = 1, max_iters: int = 50) -> int:
    """Collective over a two-rank communicator: ONE problem on ONE connected template (both ranks pass the same frame); rank g factors
    part g of the band, the separator + camera system is all-reduced (include/defslam_hip.h: dsh_sft_connected_solve)."""
    keep: list = []
    fc = ctx._frame_c(pFrame, RegLap, RegInex, RegTemp, NeighboursLayers, max_iters, keep)
    b, r = _result_buffers(pFrame, max_iters)
    ctx._check(ctx._L.dsh_sft_connected_solve(ctx._h, comm._h, C.byref(fc), C.byref(r)), "dsh_sft_connected_solve")
    return _write_back(pFrame, b, r)


def ConnectedPoseOptimizationGroup(ctx0: Context, ctx1: Context, frames: Sequence[Frame], RegLap: float = 5000, RegInex: float = 5000, RegTemp: float = 0,
                                   NeighboursLayers: int = 1, max_iters: int = 50) -> List[int]:
    """The connected-mesh protocol inside one process over two contexts (the all-reduces are summation kernels).  frames: two Frame objects
    holding the SAME problem (one per context, each receives its context's copy of the result)."""
    keep: list = []
    fc = ctx0._frame_c(frames[0], RegLap, RegInex, RegTemp, NeighboursLayers, max_iters, keep)
    res = (_lib.SftResultC * 2)()
    bufs = []
    for g in range(2):
        b, r = _result_buffers(frames[g], max_iters)
        bufs.append(b)
        res[g] = r
    ctx0._check(ctx0._L.dsh_sft_connected_solve_group(ctx0._h, ctx1._h, C.byref(fc), res), "dsh_sft_connected_solve_group")
    return [_write_back(frames[g], bufs[g], res[g]) for g in range(2)]


def two_sided_cut(Dn: int, kd: int):
    """The cut of the two-sided factorisation / the connected-mesh mode (dsh_api.cpp, SftPart): (c0, s, n1p, pad) -- part 0 = scalars
    [0, c0), separator [c0, c0 + s) with s = 16 ceil(kd / 16) >= the half-bandwidth, part 1 = the rest (reversed, behind `pad` identity
    scalars that align its separator rows to a tile boundary).  None when the band is too short to cut."""
    sT = -(-kd // 16)
    s = 16 * sT
    c0 = ((Dn - s) // 2 // 16) * 16
    n1 = Dn - s - c0
    if sT < 2 or c0 < 64 or n1 < 64:
        return None
    n1p = -(-n1 // 16) * 16
    return c0, s, n1p, n1p - n1


def frame_from_synth(fr) -> Frame:
    return Frame(Tcw=fr.Tcw.copy(), K=fr.K.copy(), N=fr.n_frame, obs_nodes=fr.obs_nodes, obs_bary=fr.obs_bary, obs_uv=fr.obs_uv,
                 obs_invsig2=fr.obs_invsig2, nodes_xyz=fr.xyz.copy())
