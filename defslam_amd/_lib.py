"""ctypes binding of libdefslam_hip.so (the C ABI in include/defslam_hip.h) and of the lab build
libdefslam_hip_lab.so (the same ABI + include/defslam_hip_debug.h: test hooks, timers, A/B solver switches).

Both are built in-tree by `__graft_entry__.build()` / `make -C defslam_amd/csrc`.
There is no CPU fallback: if the shared object is missing, importing a symbol raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The in-tree libraries, always: neither this binding nor the .so reads environment variables (tools that A/B two builds assign
# _lib.LIB_PATH / _lib.LAB_LIB_PATH before the first load()).
LIB_PATH = os.path.join(_HERE, "lib", "libdefslam_hip.so")
LAB_LIB_PATH = os.path.join(_HERE, "lib", "libdefslam_hip_lab.so")

DSH_OK = 0
DSH_TRACE_STRIDE = 8
DSH_MAX_ITERS = 64

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_i32_p = C.POINTER(C.c_int32)
c_u8_p = C.POINTER(C.c_uint8)


class SftFrameC(C.Structure):
    _fields_ = [
        ("Tcw", c_float_p),
        ("K", C.c_double * 4),
        ("n_frame", C.c_int32),
        ("M", C.c_int32),
        ("obs_nodes", c_i32_p),
        ("obs_bary", c_double_p),
        ("obs_uv", c_double_p),
        ("obs_invsig2", c_double_p),
        ("xyz", c_double_p),
        ("reg_lap", C.c_double),
        ("reg_inex", C.c_double),
        ("reg_temp", C.c_double),
        ("neighbour_layers", C.c_int32),
        ("max_iters", C.c_int32),
    ]


class SftResultC(C.Structure):
    _fields_ = [
        ("Tcw", c_float_p),
        ("pose7", c_double_p),
        ("xyz", c_double_p),
        ("chi2_obs", c_double_p),
        ("outlier", c_u8_p),
        ("mappoint_xyz", c_float_p),
        ("rep_error", C.c_double),
        ("inliers", C.c_int32),
        ("iters", C.c_int32),
        ("trials", C.c_int32),
        ("dim", C.c_int32),
        ("half_bandwidth", C.c_int32),
        ("status", C.c_int32),
        ("trace", c_double_p),
    ]


class SchwarpProblemC(C.Structure):
    pass


class BbsC(C.Structure):
    _fields_ = [("umin", C.c_double), ("umax", C.c_double), ("nptsu", C.c_int32), ("vmin", C.c_double), ("vmax", C.c_double),
                ("nptsv", C.c_int32), ("valdim", C.c_int32)]


# the pointer members are declared void*: the mirror fills them with plain addresses (base of a pooled array + offset), which costs a
# fraction of a typed ctypes pointer per field
SchwarpProblemC._fields_ = [("bbs", BbsC), ("P", C.c_int32), ("kp1", C.c_void_p), ("kp2", C.c_void_p), ("invsig", C.c_void_p), ("fx_slot", C.c_double),
                            ("fy_slot", C.c_double), ("lam", C.c_double), ("fx", C.c_float), ("fy", C.c_float), ("max_iters", C.c_int32), ("x", C.c_void_p),
                            ("diff", C.c_void_p), ("drop", C.c_void_p), ("info", C.c_int32 * 2), ("costs", C.c_double * 2), ("init_lambda", C.c_double), ("init_ok", C.c_int32)]



class SchwarpStoreC(C.Structure):
    _fields_ = [("point_id", C.c_void_p), ("idx2", C.c_void_p), ("tag", C.c_int32)]


DIFFPROP_FIELDS = ["I1u", "I1v", "I2u", "I2v", "J12a", "J12b", "J12c", "J12d", "J21a", "J21b", "J21c", "J21d",
                   "H12uux", "H12uuy", "H12uvx", "H12uvy", "H12vvx", "H12vvy"]


# Every symbol include/defslam_hip.h declares (checked by tests/test_abi.py).
EXPORTED_SYMBOLS = [
    "dsh_create", "dsh_destroy", "dsh_last_error", "dsh_stream", "dsh_synchronize",
    "dsh_template_build", "dsh_template_set", "dsh_template_dims", "dsh_template_get", "dsh_template_embed",
    "dsh_sft_solve", "dsh_sft_batch_upload", "dsh_sft_batch_run", "dsh_sft_batch_download",
    "dsh_sft_batch_counts", "dsh_sft_batch_problem_info",
    "dsh_bbs_eval", "dsh_bbs_coloc", "dsh_normals_estimate", "dsh_schwarp_eval", "dsh_schwarp_fit", "dsh_schwarp_fit_batch",
    "dsh_sfn_estimate", "dsh_bbs_bending", "dsh_warp_initialize", "dsh_search_by_schwarp",
    "dsh_template_embed_device", "dsh_scale_min_median", "dsh_optimize_horn", "dsh_surface_register",
    "dsh_comm_unique_id", "dsh_comm_create", "dsh_comm_destroy", "dsh_sft_shared_solve", "dsh_sft_shared_solve_group", "dsh_sft_connected_solve", "dsh_sft_connected_solve_group",
    "dsh_diffdb_create", "dsh_diffdb_destroy", "dsh_diffdb_clear", "dsh_diffdb_count", "dsh_diffdb_append", "dsh_schwarp_fit_batch_store",
    "dsh_normals_estimate_db", "dsh_sfn_estimate_db",
]
DSH_COMM_ID_BYTES = 128

# include/defslam_hip_debug.h: only libdefslam_hip_lab.so exports these
LAB_SYMBOLS = ["dsh_lab_set_option", "dsh_lab_sft_run_timed", "dsh_lab_sft_assemble_timed", "dsh_lab_sft_phase_ms", "dsh_lab_sft_step_trace",
               "dsh_lab_sft_system", "dsh_lab_sft_solver_info", "dsh_lab_sft_wave_check", "dsh_lab_sft_dump", "dsh_lab_sft_rounds_timed"]

_lib = None
_lab = None


def load() -> C.CDLL:
    """Load the product library; raises OSError when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        _lib = _bind(LIB_PATH, lab=False)
    return _lib


def load_lab() -> C.CDLL:
    """Load the lab build (product ABI + the measurement / debugging entry points)."""
    global _lab
    if _lab is None:
        _lab = _bind(LAB_LIB_PATH, lab=True)
    return _lab


def _bind(path: str, lab: bool) -> C.CDLL:
    if not os.path.exists(path):
        raise OSError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"(or `make -C defslam_amd/csrc`). There is no CPU fallback.")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.dsh_create.argtypes = [C.POINTER(vp), C.c_int]
    L.dsh_destroy.argtypes = [vp]
    L.dsh_last_error.argtypes = [vp]
    L.dsh_last_error.restype = C.c_char_p
    L.dsh_stream.argtypes = [vp]
    L.dsh_stream.restype = vp
    L.dsh_synchronize.argtypes = [vp]
    L.dsh_template_build.argtypes = [vp, C.c_int, c_double_p, C.c_int, c_i32_p]
    L.dsh_template_set.argtypes = [vp, C.c_int, c_double_p, c_u8_p, c_i32_p, c_i32_p, c_double_p, c_double_p, C.c_int, c_i32_p,
                                   c_double_p, C.c_double]
    L.dsh_template_dims.argtypes = [vp, c_i32_p, c_i32_p, c_i32_p]
    L.dsh_template_get.argtypes = [vp, c_u8_p, c_i32_p, c_i32_p, c_double_p, c_double_p, c_i32_p, c_double_p, c_double_p]
    L.dsh_template_embed.argtypes = [vp, C.c_int, c_float_p, c_i32_p, c_i32_p, c_float_p]
    L.dsh_sft_solve.argtypes = [vp, C.POINTER(SftFrameC), C.POINTER(SftResultC)]
    L.dsh_sft_batch_upload.argtypes = [vp, C.c_int, C.POINTER(SftFrameC)]
    L.dsh_sft_batch_run.argtypes = [vp]
    L.dsh_sft_batch_download.argtypes = [vp, C.c_int, C.POINTER(SftResultC)]
    L.dsh_sft_batch_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.dsh_sft_batch_problem_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), c_i32_p]
    L.dsh_bbs_eval.argtypes = [vp, C.POINTER(BbsC), c_double_p, c_double_p, c_double_p, C.c_int, C.c_int, C.c_int, c_double_p, c_u8_p]
    L.dsh_bbs_coloc.argtypes = [vp, C.POINTER(BbsC), c_double_p, c_double_p, C.c_int, C.c_int, C.c_int, c_i32_p, c_double_p, c_i32_p]
    L.dsh_normals_estimate.argtypes = [vp, C.c_int, c_i32_p, c_float_p, c_u8_p, c_float_p, c_u8_p, c_float_p, c_u8_p, c_float_p,
                                       c_double_p, c_double_p, c_i32_p, c_float_p, c_float_p, c_u8_p, c_i32_p]
    L.dsh_schwarp_eval.argtypes = [vp, C.POINTER(BbsC), C.c_int, c_float_p, c_float_p, c_float_p, C.c_double, C.c_double, C.c_double, c_double_p,
                                   c_double_p, c_double_p]
    L.dsh_schwarp_fit.argtypes = [vp, C.POINTER(BbsC), C.c_int, c_float_p, c_float_p, c_float_p, C.c_double, C.c_double, C.c_double, C.c_float,
                                  C.c_float, C.c_int, c_double_p, c_float_p, c_u8_p, c_i32_p, c_double_p]
    L.dsh_schwarp_fit_batch.argtypes = [vp, C.c_int, C.POINTER(SchwarpProblemC)]
    L.dsh_diffdb_create.argtypes = [vp, C.c_int64, C.POINTER(vp)]
    L.dsh_diffdb_destroy.argtypes = [vp]
    L.dsh_diffdb_clear.argtypes = [vp]
    L.dsh_diffdb_count.argtypes = [vp]
    L.dsh_diffdb_count.restype = C.c_int64
    L.dsh_diffdb_append.argtypes = [vp, C.c_int, c_float_p, c_i32_p, c_i32_p, c_i32_p]
    L.dsh_schwarp_fit_batch_store.argtypes = [vp, C.c_int, C.POINTER(SchwarpProblemC), C.POINTER(SchwarpStoreC), vp]
    L.dsh_normals_estimate_db.argtypes = [vp, vp, C.c_int, c_i32_p, c_float_p, c_u8_p, c_float_p, c_double_p, c_double_p, c_i32_p, c_float_p, c_i32_p,
                                          C.c_int32, c_i32_p, c_i32_p, c_i32_p, c_i32_p, c_float_p, c_u8_p]
    L.dsh_sfn_estimate.argtypes = [vp, C.POINTER(BbsC), C.c_int, c_double_p, c_double_p, c_float_p, C.c_double, C.c_double, C.c_int, c_double_p, c_double_p,
                                   c_double_p, c_double_p, c_float_p, c_i32_p]
    L.dsh_sfn_estimate_db.argtypes = [vp, C.POINTER(BbsC), vp, C.c_int, c_i32_p, c_double_p, c_double_p, C.c_double, C.c_double, C.c_int, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_float_p, c_i32_p]
    L.dsh_bbs_bending.argtypes = [C.POINTER(BbsC), C.c_double, c_double_p]
    L.dsh_search_by_schwarp.argtypes = [vp, C.POINTER(BbsC), c_double_p, C.c_int, c_float_p, c_u8_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int, c_float_p,
                                        c_u8_p, c_u8_p, C.c_float, C.c_int, c_i32_p, c_i32_p]
    L.dsh_warp_initialize.argtypes = [vp, C.POINTER(BbsC), C.c_int, c_float_p, c_float_p, C.c_double, c_double_p, c_i32_p]
    c_i64_p = C.POINTER(C.c_int64)
    L.dsh_template_embed_device.argtypes = [vp, C.c_int, c_float_p, c_i32_p, c_i32_p, c_float_p]
    L.dsh_scale_min_median.argtypes = [vp, C.c_int, c_float_p, c_float_p, c_double_p, C.c_int64, c_float_p, c_i64_p, c_i32_p]
    L.dsh_optimize_horn.argtypes = [vp, C.c_int, c_float_p, c_float_p, c_double_p, C.c_double, C.c_double, c_i32_p, c_double_p]
    L.dsh_surface_register.argtypes = [vp, C.c_int, c_float_p, c_float_p, c_double_p, C.c_int64, c_float_p, C.c_double, C.c_int, c_i32_p, c_double_p,
                                       c_double_p, c_float_p, c_double_p]
    L.dsh_comm_unique_id.argtypes = [vp]
    L.dsh_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.dsh_comm_destroy.argtypes = [vp]
    L.dsh_sft_shared_solve.argtypes = [vp, vp, C.POINTER(SftFrameC), C.POINTER(SftResultC)]
    L.dsh_sft_shared_solve_group.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(SftFrameC), C.POINTER(SftResultC)]
    L.dsh_sft_connected_solve.argtypes = [vp, vp, C.POINTER(SftFrameC), C.POINTER(SftResultC)]
    L.dsh_sft_connected_solve_group.argtypes = [vp, vp, C.POINTER(SftFrameC), C.POINTER(SftResultC)]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(L, name)
        if name not in ("dsh_last_error", "dsh_stream"):
            fn.restype = C.c_int
    if lab:
        L.dsh_lab_set_option.argtypes = [vp, C.c_char_p, C.c_int]
        L.dsh_lab_sft_run_timed.argtypes = [vp, C.c_int, c_double_p]
        L.dsh_lab_sft_assemble_timed.argtypes = [vp, C.c_int, c_double_p]
        L.dsh_lab_sft_phase_ms.argtypes = [vp, C.c_int, c_double_p]
        L.dsh_lab_sft_step_trace.argtypes = [vp, C.c_int, c_double_p]
        L.dsh_lab_sft_system.argtypes = [vp, C.c_int, C.c_int32, c_double_p, c_double_p, c_double_p]
        L.dsh_lab_sft_solver_info.argtypes = [vp, C.c_int, c_i32_p]
        L.dsh_lab_sft_wave_check.argtypes = [vp, C.c_double, C.c_int, C.c_int, c_double_p, c_double_p, c_i32_p, c_double_p]
        L.dsh_lab_sft_dump.argtypes = [vp, C.c_int, C.c_int, C.c_int64, c_double_p]
        L.dsh_lab_sft_rounds_timed.argtypes = [vp, c_double_p, c_i32_p]
        for name in LAB_SYMBOLS:
            getattr(L, name).restype = C.c_int
    return L


class HipEvents:
    """Two HIP events on a caller-chosen stream (libamdhip64 through ctypes): how bench.py times the launches of the
    PRODUCT library on the stream it launches on (dsh_stream) without any timing entry point in the ABI."""

    def __init__(self):
        self._hip = C.CDLL("libamdhip64.so")
        self._hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self._hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self._hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self._hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self._hip.hipEventDestroy.argtypes = [C.c_void_p]
        self.e0, self.e1 = C.c_void_p(), C.c_void_p()
        for e in (self.e0, self.e1):
            if self._hip.hipEventCreate(C.byref(e)) != 0:
                raise OSError("hipEventCreate failed")

    def start(self, stream: int):
        if self._hip.hipEventRecord(self.e0, C.c_void_p(stream)) != 0:
            raise OSError("hipEventRecord failed")

    def stop_ms(self, stream: int) -> float:
        ms = C.c_float()
        if (self._hip.hipEventRecord(self.e1, C.c_void_p(stream)) != 0 or self._hip.hipEventSynchronize(self.e1) != 0
                or self._hip.hipEventElapsedTime(C.byref(ms), self.e0, self.e1) != 0):
            raise OSError("HIP event timing failed")
        return float(ms.value)

    def close(self):
        for e in (self.e0, self.e1):
            if e:
                self._hip.hipEventDestroy(e)
        self.e0 = self.e1 = None
