"""Host-side mirror of the surface registration entry points, on top of the C ABI.

Reference interfaces being mirrored:
  * defSLAM::GroundTruthTools::scaleMinMedian(PosMono, PosStereo)        (Modules/GroundTruth/GroundTruthCalculator.cc:54)
  * defSLAM::Optimizer::OptimizeHorn(pts1, pts2, g2oS12, chi, huber)     (Modules/Tracking/DefOptimizer.cc:840)
  * defSLAM::SurfaceRegistration::registerSurfaces()                     (Modules/Mapping/SurfaceRegistration.cc:48)
The reference's rand() draws are an explicit input (`u`, uniform numbers in the order the reference consumes them).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .sft import Context, _ptr


def scaleMinMedian(ctx: Context, pos_mono, pos_stereo, u):
    """Returns dict(scale, consumed, status)."""
    m = np.ascontiguousarray(pos_mono, np.float32).reshape(-1, 3)
    s = np.ascontiguousarray(pos_stereo, np.float32).reshape(-1, 3)
    u = np.ascontiguousarray(u, np.float64)
    scale = C.c_float()
    consumed = C.c_int64()
    status = C.c_int32()
    ctx._check(ctx._L.dsh_scale_min_median(ctx._h, m.shape[0], _ptr(m, C.c_float), _ptr(s, C.c_float), _ptr(u, C.c_double), u.shape[0], C.byref(scale),
                                           C.byref(consumed), C.byref(status)), "dsh_scale_min_median")
    return dict(scale=float(scale.value), consumed=int(consumed.value), status=int(status.value))


def OptimizeHorn(ctx: Context, pts1, pts2, sim3, chi: float, huber: float = 0.01):
    """sim3 = [qx qy qz qw tx ty tz s].  Returns dict(ok, sim3, chi2, count, iters[2], trials[2])."""
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 3)
    p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 3)
    s = np.array(sim3, np.float64).copy()
    ok = C.c_int32()
    info = np.zeros(6)
    ctx._check(ctx._L.dsh_optimize_horn(ctx._h, p1.shape[0], _ptr(p1, C.c_float), _ptr(p2, C.c_float), _ptr(s, C.c_double), float(chi), float(huber),
                                        C.byref(ok), _ptr(info, C.c_double)), "dsh_optimize_horn")
    return dict(ok=bool(ok.value), sim3=s, chi2=info[0], count=int(info[1]), iters=info[2:4].astype(np.int32), trials=info[4:6].astype(np.int32))


def registerSurfaces(ctx: Context, cloud_surface, cloud_map, u, Twc, chi_limit: float, check_chi: bool = True):
    """Returns dict(registered, sim3, s22, Tcw, scale0, chi2, count, iters, trials, acceptable)."""
    a = np.ascontiguousarray(cloud_surface, np.float32).reshape(-1, 3)
    b = np.ascontiguousarray(cloud_map, np.float32).reshape(-1, 3)
    u = np.ascontiguousarray(u, np.float64)
    T = np.ascontiguousarray(Twc, np.float32).reshape(16)
    reg = C.c_int32()
    sim3 = np.zeros(8)
    s22 = C.c_double()
    Tcw = np.zeros(16, np.float32)
    info = np.zeros(8)
    ctx._check(ctx._L.dsh_surface_register(ctx._h, a.shape[0], _ptr(a, C.c_float), _ptr(b, C.c_float), _ptr(u, C.c_double), u.shape[0], _ptr(T, C.c_float),
                                           float(chi_limit), int(bool(check_chi)), C.byref(reg), _ptr(sim3, C.c_double), C.byref(s22), _ptr(Tcw, C.c_float),
                                           _ptr(info, C.c_double)), "dsh_surface_register")
    return dict(registered=bool(reg.value), sim3=sim3, s22=float(s22.value), Tcw=Tcw.reshape(4, 4), scale0=float(info[0]), chi2=info[1], count=int(info[2]),
                iters=info[3:5].astype(np.int32), trials=info[5:7].astype(np.int32), acceptable=bool(info[7]))
