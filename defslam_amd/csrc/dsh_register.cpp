// Host side of dsh_scale_min_median / dsh_optimize_horn / dsh_surface_register (include/defslam_hip.h):
// SurfaceRegistration::registerSurfaces (Modules/Mapping/SurfaceRegistration.cc:48-153) with its two numeric callees.
// The host only walks the caller's random stream (which points are candidates) and composes the 4x4 result; the
// residual lists, the medians, the Levenberg-Marquardt loop and its sums run in register_kernels.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/defslam_hip.h"
#include "dsh_ctx.h"

extern "C" hipError_t reg_scale_min_median(int, int, const float*, const float*, const double*, const int32_t*, const int64_t*, float*, double*, double*,
                                           hipStream_t);
extern "C" hipError_t reg_horn(int, const float*, const float*, const double*, double, double, double*, double*, hipStream_t);

namespace {
#define HIPCHK(c, call)                                                                                        \
  do {                                                                                                         \
    hipError_t e__ = (call);                                                                                   \
    if (e__ != hipSuccess) {   /* copies from local host buffers may be in flight: drain the stream before they go away */    \
      (void)hipStreamSynchronize((c)->stream);                                                                  \
      return dsh_fail(c, DSH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));                      \
    }                                                                                                           \
  } while (0)

constexpr int kMaxPairs = 8000;   // the finishing kernel keeps two floats per pair in LDS

struct Clouds {   // device copies of the two clouds (slices of the context scratch)
  float *a = nullptr, *b = nullptr;
};

int upload_clouds(dsh_ctx_base* c, int n, const float* a, const float* b, Clouds& d) {
  void* p = nullptr;
  HIPCHK(c, c->scratch.take(12 * (size_t)n, &p)); d.a = static_cast<float*>(p);
  HIPCHK(c, c->scratch.take(12 * (size_t)n, &p)); d.b = static_cast<float*>(p);
  HIPCHK(c, hipMemcpyAsync(d.a, a, 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d.b, b, 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  return DSH_OK;
}

// scaleMinMedian on device clouds; returns status through *status (0, 1, 2)
int scale_min_median_dev(dsh_ctx_base* c, int n, const float* d_mono, const float* d_stereo, const double* u, int64_t nu, float* scale,
                         int64_t* consumed, int32_t* status) {
  // which i are candidates and where their j-draws start (GroundTruthCalculator.cc:64-84)
  std::vector<int32_t> cand;
  std::vector<int64_t> off;
  int64_t k = 0;
  bool too_short = false;
  for (int i = 0; i < n; i++) {
    if (k >= nu) { too_short = true; break; }
    const double r_i = u[k++];
    if (r_i > 0.25) continue;
    if (k + (n - 1) > nu) { too_short = true; break; }
    cand.push_back(i);
    off.push_back(k);
    k += n - 1;
  }
  if (too_short) {
    *status = 1;
    *scale = 0.f;
    return dsh_fail(c, DSH_ERR_ARG, "dsh_scale_min_median: the uniform stream is shorter than the draws the reference makes");
  }
  if (consumed) *consumed = k;
  const int nc = (int)cand.size();
  hipStream_t st = c->stream;
  void *du = nullptr, *dc = nullptr, *doff = nullptr, *dmed = nullptr, *dsc = nullptr, *dout = nullptr;
  HIPCHK(c, c->scratch.take(8 * (size_t)(k > 0 ? k : 1), &du));
  HIPCHK(c, c->scratch.take(4 * (size_t)(nc + 1), &dc));
  HIPCHK(c, c->scratch.take(8 * (size_t)(nc + 1), &doff));
  HIPCHK(c, c->scratch.take(4 * (size_t)(nc + 1), &dmed));
  HIPCHK(c, c->scratch.take(8 * (size_t)(nc + 1), &dsc));
  HIPCHK(c, c->scratch.take(64, &dout));
  if (k > 0) HIPCHK(c, hipMemcpyAsync(du, u, 8 * (size_t)k, hipMemcpyHostToDevice, st));
  if (nc > 0) {
    HIPCHK(c, hipMemcpyAsync(dc, cand.data(), 4 * (size_t)nc, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(doff, off.data(), 8 * (size_t)nc, hipMemcpyHostToDevice, st));
  }
  HIPCHK(c, reg_scale_min_median(n, nc, d_mono, d_stereo, static_cast<double*>(du), static_cast<int32_t*>(dc), static_cast<int64_t*>(doff),
                                 static_cast<float*>(dmed), static_cast<double*>(dsc), static_cast<double*>(dout), st));
  double out[4];
  HIPCHK(c, hipMemcpyAsync(out, dout, sizeof out, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));   // also keeps cand/off alive until the copies are done
  *scale = (float)out[0];
  *status = (int32_t)out[1];
  return DSH_OK;
}

int optimize_horn_dev(dsh_ctx_base* c, int n, const float* d1, const float* d2, double* sim3, double chi, double huber, int32_t* acceptable,
                      double* info) {
  hipStream_t st = c->stream;
  void *ds = nullptr, *derr = nullptr, *dout = nullptr;
  HIPCHK(c, c->scratch.take(64, &ds));
  HIPCHK(c, c->scratch.take(24 * (size_t)(n > 0 ? n : 1), &derr));
  HIPCHK(c, c->scratch.take(128, &dout));
  HIPCHK(c, hipMemcpyAsync(ds, sim3, 64, hipMemcpyHostToDevice, st));
  const float delta_huber = (float)std::sqrt(huber);   // DefOptimizer.cc:869
  HIPCHK(c, reg_horn(n, d1, d2, static_cast<double*>(ds), chi, (double)delta_huber, static_cast<double*>(derr), static_cast<double*>(dout), st));
  double out[16];
  HIPCHK(c, hipMemcpyAsync(out, dout, sizeof out, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  std::memcpy(sim3, out, 64);
  *acceptable = out[14] != 0.0 ? 1 : 0;
  if (info) { info[0] = out[8]; info[1] = out[9]; info[2] = out[10]; info[3] = out[11]; info[4] = out[12]; info[5] = out[13]; }
  return DSH_OK;
}

// SurfaceRegistration.cc:132-150: mScw = [s R | t] as float32, Twc' = mScw * Twc, scale from the first row of the rotation
// block, new Tcw = inverse of the unscaled pose
void compose(const double* sim3, const float* Twc, double* s22_out, float* Tcw) {
  const double x = sim3[0], y = sim3[1], z = sim3[2], w = sim3[3], s = sim3[7];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  float S[16], T[16];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) S[4 * i + j] = (float)(s * R[3 * i + j]);
    S[4 * i + 3] = (float)sim3[4 + i];
  }
  S[12] = S[13] = S[14] = 0.f;
  S[15] = 1.f;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float acc = 0.f;
      for (int k = 0; k < 4; k++) acc += S[4 * i + k] * Twc[4 * k + j];
      T[4 * i + j] = acc;
    }
  float tt = 0.f;
  for (int k = 0; k < 3; k++) tt += T[k] * T[k];
  const double s22 = std::sqrt((double)tt);
  *s22_out = s22;
  float m[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m[3 * i + j] = T[4 * i + j] / (float)s22;
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[3] * m[8] - m[5] * m[6], c02 = m[3] * m[7] - m[4] * m[6];
  const float det = m[0] * c00 - m[1] * c01 + m[2] * c02;
  const float inv[9] = {c00 / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                        (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                        c02 / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
  for (int r = 0; r < 3; r++) {
    for (int cc = 0; cc < 3; cc++) Tcw[4 * r + cc] = inv[3 * r + cc];
    Tcw[4 * r + 3] = -(inv[3 * r] * T[3] + inv[3 * r + 1] * T[7] + inv[3 * r + 2] * T[11]);
  }
  Tcw[12] = Tcw[13] = Tcw[14] = 0.f;
  Tcw[15] = 1.f;
}

int enter(dsh_ctx_base* c, const char* what) {
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, std::string(what) + ": host-only context, no GPU (there is no CPU fallback)");
  if (hipSetDevice(c->device) != hipSuccess) return dsh_fail(c, DSH_ERR_HIP, std::string(what) + ": hipSetDevice failed");
  c->scratch.reset();
  return DSH_OK;
}
}  // namespace

extern "C" {

int dsh_scale_min_median(dsh_ctx* ctx, int n, const float* pos_mono, const float* pos_stereo, const double* u, int64_t nu, float* scale,
                         int64_t* consumed, int32_t* status) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (n <= 0 || n > kMaxPairs || !pos_mono || !pos_stereo || nu < 0 || (nu > 0 && !u) || !scale || !status)
    return dsh_fail(c, DSH_ERR_ARG, "dsh_scale_min_median: bad argument (1 <= n <= 8000)");
  int rc = enter(c, "dsh_scale_min_median");
  if (rc != DSH_OK) return rc;
  Clouds d;
  rc = upload_clouds(c, n, pos_mono, pos_stereo, d);
  if (rc != DSH_OK) return rc;
  return scale_min_median_dev(c, n, d.a, d.b, u, nu, scale, consumed, status);
}

int dsh_optimize_horn(dsh_ctx* ctx, int n, const float* pts1, const float* pts2, double* sim3, double chi, double huber, int32_t* acceptable,
                      double* info) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (n <= 0 || !pts1 || !pts2 || !sim3 || !acceptable || !(huber >= 0.0)) return dsh_fail(c, DSH_ERR_ARG, "dsh_optimize_horn: bad argument");
  int rc = enter(c, "dsh_optimize_horn");
  if (rc != DSH_OK) return rc;
  Clouds d;
  rc = upload_clouds(c, n, pts1, pts2, d);
  if (rc != DSH_OK) return rc;
  return optimize_horn_dev(c, n, d.a, d.b, sim3, chi, huber, acceptable, info);
}

int dsh_surface_register(dsh_ctx* ctx, int n, const float* cloud_surface, const float* cloud_map, const double* u, int64_t nu, const float* Twc,
                         double chi_limit, int check_chi, int32_t* registered, double* sim3, double* s22, float* Tcw_new, double* info) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (n < 0 || n > kMaxPairs || (n > 0 && (!cloud_surface || !cloud_map)) || nu < 0 || (nu > 0 && !u) || !Twc || !registered || !sim3 || !s22 || !Tcw_new)
    return dsh_fail(c, DSH_ERR_ARG, "dsh_surface_register: bad argument (n <= 8000)");
  *registered = 0;
  if (info) std::memset(info, 0, 8 * sizeof(double));
  if (n < 15) return DSH_OK;   // SurfaceRegistration.cc:108-109
  int rc = enter(c, "dsh_surface_register");
  if (rc != DSH_OK) return rc;
  Clouds d;
  rc = upload_clouds(c, n, cloud_surface, cloud_map, d);
  if (rc != DSH_OK) return rc;
  float scale = 0.f;
  int32_t status = 0;
  rc = scale_min_median_dev(c, n, d.a, d.b, u, nu, &scale, nullptr, &status);
  if (rc != DSH_OK) return rc;
  // g2o::Sim3(identity rotation, zero translation, scale): Quaterniond(Matrix3d::Identity()) = (0, 0, 0, 1)
  sim3[0] = sim3[1] = sim3[2] = 0.0; sim3[3] = 1.0;
  sim3[4] = sim3[5] = sim3[6] = 0.0; sim3[7] = (double)scale;
  int32_t acceptable = 0;
  double hi[6];
  rc = optimize_horn_dev(c, n, d.a, d.b, sim3, chi_limit * chi_limit, 0.01, &acceptable, hi);
  if (rc != DSH_OK) return rc;
  if (info) {
    info[0] = scale;
    for (int i = 0; i < 6; i++) info[1 + i] = hi[i];
    info[7] = acceptable;
  }
  if (!acceptable && check_chi) return DSH_OK;
  compose(sim3, Twc, s22, Tcw_new);
  *registered = 1;
  return DSH_OK;
}

}  // extern "C"
