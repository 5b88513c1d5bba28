// C ABI of the NRSfM mapping-side entry points (include/defslam_hip.h): B-spline evaluation / colocation and
// the per-map-point normal solve.  One-shot calls: host buffers in, device kernels, host buffers out.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/defslam_hip.h"
#include "dsh_ctx.h"

extern "C" hipError_t nrsfm_launch_bbs_eval(double, double, int, double, double, int, int, const double*, const double*, const double*, int, int, int, double*,
                                            uint8_t*, hipStream_t);
extern "C" hipError_t nrsfm_launch_bbs_coloc(double, double, int, double, double, int, const double*, const double*, int, int, int, int32_t*, double*,
                                             int32_t*, hipStream_t);
extern "C" hipError_t nrsfm_launch_normals(int, int, const int32_t*, const int32_t*, const float*, const uint8_t*, const float*, const uint8_t*, const float*,
                                           const uint8_t*, const float*, double*, double*, double*, int32_t*, float*, float*, uint8_t*, int32_t*, hipStream_t);

namespace {

#define HIPCHK(c, call)                                                                                        \
  do {                                                                                                         \
    hipError_t e__ = (call);                                                                                   \
    if (e__ != hipSuccess) {   /* copies from local host buffers may be in flight: drain the stream before they go away */    \
      (void)hipStreamSynchronize((c)->stream);                                                                  \
      return dsh_fail(c, DSH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));                      \
    }                                                                                                           \
  } while (0)

// scoped device buffer
struct DevBuf {   // a slice of the context's scratch (dsh_ctx.h); nothing to free
  void* p = nullptr;
  hipError_t alloc(dsh_ctx_base* c, size_t bytes) { return c->scratch.take(bytes, &p); }
  template <class T> T* as() { return static_cast<T*>(p); }
};

int gpu_ready(dsh_ctx_base* c, const char* who) {
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, std::string(who) + ": host-only context, no GPU (there is no CPU fallback)");
  if (hipSetDevice(c->device) != hipSuccess) return dsh_fail(c, DSH_ERR_HIP, std::string(who) + ": hipSetDevice failed");
  c->scratch.reset();   // temporaries of this call come out of the context's scratch
  return DSH_OK;
}

bool bbs_ok(const dsh_bbs* b) { return b && b->nptsu >= 4 && b->nptsv >= 4 && b->valdim >= 1 && b->umax > b->umin && b->vmax > b->vmin; }

}  // namespace

extern "C" {

int dsh_bbs_eval(dsh_ctx* ctx, const dsh_bbs* bbs, const double* ctrl, const double* u, const double* v, int n, int du, int dv, double* val, uint8_t* outside) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  int rc = gpu_ready(c, "dsh_bbs_eval");
  if (rc != DSH_OK) return rc;
  if (!bbs_ok(bbs) || !ctrl || n < 0 || (n > 0 && (!u || !v || !val)) || du < 0 || du > 2 || dv < 0 || dv > 2) return dsh_fail(c, DSH_ERR_ARG, "dsh_bbs_eval: bad argument");
  if (n == 0) return DSH_OK;
  const size_t nctrl = (size_t)bbs->valdim * bbs->nptsu * bbs->nptsv;
  DevBuf dctrl, du_, dv_, dval, dout;
  HIPCHK(c, dctrl.alloc(c, 8 * nctrl)); HIPCHK(c, du_.alloc(c, 8 * (size_t)n)); HIPCHK(c, dv_.alloc(c, 8 * (size_t)n));
  HIPCHK(c, dval.alloc(c, 8 * (size_t)n * bbs->valdim)); HIPCHK(c, dout.alloc(c, n));
  HIPCHK(c, hipMemcpyAsync(dctrl.p, ctrl, 8 * nctrl, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(du_.p, u, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dv_.p, v, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, nrsfm_launch_bbs_eval(bbs->umin, bbs->umax, bbs->nptsu, bbs->vmin, bbs->vmax, bbs->nptsv, bbs->valdim, dctrl.as<double>(), du_.as<double>(),
                                  dv_.as<double>(), n, du, dv, dval.as<double>(), dout.as<uint8_t>(), c->stream));
  HIPCHK(c, hipMemcpyAsync(val, dval.p, 8 * (size_t)n * bbs->valdim, hipMemcpyDeviceToHost, c->stream));
  if (outside) HIPCHK(c, hipMemcpyAsync(outside, dout.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return DSH_OK;
}

int dsh_bbs_coloc(dsh_ctx* ctx, const dsh_bbs* bbs, const double* u, const double* v, int n, int du, int dv, int32_t* cols, double* w, int32_t* n_outside) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  int rc = gpu_ready(c, "dsh_bbs_coloc");
  if (rc != DSH_OK) return rc;
  if (!bbs_ok(bbs) || n < 0 || (n > 0 && (!u || !v || !cols || !w)) || du < 0 || du > 2 || dv < 0 || dv > 2) return dsh_fail(c, DSH_ERR_ARG, "dsh_bbs_coloc: bad argument");
  if (n_outside) *n_outside = 0;
  if (n == 0) return DSH_OK;
  DevBuf du_, dv_, dcols, dw, dcnt;
  HIPCHK(c, du_.alloc(c, 8 * (size_t)n)); HIPCHK(c, dv_.alloc(c, 8 * (size_t)n)); HIPCHK(c, dcols.alloc(c, 4 * 16 * (size_t)n)); HIPCHK(c, dw.alloc(c, 8 * 16 * (size_t)n));
  HIPCHK(c, dcnt.alloc(c, 4));
  HIPCHK(c, hipMemcpyAsync(du_.p, u, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dv_.p, v, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dcnt.p, 0, 4, c->stream));
  HIPCHK(c, nrsfm_launch_bbs_coloc(bbs->umin, bbs->umax, bbs->nptsu, bbs->vmin, bbs->vmax, bbs->nptsv, du_.as<double>(), dv_.as<double>(), n, du, dv,
                                   dcols.as<int32_t>(), dw.as<double>(), dcnt.as<int32_t>(), c->stream));
  HIPCHK(c, hipMemcpyAsync(cols, dcols.p, 4 * 16 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(w, dw.p, 8 * 16 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  int32_t cnt = 0;
  HIPCHK(c, hipMemcpyAsync(&cnt, dcnt.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_outside) *n_outside = cnt;
  return DSH_OK;
}

int dsh_normals_estimate(dsh_ctx* ctx, int P, const int32_t* rec_ptr, const dsh_diffprop* recs, const uint8_t* rec_is_ref, const float* rec_first_normal,
                         const uint8_t* rec_has_first_normal, const float* x0, const uint8_t* has_x0, const float* ref_uv, double* k1k2, double* cov,
                         int32_t* status, float* normal_ref, float* normal_rec, uint8_t* rec_written, int32_t* iters) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  int rc = gpu_ready(c, "dsh_normals_estimate");
  if (rc != DSH_OK) return rc;
  if (P < 0 || (P > 0 && (!rec_ptr || !x0 || !has_x0 || !ref_uv || !k1k2 || !status))) return dsh_fail(c, DSH_ERR_ARG, "dsh_normals_estimate: bad argument");
  if (P == 0) return DSH_OK;
  const int R = rec_ptr[P];
  if (R < 0 || rec_ptr[0] != 0) return dsh_fail(c, DSH_ERR_ARG, "dsh_normals_estimate: bad rec_ptr");
  for (int p = 0; p < P; p++)
    if (rec_ptr[p + 1] < rec_ptr[p]) return dsh_fail(c, DSH_ERR_ARG, "dsh_normals_estimate: rec_ptr not monotone");
  if (R > 0 && (!recs || !rec_is_ref || !rec_first_normal || !rec_has_first_normal)) return dsh_fail(c, DSH_ERR_ARG, "dsh_normals_estimate: null records");
  // SoA transpose of the reference's DiffProp records + owner point of every record
  constexpr int NF = 18;
  std::vector<float> soa((size_t)NF * (R > 0 ? R : 1));
  std::vector<int32_t> owner(R > 0 ? R : 1);
  for (int p = 0; p < P; p++)
    for (int r = rec_ptr[p]; r < rec_ptr[p + 1]; r++) owner[r] = p;
  for (int r = 0; r < R; r++) {
    const float* f = reinterpret_cast<const float*>(&recs[r]);
    for (int k = 0; k < NF; k++) soa[(size_t)k * R + r] = f[k];
  }
  DevBuf d_ptr, d_owner, d_rec, d_isref, d_fn, d_hfn, d_x0, d_hx0, d_uv, d_Q, d_k, d_cov, d_st, d_nref, d_nrec, d_wr, d_it;
  const size_t Rn = R > 0 ? R : 1;
  HIPCHK(c, d_ptr.alloc(c, 4 * (size_t)(P + 1))); HIPCHK(c, d_owner.alloc(c, 4 * Rn)); HIPCHK(c, d_rec.alloc(c, 4 * NF * Rn)); HIPCHK(c, d_isref.alloc(c, Rn));
  HIPCHK(c, d_fn.alloc(c, 8 * Rn)); HIPCHK(c, d_hfn.alloc(c, Rn)); HIPCHK(c, d_x0.alloc(c, 8 * (size_t)P)); HIPCHK(c, d_hx0.alloc(c, P)); HIPCHK(c, d_uv.alloc(c, 8 * (size_t)P));
  HIPCHK(c, d_Q.alloc(c, 8 * 20 * Rn)); HIPCHK(c, d_k.alloc(c, 16 * (size_t)P)); HIPCHK(c, d_cov.alloc(c, 32 * (size_t)P)); HIPCHK(c, d_st.alloc(c, 4 * (size_t)P));
  HIPCHK(c, d_nref.alloc(c, 12 * (size_t)P)); HIPCHK(c, d_nrec.alloc(c, 12 * Rn)); HIPCHK(c, d_wr.alloc(c, Rn)); HIPCHK(c, d_it.alloc(c, 4 * (size_t)P));
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(d_ptr.p, rec_ptr, 4 * (size_t)(P + 1), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_x0.p, x0, 8 * (size_t)P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_hx0.p, has_x0, P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_uv.p, ref_uv, 8 * (size_t)P, hipMemcpyHostToDevice, st));
  if (R > 0) {
    HIPCHK(c, hipMemcpyAsync(d_owner.p, owner.data(), 4 * Rn, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_rec.p, soa.data(), 4 * NF * Rn, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_isref.p, rec_is_ref, Rn, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_fn.p, rec_first_normal, 8 * Rn, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_hfn.p, rec_has_first_normal, Rn, hipMemcpyHostToDevice, st));
  }
  HIPCHK(c, hipMemsetAsync(d_cov.p, 0, 32 * (size_t)P, st));
  HIPCHK(c, hipMemsetAsync(d_nref.p, 0, 12 * (size_t)P, st));
  HIPCHK(c, hipMemsetAsync(d_nrec.p, 0, 12 * Rn, st));
  HIPCHK(c, nrsfm_launch_normals(P, R, d_ptr.as<int32_t>(), d_owner.as<int32_t>(), d_rec.as<float>(), d_isref.as<uint8_t>(), d_fn.as<float>(),
                                 d_hfn.as<uint8_t>(), d_x0.as<float>(), d_hx0.as<uint8_t>(), d_uv.as<float>(), d_Q.as<double>(), d_k.as<double>(),
                                 d_cov.as<double>(), d_st.as<int32_t>(), d_nref.as<float>(), d_nrec.as<float>(), d_wr.as<uint8_t>(), d_it.as<int32_t>(), st));
  HIPCHK(c, hipMemcpyAsync(k1k2, d_k.p, 16 * (size_t)P, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(status, d_st.p, 4 * (size_t)P, hipMemcpyDeviceToHost, st));
  if (cov) HIPCHK(c, hipMemcpyAsync(cov, d_cov.p, 32 * (size_t)P, hipMemcpyDeviceToHost, st));
  if (normal_ref) HIPCHK(c, hipMemcpyAsync(normal_ref, d_nref.p, 12 * (size_t)P, hipMemcpyDeviceToHost, st));
  if (iters) HIPCHK(c, hipMemcpyAsync(iters, d_it.p, 4 * (size_t)P, hipMemcpyDeviceToHost, st));
  if (R > 0 && normal_rec) HIPCHK(c, hipMemcpyAsync(normal_rec, d_nrec.p, 12 * Rn, hipMemcpyDeviceToHost, st));
  if (R > 0 && rec_written) HIPCHK(c, hipMemcpyAsync(rec_written, d_wr.p, Rn, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return DSH_OK;
}

}  // extern "C"
