// C ABI of the Schwarzian warp fit (include/defslam_hip.h: dsh_schwarp_fit, dsh_schwarp_eval).
// The trust-region loop (3 iterations in the reference, SchwarpDatabase.cc:211-222) is sequenced on the host;
// residuals, Jacobian, normal equations, the 2N x 2N Cholesky solve and the DiffProp extraction run on the GPU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/defslam_hip.h"
#include "dsh_ctx.h"

extern "C" hipError_t nrsfm_swp_eval(double, double, int, double, double, int, int, double, double, double, const float*, const float*, const float*,
                                     const double*, double*, double*, int, hipStream_t);
extern "C" hipError_t nrsfm_swp_loss(int, int, const double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_swp_normal(int, int, int, double*, double*, const double*, const double*, double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_swp_colscale(int, const double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_swp_solve(int, const double*, const double*, double, double*, double*, double*, double*, int, int, hipStream_t);
extern "C" int nrsfm_swp_solve_np(int);
extern "C" hipError_t nrsfm_swp_step(int, const double*, const double*, const double*, const double*, double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_swp_diffprop(double, double, int, double, double, int, int, const float*, const float*, const double*, float, float, float*,
                                         uint8_t*, hipStream_t);

namespace {
#define HIPCHK(c, call)                                                                                        \
  do {                                                                                                         \
    hipError_t e__ = (call);                                                                                   \
    if (e__ != hipSuccess) return dsh_fail(c, DSH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

struct DevBuf {   // a slice of the context's scratch (dsh_ctx.h); nothing to free
  void* p = nullptr;
  hipError_t alloc(dsh_ctx_base* c, size_t bytes) { return c->scratch.take(bytes, &p); }
  template <class T> T* as() { return static_cast<T*>(p); }
};

struct Fit {
  dsh_ctx_base* c;
  const dsh_bbs* b;
  int P, N, n2, m;
  double fxs, fys, lambda;
  DevBuf kp1, kp2, isg, x, xn, cs, g, dx, r, J, A, M, W, scal;
  hipStream_t st;

  int eval(const double* xdev, bool with_j) {
    HIPCHK(c, nrsfm_swp_eval(b->umin, b->umax, b->nptsu, b->vmin, b->vmax, b->nptsv, P, fxs, fys, lambda, kp1.as<float>(), kp2.as<float>(), isg.as<float>(),
                             xdev, r.as<double>(), J.as<double>(), with_j ? 1 : 0, st));
    HIPCHK(c, nrsfm_swp_loss(2 * P, m, r.as<double>(), scal.as<double>(), st));
    if (with_j) HIPCHK(c, nrsfm_swp_normal(2 * P, m, n2, J.as<double>(), r.as<double>(), cs.as<double>(), scal.as<double>(), A.as<double>(), g.as<double>(), st));
    return DSH_OK;
  }
  int scalars(double* out8) {
    HIPCHK(c, hipMemcpyAsync(out8, scal.p, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return DSH_OK;
  }
};

int setup(Fit& f, dsh_ctx_base* c, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, const float* invsig, double fxs, double fys, double lambda,
          const double* x) {
  f.c = c; f.b = bbs; f.P = P; f.N = bbs->nptsu * bbs->nptsv; f.n2 = 2 * f.N; f.m = 2 * P + 4 * f.N;
  f.fxs = fxs; f.fys = fys; f.lambda = lambda; f.st = c->stream;
  c->scratch.reset();
  HIPCHK(c, f.kp1.alloc(c, 8 * (size_t)P)); HIPCHK(c, f.kp2.alloc(c, 8 * (size_t)P)); HIPCHK(c, f.isg.alloc(c, 4 * (size_t)P));
  HIPCHK(c, f.x.alloc(c, 8 * (size_t)f.n2)); HIPCHK(c, f.xn.alloc(c, 8 * (size_t)f.n2)); HIPCHK(c, f.cs.alloc(c, 8 * (size_t)f.n2)); HIPCHK(c, f.g.alloc(c, 8 * (size_t)f.n2));
  HIPCHK(c, f.dx.alloc(c, 8 * (size_t)f.n2)); HIPCHK(c, f.r.alloc(c, 8 * (size_t)f.m)); HIPCHK(c, f.J.alloc(c, 8 * (size_t)f.m * f.n2));
  HIPCHK(c, f.A.alloc(c, 8 * (size_t)f.n2 * f.n2)); {
    const size_t np = (size_t)nrsfm_swp_solve_np(f.n2);
    HIPCHK(c, f.M.alloc(c, 8 * np * np)); HIPCHK(c, f.W.alloc(c, 8 * np * 16));
  }
  HIPCHK(c, f.scal.alloc(c, 64));
  HIPCHK(c, hipMemcpyAsync(f.kp1.p, kp1, 8 * (size_t)P, hipMemcpyHostToDevice, f.st));
  HIPCHK(c, hipMemcpyAsync(f.kp2.p, kp2, 8 * (size_t)P, hipMemcpyHostToDevice, f.st));
  HIPCHK(c, hipMemcpyAsync(f.isg.p, invsig, 4 * (size_t)P, hipMemcpyHostToDevice, f.st));
  HIPCHK(c, hipMemcpyAsync(f.x.p, x, 8 * (size_t)f.n2, hipMemcpyHostToDevice, f.st));
  std::vector<double> ones(f.n2, 1.0);
  HIPCHK(c, hipMemcpyAsync(f.cs.p, ones.data(), 8 * (size_t)f.n2, hipMemcpyHostToDevice, f.st));
  HIPCHK(c, hipMemsetAsync(f.scal.p, 0, 64, f.st));
  HIPCHK(c, hipMemsetAsync(f.dx.p, 0, 8 * (size_t)f.n2, f.st));
  HIPCHK(c, hipStreamSynchronize(f.st));
  return DSH_OK;
}

bool args_ok(const dsh_bbs* b, int P, const float* kp1, const float* kp2, const float* invsig, const double* x) {
  return b && b->nptsu >= 4 && b->nptsv >= 4 && b->umax > b->umin && b->vmax > b->vmin && P > 0 && kp1 && kp2 && invsig && x && b->nptsu * b->nptsv <= 4096;
}
}  // namespace

extern "C" {

int dsh_schwarp_eval(dsh_ctx* ctx, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, const float* invsig, double fx_slot, double fy_slot,
                     double lambda, const double* x, double* residuals, double* jacobian) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_schwarp_eval: host-only context, no GPU (there is no CPU fallback)");
  if (!args_ok(bbs, P, kp1, kp2, invsig, x) || !residuals) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_eval: bad argument");
  (void)hipSetDevice(c->device);
  Fit f;
  int rc = setup(f, c, bbs, P, kp1, kp2, invsig, fx_slot, fy_slot, lambda, x);
  if (rc != DSH_OK) return rc;
  HIPCHK(c, nrsfm_swp_eval(bbs->umin, bbs->umax, bbs->nptsu, bbs->vmin, bbs->vmax, bbs->nptsv, P, fx_slot, fy_slot, lambda, f.kp1.as<float>(), f.kp2.as<float>(),
                           f.isg.as<float>(), f.x.as<double>(), f.r.as<double>(), f.J.as<double>(), jacobian ? 1 : 0, f.st));
  HIPCHK(c, hipMemcpyAsync(residuals, f.r.p, 8 * (size_t)f.m, hipMemcpyDeviceToHost, f.st));
  if (jacobian) HIPCHK(c, hipMemcpyAsync(jacobian, f.J.p, 8 * (size_t)f.m * f.n2, hipMemcpyDeviceToHost, f.st));
  HIPCHK(c, hipStreamSynchronize(f.st));
  return DSH_OK;
}

int dsh_schwarp_fit(dsh_ctx* ctx, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, const float* invsig, double fx_slot, double fy_slot,
                    double lambda, float fx, float fy, int max_iters, double* x, dsh_diffprop* diff, uint8_t* drop, int32_t* info, double* costs) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_schwarp_fit: host-only context, no GPU (there is no CPU fallback)");
  if (!args_ok(bbs, P, kp1, kp2, invsig, x) || max_iters < 0) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_fit: bad argument");
  if (bbs->nptsu * bbs->nptsv > 256) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_fit: more than 256 control points (the one-workgroup solve handles 2N <= 512 unknowns; the reference uses 13 x 15 = 195)");
  (void)hipSetDevice(c->device);
  Fit f;
  int rc = setup(f, c, bbs, P, kp1, kp2, invsig, fx_slot, fy_slot, lambda, x);
  if (rc != DSH_OK) return rc;
  // scal: [0] cost, [1] sqrt(rho'), [2] solve ok, [3] model cost change, [4] |step|, [5] |x|, [6] max |g|
  double s[8];
  double* scal = f.scal.as<double>();
  // Jacobi scaling from the initial Jacobian (cs = 1 first), then the scaled linearisation of the start
  if ((rc = f.eval(f.x.as<double>(), true)) != DSH_OK) return rc;
  HIPCHK(c, nrsfm_swp_colscale(f.n2, f.A.as<double>(), f.cs.as<double>(), f.st));
  if ((rc = f.eval(f.x.as<double>(), true)) != DSH_OK) return rc;
  HIPCHK(c, nrsfm_swp_step(f.n2, f.x.as<double>(), f.dx.as<double>(), f.cs.as<double>(), f.g.as<double>(), f.xn.as<double>(), scal + 2, f.st));
  if ((rc = f.scalars(s)) != DSH_OK) return rc;
  double cost = s[0];
  const double cost0 = cost;
  const double ftol = 1e-6, gtol = 1e-10, ptol = 1e-8, min_rel_dec = 1e-3;
  double radius = 1e4, nu = 2.0;
  int it = 0, good = 0, invalid = 0;
  // One host round trip per iteration: the trial point is evaluated speculatively right behind the solve (a failed solve or a
  // vanishing step only wastes that evaluation), and the gradient test of an accepted step -- `max |g| <= gtol` after the new
  // linearisation -- is read with the scalars of the next iteration (whose speculative work is dropped if it fires).
  bool pending_gtol = false;     // an accepted step was linearised; its max |g| has not been looked at yet
  if (s[6] > gtol)
    while (it < max_iters) {
      it++;
      HIPCHK(c, nrsfm_swp_solve(f.n2, f.A.as<double>(), f.g.as<double>(), radius, f.M.as<double>(), f.W.as<double>(), f.dx.as<double>(), scal + 2, 1, 2 * (3 * bbs->nptsv + 3) + 1, f.st));
      HIPCHK(c, nrsfm_swp_step(f.n2, f.x.as<double>(), f.dx.as<double>(), f.cs.as<double>(), f.g.as<double>(), f.xn.as<double>(), scal + 2, f.st));
      if ((rc = f.eval(f.xn.as<double>(), false)) != DSH_OK) return rc;   // residuals only: J, A, g still belong to x
      if ((rc = f.scalars(s)) != DSH_OK) return rc;
      if (pending_gtol && s[6] <= gtol) { it--; break; }                  // the previous iteration had already converged
      pending_gtol = false;
      const bool ok = s[2] != 0.0;
      const double model = s[3];
      if (!ok) { if (++invalid >= 5) break; radius *= 0.5; continue; }
      invalid = 0;
      if (s[4] <= ptol * (s[5] + ptol)) break;
      const double cost_new = s[0];
      const double rel = (cost - cost_new) / model;
      if (rel > min_rel_dec) {
        const double change = cost - cost_new, old = cost;
        HIPCHK(c, hipMemcpyAsync(f.x.p, f.xn.p, 8 * (size_t)f.n2, hipMemcpyDeviceToDevice, f.st));
        radius = std::fmin(1e16, radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
        nu = 2.0;
        good++;
        if ((rc = f.eval(f.x.as<double>(), true)) != DSH_OK) return rc;   // same residuals as the trial evaluation: cost == cost_new
        cost = cost_new;
        pending_gtol = true;
        if (std::fabs(change) <= ftol * old) break;
      } else {
        radius /= nu; nu *= 2.0;
        if (radius < 1e-32) break;
      }
    }
  if (info) { info[0] = it; info[1] = good; }
  if (costs) { costs[0] = cost0; costs[1] = cost; }
  HIPCHK(c, hipMemcpyAsync(x, f.x.p, 8 * (size_t)f.n2, hipMemcpyDeviceToHost, f.st));
  if (diff && drop) {
    DevBuf dd, dr;
    HIPCHK(c, dd.alloc(c, 72 * (size_t)P)); HIPCHK(c, dr.alloc(c, P));
    HIPCHK(c, nrsfm_swp_diffprop(bbs->umin, bbs->umax, bbs->nptsu, bbs->vmin, bbs->vmax, bbs->nptsv, P, f.kp1.as<float>(), f.kp2.as<float>(), f.x.as<double>(), fx,
                                 fy, dd.as<float>(), dr.as<uint8_t>(), f.st));
    HIPCHK(c, hipMemcpyAsync(diff, dd.p, 72 * (size_t)P, hipMemcpyDeviceToHost, f.st));
    HIPCHK(c, hipMemcpyAsync(drop, dr.p, P, hipMemcpyDeviceToHost, f.st));
    HIPCHK(c, hipStreamSynchronize(f.st));
  }
  HIPCHK(c, hipStreamSynchronize(f.st));
  return DSH_OK;
}

}  // extern "C"
