// Host side of dsh_sfn_estimate / dsh_bbs_bending (include/defslam_hip.h): Shape from Normals,
// Modules/Mapping/ShapeFromNormals.cc.  The stacked least squares is solved on the device by corrected semi-normal
// equations: G = A^T A and A^T b on FP64 MFMA (swp_normal_kernel), tile Cholesky (swp_solve_kernel), two steps of
// iterative refinement with the residual formed from A itself, which restores the accuracy a QR factorisation has.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/defslam_hip.h"
#include "dsh_ctx.h"
#include "dsh_diffdb.h"

extern "C" hipError_t nrsfm_swp_normal(int, int, int, double*, double*, const double*, const double*, double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_swp_solve(int, const double*, const double*, double, double*, double*, double*, double*, int, int, hipStream_t);
extern "C" hipError_t nrsfm_swp_resolve(int, const double*, const double*, const double*, double*, int, int, hipStream_t);
extern "C" int nrsfm_swp_solve_np(int);
extern "C" hipError_t nrsfm_sfn_rows(double, double, int, double, double, int, int, const double*, const double*, const float*, double*, hipStream_t);
extern "C" hipError_t nrsfm_sfn_residual(int, int, const double*, const double*, const double*, double, double*, hipStream_t);
extern "C" hipError_t nrsfm_sfn_axpy(int, const double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_warp_coloc(double, double, int, double, double, int, int, const float*, const float*, double*, double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_mat_add(size_t, const double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_match_search(double, double, int, double, double, int, const double*, int, const float*, const uint32_t*, const float*, const float*, int, int,
                                         int, const float*, const uint32_t*, const uint8_t*, float, int, int32_t*, int32_t*, hipStream_t);
extern "C" hipError_t ddb_pick_normals(int, const int32_t*, const float*, const float*, float*, hipStream_t);
extern "C" hipError_t nrsfm_sfn_points(double, double, int, double, double, int, const double*, int, const double*, const double*, float*, hipStream_t);

namespace {
#define HIPCHK(c, call)                                                                                        \
  do {                                                                                                         \
    hipError_t e__ = (call);                                                                                   \
    if (e__ != hipSuccess) {   /* copies from local host buffers may be in flight: drain the stream before they go away */    \
      (void)hipStreamSynchronize((c)->stream);                                                                  \
      return dsh_fail(c, DSH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));                      \
    }                                                                                                           \
  } while (0)

struct DevBuf {   // a slice of the context's scratch (dsh_ctx.h); nothing to free
  void* p = nullptr;
  hipError_t alloc(dsh_ctx_base* c, size_t bytes) { return c->scratch.take(bytes, &p); }
  template <class T> T* as() { return static_cast<T*>(p); }
};

// int_0^1 b_p^(k) b_q^(k) dt for the four cubic B-spline pieces on a knot interval, k = 0, 1, 2: the three coefficient
// tables of the reference's bending code are products of these numbers.
void spline_integrals(double I[3][4][4]) {
  const double piece[4][4] = {{1.0 / 6, -3.0 / 6, 3.0 / 6, -1.0 / 6}, {4.0 / 6, 0.0, -6.0 / 6, 3.0 / 6}, {1.0 / 6, 3.0 / 6, 3.0 / 6, -3.0 / 6}, {0.0, 0.0, 0.0, 1.0 / 6}};
  for (int k = 0; k < 3; k++)
    for (int p = 0; p < 4; p++)
      for (int q = 0; q < 4; q++) {
        double a[4], b[4];
        for (int t = 0; t < 4; t++) { a[t] = piece[p][t]; b[t] = piece[q][t]; }
        for (int s = 0; s < k; s++) {
          const double da[4] = {a[1], 2 * a[2], 3 * a[3], 0.0}, db[4] = {b[1], 2 * b[2], 3 * b[3], 0.0};
          std::memcpy(a, da, sizeof a);
          std::memcpy(b, db, sizeof b);
        }
        double s = 0.0;
        for (int i = 0; i < 4; i++)
          for (int j = 0; j < 4; j++) s += a[i] * b[j] / (double)(i + j + 1);
        I[k][p][q] = s;
      }
}

void bending_dense(const dsh_bbs* b, double lambda, double* Bm) {
  const int nx = b->nptsv, ny = b->nptsu, N = nx * ny;
  const double sy = (b->umax - b->umin) / (b->nptsu - 3), sx = (b->vmax - b->vmin) / (b->nptsv - 3);
  double I[3][4][4];
  spline_integrals(I);
  double coeff[16][16];
  for (int d = 0; d < 16; d++)
    for (int c = 0; c <= d; c++) {
      const int e1 = c / 4, f1 = c % 4, e2 = d / 4, f2 = d % 4;
      const double bxx = I[2][f1][f2] * I[0][e1][e2], byy = I[0][f1][f2] * I[2][e1][e2], bxy = 2.0 * I[1][f1][f2] * I[1][e1][e2];
      coeff[d][c] = sy * bxx / std::pow(sx, 3) + bxy / (sx * sy) + sx * byy / std::pow(sy, 3);
    }
  std::fill(Bm, Bm + (size_t)N * N, 0.0);
  for (int cb = 0; cb < ny - 3; cb++)        // knot cells, u index outer like the reference
    for (int ca = 0; ca < nx - 3; ca++)
      for (int c = 0; c < 16; c++)
        for (int d = c; d < 16; d++) {
          const int i = (cb + c / 4) * nx + ca + c % 4, j = (cb + d / 4) * nx + ca + d % 4;
          Bm[(size_t)i * N + j] += lambda * coeff[d][c];
          if (i != j) Bm[(size_t)j * N + i] = Bm[(size_t)i * N + j];
        }
}

bool bbs_ok(const dsh_bbs* b) { return b && b->nptsu >= 4 && b->nptsv >= 4 && b->umax > b->umin && b->vmax > b->vmin; }
}  // namespace

namespace dsh {
// dense N x N bending matrix of the grid (used by the batched Schwarp fit for its Warp::initialize stage, dsh_schwarp.cpp)
void bbs_bending_dense(const dsh_bbs* b, double lambda, double* Bm) { bending_dense(b, lambda, Bm); }
}  // namespace dsh

extern "C" {

int dsh_bbs_bending(const dsh_bbs* bbs, double lambda, double* bending) {
  if (!bbs_ok(bbs) || !bending) return DSH_ERR_ARG;
  bending_dense(bbs, lambda, bending);
  return DSH_OK;
}

}  // extern "C"

// The body of dsh_sfn_estimate / dsh_sfn_estimate_db: the normals come from the host (normals) or are picked on the device out of the
// last normal solve of a database (db, sel).
static int sfn_estimate(dsh_ctx* ctx, const dsh_bbs* bbs, int n, const double* u, const double* v, const float* normals, const dsh_diffdb* ndb, const int32_t* sel,
                        double bending_weight, double mean_depth, int n_all, const double* u_all, const double* v_all, double* ctrl_raw, double* ctrl, float* pts,
                        int32_t* ok) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_sfn_estimate: host-only context, no GPU (there is no CPU fallback)");
  if (!bbs_ok(bbs) || n < 0 || n_all < 0 || (n > 0 && (!u || !v || (!normals && !(ndb && sel)))) || (n_all > 0 && (!u_all || !v_all || !pts)) || !ctrl || !ok)
    return dsh_fail(c, DSH_ERR_ARG, "dsh_sfn_estimate: bad argument");
  if (ndb) {
    if (ndb->ctx != c) return dsh_fail(c, DSH_ERR_ARG, "dsh_sfn_estimate_db: the database belongs to another context");
    for (int i = 0; i < n; i++)
      if (sel[i] >= ndb->last_P || -1 - (long long)sel[i] >= ndb->last_R)
        return dsh_fail(c, DSH_ERR_ARG, "dsh_sfn_estimate_db: sel[" + std::to_string(i) + "] is outside the last normal solve of the database");
  }
  const int N = bbs->nptsu * bbs->nptsv;
  if (N > 512) return dsh_fail(c, DSH_ERR_ARG, "dsh_sfn_estimate: more than 512 control points (one-workgroup solve)");
  *ok = 0;
  if (hipSetDevice(c->device) != hipSuccess) return dsh_fail(c, DSH_ERR_HIP, "dsh_sfn_estimate: hipSetDevice failed");
  c->scratch.reset();
  hipStream_t st = c->stream;
  const int m = 2 * n + N + 1, np = nrsfm_swp_solve_np(N);
  DevBuf dA, db, dr, dx, ddx, dG, dg, dM, dW, dones, dscal, du, dv, dn, dua, dva, dctrl, dpts;
  HIPCHK(c, dA.alloc(c, 8 * (size_t)m * N)); HIPCHK(c, db.alloc(c, 8 * (size_t)m)); HIPCHK(c, dr.alloc(c, 8 * (size_t)m));
  HIPCHK(c, dx.alloc(c, 8 * (size_t)N)); HIPCHK(c, ddx.alloc(c, 8 * (size_t)N)); HIPCHK(c, dG.alloc(c, 8 * (size_t)N * N)); HIPCHK(c, dg.alloc(c, 8 * (size_t)N));
  HIPCHK(c, dM.alloc(c, 8 * (size_t)np * np)); HIPCHK(c, dW.alloc(c, 8 * (size_t)np * 16)); HIPCHK(c, dones.alloc(c, 8 * (size_t)N)); HIPCHK(c, dscal.alloc(c, 256));
  HIPCHK(c, du.alloc(c, 8 * (size_t)n)); HIPCHK(c, dv.alloc(c, 8 * (size_t)n)); HIPCHK(c, dn.alloc(c, 12 * (size_t)n));
  HIPCHK(c, dua.alloc(c, 8 * (size_t)n_all)); HIPCHK(c, dva.alloc(c, 8 * (size_t)n_all)); HIPCHK(c, dctrl.alloc(c, 8 * (size_t)N)); HIPCHK(c, dpts.alloc(c, 12 * (size_t)n_all));
  // constant rows: bending block, the row of ones, right-hand side (zero except N * mean_depth in the last row)
  std::vector<double> tail((size_t)(N + 1) * N), bvec((size_t)m, 0.0), ones((size_t)N, 1.0);
  bending_dense(bbs, bending_weight, tail.data());
  std::fill(tail.begin() + (size_t)N * N, tail.end(), 1.0);
  bvec[m - 1] = (double)N * mean_depth;
  HIPCHK(c, hipMemsetAsync(dA.p, 0, 8 * (size_t)2 * n * N, st));
  HIPCHK(c, hipMemcpyAsync(dA.as<double>() + (size_t)2 * n * N, tail.data(), 8 * tail.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(db.p, bvec.data(), 8 * (size_t)m, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dones.p, ones.data(), 8 * (size_t)N, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemsetAsync(dx.p, 0, 8 * (size_t)N, st));
  HIPCHK(c, hipMemsetAsync(dscal.p, 0, 256, st));
  if (n > 0) {
    HIPCHK(c, hipMemcpyAsync(du.p, u, 8 * (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dv.p, v, 8 * (size_t)n, hipMemcpyHostToDevice, st));
    if (ndb) {
      DevBuf dsel;
      HIPCHK(c, dsel.alloc(c, 4 * (size_t)n));
      HIPCHK(c, hipMemcpyAsync(dsel.p, sel, 4 * (size_t)n, hipMemcpyHostToDevice, st));
      HIPCHK(c, ddb_pick_normals(n, dsel.as<int32_t>(), ndb->last_normals, ndb->last_normals + 3 * (size_t)ndb->last_P, dn.as<float>(), st));
    } else {
      HIPCHK(c, hipMemcpyAsync(dn.p, normals, 12 * (size_t)n, hipMemcpyHostToDevice, st));
    }
  }
  HIPCHK(c, nrsfm_sfn_rows(bbs->umin, bbs->umax, bbs->nptsu, bbs->vmin, bbs->vmax, bbs->nptsv, n, du.as<double>(), dv.as<double>(), dn.as<float>(), dA.as<double>(), st));
  // x0: G x = A^T b  (the solve kernel returns M dx = -g, so it is fed g = A^T (-(b - A x)))
  double* scal = dscal.as<double>();
  for (int it = 0; it < 3; it++) {
    HIPCHK(c, nrsfm_sfn_residual(m, N, dA.as<double>(), dx.as<double>(), db.as<double>(), -1.0, dr.as<double>(), st));
    HIPCHK(c, nrsfm_swp_normal(0, m, N, dA.as<double>(), dr.as<double>(), dones.as<double>(), scal, dG.as<double>(), dg.as<double>(), st));
    if (it == 0) HIPCHK(c, nrsfm_swp_solve(N, dG.as<double>(), dg.as<double>(), 1e300, dM.as<double>(), dW.as<double>(), ddx.as<double>(), scal + 2, 0, N, st));   // dense: the mean-depth row couples every pair of control points
    else HIPCHK(c, nrsfm_swp_resolve(N, dg.as<double>(), dM.as<double>(), dW.as<double>(), ddx.as<double>(), 0, N, st));
    HIPCHK(c, nrsfm_sfn_axpy(N, ddx.as<double>(), dx.as<double>(), st));
  }
  std::vector<double> x((size_t)N);
  double s8[8];
  HIPCHK(c, hipMemcpyAsync(x.data(), dx.p, 8 * (size_t)N, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(s8, dscal.p, 64, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (ctrl_raw) std::memcpy(ctrl_raw, x.data(), 8 * (size_t)N);
  const bool factor_ok = s8[2] != 0.0 || s8[3] != 0.0;   // out[0] also folds in "model > 0"; a positive-definite G always has it
  bool finite = true;
  for (double t : x) finite = finite && std::isfinite(t);
  if (!factor_ok || !finite || n_all == 0) return DSH_OK;
  // the reference's scale: 1 / median of the float32 control points (ShapeFromNormals.cc:123-135)
  std::vector<float> dvec((size_t)N);
  for (int i = 0; i < N; i++) dvec[i] = (float)x[i];
  std::sort(dvec.begin(), dvec.end());
  const float corr = 1 / dvec[dvec.size() / 2];
  for (int i = 0; i < N; i++) ctrl[i] = corr * x[i];
  HIPCHK(c, hipMemcpyAsync(dctrl.p, ctrl, 8 * (size_t)N, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dua.p, u_all, 8 * (size_t)n_all, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dva.p, v_all, 8 * (size_t)n_all, hipMemcpyHostToDevice, st));
  HIPCHK(c, nrsfm_sfn_points(bbs->umin, bbs->umax, bbs->nptsu, bbs->vmin, bbs->vmax, bbs->nptsv, dctrl.as<double>(), n_all, dua.as<double>(), dva.as<double>(),
                             dpts.as<float>(), st));
  HIPCHK(c, hipMemcpyAsync(pts, dpts.p, 12 * (size_t)n_all, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  *ok = 1;
  return DSH_OK;
}

extern "C" {

int dsh_sfn_estimate(dsh_ctx* ctx, const dsh_bbs* bbs, int n, const double* u, const double* v, const float* normals, double bending_weight,
                     double mean_depth, int n_all, const double* u_all, const double* v_all, double* ctrl_raw, double* ctrl, float* pts, int32_t* ok) {
  return sfn_estimate(ctx, bbs, n, u, v, normals, nullptr, nullptr, bending_weight, mean_depth, n_all, u_all, v_all, ctrl_raw, ctrl, pts, ok);
}

int dsh_sfn_estimate_db(dsh_ctx* ctx, const dsh_bbs* bbs, const dsh_diffdb* db, int n, const int32_t* sel, const double* u, const double* v, double bending_weight,
                        double mean_depth, int n_all, const double* u_all, const double* v_all, double* ctrl_raw, double* ctrl, float* pts, int32_t* ok) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (!db || (n > 0 && !sel)) return dsh_fail(c, DSH_ERR_ARG, "dsh_sfn_estimate_db: bad argument");
  return sfn_estimate(ctx, bbs, n, u, v, nullptr, db, sel, bending_weight, mean_depth, n_all, u_all, v_all, ctrl_raw, ctrl, pts, ok);
}

int dsh_warp_initialize(dsh_ctx* ctx, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, double lambda, double* x, int32_t* ok) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_warp_initialize: host-only context, no GPU (there is no CPU fallback)");
  if (!bbs_ok(bbs) || P <= 0 || !kp1 || !kp2 || !x || !ok) return dsh_fail(c, DSH_ERR_ARG, "dsh_warp_initialize: bad argument");
  const int N = bbs->nptsu * bbs->nptsv;
  if (N > 512) return dsh_fail(c, DSH_ERR_ARG, "dsh_warp_initialize: more than 512 control points (one-workgroup solve)");
  *ok = 0;
  if (hipSetDevice(c->device) != hipSuccess) return dsh_fail(c, DSH_ERR_HIP, "dsh_warp_initialize: hipSetDevice failed");
  c->scratch.reset();
  hipStream_t st = c->stream;
  const int np = nrsfm_swp_solve_np(N);
  DevBuf dC, dr0, dr1, dG, dB, dg0, dg1, dM, dW, dx, dones, dscal, dk1, dk2;
  HIPCHK(c, dC.alloc(c, 8 * (size_t)P * N)); HIPCHK(c, dr0.alloc(c, 8 * (size_t)P)); HIPCHK(c, dr1.alloc(c, 8 * (size_t)P));
  HIPCHK(c, dG.alloc(c, 8 * (size_t)N * N)); HIPCHK(c, dB.alloc(c, 8 * (size_t)N * N)); HIPCHK(c, dg0.alloc(c, 8 * (size_t)N)); HIPCHK(c, dg1.alloc(c, 8 * (size_t)N));
  HIPCHK(c, dM.alloc(c, 8 * (size_t)np * np)); HIPCHK(c, dW.alloc(c, 8 * (size_t)np * 16)); HIPCHK(c, dx.alloc(c, 8 * (size_t)2 * N));
  HIPCHK(c, dones.alloc(c, 8 * (size_t)N)); HIPCHK(c, dscal.alloc(c, 256)); HIPCHK(c, dk1.alloc(c, 8 * (size_t)P)); HIPCHK(c, dk2.alloc(c, 8 * (size_t)P));
  std::vector<double> Bm((size_t)N * N), ones((size_t)N, 1.0);
  bending_dense(bbs, lambda, Bm.data());
  HIPCHK(c, hipMemcpyAsync(dB.p, Bm.data(), 8 * Bm.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dones.p, ones.data(), 8 * (size_t)N, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dk1.p, kp1, 8 * (size_t)P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dk2.p, kp2, 8 * (size_t)P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemsetAsync(dC.p, 0, 8 * (size_t)P * N, st));
  HIPCHK(c, hipMemsetAsync(dscal.p, 0, 256, st));
  HIPCHK(c, nrsfm_warp_coloc(bbs->umin, bbs->umax, bbs->nptsu, bbs->vmin, bbs->vmax, bbs->nptsv, P, dk1.as<float>(), dk2.as<float>(), dC.as<double>(), dr0.as<double>(),
                             dr1.as<double>(), st));
  double* scal = dscal.as<double>();
  // g1 = C^T (-q2y) first (G is rebuilt by the second call), then G = C^T C, g0 = C^T (-q2x); G += Bending
  HIPCHK(c, nrsfm_swp_normal(0, P, N, dC.as<double>(), dr1.as<double>(), dones.as<double>(), scal, dG.as<double>(), dg1.as<double>(), st));
  HIPCHK(c, nrsfm_swp_normal(0, P, N, dC.as<double>(), dr0.as<double>(), dones.as<double>(), scal, dG.as<double>(), dg0.as<double>(), st));
  HIPCHK(c, nrsfm_mat_add((size_t)N * N, dB.as<double>(), dG.as<double>(), st));
  HIPCHK(c, nrsfm_swp_solve(N, dG.as<double>(), dg0.as<double>(), 1e300, dM.as<double>(), dW.as<double>(), dx.as<double>(), scal + 2, 0, 3 * bbs->nptsv + 3, st));   // colocation and bending couple a 4 x 4 patch: banded
  HIPCHK(c, nrsfm_swp_resolve(N, dg1.as<double>(), dM.as<double>(), dW.as<double>(), dx.as<double>() + N, 0, 3 * bbs->nptsv + 3, st));
  double s8[8];
  HIPCHK(c, hipMemcpyAsync(x, dx.p, 8 * (size_t)2 * N, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(s8, dscal.p, 64, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  bool finite = true;
  for (int i = 0; i < 2 * N; i++) finite = finite && std::isfinite(x[i]);
  *ok = ((s8[2] != 0.0 || s8[3] != 0.0) && finite) ? 1 : 0;
  return DSH_OK;
}

int dsh_search_by_schwarp(dsh_ctx* ctx, const dsh_bbs* bbs, const double* x, int Q, const float* kp1, const uint8_t* desc1, const float* cam2,
                          const float* bounds2, int grid_cols, int grid_rows, int N2, const float* kp2, const uint8_t* desc2, const uint8_t* has_mp2,
                          float radius, int th_low, int32_t* match, int32_t* nmatches) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_search_by_schwarp: host-only context, no GPU (there is no CPU fallback)");
  if (!bbs_ok(bbs) || !x || Q < 0 || N2 < 0 || (Q > 0 && (!kp1 || !desc1 || !match)) || (N2 > 0 && (!kp2 || !desc2 || !has_mp2)) || !cam2 || !bounds2 ||
      grid_cols <= 0 || grid_rows <= 0 || grid_cols * grid_rows > 8192 || !(bounds2[1] > bounds2[0]) || !(bounds2[3] > bounds2[2]) || th_low > 256)
    return dsh_fail(c, DSH_ERR_ARG, "dsh_search_by_schwarp: bad argument");
  if (nmatches) *nmatches = 0;
  if (Q == 0) return DSH_OK;
  if (hipSetDevice(c->device) != hipSuccess) return dsh_fail(c, DSH_ERR_HIP, "dsh_search_by_schwarp: hipSetDevice failed");
  c->scratch.reset();
  hipStream_t st = c->stream;
  const int N = bbs->nptsu * bbs->nptsv;
  DevBuf dx, dk1, dd1, dk2, dd2, dmp, dcell, dmatch;
  HIPCHK(c, dx.alloc(c, 8 * (size_t)2 * N)); HIPCHK(c, dk1.alloc(c, 8 * (size_t)Q)); HIPCHK(c, dd1.alloc(c, 32 * (size_t)Q));
  HIPCHK(c, dk2.alloc(c, 8 * (size_t)N2)); HIPCHK(c, dd2.alloc(c, 32 * (size_t)N2)); HIPCHK(c, dmp.alloc(c, (size_t)N2));
  HIPCHK(c, dcell.alloc(c, 4 * (size_t)N2)); HIPCHK(c, dmatch.alloc(c, 4 * (size_t)Q));
  HIPCHK(c, hipMemcpyAsync(dx.p, x, 8 * (size_t)2 * N, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dk1.p, kp1, 8 * (size_t)Q, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dd1.p, desc1, 32 * (size_t)Q, hipMemcpyHostToDevice, st));
  if (N2 > 0) {
    HIPCHK(c, hipMemcpyAsync(dk2.p, kp2, 8 * (size_t)N2, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dd2.p, desc2, 32 * (size_t)N2, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dmp.p, has_mp2, (size_t)N2, hipMemcpyHostToDevice, st));
  }
  HIPCHK(c, nrsfm_match_search(bbs->umin, bbs->umax, bbs->nptsu, bbs->vmin, bbs->vmax, bbs->nptsv, dx.as<double>(), Q, dk1.as<float>(), dd1.as<uint32_t>(), cam2,
                               bounds2, grid_cols, grid_rows, N2, dk2.as<float>(), dd2.as<uint32_t>(), dmp.as<uint8_t>(), radius, th_low, dcell.as<int32_t>(),
                               dmatch.as<int32_t>(), st));
  HIPCHK(c, hipMemcpyAsync(match, dmatch.p, 4 * (size_t)Q, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (nmatches) {
    int n = 0;
    for (int q = 0; q < Q; q++) n += match[q] >= 0;
    *nmatches = n;
  }
  return DSH_OK;
}

}  // extern "C"
