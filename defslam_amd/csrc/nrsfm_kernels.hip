// NRSfM mapping-side kernels for gfx950, FP64.
//
//   bbs_eval_kernel / bbs_coloc_kernel .. uniform bicubic B-spline evaluation with derivative orders and the
//                                         16-tap colocation rows (Thirdparty/BBS/bbs.cc:70-195, 214-355)
//   normals_coeff_kernel ................ two bicubic polynomials per keyframe pair (PolySolver.cc:50-149, with the
//                                         float32 intermediates of NormalEstimator.cc:78-104)
//   normals_solve_kernel ................ one thread per map point: Levenberg-Marquardt on the 2-unknown system,
//                                         covariance gate, normal of the reference keyframe (NormalEstimator.cc:112-170)
//   normals_propagate_kernel ............ normals in the other keyframes (NormalEstimator.cc:173-224)
//
// Compiled with -ffp-contract=off: the float32/float64 expression order of the reference is kept, no FMA fusion.
// All kernels are embarrassingly parallel and bandwidth-trivial; records are stored SoA so neighbouring
// threads read neighbouring addresses.
#include <hip/hip_runtime.h>
#include <cstring>
#include "tile_chol.h"
#include <math.h>
#include <stdint.h>

namespace {

// ------------------------------------------------------------------------------------------------
// B-spline
// ------------------------------------------------------------------------------------------------
struct BbsPar { double umin, umax, vmin, vmax; int nptsu, nptsv, valdim, pad; };

__device__ __forceinline__ void norm_inter(double xmin, double xmax, int npts, double x, double& nx, int& inter) {
  const int ninter = npts - 3;
  const double width = (xmax - xmin) / ninter;
  if (x == xmax) { nx = 1.0; inter = ninter - 1; }
  else if (x < xmin) { nx = (x - xmin) / width; inter = -1; }
  else if (x > xmax) { nx = (x - xmin) / width - ninter; inter = ninter; }
  else { const double s = (x - xmin) / width; inter = (int)floor(s); nx = s - inter; }
}

__device__ __forceinline__ void cubic_basis(int order, double t, double* b) {
  const double t2 = t * t, t3 = t2 * t;
  if (order == 0) {
    b[0] = (-t3 + 3.0 * t2 - 3.0 * t + 1.0) / 6.0;
    b[1] = (3.0 * t3 - 6.0 * t2 + 4.0) / 6.0;
    b[2] = (-3.0 * t3 + 3.0 * t2 + 3.0 * t + 1.0) / 6.0;
    b[3] = t3 / 6.0;
  } else if (order == 1) {
    b[0] = (-t2 + 2 * t - 1) / 2.0;
    b[1] = (3.0 * t2 - 4.0 * t) / 2.0;
    b[2] = (-3 * t2 + 2 * t + 1) / 2.0;
    b[3] = t2 / 2.0;
  } else {
    b[0] = -t + 1.0;
    b[1] = 3.0 * t - 2.0;
    b[2] = -3.0 * t + 1.0;
    b[3] = t;
  }
}

__device__ __forceinline__ double deriv_fact(const BbsPar& p, int du, int dv) {
  const double su = (p.umax - p.umin) / (p.nptsu - 3);
  const double sv = (p.vmax - p.vmin) / (p.nptsv - 3);
  // orders are 0, 1 or 2: products reproduce pow() for these exponents (correctly rounded square)
  const double pu = du == 0 ? 1.0 : (du == 1 ? su : su * su);
  const double pv = dv == 0 ? 1.0 : (dv == 1 ? sv : sv * sv);
  return 1.0 / (pu * pv);
}

__global__ void bbs_eval_kernel(BbsPar p, const double* __restrict__ ctrl, const double* __restrict__ u, const double* __restrict__ v,
                                int n, int du, int dv, double* __restrict__ val, uint8_t* __restrict__ outside, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) double s_ctrl[];
  const int nctrl = p.valdim * p.nptsu * p.nptsv;
  if (use_lds) {
    for (int i = threadIdx.x; i < nctrl; i += blockDim.x) s_ctrl[i] = ctrl[i];
    __syncthreads();
  }
  const double* cp = use_lds ? s_ctrl : ctrl;
  const double fact = deriv_fact(p, du, dv);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    double nu, nv, bu[4], bv[4];
    int Iu, Iv;
    norm_inter(p.umin, p.umax, p.nptsu, u[k], nu, Iu);
    norm_inter(p.vmin, p.vmax, p.nptsv, v[k], nv, Iv);
    cubic_basis(du, nu, bu);
    cubic_basis(dv, nv, bv);
    const bool bad = Iu < 0 || Iu > p.nptsu - 4 || Iv < 0 || Iv > p.nptsv - 4;
    if (outside) outside[k] = bad ? 1 : 0;
    for (int d = 0; d < p.valdim; d++) {
      double acc = 0.0;
      if (!bad) {
        for (int iu = 0; iu < 4; iu++)
          for (int iv = 0; iv < 4; iv++) {
            const double bas = bu[iu] * bv[iv];
            acc += cp[p.valdim * ((iu + Iu) * p.nptsv + iv + Iv) + d] * bas;
          }
        acc *= fact;
      }
      val[(size_t)p.valdim * k + d] = acc;
    }
  }
}

__global__ void bbs_coloc_kernel(BbsPar p, const double* __restrict__ u, const double* __restrict__ v, int n, int du, int dv,
                                 int32_t* __restrict__ cols, double* __restrict__ w, int32_t* __restrict__ n_outside) {
  const double fact = deriv_fact(p, du, dv);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    double nu, nv, bu[4], bv[4];
    int Iu, Iv;
    norm_inter(p.umin, p.umax, p.nptsu, u[k], nu, Iu);
    norm_inter(p.vmin, p.vmax, p.nptsv, v[k], nv, Iv);
    const bool bad = Iu < 0 || Iu > p.nptsu - 4 || Iv < 0 || Iv > p.nptsv - 4;
    if (bad) {
      atomicAdd(n_outside, 1);
      for (int t = 0; t < 16; t++) { cols[16 * (size_t)k + t] = -1; w[16 * (size_t)k + t] = 0.0; }
      continue;
    }
    cubic_basis(du, nu, bu);
    cubic_basis(dv, nv, bv);
    for (int iu = 0; iu < 4; iu++)
      for (int iv = 0; iv < 4; iv++) {
        cols[16 * (size_t)k + 4 * iu + iv] = (iu + Iu) * p.nptsv + iv + Iv;
        w[16 * (size_t)k + 4 * iu + iv] = (du == 0 && dv == 0) ? bu[iu] * bv[iv] : fact * bu[iu] * bv[iv];
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Normals
// ------------------------------------------------------------------------------------------------
enum { F_I1u, F_I1v, F_I2u, F_I2v, F_J12a, F_J12b, F_J12c, F_J12d, F_J21a, F_J21b, F_J21c, F_J21d, F_Huux, F_Huuy, F_Huvx, F_Huvy, F_Hvvx, F_Hvvy, NF };

// recs are SoA: field f of record r at recs[f * R + r]; coefficients SoA: q[c * R + r], c = 0..19
__global__ void normals_coeff_kernel(int R, const float* __restrict__ recs, const uint8_t* __restrict__ is_ref, double* __restrict__ Q) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R || !is_ref[r]) return;
  auto F = [&](int f) { return recs[(size_t)f * R + r]; };
  const float af = F(F_J12a), bf = F(F_J12b), cf = F(F_J12c), df = F(F_J12d);
  const float t1f = -F(F_J12b) * F(F_Hvvx) / 2 + F(F_J12a) * F(F_Hvvy) / 2;
  const float t2f = -(F(F_J12d) * F(F_Hvvx)) / 2 + (F(F_J12c) * F(F_Hvvy)) / 2;
  const float I1u = F(F_I1u), I1v = F(F_I1v), I2u = F(F_I2u), I2v = F(F_I2v);
  const float e2f = 1 + I2u * I2u + I2v * I2v;
  const float e1f = 1 + I1u * I1u + I1v * I1v;
  const double a = af, b = bf, c = cf, d = df, t1 = t1f, t2 = t2f, e1 = e1f, e2 = e2f, x1 = I1u, y1 = I1v, x2 = I2u, y2 = I2v;
  const double D = a * d - c * b;
  double q[20];
  q[0] = D * (t1 * e1 * e2 - D * (e1 * (c * x2 + d * y2) - y1 * e2));
  q[1] = -D * (t2 * e1 * e2 - D * (e1 * (a * x2 + b * y2) - x1 * e2));
  q[2] = 0;
  q[3] = 0;
  q[4] = t2 * (e1 * e2 * t1 - D * (x2 * e1 * c + y2 * e1 * d - 2 * e2 * y1)) - t1 * D * (x2 * e1 * a + e1 * b * y2 + 2 * e2 * x1) +
         D * D * (e1 * (a * c + b * d) - 2 * (a * x2 * y1 - c * x1 * x2 + b * y1 * y2 - d * x1 * y2));
  q[5] = (e1 * (-e2 * (t2 * t2) + 2 * x2 * t2 * a * D + 2 * y2 * t2 * b * D - ((a * a) + (b * b)) * D * D) + e2 * D * D);
  q[6] = 0;
  q[7] = t1 * (e2 * D + 2 * a * x1 * x2 * (D) + 2 * x1 * y2 * b * D) - t2 * 2 * (e2 * x1 * t1 + D * (x2 * y1 * a - c * x1 * x2 + y1 * y2 * b - x1 * y2 * d)) +
         e2 * y1 * t2 * t2 + D * D * (-2 * x1 * (a * c + b * d) + y1 * (a * a + b * b) - c * x2 - d * y2);
  q[8] = t2 * (D * (e2 - 2 * a * x1 * x2 - 2 * b * x1 * y2)) + x1 * e2 * t2 * t2 + (D * D) * (-y2 * b - x2 * a + x1 * (a * a + b * b));
  q[9] = t2 * (e2 * t1 - D * (c * x2 + d * y2)) - t1 * (D * (a * x2 + b * y2)) + (a * c + b * d) * D * D;
  q[10] = 0;
  q[11] = 0;
  q[12] = -D * (e1 * e2 * t1 - D * (e1 * (c * x2 + d * y2) - e2 * y1));
  q[13] = D * (e1 * e2 * t2 - (D * (e1 * (a * x2 + b * y2) - e2 * x1)));
  q[14] = 0;
  q[15] = e1 * (-e2 * t1 * t1 + (D * (-(c * c + d * d) * D + 2 * t1 * c * x2 + 2 * d * y2 * t1))) + e2 * D * D;
  q[16] = t2 * (e1 * e2 * t1 - D * (e1 * c * x2 + e1 * d * y2 + 2 * e2 * y1)) - t1 * D * (e1 * (a * x2 + b * y2) - 2 * e2 * x1) +
          D * D * ((e1 * (a * c + b * d) + 2 * (a * x2 * y1 - c * x1 * x2 + b * y1 * y2 - d * x1 * y2)));
  q[17] = t1 * D * (e2 - 2 * c * x2 * y1 - 2 * d * y1 * y2) + y1 * (e2 * t1 * t1 + D * D * (c * c + d * d)) - D * D * (c * x2 + d * y2);
  q[18] = t2 * (e2 * D + 2 * y1 * D * (c * x2 + d * y2)) + t1 * (-2 * e2 * y1 * t2 + 2 * D * (a * x2 * y1 - c * x1 * x2 + b * y1 * y2 - d * x1 * y2)) +
          e2 * x1 * t1 * t1 - 2 * D * D * (a * c * y1 + 0.5 * a * x2 - 0.5 * c * c * x1 + b * d * y1 + 0.5 * b * y2 - 0.5 * d * d * x1);
  q[19] = t2 * (e2 * t1 - D * (c * x2 + d * y2)) - t1 * (D * (a * x2 + b * y2)) + D * D * (a * c + b * d);
#pragma unroll
  for (int i = 0; i < 20; i++) Q[(size_t)i * R + r] = q[i];
}

struct Lsq { double cost, g0, g1, a00, a01, a11; };

// 1/2 |r|^2 (+ scaled gradient and J^T J when WITH_J) over the reference records of one point
template <bool WITH_J>
__device__ Lsq lsq_eval(int r0, int r1, int R, const uint8_t* is_ref, const double* Q, double x, double y, double s0, double s1) {
  Lsq o = {0, 0, 0, 0, 0, 0};
  const double x2 = x * x, y2 = y * y, x3 = x2 * x, y3 = y2 * y;
  for (int r = r0; r < r1; r++) {
    if (!is_ref[r]) continue;
    double q[20];
#pragma unroll
    for (int i = 0; i < 20; i++) q[i] = Q[(size_t)i * R + r];
    const double e0 = q[0] * x3 + q[1] * x2 * y + q[2] * x * y2 + q[3] * y3 + q[4] * x2 + q[5] * x * y + q[6] * y2 + q[7] * x + q[8] * y + q[9];
    const double e1 = q[10] * x3 + q[11] * x2 * y + q[12] * x * y2 + q[13] * y3 + q[14] * x2 + q[15] * x * y + q[16] * y2 + q[17] * x + q[18] * y + q[19];
    o.cost += e0 * e0 + e1 * e1;
    if (WITH_J) {
      const double j00 = (3 * q[0] * x2 + 2 * q[1] * x * y + q[2] * y2 + 2 * q[4] * x + q[5] * y + q[7]) * s0;
      const double j01 = (q[1] * x2 + 2 * q[2] * x * y + 3 * q[3] * y2 + q[5] * x + 2 * q[6] * y + q[8]) * s1;
      const double j10 = (3 * q[10] * x2 + 2 * q[11] * x * y + q[12] * y2 + 2 * q[14] * x + q[15] * y + q[17]) * s0;
      const double j11 = (q[11] * x2 + 2 * q[12] * x * y + 3 * q[13] * y2 + q[15] * x + 2 * q[16] * y + q[18]) * s1;
      o.g0 += j00 * e0 + j10 * e1;
      o.g1 += j01 * e0 + j11 * e1;
      o.a00 += j00 * j00 + j10 * j10;
      o.a01 += j00 * j01 + j10 * j11;
      o.a11 += j01 * j01 + j11 * j11;
    }
  }
  o.cost *= 0.5;
  return o;
}

// Trust-region Levenberg-Marquardt as documented for Ceres' TRUST_REGION / LEVENBERG_MARQUARDT / DENSE_NORMAL_CHOLESKY
// with the options of NormalEstimator.cc:139-148 (see oracle/nrsfm_oracle.c for the statement of the algorithm).
__global__ void normals_solve_kernel(int P, int R, const int32_t* __restrict__ rec_ptr, const uint8_t* __restrict__ is_ref,
                                     const double* __restrict__ Q, const float* __restrict__ x0, const uint8_t* __restrict__ has_x0,
                                     const float* __restrict__ ref_uv, double* __restrict__ k1k2, double* __restrict__ cov,
                                     int32_t* __restrict__ status, float* __restrict__ normal_ref, int32_t* __restrict__ iters) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int r0 = rec_ptr[p], r1 = rec_ptr[p + 1];
  int K = 0;
  for (int r = r0; r < r1; r++) K += is_ref[r] ? 1 : 0;
  double x = 0.0, y = -0.0;
  status[p] = 1;
  iters[p] = 0;
  k1k2[2 * p] = x; k1k2[2 * p + 1] = y;
  if (K == 0) return;
  if (has_x0[p]) { x = x0[2 * p]; y = x0[2 * p + 1]; }
  const double ftol = 1e-10, gtol = 1e-8, ptol = 1e-8, min_rel_dec = 1e-3;
  double radius = 1e4, nu = 2.0;
  Lsq L = lsq_eval<true>(r0, r1, R, is_ref, Q, x, y, 1.0, 1.0);
  const double s0 = 1.0 / (1.0 + sqrt(L.a00)), s1 = 1.0 / (1.0 + sqrt(L.a11));
  L = lsq_eval<true>(r0, r1, R, is_ref, Q, x, y, s0, s1);
  int it = 0, invalid = 0;
  if (!(fmax(fabs(L.g0), fabs(L.g1)) <= gtol)) {
    while (it < 200) {
      it++;
      const double d0 = fmin(fmax(L.a00, 1e-6), 1e32) / radius, d1 = fmin(fmax(L.a11, 1e-6), 1e32) / radius;
      const double m00 = L.a00 + d0, m01 = L.a01, m11 = L.a11 + d1;
      const double l00 = sqrt(m00), l10 = m01 / l00, l11sq = m11 - l10 * l10;
      bool ok = (m00 > 0) && (l11sq > 0);
      double dx0 = 0, dx1 = 0, model = 0;
      if (ok) {
        const double l11 = sqrt(l11sq);
        const double y0 = -L.g0 / l00, y1 = (-L.g1 - l10 * y0) / l11;
        dx1 = y1 / l11;
        dx0 = (y0 - l10 * dx1) / l00;
        ok = isfinite(dx0) && isfinite(dx1);
        model = -(dx0 * L.g0 + dx1 * L.g1 + 0.5 * (dx0 * (L.a00 * dx0 + L.a01 * dx1) + dx1 * (L.a01 * dx0 + L.a11 * dx1)));
        if (!(model > 0)) ok = false;
      }
      if (!ok) {
        if (++invalid >= 5) break;
        radius *= 0.5;
        continue;
      }
      invalid = 0;
      const double st0 = dx0 * s0, st1 = dx1 * s1;
      const double xn = x + st0, yn = y + st1;
      const double snorm = sqrt(st0 * st0 + st1 * st1), xnorm = sqrt(x * x + y * y);
      if (snorm <= ptol * (xnorm + ptol)) break;
      const Lsq N = lsq_eval<false>(r0, r1, R, is_ref, Q, xn, yn, s0, s1);
      const double rel = (L.cost - N.cost) / model;
      if (rel > min_rel_dec) {
        const double cost_change = L.cost - N.cost, old_cost = L.cost;
        x = xn; y = yn;
        radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3.0));
        radius = fmin(1e16, radius);
        nu = 2.0;
        L = lsq_eval<true>(r0, r1, R, is_ref, Q, x, y, s0, s1);
        if (fmax(fabs(L.g0), fabs(L.g1)) <= gtol) break;
        if (fabs(cost_change) <= ftol * old_cost) break;
      } else {
        radius = radius / nu;
        nu *= 2.0;
        if (radius < 1e-32) break;
      }
    }
  }
  iters[p] = it;
  k1k2[2 * p] = x; k1k2[2 * p + 1] = y;
  const Lsq C = lsq_eval<true>(r0, r1, R, is_ref, Q, x, y, 1.0, 1.0);
  const double tr = C.a00 + C.a11, det = C.a00 * C.a11 - C.a01 * C.a01;
  const double disc = sqrt(fmax(0.0, 0.25 * tr * tr - det));
  const double lmax = 0.5 * tr + disc, lmin = det / lmax;
  if (!(lmax > 0) || !(lmin / lmax >= 1e-14)) { status[p] = 2; return; }
  cov[4 * p] = C.a11 / det; cov[4 * p + 1] = -C.a01 / det; cov[4 * p + 2] = -C.a01 / det; cov[4 * p + 3] = C.a00 / det;
  const float I1u = ref_uv[2 * p], I1v = ref_uv[2 * p + 1];
  normal_ref[3 * p] = (float)x;
  normal_ref[3 * p + 1] = (float)y;
  normal_ref[3 * p + 2] = (float)(1 - x * I1u - y * I1v);
  status[p] = 0;
}

__global__ void normals_propagate_kernel(int R, const float* __restrict__ recs, const int32_t* __restrict__ rec_point,
                                         const uint8_t* __restrict__ is_ref, const float* __restrict__ first_n, const uint8_t* __restrict__ has_first_n,
                                         const double* __restrict__ k1k2, const int32_t* __restrict__ status,
                                         float* __restrict__ normal_rec, uint8_t* __restrict__ written) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  written[r] = 0;
  const int p = rec_point[r];
  if (status[p] == 2) return;    // covariance failed: the reference skips the whole point (NormalEstimator.cc:157-158)
  double n0, n1;
  if (is_ref[r]) { n0 = k1k2[2 * p]; n1 = k1k2[2 * p + 1]; }
  else if (has_first_n[r]) { n0 = first_n[2 * r]; n1 = first_n[2 * r + 1]; }
  else return;
  auto F = [&](int f) { return recs[(size_t)f * R + r]; };
  const float j21_11 = F(F_J21a), j21_12 = F(F_J21c), j21_21 = F(F_J21b), j21_22 = F(F_J21d);
  const float a = F(F_J12a), b = F(F_J12b), c = F(F_J12c), d = F(F_J12d);
  const float detJ12 = a * d - c * b;
  const float t1 = -b * F(F_Hvvx) / 2 + a * F(F_Hvvy) / 2;
  const float t2 = (d * F(F_Huux)) / 2 - (c * F(F_Huuy)) / 2;   // H12uu here vs H12vv in the polynomials: reference quirk, kept
  const double k1 = j21_11 * n0 + j21_12 * n1 + (d * t2 - b * t1) / (detJ12 * detJ12);
  const double k2 = j21_21 * n0 + j21_22 * n1 + (a * t1 - c * t2) / (detJ12 * detJ12);
  const float I2u = F(F_I2u), I2v = F(F_I2v);
  normal_rec[3 * r] = (float)k1;
  normal_rec[3 * r + 1] = (float)k2;
  normal_rec[3 * r + 2] = (float)(1 - k1 * I2u - k2 * I2v);
  written[r] = 1;
}


// ------------------------------------------------------------------------------------------------
// Schwarzian-regularised warp fit: Schwarp.cc:38-97,235-543, SchwarpDatabase.cc:145-349
// Parameter layout x[0..N) first coordinate, x[N..2N) second; dense row-major Jacobian (2P+4N) x 2N.
// ------------------------------------------------------------------------------------------------
typedef double v2d_t __attribute__((ext_vector_type(2)));
struct SwpPar { double umin, umax, vmin, vmax, fxs, fys, lambda; int nu, nv, N, P; };

__device__ __forceinline__ void swp_eval16(const SwpPar& p, const double* x, double u, double v, int du, int dv, double& ox, double& oy) {
  BbsPar b = {p.umin, p.umax, p.vmin, p.vmax, p.nu, p.nv, 2, 0};
  double nu, nv, bu[4], bv[4];
  int Iu, Iv;
  norm_inter(p.umin, p.umax, p.nu, u, nu, Iu);
  norm_inter(p.vmin, p.vmax, p.nv, v, nv, Iv);
  cubic_basis(du, nu, bu);
  cubic_basis(dv, nv, bv);
  double ax = 0.0, ay = 0.0;
  if (!(Iu < 0 || Iu > p.nu - 4 || Iv < 0 || Iv > p.nv - 4)) {
    for (int iu = 0; iu < 4; iu++)
      for (int iv = 0; iv < 4; iv++) {
        const double bas = bu[iu] * bv[iv];
        const int l = (iu + Iu) * p.nv + iv + Iv;
        ax += x[l] * bas;
        ay += x[p.N + l] * bas;
      }
    const double fact = deriv_fact(b, du, dv);
    ax *= fact; ay *= fact;
  }
  ox = ax; oy = ay;
}

// taps of the site: columns (16) and weights for derivative order (du,dv)
__device__ __forceinline__ bool swp_taps(const SwpPar& p, double u, double v, int du, int dv, int* cols, double* w) {
  BbsPar b = {p.umin, p.umax, p.vmin, p.vmax, p.nu, p.nv, 2, 0};
  double nu, nv, bu[4], bv[4];
  int Iu, Iv;
  norm_inter(p.umin, p.umax, p.nu, u, nu, Iu);
  norm_inter(p.vmin, p.vmax, p.nv, v, nv, Iv);
  if (Iu < 0 || Iu > p.nu - 4 || Iv < 0 || Iv > p.nv - 4) return false;
  cubic_basis(du, nu, bu);
  cubic_basis(dv, nv, bv);
  const double fact = deriv_fact(b, du, dv);
  for (int iu = 0; iu < 4; iu++)
    for (int iv = 0; iv < 4; iv++) {
      cols[4 * iu + iv] = (iu + Iu) * p.nv + iv + Iv;
      w[4 * iu + iv] = (du == 0 && dv == 0) ? bu[iu] * bv[iv] : fact * bu[iu] * bv[iv];
    }
  return true;
}

template <bool WITH_J>
__device__ __forceinline__ void swp_eval_body(SwpPar p, const float* __restrict__ kp1, const float* __restrict__ kp2, const float* __restrict__ invsig,
                                const double* __restrict__ x, double* __restrict__ r, double* __restrict__ J) {
  const int n2 = 2 * p.N;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < p.P) {
    const int i = t;
    const double u = kp1[2 * i], v = kp1[2 * i + 1];
    double ex, ey;
    swp_eval16(p, x, u, v, 0, 0, ex, ey);
    r[i] = invsig[i] * ((double)kp2[2 * i] - ex) * p.fxs;
    r[i + p.P] = invsig[i] * ((double)kp2[2 * i + 1] - ey) * p.fys;
    if (WITH_J) {
      int cols[16]; double w[16];
      if (swp_taps(p, u, v, 0, 0, cols, w))
        for (int k = 0; k < 16; k++) {
          const double jv = -w[k] * p.fxs;
          J[(size_t)i * n2 + cols[k]] = jv;            // x row
          J[(size_t)(i + p.P) * n2 + cols[k]] = jv;    // y row: the reference overwrites it with the x row (Schwarp.cc:291-298)
        }
    }
  } else if (t < p.P + p.N) {
    const int k = t - p.P;
    const int iu = k / p.nv, iv = k % p.nv;
    const double X = (double)((p.umax - p.umin) * iu) / (p.nu - 1) + p.umin;
    const double Y = (double)((p.vmax - p.vmin) * iv) / (p.nv - 1) + p.vmin;
    double xu, yu, xv, yv, xuu, yuu, xvv, yvv, xuv, yuv;
    swp_eval16(p, x, X, Y, 1, 0, xu, yu);
    swp_eval16(p, x, X, Y, 0, 1, xv, yv);
    swp_eval16(p, x, X, Y, 2, 0, xuu, yuu);
    swp_eval16(p, x, X, Y, 0, 2, xvv, yvv);
    swp_eval16(p, x, X, Y, 1, 1, xuv, yuv);
    const double lam = p.lambda;
    double* rs = r + 2 * p.P;
    rs[k] = ((xuu * yu - yuu * xu)) * lam;
    rs[p.N + k] = ((yvv * xv - xvv * yv)) * lam;
    rs[2 * p.N + k] = ((xuu * yv - yuu * xv + 2 * (xuv * yu - yuv * xu))) * lam;
    rs[3 * p.N + k] = ((yvv * xu - xvv * yu + 2 * (yuv * xv - xuv * yv))) * lam;
    if (WITH_J) {
      int c[16]; double wu[16], wv[16], wuu[16], wvv[16], wuv[16];
      if (swp_taps(p, X, Y, 1, 0, c, wu)) {
        swp_taps(p, X, Y, 0, 1, c, wv); swp_taps(p, X, Y, 2, 0, c, wuu); swp_taps(p, X, Y, 0, 2, c, wvv); swp_taps(p, X, Y, 1, 1, c, wuv);
        double* Js = J + (size_t)2 * p.P * n2;
        const int N = p.N;
        for (int q = 0; q < 16; q++) {
          const int col = c[q];
          const double Cu = wu[q], Cv = wv[q], Cuu = wuu[q], Cvv = wvv[q], Cuv = wuv[q];
          Js[(size_t)k * n2 + col] = lam * (yu * Cuu - yuu * Cu);
          Js[(size_t)k * n2 + N + col] = lam * (xuu * Cu - xu * Cuu);
          Js[(size_t)(N + k) * n2 + col] = lam * (yvv * Cv - yv * Cvv);
          Js[(size_t)(N + k) * n2 + N + col] = lam * (xv * Cvv - xvv * Cv);
          Js[(size_t)(2 * N + k) * n2 + col] = lam * (yv * Cuu - yuu * Cv + 2 * yu * Cuv - 2 * yuv * Cu);
          Js[(size_t)(2 * N + k) * n2 + N + col] = lam * (xuu * Cv - xv * Cuu + 2 * xuv * Cu - 2 * xu * Cuv);
          Js[(size_t)(3 * N + k) * n2 + col] = lam * (yvv * Cu - yu * Cvv - 2 * yv * Cuv + 2 * yuv * Cv);
          Js[(size_t)(3 * N + k) * n2 + N + col] = lam * (xu * Cvv - xvv * Cu - 2 * xuv * Cv + 2 * xv * Cuv);
        }
      }
    }
  }
}

// The same residuals with the Jacobian in its structured form (SwpFit::Jw, Js): no dense (2P+4N) x 2N matrix is ever written.
__device__ __forceinline__ void swp_evalc_body(SwpPar p, const float* __restrict__ kp1, const float* __restrict__ kp2, const float* __restrict__ invsig,
                                               const double* __restrict__ x, double* __restrict__ r, double* __restrict__ Jw, double* __restrict__ Js) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < p.P) {
    const int i = t;
    const double u = kp1[2 * i], v = kp1[2 * i + 1];
    double ex, ey;
    swp_eval16(p, x, u, v, 0, 0, ex, ey);
    r[i] = invsig[i] * ((double)kp2[2 * i] - ex) * p.fxs;
    r[i + p.P] = invsig[i] * ((double)kp2[2 * i + 1] - ey) * p.fys;
    int cols[16]; double w[16];
    const bool in = swp_taps(p, u, v, 0, 0, cols, w);
#pragma unroll
    for (int k = 0; k < 16; k++) Jw[(size_t)i * 16 + k] = in ? -w[k] * p.fxs : 0.0;   // x row = y row (Schwarp.cc:291-298)
  } else if (t < p.P + p.N) {
    const int k = t - p.P;
    const int iu = k / p.nv, iv = k % p.nv;
    const double X = (double)((p.umax - p.umin) * iu) / (p.nu - 1) + p.umin;
    const double Y = (double)((p.vmax - p.vmin) * iv) / (p.nv - 1) + p.vmin;
    double xu, yu, xv, yv, xuu, yuu, xvv, yvv, xuv, yuv;
    swp_eval16(p, x, X, Y, 1, 0, xu, yu);
    swp_eval16(p, x, X, Y, 0, 1, xv, yv);
    swp_eval16(p, x, X, Y, 2, 0, xuu, yuu);
    swp_eval16(p, x, X, Y, 0, 2, xvv, yvv);
    swp_eval16(p, x, X, Y, 1, 1, xuv, yuv);
    const double lam = p.lambda;
    double* rs = r + 2 * p.P;
    rs[k] = ((xuu * yu - yuu * xu)) * lam;
    rs[p.N + k] = ((yvv * xv - xvv * yv)) * lam;
    rs[2 * p.N + k] = ((xuu * yv - yuu * xv + 2 * (xuv * yu - yuv * xu))) * lam;
    rs[3 * p.N + k] = ((yvv * xu - xvv * yu + 2 * (yuv * xv - xuv * yv))) * lam;
    int c[16]; double wu[16], wv[16], wuu[16], wvv[16], wuv[16];
    const bool in = swp_taps(p, X, Y, 1, 0, c, wu);
    if (in) { swp_taps(p, X, Y, 0, 1, c, wv); swp_taps(p, X, Y, 2, 0, c, wuu); swp_taps(p, X, Y, 0, 2, c, wvv); swp_taps(p, X, Y, 1, 1, c, wuv); }
    double* J0 = Js + (size_t)k * 128;
    for (int q = 0; q < 16; q++) {
      const double Cu = in ? wu[q] : 0.0, Cv = in ? wv[q] : 0.0, Cuu = in ? wuu[q] : 0.0, Cvv = in ? wvv[q] : 0.0, Cuv = in ? wuv[q] : 0.0;
      J0[q] = lam * (yu * Cuu - yuu * Cu);
      J0[16 + q] = lam * (xuu * Cu - xu * Cuu);
      J0[32 + q] = lam * (yvv * Cv - yv * Cvv);
      J0[48 + q] = lam * (xv * Cvv - xvv * Cv);
      J0[64 + q] = lam * (yv * Cuu - yuu * Cv + 2 * yu * Cuv - 2 * yuv * Cu);
      J0[80 + q] = lam * (xuu * Cv - xv * Cuu + 2 * xuv * Cu - 2 * xu * Cuv);
      J0[96 + q] = lam * (yvv * Cu - yu * Cvv - 2 * yv * Cuv + 2 * yuv * Cv);
      J0[112 + q] = lam * (xu * Cvv - xvv * Cu - 2 * xuv * Cv + 2 * xv * Cuv);
    }
  }
}

// Rows bucketed by knot cell, once per fit (the key points do not move).  First the cell of every match and grid site (one lane each;
// -1 outside the spline domain: in no bucket), then one 256-thread workgroup per fit: thread c owns cell c and walks the cell ids in
// index order (a fixed order of the sums below).
__device__ __forceinline__ void swp_cellid_body(SwpPar p, const float* __restrict__ kp1, int32_t* __restrict__ cid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.P + p.N) return;
  const int ncv = p.nv - 3;
  double u, v;
  if (t < p.P) { u = kp1[2 * t]; v = kp1[2 * t + 1]; }
  else {
    const int k = t - p.P, iu = k / p.nv, iv = k % p.nv;
    u = (double)((p.umax - p.umin) * iu) / (p.nu - 1) + p.umin;
    v = (double)((p.vmax - p.vmin) * iv) / (p.nv - 1) + p.vmin;
  }
  double nx; int Iu, Iv;
  norm_inter(p.umin, p.umax, p.nu, u, nx, Iu);
  norm_inter(p.vmin, p.vmax, p.nv, v, nx, Iv);
  cid[t] = (Iu < 0 || Iu > p.nu - 4 || Iv < 0 || Iv > p.nv - 4) ? -1 : Iu * ncv + Iv;
}
__device__ __forceinline__ void swp_buckets_body(SwpPar p, const int32_t* cid, int32_t* __restrict__ bw_ptr, int32_t* __restrict__ bw_idx,
                                                 int32_t* __restrict__ bs_ptr, int32_t* __restrict__ bs_idx, bool use_lds) {
  // A stable counting sort of the rows by knot cell (rows of a cell in index order: the order the gather adds them in).  Histogram by LDS
  // atomics, the row pointers by one lane per set, placement by one thread per cell.
  __shared__ int cnt[2][512];
  extern __shared__ __attribute__((aligned(16))) int cid_l[];
  const int ncell = (p.nu - 3) * (p.nv - 3);
  const int t = threadIdx.x;
  if (use_lds) {
    for (int i = t; i < p.P + p.N; i += blockDim.x) cid_l[i] = cid[i];
    cid = cid_l;
  }
  for (int c = t; c < 2 * 512; c += blockDim.x) (&cnt[0][0])[c] = 0;
  __syncthreads();
  for (int i = t; i < p.P + p.N; i += blockDim.x) {
    const int c = cid[i];
    if (c >= 0) atomicAdd(&cnt[i < p.P ? 0 : 1][c], 1);
  }
  __syncthreads();
  if (t < 2) {
    int32_t* ptr = t == 0 ? bw_ptr : bs_ptr;
    int acc = 0;
    for (int c = 0; c < ncell; c++) { const int k = cnt[t][c]; ptr[c] = acc; cnt[t][c] = acc; acc += k; }   // cnt becomes the cursor of the cell
    ptr[ncell] = acc;
  }
  __syncthreads();
  // placement: one thread per (set, cell) walks the ids in index order, four per LDS read (the ids are the same address for every lane: a
  // broadcast), and appends the rows of its cell -- stable by construction
  for (int w = t; w < 2 * ncell; w += blockDim.x) {
    const int set = w >= ncell, c = set ? w - ncell : w;
    const int off = set ? p.P : 0, count = set ? p.N : p.P;
    int32_t* idx = set ? bs_idx : bw_idx;
    int q = cnt[set][c];
    int i = 0;
    if (use_lds) {
      // (4-aligned stretch of the LDS copy; the warp rows start at 0, the Schwarzian rows at P)
      for (; i < count && ((off + i) & 3); i++) if (cid[off + i] == c) idx[q++] = i;
      for (; i + 4 <= count; i += 4) {
        const int4 v = *reinterpret_cast<const int4*>(cid + off + i);
        if (v.x == c) idx[q++] = i;
        if (v.y == c) idx[q++] = i + 1;
        if (v.z == c) idx[q++] = i + 2;
        if (v.w == c) idx[q++] = i + 3;
      }
    }
    for (; i < count; i++) if (cid[off + i] == c) idx[q++] = i;
  }
}

// A = (J S)^T (J S) and g = (J S)^T r from the structured Jacobian, as a GATHER: one wavefront per control point l1; lane nb < 49 owns the
// pair (l1, l2) with l2 in the 7 x 7 neighbourhood of l1 (two control points share a row of J only if a knot cell's 4 x 4 patch holds
// both) and sums, over the <= 16 cells that hold both and over the rows of each cell in bucket order, the four products
// (x|y of l1) x (x|y of l2).  The Huber weight of the reprojection block (one scalar for all 2P rows) and the duplicated y row enter as
// the factor 2 rho' on the warp rows' sum.  Rows l1 and N + l1 of A are written completely (zeros outside the neighbourhood): no
// memset, no atomics, both triangles from the same commutative products (bit-symmetric).  ~2.6 MFLOP where the dense product spent 460.
__device__ __forceinline__ void swp_normalc_body(SwpPar p, const double* __restrict__ Jw, const double* __restrict__ Js, const int32_t* __restrict__ bw_ptr,
                                                 const int32_t* __restrict__ bw_idx, const int32_t* __restrict__ bs_ptr, const int32_t* __restrict__ bs_idx,
                                                 const double* __restrict__ r, const double* __restrict__ cs, const double* __restrict__ scal,
                                                 double* __restrict__ A, double* __restrict__ g) {
  const int N = p.N, n2 = 2 * N, nv = p.nv, ncv = p.nv - 3, ncu = p.nu - 3;
  const int l1 = blockIdx.x, lane = threadIdx.x;
  const int iu1 = l1 / nv, iv1 = l1 % nv;
  const double rho1 = scal[1] * scal[1];        // scal[1] = sqrt(rho')
  for (int j = lane; j < n2; j += 64) { A[(size_t)l1 * n2 + j] = 0.0; A[(size_t)(N + l1) * n2 + j] = 0.0; }
  __syncthreads();
  if (lane < 49) {
    const int iu2 = iu1 + lane / 7 - 3, iv2 = iv1 + lane % 7 - 3;
    if (iu2 >= 0 && iu2 < p.nu && iv2 >= 0 && iv2 < nv) {
      const int l2 = iu2 * nv + iv2;
      double wxx = 0.0, sxx = 0.0, sxy = 0.0, syx = 0.0, syy = 0.0;
      for (int Iu = max(max(iu1, iu2) - 3, 0); Iu <= min(min(iu1, iu2), ncu - 1); Iu++)
        for (int Iv = max(max(iv1, iv2) - 3, 0); Iv <= min(min(iv1, iv2), ncv - 1); Iv++) {
          const int cell = Iu * ncv + Iv;
          const int a = 4 * (iu1 - Iu) + (iv1 - Iv), b = 4 * (iu2 - Iu) + (iv2 - Iv);
          const int w0 = bw_ptr[cell], w1 = bw_ptr[cell + 1];
#pragma unroll 4
          for (int q = w0; q < w1; q++) { const size_t i = (size_t)bw_idx[q] * 16; wxx = fma(Jw[i + a], Jw[i + b], wxx); }
          const int s0 = bs_ptr[cell], s1 = bs_ptr[cell + 1];
          for (int q = s0; q < s1; q++) {
            const double* Jk = Js + (size_t)bs_idx[q] * 128;
#pragma unroll
            for (int rw = 0; rw < 4; rw++) {
              const double xa = Jk[32 * rw + a], ya = Jk[32 * rw + 16 + a], xb = Jk[32 * rw + b], yb = Jk[32 * rw + 16 + b];
              sxx = fma(xa, xb, sxx); sxy = fma(xa, yb, sxy); syx = fma(ya, xb, syx); syy = fma(ya, yb, syy);
            }
          }
        }
      const double c1x = cs[l1], c1y = cs[N + l1], c2x = cs[l2], c2y = cs[N + l2];
      A[(size_t)l1 * n2 + l2] = (2.0 * rho1 * wxx + sxx) * (c1x * c2x);
      A[(size_t)l1 * n2 + N + l2] = sxy * (c1x * c2y);
      A[(size_t)(N + l1) * n2 + l2] = syx * (c1y * c2x);
      A[(size_t)(N + l1) * n2 + N + l2] = syy * (c1y * c2y);
    }
  }
  // g: sixteen lanes, one per knot cell that holds l1; their partial sums are added in cell order
  double gx = 0.0, gy = 0.0;
  if (lane < 16) {
    const int Iu = iu1 - 3 + lane / 4, Iv = iv1 - 3 + lane % 4;
    if (Iu >= 0 && Iu < ncu && Iv >= 0 && Iv < ncv) {
      const int cell = Iu * ncv + Iv, a = 4 * (iu1 - Iu) + (iv1 - Iv);
      double gw = 0.0;
      for (int q = bw_ptr[cell]; q < bw_ptr[cell + 1]; q++) { const int i = bw_idx[q]; gw = fma(Jw[(size_t)i * 16 + a], r[i] + r[i + p.P], gw); }
      gx = rho1 * gw;      // (sqrt(rho') J)^T (sqrt(rho') r) for the x row and its copy, the y row
      const double* rs = r + 2 * p.P;
      for (int q = bs_ptr[cell]; q < bs_ptr[cell + 1]; q++) {
        const int k = bs_idx[q];
        const double* Jk = Js + (size_t)k * 128;
#pragma unroll
        for (int rw = 0; rw < 4; rw++) { gx = fma(Jk[32 * rw + a], rs[rw * N + k], gx); gy = fma(Jk[32 * rw + 16 + a], rs[rw * N + k], gy); }
      }
    }
  }
  double tx = 0.0, ty = 0.0;
  for (int j = 0; j < 16; j++) { tx += __shfl(gx, j, 64); ty += __shfl(gy, j, 64); }
  if (lane == 0) { g[l1] = tx * cs[l1]; g[N + l1] = ty * cs[N + l1]; }
}

// Fixed-tree block sum: lane t sums its contiguous chunk in ascending order, then a binary tree over the 256 partial
// sums (bit-reproducible run to run; differs from a sequential sum only in the last bits).
__device__ double swp_block_sum256(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const double tot = red[0];
  __syncthreads();
  return tot;
}

// scal[0] = cost = 1/2 (rho(|r_warp|^2) + |r_schw|^2), scal[1] = sqrt(rho'); one 256-thread workgroup.
__device__ __forceinline__ void swp_loss_body(int P2, int m, const double* __restrict__ r, double* __restrict__ scal) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  const double a = 5.77;   // HuberLoss(5.77), SchwarpDatabase.cc:208
  double s1 = 0.0, s2 = 0.0;
  {
    const int ch = (P2 + 255) / 256;
    for (int i = t * ch; i < min(P2, (t + 1) * ch); i++) s1 += r[i] * r[i];
    const int m2 = m - P2, ch2 = (m2 + 255) / 256;
    for (int i = t * ch2; i < min(m2, (t + 1) * ch2); i++) s2 += r[P2 + i] * r[P2 + i];
  }
  const double sq = swp_block_sum256(s1, red);
  const double rest = swp_block_sum256(s2, red);
  if (t == 0) {
    double rho0 = sq, rho1 = 1.0;
    if (sq > a * a) { const double rt = sqrt(sq); rho0 = 2 * a * rt - a * a; rho1 = a / rt; }
    scal[0] = (rho0 + rest) * 0.5;
    scal[1] = sqrt(rho1);
  }
}

__device__ __forceinline__ void swp_scale_body(int P2, int n2, const double* __restrict__ scal, double* __restrict__ r, double* __restrict__ J) {
  const double sc = scal[1];
  const size_t tot = (size_t)P2 * n2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) J[i] *= sc;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)P2; i += (size_t)gridDim.x * blockDim.x) r[i] *= sc;
}

// A = (J S)^T (J S) (lower and mirrored), g = (J S)^T r on FP64 MFMA: one wavefront per 16x16 tile of A (lower triangle),
// J is read straight from global memory as both operands (lane (i, k) reads J[row0 + k][tile + i]: 16 consecutive doubles
// of 4 rows per MFMA), four accumulators in flight; the four wavefronts of a workgroup split the rows of J (split-K) and
// their partial tiles are added in a fixed order (bit-reproducible run to run).  The column scaling S = diag(cs) is
// applied to the finished tile.  Tiles of block column 0 also accumulate g on the vector ALU.
__device__ __forceinline__ void swp_normal_body(int m, int n2, int nt, const double* __restrict__ J, const double* __restrict__ r,
                                                         const double* __restrict__ cs, double* __restrict__ A, double* __restrict__ g) {
  __shared__ double part[3][4][64];
  __shared__ double gpart[3][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x;
  int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
  while (ti * (ti + 1) / 2 > t) ti--;
  const int tj = t - ti * (ti + 1) / 2;
  const int ta = 16 * ti, tb = 16 * tj;
  const int i = lane & 15, k = lane >> 4;
  const bool va = ta + i < n2, vb = tb + i < n2;
  v4d acc[4];
#pragma unroll
  for (int u = 0; u < 4; u++) acc[u] = (v4d){0.0, 0.0, 0.0, 0.0};
  double gacc = 0.0;
  const double* Ja = J + ta + i;
  const double* Jb = J + tb + i;
  for (int r0 = 16 * wave; r0 < m; r0 += 64) {
    double av[4], bv[4], rv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int row = r0 + 4 * u + k;
      const bool vr = row < m;
      av[u] = (vr && va) ? Ja[(size_t)row * n2] : 0.0;
      bv[u] = (vr && vb) ? Jb[(size_t)row * n2] : 0.0;
      rv[u] = (tj == 0 && vr) ? r[row] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc[u], 0, 0, 0);
      gacc = fma(av[u], rv[u], gacc);
    }
  }
  v4d tot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (tj == 0) {   // g[ta + i]: the four k-groups hold partial sums of disjoint rows
    gacc += __shfl_xor(gacc, 16, 64);
    gacc += __shfl_xor(gacc, 32, 64);
  }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) part[wave - 1][q][lane] = tot[q];
    if (lane < 16) gpart[wave - 1][lane] = gacc;
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; w++) {
#pragma unroll
    for (int q = 0; q < 4; q++) tot[q] += part[w][q][lane];
    if (lane < 16) gacc += gpart[w][lane];
  }
  const int g4 = lane >> 4, c = lane & 15;
  const double csb = (tb + c < n2) ? cs[tb + c] : 0.0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ra = ta + g4 + 4 * q, cb = tb + c;
    if (ra < n2 && cb < n2 && cb <= ra) {
      const double v = tot[q] * (cs[ra] * csb);
      A[(size_t)ra * n2 + cb] = v;
      A[(size_t)cb * n2 + ra] = v;
    }
  }
  if (tj == 0 && lane < 16 && va) g[ta + i] = gacc * cs[ta + i];
}

// One workgroup (8 wavefronts): M = A + diag(clamp(diag A)/radius) padded to np = 16*ceil(n/16) with an identity block,
// right-looking Cholesky on 16x16 FP64 MFMA tiles (M stays in global memory: 1.2 MB at n = 390, L2 resident):
//   per block column K: wave 0 factors the diagonal tile and inverts the factor in registers (chol_inv_blocked, W = L^-1)
//   and finishes block K of the forward substitution z_K = W y_K; TRSM X_I = A_IK W^T (4 MFMAs per tile, tiles over the
//   waves) folds y_I -= X_I z_K in as soon as a tile is known; trailing A_IJ -= X_I X_J^T (4 MFMAs per tile, 4 in flight);
// then the backward substitution L^T dx = z (operands of the next block prefetched), and the model decrease
// -(dx.g + 1/2 dx^T A dx).  out[0] = ok, out[1] = model.  Winv: np x 16 doubles, row-major W tiles.  Needs np <= 512.
#define SWS_TP 17
#define SWS_TILE (16 * SWS_TP)
// M = A + diag(clamp(diag A) / radius), padded to np x np with an identity block (whole GPU: one workgroup would be
// load-latency bound on this 1.2 MB copy)
// The damped matrix / its factor M is kept as 16x16 tiles in MFMA accumulator order (tile (I,J) at (I*NT + J)*256, element
// (row, col) at ((row&3)*16 + col)*4 + (row>>2)): the tile Cholesky moves every tile as two 16-byte accesses per lane.
__device__ __forceinline__ size_t swp_mi(int np, int r, int c) {
  return ((size_t)(r >> 4) * (np >> 4) + (c >> 4)) * 256 + ((((r & 15) & 3) << 4) + (c & 15)) * 4 + ((r & 15) >> 2);
}
// Unknown ordering inside the solver.  A two-coordinate problem (Schwarp: n = 2N, first coordinates then second ones) is
// interleaved (x0, y0, x1, y1, ...): control points couple within a 4 x 4 patch of the grid, so the interleaved normal matrix is
// banded (half-bandwidth 2 (3 nptsv + 3) + 1) and the factorisation only visits the tiles of the band.
__device__ __forceinline__ int swp_perm(int n, int il, int i) { return (il && i < n) ? ((i < n / 2) ? 2 * i : 2 * (i - n / 2) + 1) : i; }
__device__ __forceinline__ void swp_damp_body(int n, int np, int il, const double* __restrict__ A, double radius, double* __restrict__ M) {
  const int cc = blockIdx.x * 256 + threadIdx.x, rr = blockIdx.y;
  if (cc >= np) return;
  double v = (rr == cc) ? 1.0 : 0.0;
  if (rr < n && cc < n) {
    v = A[(size_t)rr * n + cc];
    if (rr == cc) v += fmin(fmax(v, 1e-6), 1e32) / radius;
  }
  M[swp_mi(np, swp_perm(n, il, rr), swp_perm(n, il, cc))] = v;
}
// sum over the 16 lanes of a row group (lanes sharing l >> 4), result in every lane of the group
__device__ __forceinline__ double swp_row16_sum(double v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
// Workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains the global stores (the L tiles that nobody reads
// before the end of the factorisation), a round trip to L2 in every step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Factor loop of the solve for a band of at most 7 sub-diagonal tiles (the Schwarp fit: exactly 7; Warp::initialize: 3): the trailing
// window lives in REGISTERS -- wavefront w owns the tile rows I = w (mod 8), tile (I, J) transposed in acc[J & 7], loaded once from the
// natural-order matrix A by the wave that owns the row (eight steps before it is needed: a second register set keeps the row after next in
// flight, a row fetched one step ahead made every step wait for its gather).  With T^T in accumulator layout every product takes registers as they are:
//   TRSM    X_I^T = W_K T(I,K)^T        A = W_K (LDS), B = acc[K & 7]
//   update  T(I,J)^T -= X_J X_I^T       A = X_J (LDS, written by the owner of row J in lane order: the reader lane reads what the same lane
//                                        wrote), B = X_I^T (the TRSM result, still in registers); J = I: A = the same registers
// so a step is two barriers and no global round trip (the version below that keeps M in memory: four dependent ones, 7 us per step
// against 1.8).  L tiles are stored (fire and forget) in the layout the back substitution reads.  Same products in the same order as the
// memory version: the factor is bit-identical, the forward substitution sums in another order.
// The rows come straight from A (natural ordering, row-major n x n): element (r, c) of the interleaved, damped, identity-padded matrix
// is A[unperm(r)][unperm(c)] (+ clamp(diag) / radius on the diagonal) -- the damp kernel and its 1.3 MB copy per solve are not needed.
__device__ __forceinline__ void swp_factor_band8(int n, int np, int il, const double* __restrict__ A_, double radius, double* __restrict__ M_,
                                                 double* __restrict__ Winv_, double* Xl, double* Wk, double* yv, int* bad) {
  // The pointers come out of a descriptor in memory: generic address space, i.e. FLAT loads and stores, which count on the LDS counter as
  // well -- every LDS wait (the barriers of the loop) would then wait for the rows in flight.  Global-address-space views:
  typedef __attribute__((address_space(1))) double gdbl;
  const gdbl* A = (const gdbl*)A_;
  gdbl* M = (gdbl*)M_;
  gdbl* Winv = (gdbl*)Winv_;
  const int NT = np / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int crow = lane >> 4, ccol = lane & 15;
  v4d acc[8];
  const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
  auto unperm = [&](int i) { return il ? ((i & 1) ? n / 2 + (i >> 1) : (i >> 1)) : i; };
  v4d nxt[8];                                        // the row after next of this wave, in flight for eight steps
  auto load_row = [&](v4d (&acc)[8], int I) {
    const int c = 16 * I + ccol, pc = unperm(c);
#pragma unroll
    for (int s8 = 0; s8 < 8; s8++) {
      // slot s8 holds column J = the one of I-7 .. I with J & 7 == s8; acc[s8][q] = element (16 J + crow + 4 q, 16 I + ccol)
      const int J = I - ((I - s8) & 7);
      v4d t = zero4;
      if (J >= 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int r = 16 * J + crow + 4 * q;
          t[q] = (r < n && c < n) ? A[(size_t)unperm(r) * n + pc] : ((r == c) ? 1.0 : 0.0);
        }
      }
      acc[s8] = t;
    }
    // the damping of the diagonal tile (slot I & 7 == wave), behind ALL the loads: written into the load loop it made the compiler wait for
    // every element before the next one was requested -- 32 round trips in a row per tile row
#pragma unroll
    for (int s8 = 0; s8 < 8; s8++)
      if (s8 == wave && crow == (ccol & 3) && c < n) {
        const int qd = ccol >> 2;
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (q == qd) acc[s8][q] += fmin(fmax(acc[s8][q], 1e-6), 1e32) / radius;
      }
  };
  auto factor_diag = [&](int K, v4d a) {
    v4d w;
    if (!chol_inv_blocked(a, w) && lane == 0) *bad = 1;
    const double yk = yv[16 * K + ccol];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      Wk[(crow + 4 * q) * SWS_TP + ccol] = w[q];                        // W row-major: the TRSM reads it as the A operand
      Winv[(size_t)(16 * K + crow + 4 * q) * 16 + ccol] = w[q];
    }
    double z[4];
#pragma unroll
    for (int q = 0; q < 4; q++) z[q] = swp_row16_sum(w[q] * yk);      // z_K = W y_K, row crow + 4q
    if (ccol == 0) {
#pragma unroll
      for (int q = 0; q < 4; q++) yv[16 * K + crow + 4 * q] = z[q];
    }
  };
  int myrow = wave;                                  // the row whose tiles acc holds
  if (myrow < NT) load_row(acc, myrow);
  if (myrow + 8 < NT) load_row(nxt, myrow + 8);
  if (wave == 0) factor_diag(0, acc[0]);
  __syncthreads();
  for (int K = 0; K < NT; K++) {
    v4d xt = zero4;
    const bool pivot_row = myrow == K;
    const bool active = myrow > K && myrow < NT;     // K < myrow <= K + 7
    if (pivot_row) {
      myrow = K + 8;                                  // this wave's next row has been on its way since step K - 8; the one after it starts now
#pragma unroll
      for (int s8 = 0; s8 < 8; s8++) acc[s8] = nxt[s8];
      if (myrow + 8 < NT) load_row(nxt, myrow + 8);
    } else if (active) {
      const int I = myrow;
      double a[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) a[kk] = Wk[ccol * SWS_TP + crow + 4 * kk];
      v4d b = zero4;                                  // tile (I, K): slot K & 7 (wave-uniform branches, no run-time register index)
#pragma unroll
      for (int s8 = 0; s8 < 8; s8++)
        if (s8 == (K & 7)) b = acc[s8];
      v4d x = zero4, x2 = zero4;
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], x, 0, 0, 0);
      x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], x2, 0, 0, 0);
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], x, 0, 0, 0);
      x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], x2, 0, 0, 0);
      xt = x + x2;                                    // xt[q] = X_I[ccol][crow + 4q]
      gdbl* Lt = M + ((size_t)I * NT + K) * 256 + (size_t)((ccol & 3) * 16 + crow) * 4 + (ccol >> 2);
      double p = 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        Lt[16 * q] = xt[q];                           // element (ccol, crow + 4q) of tile (I, K), accumulator-order storage
        Xl[(size_t)(I & 7) * 256 + q * 64 + lane] = xt[q];
        p = fma(xt[q], yv[16 * K + crow + 4 * q], p);
      }
      p += __shfl_xor(p, 16, 64);
      p += __shfl_xor(p, 32, 64);
      if (crow == 0) yv[16 * I + ccol] -= p;          // forward substitution of block row I: only this wave touches it in this step
    }
    lds_barrier();
    if (active) {
      const int I = myrow;
      // the diagonal tile first: the owner of row K+1 factors it while the longer rows are still being updated
      v4d d = zero4;
#pragma unroll
      for (int s8 = 0; s8 < 8; s8++)
        if (s8 == wave) d = acc[s8];                  // I & 7 == wave
#pragma unroll
      for (int kk = 0; kk < 4; kk++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(-xt[kk], xt[kk], d, 0, 0, 0);
      if (I == K + 1) {
        __builtin_amdgcn_s_setprio(3);                // the chain of the factorisation runs through this tile: ahead of the wave it shares the SIMD with
        factor_diag(I, d);
        __builtin_amdgcn_s_setprio(0);
      } else {
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {
          if (s8 == wave) acc[s8] = d;
          const int J = I - ((I - s8) & 7);           // the column slot s8 holds for row I; X_J sits in slot J & 7 == s8 of Xl as well
          if (J > K && J < I) {
            v4d t = acc[s8];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) t = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xl[(size_t)s8 * 256 + kk * 64 + lane], xt[kk], t, 0, 0, 0);
            acc[s8] = t;
          }
        }
      }
    }
    lds_barrier();
  }
}

// gnv > 0: A is the normal matrix of a two-coordinate problem on a control grid with gnv points along v (rows of A outside the 7 x 7
// neighbourhood of their control point are zero: the model decrease only reads that neighbourhood); 0: dense rows
__device__ __forceinline__ void swp_solve_body(int n, int np, int il, int bwt, const double* __restrict__ A_, const double* __restrict__ g_, double radius,
                                                        double* __restrict__ M_, double* __restrict__ Winv_, double* __restrict__ dx_, double* __restrict__ out_, int gnv = 0) {
  // (global-address-space views: pointers read from a descriptor are generic, and flat accesses also count on the LDS counter)
  typedef __attribute__((address_space(1))) double gdbl;
  typedef __attribute__((address_space(1))) v4d gv4d;
  const gdbl* A = (const gdbl*)A_;
  const gdbl* g = (const gdbl*)g_;
  gdbl* M = (gdbl*)M_;
  gdbl* Winv = (gdbl*)Winv_;
  gdbl* dx = (gdbl*)dx_;
  gdbl* out = (gdbl*)out_;
  // il: interleaved unknown ordering (swp_perm); bwt: sub-diagonal tiles of the band (NT - 1: dense)
  extern __shared__ double sws[];
  const int NT = np / 16;
  double* Xp = sws;                       // NT panel tiles, k-major padded: Xp[T*SWS_TILE + k*SWS_TP + i] = X_T[i][k]
  double* Wk = Xp + (size_t)max(NT * SWS_TILE, 8 * 256);  // W^T of the current block, k-major: Wk[k*SWS_TP + j] = W[j][k] (band version: W row-major; its X ring takes 8 x 256 doubles of Xp)
  double* yv = Wk + SWS_TILE;             // np: right-hand side -> forward-substituted -> solution
  double* red = yv + np;                  // 16 partial sums
  __shared__ int bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int crow = lane >> 4, ccol = lane & 15;
  for (int i = tid; i < np; i += 512) yv[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < n; i += 512) yv[swp_perm(n, il, i)] = -g[i];     // (wide bands: M was prepared by swp_damp_kernel)
  if (tid == 0) bad = 0;
  __syncthreads();
  // Diagonal tile K: Cholesky + inverse (wave 0), W^T into LDS for the TRSM, z_K = W y_K.  `a` = the updated tile.
  auto factor_diag = [&](int K, v4d a) {
    v4d w;
    if (!chol_inv_blocked(a, w) && lane == 0) bad = 1;
    const double yk = yv[16 * K + ccol];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      Wk[ccol * SWS_TP + crow + 4 * q] = w[q];
      Winv[(size_t)(16 * K + crow + 4 * q) * 16 + ccol] = w[q];
    }
    double z[4];
#pragma unroll
    for (int q = 0; q < 4; q++) z[q] = swp_row16_sum(w[q] * yk);     // z_K = W y_K, row crow + 4q
    if (ccol == 0) {
#pragma unroll
      for (int q = 0; q < 4; q++) yv[16 * K + crow + 4 * q] = z[q];
    }
  };
  if (bwt <= 7) swp_factor_band8(n, np, il, A_, radius, M_, Winv_, Xp, Wk, yv, &bad);   // (wave-uniform: the whole workgroup takes one path)
  if (bwt > 7 && wave == 0) factor_diag(0, *reinterpret_cast<const gv4d*>(M + 4 * lane));
  __syncthreads();
  for (int K = 0; K < (bwt > 7 ? NT : 0); K++) {
    // TRSM: X_I = A_IK W^T, and the forward substitution of block row I: y_I -= X_I z_K
    const double zk = yv[16 * K + ccol];
    constexpr int TU = 4;   // tiles of this wave in flight (NT <= 32)
    const int Iend = min(NT, K + 1 + bwt);          // tile rows below the band hold zeros: not visited
    for (int I0 = K + 1 + wave; I0 < Iend; I0 += 8 * TU) {
      double av[TU][4], bv[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) bv[kk] = Wk[(4 * kk + crow) * SWS_TP + ccol];                        // B[k][j] = W[j][k]
#pragma unroll
      for (int u = 0; u < TU; u++) {
        const int I = min(I0 + 8 * u, Iend - 1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) av[u][kk] = M[swp_mi(np, 16 * I + ccol, 16 * K + 4 * kk + crow)];   // A operand: lane (i = ccol, k = crow)
      }
#pragma unroll
      for (int u = 0; u < TU; u++) {
        const int I = I0 + 8 * u;
        if (I >= Iend) break;
        v4d x = {0.0, 0.0, 0.0, 0.0}, x2 = x;
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][0], bv[0], x, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][1], bv[1], x2, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][2], bv[2], x, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][3], bv[3], x2, 0, 0, 0);
        x += x2;
        *reinterpret_cast<gv4d*>(M + ((size_t)I * NT + K) * 256 + 4 * lane) = x;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          Xp[(size_t)I * SWS_TILE + ccol * SWS_TP + crow + 4 * q] = x[q];
          const double d = swp_row16_sum(x[q] * zk);
          if (ccol == 0) yv[16 * I + crow + 4 * q] -= d;     // only this wave touches block row I in this step
        }
      }
    }
    __syncthreads();
    // Trailing update of the lower triangle: tiles (I, J), K < J <= I.  Wave 0 looks ahead: it updates the next diagonal
    // tile and factors it while the other seven waves update the rest, so no step waits for a Cholesky.  The other tile rows
    // are dealt from both ends (row lengths grow linearly: a long row is paired with a short one); the A operand of a
    // row is read once, 4 tiles of the row are in flight.
    const int ntr = Iend - 1 - K;
    if (wave == 0) {
      if (ntr > 0) {
        const int I = K + 1;
        v4d acc = *reinterpret_cast<const gv4d*>(M + ((size_t)I * NT + I) * 256 + 4 * lane);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const double xv = Xp[(size_t)I * SWS_TILE + (4 * kk + crow) * SWS_TP + ccol];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xv, xv, acc, 0, 0, 0);
        }
        factor_diag(I, acc);
      }
    } else {
      const int R = ntr - 1;                             // tile rows K+2 .. NT-1
      for (int p = wave - 1; p < (R + 1) / 2; p += 7) {
#pragma unroll 1
        for (int side = 0; side < 2; side++) {
          const int rr = side == 0 ? p : R - 1 - p;
          if (side == 1 && rr == p) break;
          const int I = K + 2 + rr;
          double an[4];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) an[kk] = -Xp[(size_t)I * SWS_TILE + (4 * kk + crow) * SWS_TP + ccol];
          constexpr int UNR = 4;
          for (int J0 = K + 1; J0 <= I; J0 += UNR) {
            v4d acc[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
              const int J = min(J0 + u, I);
              acc[u] = *reinterpret_cast<const gv4d*>(M + ((size_t)I * NT + J) * 256 + 4 * lane);
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
#pragma unroll
              for (int u = 0; u < UNR; u++) {
                const int J = min(J0 + u, I);
                const double bb = Xp[(size_t)J * SWS_TILE + (4 * kk + crow) * SWS_TP + ccol];
                acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[kk], bb, acc[u], 0, 0, 0);
              }
            }
#pragma unroll
            for (int u = 0; u < UNR; u++)
              if (J0 + u <= I) *reinterpret_cast<gv4d*>(M + ((size_t)I * NT + J0 + u) * 256 + 4 * lane) = acc[u];
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- backward substitution L^T x = z, right-looking: x_K = W_K^T z_K, then z_c -= L[16K.., c]^T x_K for c < 16K.
  //      Thread c owns column c (np <= 512); the operands of block K-1 are fetched while block K is processed. ----------
  double wt[16], lc[16];
  auto fetch = [&](int K) {
    if (K < 0) return;
    if (tid < 16) {
#pragma unroll
      for (int j = 0; j < 16; j++) wt[j] = Winv[(size_t)(16 * K + j) * 16 + tid];     // column tid of W_K = row of W_K^T
    }
    if (tid < 16 * K && tid >= 16 * (K - bwt)) {
#pragma unroll
      for (int j = 0; j < 16; j++) lc[j] = M[swp_mi(np, 16 * K + j, tid)];
    }
  };
  fetch(NT - 1);
  for (int K = NT - 1; K >= 0; K--) {
    double z = 0.0;
    if (tid < 16) {
#pragma unroll
      for (int j = 0; j < 16; j++) z = fma(wt[j], yv[16 * K + j], z);
    }
    double lcur[16];
#pragma unroll
    for (int j = 0; j < 16; j++) lcur[j] = lc[j];
    const bool upd = tid < 16 * K && tid >= 16 * (K - bwt);
    __syncthreads();
    if (tid < 16) yv[16 * K + tid] = z;
    fetch(K - 1);
    __syncthreads();
    if (upd) {
      double sacc = yv[tid];
#pragma unroll
      for (int j = 0; j < 16; j++) sacc = fma(-lcur[j], yv[16 * K + j], sacc);
      yv[tid] = sacc;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += 512) dx[i] = yv[swp_perm(n, il, i)];
  // ---- model decrease -(dx.g + 1/2 dx^T A dx): a wave per row, lanes across the columns --------------------------
  double part = 0.0;
  if (gnv > 0) {
    // two-coordinate grid problem (Schwarp): row a only holds the 7 x 7 neighbourhood of its control point in both coordinate blocks.  One
    // THREAD per row (n <= 512): its 2 x 49 loads are independent and go out together (a wave per row and four rows in flight spent 46 us on
    // thirteen dependent round trips); products added in neighbourhood order, x block then y block.
    const int N = n / 2, gnu = N / gnv;
    const int a = tid;
    if (a < n) {
      const int l1 = a < N ? a : a - N;
      const int iu1 = l1 / gnv, iv1 = l1 % gnv;
      const gdbl* row = A + (size_t)a * n;
      double t = 0.0;
#pragma unroll 1
      for (int blk = 0; blk < 2; blk++) {
        double av[49];
#pragma unroll
        for (int e = 0; e < 49; e++) {
          const int iu2 = iu1 + e / 7 - 3, iv2 = iv1 + e % 7 - 3;
          const bool in = iu2 >= 0 && iu2 < gnu && iv2 >= 0 && iv2 < gnv;
          av[e] = in ? row[blk * N + iu2 * gnv + iv2] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 49; e++) {
          const int iu2 = iu1 + e / 7 - 3, iv2 = iv1 + e % 7 - 3;
          const bool in = iu2 >= 0 && iu2 < gnu && iv2 >= 0 && iv2 < gnv;
          if (in) t = fma(av[e], yv[swp_perm(n, il, blk * N + iu2 * gnv + iv2)], t);
        }
      }
      part = yv[swp_perm(n, il, a)] * (g[a] + 0.5 * t);
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
  } else
  for (int a0 = 4 * wave; a0 < n; a0 += 32) {   // four rows per wave in flight
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b2 = lane; b2 < n; b2 += 64) {
      const double xb = yv[swp_perm(n, il, b2)];
#pragma unroll
      for (int u = 0; u < 4; u++) t[u] = fma((a0 + u < n) ? A[(size_t)(a0 + u) * n + b2] : 0.0, xb, t[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      for (int o = 32; o > 0; o >>= 1) t[u] += __shfl_down(t[u], o, 64);
      if (lane == 0 && a0 + u < n) part += yv[swp_perm(n, il, a0 + u)] * (g[a0 + u] + 0.5 * t[u]);
    }
  }
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int w = 0; w < 8; w++) tot += red[w];
    const double model = -tot;
    out[1] = bad ? 0.0 : model;
    out[0] = (bad || !(model > 0)) ? 0.0 : 1.0;
  }
}

// Substitutions only, with the factor left in M / Winv by swp_solve_kernel: M dx = -g (iterative refinement of the
// Shape-from-Normals least squares).  One workgroup, np <= 512.
__device__ __forceinline__ void swp_resolve_body(int n, int np, int il, int bwt, const double* __restrict__ g, const double* __restrict__ M,
                                                 const double* __restrict__ Winv, double* __restrict__ dx) {
  __shared__ double yv[512];
  const int tid = threadIdx.x, NT = np / 16;
  yv[tid] = 0.0;
  __syncthreads();
  if (tid < n) yv[swp_perm(n, il, tid)] = -g[tid];
  __syncthreads();
  for (int K = 0; K < NT; K++) {
    double z = 0.0;
    if (tid < 16) {
#pragma unroll
      for (int k = 0; k < 16; k++) z = fma(Winv[(size_t)(16 * K + tid) * 16 + k], yv[16 * K + k], z);
    }
    double lr[16];
    const bool upd = tid >= 16 * (K + 1) && tid < min(np, 16 * (K + 1 + bwt));
    if (upd) {
#pragma unroll
      for (int j = 0; j < 16; j++) lr[j] = M[swp_mi(np, tid, 16 * K + j)];
    }
    __syncthreads();
    if (tid < 16) yv[16 * K + tid] = z;
    __syncthreads();
    if (upd) {
      double sacc = yv[tid];
#pragma unroll
      for (int j = 0; j < 16; j++) sacc = fma(-lr[j], yv[16 * K + j], sacc);
      yv[tid] = sacc;
    }
    __syncthreads();
  }
  for (int K = NT - 1; K >= 0; K--) {
    double z = 0.0;
    if (tid < 16) {
#pragma unroll
      for (int j = 0; j < 16; j++) z = fma(Winv[(size_t)(16 * K + j) * 16 + tid], yv[16 * K + j], z);
    }
    double lc[16];
    const bool upd = tid < 16 * K && tid >= 16 * (K - bwt);
    if (upd) {
#pragma unroll
      for (int j = 0; j < 16; j++) lc[j] = M[swp_mi(np, 16 * K + j, tid)];
    }
    __syncthreads();
    if (tid < 16) yv[16 * K + tid] = z;
    __syncthreads();
    if (upd) {
      double sacc = yv[tid];
#pragma unroll
      for (int j = 0; j < 16; j++) sacc = fma(-lc[j], yv[16 * K + j], sacc);
      yv[tid] = sacc;
    }
    __syncthreads();
  }
  if (tid < n) dx[tid] = yv[swp_perm(n, il, tid)];
}
__global__ __launch_bounds__(512) void swp_resolve_kernel(int n, int np, int il, int bwt, const double* __restrict__ g, const double* __restrict__ M,
                                                          const double* __restrict__ Winv, double* __restrict__ dx) {
  swp_resolve_body(n, np, il, bwt, g, M, Winv, dx);
}

// ------------------------------------------------------------------------------------------------
// Shape from Normals (Modules/Mapping/ShapeFromNormals.cc): stacked least squares [M; Bending; 1^T] x = [0; 0; N mean]
// ------------------------------------------------------------------------------------------------
// Rows of obtainM (ShapeFromNormals.cc:178-260) for one site per lane: row k = (n.eta) coloc_du + n_x coloc,
// row k + n = (n.eta) coloc_dv + n_y coloc; 16 taps each, A is m x N row-major and zeroed beforehand.
__global__ void sfn_rows_kernel(BbsPar p, int n, const double* __restrict__ u, const double* __restrict__ v, const float* __restrict__ normals, int N,
                                double* __restrict__ A) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double nu, nv, b0u[4], b1u[4], b0v[4], b1v[4];
  int Iu, Iv;
  norm_inter(p.umin, p.umax, p.nptsu, u[k], nu, Iu);
  norm_inter(p.vmin, p.vmax, p.nptsv, v[k], nv, Iv);
  if (Iu < 0 || Iu > p.nptsu - 4 || Iv < 0 || Iv > p.nptsv - 4) return;   // outside the definition domain: no constraint
  cubic_basis(0, nu, b0u); cubic_basis(1, nu, b1u);
  cubic_basis(0, nv, b0v); cubic_basis(1, nv, b1v);
  const double fu = deriv_fact(p, 1, 0), fv = deriv_fact(p, 0, 1);
  double nx = normals[3 * k], ny = normals[3 * k + 1], nz = normals[3 * k + 2];
  const double nn = sqrt(nx * nx + ny * ny + nz * nz);
  nx /= nn; ny /= nn; nz /= nn;
  const double ne = nx * u[k] + ny * v[k] + nz;
  for (int iu = 0; iu < 4; iu++)
    for (int iv = 0; iv < 4; iv++) {
      const int col = (iu + Iu) * p.nptsv + iv + Iv;
      const double w0 = b0u[iu] * b0v[iv], wu = fu * b1u[iu] * b0v[iv], wv = fv * b0u[iu] * b1v[iv];
      A[(size_t)k * N + col] = ne * wu + nx * w0;
      A[(size_t)(k + n) * N + col] = ne * wv + ny * w0;
    }
}

// Colocation rows of Warps::Warp::initialize (Schwarp.cc:136-139): C[k, :] = 16 B-spline weights of key point k (float32 pair),
// rhs[k] = -kp2 coordinate `coord` (negated: swp_solve_kernel returns M dx = -g).  C is P x N row-major and zeroed beforehand.
__device__ __forceinline__ void warp_coloc_body(BbsPar p, int P, const float* __restrict__ kp1, const float* __restrict__ kp2, int N, double* __restrict__ Cm,
                                                double* __restrict__ rhs0, double* __restrict__ rhs1) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  rhs0[k] = -(double)kp2[2 * k];
  rhs1[k] = -(double)kp2[2 * k + 1];
  double nu, nv, bu[4], bv[4];
  int Iu, Iv;
  norm_inter(p.umin, p.umax, p.nptsu, (double)kp1[2 * k], nu, Iu);
  norm_inter(p.vmin, p.vmax, p.nptsv, (double)kp1[2 * k + 1], nv, Iv);
  if (Iu < 0 || Iu > p.nptsu - 4 || Iv < 0 || Iv > p.nptsv - 4) return;
  cubic_basis(0, nu, bu);
  cubic_basis(0, nv, bv);
  for (int iu = 0; iu < 4; iu++)
    for (int iv = 0; iv < 4; iv++) Cm[(size_t)k * N + (iu + Iu) * p.nptsv + iv + Iv] = bu[iu] * bv[iv];
}
__global__ void warp_coloc_kernel(BbsPar p, int P, const float* __restrict__ kp1, const float* __restrict__ kp2, int N, double* __restrict__ Cm,
                                  double* __restrict__ rhs0, double* __restrict__ rhs1) {
  warp_coloc_body(p, P, kp1, kp2, N, Cm, rhs0, rhs1);
}

__global__ void mat_add_kernel(size_t n, const double* __restrict__ B, double* __restrict__ A) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) A[i] += B[i];
}

// ------------------------------------------------------------------------------------------------
// Warp-guided match search (DefORBmatcher::searchBySchwarp, Modules/Matching/DefORBmatcher.cc:189-294)
// ------------------------------------------------------------------------------------------------
struct MatchPar { float fx, fy, cx, cy, minX, maxX, minY, maxY, winv, hinv, radius; int cols, rows, th_low; };

// grid cell of every key point of keyframe 2 (Frame::PosInGrid, Frame.cc:484-496): cell = px * rows + py or -1 (not in the grid)
__global__ void match_cells_kernel(MatchPar p, int N2, const float* __restrict__ kp2, int32_t* __restrict__ cell) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N2) return;
  const int px = (int)roundf((kp2[2 * j] - p.minX) * p.winv), py = (int)roundf((kp2[2 * j + 1] - p.minY) * p.hinv);
  cell[j] = (px < 0 || px >= p.cols || py < 0 || py >= p.rows) ? -1 : px * p.rows + py;
}

// One wavefront per query: prediction through the warp (16 taps, float32 key point like Warp::getEstimates), image and window
// tests of KeyFrame::GetFeaturesInArea, 256-bit Hamming distance to every candidate.  The reference walks the grid cells
// column by column and keeps the first strictly better candidate; a brute-force scan reproduces that with the key
// (distance, cell, index): lanes scan candidates j = lane, lane + 64, ..., then a wave-wide lexicographic minimum.
__global__ __launch_bounds__(256) void match_search_kernel(BbsPar b, MatchPar p, const double* __restrict__ x, int Q, const float* __restrict__ kp1,
                                                           const uint32_t* __restrict__ desc1, int N2, const float* __restrict__ kp2,
                                                           const uint32_t* __restrict__ desc2, const uint8_t* __restrict__ has_mp2,
                                                           const int32_t* __restrict__ cell, int32_t* __restrict__ match) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= Q) return;
  const int N = b.nptsu * b.nptsv;
  double nu, nv, bu[4], bv[4];
  int Iu, Iv;
  const double u = kp1[2 * q], v = kp1[2 * q + 1];
  norm_inter(b.umin, b.umax, b.nptsu, u, nu, Iu);
  norm_inter(b.vmin, b.vmax, b.nptsv, v, nv, Iv);
  cubic_basis(0, nu, bu);
  cubic_basis(0, nv, bv);
  double ax = 0.0, ay = 0.0;
  if (!(Iu < 0 || Iu > b.nptsu - 4 || Iv < 0 || Iv > b.nptsv - 4)) {
    for (int iu = 0; iu < 4; iu++)
      for (int iv = 0; iv < 4; iv++) {
        const double bas = bu[iu] * bv[iv];
        const int c = (iu + Iu) * b.nptsv + iv + Iv;
        ax += x[c] * bas;
        ay += x[N + c] * bas;
      }
  }
  const float ex = (float)ax, ey = (float)ay;
  const float px = ex * p.fx + p.cx, py = ey * p.fy + p.cy;
  int result = -1;
  bool live = px >= p.minX && px < p.maxX && py >= p.minY && py < p.maxY;
  int c0 = 0, c1 = 0, r0 = 0, r1 = 0;
  if (live) {
    c0 = max(0, (int)floorf((px - p.minX - p.radius) * p.winv));
    c1 = min(p.cols - 1, (int)ceilf((px - p.minX + p.radius) * p.winv));
    r0 = max(0, (int)floorf((py - p.minY - p.radius) * p.hinv));
    r1 = min(p.rows - 1, (int)ceilf((py - p.minY + p.radius) * p.hinv));
    live = c0 < p.cols && c1 >= 0 && r0 < p.rows && r1 >= 0;
  }
  if (live) {
    uint32_t d1[8];
#pragma unroll
    for (int w = 0; w < 8; w++) d1[w] = desc1[8 * (size_t)q + w];
    // key = distance (9 bits) | cell (13 bits at most 64*48) | index: smaller is better, exactly the reference's visiting order among equal distances
    unsigned long long best = ~0ull;
    for (int j = lane; j < N2; j += 64) {
      const int cj = cell[j];
      if (cj < 0 || has_mp2[j]) continue;
      const int cx = cj / p.rows, cy = cj - cx * p.rows;
      if (cx < c0 || cx > c1 || cy < r0 || cy > r1) continue;
      const float dx = kp2[2 * j] - px, dy = kp2[2 * j + 1] - py;
      if (!(fabsf(dx) < p.radius && fabsf(dy) < p.radius)) continue;
      int dist = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) dist += __popc(d1[w] ^ desc2[8 * (size_t)j + w]);
      if (dist >= p.th_low) continue;
      const unsigned long long key = ((unsigned long long)dist << 48) | ((unsigned long long)cj << 32) | (unsigned)j;
      best = key < best ? key : best;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(best, o, 64);
      best = other < best ? other : best;
    }
    if (best != ~0ull) result = (int)(best & 0xFFFFFFFFull);
  }
  if (lane == 0) match[q] = result;
}

// out[i] = sign * (b[i] - A[i,:] x): one wavefront per row
__global__ __launch_bounds__(256) void sfn_residual_kernel(int m, int N, const double* __restrict__ A, const double* __restrict__ x, const double* __restrict__ b,
                                                           double sign, double* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  double s = 0.0;
  for (int j = lane; j < N; j += 64) s = fma(A[(size_t)row * N + j], x[j], s);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) out[row] = sign * (b[row] - s);
}

__global__ void sfn_axpy_kernel(int n, const double* __restrict__ dx, double* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += dx[i];
}

// Surface points (ShapeFromNormals.cc:150-166): depth d = BBS eval of the scaled control points, float32 (u d, v d, d)
__global__ void sfn_points_kernel(BbsPar p, const double* __restrict__ ctrl, int n, const double* __restrict__ u, const double* __restrict__ v,
                                  float* __restrict__ pts) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double nu, nv, bu[4], bv[4];
  int Iu, Iv;
  norm_inter(p.umin, p.umax, p.nptsu, u[k], nu, Iu);
  norm_inter(p.vmin, p.vmax, p.nptsv, v[k], nv, Iv);
  cubic_basis(0, nu, bu);
  cubic_basis(0, nv, bv);
  double d = 0.0;
  if (!(Iu < 0 || Iu > p.nptsu - 4 || Iv < 0 || Iv > p.nptsv - 4)) {
    for (int iu = 0; iu < 4; iu++)
      for (int iv = 0; iv < 4; iv++) d += ctrl[(iu + Iu) * p.nptsv + iv + Iv] * (bu[iu] * bv[iv]);
  }
  pts[3 * k] = (float)(u[k] * d);
  pts[3 * k + 1] = (float)(v[k] * d);
  pts[3 * k + 2] = (float)d;
}

// xn = x + dx*cs; out[2] = |step|, out[3] = |x|, out[4] = max |g|; one 256-thread workgroup
__device__ __forceinline__ void swp_step_body(int n, const double* __restrict__ x, const double* __restrict__ dx, const double* __restrict__ cs,
                                                       const double* __restrict__ g, double* __restrict__ xn, double* __restrict__ out) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  double sn = 0, xnrm = 0, gm = 0;
  const int ch = (n + 255) / 256;
  for (int j = t * ch; j < min(n, (t + 1) * ch); j++) {
    const double st = dx[j] * cs[j];
    xn[j] = x[j] + st;
    sn += st * st;
    xnrm += x[j] * x[j];
    gm = fmax(gm, fabs(g[j]));
  }
  const double snt = swp_block_sum256(sn, red), xt = swp_block_sum256(xnrm, red);
  red[t] = gm;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] = fmax(red[t], red[t + o]);
    __syncthreads();
  }
  if (t == 0) { out[2] = sqrt(snt); out[3] = sqrt(xt); out[4] = red[0]; }
}

__device__ __forceinline__ void swp_colscale_body(int n, const double* __restrict__ A, double* __restrict__ cs) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) cs[j] = 1.0 / (1.0 + sqrt(A[(size_t)j * n + j]));
}

// DiffProp records of the fitted warp (SchwarpDatabase.cc:243-345): six evaluations -> float32 key points
__device__ __forceinline__ void swp_diffprop_body(SwpPar p, const float* __restrict__ kp1, const float* __restrict__ kp2, const double* __restrict__ x,
                                    float fx_true, float fy_true, float* __restrict__ diff, uint8_t* __restrict__ drop) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.P) return;
  const double u = kp1[2 * i], v = kp1[2 * i + 1];
  double ax, ay;
  float qe[2], dqu[2], dqv[2], dquv[2], dquu[2], dqvv[2];
  swp_eval16(p, x, u, v, 0, 0, ax, ay); qe[0] = (float)ax; qe[1] = (float)ay;
  swp_eval16(p, x, u, v, 1, 0, ax, ay); dqu[0] = (float)ax; dqu[1] = (float)ay;
  swp_eval16(p, x, u, v, 0, 1, ax, ay); dqv[0] = (float)ax; dqv[1] = (float)ay;
  swp_eval16(p, x, u, v, 1, 1, ax, ay); dquv[0] = (float)ax; dquv[1] = (float)ay;
  swp_eval16(p, x, u, v, 2, 0, ax, ay); dquu[0] = (float)ax; dquu[1] = (float)ay;
  swp_eval16(p, x, u, v, 0, 2, ax, ay); dqvv[0] = (float)ax; dqvv[1] = (float)ay;
  float ex = qe[0] - kp2[2 * i], ey = qe[1] - kp2[2 * i + 1];
  ex *= fx_true; ey *= fy_true;
  drop[i] = sqrt((double)ex * ex + (double)ey * ey) > 10 ? 1 : 0;
  float* d = diff + 18 * (size_t)i;
  d[0] = kp1[2 * i]; d[1] = kp1[2 * i + 1]; d[2] = kp2[2 * i]; d[3] = kp2[2 * i + 1];
  d[4] = dqu[0]; d[5] = dqu[1]; d[6] = dqv[0]; d[7] = dqv[1];
  const float det = dqu[0] * dqv[1] - dqv[0] * dqu[1];
  d[8] = d[7] / det; d[9] = -d[6] / det; d[10] = -d[5] / det; d[11] = d[4] / det;
  d[12] = dquu[0]; d[13] = dquu[1]; d[14] = dquv[0]; d[15] = dquv[1]; d[16] = dqvv[0]; d[17] = dqvv[1];
}

// ---- thin kernels over the bodies above (one fit per launch: dsh_schwarp_eval, Shape from Normals, warp initialisation) ----------
template <bool WITH_J>
__global__ void swp_eval_kernel(SwpPar p, const float* __restrict__ kp1, const float* __restrict__ kp2, const float* __restrict__ invsig,
                                const double* __restrict__ x, double* __restrict__ r, double* __restrict__ J) { swp_eval_body<WITH_J>(p, kp1, kp2, invsig, x, r, J); }
__global__ __launch_bounds__(256) void swp_loss_kernel(int P2, int m, const double* __restrict__ r, double* __restrict__ scal) { swp_loss_body(P2, m, r, scal); }
__global__ void swp_scale_kernel(int P2, int n2, const double* __restrict__ scal, double* __restrict__ r, double* __restrict__ J) { swp_scale_body(P2, n2, scal, r, J); }
__global__ __launch_bounds__(256) void swp_normal_kernel(int m, int n2, int nt, const double* __restrict__ J, const double* __restrict__ r,
                                                         const double* __restrict__ cs, double* __restrict__ A, double* __restrict__ g) { swp_normal_body(m, n2, nt, J, r, cs, A, g); }
__global__ __launch_bounds__(256) void swp_damp_kernel(int n, int np, int il, const double* __restrict__ A, double radius, double* __restrict__ M) { swp_damp_body(n, np, il, A, radius, M); }
__global__ __launch_bounds__(512) void swp_solve_kernel(int n, int np, int il, int bwt, const double* __restrict__ A, const double* __restrict__ g, double radius,
                                                        double* __restrict__ M, double* __restrict__ Winv, double* __restrict__ dx, double* __restrict__ out) { swp_solve_body(n, np, il, bwt, A, g, radius, M, Winv, dx, out); }
__global__ __launch_bounds__(256) void swp_step_kernel(int n, const double* __restrict__ x, const double* __restrict__ dx, const double* __restrict__ cs,
                                                       const double* __restrict__ g, double* __restrict__ xn, double* __restrict__ out) { swp_step_body(n, x, dx, cs, g, xn, out); }
__global__ void swp_colscale_kernel(int n, const double* __restrict__ A, double* __restrict__ cs) { swp_colscale_body(n, A, cs); }
__global__ void swp_diffprop_kernel(SwpPar p, const float* __restrict__ kp1, const float* __restrict__ kp2, const double* __restrict__ x,
                                    float fx_true, float fy_true, float* __restrict__ diff, uint8_t* __restrict__ drop) { swp_diffprop_body(p, kp1, kp2, x, fx_true, fy_true, diff, drop); }

// ------------------------------------------------------------------------------------------------
// Batched Schwarp fit (SchwarpDatabase::add fits one warp per anchor keyframe, SchwarpDatabase.cc:50-128): B fits advance
// together, one launch per stage with the fit in blockIdx.y (blockIdx.z for the 2D damp grid); each kernel reads its arguments
// from the fit's descriptor.  The trust-region control of the reference's Ceres run (dsh_schwarp.cpp restated it on the host
// with one round trip per iteration) runs in swp_ctl_kernel on the device: the whole batch is a fixed sequence of launches
// without a single host synchronisation, and a finished fit skips its stages by a flag.
// ------------------------------------------------------------------------------------------------
struct SwpFit {
  SwpPar p;                      // domain, grid, P, N, slots, lambda
  float fx, fy;                  // true focal lengths (DiffProp drop test)
  int n2, m, np, il, bwt, max_iters;
  const float *kp1, *kp2, *isg;
  double *x, *xn, *cs, *g, *dx, *r, *J, *A, *M, *W, *scal;   // scal: [0] cost [1] sqrt(rho') [2] solve ok [3] model change [4] |step| [5] |x| [6] max |g|
  float* diff;
  uint8_t* drop;
  int32_t* info;                 // [0] iterations [1] accepted steps
  double* costs;                 // [0] initial [1] final
  // trust-region state (Ceres LM as restated in dsh_schwarp.cpp / oracle/schwarp_oracle.c)
  double radius, nu, cost, cost0, change, old;
  int it, good, invalid, done, accepted, pending;   // pending: an accepted step was re-linearised, its max |g| has not been tested yet
  // optional first stage, Warps::Warp::initialize (Schwarp.cc:99-160): x = the regularised linear fit of the warp with this bending
  // matrix (N x N, shared by the fits of one grid and weight; NULL: x holds the caller's start value).  It borrows the buffers of
  // the fit: C in J, the two right-hand sides in r, C^T C + Bending in A, C^T kp2 in g, the factor in M / W.
  const double* bend;
  int npi, bwti;                 // padded size and band width (in tiles) of the N x N system
  // Structured Jacobian of the fit (every row touches the 4 x 4 patch of control points of ONE knot cell -- SURVEY 7 K11): the warp rows
  // as 16 values per match (the x row; the reference's y row is a copy of it, Schwarp.cc:291-298), the Schwarzian rows as 32 values per
  // row (16 for each coordinate), and the rows bucketed by knot cell (matches in index order: a fixed summation order).
  double *Jw, *Js;               // P x 16;  N x 4 x 32 (site, row, [x taps | y taps])
  int32_t *bw_ptr, *bw_idx;      // ncell + 1, P: matches of cell (Iu, Iv) = Iu * (nv - 3) + Iv
  int32_t *bs_ptr, *bs_idx;      // ncell + 1, N: grid sites of the cell
  int32_t* cid;                  // P + N: knot cell of every match / grid site (-1: outside the domain)
};
#define SWP_STAGE_ALWAYS 0      // setup stages: run for every fit
#define SWP_STAGE_ACTIVE 1      // stages of an iteration: skipped once the fit is done
#define SWP_STAGE_ACCEPTED 2    // re-linearisation: only after an accepted step

__device__ __forceinline__ bool swp_on(const SwpFit& f, int stage) {
  return stage == SWP_STAGE_ALWAYS || (stage == SWP_STAGE_ACTIVE && !f.done) || (stage == SWP_STAGE_ACCEPTED && f.accepted);
}

template <bool WITH_J>
__global__ void swpb_eval_kernel(const SwpFit* fits, int stage, int at_xn) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage)) return;
  swp_eval_body<WITH_J>(f.p, f.kp1, f.kp2, f.isg, at_xn ? f.xn : f.x, f.r, f.J);
}
__global__ void swpb_zero_j_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage) || !f.bend) return;      // (only the Warp::initialize stage has a dense buffer)
  const size_t tot = (size_t)f.m * f.n2 / 2;
  v2d_t* J2 = reinterpret_cast<v2d_t*>(f.J);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) J2[i] = (v2d_t){0.0, 0.0};
}
__global__ __launch_bounds__(256) void swpb_loss_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage)) return;
  swp_loss_body(2 * f.p.P, f.m, f.r, f.scal);
}
__global__ void swpb_scale_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage)) return;
  swp_scale_body(2 * f.p.P, f.n2, f.scal, f.r, f.J);
}
__global__ __launch_bounds__(256) void swpb_normal_kernel(const SwpFit* fits, int stage, int nt) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage)) return;
  swp_normal_body(f.m, f.n2, nt, f.J, f.r, f.cs, f.A, f.g);
}
__global__ void swpb_evalc_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage)) return;
  swp_evalc_body(f.p, f.kp1, f.kp2, f.isg, f.x, f.r, f.Jw, f.Js);
}
__global__ void swpb_cellid_kernel(const SwpFit* fits) {
  const SwpFit& f = fits[blockIdx.y];
  swp_cellid_body(f.p, f.kp1, f.cid);
}
__global__ __launch_bounds__(256) void swpb_buckets_kernel(const SwpFit* fits, int use_lds) {
  const SwpFit& f = fits[blockIdx.y];
  swp_buckets_body(f.p, f.cid, f.bw_ptr, f.bw_idx, f.bs_ptr, f.bs_idx, use_lds != 0);
}
__global__ __launch_bounds__(64) void swpb_normalc_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage) || (int)blockIdx.x >= f.p.N) return;
  swp_normalc_body(f.p, f.Jw, f.Js, f.bw_ptr, f.bw_idx, f.bs_ptr, f.bs_idx, f.r, f.cs, f.scal, f.A, f.g);
}
__global__ void swpb_colscale_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage)) return;
  swp_colscale_body(f.n2, f.A, f.cs);
}
// The Jacobi scaling enters the normal equations as a factor on sums that do not depend on it (swp_normalc_body: sum * (c1 * c2), g: sum * c):
// the equations linearised with cs = 1 become the scaled ones by those factors -- the same bits as a second linearisation at the same x.
__global__ __launch_bounds__(256) void swpb_rescale_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.z];
  const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (!swp_on(f, stage) || i >= f.n2 || j >= f.n2) return;
  const double ci = f.cs[i];
  f.A[(size_t)i * f.n2 + j] *= ci * f.cs[j];
  if (j == 0) f.g[i] *= ci;
}
__global__ __launch_bounds__(256) void swpb_damp_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.z];
  if (!swp_on(f, stage) || (int)blockIdx.y >= f.np || f.bwt <= 7) return;   // (the band solver reads A itself)
  swp_damp_body(f.n2, f.np, f.il, f.A, f.radius, f.M);
}
__global__ __launch_bounds__(512) void swpb_solve_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage)) return;
  swp_solve_body(f.n2, f.np, f.il, f.bwt, f.A, f.g, f.radius, f.M, f.W, f.dx, f.scal + 2, f.il ? f.p.nv : 0);
}
__global__ __launch_bounds__(256) void swpb_step_kernel(const SwpFit* fits, int stage) {
  const SwpFit& f = fits[blockIdx.y];
  if (!swp_on(f, stage)) return;
  swp_step_body(f.n2, f.x, f.dx, f.cs, f.g, f.xn, f.scal + 2);
}
__global__ void swpb_diffprop_kernel(const SwpFit* fits) {
  const SwpFit& f = fits[blockIdx.y];
  if (!f.diff || !f.drop) return;
  swp_diffprop_body(f.p, f.kp1, f.kp2, f.x, f.fx, f.fy, f.diff, f.drop);
}
// ---- Warp::initialize as a first stage of the batch (fits with f.bend) ----
__global__ void wib_coloc_kernel(const SwpFit* fits) {
  const SwpFit& f = fits[blockIdx.y];
  if (!f.bend) return;
  warp_coloc_body(BbsPar{f.p.umin, f.p.umax, f.p.vmin, f.p.vmax, f.p.nu, f.p.nv, 2, 0}, f.p.P, f.kp1, f.kp2, f.p.N, f.J, f.r, f.r + f.p.P);
}
__global__ __launch_bounds__(256) void wib_normal_kernel(const SwpFit* fits, int second, int nt) {   // C^T C and C^T (-kp2 coordinate)
  const SwpFit& f = fits[blockIdx.y];
  if (!f.bend) return;
  swp_normal_body(f.p.P, f.p.N, nt, f.J, f.r + (second ? f.p.P : 0), f.cs, f.A, f.g + (second ? f.p.N : 0));
}
__global__ void wib_bend_kernel(const SwpFit* fits) {
  const SwpFit& f = fits[blockIdx.y];
  if (!f.bend) return;
  const size_t tot = (size_t)f.p.N * f.p.N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) f.A[i] += f.bend[i];
}
__global__ __launch_bounds__(256) void wib_damp_kernel(const SwpFit* fits) {
  const SwpFit& f = fits[blockIdx.z];
  if (!f.bend || (int)blockIdx.y >= f.npi || f.bwti <= 7) return;
  swp_damp_body(f.p.N, f.npi, 0, f.A, 1e300, f.M);
}
__global__ __launch_bounds__(512) void wib_solve_kernel(const SwpFit* fits) {
  const SwpFit& f = fits[blockIdx.y];
  if (!f.bend) return;
  swp_solve_body(f.p.N, f.npi, 0, f.bwti, f.A, f.g, 1e300, f.M, f.W, f.x, f.scal + 2);
}
__global__ __launch_bounds__(512) void wib_resolve_kernel(const SwpFit* fits) {
  const SwpFit& f = fits[blockIdx.y];
  if (!f.bend) return;
  swp_resolve_body(f.p.N, f.npi, 0, f.bwti, f.g + f.p.N, f.M, f.W, f.x + f.p.N);
}
// info[2] = the reference's verdict on the initialisation (positive definite system, finite control points); the scalars go back to
// zero for the fit
__global__ void wib_finish_kernel(const SwpFit* fits, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const SwpFit& f = fits[b];
  if (!f.bend) { f.info[2] = 1; return; }
  bool finite = true;
  for (int i = 0; i < 2 * f.p.N; i++) finite = finite && isfinite(f.x[i]);
  f.info[2] = ((f.scal[2] != 0.0 || f.scal[3] != 0.0) && finite) ? 1 : 0;
  for (int i = 0; i < 16; i++) f.scal[i] = 0.0;
}

// x <- xn after an accepted step
__global__ void swpb_accept_kernel(SwpFit* fits) {
  const SwpFit& f = fits[blockIdx.y];
  if (!f.accepted) return;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < f.n2; j += gridDim.x * blockDim.x) f.x[j] = f.xn[j];
}

// The trust-region controller, one thread per fit.  phase 0: after the scaled linearisation of the start (and a zero step that
// measures max |g|); 1: top of an iteration; 5: behind solve + step -- the gradient test of the previous accepted step (the step
// kernel measures max |g| of the current linearisation); 2: after the trial evaluation (accept / reject, radius update);
// 3: after the re-linearisation of an accepted step (function tolerance); 4: results.  Same decisions, in the same order, as
// the host loop this replaces (oracle/schwarp_oracle.c is the restatement both are tested against).
__global__ void swpb_ctl_kernel(SwpFit* fits, int B, int phase) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  SwpFit& f = fits[b];
  const double ftol = 1e-6, gtol = 1e-10, ptol = 1e-8, min_rel_dec = 1e-3;
  double* s = f.scal;
  if (phase == 0) {
    f.cost = s[0]; f.cost0 = s[0];
    f.radius = 1e4; f.nu = 2.0;
    f.it = 0; f.good = 0; f.invalid = 0; f.accepted = 0; f.pending = 0;
    f.done = !(s[6] > gtol) || f.max_iters <= 0;
  } else if (phase == 1) {
    f.accepted = 0;
    if (!f.done) {
      if (f.it >= f.max_iters) f.done = 1;
      else f.it++;
    }
  } else if (phase == 5) {
    if (!f.done && f.pending && s[6] <= gtol) { f.it--; f.done = 1; }   // the previous iteration had already converged
    f.pending = 0;
  } else if (phase == 2) {
    f.accepted = 0;
    if (f.done) return;
    const bool ok = s[2] != 0.0;
    const double model = s[3];
    if (!ok) { if (++f.invalid >= 5) f.done = 1; f.radius *= 0.5; return; }
    f.invalid = 0;
    if (s[4] <= ptol * (s[5] + ptol)) { f.done = 1; return; }
    const double cost_new = s[0];
    const double rel = (f.cost - cost_new) / model;
    if (rel > min_rel_dec) {
      f.change = f.cost - cost_new; f.old = f.cost;
      f.radius = fmin(1e16, f.radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
      f.nu = 2.0;
      f.good++;
      f.cost = cost_new;
      f.accepted = 1;
    } else {
      f.radius /= f.nu; f.nu *= 2.0;
      if (f.radius < 1e-32) f.done = 1;
    }
  } else if (phase == 3) {
    if (!f.accepted) return;
    f.pending = 1;
    if (fabs(f.change) <= ftol * f.old) f.done = 1;   // function tolerance, tested behind the re-linearisation like on the host
  } else {
    if (f.info) { f.info[0] = f.it; f.info[1] = f.good; }
    if (f.costs) { f.costs[0] = f.cost0; f.costs[1] = f.cost; }
  }
}

}  // namespace

// ---- launchers (called from dsh_nrsfm.cpp) ---------------------------------------------------------
extern "C" hipError_t nrsfm_launch_bbs_eval(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, int valdim, const double* ctrl,
                                            const double* u, const double* v, int n, int du, int dv, double* val, uint8_t* outside, hipStream_t st) {
  BbsPar p = {umin, umax, vmin, vmax, nptsu, nptsv, valdim, 0};
  const size_t bytes = sizeof(double) * (size_t)valdim * nptsu * nptsv;
  const int use_lds = bytes <= 64 * 1024;
  const int block = 256;
  const int grid = n > 0 ? (n + block - 1) / block : 1;
  hipLaunchKernelGGL(bbs_eval_kernel, dim3(grid < 2048 ? grid : 2048), dim3(block), use_lds ? bytes : 0, st, p, ctrl, u, v, n, du, dv, val, outside, use_lds);
  return hipGetLastError();
}

extern "C" hipError_t nrsfm_launch_bbs_coloc(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, const double* u, const double* v,
                                             int n, int du, int dv, int32_t* cols, double* w, int32_t* n_outside, hipStream_t st) {
  BbsPar p = {umin, umax, vmin, vmax, nptsu, nptsv, 1, 0};
  const int block = 256;
  const int grid = n > 0 ? (n + block - 1) / block : 1;
  hipLaunchKernelGGL(bbs_coloc_kernel, dim3(grid < 2048 ? grid : 2048), dim3(block), 0, st, p, u, v, n, du, dv, cols, w, n_outside);
  return hipGetLastError();
}

extern "C" hipError_t nrsfm_launch_normals(int P, int R, const int32_t* rec_ptr, const int32_t* rec_point, const float* recs, const uint8_t* is_ref,
                                           const float* first_n, const uint8_t* has_first_n, const float* x0, const uint8_t* has_x0, const float* ref_uv,
                                           double* Q, double* k1k2, double* cov, int32_t* status, float* normal_ref, float* normal_rec, uint8_t* written,
                                           int32_t* iters, hipStream_t st) {
  const int block = 128;
  if (R > 0) hipLaunchKernelGGL(normals_coeff_kernel, dim3((R + block - 1) / block), dim3(block), 0, st, R, recs, is_ref, Q);
  if (P > 0)
    hipLaunchKernelGGL(normals_solve_kernel, dim3((P + block - 1) / block), dim3(block), 0, st, P, R, rec_ptr, is_ref, Q, x0, has_x0, ref_uv, k1k2, cov, status,
                       normal_ref, iters);
  if (R > 0)
    hipLaunchKernelGGL(normals_propagate_kernel, dim3((R + block - 1) / block), dim3(block), 0, st, R, recs, rec_point, is_ref, first_n, has_first_n, k1k2,
                       status, normal_rec, written);
  return hipGetLastError();
}

// ---- Schwarp launchers ------------------------------------------------------------------------------
extern "C" hipError_t nrsfm_swp_eval(double umin, double umax, int nu, double vmin, double vmax, int nv, int P, double fxs, double fys, double lambda,
                                     const float* kp1, const float* kp2, const float* invsig, const double* x, double* r, double* J, int with_j, hipStream_t st) {
  SwpPar p = {umin, umax, vmin, vmax, fxs, fys, lambda, nu, nv, nu * nv, P};
  const int tot = P + p.N, block = 128;
  if (with_j) {
    hipError_t e = hipMemsetAsync(J, 0, sizeof(double) * (size_t)(2 * P + 4 * p.N) * 2 * p.N, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(swp_eval_kernel<true>, dim3((tot + block - 1) / block), dim3(block), 0, st, p, kp1, kp2, invsig, x, r, J);
  } else {
    hipLaunchKernelGGL(swp_eval_kernel<false>, dim3((tot + block - 1) / block), dim3(block), 0, st, p, kp1, kp2, invsig, x, r, J);
  }
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_loss(int P2, int m, const double* r, double* scal, hipStream_t st) {
  hipLaunchKernelGGL(swp_loss_kernel, dim3(1), dim3(256), 0, st, P2, m, r, scal);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_normal(int P2, int m, int n2, double* J, double* r, const double* cs, const double* scal, double* A, double* g, hipStream_t st) {
  hipLaunchKernelGGL(swp_scale_kernel, dim3(256), dim3(256), 0, st, P2, n2, scal, r, J);
  const int nt = (n2 + 15) / 16, tiles = nt * (nt + 1) / 2;
  hipLaunchKernelGGL(swp_normal_kernel, dim3(tiles), dim3(256), 0, st, m, n2, nt, J, r, cs, A, g);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_colscale(int n2, const double* A, double* cs, hipStream_t st) {
  hipLaunchKernelGGL(swp_colscale_kernel, dim3((n2 + 127) / 128), dim3(128), 0, st, n2, A, cs);
  return hipGetLastError();
}
// M: np*np doubles, Winv: np*16 doubles with np = nrsfm_swp_solve_np(n2)
extern "C" int nrsfm_swp_solve_np(int n2) { return 16 * ((n2 + 15) / 16); }
// interleave: two-coordinate unknown vector (first coordinates, then second ones) reordered inside the solver; kd: scalar
// half-bandwidth of the (reordered) matrix, >= n for a dense one.
extern "C" hipError_t nrsfm_swp_solve(int n2, const double* A, const double* g, double radius, double* M, double* Winv, double* dx, double* out, int interleave,
                                      int kd, hipStream_t st) {
  const int np = nrsfm_swp_solve_np(n2), NT = np / 16;
  const int bwt = min(NT - 1, (max(kd, 0) + 15) / 16);
  const size_t lds = sizeof(double) * ((size_t)max(NT * SWS_TILE, 8 * 256) + SWS_TILE + np + 16);
  if (lds > 150 * 1024 || np > 512) return hipErrorInvalidValue;   // one thread per unknown in the backward substitution
  {   // the attribute is per device: set every time (microseconds) rather than cached per process
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(swp_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (bwt > 7) hipLaunchKernelGGL(swp_damp_kernel, dim3((np + 255) / 256, np), dim3(256), 0, st, n2, np, interleave, A, radius, M);   // (the band solver reads A itself)
  hipLaunchKernelGGL(swp_solve_kernel, dim3(1), dim3(512), lds, st, n2, np, interleave, bwt, A, g, radius, M, Winv, dx, out);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_step(int n2, const double* x, const double* dx, const double* cs, const double* g, double* xn, double* out, hipStream_t st) {
  hipLaunchKernelGGL(swp_step_kernel, dim3(1), dim3(256), 0, st, n2, x, dx, cs, g, xn, out);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_diffprop(double umin, double umax, int nu, double vmin, double vmax, int nv, int P, const float* kp1, const float* kp2,
                                         const double* x, float fx_true, float fy_true, float* diff, uint8_t* drop, hipStream_t st) {
  SwpPar p = {umin, umax, vmin, vmax, 0.0, 0.0, 0.0, nu, nv, nu * nv, P};
  hipLaunchKernelGGL(swp_diffprop_kernel, dim3((P + 127) / 128), dim3(128), 0, st, p, kp1, kp2, x, fx_true, fy_true, diff, drop);
  return hipGetLastError();
}

// The batched Schwarp fit: a fixed sequence of launches over B fit descriptors (device array), no host synchronisation inside.
// maxP / maxN / maxn2 / maxnp: the largest sizes in the batch (grid extents); max_iters: the largest iteration limit.
extern "C" size_t nrsfm_swp_fit_bytes() { return sizeof(SwpFit); }
extern "C" void nrsfm_swp_fit_fill(void* host_slot, double umin, double umax, int nu, double vmin, double vmax, int nv, int P, double fxs, double fys, double lambda,
                                   float fx, float fy, int max_iters, const float* kp1, const float* kp2, const float* isg, double* x, double* xn, double* cs,
                                   double* g, double* dx, double* r, double* J, double* A, double* M, double* W, double* scal, float* diff, uint8_t* drop,
                                   int32_t* info, double* costs, const double* bend, void* compact) {
  SwpFit f{};
  f.p = SwpPar{umin, umax, vmin, vmax, fxs, fys, lambda, nu, nv, nu * nv, P};
  f.fx = fx; f.fy = fy;
  f.n2 = 2 * nu * nv; f.m = 2 * P + 4 * nu * nv; f.np = nrsfm_swp_solve_np(f.n2); f.il = 1;
  f.bwt = min(f.np / 16 - 1, (2 * (3 * nv + 3) + 1 + 15) / 16);
  f.max_iters = max_iters;
  f.kp1 = kp1; f.kp2 = kp2; f.isg = isg; f.x = x; f.xn = xn; f.cs = cs; f.g = g; f.dx = dx; f.r = r; f.J = J; f.A = A; f.M = M; f.W = W; f.scal = scal;
  f.diff = diff; f.drop = drop; f.info = info; f.costs = costs;
  f.bend = bend;
  f.npi = nrsfm_swp_solve_np(nu * nv);
  f.bwti = min(f.npi / 16 - 1, (3 * nv + 3 + 15) / 16);   // colocation and bending couple a 4 x 4 patch of control points
  {   // structured Jacobian + row buckets (nrsfm_swp_compact_bytes)
    const int N = nu * nv, ncell = (nu - 3) * (nv - 3);
    char* cb = static_cast<char*>(compact);
    f.Jw = reinterpret_cast<double*>(cb); cb += 8 * (size_t)P * 16;
    f.Js = reinterpret_cast<double*>(cb); cb += 8 * (size_t)N * 128;
    f.bw_ptr = reinterpret_cast<int32_t*>(cb); cb += 4 * (size_t)(ncell + 1);
    f.bw_idx = reinterpret_cast<int32_t*>(cb); cb += 4 * (size_t)P;
    f.bs_ptr = reinterpret_cast<int32_t*>(cb); cb += 4 * (size_t)(ncell + 1);
    f.bs_idx = reinterpret_cast<int32_t*>(cb); cb += 4 * (size_t)N;
    f.cid = reinterpret_cast<int32_t*>(cb);
  }
  memcpy(host_slot, &f, sizeof f);
}
extern "C" size_t nrsfm_swp_compact_bytes(int P, int nu, int nv) {
  const size_t N = (size_t)nu * nv, ncell = (size_t)(nu - 3) * (nv - 3);
  return 8 * (size_t)P * 16 + 8 * N * 128 + 4 * (ncell + 1) * 2 + 4 * (size_t)P + 4 * N + 4 * ((size_t)P + N) + 64;
}
extern "C" hipError_t nrsfm_swp_fit_batch(void* d_fits_v, int B, int maxP, int maxN, int max_iters, int with_init, hipStream_t st) {
  SwpFit* fits = static_cast<SwpFit*>(d_fits_v);
  const int maxn2 = 2 * maxN, maxnp = nrsfm_swp_solve_np(maxn2), NT = maxnp / 16;
  const size_t lds = sizeof(double) * ((size_t)max(NT * SWS_TILE, 8 * 256) + SWS_TILE + maxnp + 16);
  if (lds > 150 * 1024 || maxnp > 512) return hipErrorInvalidValue;
  {   // the attribute is per device: set every time (microseconds) rather than cached per process
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(swpb_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(wib_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const int nt = (maxn2 + 15) / 16, tiles = nt * (nt + 1) / 2;
  const dim3 g_eval((maxP + maxN + 127) / 128, B), g_one(1, B);
  (void)tiles;
  auto linearise = [&](int stage) {   // residuals + structured Jacobian at x, loss (the Huber weight of the warp block), normal equations by gather
    hipLaunchKernelGGL(swpb_evalc_kernel, g_eval, dim3(128), 0, st, fits, stage);
    hipLaunchKernelGGL(swpb_loss_kernel, g_one, dim3(256), 0, st, fits, stage);
    hipLaunchKernelGGL(swpb_normalc_kernel, dim3(maxN, B), dim3(64), 0, st, fits, stage);
  };
  hipLaunchKernelGGL(swpb_cellid_kernel, g_eval, dim3(128), 0, st, fits);
  {
    const size_t cid_bytes = sizeof(int) * (size_t)(maxP + maxN);
    const int use_lds = cid_bytes <= 48 * 1024;      // (very many matches: the cell ids stay in memory)
    hipLaunchKernelGGL(swpb_buckets_kernel, g_one, dim3(256), use_lds ? cid_bytes : 0, st, fits, use_lds);
  }
  const dim3 g_ctl((B + 63) / 64);
  if (with_init) {   // Warp::initialize for the fits that ask for it: same launches for everybody, a fit without it leaves at once
    const int npi = nrsfm_swp_solve_np(maxN), nti = (maxN + 15) / 16, tiles_i = nti * (nti + 1) / 2;
    hipLaunchKernelGGL(swpb_zero_j_kernel, dim3(64, B), dim3(256), 0, st, fits, SWP_STAGE_ALWAYS);
    hipLaunchKernelGGL(wib_coloc_kernel, dim3((maxP + 127) / 128, B), dim3(128), 0, st, fits);
    hipLaunchKernelGGL(wib_normal_kernel, dim3(tiles_i, B), dim3(256), 0, st, fits, 1, nti);
    hipLaunchKernelGGL(wib_normal_kernel, dim3(tiles_i, B), dim3(256), 0, st, fits, 0, nti);
    hipLaunchKernelGGL(wib_bend_kernel, dim3(32, B), dim3(256), 0, st, fits);
    hipLaunchKernelGGL(wib_damp_kernel, dim3((npi + 255) / 256, npi, B), dim3(256), 0, st, fits);
    hipLaunchKernelGGL(wib_solve_kernel, g_one, dim3(512), lds, st, fits);
    hipLaunchKernelGGL(wib_resolve_kernel, g_one, dim3(512), 0, st, fits);
  }
  hipLaunchKernelGGL(wib_finish_kernel, g_ctl, dim3(64), 0, st, fits, B);
  // Jacobi scaling from the initial Jacobian (cs = 1 first), then the start's normal equations rescaled; a zero step measures |x|, max |g|
  linearise(SWP_STAGE_ALWAYS);
  hipLaunchKernelGGL(swpb_colscale_kernel, dim3((maxn2 + 127) / 128, B), dim3(128), 0, st, fits, SWP_STAGE_ALWAYS);
  hipLaunchKernelGGL(swpb_rescale_kernel, dim3((maxn2 + 255) / 256, maxn2, B), dim3(256), 0, st, fits, SWP_STAGE_ALWAYS);
  hipLaunchKernelGGL(swpb_step_kernel, g_one, dim3(256), 0, st, fits, SWP_STAGE_ALWAYS);
  hipLaunchKernelGGL(swpb_ctl_kernel, g_ctl, dim3(64), 0, st, fits, B, 0);
  for (int it = 0; it <= max_iters; it++) {   // the last round only lets phase 1 retire the fits that used every iteration
    hipLaunchKernelGGL(swpb_ctl_kernel, g_ctl, dim3(64), 0, st, fits, B, 1);
    if (it == max_iters) break;
    hipLaunchKernelGGL(swpb_damp_kernel, dim3((maxnp + 255) / 256, maxnp, B), dim3(256), 0, st, fits, SWP_STAGE_ACTIVE);
    hipLaunchKernelGGL(swpb_solve_kernel, g_one, dim3(512), lds, st, fits, SWP_STAGE_ACTIVE);
    hipLaunchKernelGGL(swpb_step_kernel, g_one, dim3(256), 0, st, fits, SWP_STAGE_ACTIVE);
    hipLaunchKernelGGL(swpb_ctl_kernel, g_ctl, dim3(64), 0, st, fits, B, 5);
    hipLaunchKernelGGL(swpb_eval_kernel<false>, g_eval, dim3(128), 0, st, fits, SWP_STAGE_ACTIVE, 1);   // residuals at the trial point
    hipLaunchKernelGGL(swpb_loss_kernel, g_one, dim3(256), 0, st, fits, SWP_STAGE_ACTIVE);
    hipLaunchKernelGGL(swpb_ctl_kernel, g_ctl, dim3(64), 0, st, fits, B, 2);
    hipLaunchKernelGGL(swpb_accept_kernel, dim3(2, B), dim3(256), 0, st, fits);
    linearise(SWP_STAGE_ACCEPTED);
    hipLaunchKernelGGL(swpb_ctl_kernel, g_ctl, dim3(64), 0, st, fits, B, 3);
  }
  hipLaunchKernelGGL(swpb_diffprop_kernel, dim3((maxP + 127) / 128, B), dim3(128), 0, st, fits);
  hipLaunchKernelGGL(swpb_ctl_kernel, g_ctl, dim3(64), 0, st, fits, B, 4);
  return hipGetLastError();
}

extern "C" hipError_t nrsfm_swp_resolve(int n2, const double* g, const double* M, const double* Winv, double* dx, int interleave, int kd, hipStream_t st) {
  const int np = nrsfm_swp_solve_np(n2), NT = np / 16;
  if (np > 512) return hipErrorInvalidValue;
  const int bwt = min(NT - 1, (max(kd, 0) + 15) / 16);
  hipLaunchKernelGGL(swp_resolve_kernel, dim3(1), dim3(512), 0, st, n2, np, interleave, bwt, g, M, Winv, dx);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_sfn_rows(double umin, double umax, int nu, double vmin, double vmax, int nv, int n, const double* u, const double* v,
                                     const float* normals, double* A, hipStream_t st) {
  BbsPar p = {umin, umax, vmin, vmax, nu, nv, 1, 0};
  if (n > 0) hipLaunchKernelGGL(sfn_rows_kernel, dim3((n + 127) / 128), dim3(128), 0, st, p, n, u, v, normals, nu * nv, A);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_sfn_residual(int m, int N, const double* A, const double* x, const double* b, double sign, double* out, hipStream_t st) {
  hipLaunchKernelGGL(sfn_residual_kernel, dim3((m + 3) / 4), dim3(256), 0, st, m, N, A, x, b, sign, out);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_sfn_axpy(int n, const double* dx, double* x, hipStream_t st) {
  hipLaunchKernelGGL(sfn_axpy_kernel, dim3((n + 127) / 128), dim3(128), 0, st, n, dx, x);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_sfn_points(double umin, double umax, int nu, double vmin, double vmax, int nv, const double* ctrl, int n, const double* u,
                                       const double* v, float* pts, hipStream_t st) {
  BbsPar p = {umin, umax, vmin, vmax, nu, nv, 1, 0};
  if (n > 0) hipLaunchKernelGGL(sfn_points_kernel, dim3((n + 127) / 128), dim3(128), 0, st, p, ctrl, n, u, v, pts);
  return hipGetLastError();
}

extern "C" hipError_t nrsfm_warp_coloc(double umin, double umax, int nu, double vmin, double vmax, int nv, int P, const float* kp1, const float* kp2, double* Cm,
                                       double* rhs0, double* rhs1, hipStream_t st) {
  BbsPar p = {umin, umax, vmin, vmax, nu, nv, 1, 0};
  hipLaunchKernelGGL(warp_coloc_kernel, dim3((P + 127) / 128), dim3(128), 0, st, p, P, kp1, kp2, nu * nv, Cm, rhs0, rhs1);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_mat_add(size_t n, const double* B, double* A, hipStream_t st) {
  hipLaunchKernelGGL(mat_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, B, A);
  return hipGetLastError();
}

extern "C" hipError_t nrsfm_match_search(double umin, double umax, int nu, double vmin, double vmax, int nv, const double* x, int Q, const float* kp1,
                                         const uint32_t* desc1, const float* cam2, const float* bounds2, int cols, int rows, int N2, const float* kp2,
                                         const uint32_t* desc2, const uint8_t* has_mp2, float radius, int th_low, int32_t* cell, int32_t* match, hipStream_t st) {
  BbsPar b = {umin, umax, vmin, vmax, nu, nv, 2, 0};
  MatchPar p;
  p.fx = cam2[0]; p.fy = cam2[1]; p.cx = cam2[2]; p.cy = cam2[3];
  p.minX = bounds2[0]; p.maxX = bounds2[1]; p.minY = bounds2[2]; p.maxY = bounds2[3];
  p.winv = (float)cols / (p.maxX - p.minX); p.hinv = (float)rows / (p.maxY - p.minY);
  p.radius = radius; p.cols = cols; p.rows = rows; p.th_low = th_low;
  if (N2 > 0) hipLaunchKernelGGL(match_cells_kernel, dim3((N2 + 255) / 256), dim3(256), 0, st, p, N2, kp2, cell);
  if (Q > 0) hipLaunchKernelGGL(match_search_kernel, dim3((Q + 3) / 4), dim3(256), 0, st, b, p, x, Q, kp1, desc1, N2, kp2, desc2, has_mp2, cell, match);
  return hipGetLastError();
}
