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
__global__ void swp_eval_kernel(SwpPar p, const float* __restrict__ kp1, const float* __restrict__ kp2, const float* __restrict__ invsig,
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

// scal[0] = cost = 1/2 (rho(|r_warp|^2) + |r_schw|^2), scal[1] = sqrt(rho'), sequential sums (oracle order), one lane.
__global__ void swp_loss_kernel(int P2, int m, const double* __restrict__ r, double* __restrict__ scal) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const double a = 5.77;   // HuberLoss(5.77), SchwarpDatabase.cc:208
  double sq = 0.0;
  for (int i = 0; i < P2; i++) sq += r[i] * r[i];
  double rho0 = sq, rho1 = 1.0;
  if (sq > a * a) { const double rt = sqrt(sq); rho0 = 2 * a * rt - a * a; rho1 = a / rt; }
  double cost = rho0;
  for (int i = P2; i < m; i++) cost += r[i] * r[i];
  scal[0] = cost * 0.5;
  scal[1] = sqrt(rho1);
}

__global__ void swp_scale_kernel(int P2, int n2, const double* __restrict__ scal, double* __restrict__ r, double* __restrict__ J) {
  const double sc = scal[1];
  const size_t tot = (size_t)P2 * n2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) J[i] *= sc;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)P2; i += (size_t)gridDim.x * blockDim.x) r[i] *= sc;
}

// A = (J S)^T (J S) (lower and mirrored), g = (J S)^T r; one lane per (a,b), rows summed in ascending order.
__global__ void swp_normal_kernel(int m, int n2, const double* __restrict__ J, const double* __restrict__ r, const double* __restrict__ cs,
                                  double* __restrict__ A, double* __restrict__ g) {
  __shared__ double Ja[16][17], Jb[16][17], rr[16];
  const int ta = blockIdx.y * 16, tb = blockIdx.x * 16;
  if (tb > ta) return;
  const int la = threadIdx.y, lb = threadIdx.x;
  const int a = ta + la, b = tb + lb;
  const double csa = a < n2 ? cs[a] : 0.0, csb = b < n2 ? cs[b] : 0.0;
  double acc = 0.0, gacc = 0.0;
  for (int i0 = 0; i0 < m; i0 += 16) {
    // stage 16 rows x 16 columns of both column tiles (thread (la, lb) loads row i0+la)
    const int i = i0 + la;
    Ja[la][lb] = (i < m && ta + lb < n2) ? J[(size_t)i * n2 + ta + lb] : 0.0;
    Jb[la][lb] = (i < m && tb + lb < n2) ? J[(size_t)i * n2 + tb + lb] : 0.0;
    if (lb == 0) rr[la] = i < m ? r[i] : 0.0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const double ja = Ja[k][la], jb = Jb[k][lb];
      if (ja != 0.0) {
        const double jas = ja * csa;
        if (jb != 0.0) acc += jas * (jb * csb);
        if (tb == 0 && lb == 0) gacc += jas * rr[k];
      }
    }
    __syncthreads();
  }
  if (a < n2 && b < n2 && b <= a) { A[(size_t)a * n2 + b] = acc; A[(size_t)b * n2 + a] = acc; }
  if (tb == 0 && lb == 0 && a < n2) g[a] = gacc;
}

// One workgroup: M = A + diag(clamp(diag A)/radius), Cholesky (row-wise left-looking, oracle summation order), solve M dx = -g,
// model = -(dx.g + 1/2 dx^T A dx).  out[0] = ok, out[1] = model.
__global__ void swp_solve_kernel(int n, const double* __restrict__ A, const double* __restrict__ g, double radius, double* __restrict__ M,
                                 double* __restrict__ dx, double* __restrict__ out) {
  __shared__ int bad;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (size_t i = tid; i < (size_t)n * n; i += nt) {
    const int rr = (int)(i / n), cc = (int)(i % n);
    double v = A[i];
    if (rr == cc) v += fmin(fmax(v, 1e-6), 1e32) / radius;
    M[i] = v;
  }
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int k = 0; k < n; k++) {
    // every lane forms the pivot itself (same sequential sum), rows r > k form their entry of column k
    double d = M[(size_t)k * n + k];
    for (int j = 0; j < k; j++) d -= M[(size_t)k * n + j] * M[(size_t)k * n + j];
    if (!(d > 0)) { if (tid == 0) bad = 1; }
    const double piv = sqrt(d);
    for (int rI = k + 1 + tid; rI < n; rI += nt) {
      double v = M[(size_t)rI * n + k];
      for (int j = 0; j < k; j++) v -= M[(size_t)rI * n + j] * M[(size_t)k * n + j];
      M[(size_t)rI * n + k] = v / piv;
    }
    __syncthreads();
    if (tid == 0) M[(size_t)k * n + k] = piv;
    __syncthreads();
  }
  if (tid == 0) {
    out[0] = bad ? 0.0 : 1.0;
    out[1] = 0.0;
    if (!bad) {
      for (int i = 0; i < n; i++) { double v = -g[i]; for (int j = 0; j < i; j++) v -= M[(size_t)i * n + j] * dx[j]; dx[i] = v / M[(size_t)i * n + i]; }
      for (int i = n - 1; i >= 0; i--) { double v = dx[i]; for (int j = i + 1; j < n; j++) v -= M[(size_t)j * n + i] * dx[j]; dx[i] = v / M[(size_t)i * n + i]; }
      double dg = 0, q = 0;
      for (int a = 0; a < n; a++) {
        dg += dx[a] * g[a];
        double t = 0;
        for (int b = 0; b < n; b++) t += A[(size_t)a * n + b] * dx[b];
        q += dx[a] * t;
      }
      const double model = -(dg + 0.5 * q);
      out[1] = model;
      if (!(model > 0)) out[0] = 0.0;
    }
  }
}

// xn = x + dx*cs; out[2] = |step|, out[3] = |x|, out[4] = max |g|
__global__ void swp_step_kernel(int n, const double* __restrict__ x, const double* __restrict__ dx, const double* __restrict__ cs,
                                const double* __restrict__ g, double* __restrict__ xn, double* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double sn = 0, xnrm = 0, gm = 0;
  for (int j = 0; j < n; j++) {
    const double st = dx[j] * cs[j];
    xn[j] = x[j] + st;
    sn += st * st;
    xnrm += x[j] * x[j];
    gm = fmax(gm, fabs(g[j]));
  }
  out[2] = sqrt(sn); out[3] = sqrt(xnrm); out[4] = gm;
}

__global__ void swp_colscale_kernel(int n, const double* __restrict__ A, double* __restrict__ cs) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) cs[j] = 1.0 / (1.0 + sqrt(A[(size_t)j * n + j]));
}

// DiffProp records of the fitted warp (SchwarpDatabase.cc:243-345): six evaluations -> float32 key points
__global__ void swp_diffprop_kernel(SwpPar p, const float* __restrict__ kp1, const float* __restrict__ kp2, const double* __restrict__ x,
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
  hipLaunchKernelGGL(swp_loss_kernel, dim3(1), dim3(64), 0, st, P2, m, r, scal);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_normal(int P2, int m, int n2, double* J, double* r, const double* cs, const double* scal, double* A, double* g, hipStream_t st) {
  hipLaunchKernelGGL(swp_scale_kernel, dim3(256), dim3(256), 0, st, P2, n2, scal, r, J);
  const int nt = (n2 + 15) / 16;
  hipLaunchKernelGGL(swp_normal_kernel, dim3(nt, nt), dim3(16, 16), 0, st, m, n2, J, r, cs, A, g);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_colscale(int n2, const double* A, double* cs, hipStream_t st) {
  hipLaunchKernelGGL(swp_colscale_kernel, dim3((n2 + 127) / 128), dim3(128), 0, st, n2, A, cs);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_solve(int n2, const double* A, const double* g, double radius, double* M, double* dx, double* out, hipStream_t st) {
  hipLaunchKernelGGL(swp_solve_kernel, dim3(1), dim3(512), 0, st, n2, A, g, radius, M, dx, out);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_step(int n2, const double* x, const double* dx, const double* cs, const double* g, double* xn, double* out, hipStream_t st) {
  hipLaunchKernelGGL(swp_step_kernel, dim3(1), dim3(64), 0, st, n2, x, dx, cs, g, xn, out);
  return hipGetLastError();
}
extern "C" hipError_t nrsfm_swp_diffprop(double umin, double umax, int nu, double vmin, double vmax, int nv, int P, const float* kp1, const float* kp2,
                                         const double* x, float fx_true, float fy_true, float* diff, uint8_t* drop, hipStream_t st) {
  SwpPar p = {umin, umax, vmin, vmax, 0.0, 0.0, 0.0, nu, nv, nu * nv, P};
  hipLaunchKernelGGL(swp_diffprop_kernel, dim3((P + 127) / 128), dim3(128), 0, st, p, kp1, kp2, x, fx_true, fy_true, diff, drop);
  return hipGetLastError();
}
