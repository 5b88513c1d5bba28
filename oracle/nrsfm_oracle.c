/*
 * nrsfm_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never on the product path).
 *
 * Restatement of the per-map-point surface-normal solve of the NRSfM mapping path
 * (SURVEY.md section 8a rows B2a-B2c):
 *   polynomial coefficients ... Modules/Mapping/PolySolver.cc:50-149  (order x^3,x^2y,xy^2,y^3,x^2,xy,y^2,x,y,1)
 *   residuals + 2x2 Jacobian ... Modules/Mapping/PolySolver.cc:152-193
 *   per-point driver ............ Modules/Mapping/NormalEstimator.cc:38-229 (float32 DiffProp fields, float32
 *                                 t1/t2/e1/e2, initial guess, covariance gate, normal write-back, propagation)
 *
 * PARITY UNPINNED: the reference minimises with Ceres (un-vendored, version unpinned; README suggests vcpkg
 * "ceres[suitesparse,lapack,eigensparse,tools]"), which is not available here.  The Levenberg-Marquardt
 * trust-region loop below restates the algorithm Ceres documents for
 *   TRUST_REGION / LEVENBERG_MARQUARDT / DENSE_NORMAL_CHOLESKY, jacobi_scaling = true,
 *   initial_trust_region_radius 1e4, min_relative_decrease 1e-3, min/max_lm_diagonal 1e-6/1e32,
 *   function_tolerance 1e-10, gradient_tolerance 1e-8 (NormalEstimator.cc:139-148), parameter_tolerance 1e-8,
 *   max_num_iterations 200, radius /= max(1/3, 1-(2 rho-1)^3) on success, radius /= nu, nu *= 2 on failure,
 * and ceres::Covariance as "(J^T J)^-1 unless J is rank deficient (reciprocal condition < 1e-14)".
 * It is anchored by known-answer tests (common roots of both cubics on synthetic scenes), not by Ceres bits.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NREC_F 18 /* floats per DiffProp record: I1u I1v I2u I2v J12a J12b J12c J12d J21a J21b J21c J21d H12uux H12uuy H12uvx H12uvy H12vvx H12vvy */
enum { F_I1u, F_I1v, F_I2u, F_I2v, F_J12a, F_J12b, F_J12c, F_J12d, F_J21a, F_J21b, F_J21c, F_J21d, F_Huux, F_Huuy, F_Huvx, F_Huvy, F_Hvvx, F_Hvvy };

/* PolySolver::getCoefficients, both polynomials (PolySolver.cc:50-149). */
static void poly_coeffs(double a, double b, double c, double d, double t1, double t2, double e1, double e2,
                        double x1, double y1, double x2, double y2, double* q1, double* q2) {
  double detJ12 = a * d - c * b;
  /* first polynomial */
  q1[0] = detJ12 * (t1 * e1 * e2 - detJ12 * (e1 * (c * x2 + d * y2) - y1 * e2));
  q1[1] = -detJ12 * (t2 * e1 * e2 - detJ12 * (e1 * (a * x2 + b * y2) - x1 * e2));
  q1[2] = 0;
  q1[3] = 0;
  q1[4] = t2 * (e1 * e2 * t1 - detJ12 * (x2 * e1 * c + y2 * e1 * d - 2 * e2 * y1)) -
          t1 * detJ12 * (x2 * e1 * a + e1 * b * y2 + 2 * e2 * x1) +
          detJ12 * detJ12 * (e1 * (a * c + b * d) - 2 * (a * x2 * y1 - c * x1 * x2 + b * y1 * y2 - d * x1 * y2));
  q1[5] = (e1 * (-e2 * pow(t2, 2) + 2 * x2 * t2 * a * detJ12 + 2 * y2 * t2 * b * detJ12 - (pow(a, 2) + pow(b, 2)) * detJ12 * detJ12) +
           e2 * detJ12 * detJ12);
  q1[6] = 0;
  q1[7] = t1 * (e2 * detJ12 + 2 * a * x1 * x2 * (detJ12) + 2 * x1 * y2 * b * detJ12) -
          t2 * 2 * (e2 * x1 * t1 + detJ12 * (x2 * y1 * a - c * x1 * x2 + y1 * y2 * b - x1 * y2 * d)) + e2 * y1 * t2 * t2 +
          detJ12 * detJ12 * (-2 * x1 * (a * c + b * d) + y1 * (a * a + b * b) - c * x2 - d * y2);
  q1[8] = t2 * (detJ12 * (e2 - 2 * a * x1 * x2 - 2 * b * x1 * y2)) + x1 * e2 * t2 * t2 +
          (detJ12 * detJ12) * (-y2 * b - x2 * a + x1 * (a * a + b * b));
  q1[9] = t2 * (e2 * t1 - detJ12 * (c * x2 + d * y2)) - t1 * (detJ12 * (a * x2 + b * y2)) + (a * c + b * d) * detJ12 * detJ12;
  /* second polynomial */
  q2[0] = 0;
  q2[1] = 0;
  q2[2] = -detJ12 * (e1 * e2 * t1 - detJ12 * (e1 * (c * x2 + d * y2) - e2 * y1));
  q2[3] = detJ12 * (e1 * e2 * t2 - (detJ12 * (e1 * (a * x2 + b * y2) - e2 * x1)));
  q2[4] = 0;
  q2[5] = e1 * (-e2 * t1 * t1 + (detJ12 * (-(c * c + d * d) * detJ12 + 2 * t1 * c * x2 + 2 * d * y2 * t1))) + e2 * detJ12 * detJ12;
  q2[6] = t2 * (e1 * e2 * t1 - detJ12 * (e1 * c * x2 + e1 * d * y2 + 2 * e2 * y1)) - t1 * detJ12 * (e1 * (a * x2 + b * y2) - 2 * e2 * x1) +
          detJ12 * detJ12 * ((e1 * (a * c + b * d) + 2 * (a * x2 * y1 - c * x1 * x2 + b * y1 * y2 - d * x1 * y2)));
  q2[7] = t1 * detJ12 * (e2 - 2 * c * x2 * y1 - 2 * d * y1 * y2) + y1 * (e2 * t1 * t1 + detJ12 * detJ12 * (c * c + d * d)) -
          detJ12 * detJ12 * (c * x2 + d * y2);
  q2[8] = t2 * (e2 * detJ12 + 2 * y1 * detJ12 * (c * x2 + d * y2)) +
          t1 * (-2 * e2 * y1 * t2 + 2 * detJ12 * (a * x2 * y1 - c * x1 * x2 + b * y1 * y2 - d * x1 * y2)) + e2 * x1 * t1 * t1 -
          2 * detJ12 * detJ12 * (a * c * y1 + 0.5 * a * x2 - 0.5 * c * c * x1 + b * d * y1 + 0.5 * b * y2 - 0.5 * d * d * x1);
  q2[9] = t2 * (e2 * t1 - detJ12 * (c * x2 + d * y2)) - t1 * (detJ12 * (a * x2 + b * y2)) + detJ12 * detJ12 * (a * c + b * d);
}

/* NormalEstimator.cc:78-110: float32 intermediates, then the double-precision coefficient routine */
void nrsfm_oracle_record_coeffs(const float* rec, double* q1, double* q2) {
  float a = rec[F_J12a], b = rec[F_J12b], c = rec[F_J12c], d = rec[F_J12d];
  float t1 = -rec[F_J12b] * rec[F_Hvvx] / 2 + rec[F_J12a] * rec[F_Hvvy] / 2;
  float t2 = -(rec[F_J12d] * rec[F_Hvvx]) / 2 + (rec[F_J12c] * rec[F_Hvvy]) / 2;
  float I1u = rec[F_I1u], I1v = rec[F_I1v], I2u = rec[F_I2u], I2v = rec[F_I2v];
  float e2 = 1 + I2u * I2u + I2v * I2v;
  float e1 = 1 + I1u * I1u + I1v * I1v;
  poly_coeffs(a, b, c, d, t1, t2, e1, e2, I1u, I1v, I2u, I2v, q1, q2);
}

/* PolySolver::Evaluate (PolySolver.cc:152-193): residuals and row-major 2x2 Jacobian of one block */
static void poly_eval(const double* q1, const double* q2, const double* x, double* e, double* J) {
  e[0] = q1[0] * pow(x[0], 3) + q1[1] * pow(x[0], 2) * pow(x[1], 1) + q1[2] * pow(x[0], 1) * pow(x[1], 2) + q1[3] * pow(x[1], 3) +
         q1[4] * pow(x[0], 2) + q1[5] * x[0] * x[1] + q1[6] * pow(x[1], 2) + q1[7] * x[0] + q1[8] * x[1] + q1[9];
  e[1] = q2[0] * pow(x[0], 3) + q2[1] * pow(x[0], 2) * pow(x[1], 1) + q2[2] * pow(x[0], 1) * pow(x[1], 2) + q2[3] * pow(x[1], 3) +
         q2[4] * pow(x[0], 2) + q2[5] * x[0] * x[1] + q2[6] * pow(x[1], 2) + q2[7] * x[0] + q2[8] * x[1] + q2[9];
  if (J) {
    J[0] = 3 * q1[0] * pow(x[0], 2) + 2 * q1[1] * pow(x[0], 1) * pow(x[1], 1) + q1[2] * pow(x[1], 2) + 2 * q1[4] * pow(x[0], 1) + q1[5] * x[1] + q1[7];
    J[1] = q1[1] * pow(x[0], 2) + 2 * q1[2] * pow(x[0], 1) * pow(x[1], 1) + 3 * q1[3] * pow(x[1], 2) + q1[5] * x[0] + 2 * q1[6] * pow(x[1], 1) + q1[8];
    J[2] = 3 * q2[0] * pow(x[0], 2) + 2 * q2[1] * pow(x[0], 1) * pow(x[1], 1) + q2[2] * pow(x[1], 2) + 2 * q2[4] * pow(x[0], 1) + q2[5] * x[1] + q2[7];
    J[3] = q2[1] * pow(x[0], 2) + 2 * q2[2] * pow(x[0], 1) * pow(x[1], 1) + 3 * q2[3] * pow(x[1], 2) + q2[5] * x[0] + 2 * q2[6] * pow(x[1], 1) + q2[8];
  }
}

void nrsfm_oracle_poly_eval(const double* q1, const double* q2, const double* x, double* e, double* J) { poly_eval(q1, q2, x, e, J); }

/* cost = 1/2 |r|^2 over the blocks; optionally g = J^T r (2) and A = J^T J (3: xx, xy, yy) with column scaling s */
static double lsq_eval(int K, const double* Q, const double* x, const double* s, double* g, double* A) {
  double cost = 0.0;
  if (g) { g[0] = g[1] = 0.0; A[0] = A[1] = A[2] = 0.0; }
  for (int k = 0; k < K; k++) {
    double e[2], J[4];
    poly_eval(Q + 20 * k, Q + 20 * k + 10, x, e, g ? J : 0);
    cost += e[0] * e[0] + e[1] * e[1];
    if (g) {
      double j00 = J[0] * s[0], j01 = J[1] * s[1], j10 = J[2] * s[0], j11 = J[3] * s[1];
      g[0] += j00 * e[0] + j10 * e[1];
      g[1] += j01 * e[0] + j11 * e[1];
      A[0] += j00 * j00 + j10 * j10;
      A[1] += j00 * j01 + j10 * j11;
      A[2] += j01 * j01 + j11 * j11;
    }
  }
  return 0.5 * cost;
}

/* The documented Ceres trust-region Levenberg-Marquardt loop on 2 unknowns. Returns the iteration count. */
static int lm_solve2(int K, const double* Q, double* x, int* term) {
  const double ftol = 1e-10, gtol = 1e-8, ptol = 1e-8, min_rel_dec = 1e-3;
  const int max_iter = 200;
  double radius = 1e4, nu = 2.0;
  double s[2] = {1.0, 1.0}, g[2], A[3];
  double cost = lsq_eval(K, Q, x, s, g, A);
  /* Jacobi scaling from the initial Jacobian: 1 / (1 + sqrt(column squared norm)) */
  s[0] = 1.0 / (1.0 + sqrt(A[0]));
  s[1] = 1.0 / (1.0 + sqrt(A[2]));
  cost = lsq_eval(K, Q, x, s, g, A);
  *term = 0;
  if (fmax(fabs(g[0]), fabs(g[1])) <= gtol) { *term = 1; return 0; }
  int it = 0, invalid = 0;
  while (it < max_iter) {
    it++;
    double d0 = fmin(fmax(A[0], 1e-6), 1e32) / radius, d1 = fmin(fmax(A[2], 1e-6), 1e32) / radius;
    double m00 = A[0] + d0, m01 = A[1], m11 = A[2] + d1;
    /* Cholesky of the 2x2 system, step = -(M)^-1 g */
    double l00 = sqrt(m00), l10 = m01 / l00, l11sq = m11 - l10 * l10;
    int ok = (m00 > 0) && (l11sq > 0);
    double dx0 = 0, dx1 = 0, model = 0;
    if (ok) {
      double l11 = sqrt(l11sq);
      double y0 = -g[0] / l00, y1 = (-g[1] - l10 * y0) / l11;
      dx1 = y1 / l11;
      dx0 = (y0 - l10 * dx1) / l00;
      ok = isfinite(dx0) && isfinite(dx1);
      /* model_cost_change = -(J d).(r + J d / 2) = -(d.g + 1/2 d^T A d) */
      model = -(dx0 * g[0] + dx1 * g[1] + 0.5 * (dx0 * (A[0] * dx0 + A[1] * dx1) + dx1 * (A[1] * dx0 + A[2] * dx1)));
      if (!(model > 0)) ok = 0;
    }
    if (!ok) {
      if (++invalid >= 5) { *term = 5; break; }
      radius *= 0.5;
      continue;
    }
    invalid = 0;
    double step[2] = {dx0 * s[0], dx1 * s[1]};
    double xn[2] = {x[0] + step[0], x[1] + step[1]};
    double snorm = sqrt(step[0] * step[0] + step[1] * step[1]), xnorm = sqrt(x[0] * x[0] + x[1] * x[1]);
    if (snorm <= ptol * (xnorm + ptol)) { *term = 3; break; }
    double cost_new = lsq_eval(K, Q, xn, s, 0, 0);
    double rel = (cost - cost_new) / model;
    if (rel > min_rel_dec) {
      double cost_change = cost - cost_new;
      double old_cost = cost;
      x[0] = xn[0]; x[1] = xn[1];
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
      radius = fmin(1e16, radius);
      nu = 2.0;
      cost = lsq_eval(K, Q, x, s, g, A);
      if (fmax(fabs(g[0]), fabs(g[1])) <= gtol) { *term = 1; break; }
      if (fabs(cost_change) <= ftol * old_cost) { *term = 2; break; }
    } else {
      radius = radius / nu;
      nu *= 2.0;
      if (radius < 1e-32) { *term = 4; break; }
    }
  }
  return it;
}

/*
 * One NormalEstimator::ObtainK1K2 pass over P map points that have new observations.
 *   rec_ptr[P+1]      CSR of DiffProp records per point;  rec[R*18] float32 fields (order above)
 *   rec_is_ref[R]     refKF == record.KFToKF.first
 *   rec_first_n[R*2], rec_has_first_n[R]  normal (k1,k2) already stored for the record's first keyframe (non-ref records)
 *   x0[P*2], has_x0[P] previous normal of the reference keyframe (else the start is (0,-0))
 *   ref_uv[P*2]        normalised keypoint of the point in its reference keyframe
 * Outputs: k1k2[P*2], cov[P*4], status[P] (0 solved, 1 no equation, 2 covariance failed), normal_ref[P*3] float,
 *   normal_rec[R*3] float + rec_written[R], iters[P], term[P].
 */
void nrsfm_oracle_normals(int P, const int32_t* rec_ptr, const float* rec, const uint8_t* rec_is_ref,
                          const float* rec_first_n, const uint8_t* rec_has_first_n,
                          const float* x0, const uint8_t* has_x0, const float* ref_uv,
                          double* k1k2, double* cov, int32_t* status, float* normal_ref, float* normal_rec, uint8_t* rec_written,
                          int32_t* iters, int32_t* term) {
  for (int p = 0; p < P; p++) {
    const int r0 = rec_ptr[p], r1 = rec_ptr[p + 1];
    /* one residual block per record whose first keyframe is the reference keyframe -- no upper limit (NormalEstimator.cc:77-118 adds
     * a block per such record, however many there are) */
    double* Q = (double*)malloc(sizeof(double) * 20 * (size_t)(r1 > r0 ? r1 - r0 : 1));
    int K = 0;
    for (int r = r0; r < r1; r++)
      if (rec_is_ref[r]) { nrsfm_oracle_record_coeffs(rec + NREC_F * r, Q + 20 * K, Q + 20 * K + 10); K++; }
    double x[2] = {0.0, -0.0};
    status[p] = 1; iters[p] = 0; term[p] = 0;
    for (int r = r0; r < r1; r++) rec_written[r] = 0;
    if (K > 0) {
      if (has_x0[p]) { x[0] = x0[2 * p]; x[1] = x0[2 * p + 1]; }
      iters[p] = lm_solve2(K, Q, x, &term[p]);
      /* covariance (NormalEstimator.cc:153-159): (J^T J)^-1 unless rank deficient */
      double s1[2] = {1.0, 1.0}, g[2], A[3];
      lsq_eval(K, Q, x, s1, g, A);
      double tr = A[0] + A[2], det = A[0] * A[2] - A[1] * A[1];
      double disc = sqrt(fmax(0.0, 0.25 * tr * tr - det));
      double lmax = 0.5 * tr + disc, lmin = det / lmax;
      k1k2[2 * p] = x[0]; k1k2[2 * p + 1] = x[1];
      if (!(lmax > 0) || !(lmin / lmax >= 1e-14)) { status[p] = 2; free(Q); continue; }
      cov[4 * p] = A[2] / det; cov[4 * p + 1] = -A[1] / det; cov[4 * p + 2] = -A[1] / det; cov[4 * p + 3] = A[0] / det;
      float I1u = ref_uv[2 * p], I1v = ref_uv[2 * p + 1];
      normal_ref[3 * p] = (float)x[0];
      normal_ref[3 * p + 1] = (float)x[1];
      normal_ref[3 * p + 2] = (float)(1 - x[0] * I1u - x[1] * I1v);
      status[p] = 0;
    }
    /* propagation to the other keyframes (NormalEstimator.cc:173-224) */
    for (int r = r0; r < r1; r++) {
      const float* f = rec + NREC_F * r;
      double n0, n1;
      if (rec_is_ref[r]) { n0 = x[0]; n1 = x[1]; }
      else if (rec_has_first_n[r]) { n0 = rec_first_n[2 * r]; n1 = rec_first_n[2 * r + 1]; }
      else continue;
      float j21_11 = f[F_J21a], j21_12 = f[F_J21c], j21_21 = f[F_J21b], j21_22 = f[F_J21d];
      float a = f[F_J12a], b = f[F_J12b], c = f[F_J12c], d = f[F_J12d];
      float detJ12 = a * d - c * b;
      float t1 = -b * f[F_Hvvx] / 2 + a * f[F_Hvvy] / 2;
      float t2 = (d * f[F_Huux]) / 2 - (c * f[F_Huuy]) / 2;   /* H12uu here, H12vv when forming the polynomials: reference quirk */
      double k1 = j21_11 * n0 + j21_12 * n1 + (d * t2 - b * t1) / (detJ12 * detJ12);
      double k2 = j21_21 * n0 + j21_22 * n1 + (a * t1 - c * t2) / (detJ12 * detJ12);
      float I2u = f[F_I2u], I2v = f[F_I2v];
      normal_rec[3 * r] = (float)k1;
      normal_rec[3 * r + 1] = (float)k2;
      normal_rec[3 * r + 2] = (float)(1 - k1 * I2u - k2 * I2v);
      rec_written[r] = 1;
    }
    free(Q);
  }
}
