/*
 * CPU oracle for the surface registration of a new keyframe surface against the map.
 *
 * TEST INFRASTRUCTURE ONLY (parity checker for tests/, smoke(); never linked into the product).
 *
 * Follows, by file:line of /root/reference:
 *   Modules/GroundTruth/GroundTruthCalculator.cc:54-160   scaleMinMedian (float/double mixing kept as written)
 *   Modules/Tracking/DefOptimizer.cc:840-922              Optimizer::OptimizeHorn (two optimize(50) calls, stale-error chi2)
 *   Thirdparty/g2o/g2o/types/sim3.h:71-140,146-148,247-253  Sim3(update) = exp, map, operator*
 *   Thirdparty/g2o/g2o/types/types_seven_dof_expmap.h:96-123,159-188  VertexSim3ExpmapNoProj::oplusImpl, EdgeSim3Simple
 *   Thirdparty/g2o/g2o/core/base_unary_edge.hpp:44-125     numeric Jacobian (delta 1e-9, central), quadratic form
 *   Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189  LM controller (same restatement as sft_oracle.c)
 *   Modules/Mapping/SurfaceRegistration.cc:112-152         composition of the Sim(3) with the keyframe pose, scale recovery
 *
 * Parity status: UNPINNED.  g2o needs Eigen and the call site needs OpenCV; neither is in the image and the reference
 * has no tests or golden vectors for this path.  The reference draws from rand(); the stream of uniform numbers is an
 * INPUT here (u[k] stands for the k-th `(double)rand() / RAND_MAX`), consumed in the reference's order, so a caller
 * that fills it from rand() reproduces the reference's choices.  Two reads past the end of a vector that the
 * reference can make (all residuals unselected) are defined here as "no candidate" (see scale_min_median).
 * Cross-checks in tests/: closed-form Sim(3) recovery on noise-free clouds, the numeric Jacobian against the analytic
 * derivative, exp against scipy, the median selection against numpy.
 */
#include "small_algebra.h"

typedef struct { quat_t r; double t[3]; double s; } sim3_t;

static void mat3_add3(const double* A, double a, const double* B, double b, const double* C, double c, double* O) {
  /* (a*A + b*B) + c*C elementwise, the order Eigen's expression tree evaluates */
  for (int i = 0; i < 9; i++) O[i] = (a * A[i] + b * B[i]) + c * C[i];
}

/* sim3.h:71-140 */
static sim3_t sim3_exp(const double u[7]) {
  const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
  const double sigma = u[6];
  const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double Om2[9], R[9], W[9];
  mat3_mul(Om, Om, Om2);
  sim3_t S;
  S.s = exp(sigma);
  const double eps = 0.00001;
  double A, B, C;
  if (fabs(sigma) < eps) {
    C = 1;
    if (theta < eps) {
      A = 1. / 2.;
      B = 1. / 6.;
      for (int i = 0; i < 9; i++) R[i] = (I3[i] + Om[i]) + Om2[i];
    } else {
      const double theta2 = theta * theta;
      A = (1 - cos(theta)) / (theta2);
      B = (theta - sin(theta)) / (theta2 * theta);
      const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
      for (int i = 0; i < 9; i++) R[i] = (I3[i] + a * Om[i]) + b * Om2[i];
    }
  } else {
    C = (S.s - 1) / sigma;
    if (theta < eps) {
      const double sigma2 = sigma * sigma;
      A = ((sigma - 1) * S.s + 1) / sigma2;
      B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma);
      for (int i = 0; i < 9; i++) R[i] = (I3[i] + Om[i]) + Om2[i];
    } else {
      const double ra = sin(theta) / theta, rb = (1 - cos(theta)) / (theta * theta);
      for (int i = 0; i < 9; i++) R[i] = (I3[i] + ra * Om[i]) + rb * Om2[i];
      const double a = S.s * sin(theta);
      const double b = S.s * cos(theta);
      const double theta2 = theta * theta;
      const double sigma2 = sigma * sigma;
      const double c = theta2 + sigma2;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
    }
  }
  S.r = quat_from_R(R);
  mat3_add3(Om, A, Om2, B, I3, C, W);
  for (int i = 0; i < 3; i++) S.t[i] = (W[3 * i] * up[0] + W[3 * i + 1] * up[1]) + W[3 * i + 2] * up[2];
  return S;
}

/* sim3.h:146-148 */
static void sim3_map(const sim3_t* S, const double p[3], double o[3]) {
  double rp[3];
  quat_rot(&S->r, p, rp);
  for (int k = 0; k < 3; k++) o[k] = S->s * rp[k] + S->t[k];
}

/* sim3.h:247-253 (no renormalisation of the quaternion) */
static sim3_t sim3_mul(const sim3_t* a, const sim3_t* b) {
  sim3_t r;
  double rt[3];
  r.r = quat_mul(&a->r, &b->r);
  quat_rot(&a->r, b->t, rt);
  for (int k = 0; k < 3; k++) r.t[k] = a->s * rt[k] + a->t[k];
  r.s = a->s * b->s;
  return r;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* GroundTruthCalculator.cc:54-160.  mono/stereo: n*3 float.  u: uniform stream, nu entries.                      */
/* Returns the scale; *consumed = draws used, *status = 0 ok, 1 stream too short, 2 "return 0.0" of the reference */
/* (a candidate whose selected set holds fewer than two residuals; the reference reads past the vector when it    */
/* holds none: defined here as the same early return).                                                            */
/* ------------------------------------------------------------------------------------------------------------- */
static int cmp_float(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

float horn_oracle_scale_min_median(int n, const float* mono, const float* stereo, const double* u, int nu,
                                   int32_t* consumed, int32_t* status, float* medians /* n, -1 where not a candidate, may be NULL */) {
  float min_med = 10000.0;
  int final_points = 0;
  double best_scale = 0.0;
  int k = 0;
  float* res = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  *status = 0;
  if (medians) for (int i = 0; i < n; i++) medians[i] = -1.f;
  for (int i = 0; i < n; i++) {
    if (k >= nu) { *status = 1; break; }
    const double r_i = u[k++];
    if (r_i > 0.25) continue;
    const double scale = stereo[3 * i + 2] / mono[3 * i + 2]; /* float division, widened */
    int m = 0;
    for (int j = 0; j < n; j++) {
      res[j] = -1;
      if (i == j) continue;
      if (k >= nu) { *status = 1; break; }
      const double r_j = u[k++];
      if (r_j > 0.25) continue;
      float r2 = 0.0;
      for (int c = 0; c < 3; c++) {
        const double d = (scale * mono[3 * j + c] - stereo[3 * j + c]);
        r2 = r2 + d * d;
      }
      res[j] = sqrtf(r2);
      m++;
    }
    if (*status) break;
    qsort(res, n, sizeof(float), cmp_float);
    /* `while (squared_res[NumberNonZero++] < 0)`: the copy starts ONE PAST the first non-negative entry */
    final_points++;
    if (m <= 1) { *status = 2; *consumed = k; free(res); return 0.0f; }
    const int first = n - m;         /* index of the first non-negative entry */
    const int size = m - 1;
    const int median_index = size / 2;
    const float med = res[first + 1 + median_index];
    if (medians) medians[i] = med;
    if (med < min_med) { min_med = med; best_scale = scale; }
  }
  *consumed = k;
  free(res);
  if (*status) return 0.0f;
  const float desv = 1.4826 * (1.0 - (5.0 / (final_points - 1.0))) * sqrtf(min_med);
  float sum_num = 0.0, sum_den = 0.0;
  for (int i = 0; i < n; i++) {
    float residual = 0.0;
    for (int c = 0; c < 3; c++) {
      const double d = (best_scale * mono[3 * i + c] - stereo[3 * i + c]);
      residual = residual + d * d;
    }
    residual = sqrtf(residual);
    if ((residual / desv) < 2.5) {
      sum_num += (stereo[3 * i + 2] * mono[3 * i + 2]);
      sum_den += (mono[3 * i + 2] * mono[3 * i + 2]);
    }
  }
  best_scale = sum_num / sum_den;
  return (float)best_scale;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* OptimizeHorn, DefOptimizer.cc:840-922                                                                          */
/* ------------------------------------------------------------------------------------------------------------- */
typedef struct {
  int n;
  const float *p1, *p2;
  sim3_t est;
  double delta, dsqr;
  double* err; /* n*3: the edges' _error as last computed */
  int tree;    /* 0: sums in edge order (the reference); 1: the fixed tree of the device kernel (see horn_sum) */
  double* term; /* n*35 scratch */
} horn_t;

/* Sum of n terms (stride apart).  tree == 0: in edge order, like g2o.  tree == 1: the order the 256-thread device kernel
 * uses (register_kernels.hip block_sum256): thread t adds the terms t, t+256, ... in order, 64 lanes are combined by an
 * xor butterfly (32, 16, ... 1), the four wavefronts as (w0 + w1) + (w2 + w3).  Only used to show that, given the same
 * summation order, the device follows the same Levenberg-Marquardt trajectory decision for decision. */
static double horn_sum(const double* v, int n, int stride, int tree, int negate) {
  if (!tree) {
    double s = 0.0;
    for (int i = 0; i < n; i++) { if (negate) s -= v[(size_t)i * stride]; else s += v[(size_t)i * stride]; }
    return s;
  }
  double part[256];
  for (int t = 0; t < 256; t++) {
    double s = 0.0;
    for (int i = t; i < n; i += 256) { if (negate) s -= v[(size_t)i * stride]; else s += v[(size_t)i * stride]; }
    part[t] = s;
  }
  double w[4];
  for (int q = 0; q < 4; q++) {
    double a[64], b[64];
    for (int l = 0; l < 64; l++) a[l] = part[64 * q + l];
    for (int m = 32; m >= 1; m >>= 1) {
      for (int l = 0; l < 64; l++) b[l] = a[l] + a[l ^ m];
      for (int l = 0; l < 64; l++) a[l] = b[l];
    }
    w[q] = a[0];
  }
  return (w[0] + w[1]) + (w[2] + w[3]);
}

static void horn_errors(horn_t* g) {
  for (int i = 0; i < g->n; i++) {
    const double a[3] = {g->p1[3 * i], g->p1[3 * i + 1], g->p1[3 * i + 2]};
    double m[3];
    sim3_map(&g->est, a, m);
    for (int k = 0; k < 3; k++) g->err[3 * i + k] = (double)g->p2[3 * i + k] - m[k];
  }
}
static double horn_edge_chi2(const horn_t* g, int i) {
  const double* e = g->err + 3 * i;
  return (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];
}
static void horn_huber(const horn_t* g, double e2, double rho[2]) {
  if (e2 <= g->dsqr) { rho[0] = e2; rho[1] = 1.; }
  else { const double sq = sqrt(e2); rho[0] = 2 * sq * g->delta - g->dsqr; rho[1] = g->delta / sq; }
}
static double horn_robust_chi2(const horn_t* g) {
  double rho[2];
  for (int i = 0; i < g->n; i++) { horn_huber(g, horn_edge_chi2(g, i), rho); g->term[i] = rho[0]; }
  return horn_sum(g->term, g->n, 1, g->tree, 0);
}
/* base_unary_edge.hpp:81-125 + 44-74; H column-major 7x7 (full), b 7 */
static void horn_build(horn_t* g, double* H, double* b) {
  const double delta = 1e-9, scalar = 1.0 / (2 * delta);
  sim3_t plus[7], minus[7];
  for (int d = 0; d < 7; d++) {
    double add[7] = {0, 0, 0, 0, 0, 0, 0};
    add[d] = delta;
    sim3_t up = sim3_exp(add);
    plus[d] = sim3_mul(&up, &g->est);
    add[d] = -delta;
    up = sim3_exp(add);
    minus[d] = sim3_mul(&up, &g->est);
  }
  for (int i = 0; i < g->n; i++) {
    const double a[3] = {g->p1[3 * i], g->p1[3 * i + 1], g->p1[3 * i + 2]};
    const double z[3] = {g->p2[3 * i], g->p2[3 * i + 1], g->p2[3 * i + 2]};
    double J[3][7];
    for (int d = 0; d < 7; d++) {
      double mp[3], mm[3];
      sim3_map(&plus[d], a, mp);
      sim3_map(&minus[d], a, mm);
      for (int k = 0; k < 3; k++) J[k][d] = scalar * ((z[k] - mp[k]) - (z[k] - mm[k]));
    }
    const double* e = g->err + 3 * i;
    double rho[2];
    horn_huber(g, horn_edge_chi2(g, i), rho);
    double* tm = g->term + (size_t)35 * i;
    int q = 0;
    for (int c = 0; c < 7; c++)
      for (int r = c; r < 7; r++) tm[q++] = ((J[0][r] * rho[1]) * J[0][c] + (J[1][r] * rho[1]) * J[1][c]) + (J[2][r] * rho[1]) * J[2][c];
    for (int r = 0; r < 7; r++) tm[28 + r] = ((rho[1] * J[0][r]) * e[0] + (rho[1] * J[1][r]) * e[1]) + (rho[1] * J[2][r]) * e[2];
  }
  /* from->A() += A^T W A, from->b() -= rho1 A^T Omega e, edge by edge; only the lower triangle is read by the LDLT */
  int q = 0;
  for (int c = 0; c < 7; c++)
    for (int r = c; r < 7; r++) { H[r + 7 * c] = horn_sum(g->term + q, g->n, 35, g->tree, 0); H[c + 7 * r] = H[r + 7 * c]; q++; }
  for (int r = 0; r < 7; r++) b[r] = horn_sum(g->term + 28 + r, g->n, 35, g->tree, 1);
}

static void horn_lm(horn_t* g, int max_iters, int32_t* iters, int32_t* trials) {
  double H[49], Hs[49], b[7], x[7] = {0, 0, 0, 0, 0, 0, 0}, tmp[7 * 48];
  int perm[7];
  double lambda = -1., ni = 2.;
  int nBad = 0, it_count = 0, total_trials = 0;
  const double tau = 1e-5, goodUp = 2. / 3., goodLo = 1. / 3.;
  const int maxTrials = 10;
  for (int it = 0; it < max_iters && g->n > 0; it++) {
    horn_errors(g);
    double currentChi = horn_robust_chi2(g), tempChi = currentChi;
    const double iniChi = currentChi;
    horn_build(g, H, b);
    if (it == 0) {
      double maxDiag = 0.;
      for (int j = 0; j < 7; j++) { const double v = fabs(H[j + 7 * j]); if (v > maxDiag) maxDiag = v; }
      lambda = tau * maxDiag; ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const sim3_t bak = g->est;
      memcpy(Hs, H, sizeof(H));
      for (int j = 0; j < 7; j++) Hs[j + 7 * j] += lambda;
      const int ok = ldlt_pivoted(Hs, 7, perm, tmp);
      if (ok) ldlt_pivoted_solve(Hs, 7, perm, b, x);
      { const sim3_t up = sim3_exp(x); g->est = sim3_mul(&up, &g->est); }
      horn_errors(g);
      tempChi = horn_robust_chi2(g);
      if (!ok) tempChi = DBL_MAX;
      rho = (currentChi - tempChi);
      double scale = 0.;
      for (int j = 0; j < 7; j++) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = alpha < goodUp ? alpha : goodUp;
        const double sf = goodLo > alpha ? goodLo : alpha;
        lambda *= sf; ni = 2; currentChi = tempChi;
      } else {
        lambda *= ni; ni *= 2;
        g->est = bak;
      }
      qmax++;
    } while (rho < 0 && qmax < maxTrials);
    total_trials += qmax;
    it_count++;
    if (qmax == maxTrials || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) break;
  }
  *iters = it_count;
  *trials = total_trials;
}

/* sim3: qx qy qz qw tx ty tz s (in: initial, out: estimate).  Returns `aceptable`. */
int horn_oracle_optimize(int n, const float* pts1, const float* pts2, double* sim3, double chi, double huber, int sum_mode,
                         double* chi2_final, int32_t* count_out, int32_t* iters /* 2 */, int32_t* trials /* 2 */) {
  horn_t g;
  g.n = n; g.p1 = pts1; g.p2 = pts2; g.tree = sum_mode;
  g.term = (double*)calloc((size_t)(n > 0 ? n : 1) * 35, sizeof(double));
  g.est.r.x = sim3[0]; g.est.r.y = sim3[1]; g.est.r.z = sim3[2]; g.est.r.w = sim3[3];
  g.est.t[0] = sim3[4]; g.est.t[1] = sim3[5]; g.est.t[2] = sim3[6]; g.est.s = sim3[7];
  const float deltaHuber = sqrt(huber);
  g.delta = deltaHuber; g.dsqr = g.delta * g.delta;
  g.err = (double*)calloc((size_t)(n > 0 ? n : 1) * 3, sizeof(double));
  horn_lm(&g, 50, &iters[0], &trials[0]);
  const sim3_t first = g.est;
  int count = 0;
  for (int i = 0; i < n; i++) if (!(horn_edge_chi2(&g, i) > chi)) count++;
  horn_lm(&g, 50, &iters[1], &trials[1]);
  for (int i = 0; i < n; i++) g.term[i] = horn_edge_chi2(&g, i);   /* OptimizableGraph::chi2(): plain, stale errors */
  const double total = horn_sum(g.term, n, 1, g.tree, 0);
  /* g2oS12 = vert0->estimate() is read BEFORE the second optimize (DefOptimizer.cc:896) */
  sim3[0] = first.r.x; sim3[1] = first.r.y; sim3[2] = first.r.z; sim3[3] = first.r.w;
  sim3[4] = first.t[0]; sim3[5] = first.t[1]; sim3[6] = first.t[2]; sim3[7] = first.s;
  if (chi2_final) *chi2_final = total;
  if (count_out) *count_out = count;
  free(g.err); free(g.term);
  if (isnan(total) || isinf(total)) return 0;
  return (total / count < chi);
}

/* Unit-test hooks */
void horn_oracle_sim3_exp(const double u[7], double out[8]) {
  const sim3_t S = sim3_exp(u);
  out[0] = S.r.x; out[1] = S.r.y; out[2] = S.r.z; out[3] = S.r.w; out[4] = S.t[0]; out[5] = S.t[1]; out[6] = S.t[2]; out[7] = S.s;
}
void horn_oracle_system(int n, const float* pts1, const float* pts2, const double* sim3, double huber, double* H, double* b, double* chi) {
  horn_t g;
  g.n = n; g.p1 = pts1; g.p2 = pts2; g.tree = 0;
  g.term = (double*)calloc((size_t)(n > 0 ? n : 1) * 35, sizeof(double));
  g.est.r.x = sim3[0]; g.est.r.y = sim3[1]; g.est.r.z = sim3[2]; g.est.r.w = sim3[3];
  g.est.t[0] = sim3[4]; g.est.t[1] = sim3[5]; g.est.t[2] = sim3[6]; g.est.s = sim3[7];
  const float deltaHuber = sqrt(huber);
  g.delta = deltaHuber; g.dsqr = g.delta * g.delta;
  g.err = (double*)calloc((size_t)(n > 0 ? n : 1) * 3, sizeof(double));
  horn_errors(&g);
  *chi = horn_robust_chi2(&g);
  horn_build(&g, H, b);
  free(g.err); free(g.term);
}

/* ------------------------------------------------------------------------------------------------------------- */
/* SurfaceRegistration.cc:112-152: compose with the keyframe pose, recover the scale, new Tcw (float32, row-major) */
/* ------------------------------------------------------------------------------------------------------------- */
void horn_oracle_compose(const double* sim3, const float* Twc, double* s22_out, float* Tcw_out) {
  quat_t q = {sim3[0], sim3[1], sim3[2], sim3[3]};
  double R[9];
  quat_to_R(&q, R);
  float S[16], T[16];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) S[4 * i + j] = (float)(sim3[7] * R[3 * i + j]);   /* Converter::toCvMat(Sim3): s*R, t */
    S[4 * i + 3] = (float)sim3[4 + i];
  }
  S[12] = S[13] = S[14] = 0.f; S[15] = 1.f;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float a = 0.f;
      for (int k = 0; k < 4; k++) a += S[4 * i + k] * Twc[4 * k + j];
      T[4 * i + j] = a;
    }
  float tt = 0.f;
  for (int k = 0; k < 3; k++) tt += T[k] * T[k];   /* (R R^T)(0,0) */
  const double s22 = sqrt((double)tt);
  *s22_out = s22;
  float Rn[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rn[3 * i + j] = T[4 * i + j] / (float)s22;
  /* inverse of [Rn | t] with Rn orthonormal up to rounding: the reference uses Eigen's general 4x4 inverse (float);
     restated as the cofactor inverse of the 3x3 block, which is what a general inverse reduces to for this pattern */
  const float a = Rn[0], b = Rn[1], c = Rn[2], d = Rn[3], e = Rn[4], f = Rn[5], g = Rn[6], h = Rn[7], i2 = Rn[8];
  const float det = a * (e * i2 - f * h) - b * (d * i2 - f * g) + c * (d * h - e * g);
  const float inv[9] = {(e * i2 - f * h) / det, (c * h - b * i2) / det, (b * f - c * e) / det,
                        (f * g - d * i2) / det, (a * i2 - c * g) / det, (c * d - a * f) / det,
                        (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
  for (int r = 0; r < 3; r++) {
    for (int cc = 0; cc < 3; cc++) Tcw_out[4 * r + cc] = inv[3 * r + cc];
    Tcw_out[4 * r + 3] = -(inv[3 * r] * T[3] + inv[3 * r + 1] * T[7] + inv[3 * r + 2] * T[11]);
  }
  Tcw_out[12] = Tcw_out[13] = Tcw_out[14] = 0.f; Tcw_out[15] = 1.f;
}
