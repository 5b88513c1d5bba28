/*
 * schwarp_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never on the product path).
 *
 * Restatement of the Schwarzian-regularised B-spline warp fit between two keyframes
 * (SURVEY.md section 8a rows B1a-B1c):
 *   reprojection block ...... Warps::Warp ctor + Evaluate   Modules/Mapping/Schwarp.cc:38-97,235-303
 *   Schwarzian block ......... Warps::Schwarzian ctor + Evaluate   Schwarp.cc:305-543
 *   solve + DiffProp ......... SchwarpDatabase::calculateSchwarps   Modules/Mapping/SchwarpDatabase.cc:145-349
 *
 * Reference quirks kept on purpose:
 *   - the caller passes (fy, fx) into the (fx, fy) slots of Warp (SchwarpDatabase.cc:200-201) -> here `fxs`, `fys`
 *     are simply "the value in the fx slot / fy slot";
 *   - Warp's Jacobian is the constant -[coloc*fxs, 0; 0, coloc*fys] WITHOUT the invSigma factor, and the copy loop of
 *     Evaluate overwrites the y rows with the x rows (Schwarp.cc:291-298): rows i and i+P are both [-coloc_i*fxs, 0];
 *   - the warp block is ONE residual block, so HuberLoss(5.77) weighs the sum of all its squared residuals.
 * Parameter layout: x[0..N) first coordinate of the N = nptsu*nptsv control points, x[N..2N) second coordinate
 * (the Eigen::Map in Schwarp.cc:240-244 is column-major; index l = iu*nptsv + iv).
 *
 * PARITY UNPINNED: the minimiser is Ceres (un-vendored, unversioned).  The loop below is Ceres' documented
 * trust-region Levenberg-Marquardt (see oracle/nrsfm_oracle.c) with the options of SchwarpDatabase.cc:211-218:
 * max_num_iterations 3, default tolerances (function 1e-6, gradient 1e-10, parameter 1e-8), Jacobi scaling,
 * normal equations solved by Cholesky (SPARSE_NORMAL_CHOLESKY solves the same system).  The B-spline pieces it
 * is built from are pinned to the reference's bbs.cc (oracle/bbs_oracle.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void bbs_oracle_eval(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, int valdim, const double* ctrl, const double* u,
                     const double* v, int n, int du, int dv, double* val, uint8_t* status);
int bbs_oracle_coloc(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, const double* u, const double* v, int n, int du, int dv,
                     int32_t* cols, double* w);

typedef struct {
  double umin, umax, vmin, vmax;
  int nu, nv, N, P;
  const float *kp1, *kp2, *invsig;
  double fxs, fys, lambda;
} swp_t;

/* Array[valdim*l + n] = ControlPoints(l, n) (Schwarp.cc:253-261) */
static void interleave(const swp_t* s, const double* x, double* arr) {
  for (int l = 0; l < s->N; l++) { arr[2 * l] = x[l]; arr[2 * l + 1] = x[s->N + l]; }
}

static void grid_sites(const swp_t* s, double* X, double* Y) { /* Schwarp.cc:322-331 */
  int us = 0;
  for (int i = 0; i < s->nu; i++)
    for (int j = 0; j < s->nv; j++) {
      X[us] = (double)((s->umax - s->umin) * i) / (s->nu - 1) + s->umin;
      Y[us] = (double)((s->vmax - s->vmin) * j) / (s->nv - 1) + s->vmin;
      us++;
    }
}

/* residuals r[2P + 4N]; J (row-major, (2P+4N) x 2N) may be NULL. Raw (no loss correction). */
static void swp_eval(const swp_t* s, const double* x, double* r, double* J) {
  const int N = s->N, P = s->P, n2 = 2 * N;
  double* arr = (double*)malloc(sizeof(double) * n2);
  interleave(s, x, arr);
  /* ---- reprojection block */
  double* u = (double*)malloc(sizeof(double) * (P > N ? P : N));
  double* v = (double*)malloc(sizeof(double) * (P > N ? P : N));
  double* val = (double*)malloc(sizeof(double) * 2 * (P > N ? P : N));
  for (int i = 0; i < P; i++) { u[i] = s->kp1[2 * i]; v[i] = s->kp1[2 * i + 1]; }
  bbs_oracle_eval(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, 2, arr, u, v, P, 0, 0, val, 0);
  for (int i = 0; i < P; i++) {
    r[i] = s->invsig[i] * ((double)s->kp2[2 * i] - val[2 * i]) * s->fxs;
    r[i + P] = s->invsig[i] * ((double)s->kp2[2 * i + 1] - val[2 * i + 1]) * s->fys;
  }
  if (J) {
    memset(J, 0, sizeof(double) * (size_t)(2 * P + 4 * N) * n2);
    int32_t* cols = (int32_t*)malloc(sizeof(int32_t) * 16 * P);
    double* w = (double*)malloc(sizeof(double) * 16 * P);
    bbs_oracle_coloc(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, u, v, P, 0, 0, cols, w);
    for (int i = 0; i < P; i++)
      for (int t = 0; t < 16; t++) {
        if (cols[16 * i + t] < 0) continue;
        const double jv = -w[16 * i + t] * s->fxs;
        J[(size_t)i * n2 + cols[16 * i + t]] = jv;           /* x row */
        J[(size_t)(i + P) * n2 + cols[16 * i + t]] = jv;     /* y row: overwritten with the x row (quirk) */
      }
    free(cols); free(w);
  }
  /* ---- Schwarzian block at the grid sites */
  double* X = u; double* Y = v;
  grid_sites(s, X, Y);
  double *d10 = (double*)malloc(sizeof(double) * 2 * N), *d01 = (double*)malloc(sizeof(double) * 2 * N), *d20 = (double*)malloc(sizeof(double) * 2 * N),
         *d02 = (double*)malloc(sizeof(double) * 2 * N), *d11 = (double*)malloc(sizeof(double) * 2 * N);
  bbs_oracle_eval(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, 2, arr, X, Y, N, 1, 0, d10, 0);
  bbs_oracle_eval(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, 2, arr, X, Y, N, 0, 1, d01, 0);
  bbs_oracle_eval(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, 2, arr, X, Y, N, 2, 0, d20, 0);
  bbs_oracle_eval(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, 2, arr, X, Y, N, 0, 2, d02, 0);
  bbs_oracle_eval(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, 2, arr, X, Y, N, 1, 1, d11, 0);
  double* rs = r + 2 * P;
  const double lam = s->lambda;
  for (int k = 0; k < N; k++) {
    const double xu = d10[2 * k], yu = d10[2 * k + 1], xv = d01[2 * k], yv = d01[2 * k + 1];
    const double xuu = d20[2 * k], yuu = d20[2 * k + 1], xvv = d02[2 * k], yvv = d02[2 * k + 1], xuv = d11[2 * k], yuv = d11[2 * k + 1];
    rs[k] = ((xuu * yu - yuu * xu)) * lam;
    rs[N + k] = ((yvv * xv - xvv * yv)) * lam;
    rs[2 * N + k] = ((xuu * yv - yuu * xv + 2 * (xuv * yu - yuv * xu))) * lam;
    rs[3 * N + k] = ((yvv * xu - xvv * yu + 2 * (yuv * xv - xuv * yv))) * lam;
  }
  if (J) {
    int32_t* c = (int32_t*)malloc(sizeof(int32_t) * 16 * N);
    double *wu = (double*)malloc(sizeof(double) * 16 * N), *wv = (double*)malloc(sizeof(double) * 16 * N), *wuu = (double*)malloc(sizeof(double) * 16 * N),
           *wvv = (double*)malloc(sizeof(double) * 16 * N), *wuv = (double*)malloc(sizeof(double) * 16 * N);
    bbs_oracle_coloc(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, X, Y, N, 1, 0, c, wu);
    bbs_oracle_coloc(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, X, Y, N, 0, 1, c, wv);
    bbs_oracle_coloc(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, X, Y, N, 2, 0, c, wuu);
    bbs_oracle_coloc(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, X, Y, N, 0, 2, c, wvv);
    bbs_oracle_coloc(s->umin, s->umax, s->nu, s->vmin, s->vmax, s->nv, X, Y, N, 1, 1, c, wuv);
    double* Js = J + (size_t)2 * P * n2;
    for (int k = 0; k < N; k++) {
      const double xu = d10[2 * k], yu = d10[2 * k + 1], xv = d01[2 * k], yv = d01[2 * k + 1];
      const double xuu = d20[2 * k], yuu = d20[2 * k + 1], xvv = d02[2 * k], yvv = d02[2 * k + 1], xuv = d11[2 * k], yuv = d11[2 * k + 1];
      for (int t = 0; t < 16; t++) {
        const int col = c[16 * k + t];
        if (col < 0) continue;
        const double Cu = wu[16 * k + t], Cv = wv[16 * k + t], Cuu = wuu[16 * k + t], Cvv = wvv[16 * k + t], Cuv = wuv[16 * k + t];
        /* Schwarp.cc:497-518 (jI, jJ, jM, jN), x parameters then y parameters */
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
    free(c); free(wu); free(wv); free(wuu); free(wvv); free(wuv);
  }
  free(arr); free(u); free(v); free(val); free(d10); free(d01); free(d20); free(d02); free(d11);
}

/* Raw evaluation for tests */
void schwarp_oracle_eval(double umin, double umax, int nu, double vmin, double vmax, int nv, int P, const float* kp1, const float* kp2,
                         const float* invsig, double fxs, double fys, double lambda, const double* x, double* r, double* J) {
  swp_t s = {umin, umax, vmin, vmax, nu, nv, nu * nv, P, kp1, kp2, invsig, fxs, fys, lambda};
  swp_eval(&s, x, r, J);
}

#define HUBER_A 5.77 /* SchwarpDatabase.cc:208 */

/* ceres::Problem::Evaluate(EvaluateOptions(), &cost, &residuals, ...) of the problem CalculateInitialSchwarp builds
 * (DefORBmatcher.cc:155-166): ONE residual block (Warps::Warp, 2P residuals) under HuberLoss(5.77), apply_loss_function = true
 * (the default), so the residuals come back loss-corrected.  Ceres' Corrector (corrector.cc): for rho'' <= 0 -- Huber beyond its
 * threshold has rho'' = -a / (2 s^1.5) -- the block is scaled by sqrt(rho'(s)), s = |r|^2; rho' = 1 up to s = a^2, a / sqrt(s) beyond.
 * out[0 .. 2P): the corrected residuals; returns the cost 1/2 rho(s). */
double schwarp_oracle_eval_initial(double umin, double umax, int nu, double vmin, double vmax, int nv, int P, const float* kp1, const float* kp2,
                                   const float* invsig, double fxs, double fys, const double* x, double* out) {
  swp_t s = {umin, umax, vmin, vmax, nu, nv, nu * nv, P, kp1, kp2, invsig, fxs, fys, 0.0};
  double* r = (double*)malloc(sizeof(double) * (2 * (size_t)P + 4 * (size_t)s.N));
  swp_eval(&s, x, r, 0);
  double sq = 0.0;
  for (int i = 0; i < 2 * P; i++) sq += r[i] * r[i];
  double rho0 = sq, rho1 = 1.0;
  if (sq > HUBER_A * HUBER_A) { const double rt = sqrt(sq); rho0 = 2 * HUBER_A * rt - HUBER_A * HUBER_A; rho1 = HUBER_A / rt; }
  const double sc = sqrt(rho1);
  for (int i = 0; i < 2 * P; i++) out[i] = sc * r[i];
  free(r);
  return 0.5 * rho0;
}

/* cost = 1/2 (rho(|r_warp|^2) + |r_schw|^2); optionally the loss-corrected, column-scaled normal equations */
static double swp_normal(const swp_t* s, const double* x, const double* cs, double* r, double* J, double* A, double* g) {
  const int n2 = 2 * s->N, m = 2 * s->P + 4 * s->N, P2 = 2 * s->P;
  swp_eval(s, x, r, A ? J : 0);
  double sq = 0.0;
  for (int i = 0; i < P2; i++) sq += r[i] * r[i];
  double rho0 = sq, rho1 = 1.0;
  if (sq > HUBER_A * HUBER_A) { const double rt = sqrt(sq); rho0 = 2 * HUBER_A * rt - HUBER_A * HUBER_A; rho1 = HUBER_A / rt; }
  double cost = rho0;
  for (int i = P2; i < m; i++) cost += r[i] * r[i];
  cost *= 0.5;
  if (A) {
    const double sc = sqrt(rho1);   /* Ceres Corrector with rho'' <= 0: residuals and Jacobian scaled by sqrt(rho') */
    for (int i = 0; i < P2; i++) {
      r[i] *= sc;
      for (int j = 0; j < n2; j++) J[(size_t)i * n2 + j] *= sc;
    }
    memset(A, 0, sizeof(double) * (size_t)n2 * n2);
    memset(g, 0, sizeof(double) * n2);
    for (int i = 0; i < m; i++) {
      const double* row = J + (size_t)i * n2;
      for (int a = 0; a < n2; a++) {
        const double ja = row[a];
        if (ja == 0.0) continue;
        const double jas = ja * cs[a];
        g[a] += jas * r[i];
        for (int b = 0; b <= a; b++) {
          const double jb = row[b];
          if (jb != 0.0) A[(size_t)a * n2 + b] += jas * (jb * cs[b]);
        }
      }
    }
    for (int a = 0; a < n2; a++)
      for (int b = a + 1; b < n2; b++) A[(size_t)a * n2 + b] = A[(size_t)b * n2 + a];
  }
  return cost;
}

static int chol_solve(double* M, int n, const double* b, double* x) { /* M lower, in place */
  for (int k = 0; k < n; k++) {
    double d = M[(size_t)k * n + k];
    for (int j = 0; j < k; j++) d -= M[(size_t)k * n + j] * M[(size_t)k * n + j];
    if (!(d > 0)) return 0;
    d = sqrt(d);
    M[(size_t)k * n + k] = d;
    for (int r = k + 1; r < n; r++) {
      double v = M[(size_t)r * n + k];
      for (int j = 0; j < k; j++) v -= M[(size_t)r * n + j] * M[(size_t)k * n + j];
      M[(size_t)r * n + k] = v / d;
    }
  }
  for (int i = 0; i < n; i++) { double v = b[i]; for (int j = 0; j < i; j++) v -= M[(size_t)i * n + j] * x[j]; x[i] = v / M[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int j = i + 1; j < n; j++) v -= M[(size_t)j * n + i] * x[j]; x[i] = v / M[(size_t)i * n + i]; }
  return 1;
}

/*
 * SchwarpDatabase::calculateSchwarps: fit (x is in/out, 2N doubles) and DiffProp extraction.
 * diff[P*18] float32 in the DiffProp field order of nrsfm_oracle.c; drop[P] = 1 when the reprojection error of the
 * match exceeds 10 px (SchwarpDatabase.cc:283-293; fx_true/fy_true are KF->fx, KF->fy). info[0] = LM iterations,
 * info[1] = successful steps; costs[0] initial, costs[1] final.
 */
void schwarp_oracle_fit(double umin, double umax, int nu, double vmin, double vmax, int nv, int P, const float* kp1, const float* kp2,
                        const float* invsig, double fxs, double fys, double lambda, float fx_true, float fy_true, int max_iters,
                        double* x, float* diff, uint8_t* drop, int32_t* info, double* costs) {
  swp_t s = {umin, umax, vmin, vmax, nu, nv, nu * nv, P, kp1, kp2, invsig, fxs, fys, lambda};
  const int N = s.N, n2 = 2 * N, m = 2 * P + 4 * N;
  double* r = (double*)malloc(sizeof(double) * m);
  double* J = (double*)malloc(sizeof(double) * (size_t)m * n2);
  double* A = (double*)malloc(sizeof(double) * (size_t)n2 * n2);
  double* M = (double*)malloc(sizeof(double) * (size_t)n2 * n2);
  double *g = (double*)malloc(sizeof(double) * n2), *cs = (double*)malloc(sizeof(double) * n2), *dx = (double*)malloc(sizeof(double) * n2),
         *xn = (double*)malloc(sizeof(double) * n2), *rhs = (double*)malloc(sizeof(double) * n2);
  for (int j = 0; j < n2; j++) cs[j] = 1.0;
  double cost = swp_normal(&s, x, cs, r, J, A, g);
  for (int j = 0; j < n2; j++) cs[j] = 1.0 / (1.0 + sqrt(A[(size_t)j * n2 + j]));
  cost = swp_normal(&s, x, cs, r, J, A, g);
  costs[0] = cost;
  const double ftol = 1e-6, gtol = 1e-10, ptol = 1e-8, min_rel_dec = 1e-3;
  double radius = 1e4, nu_f = 2.0;
  int it = 0, good = 0, invalid = 0;
  double gmax = 0;
  for (int j = 0; j < n2; j++) gmax = fmax(gmax, fabs(g[j]));
  if (gmax > gtol)
    while (it < max_iters) {
      it++;
      memcpy(M, A, sizeof(double) * (size_t)n2 * n2);
      for (int j = 0; j < n2; j++) { M[(size_t)j * n2 + j] += fmin(fmax(A[(size_t)j * n2 + j], 1e-6), 1e32) / radius; rhs[j] = -g[j]; }
      int ok = chol_solve(M, n2, rhs, dx);
      double model = 0;
      if (ok) {
        double dg = 0, q = 0;
        for (int a = 0; a < n2; a++) {
          dg += dx[a] * g[a];
          double t = 0;
          for (int b = 0; b < n2; b++) t += A[(size_t)a * n2 + b] * dx[b];
          q += dx[a] * t;
        }
        model = -(dg + 0.5 * q);
        if (!(model > 0)) ok = 0;
      }
      if (!ok) { if (++invalid >= 5) break; radius *= 0.5; continue; }
      invalid = 0;
      double sn = 0, xnrm = 0;
      for (int j = 0; j < n2; j++) { const double st = dx[j] * cs[j]; xn[j] = x[j] + st; sn += st * st; xnrm += x[j] * x[j]; }
      if (sqrt(sn) <= ptol * (sqrt(xnrm) + ptol)) break;
      const double cost_new = swp_normal(&s, xn, cs, r, 0, 0, 0);
      const double rel = (cost - cost_new) / model;
      if (rel > min_rel_dec) {
        const double change = cost - cost_new, old = cost;
        memcpy(x, xn, sizeof(double) * n2);
        radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
        nu_f = 2.0;
        good++;
        cost = swp_normal(&s, x, cs, r, J, A, g);
        gmax = 0;
        for (int j = 0; j < n2; j++) gmax = fmax(gmax, fabs(g[j]));
        if (gmax <= gtol) break;
        if (fabs(change) <= ftol * old) break;
      } else {
        radius /= nu_f; nu_f *= 2.0;
        if (radius < 1e-32) break;
      }
    }
  costs[1] = cost;
  info[0] = it; info[1] = good;
  /* ---- DiffProp (SchwarpDatabase.cc:243-345): six evaluations, stored as float32 key points */
  double* arr = (double*)malloc(sizeof(double) * n2);
  interleave(&s, x, arr);
  double *u = (double*)malloc(sizeof(double) * P), *v = (double*)malloc(sizeof(double) * P), *val = (double*)malloc(sizeof(double) * 2 * P);
  for (int i = 0; i < P; i++) { u[i] = kp1[2 * i]; v[i] = kp1[2 * i + 1]; }
  static const int ord[6][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}, {2, 0}, {0, 2}};
  float* ev = (float*)malloc(sizeof(float) * 12 * P);   /* qe, dqu, dqv, dquv, dquu, dqvv (x,y each) */
  for (int o = 0; o < 6; o++) {
    bbs_oracle_eval(umin, umax, nu, vmin, vmax, nv, 2, arr, u, v, P, ord[o][0], ord[o][1], val, 0);
    for (int i = 0; i < P; i++) { ev[(o * P + i) * 2] = (float)val[2 * i]; ev[(o * P + i) * 2 + 1] = (float)val[2 * i + 1]; }
  }
  for (int i = 0; i < P; i++) {
    const float *qe = &ev[(0 * P + i) * 2], *dqu = &ev[(1 * P + i) * 2], *dqv = &ev[(2 * P + i) * 2], *dquv = &ev[(3 * P + i) * 2],
                *dquu = &ev[(4 * P + i) * 2], *dqvv = &ev[(5 * P + i) * 2];
    float ex = qe[0] - kp2[2 * i], ey = qe[1] - kp2[2 * i + 1];
    ex *= fx_true; ey *= fy_true;
    drop[i] = sqrt((double)ex * ex + (double)ey * ey) > 10;
    float* d = diff + 18 * i;
    d[0] = kp1[2 * i]; d[1] = kp1[2 * i + 1]; d[2] = kp2[2 * i]; d[3] = kp2[2 * i + 1];
    d[4] = dqu[0]; d[5] = dqu[1]; d[6] = dqv[0]; d[7] = dqv[1];     /* J12a, J12b, J12c, J12d */
    const float det = dqu[0] * dqv[1] - dqv[0] * dqu[1];
    d[8] = d[7] / det; d[9] = -d[6] / det; d[10] = -d[5] / det; d[11] = d[4] / det;   /* J21a, J21b, J21c, J21d */
    d[12] = dquu[0]; d[13] = dquu[1]; d[14] = dquv[0]; d[15] = dquv[1]; d[16] = dqvv[0]; d[17] = dqvv[1];
  }
  free(arr); free(u); free(v); free(val); free(ev);
  free(r); free(J); free(A); free(M); free(g); free(cs); free(dx); free(xn); free(rhs);
}
