/*
 * sfn_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never on the product path).
 *
 * Restatement of the reference's Shape-from-Normals step: the depth B-spline of a
 * keyframe from the surface normals NRSfM estimated.
 *   bending matrix ........................ Thirdparty/BBS/bbs.cc:556-641 (bending_ur), bbs_coloc.cc:406-508
 *   normal-constraint rows M .............. Modules/Mapping/ShapeFromNormals.cc:178-260 (obtainM)
 *   stacked system + mean-depth row, QR ... Modules/Mapping/ShapeFromNormals.cc:38-100 (ctor, estimate)
 *   median scale, surface points .......... Modules/Mapping/ShapeFromNormals.cc:101-171
 *
 * The reference's bending code carries three 256-entry tables of precomputed coefficients.  They are not copied:
 * bend_tables() derives them from what they are -- integrals over one knot cell of products of the cubic B-spline
 * pieces and their derivatives ( B_xx[c][d] = I2(f_c, f_d) I0(e_c, e_d), B_yy = I0 I2, B_xy = 2 I1 I1 with
 * I_k(p, q) = int_0^1 b_p^(k) b_q^(k) dt ) -- by exact polynomial integration.
 *
 * PARITY: the bending matrix is PINNED against the reference's own bending_ur compiled into oracle/_ref/libbbs_ref.so
 * (tests/test_oracle_sfn.py, to 1e-15 relative: the reference sums the same terms in the same order from rounded table
 * entries).  The least-squares solve is UNPINNED (Eigen's HouseholderQR is not in this image): it is restated as a plain
 * unpivoted Householder QR and cross-checked against numpy.linalg.lstsq.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int bbs_oracle_coloc(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, const double* u, const double* v, int n, int du, int dv,
                     int32_t* cols, double* w);
void bbs_oracle_eval(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, int valdim, const double* ctrl, const double* u, const double* v,
                     int n, int du, int dv, double* val, uint8_t* status);

/* cubic B-spline pieces on t in [0,1] as polynomial coefficients c0 + c1 t + c2 t^2 + c3 t^3 (bbs.cc:95-121, order 0) */
static const double PIECE[4][4] = {
    {1.0 / 6, -3.0 / 6, 3.0 / 6, -1.0 / 6}, {4.0 / 6, 0.0, -6.0 / 6, 3.0 / 6}, {1.0 / 6, 3.0 / 6, 3.0 / 6, -3.0 / 6}, {0.0, 0.0, 0.0, 1.0 / 6}};

static void poly_deriv(const double* p, int k, double* out) { /* k-th derivative of a cubic, 4 coefficients out */
  double c[4] = {p[0], p[1], p[2], p[3]};
  for (int s = 0; s < k; s++) {
    double d[4] = {c[1], 2 * c[2], 3 * c[3], 0.0};
    memcpy(c, d, sizeof c);
  }
  memcpy(out, c, sizeof c);
}

static double poly_prod_integral(const double* a, const double* b) { /* int_0^1 a(t) b(t) dt */
  double s = 0.0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) s += a[i] * b[j] / (double)(i + j + 1);
  return s;
}

/* I[k][p][q] = int_0^1 b_p^(k) b_q^(k) dt, k = 0, 1, 2 */
static void bend_tables(double I[3][4][4]) {
  for (int k = 0; k < 3; k++)
    for (int p = 0; p < 4; p++)
      for (int q = 0; q < 4; q++) {
        double a[4], b[4];
        poly_deriv(PIECE[p], k, a);
        poly_deriv(PIECE[q], k, b);
        I[k][p][q] = poly_prod_integral(a, b);
      }
}

/* Dense symmetric N x N bending matrix, N = nptsu * nptsv, control point (iu, iv) at index iu * nptsv + iv.
 * One knot cell at a time, pairs (c <= d) of the 16 local basis functions, local index c = 4 e + f with e along u and f along v;
 * "x" of the reference's formulas is the v direction (fast index), "y" the u direction (bbs.cc:566-567, 606-613). */
void sfn_oracle_bending(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, double lambda, double* Bm) {
  const int nx = nptsv, ny = nptsu, N = nptsu * nptsv;
  const double sy = (umax - umin) / (nptsu - 3), sx = (vmax - vmin) / (nptsv - 3);
  double I[3][4][4];
  bend_tables(I);
  double coeff[16][16];
  for (int d = 0; d < 16; d++)
    for (int c = 0; c <= d; c++) {
      const int e1 = c / 4, f1 = c % 4, e2 = d / 4, f2 = d % 4;
      const double bxx = I[2][f1][f2] * I[0][e1][e2], byy = I[0][f1][f2] * I[2][e1][e2], bxy = 2.0 * I[1][f1][f2] * I[1][e1][e2];
      coeff[d][c] = sy * bxx / pow(sx, 3) + bxy / (sx * sy) + sx * byy / pow(sy, 3);
    }
  memset(Bm, 0, sizeof(double) * (size_t)N * N);
  for (int b = 0; b < ny - 3; b++)
    for (int a = 0; a < nx - 3; a++)
      for (int c = 0; c < 16; c++)
        for (int d = c; d < 16; d++) {
          const int i = (b + c / 4) * nx + a + c % 4, j = (b + d / 4) * nx + a + d % 4;
          Bm[(size_t)i * N + j] += lambda * coeff[d][c];
          if (i != j) Bm[(size_t)j * N + i] = Bm[(size_t)i * N + j];
        }
}

/* Rows of obtainM for n sites with unit-normalised normals: row i = (n.eta) coloc_du_i + n_x coloc_i,
 * row i + n = (n.eta) coloc_dv_i + n_y coloc_i, eta = (u, v, 1).  M is (2n) x N row-major. */
void sfn_oracle_rows(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, int n, const double* u, const double* v, const float* normals,
                     double* M) {
  const int N = nptsu * nptsv;
  int32_t* cols = (int32_t*)malloc(sizeof(int32_t) * 16 * (size_t)n);
  double* w0 = (double*)malloc(sizeof(double) * 16 * (size_t)n);
  double* wu = (double*)malloc(sizeof(double) * 16 * (size_t)n);
  double* wv = (double*)malloc(sizeof(double) * 16 * (size_t)n);
  bbs_oracle_coloc(umin, umax, nptsu, vmin, vmax, nptsv, u, v, n, 0, 0, cols, w0);
  bbs_oracle_coloc(umin, umax, nptsu, vmin, vmax, nptsv, u, v, n, 1, 0, cols, wu);
  bbs_oracle_coloc(umin, umax, nptsu, vmin, vmax, nptsv, u, v, n, 0, 1, cols, wv);
  memset(M, 0, sizeof(double) * 2 * (size_t)n * N);
  for (int i = 0; i < n; i++) {
    double nx = normals[3 * i], ny = normals[3 * i + 1], nz = normals[3 * i + 2];   /* cv::Vec3f -> Eigen::Vector3d, normalised in double */
    const double nn = sqrt(nx * nx + ny * ny + nz * nz);
    nx /= nn; ny /= nn; nz /= nn;
    const double ne = nx * u[i] + ny * v[i] + nz;
    for (int k = 0; k < 16; k++) {
      const int col = cols[16 * i + k];
      if (col < 0) continue;   /* site outside the definition domain: no constraint */
      M[(size_t)i * N + col] += ne * wu[16 * i + k] + nx * w0[16 * i + k];
      M[(size_t)(i + n) * N + col] += ne * wv[16 * i + k] + ny * w0[16 * i + k];
    }
  }
  free(cols); free(w0); free(wu); free(wv);
}

/* min |A x - b| by unpivoted Householder QR (what Eigen::HouseholderQR::solve computes), A is m x n row-major, overwritten. */
static int householder_lstsq(int m, int n, double* A, double* b, double* x) {
  for (int k = 0; k < n; k++) {
    double norm = 0.0;
    for (int i = k; i < m; i++) norm += A[(size_t)i * n + k] * A[(size_t)i * n + k];
    norm = sqrt(norm);
    if (norm == 0.0) return 0;
    const double alpha = A[(size_t)k * n + k] > 0 ? -norm : norm;
    const double v0 = A[(size_t)k * n + k] - alpha;
    double vnorm2 = v0 * v0;
    for (int i = k + 1; i < m; i++) vnorm2 += A[(size_t)i * n + k] * A[(size_t)i * n + k];
    /* apply H = I - 2 v v^T / (v^T v) to the remaining columns and to b; v = (v0, A[k+1.., k]) */
    for (int j = k + 1; j < n; j++) {
      double s = v0 * A[(size_t)k * n + j];
      for (int i = k + 1; i < m; i++) s += A[(size_t)i * n + k] * A[(size_t)i * n + j];
      s = 2.0 * s / vnorm2;
      A[(size_t)k * n + j] -= s * v0;
      for (int i = k + 1; i < m; i++) A[(size_t)i * n + j] -= s * A[(size_t)i * n + k];
    }
    double s = v0 * b[k];
    for (int i = k + 1; i < m; i++) s += A[(size_t)i * n + k] * b[i];
    s = 2.0 * s / vnorm2;
    b[k] -= s * v0;
    for (int i = k + 1; i < m; i++) b[i] -= s * A[(size_t)i * n + k];
    A[(size_t)k * n + k] = alpha;
  }
  for (int k = n - 1; k >= 0; k--) {
    double s = b[k];
    for (int j = k + 1; j < n; j++) s -= A[(size_t)k * n + j] * x[j];
    x[k] = s / A[(size_t)k * n + k];
  }
  return 1;
}

static int cmp_float(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

/* ShapeFromNormals::estimate.  n sites with normals; n_all key points get a surface point.
 * Returns 1 on success, 0 on failure (no key points / NaN / Inf / rank deficiency).
 * ctrl_raw: least-squares solution before the median scaling (for diagnostics), ctrl: scaled control points (Surface::saveArray),
 * pts: float32 (u d, v d, d) per key point (Surface::set3DSurfacePoint). */
int sfn_oracle_estimate(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, int n, const double* u, const double* v, const float* normals,
                        double bending_weight, double mean_depth, int n_all, const double* u_all, const double* v_all, double* ctrl_raw, double* ctrl,
                        float* pts) {
  const int N = nptsu * nptsv, m = 2 * n + N + 1;
  double* A = (double*)calloc((size_t)m * N, sizeof(double));
  double* b = (double*)calloc((size_t)m, sizeof(double));
  sfn_oracle_rows(umin, umax, nptsu, vmin, vmax, nptsv, n, u, v, normals, A);
  sfn_oracle_bending(umin, umax, nptsu, vmin, vmax, nptsv, bending_weight, A + (size_t)2 * n * N);
  for (int j = 0; j < N; j++) A[(size_t)(m - 1) * N + j] = 1.0;
  b[m - 1] = (double)N * mean_depth;
  int ok = householder_lstsq(m, N, A, b, ctrl_raw);
  free(A); free(b);
  if (!ok || n_all == 0) return 0;
  for (int i = 0; i < N; i++)
    if (isnan(ctrl_raw[i]) || isinf(ctrl_raw[i])) return 0;
  float* dv = (float*)malloc(sizeof(float) * (size_t)N);
  for (int i = 0; i < N; i++) dv[i] = (float)ctrl_raw[i];
  qsort(dv, (size_t)N, sizeof(float), cmp_float);
  const float corr = 1 / dv[N / 2];
  free(dv);
  for (int i = 0; i < N; i++) ctrl[i] = corr * ctrl_raw[i];
  /* depth at every key point (BBS::EvalEigen, order 0: the pinned tensor-product evaluation) */
  double* depth = (double*)malloc(sizeof(double) * (size_t)n_all);
  bbs_oracle_eval(umin, umax, nptsu, vmin, vmax, nptsv, 1, ctrl, u_all, v_all, n_all, 0, 0, depth, 0);
  for (int i = 0; i < n_all; i++) {
    const double d = depth[i];
    pts[3 * i] = (float)(u_all[i] * d);
    pts[3 * i + 1] = (float)(v_all[i] * d);
    pts[3 * i + 2] = (float)d;
  }
  free(depth);
  return 1;
}


/* Warps::Warp::initialize (Modules/Mapping/Schwarp.cc:99-160): control points of the warp kp1 -> kp2 from the regularised
 * linear least squares  (C^T C + Bending(lambda)) X = C^T q2,  C = colocation matrix of kp1 (P x N), X is N x 2 stored
 * column-major in x[2N].  The reference solves with Eigen::SimplicialLDLT (sparse LDLT) -> UNPINNED; restated with a dense
 * Cholesky.  Returns 1 on success, 0 when the matrix is not positive definite. */
int warp_oracle_initialize(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, int P, const float* kp1, const float* kp2, double lambda,
                           double* x) {
  const int N = nptsu * nptsv;
  double* u = (double*)malloc(sizeof(double) * (size_t)P);
  double* v = (double*)malloc(sizeof(double) * (size_t)P);
  int32_t* cols = (int32_t*)malloc(sizeof(int32_t) * 16 * (size_t)P);
  double* w = (double*)malloc(sizeof(double) * 16 * (size_t)P);
  double* G = (double*)malloc(sizeof(double) * (size_t)N * N);
  for (int i = 0; i < P; i++) { u[i] = kp1[2 * i]; v[i] = kp1[2 * i + 1]; }
  bbs_oracle_coloc(umin, umax, nptsu, vmin, vmax, nptsv, u, v, P, 0, 0, cols, w);
  sfn_oracle_bending(umin, umax, nptsu, vmin, vmax, nptsv, lambda, G);
  for (int j = 0; j < 2 * N; j++) x[j] = 0.0;
  for (int i = 0; i < P; i++)
    for (int a = 0; a < 16; a++) {
      const int ca = cols[16 * i + a];
      if (ca < 0) continue;
      for (int b = 0; b < 16; b++) G[(size_t)ca * N + cols[16 * i + b]] += w[16 * i + a] * w[16 * i + b];
      x[ca] += w[16 * i + a] * (double)kp2[2 * i];
      x[N + ca] += w[16 * i + a] * (double)kp2[2 * i + 1];
    }
  int ok = 1;
  for (int k = 0; k < N && ok; k++) {          /* in-place Cholesky, lower */
    double d = G[(size_t)k * N + k];
    for (int j = 0; j < k; j++) d -= G[(size_t)k * N + j] * G[(size_t)k * N + j];
    if (!(d > 0)) { ok = 0; break; }
    const double piv = sqrt(d);
    G[(size_t)k * N + k] = piv;
    for (int r = k + 1; r < N; r++) {
      double t = G[(size_t)r * N + k];
      for (int j = 0; j < k; j++) t -= G[(size_t)r * N + j] * G[(size_t)k * N + j];
      G[(size_t)r * N + k] = t / piv;
    }
  }
  for (int c = 0; c < 2 && ok; c++) {
    double* y = x + (size_t)c * N;
    for (int i = 0; i < N; i++) { double t = y[i]; for (int j = 0; j < i; j++) t -= G[(size_t)i * N + j] * y[j]; y[i] = t / G[(size_t)i * N + i]; }
    for (int i = N - 1; i >= 0; i--) { double t = y[i]; for (int j = i + 1; j < N; j++) t -= G[(size_t)j * N + i] * y[j]; y[i] = t / G[(size_t)i * N + i]; }
  }
  free(u); free(v); free(cols); free(w); free(G);
  return ok;
}
