/* Small fixed-size algebra shared by the CPU oracles (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py):
 * Eigen quaternion conventions, g2o SE3Quat, and Eigen::LDLT restated (unblocked, diagonal pivoting).
 * Everything is static: each oracle translation unit gets its own copy. */
#ifndef DEFSLAM_ORACLE_SMALL_ALGEBRA_H
#define DEFSLAM_ORACLE_SMALL_ALGEBRA_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#if defined(__GNUC__)
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wunused-function"
#endif

typedef struct { double x, y, z, w; } quat_t;
typedef struct { quat_t r; double t[3]; } se3_t;

static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* Eigen: QuaternionBase::operator=(MatrixBase) (rotation matrix -> quaternion). */
static quat_t quat_from_R(const double R[9]) {
  quat_t q; double c[4];
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    c[3] = 0.5 * t;
    t = 0.5 / t;
    c[0] = (R[7] - R[5]) * t;
    c[1] = (R[2] - R[6]) * t;
    c[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    c[i] = 0.5 * t;
    t = 0.5 / t;
    c[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    c[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    c[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  q.x = c[0]; q.y = c[1]; q.z = c[2]; q.w = c[3];
  return q;
}

/* se3quat.h:280-285 normalizeRotation */
static void quat_normalize_pos(quat_t* q) {
  if (q->w < 0) { q->x *= -1; q->y *= -1; q->z *= -1; q->w *= -1; }
  double n = sqrt(q->x * q->x + q->y * q->y + q->z * q->z + q->w * q->w);
  q->x /= n; q->y /= n; q->z /= n; q->w /= n;
}

/* Eigen: QuaternionBase::_transformVector */
static void quat_rot(const quat_t* q, const double v[3], double o[3]) {
  double qv[3] = {q->x, q->y, q->z}, uv[3], c2[3];
  cross3(qv, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(qv, uv, c2);
  o[0] = v[0] + q->w * uv[0] + c2[0];
  o[1] = v[1] + q->w * uv[1] + c2[1];
  o[2] = v[2] + q->w * uv[2] + c2[2];
}

/* Eigen: quaternion product a*b */
static quat_t quat_mul(const quat_t* a, const quat_t* b) {
  quat_t r;
  r.w = a->w * b->w - a->x * b->x - a->y * b->y - a->z * b->z;
  r.x = a->w * b->x + a->x * b->w + a->y * b->z - a->z * b->y;
  r.y = a->w * b->y + a->y * b->w + a->z * b->x - a->x * b->z;
  r.z = a->w * b->z + a->z * b->w + a->x * b->y - a->y * b->x;
  return r;
}

/* Eigen: QuaternionBase::toRotationMatrix */
static void quat_to_R(const quat_t* q, double R[9]) {
  double tx = 2 * q->x, ty = 2 * q->y, tz = 2 * q->z;
  double twx = tx * q->w, twy = ty * q->w, twz = tz * q->w;
  double txx = tx * q->x, txy = ty * q->x, txz = tz * q->x;
  double tyy = ty * q->y, tyz = tz * q->y, tzz = tz * q->z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* se3quat.h:217-220 */
static void se3_map(const se3_t* T, const double p[3], double o[3]) {
  quat_rot(&T->r, p, o);
  o[0] += T->t[0]; o[1] += T->t[1]; o[2] += T->t[2];
}

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}

/* se3quat.h:223-257: exp of [omega, upsilon] */
static se3_t se3_exp(const double u[6]) {
  double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
  double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double Om2[9], R[9], V[9];
  mat3_mul(Om, Om, Om2);
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = (I3[i] + Om[i]) + Om2[i]; V[i] = R[i]; }
  } else {
    double a = sin(theta) / theta;
    double b = (1 - cos(theta)) / (theta * theta);
    double c = (theta - sin(theta)) / (pow(theta, 3));
    for (int i = 0; i < 9; i++) {
      R[i] = (I3[i] + a * Om[i]) + b * Om2[i];
      V[i] = (I3[i] + b * Om[i]) + c * Om2[i];
    }
  }
  se3_t T;
  T.r = quat_from_R(R);
  for (int i = 0; i < 3; i++) T.t[i] = V[i * 3 + 0] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
  quat_normalize_pos(&T.r);
  return T;
}

/* se3quat.h:104-110: a * b */
static se3_t se3_mul(const se3_t* a, const se3_t* b) {
  se3_t r = *a;
  double rt[3];
  quat_rot(&a->r, b->t, rt);
  r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
  r.r = quat_mul(&a->r, &b->r);
  quat_normalize_pos(&r.r);
  return r;
}

/* Converter.cc:35-45 (float32 4x4 row-major -> SE3Quat) */
static se3_t se3_from_f32(const float* Tcw) {
  double R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i * 3 + j] = (double)Tcw[i * 4 + j];
  se3_t T;
  T.r = quat_from_R(R);
  T.t[0] = (double)Tcw[3]; T.t[1] = (double)Tcw[7]; T.t[2] = (double)Tcw[11];
  quat_normalize_pos(&T.r);
  return T;
}

/* Converter.cc:47-66 + se3quat.h:269-277 */
static void se3_to_f32(const se3_t* T, float* out) {
  double R[9];
  quat_to_R(&T->r, R);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) out[i * 4 + j] = (float)R[i * 3 + j];
    out[i * 4 + 3] = (float)T->t[i];
  }
  out[12] = 0.f; out[13] = 0.f; out[14] = 0.f; out[15] = 1.f;
}


/* Eigen::LDLT restated (unblocked, diagonal pivoting, column-major, in place, lower). Returns 1 if "isPositive()". */
static int ldlt_pivoted(double* A, int n, int* perm, double* tmp) {
  /* A column-major n x n, lower triangle significant. */
  int sign = 0; /* 0 zero, 1 possemidef, -1 negsemidef, 2 indefinite */
  int found_zero_pivot = 0;
  if (n <= 1) {
    perm[0] = 0;
    if (n == 1) { if (A[0] > 0) sign = 1; else if (A[0] < 0) sign = -1; }
    return sign == 1 || sign == 0;
  }
  for (int k = 0; k < n; k++) {
    /* biggest |diag| in the trailing corner */
    int p = k; double big = fabs(A[k + (size_t)k * n]);
    for (int i = k + 1; i < n; i++) {
      double v = fabs(A[i + (size_t)i * n]);
      if (v > big) { big = v; p = i; }
    }
    perm[k] = p;
    if (p != k) {
      /* symmetric swap of rows/cols k and p on the lower triangle */
      for (int j = 0; j < k; j++) { double t = A[k + (size_t)j * n]; A[k + (size_t)j * n] = A[p + (size_t)j * n]; A[p + (size_t)j * n] = t; }
      for (int i = p + 1; i < n; i++) { double t = A[i + (size_t)k * n]; A[i + (size_t)k * n] = A[i + (size_t)p * n]; A[i + (size_t)p * n] = t; }
      { double t = A[k + (size_t)k * n]; A[k + (size_t)k * n] = A[p + (size_t)p * n]; A[p + (size_t)p * n] = t; }
      for (int i = k + 1; i < p; i++) { double t = A[i + (size_t)k * n]; A[i + (size_t)k * n] = A[p + (size_t)i * n]; A[p + (size_t)i * n] = t; }
    }
    int rs = n - k - 1;
    if (k > 0) {
      /* tmp = D(0:k) * A10^T ; A(k,k) -= A10*tmp ; A21 -= A20*tmp */
      double s = 0;
      for (int j = 0; j < k; j++) { tmp[j] = A[j + (size_t)j * n] * A[k + (size_t)j * n]; s += A[k + (size_t)j * n] * tmp[j]; }
      A[k + (size_t)k * n] -= s;
      if (rs > 0) {
        double* a21 = &A[(k + 1) + (size_t)k * n];
        for (int j = 0; j < k; j++) {
          const double tj = tmp[j];
          const double* a20 = &A[(k + 1) + (size_t)j * n];
          for (int i = 0; i < rs; i++) a21[i] -= a20[i] * tj;
        }
      }
    }
    double akk = A[k + (size_t)k * n];
    int pivot_valid = fabs(akk) > 0.0;
    if (k == 0 && !pivot_valid) {
      sign = 0;
      for (int j = 0; j < n; j++) perm[j] = j;
      return 1;
    }
    if (rs > 0 && pivot_valid) {
      double* a21 = &A[(k + 1) + (size_t)k * n];
      for (int i = 0; i < rs; i++) a21[i] /= akk;
    } else if (rs > 0) {
      /* zero pivot: Eigen checks the column is (near) zero too; irrelevant for SPD input */
      found_zero_pivot = 1;
    }
    if (sign == 1) { if (akk < 0) sign = 2; }
    else if (sign == -1) { if (akk > 0) sign = 2; }
    else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
  }
  (void)found_zero_pivot;
  return sign == 1 || sign == 0;
}

static void ldlt_pivoted_solve(const double* A, int n, const int* perm, const double* b, double* x) {
  for (int i = 0; i < n; i++) x[i] = b[i];
  for (int k = 0; k < n; k++) { int p = perm[k]; if (p != k) { double t = x[k]; x[k] = x[p]; x[p] = t; } }
  /* L y = Pb (unit lower), column oriented */
  for (int j = 0; j < n; j++) {
    double xj = x[j];
    const double* col = &A[(size_t)j * n];
    for (int i = j + 1; i < n; i++) x[i] -= col[i] * xj;
  }
  /* D: pseudo-inverse with Eigen's tolerance 1/highest */
  const double tol = 1.0 / DBL_MAX;
  for (int i = 0; i < n; i++) {
    double d = A[i + (size_t)i * n];
    if (fabs(d) > tol) x[i] /= d; else x[i] = 0;
  }
  /* L^T z = y */
  for (int j = n - 1; j >= 0; j--) {
    const double* col = &A[(size_t)j * n];
    double s = x[j];
    for (int i = j + 1; i < n; i++) s -= col[i] * x[i];
    x[j] = s;
  }
  for (int k = n - 1; k >= 0; k--) { int p = perm[k]; if (p != k) { double t = x[k]; x[k] = x[p]; x[p] = t; } }
}


#if defined(__GNUC__)
#pragma GCC diagnostic pop
#endif
#endif
