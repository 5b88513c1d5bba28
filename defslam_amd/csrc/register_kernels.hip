// Surface registration and template embedding on the device, gfx950.
//
//   embed_kernel        TriangularMesh::calculateFeaturesCoordinates / pointInTriangle  (TriangularMesh.cc:133-236)
//   smm_*_kernel        GroundTruthTools::scaleMinMedian                                 (GroundTruthCalculator.cc:54-160)
//   horn_lm_kernel      Optimizer::OptimizeHorn: g2o LM on one Sim(3) vertex, numeric Jacobians, Huber
//                       (DefOptimizer.cc:840-922, sim3.h:71-140, base_unary_edge.hpp:44-125,
//                        optimization_algorithm_levenberg.cpp:61-189)
//
// All of it is small (a keyframe has ~1000 point pairs): the point of running it here is that the clouds, the
// template and the result stay in HBM between the mapping kernels and the next SfT solve.  One workgroup runs the
// whole Levenberg-Marquardt loop (two optimize(50) calls) like the SfT kernel does; sums use a fixed reduction tree.
// Compiled with -ffp-contract=off: the reference's float32/float64 expression order is kept.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

namespace {

// ------------------------------------------------------------------------------------------------------------------
// fixed-tree block sum of K doubles per thread (256 threads); result in out[0..K) (LDS), valid after the call
// ------------------------------------------------------------------------------------------------------------------
template <int K>
__device__ void block_sum256(double* v, double* red /* 4*K */, double* out /* K */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < K; i++) {
    double x = v[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    if (lane == 0) red[w * K + i] = x;
  }
  __syncthreads();
  if (threadIdx.x < K) out[threadIdx.x] = (red[threadIdx.x] + red[K + threadIdx.x]) + (red[2 * K + threadIdx.x] + red[3 * K + threadIdx.x]);
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------------
// Embedding: one wavefront per point
// ------------------------------------------------------------------------------------------------------------------
__device__ bool point_in_triangle(const float* q, const float* v0, const float* v1, const float* v2, float* bary) {
  float u[3], v[3], w[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { u[k] = v1[k] - v0[k]; v[k] = v2[k] - v0[k]; w[k] = q[k] - v0[k]; }
  const float nx = u[1] * v[2] - u[2] * v[1], ny = u[2] * v[0] - u[0] * v[2], nz = u[0] * v[1] - u[1] * v[0];
  const float ax = u[1] * w[2] - u[2] * w[1], ay = u[2] * w[0] - u[0] * w[2], az = u[0] * w[1] - u[1] * w[0];
  const float bx = w[1] * v[2] - w[2] * v[1], by = w[2] * v[0] - w[0] * v[2], bz = w[0] * v[1] - w[1] * v[0];
  const float n2 = nx * nx + ny * ny + nz * nz;
  const float gamma = (ax * nx + ay * ny + az * nz) / n2;
  const float beta = (bx * nx + by * ny + bz * nz) / n2;
  const float alpha = 1 - gamma - beta;
  bary[0] = alpha; bary[1] = beta; bary[2] = gamma;
  float d2 = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float proj = v0[k] * alpha + v1[k] * beta + v2[k] * gamma;
    const float df = proj - q[k];
    d2 += df * df;
  }
  if ((double)d2 > 1E-1) return false;
  return (0 <= alpha) && (alpha <= 1) && (0 <= beta) && (beta <= 1) && (0 <= gamma) && (gamma <= 1);
}

__global__ __launch_bounds__(256) void embed_kernel(int P, const float* __restrict__ pts, int n, const double* __restrict__ xyz0,
                                                    const int32_t* __restrict__ facets, const int32_t* __restrict__ nf_ptr,
                                                    const int32_t* __restrict__ nf_idx, int32_t* __restrict__ facet_id,
                                                    int32_t* __restrict__ nodes, float* __restrict__ bary) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const float mp[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
  // closest node: first index of the minimum distance below 100 (TriangularMesh.cc:152-163)
  double best = 100;
  int closest = -1;
  for (int i = lane; i < n; i += 64) {
    const double dx = xyz0[3 * i] - mp[0], dy = xyz0[3 * i + 1] - mp[1], dz = xyz0[3 * i + 2] - mp[2];
    const double dist = sqrt((dx * dx + dy * dy) + dz * dz);
    if (dist < best) { best = dist; closest = i; }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const double ob = __shfl_xor(best, m, 64);
    const int oc = __shfl_xor(closest, m, 64);
    if (oc >= 0 && (closest < 0 || ob < best || (ob == best && oc < closest))) { best = ob; closest = oc; }
  }
  int fid = -1, nd[3] = {-1, -1, -1};
  float bb[3] = {0.f, 0.f, 0.f};
  if (closest >= 0) {
    const int q0 = nf_ptr[closest], q1 = nf_ptr[closest + 1];
    for (int base = q0; base < q1 && fid < 0; base += 64) {
      const int q = base + lane;
      bool hit = false;
      float b[3] = {0.f, 0.f, 0.f};
      int f = -1;
      if (q < q1) {
        f = nf_idx[q];
        float v[3][3];
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
          for (int k = 0; k < 3; k++) v[s][k] = (float)xyz0[3 * facets[3 * f + s] + k];
        hit = point_in_triangle(mp, v[0], v[1], v[2], b);
      }
      const unsigned long long mask = __ballot(hit);
      if (mask) {
        const int src = __ffsll((long long)mask) - 1;   // first facet of the node's list that contains the point
        fid = __shfl(f, src, 64);
#pragma unroll
        for (int k = 0; k < 3; k++) bb[k] = __shfl(b[k], src, 64);
#pragma unroll
        for (int k = 0; k < 3; k++) nd[k] = facets[3 * fid + k];
      }
    }
  }
  if (lane == 0) {
    facet_id[p] = fid;
#pragma unroll
    for (int k = 0; k < 3; k++) { nodes[3 * p + k] = nd[k]; bary[3 * p + k] = bb[k]; }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// scaleMinMedian.  The host walks the uniform stream once (which i are candidates, where their j-draws start);
// one workgroup per candidate computes its residual list and picks the reference's "median".
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smm_median_kernel(int n, const float* __restrict__ mono, const float* __restrict__ stereo,
                                                         const double* __restrict__ u, const int32_t* __restrict__ cand,
                                                         const int64_t* __restrict__ cand_off, float* __restrict__ medians,
                                                         double* __restrict__ scales) {
  extern __shared__ float res[];   // n residuals (-1: not selected), then 4 ints
  __shared__ int cnt[4];
  __shared__ float med_s;
  const int c = blockIdx.x, i = cand[c];
  const int64_t off = cand_off[c];
  const double scale = stereo[3 * i + 2] / mono[3 * i + 2];
  int m = 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    float r = -1.f;
    if (j != i && !(u[off + (j < i ? j : j - 1)] > 0.25)) {
      float r2 = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double d = (scale * mono[3 * j + k] - stereo[3 * j + k]);
        r2 = r2 + d * d;
      }
      r = sqrtf(r2);
      m++;
    }
    res[j] = r;
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) m += __shfl_xor(m, s, 64);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = m;
  if (threadIdx.x == 0) med_s = -1.f;
  __syncthreads();
  m = cnt[0] + cnt[1] + cnt[2] + cnt[3];
  if (m > 1) {
    // sorted non-negative residuals r_0 <= ... <= r_{m-1}; the reference copies from r_1 on and takes element (m-1)/2
    const int k = 1 + (m - 1) / 2;
    for (int j = threadIdx.x; j < n; j += 256) {
      const float v = res[j];
      if (v < 0) continue;
      int less = 0, leq = 0;
      for (int t = 0; t < n; t++) {
        const float x = res[t];
        less += (x >= 0 && x < v);
        leq += (x >= 0 && x <= v);
      }
      if (less <= k && k < leq) med_s = v;   // every writer holds the same value
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) { medians[c] = med_s; scales[c] = scale; }
}

// out: [0] final scale (float widened), [1] status (0 ok, 2 the reference's early `return 0.0`), [2] min_med, [3] desv
__global__ __launch_bounds__(256) void smm_finish_kernel(int n, int ncand, const float* __restrict__ mono, const float* __restrict__ stereo,
                                                         const float* __restrict__ medians, const double* __restrict__ scales,
                                                         double* __restrict__ out) {
  extern __shared__ float prod[];   // 2n: num, den products (NaN marks an outlier)
  __shared__ double bs_s;
  __shared__ float desv_s;
  __shared__ int early;
  if (threadIdx.x == 0) {
    float min_med = 10000.0f;
    double best_scale = 0.0;
    int e = 0;
    for (int c = 0; c < ncand; c++) {
      const float md = medians[c];
      if (md < 0) { e = 1; break; }
      if (md < min_med) { min_med = md; best_scale = scales[c]; }
    }
    const float desv = 1.4826 * (1.0 - (5.0 / (ncand - 1.0))) * sqrtf(min_med);
    bs_s = best_scale; desv_s = desv; early = e;
    out[2] = min_med; out[3] = desv;
  }
  __syncthreads();
  if (early) {
    if (threadIdx.x == 0) { out[0] = 0.0; out[1] = 2.0; }
    return;
  }
  const double best_scale = bs_s;
  const float desv = desv_s;
  for (int i = threadIdx.x; i < n; i += 256) {
    float residual = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double d = (best_scale * mono[3 * i + k] - stereo[3 * i + k]);
      residual = residual + d * d;
    }
    residual = sqrtf(residual);
    const bool in = (double)(residual / desv) < 2.5;
    prod[2 * i] = in ? (stereo[3 * i + 2] * mono[3 * i + 2]) : __int_as_float(0x7fc00000);
    prod[2 * i + 1] = mono[3 * i + 2] * mono[3 * i + 2];
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // float sums in index order, like the reference
    float num = 0.0f, den = 0.0f;
    int i = 0;
    for (; i + 8 <= n; i += 8) {            // eight pairs per LDS round trip, the additions stay in index order
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { a[u] = prod[2 * (i + u)]; b[u] = prod[2 * (i + u) + 1]; }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (a[u] == a[u]) { num += a[u]; den += b[u]; }
    }
    for (; i < n; i++) {
      const float a = prod[2 * i];
      if (a != a) continue;
      num += a;
      den += prod[2 * i + 1];
    }
    out[0] = (double)(num / den);
    out[1] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Sim(3) algebra (g2o sim3.h with Eigen's quaternion conventions)
// ------------------------------------------------------------------------------------------------------------------
struct Sim3 { double qx, qy, qz, qw, t[3], s; };

__device__ void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ void quat_rot(const Sim3& S, const double* v, double* o) {
  const double qv[3] = {S.qx, S.qy, S.qz};
  double uv[3], c2[3];
  cross3(qv, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(qv, uv, c2);
#pragma unroll
  for (int k = 0; k < 3; k++) o[k] = v[k] + S.qw * uv[k] + c2[k];
}
__device__ void sim3_map(const Sim3& S, const double* p, double* o) {
  double rp[3];
  quat_rot(S, p, rp);
#pragma unroll
  for (int k = 0; k < 3; k++) o[k] = S.s * rp[k] + S.t[k];
}
__device__ Sim3 sim3_mul(const Sim3& a, const Sim3& b) {
  Sim3 r;
  r.qw = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
  r.qx = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
  r.qy = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
  r.qz = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
  double rt[3];
  quat_rot(a, b.t, rt);
#pragma unroll
  for (int k = 0; k < 3; k++) r.t[k] = a.s * rt[k] + a.t[k];
  r.s = a.s * b.s;
  return r;
}
__device__ void quat_from_R(const double* R, Sim3& S) {
  double c[4];
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
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    c[i] = 0.5 * t;
    t = 0.5 / t;
    c[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    c[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    c[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  S.qx = c[0]; S.qy = c[1]; S.qz = c[2]; S.qw = c[3];
}
__device__ Sim3 sim3_exp(const double* u) {
  const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
  const double sigma = u[6];
  const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double Om2[9], R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Om[i * 3 + k] * Om[k * 3 + j];
      Om2[i * 3 + j] = s;
    }
  Sim3 S;
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
  quat_from_R(R, S);
  for (int i = 0; i < 3; i++) {
    const double w0 = (A * Om[3 * i] + B * Om2[3 * i]) + C * I3[3 * i];
    const double w1 = (A * Om[3 * i + 1] + B * Om2[3 * i + 1]) + C * I3[3 * i + 1];
    const double w2 = (A * Om[3 * i + 2] + B * Om2[3 * i + 2]) + C * I3[3 * i + 2];
    S.t[i] = (w0 * up[0] + w1 * up[1]) + w2 * up[2];
  }
  return S;
}

// Eigen::LDLT restated for a 7x7 matrix (column-major, lower), diagonal pivoting; returns isPositive()
__device__ bool ldlt7(double* A, int* perm) {
  constexpr int n = 7;
  int sign = 0;
  double tmp[n];
  for (int k = 0; k < n; k++) {
    int p = k;
    double big = fabs(A[k + k * n]);
    for (int i = k + 1; i < n; i++) {
      const double v = fabs(A[i + i * n]);
      if (v > big) { big = v; p = i; }
    }
    perm[k] = p;
    if (p != k) {
      for (int j = 0; j < k; j++) { const double t = A[k + j * n]; A[k + j * n] = A[p + j * n]; A[p + j * n] = t; }
      for (int i = p + 1; i < n; i++) { const double t = A[i + k * n]; A[i + k * n] = A[i + p * n]; A[i + p * n] = t; }
      { const double t = A[k + k * n]; A[k + k * n] = A[p + p * n]; A[p + p * n] = t; }
      for (int i = k + 1; i < p; i++) { const double t = A[i + k * n]; A[i + k * n] = A[p + i * n]; A[p + i * n] = t; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      double s = 0;
      for (int j = 0; j < k; j++) { tmp[j] = A[j + j * n] * A[k + j * n]; s += A[k + j * n] * tmp[j]; }
      A[k + k * n] -= s;
      for (int j = 0; j < k; j++)
        for (int i = 0; i < rs; i++) A[(k + 1 + i) + k * n] -= A[(k + 1 + i) + j * n] * tmp[j];
    }
    const double akk = A[k + k * n];
    const bool pivot_valid = fabs(akk) > 0.0;
    if (k == 0 && !pivot_valid) {
      for (int j = 0; j < n; j++) perm[j] = j;
      return true;
    }
    if (pivot_valid)
      for (int i = 0; i < rs; i++) A[(k + 1 + i) + k * n] /= akk;
    if (sign == 1) { if (akk < 0) sign = 2; }
    else if (sign == -1) { if (akk > 0) sign = 2; }
    else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
  }
  return sign == 1 || sign == 0;
}
__device__ void ldlt7_solve(const double* A, const int* perm, const double* b, double* x) {
  constexpr int n = 7;
  for (int i = 0; i < n; i++) x[i] = b[i];
  for (int k = 0; k < n; k++) { const int p = perm[k]; if (p != k) { const double t = x[k]; x[k] = x[p]; x[p] = t; } }
  for (int j = 0; j < n; j++)
    for (int i = j + 1; i < n; i++) x[i] -= A[i + j * n] * x[j];
  const double tol = 1.0 / DBL_MAX;
  for (int i = 0; i < n; i++) { const double d = A[i + i * n]; if (fabs(d) > tol) x[i] /= d; else x[i] = 0; }
  for (int j = n - 1; j >= 0; j--) {
    double s = x[j];
    for (int i = j + 1; i < n; i++) s -= A[i + j * n] * x[i];
    x[j] = s;
  }
  for (int k = n - 1; k >= 0; k--) { const int p = perm[k]; if (p != k) { const double t = x[k]; x[k] = x[p]; x[p] = t; } }
}

struct HornShared {
  Sim3 est, bak, pm[14];
  double H[49], Hs[49], b[7], x[7];
  double red[4 * 35], sum[35];
  double delta, dsqr, lambda, ni, currentChi, iniChi, rho;
  int nBad, qmax, stop, it, ok;
};

__device__ inline void huber(const HornShared& S, double e2, double& rho0, double& rho1) {
  if (e2 <= S.dsqr) { rho0 = e2; rho1 = 1.; }
  else { const double sq = sqrt(e2); rho0 = 2 * sq * S.delta - S.dsqr; rho1 = S.delta / sq; }
}

// errors of every edge at S.est (stored: the reference reads the edges' last computed error afterwards); robust chi2
__device__ double horn_errors(HornShared& S, int n, const float* p1, const float* p2, double* err) {
  double chi = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double a[3] = {p1[3 * i], p1[3 * i + 1], p1[3 * i + 2]};
    double m[3], e[3];
    sim3_map(S.est, a, m);
#pragma unroll
    for (int k = 0; k < 3; k++) { e[k] = (double)p2[3 * i + k] - m[k]; err[3 * i + k] = e[k]; }
    double r0, r1;
    huber(S, (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2], r0, r1);
    chi += r0;
  }
  block_sum256<1>(&chi, S.red, S.sum);
  return S.sum[0];
}

// out: [0..7] sim3 after the FIRST optimize (what the reference returns), [8] plain chi2 of all edges after the second,
//      [9] count, [10..11] iterations, [12..13] trials, [14] acceptable
__global__ __launch_bounds__(256) void horn_lm_kernel(int n, const float* __restrict__ p1, const float* __restrict__ p2, const double* __restrict__ sim3_in,
                                                      double chi, double huber_delta, double* __restrict__ err, double* __restrict__ out) {
  __shared__ HornShared S;
  const int tid = threadIdx.x;
  if (tid == 0) {
    S.est.qx = sim3_in[0]; S.est.qy = sim3_in[1]; S.est.qz = sim3_in[2]; S.est.qw = sim3_in[3];
    S.est.t[0] = sim3_in[4]; S.est.t[1] = sim3_in[5]; S.est.t[2] = sim3_in[6]; S.est.s = sim3_in[7];
    S.delta = huber_delta; S.dsqr = huber_delta * huber_delta;
  }
  for (int i = tid; i < 3 * n; i += 256) err[i] = 0.0;
  __syncthreads();
  for (int phase = 0; phase < 2; phase++) {
    if (tid == 0) {
      S.lambda = -1.; S.ni = 2.; S.nBad = 0; S.stop = 0; S.it = 0;
      for (int i = 0; i < 7; i++) S.x[i] = 0.0;
    }
    int total_trials = 0, it_count = 0;
    __syncthreads();
    for (int it = 0; it < 50 && n > 0; it++) {
      const double chi0 = horn_errors(S, n, p1, p2, err);
      // perturbed estimates for the central differences (delta 1e-9): Sim3(+-delta e_d) * estimate
      if (tid < 14) {
        double add[7] = {0, 0, 0, 0, 0, 0, 0};
        add[tid >> 1] = (tid & 1) ? -1e-9 : 1e-9;
        S.pm[tid] = sim3_mul(sim3_exp(add), S.est);
      }
      __syncthreads();
      {
        const double scalar = 1.0 / (2 * 1e-9);
        double acc[35];
#pragma unroll
        for (int i = 0; i < 35; i++) acc[i] = 0.0;
        for (int i = tid; i < n; i += 256) {
          const double a[3] = {p1[3 * i], p1[3 * i + 1], p1[3 * i + 2]};
          const double z[3] = {p2[3 * i], p2[3 * i + 1], p2[3 * i + 2]};
          double J[3][7];
#pragma unroll
          for (int d = 0; d < 7; d++) {
            double mp[3], mm[3];
            sim3_map(S.pm[2 * d], a, mp);
            sim3_map(S.pm[2 * d + 1], a, mm);
#pragma unroll
            for (int k = 0; k < 3; k++) J[k][d] = scalar * ((z[k] - mp[k]) - (z[k] - mm[k]));
          }
          const double e0 = err[3 * i], e1 = err[3 * i + 1], e2 = err[3 * i + 2];
          double r0, r1;
          huber(S, (e0 * e0 + e1 * e1) + e2 * e2, r0, r1);
          int q = 0;
#pragma unroll
          for (int c = 0; c < 7; c++)
#pragma unroll
            for (int r = c; r < 7; r++) acc[q++] += ((J[0][r] * r1) * J[0][c] + (J[1][r] * r1) * J[1][c]) + (J[2][r] * r1) * J[2][c];
#pragma unroll
          for (int r = 0; r < 7; r++) acc[28 + r] -= ((r1 * J[0][r]) * e0 + (r1 * J[1][r]) * e1) + (r1 * J[2][r]) * e2;
        }
        block_sum256<35>(acc, S.red, S.sum);
      }
      if (tid == 0) {
        int q = 0;
        for (int c = 0; c < 7; c++)
          for (int r = c; r < 7; r++) { S.H[r + 7 * c] = S.sum[q]; S.H[c + 7 * r] = S.sum[q]; q++; }
        for (int r = 0; r < 7; r++) S.b[r] = S.sum[28 + r];
        S.currentChi = chi0; S.iniChi = chi0;
        if (it == 0) {
          double maxDiag = 0.;
          for (int j = 0; j < 7; j++) { const double v = fabs(S.H[j + 7 * j]); if (v > maxDiag) maxDiag = v; }
          S.lambda = 1e-5 * maxDiag; S.ni = 2; S.nBad = 0;
        }
        S.rho = 0; S.qmax = 0;
      }
      __syncthreads();
      bool again;
      do {
        if (tid == 0) {
          S.bak = S.est;
          for (int i = 0; i < 49; i++) S.Hs[i] = S.H[i];
          for (int j = 0; j < 7; j++) S.Hs[j + 7 * j] += S.lambda;
          int perm[7];
          S.ok = ldlt7(S.Hs, perm) ? 1 : 0;
          if (S.ok) ldlt7_solve(S.Hs, perm, S.b, S.x);
          S.est = sim3_mul(sim3_exp(S.x), S.est);
        }
        __syncthreads();
        const double chiN = horn_errors(S, n, p1, p2, err);
        if (tid == 0) {
          const double tempChi = S.ok ? chiN : DBL_MAX;
          double rho = (S.currentChi - tempChi);
          double scale = 0.;
          for (int j = 0; j < 7; j++) scale += S.x[j] * (S.lambda * S.x[j] + S.b[j]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = alpha < (2. / 3.) ? alpha : (2. / 3.);
            const double sf = (1. / 3.) > alpha ? (1. / 3.) : alpha;
            S.lambda *= sf; S.ni = 2; S.currentChi = tempChi;
          } else {
            S.lambda *= S.ni; S.ni *= 2;
            S.est = S.bak;
          }
          S.rho = rho;
          S.qmax++;
        }
        __syncthreads();
        again = S.rho < 0 && S.qmax < 10;
      } while (again);
      total_trials += S.qmax;
      it_count++;
      if (tid == 0) {
        int stop = 0;
        if (S.qmax == 10 || S.rho == 0) stop = 1;
        else {
          if ((S.iniChi - S.currentChi) * 1e3 < S.iniChi) S.nBad++; else S.nBad = 0;
          if (S.nBad >= 3) stop = 1;
        }
        S.stop = stop;
      }
      __syncthreads();
      if (S.stop) break;
    }
    if (phase == 0) {
      // g2oS12 = vert0->estimate(); count of edges whose (stale) chi2 is within the limit
      double cnt = 0.0;
      for (int i = tid; i < n; i += 256) {
        const double e0 = err[3 * i], e1 = err[3 * i + 1], e2 = err[3 * i + 2];
        if (!(((e0 * e0 + e1 * e1) + e2 * e2) > chi)) cnt += 1.0;
      }
      block_sum256<1>(&cnt, S.red, S.sum);
      if (tid == 0) {
        out[0] = S.est.qx; out[1] = S.est.qy; out[2] = S.est.qz; out[3] = S.est.qw;
        out[4] = S.est.t[0]; out[5] = S.est.t[1]; out[6] = S.est.t[2]; out[7] = S.est.s;
        out[9] = S.sum[0];
      }
    } else {
      double tot = 0.0;
      for (int i = tid; i < n; i += 256) {
        const double e0 = err[3 * i], e1 = err[3 * i + 1], e2 = err[3 * i + 2];
        tot += (e0 * e0 + e1 * e1) + e2 * e2;
      }
      block_sum256<1>(&tot, S.red, S.sum);
      if (tid == 0) {
        const double total = S.sum[0];
        out[8] = total;
        out[14] = (isnan(total) || isinf(total)) ? 0.0 : ((total / out[9] < chi) ? 1.0 : 0.0);
      }
    }
    if (tid == 0) { out[10 + phase] = it_count; out[12 + phase] = total_trials; }
    __syncthreads();
  }
}

}  // namespace

extern "C" hipError_t reg_embed(int P, const float* pts, int n, const double* xyz0, const int32_t* facets, const int32_t* nf_ptr, const int32_t* nf_idx,
                                int32_t* facet_id, int32_t* nodes, float* bary, hipStream_t st) {
  if (P > 0) hipLaunchKernelGGL(embed_kernel, dim3((P + 3) / 4), dim3(256), 0, st, P, pts, n, xyz0, facets, nf_ptr, nf_idx, facet_id, nodes, bary);
  return hipGetLastError();
}

extern "C" hipError_t reg_scale_min_median(int n, int ncand, const float* mono, const float* stereo, const double* u, const int32_t* cand,
                                           const int64_t* cand_off, float* medians, double* scales, double* out, hipStream_t st) {
  if (ncand > 0) hipLaunchKernelGGL(smm_median_kernel, dim3(ncand), dim3(256), sizeof(float) * (size_t)n, st, n, mono, stereo, u, cand, cand_off, medians, scales);
  hipLaunchKernelGGL(smm_finish_kernel, dim3(1), dim3(256), sizeof(float) * 2 * (size_t)n, st, n, ncand, mono, stereo, medians, scales, out);
  return hipGetLastError();
}

extern "C" hipError_t reg_horn(int n, const float* p1, const float* p2, const double* sim3_in, double chi, double huber_delta, double* err, double* out,
                               hipStream_t st) {
  hipLaunchKernelGGL(horn_lm_kernel, dim3(1), dim3(256), 0, st, n, p1, p2, sim3_in, chi, huber_delta, err, out);
  return hipGetLastError();
}
