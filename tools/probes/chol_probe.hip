// Probe: cycles of the 16x16 Cholesky(+inverse) tile kernels, one wavefront: the first (rank-1 MFMA) variant, kept here for the record, and the blocked one factor_tiles uses (tile_chol.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define TS 16
#include "../../defslam_amd/csrc/tile_chol.h"   // the product's chol_inv_blocked (r04: shortened chain); the copy below is the r03 code
__device__ __forceinline__ void rsqrt_sqrt_first(double d, double& inv, double& s) {
  double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double res = fma(-g, g, d);
  g = fma(res, h, g);
  s = g; inv = h + h;
}
template <int VARIANT>
__device__ __forceinline__ bool chol_inv_mfma(v4d& a, v4d& w) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  w = (v4d){(g == c) ? 1.0 : 0.0, (g + 4 == c) ? 1.0 : 0.0, (g + 8 == c) ? 1.0 : 0.0, (g + 12 == c) ? 1.0 : 0.0};
  bool bad = false;
  double pd = bcast_lane(a[0], 0);
#pragma unroll
  for (int j = 0; j < TS; j++) {
    const int gj = j & 3, qj = j >> 2;
    if (!(pd > 0.0)) bad = true;
    double inv, sq;
    rsqrt_sqrt_first(pd, inv, sq);
    const double m = (g == gj) ? inv : 0.0;
    const double la = a[qj] * m;
    if (j + 1 < TS) {
      const double an = bcast_lane(a[(j + 1) >> 2], 16 * ((j + 1) & 3) + j + 1);
      const double ln = bcast_lane(la, 16 * gj + j + 1);
      pd = fma(-ln, ln, an);
    }
    const double nla = -la;
    if (VARIANT != 2) a = __builtin_amdgcn_mfma_f64_16x16x4f64(nla, la, a, 0, 0, 0);
    if (VARIANT == 0) {
      const double wr = w[qj] * m;
      const double u = (g == gj && c == j) ? (nla + 1.0) : nla;
      w = __builtin_amdgcn_mfma_f64_16x16x4f64(u, wr, w, 0, 0, 0);
    }
  }
  return !bad;
}
__device__ __forceinline__ void rsqrt_sqrt_k(double d, double& inv, double& s) {
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double res = fma(-g, g, d);
  s = fma(res, h, g);
  inv = fma(fma(-h, g, 0.5), h + h, h + h);   // one more correction of 1/sqrt without lengthening the sqrt chain
}
__device__ __forceinline__ bool chol_inv_blocked_r03(v4d& a, v4d& w) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  w = (v4d){(g == c) ? 1.0 : 0.0, (g + 4 == c) ? 1.0 : 0.0, (g + 8 == c) ? 1.0 : 0.0, (g + 12 == c) ? 1.0 : 0.0};
  bool bad = false;
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int J = 0; J < 4; J++) {
    const double aJ = a[J];
    const int b0 = 4 * J;
    const double d00 = bcast_lane(aJ, b0), d10 = bcast_lane(aJ, 16 + b0), d11 = bcast_lane(aJ, 16 + b0 + 1);
    const double d20 = bcast_lane(aJ, 32 + b0), d21 = bcast_lane(aJ, 32 + b0 + 1), d22 = bcast_lane(aJ, 32 + b0 + 2);
    const double d30 = bcast_lane(aJ, 48 + b0), d31 = bcast_lane(aJ, 48 + b0 + 1), d32 = bcast_lane(aJ, 48 + b0 + 2), d33 = bcast_lane(aJ, 48 + b0 + 3);
    double i0, i1, i2, i3, sq;
    if (!(d00 > 0.0)) bad = true;
    rsqrt_sqrt_k(d00, i0, sq);
    const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
    const double p1 = fma(-l10, l10, d11);
    if (!(p1 > 0.0)) bad = true;
    rsqrt_sqrt_k(p1, i1, sq);
    const double l21 = fma(-l20, l10, d21) * i1, l31 = fma(-l30, l10, d31) * i1;
    const double p2 = fma(-l21, l21, fma(-l20, l20, d22));
    if (!(p2 > 0.0)) bad = true;
    rsqrt_sqrt_k(p2, i2, sq);
    const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * i2;
    const double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
    if (!(p3 > 0.0)) bad = true;
    rsqrt_sqrt_k(p3, i3, sq);
    // M = Ld^-1 (lower triangular)
    const double m10 = -(l10 * i0) * i1;
    const double m21 = -(l21 * i1) * i2;
    const double m32 = -(l32 * i2) * i3;
    const double m20 = -fma(l21, m10, l20 * i0) * i2;
    const double m31 = -fma(l32, m21, l31 * i1) * i3;
    const double m30 = -fma(l32, m20, fma(l31, m10, l30 * i0)) * i3;
    // A operand of Z = Mpad * rows: lane (i = c, k = g) holds M[i][k] for i < 4, k <= i
    double sel = 0.0;
    sel = (c == 0 && g == 0) ? i0 : sel;
    sel = (c == 1) ? (g == 0 ? m10 : (g == 1 ? i1 : 0.0)) : sel;
    sel = (c == 2) ? (g == 0 ? m20 : (g == 1 ? m21 : (g == 2 ? i2 : 0.0))) : sel;
    sel = (c == 3) ? (g == 0 ? m30 : (g == 1 ? m31 : (g == 2 ? m32 : i3))) : sel;
    const v4d zw = __builtin_amdgcn_mfma_f64_16x16x4f64(sel, w[J], zero, 0, 0, 0);
    if (J < 3) {
      const v4d z = __builtin_amdgcn_mfma_f64_16x16x4f64(sel, aJ, zero, 0, 0, 0);
      const double lp = z[0];                       // L[c][4J+g]
      const double nlp = -lp;
      a = __builtin_amdgcn_mfma_f64_16x16x4f64(nlp, lp, a, 0, 0, 0);
      const double below = (c >= 4 * J + 4) ? nlp : 0.0;
      w = __builtin_amdgcn_mfma_f64_16x16x4f64(below, zw[0], w, 0, 0, 0);
    }
    w[J] = zw[0];
  }
  return !bad;
}


template <int NEW>
__global__ void kb(const double* A, double* out, int n, long long* t, double* Wout) {
  const int l = threadIdx.x;
  v4d a0;
  for (int q = 0; q < 4; q++) a0[q] = A[((l >> 4) + 4 * q) * 16 + (l & 15)];
  v4d w, acc = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
    v4d a = a0;
    a[0] += 1e-9 * i;
    if (NEW) chol_inv_blocked(a, w); else chol_inv_blocked_r03(a, w);
    acc += w + a;
  }
  long long t1 = clock64();
  for (int q = 0; q < 4; q++) out[l * 4 + q] = acc[q];
  { v4d a = a0; if (NEW) chol_inv_blocked(a, w); else chol_inv_blocked_r03(a, w); for (int q = 0; q < 4; q++) Wout[((l >> 4) + 4 * q) * 16 + (l & 15)] = w[q]; }
  if (l == 0) t[3 + NEW] = (t1 - t0) / n;
}
// r06, the look-ahead question: how much of the tile Cholesky is a chain that WAITS for its own results (and could be hidden behind independent
// work of the same wave) and how much is instructions that have to issue anyway?  Two independent tiles per iteration, interleaved by the
// compiler: cycles per PAIR against cycles per single tile.  pair == 2 x single: issue-bound, nothing to hide; pair == single: all latency.
__global__ void kb_pair(const double* A, double* out, int n, long long* t) {
  const int l = threadIdx.x;
  v4d a0;
  for (int q = 0; q < 4; q++) a0[q] = A[((l >> 4) + 4 * q) * 16 + (l & 15)];
  v4d w, w2, acc = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
    v4d a = a0, a2 = a0;
    a[0] += 1e-9 * i;
    a2[0] += 2e-9 * i;
    chol_inv_blocked(a, w);
    chol_inv_blocked(a2, w2);
    acc += w + a + w2 + a2;
  }
  long long t1 = clock64();
  for (int q = 0; q < 4; q++) out[l * 4 + q] = acc[q];
  if (l == 0) t[5] = (t1 - t0) / n;
}
template <int VARIANT>
__global__ void k(const double* A, double* out, int n, long long* t) {
  const int l = threadIdx.x;
  v4d a0;
  for (int q = 0; q < 4; q++) a0[q] = A[((l >> 4) + 4 * q) * 16 + (l & 15)];
  v4d w, acc = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
    v4d a = a0;
    a[0] += 1e-9 * i;
    chol_inv_mfma<VARIANT>(a, w);
    acc += w + a;
  }
  long long t1 = clock64();
  for (int q = 0; q < 4; q++) out[l * 4 + q] = acc[q];
  if (l == 0) t[VARIANT] = (t1 - t0) / n;
}
int main() {
  double hA[256];
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) hA[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
  double *dA, *out; long long* t; long long h[6]; double* dW; double hW[256];
  hipMalloc(&dA, 2048); hipMalloc(&out, 4096); hipMalloc(&t, 64); hipMalloc(&dW, 2048);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice);
  k<0><<<1, 64>>>(dA, out, 2000, t); k<1><<<1, 64>>>(dA, out, 2000, t); k<2><<<1, 64>>>(dA, out, 2000, t);
  auto check = [&]() {
    hipMemcpy(hW, dW, 2048, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
      double s = 0;
      for (int p = 0; p < 16; p++) for (int q = 0; q < 16; q++) s += hW[i * 16 + p] * hA[p * 16 + q] * hW[j * 16 + q];
      const double e = s - (i == j ? 1.0 : 0.0);
      if (e * e > err) err = e * e;
    }
    return sqrt(err);
  };
  kb<0><<<1, 64>>>(dA, out, 2000, t, dW);
  hipDeviceSynchronize();
  const double e_old = check();
  kb<1><<<1, 64>>>(dA, out, 2000, t, dW);
  hipDeviceSynchronize();
  const double e_new = check();
  kb_pair<<<1, 64>>>(dA, out, 2000, t);
  hipDeviceSynchronize();
  hipMemcpy(h, t, 48, hipMemcpyDeviceToHost);
  printf("chol+inv (2 MFMA/step): %lld cycles per tile; chol only (1 MFMA/step): %lld; no MFMA (chain only): %lld\n", h[0], h[1], h[2]);
  printf("blocked chol+inv, r03 code:            %lld cycles per tile, max |W A W^T - I| = %.3e\n", h[3], e_old);
  printf("blocked chol+inv, product (tile_chol.h): %lld cycles per tile, max |W A W^T - I| = %.3e\n", h[4], e_new);
  printf("look-ahead bound (r06): TWO independent tiles interleaved by the compiler: %lld cycles per pair = %.2f x one tile (2.00 = issue-bound: nothing to hide "
         "behind other work of the same wave; 1.00 = all of it the latency of dependent operations)\n", h[5], (double)h[5] / (double)h[4]);
  return 0;
}
