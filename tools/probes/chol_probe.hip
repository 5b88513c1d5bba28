// Probe: cycles of the rank-1-MFMA 16x16 Cholesky(+inverse) used by factor_tiles, one wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
#define TS 16
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void rsqrt_sqrt(double d, double& inv, double& s) {
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
    rsqrt_sqrt(pd, inv, sq);
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
  double *dA, *out; long long* t; long long h[3];
  hipMalloc(&dA, 2048); hipMalloc(&out, 4096); hipMalloc(&t, 64);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice);
  k<0><<<1, 64>>>(dA, out, 2000, t); k<1><<<1, 64>>>(dA, out, 2000, t); k<2><<<1, 64>>>(dA, out, 2000, t);
  hipDeviceSynchronize();
  hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
  printf("chol+inv (2 MFMA/step): %lld cycles per tile; chol only (1 MFMA/step): %lld; no MFMA (chain only): %lld\n", h[0], h[1], h[2]);
  return 0;
}
