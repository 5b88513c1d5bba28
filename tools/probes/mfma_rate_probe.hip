// Probe: issue rate / dependent latency of v_mfma_f64_16x16x4_f64 and the rsqrt+Goldschmidt chain on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k_indep(double* out, int n, long long* t) {
  v4d a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[threadIdx.x + blockDim.x * blockIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_dep(double* out, int n, long long* t) {
  v4d a0 = {0,0,0,0};
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    x = a0[0] * 1e-9;   // VALU consumer of the result feeding the next MFMA
  }
  long long t1 = clock64();
  out[threadIdx.x + blockDim.x * blockIdx.x] = a0[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = t1 - t0;
}
__global__ void k_rsq(double* out, int n, long long* t) {
  double d = 2.0 + threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
    double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double res = fma(-g, g, d);
    g = fma(res, h, g);
    d = g + (h + h) + 1.5;
  }
  long long t1 = clock64();
  out[threadIdx.x] = d;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = t1 - t0;
}
__global__ void k_fma(double* out, int n, long long* t) {
  double d = 2.0 + threadIdx.x, e = 1.000001;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) { d = fma(d, e, 1e-9); d = fma(d, e, 1e-9); d = fma(d, e, 1e-9); d = fma(d, e, 1e-9); }
  long long t1 = clock64();
  out[threadIdx.x] = d;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[3] = t1 - t0;
}
int main() {
  double* out; long long* t; long long h[4];
  hipMalloc(&out, 1 << 20); hipMalloc(&t, 64);
  const int n = 2000;
  for (int waves = 1; waves <= 8; waves *= 2) {
    k_indep<<<1, 64 * waves>>>(out, n, t); hipDeviceSynchronize();
    hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
    printf("indep mfma f64 16x16x4, %d wave(s)/WG: %.1f cycles per MFMA per wave\n", waves, (double)h[0] / (4.0 * n));
  }
  k_dep<<<1, 64>>>(out, n, t); k_rsq<<<1, 64>>>(out, n, t); k_fma<<<1, 64>>>(out, n, t); hipDeviceSynchronize();
  hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
  printf("dependent mfma+valu chain: %.1f cycles/iter\nrsqrt+goldschmidt chain: %.1f cycles/iter\ndependent f64 fma: %.1f cycles\n", (double)h[1] / n, (double)h[2] / n, (double)h[3] / (4.0 * n));
  return 0;
}
