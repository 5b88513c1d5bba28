// Probe: v_mfma_f64_16x16x4_f64 accumulating into the SAME registers back to back (the four k-chunks of one tile product) against two and four
// independent accumulators interleaved, with 1, 2 (one per SIMD pair ...) up to 8 wavefronts per workgroup (4 SIMDs: 8 waves = 2 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_chain(double* out, int n, long long* t) {
  v4d a[NACC];
  for (int j = 0; j < NACC; j++) a[j] = (v4d){0, 0, 0, 0};
  const double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  const long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int j = 0; j < NACC; j++) a[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[j], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int j = 0; j < NACC; j++) s += a[j][j & 3];
  out[threadIdx.x + blockDim.x * blockIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { t[2 * (threadIdx.x >> 6)] = t0; t[2 * (threadIdx.x >> 6) + 1] = t1; }   // per wave: the arbiter favours the oldest one
}
int main() {
  double* out; long long* t; long long h, hh[64];
  hipMalloc(&out, 1 << 20); hipMalloc(&t, 512);
  const int n = 2000;
  for (int waves = 1; waves <= 16; waves *= 2) {
    k_chain<1><<<1, 64 * waves>>>(out, n, t); hipDeviceSynchronize(); hipMemcpy(hh, t, 16 * waves, hipMemcpyDeviceToHost); { long long lo = hh[0], hi = hh[1]; for (int w = 0; w < waves; w++) { if (hh[2 * w] < lo) lo = hh[2 * w]; if (hh[2 * w + 1] > hi) hi = hh[2 * w + 1]; } h = hi - lo; }
    const double c1 = (double)h / (4.0 * n);
    k_chain<2><<<1, 64 * waves>>>(out, n, t); hipDeviceSynchronize(); hipMemcpy(hh, t, 16 * waves, hipMemcpyDeviceToHost); { long long lo = hh[0], hi = hh[1]; for (int w = 0; w < waves; w++) { if (hh[2 * w] < lo) lo = hh[2 * w]; if (hh[2 * w + 1] > hi) hi = hh[2 * w + 1]; } h = hi - lo; }
    const double c2 = (double)h / (8.0 * n);
    k_chain<4><<<1, 64 * waves>>>(out, n, t); hipDeviceSynchronize(); hipMemcpy(hh, t, 16 * waves, hipMemcpyDeviceToHost); { long long lo = hh[0], hi = hh[1]; for (int w = 0; w < waves; w++) { if (hh[2 * w] < lo) lo = hh[2 * w]; if (hh[2 * w + 1] > hi) hi = hh[2 * w + 1]; } h = hi - lo; }
    const double c4 = (double)h / (16.0 * n);
    printf("%d wave(s): cycles per MFMA per wave (first start to last end) -- one accumulator chain %.1f, two interleaved %.1f, four interleaved %.1f\n", waves, c1, c2, c4);
  }
  return 0;
}
