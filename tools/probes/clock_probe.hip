// Probe: shader clock (s_memtime) vs constant 100 MHz counter (s_memrealtime) for a 1-workgroup kernel on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, int n, long long* t) {
  double d = 2.0 + threadIdx.x, e = 1.000001;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; i++) { d = fma(d, e, 1e-9); d = fma(d, e, 1e-9); d = fma(d, e, 1e-9); d = fma(d, e, 1e-9); }
  long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = d;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
  double* out; long long* t; long long h[2];
  hipMalloc(&out, 1 << 24); hipMalloc(&t, 64);
  for (int blocks : {1, 256, 1024}) {
    for (int rep = 0; rep < 2; rep++) { k<<<blocks, 512>>>(out, 200000, t); hipDeviceSynchronize(); }
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("blocks %4d: %lld shader cycles, %lld ticks@100MHz -> %.0f MHz; %.2f cycles per dependent f64 fma\n", blocks, h[0], h[1], 100.0 * h[0] / h[1], h[0] / 800000.0);
  }
  return 0;
}
