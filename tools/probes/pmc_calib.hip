// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 with this repo's access width (8 B per lane):
// copies N doubles (N*8 bytes read, N*8 bytes written), far larger than the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_copy8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] * 1.0000001;
}
int main() {
  const size_t n = 1ull << 28;  // 2 GiB in, 2 GiB out
  double *a, *b;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemset(a, 0, n * 8);
  for (int r = 0; r < 3; r++) calib_copy8<<<4096, 256>>>(a, b, n);
  hipDeviceSynchronize();
  printf("calib_copy8: %zu bytes read and %zu bytes written per launch\n", n * 8, n * 8);
  return 0;
}
