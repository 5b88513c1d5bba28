// Probe: what HBM bandwidth does this box deliver for (a) a plain coalesced copy and (b) the SfT solver's access shape --
// every workgroup streams its own 1.7 MB region in 2 KB tiles (one wavefront per tile, 32 B per lane), reading one
// region and writing another, 256 / 512 workgroups at a time like the resident problems of a batched launch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void copy8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] * 1.0000001;
}
// region_tiles tiles of 256 doubles per workgroup and pass; wave w of the workgroup handles tiles w, w+nw, ...
__global__ void tiles(const double* __restrict__ in, double* __restrict__ out, int region_tiles, int passes, int nregions) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int p = 0; p < passes; p++) {
    const size_t reg = ((size_t)blockIdx.x + (size_t)p * gridDim.x) % nregions;
    const double* src = in + reg * region_tiles * 256;
    double* dst = out + reg * region_tiles * 256;
    for (int t = wave; t < region_tiles; t += nw) {
      v4d v = *reinterpret_cast<const v4d*>(src + (size_t)t * 256 + 4 * lane);
      v[0] += 1.0;
      *reinterpret_cast<v4d*>(dst + (size_t)t * 256 + 4 * lane) = v;
    }
  }
}
int main() {
  const size_t n = 1ull << 28;
  double *a, *b;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemset(a, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  copy8<<<4096, 256>>>(a, b, n); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 3; r++) copy8<<<4096, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("coalesced copy (8 B/lane): %.2f TB/s read+write\n", 3.0 * 2.0 * n * 8 / (ms * 1e-3) / 1e12);
  const int region_tiles = 846;                       // 94 x 9 tiles = 1.73 MB, like one H
  const int nregions = (int)(n / ((size_t)region_tiles * 256));
  for (int wgs : {256, 512, 1024, 2048}) {
    for (int threads : {256, 512}) {
      const int passes = 16;
      tiles<<<wgs, threads>>>(a, b, region_tiles, 2, nregions); hipDeviceSynchronize();
      hipEventRecord(e0); tiles<<<wgs, threads>>>(a, b, region_tiles, passes, nregions); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 2.0 * (double)wgs * passes * region_tiles * 2048.0;
      printf("tile stream: %4d workgroups x %3d threads: %.2f TB/s read+write\n", wgs, threads, bytes / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
