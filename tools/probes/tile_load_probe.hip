// Per-CU tile load probe: one workgroup, W active wavefronts, each streams 2 KB tiles (32 B per lane, two dwordx4) with
// U tiles in flight per batch (straight-line), tile stride S bytes, over a region of R bytes.  Prints cycles per tile
// per wave: L2 / HBM latency and the bandwidth a single CU reaches with few waves.
// build: hipcc -O3 --offload-arch=gfx950 -o tile_load_probe tile_load_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(512) void probe(const double* __restrict__ buf, size_t region_tiles, size_t stride_tiles, int waves, int iters, long long* out, double* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= waves) return;
  v4d acc = {0, 0, 0, 0};
  size_t t = (size_t)wave * 7919 % region_tiles;
  // warm pass (brings the region into L2 when it fits)
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < U; u++) { acc += *reinterpret_cast<const v4d*>(buf + t * 256 + 4 * lane); t = (t + stride_tiles) % region_tiles; }
  }
  __syncthreads();
  t = (size_t)wave * 7919 % region_tiles;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    v4d b[U];
#pragma unroll
    for (int u = 0; u < U; u++) { b[u] = *reinterpret_cast<const v4d*>(buf + t * 256 + 4 * lane); t = (t + stride_tiles) % region_tiles; }
#pragma unroll
    for (int u = 0; u < U; u++) acc += b[u];
  }
  const long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  if (acc[0] == 123.456) sink[0] = acc[1];
}

int main() {
  const size_t bytes = (size_t)256 << 20;
  double* buf; long long* out; double* sink;
  hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&out, 64); hipMalloc(&sink, 8);
  const int iters = 64;
  struct Cfg { size_t region_kb, stride_tiles; int waves; };
  const std::vector<Cfg> cfgs = {{512, 1, 1}, {512, 16, 1}, {512, 16, 8}, {512, 1, 8}, {2048, 16, 8}, {2048, 17, 8}, {262144, 16, 1}, {262144, 16, 8}, {262144, 1029, 8}};
  for (const Cfg& c : cfgs) {
    for (int U : {1, 4, 8, 16}) {
      const size_t rt = c.region_kb / 2;
      if (U == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(512), 0, 0, buf, rt, c.stride_tiles, c.waves, iters * 16, out, sink);
      if (U == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(512), 0, 0, buf, rt, c.stride_tiles, c.waves, iters * 4, out, sink);
      if (U == 8) hipLaunchKernelGGL(probe<8>, dim3(1), dim3(512), 0, 0, buf, rt, c.stride_tiles, c.waves, iters * 2, out, sink);
      if (U == 16) hipLaunchKernelGGL(probe<16>, dim3(1), dim3(512), 0, 0, buf, rt, c.stride_tiles, c.waves, iters, out, sink);
      hipDeviceSynchronize();
      long long h[8];
      hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
      double mx = 0;
      for (int w = 0; w < c.waves; w++) mx = h[w] > mx ? h[w] : mx;
      printf("region %6zu KB stride %4zu tiles waves %d in-flight %2d: %7.0f cycles per tile per wave  (%.1f B/clk for the CU)\n", c.region_kb, c.stride_tiles, c.waves, U,
             mx / (iters * 16.0), c.waves * 2048.0 / (mx / (iters * 16.0)));
    }
  }
  return 0;
}
