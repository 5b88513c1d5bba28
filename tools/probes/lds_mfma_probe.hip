// Per-CU probe: W wavefronts each repeat { read a 2 KB tile from LDS (layout A: 32 B per lane contiguous; layout B: two
// 1 KB planes of 16 B per lane), optionally 4 x v_mfma_f64_16x16x4_f64 } and report cycles per item.
// build: hipcc -O3 --offload-arch=gfx950 -o lds_mfma_probe lds_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <int LAYOUT, int MFMA, int DEP>
__global__ __launch_bounds__(512) void probe(int waves, int iters, long long* out, double* sink) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 256; i += blockDim.x) lds[i] = 1e-3 * i;
  __syncthreads();
  if (wave >= waves) return;
  v4d s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const double* tile = lds + ((i + u + wave) & 15) * 256;
      v4d a;
      if (LAYOUT == 0) a = *reinterpret_cast<const v4d*>(tile + 4 * lane);
      else { const v2d lo = *reinterpret_cast<const v2d*>(tile + 2 * lane), hi = *reinterpret_cast<const v2d*>(tile + 128 + 2 * lane); a = (v4d){lo.x, lo.y, hi.x, hi.y}; }
      if (MFMA) {
        if (DEP) {
          s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[0], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], a[1], s1, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], a[2], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], a[3], s1, 0, 0, 0);
        } else {
          s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[0], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], a[1], s1, 0, 0, 0);
          s2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], a[2], s2, 0, 0, 0);
          s3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], a[3], s3, 0, 0, 0);
        }
      } else s0 += a;
    }
  }
  const long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  const v4d s = (s0 + s1) + (s2 + s3);
  if (s[0] == 123.456) sink[0] = s[1];
}


// Same item with the B operand streamed from global memory (L2-resident region, tiles 32 KB apart), chunks of 4 tiles double
// buffered in straight-line code -- the product loop of the wide tile factorisation.
__global__ __launch_bounds__(512) void probe_gl(const double* __restrict__ buf, int waves, int iters, long long* out, double* sink) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 256; i += blockDim.x) lds[i] = 1e-3 * i;
  __syncthreads();
  if (wave >= waves) return;
  v4d s0 = {0, 0, 0, 0}, s1 = s0;
  const double* base = buf + (size_t)wave * 16 * 4096 + 4 * lane;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    v4d b[2][4];
#pragma unroll
    for (int u = 0; u < 4; u++) b[0][u] = *reinterpret_cast<const v4d*>(base + (size_t)((i * 16 + u) & 63) * 4096);
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c < 3) {
#pragma unroll
        for (int u = 0; u < 4; u++) b[(c + 1) & 1][u] = *reinterpret_cast<const v4d*>(base + (size_t)((i * 16 + 4 * (c + 1) + u) & 63) * 4096);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double* tile = lds + ((4 * c + u + wave) & 15) * 256;
        const v2d lo = *reinterpret_cast<const v2d*>(tile + 2 * lane), hi = *reinterpret_cast<const v2d*>(tile + 128 + 2 * lane);
        const v4d bb = b[c & 1][u];
        s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo.x, bb[0], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo.y, bb[1], s1, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi.x, bb[2], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi.y, bb[3], s1, 0, 0, 0);
      }
    }
  }
  const long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  const v4d s = s0 + s1;
  if (s[0] == 123.456) sink[0] = s[1];
}

template <int L, int M, int D> void run(const char* name, long long* out, double* sink) {
  for (int waves : {1, 2, 4, 8}) {
    const int iters = 256;
    hipLaunchKernelGGL((probe<L, M, D>), dim3(1), dim3(512), 16 * 2048, 0, waves, iters, out, sink);
    (void)hipDeviceSynchronize();
    long long h[8];
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
    printf("%-46s waves %d: %6.0f cycles per item\n", name, waves, (double)mx / (iters * 8.0));
  }
}

int main() {
  long long* out; double* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 8);
  run<0, 0, 0>("LDS 32 B per lane, no MFMA", out, sink);
  run<1, 0, 0>("LDS two planes, no MFMA", out, sink);
  run<0, 1, 1>("LDS 32 B per lane + 4 MFMA (2 chains)", out, sink);
  run<1, 1, 1>("LDS two planes + 4 MFMA (2 chains)", out, sink);
  run<1, 1, 0>("LDS two planes + 4 MFMA (4 chains)", out, sink);
  double* buf;
  (void)hipMalloc(&buf, (size_t)8 * 16 * 4096 * 8 * 4);
  (void)hipMemset(buf, 0, (size_t)8 * 16 * 4096 * 8 * 4);
  for (int waves : {1, 2, 4, 8}) {
    const int iters = 64;
    hipLaunchKernelGGL(probe_gl, dim3(1), dim3(512), 16 * 2048, 0, buf, waves, iters, out, sink);
    (void)hipDeviceSynchronize();
    long long h[8];
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
    printf("%-46s waves %d: %6.0f cycles per item\n", "LDS A + global B (4-tile chunks) + 4 MFMA", waves, (double)mx / (iters * 16.0));
  }
  return 0;
}
