// Probe for the one-wavefront-per-problem factorisation (512 registers, one wave per SIMD):
//  (1) v_mfma_f64_16x16x4_f64 issue rate of ONE wave with 36 independent accumulators, four dependent k-chunks per tile
//  (2) does VALU work (f64 fma / 32-bit integer) issued between the MFMAs of the same wave hide behind them?
//  (3) the NEG bits (blgp) of the f64 MFMA: (-A) B + C without a VALU negation
//  (4) the same stream on every SIMD of the chip (1024 one-wave workgroups): cycles per MFMA and the wall clock -> sustained FP64 TFLOP/s
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NT, int FILL>   // FILL: 0 none, 1 = 4 independent f64 fma per MFMA, 2 = 8 int adds per MFMA, 3 = 8 f64 fma per MFMA
__global__ __launch_bounds__(64, 1) void k_stream(double* out, int n, long long* t) {
  v4d acc[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) acc[j] = (v4d){0, 0, 0, 0};
  double x[4], y[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { x[q] = threadIdx.x * 1e-3 + q; y[q] = 1.0 + threadIdx.x * 1e-4 - q; }
  double f0 = 1.0 + threadIdx.x, f1 = 2.0, f2 = 3.0, f3 = 4.0, f4 = 5.0, f5 = 6.0, f6 = 7.0, f7 = 8.0;
  const double e = 1.0000001;
  int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
  const long long t0 = clock64();
  for (int it = 0; it < n; it++) {
#pragma unroll
    for (int j = 0; j < NT; j++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[q], y[q], acc[j], 0, 0, 0);
        if (FILL == 1 || FILL == 3) {
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f0) : "v"(e));
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f1) : "v"(e));
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f2) : "v"(e));
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f3) : "v"(e));
        }
        if (FILL == 3) {
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f4) : "v"(e));
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f5) : "v"(e));
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f6) : "v"(e));
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f7) : "v"(e));
        }
        if (FILL == 2) {
          asm volatile("v_add_u32 %0, %0, 1" : "+v"(i0)); asm volatile("v_add_u32 %0, %0, 1" : "+v"(i1));
          asm volatile("v_add_u32 %0, %0, 1" : "+v"(i2)); asm volatile("v_add_u32 %0, %0, 1" : "+v"(i3));
          asm volatile("v_add_u32 %0, %0, 1" : "+v"(i4)); asm volatile("v_add_u32 %0, %0, 1" : "+v"(i5));
          asm volatile("v_add_u32 %0, %0, 1" : "+v"(i6)); asm volatile("v_add_u32 %0, %0, 1" : "+v"(i7));
        }
      }
    }
  }
  const long long t1 = clock64();
  double s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
#pragma unroll
  for (int j = 0; j < NT; j++) s += acc[j][j & 3];
  out[threadIdx.x + 64 * (blockIdx.x & 1023)] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

// interleaved order: chunk q of every tile before chunk q+1 (the dependent MFMAs of a tile are NT issues apart)
template <int NT>
__global__ __launch_bounds__(64, 1) void k_stream_il(double* out, int n, long long* t) {
  v4d acc[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) acc[j] = (v4d){0, 0, 0, 0};
  double x[4], y[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { x[q] = threadIdx.x * 1e-3 + q; y[q] = 1.0 + threadIdx.x * 1e-4 - q; }
  const long long t0 = clock64();
  for (int it = 0; it < n; it++) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[q], y[q], acc[j], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int j = 0; j < NT; j++) s += acc[j][j & 3];
  out[threadIdx.x + 64 * (blockIdx.x & 1023)] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

__global__ void k_neg(double* out) {
  const int lane = threadIdx.x;
  const double a = 1.0 + 0.01 * lane, b = 2.0 - 0.003 * lane;
  v4d c = {0.5, 0.25, 0.125, 1.0};
  const v4d r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, b, c, 0, 0, 0);
  const v4d r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);   // blgp bit 0: negate A
  const v4d r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 2);   // blgp bit 1: negate B
  const v4d r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 4);   // blgp bit 2: negate C
  const v4d r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, -c, 0, 0, 0);
  double d1 = 0, d2 = 0, d3 = 0;
  for (int q = 0; q < 4; q++) { d1 = fmax(d1, fabs(r0[q] - r1[q])); d2 = fmax(d2, fabs(r0[q] - r2[q])); d3 = fmax(d3, fabs(r4[q] - r3[q])); }
  out[lane] = d1; out[64 + lane] = d2; out[128 + lane] = d3;
}

template <class K>
static void run(const char* name, K kern, int blocks, int n, int nt, double* out, long long* t, int fill_ops) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 64>>>(out, 10, t); hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, 64>>>(out, n, t);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  static long long h[4096];
  hipMemcpy(h, t, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0; long long mx = 0;
  for (int b = 0; b < blocks; b++) { mean += (double)h[b]; if (h[b] > mx) mx = h[b]; }
  mean /= blocks;
  const double nm = 4.0 * nt * n;
  printf("%-44s blocks %4d: %.1f cycles per MFMA (mean wave), %.1f (slowest); wall %.3f ms -> %.2f TFLOP/s, eff. clock %.2f GHz\n", name, blocks,
         mean / nm, (double)mx / nm, ms, 2048.0 * nm * blocks / (ms * 1e-3) / 1e12, (double)mx / (ms * 1e-3) / 1e9);
}

int main() {
  double* out; long long* t;
  hipMalloc(&out, 1 << 20); hipMalloc(&t, 8 * 4096);
  k_neg<<<1, 64>>>(out); hipDeviceSynchronize();
  double h[192]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double d1 = 0, d2 = 0, d3 = 0;
  for (int i = 0; i < 64; i++) { d1 = fmax(d1, h[i]); d2 = fmax(d2, h[64 + i]); d3 = fmax(d3, h[128 + i]); }
  printf("neg bits of v_mfma_f64 (blgp): negA max diff %.3g, negB %.3g, negC %.3g (0 = the bit negates that operand)\n", d1, d2, d3);
  const int n = 400;
  for (int blocks : {1, 256, 1024, 2048}) {
    run("36 tiles, 4 dependent chunks back to back", k_stream<36, 0>, blocks, n, 36, out, t, 0);
    run("36 tiles, chunks interleaved over tiles", k_stream_il<36>, blocks, n, 36, out, t, 0);
    run("36 tiles + 4 f64 fma per MFMA", k_stream<36, 1>, blocks, n, 36, out, t, 4);
    run("36 tiles + 8 f64 fma per MFMA", k_stream<36, 3>, blocks, n, 36, out, t, 8);
    run("36 tiles + 8 int add per MFMA", k_stream<36, 2>, blocks, n, 36, out, t, 8);
  }
  run("long run, interleaved, 1024 blocks", k_stream_il<36>, 1024, 20 * n, 36, out, t, 0);
  return 0;
}
