// Probe for the 7-row camera border of the one-wavefront factorisation (r06): v_mfma_f64_4x4x4_4b_f64 on gfx950
//  (1) operand / result layout, found by one-hot inputs (no layout is assumed) and compared with the hypothesis
//        A_b[i][k] lane i + 4 b + 16 k,  B_b[k][j] lane j + 4 b + 16 k,  D_b[i][j] lane j + 4 b + 16 i
//      under which register q of a 16 x 16 tile in accumulator order IS the A operand of the four row blocks (k = 4q..4q+3)
//  (2) issue rate of one wave: independent accumulators, a dependent chain, two interleaved chains; NEG bits
//  (3) does a VALU instruction issue in its shadow (same wave)?
//  (4) two waves on one SIMD, one streaming FP64 MFMAs (16x16x4), the other vector instructions (f64 fma / int add / DPP moves):
//      does the MFMA wave keep its 64 cycles per instruction?  (the look-ahead question: could ANOTHER wave run the tile Cholesky chain)
//  (5) the same 4x4x4 stream on every SIMD of the chip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_onehot(double* out) {   // block (la, lb): A = e_la, B = e_lb -> D per lane
  const int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[((size_t)la * 64 + lb) * 64 + lane] = d;
}
__global__ void k_neg4(double* out) {
  const int lane = threadIdx.x;
  const double a = 1.0 + 0.01 * lane, b = 2.0 - 0.003 * lane, c = 0.5 + lane;
  const double r0 = __builtin_amdgcn_mfma_f64_4x4x4f64(-a, b, c, 0, 0, 0);
  const double r1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1);
  const double r2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 2);
  const double r3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 4);
  const double r4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, -c, 0, 0, 0);
  out[lane] = fabs(r0 - r1); out[64 + lane] = fabs(r0 - r2); out[128 + lane] = fabs(r4 - r3);
}

// MODE 0: NT independent accumulators round robin; 1: one dependent chain; 2: NT accumulators + FILL vector instructions per MFMA
template <int NT, int MODE, int FILL>
__global__ __launch_bounds__(64, 1) void k_rate4(double* out, int n, long long* t) {
  double acc[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) acc[j] = 0.0;
  double x[4], y[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { x[q] = threadIdx.x * 1e-3 + q; y[q] = 1.0 + threadIdx.x * 1e-4 - q; }
  double f0 = 1.0 + threadIdx.x, f1 = 2.0;
  const double e = 1.0000001;
  int i0 = threadIdx.x, i1 = 1;
  const long long t0 = clock64();
  for (int it = 0; it < n; it++) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int j = 0; j < NT; j++) {
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[MODE == 1 ? 0 : j]) : "v"(x[q]), "v"(y[q]));
        if (FILL == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f0) : "v"(e));
        if (FILL == 2) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f0) : "v"(e)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f1) : "v"(e)); }
        if (FILL == 3) { asm volatile("v_add_u32 %0, %0, 1" : "+v"(i0)); asm volatile("v_add_u32 %0, %0, 1" : "+v"(i1)); }
      }
  }
  const long long t1 = clock64();
  double s = f0 + f1 + (double)(i0 + i1);
#pragma unroll
  for (int j = 0; j < NT; j++) s += acc[j];
  out[threadIdx.x + 64 * (blockIdx.x & 1023)] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

// a border-shaped mix: per "tile" 4 x 16x16x4 (window) then 8 x 4x4x4 (border), the way a factor step would interleave them
__global__ __launch_bounds__(64, 1) void k_mix(double* out, int n, long long* t) {
  v4d big[8];
  double sm[16];
#pragma unroll
  for (int j = 0; j < 8; j++) big[j] = (v4d){0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 16; j++) sm[j] = 0.0;
  double x[4], y[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { x[q] = threadIdx.x * 1e-3 + q; y[q] = 1.0 + threadIdx.x * 1e-4 - q; }
  const long long t0 = clock64();
  for (int it = 0; it < n; it++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
      for (int q = 0; q < 4; q++) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(big[j]) : "v"(x[q]), "v"(y[q]));
#pragma unroll
      for (int q = 0; q < 4; q++) {
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(sm[2 * j]) : "v"(x[q]), "v"(y[q]));
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(sm[2 * j + 1]) : "v"(x[q]), "v"(y[q]));
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) s += big[j][j & 3];
#pragma unroll
  for (int j = 0; j < 16; j++) s += sm[j];
  out[threadIdx.x + 64 * (blockIdx.x & 1023)] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

// (4) eight waves per workgroup = two per SIMD.  Waves 0..3 stream 16x16x4 MFMAs; waves 4..7 do OTHER: 0 idle (exit), 1 f64 fma chain x 4
// independent, 2 int adds, 3 DPP moves, 4 MFMAs as well.  t[0..3]: cycles of the MFMA waves, t[4..7]: of the others; cnt: their work.
template <int OTHER>
__global__ __launch_bounds__(512, 1) void k_pair(double* out, int n, long long* t) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double x[4], y[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { x[q] = lane * 1e-3 + q; y[q] = 1.0 + lane * 1e-4 - q; }
  double s = 0;
  const long long t0 = clock64();
  if (w < 4 || OTHER == 4) {
    v4d acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = (v4d){0, 0, 0, 0};
    for (int it = 0; it < n; it++) {
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x[q]), "v"(y[q]));
    }
#pragma unroll
    for (int j = 0; j < 8; j++) s += acc[j][j & 3];
  } else if (OTHER != 0) {
    double f0 = 1.0 + lane, f1 = 2.0, f2 = 3.0, f3 = 4.0;
    const double e = 1.0000001;
    int i0 = lane, i1 = 1, i2 = 2, i3 = 3;
    for (int it = 0; it < n; it++) {
#pragma unroll
      for (int r = 0; r < 32; r++) {
        if (OTHER == 1) {
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f0) : "v"(e)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f1) : "v"(e));
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f2) : "v"(e)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f3) : "v"(e));
        } else if (OTHER == 2) {
          asm volatile("v_add_u32 %0, %0, 1" : "+v"(i0)); asm volatile("v_add_u32 %0, %0, 1" : "+v"(i1));
          asm volatile("v_add_u32 %0, %0, 1" : "+v"(i2)); asm volatile("v_add_u32 %0, %0, 1" : "+v"(i3));
        } else {
          asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(i0)); asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(i1));
          asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(i2)); asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(i3));
        }
      }
    }
    s = f0 + f1 + f2 + f3 + (double)(i0 + i1 + i2 + i3);
  }
  const long long t1 = clock64();
  out[threadIdx.x] = s;
  if (lane == 0) t[w] = t1 - t0;
}

// (6) is a DEPENDENT chain of 4x4x4 MFMAs (accumulator = the result of the instruction right in front of it) interlocked by the hardware?
// Integer-valued data (exact in FP64): NOPS wait states between the instructions; -1 = the two-accumulator alternation the factor kernel uses
// for its border tiles.  A busy neighbour wave (LDS traffic of the same workgroup's other waves) varies the timing.
template <int NOPS>
__global__ __launch_bounds__(256) void k_chain(double* out, int n) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ double sh[4 * 64];
  double acc = 0.0, acc2 = 0.0;
  double a = (double)((lane * 7 + 3) % 5 - 2), b = (double)((lane * 5 + 1) % 7 - 3);
  for (int it = 0; it < n; it++) {
    if (w != 0) {   // neighbours: LDS + VALU noise
      sh[threadIdx.x] = a + it; __syncthreads(); a = sh[(threadIdx.x + 17) & 255] * 0.5 + 1.0; __syncthreads();
      continue;
    }
    __syncthreads(); __syncthreads();
    if (NOPS == -1) {
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %3, %2, %1\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %3, %2, %1\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %3, %2, %1\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %3, %2, %1\n\ts_nop 15"
                   : "+v"(acc), "+v"(acc2) : "v"(a), "v"(b));
    } else if (NOPS == 0) {
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %0\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %0\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    } else if (NOPS == 1) {
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 0\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %0\n\ts_nop 0\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 0\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %0\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    } else if (NOPS == 4) {
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 3\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %0\n\ts_nop 3\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 3\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %0\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    } else {
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 15\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %0\n\ts_nop 15\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 15\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %0\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    }
  }
  if (w == 0) out[(size_t)blockIdx.x * 64 + lane] = acc + acc2;
}
static void chain_check(double* out) {
  const int blocks = 2048, n = 300;
  static double ref[2048 * 64], h[2048 * 64];
  k_chain<16><<<blocks, 256>>>(out, n); hipDeviceSynchronize();
  hipMemcpy(ref, out, sizeof(ref), hipMemcpyDeviceToHost);
  auto cmp = [&](const char* name) {
    hipDeviceSynchronize();
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0, badb = 0;
    for (int b = 0; b < blocks; b++) { int bb = 0; for (int l = 0; l < 64; l++) if (h[b * 64 + l] != ref[b * 64 + l]) { bad++; bb = 1; } badb += bb; }
    printf("dependent 4x4x4 chain, %-34s: %d of %d lanes differ from the chain with 16 wait states (%d of %d waves)\n", name, bad, blocks * 64, badb, blocks);
  };
  for (int rep = 0; rep < 3; rep++) {
    k_chain<0><<<blocks, 256>>>(out, n); cmp("back to back (0 wait states)");
    k_chain<1><<<blocks, 256>>>(out, n); cmp("s_nop 0 between (1 wait state)");
    k_chain<4><<<blocks, 256>>>(out, n); cmp("s_nop 3 between (4 wait states)");
  }
  // the alternation: compare against itself with the reference's operand order (acc + acc2 of the alternating chains = a different sum: just run-to-run stability)
  k_chain<-1><<<blocks, 256>>>(out, n); hipDeviceSynchronize();
  hipMemcpy(ref, out, sizeof(ref), hipMemcpyDeviceToHost);
  for (int rep = 0; rep < 3; rep++) { k_chain<-1><<<blocks, 256>>>(out, n); cmp("two alternating accumulators (self)"); }
}

template <class K>
static void run(const char* name, K kern, int blocks, int n, double per_it, double flops_per, double* out, long long* t) {
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
  const double nm = per_it * n;
  printf("%-58s blocks %4d: %.1f cycles per instruction (mean wave), %.1f (slowest); wall %.3f ms -> %.2f TFLOP/s, eff. clock %.2f GHz\n", name, blocks,
         mean / nm, (double)mx / nm, ms, flops_per * nm * blocks / (ms * 1e-3) / 1e12, (double)mx / (ms * 1e-3) / 1e9);
}
template <class K>
static void run_pair(const char* name, K kern, int n, double* out, long long* t) {
  kern<<<1, 512>>>(out, 10, t); hipDeviceSynchronize();
  kern<<<1, 512>>>(out, n, t); hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-40s MFMA waves: %.1f %.1f %.1f %.1f cycles per MFMA; other waves: %.1f %.1f %.1f %.1f cycles per instruction\n", name, h[0] / (32.0 * n), h[1] / (32.0 * n),
         h[2] / (32.0 * n), h[3] / (32.0 * n), h[4] / (128.0 * n), h[5] / (128.0 * n), h[6] / (128.0 * n), h[7] / (128.0 * n));
}

int main() {
  double* out; long long* t;
  hipMalloc(&out, 64 * 64 * 64 * 8); hipMalloc(&t, 8 * 4096);
  // (1) layout
  k_onehot<<<dim3(64, 64), 64>>>(out); hipDeviceSynchronize();
  static double h[64 * 64 * 64];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0, nz = 0;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      const int ia = la & 3, ba = (la >> 2) & 3, ka = la >> 4;
      const int jb = lb & 3, bb = (lb >> 2) & 3, kb = lb >> 4;
      const int expect_lane = (ba == bb && ka == kb) ? (jb + 4 * ba + 16 * ia) : -1;
      for (int l = 0; l < 64; l++) {
        const double v = h[((size_t)la * 64 + lb) * 64 + l];
        const double ex = (l == expect_lane) ? 1.0 : 0.0;
        if (v != 0.0) nz++;
        if (v != ex) { if (bad < 10) printf("  layout mismatch: A lane %d, B lane %d -> D lane %d = %g (hypothesis: lane %d)\n", la, lb, l, v, expect_lane); bad++; }
      }
    }
  printf("v_mfma_f64_4x4x4_4b_f64 layout, hypothesis A i+4b+16k / B j+4b+16k / D j+4b+16i: %s (%d non-zero results of 4096 pairs, %d mismatches)\n", bad ? "WRONG" : "CONFIRMED", nz, bad);
  if (bad) {   // print the actual map for the first block of rows
    for (int la = 0; la < 64; la += 1)
      for (int lb = 0; lb < 64; lb++)
        for (int l = 0; l < 64; l++)
          if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0 && la < 8) printf("  A lane %2d x B lane %2d -> D lane %2d\n", la, lb, l);
  }
  k_neg4<<<1, 64>>>(out); hipDeviceSynchronize();
  double hn[192]; hipMemcpy(hn, out, sizeof(hn), hipMemcpyDeviceToHost);
  double d1 = 0, d2 = 0, d3 = 0;
  for (int i = 0; i < 64; i++) { d1 = fmax(d1, hn[i]); d2 = fmax(d2, hn[64 + i]); d3 = fmax(d3, hn[128 + i]); }
  printf("neg bits of v_mfma_f64_4x4x4 (blgp): negA max diff %.3g, negB %.3g, negC %.3g (0 = the bit negates that operand)\n", d1, d2, d3);
  chain_check(out);
  // (2), (3)
  const int n = 2000;
  run("4x4x4: 16 independent accumulators", k_rate4<16, 0, 0>, 1, n, 64, 512, out, t);
  run("4x4x4: 2 interleaved chains", k_rate4<2, 0, 0>, 1, n, 8, 512, out, t);
  run("4x4x4: one dependent chain", k_rate4<16, 1, 0>, 1, n, 64, 512, out, t);
  run("4x4x4: 16 accumulators + 1 f64 fma per MFMA", k_rate4<16, 2, 1>, 1, n, 64, 512, out, t);
  run("4x4x4: 16 accumulators + 2 f64 fma per MFMA", k_rate4<16, 2, 2>, 1, n, 64, 512, out, t);
  run("4x4x4: 16 accumulators + 2 int add per MFMA", k_rate4<16, 2, 3>, 1, n, 64, 512, out, t);
  run("mix: per tile 4 x 16x16x4 + 8 x 4x4x4 (96 instr / it)", k_mix, 1, n, 96, (32.0 * 2048 + 64.0 * 512) / 96.0, out, t);
  // (5)
  run("4x4x4: 16 independent accumulators, every SIMD", k_rate4<16, 0, 0>, 1024, n, 64, 512, out, t);
  run("mix, every SIMD", k_mix, 1024, n, 96, (32.0 * 2048 + 64.0 * 512) / 96.0, out, t);
  // (4)
  run_pair("pair: MFMA waves alone", k_pair<0>, 400, out, t);
  run_pair("pair: + f64 fma waves", k_pair<1>, 400, out, t);
  run_pair("pair: + int add waves", k_pair<2>, 400, out, t);
  run_pair("pair: + DPP mov waves", k_pair<3>, 400, out, t);
  run_pair("pair: + MFMA waves (2 per SIMD)", k_pair<4>, 400, out, t);
  return 0;
}
