// Probe: window tiles pinned in accumulator registers by hand (inline asm, a[8T:8T+7]), operands in compiler-managed VGPRs.
//  (1) cycles per v_mfma_f64_16x16x4_f64 of ONE wave streaming over 32 AGPR tiles, operands from 9 VGPR tiles, NEG bit on A
//  (2) what issues in the shadow of an f64 MFMA from the same wave: ds_read_b64, ds_write_b64, global_load, SALU, v_accvgpr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

#define MF(T, a, b) asm volatile("v_mfma_f64_16x16x4_f64 a[%c0:%c1], %2, %3, a[%c0:%c1] neg:[1,0,0]" :: "i"(8 * (T)), "i"(8 * (T) + 7), "v"(a), "v"(b))

template <int T>
__device__ __forceinline__ void tile_update(const v4d& A, const v4d& B) {
  MF(T, A[0], B[0]); MF(T, A[1], B[1]); MF(T, A[2], B[2]); MF(T, A[3], B[3]);
}
template <int T>
__device__ __forceinline__ void tile_zero() {
#pragma unroll
  for (int r = 0; r < 8; r++) asm volatile("v_accvgpr_write_b32 a[%c0], 0" :: "i"(8 * T + r));
}
template <int T>
__device__ __forceinline__ v4d tile_read() {
  v4d v;
  asm volatile("s_nop 15\n\ts_nop 3\n\t"
               "v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c9]\n\tv_accvgpr_read_b32 %2, a[%c10]\n\tv_accvgpr_read_b32 %3, a[%c11]\n\t"
               "v_accvgpr_read_b32 %4, a[%c12]\n\tv_accvgpr_read_b32 %5, a[%c13]\n\tv_accvgpr_read_b32 %6, a[%c14]\n\tv_accvgpr_read_b32 %7, a[%c15]"
               : "=v"(((int*)&v)[0]), "=v"(((int*)&v)[1]), "=v"(((int*)&v)[2]), "=v"(((int*)&v)[3]), "=v"(((int*)&v)[4]), "=v"(((int*)&v)[5]),
                 "=v"(((int*)&v)[6]), "=v"(((int*)&v)[7])
               : "i"(8 * T), "i"(8 * T + 1), "i"(8 * T + 2), "i"(8 * T + 3), "i"(8 * T + 4), "i"(8 * T + 5), "i"(8 * T + 6), "i"(8 * T + 7));
  return v;
}

template <int T, int N> struct ForTiles {
  template <class F> static __device__ __forceinline__ void run(F&& f) { f.template operator()<T>(); ForTiles<T + 1, N>::run(f); }
};
template <int N> struct ForTiles<N, N> { template <class F> static __device__ __forceinline__ void run(F&&) {} };


template <int FILL, int T>
__device__ __forceinline__ void win_body(const v4d (&Y)[9], double& acc0, double& acc1, int& s0, int& s1, int& s2, int& s3, int& av0, int& av1, const double* gp) {
  if constexpr (T < 32) {
    constexpr int ja = T % 9, jb = (T * 5 + 1) % 9;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      MF(T, Y[ja][q], Y[jb][q]);
      if (FILL == 1) {
        double r0, r1;
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512" : "=v"(r0), "=v"(r1) : "v"((int)(threadIdx.x * 8 + ((T * 4 + q) & 7) * 1024)));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc0 = r0; acc1 = r1;
      }
      if (FILL == 2) asm volatile("ds_write_b64 %0, %1" :: "v"((int)(threadIdx.x * 8 + ((T * 4 + q) & 7) * 1024)), "v"(Y[0][q]) : "memory");
      if (FILL == 3) { double r; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(gp + 64 * ((T * 4 + q) & 31))); acc0 = r; }
      if (FILL == 4) { asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc"); }
      if (FILL == 5) { asm volatile("v_accvgpr_read_b32 %0, a255\n\tv_accvgpr_read_b32 %1, a254" : "=v"(av0), "=v"(av1)); }
    }
    win_body<FILL, T + 1>(Y, acc0, acc1, s0, s1, s2, s3, av0, av1, gp);
  }
}

// FILL between the MFMAs of a tile: 0 none, 1 two ds_read_b64, 2 one ds_write_b64, 3 one global_load_dwordx2, 4 four SALU, 5 two v_accvgpr_read (VALU)
template <int FILL>
__global__ __launch_bounds__(64, 1) void k_win(double* out, const double* in, int n, long long* t) {
  asm volatile("" ::: "a0", "a255");   // the accumulator file is ours: makes the kernel descriptor allocate all of it
  __shared__ double lds[2048];
  v4d Y[9];
#pragma unroll
  for (int j = 0; j < 9; j++) Y[j] = *(const v4d*)(in + 256 * j + 4 * threadIdx.x);
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = in[i];
  ForTiles<0, 32>::run([&]<int T>() { tile_zero<T>(); });
  __syncthreads();
  double acc0 = 0, acc1 = 0;
  int s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  int av0 = 0, av1 = 0;
  const double* gp = in + threadIdx.x;
  const long long t0 = clock64();
  for (int it = 0; it < n; it++) {
    win_body<FILL, 0>(Y, acc0, acc1, s0, s1, s2, s3, av0, av1, gp);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  double s = acc0 + acc1 + (double)(s0 + s1 + s2 + s3 + av0 + av1);
  ForTiles<0, 32>::run([&]<int T>() { const v4d v = tile_read<T>(); s += v[0] + v[1] + v[2] + v[3]; });
  out[threadIdx.x + 64 * (blockIdx.x & 1023)] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

// correctness of the hand-pinned tiles: one tile, C = -A^T-chunks * B-chunks against the builtin
__global__ __launch_bounds__(64, 1) void k_check(double* out, const double* in) {
  asm volatile("" ::: "a0", "a255");
  const v4d A = *(const v4d*)(in + 4 * threadIdx.x), B = *(const v4d*)(in + 256 + 4 * threadIdx.x);
  tile_zero<5>();
  tile_update<5>(A, B);
  const v4d r = tile_read<5>();
  v4d ref = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; q++) ref = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[q], B[q], ref, 0, 0, 0);
  double d = 0;
  for (int q = 0; q < 4; q++) d = fmax(d, fabs(r[q] - ref[q]));
  out[threadIdx.x] = d;
  out[64 + threadIdx.x] = fabs(ref[0]);
}

template <class K>
static void run(const char* name, K kern, int blocks, int n, double* out, const double* in, long long* t) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kern<<<blocks, 64>>>(out, in, 5, t); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<blocks, 64>>>(out, in, n, t);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  static long long h[4096];
  (void)hipMemcpy(h, t, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int b = 0; b < blocks; b++) mean += (double)h[b];
  mean /= blocks;
  const double nm = 128.0 * n;
  fflush(stdout); printf("%-34s blocks %4d: %.1f cycles per MFMA; wall %.3f ms -> %.2f TFLOP/s\n", name, blocks, mean / nm, ms, 2048.0 * nm * blocks / (ms * 1e-3) / 1e12);
}

int main() { setvbuf(stdout, NULL, _IONBF, 0);
  double *out, *in; long long* t;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&in, 1 << 20); (void)hipMalloc(&t, 8 * 4096);
  static double h[131072];
  for (int i = 0; i < 131072; i++) h[i] = 0.001 * ((i * 7919) % 1000) - 0.3;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  k_check<<<1, 64>>>(out, in); (void)hipDeviceSynchronize();
  double r[128]; (void)hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
  double d = 0, m = 0;
  for (int i = 0; i < 64; i++) { d = fmax(d, r[i]); m = fmax(m, r[64 + i]); }
  fflush(stdout); printf("hand-pinned tile vs builtin: max diff %.3g (magnitude %.3g)\n", d, m);
  const int n = 300;
  for (int blocks : {1, 1024}) {
    run("32 AGPR tiles, no filler", k_win<0>, blocks, n, out, in, t);
    run("+ 2 ds_read_b64 per MFMA", k_win<1>, blocks, n, out, in, t);
    run("+ 1 ds_write_b64 per MFMA", k_win<2>, blocks, n, out, in, t);
    run("+ 1 global_load_dwordx2 per MFMA", k_win<3>, blocks, n, out, in, t);
    run("+ 4 SALU per MFMA", k_win<4>, blocks, n, out, in, t);
    run("+ 2 v_accvgpr_read per MFMA", k_win<5>, blocks, n, out, in, t);
  }
  return 0;
}
