// Probe: lane layout of v_mfma_f64_16x16x4_f64 on gfx950 (A: 16x4, B: 4x16, C/D: 16x16, 4 doubles per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* C) {
  const int l = threadIdx.x;
  // guide: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; C/D: col = l&15, row = (l>>4) + 4*reg
  v4d acc = {0, 0, 0, 0};
  for (int kk = 0; kk < 4; kk++) {
    const double a = A[(l & 15) * 16 + 4 * kk + (l >> 4)];
    const double b = B[(4 * kk + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
int main() {
  double hA[256], hB[256], hC[256], ref[256];
  for (int i = 0; i < 256; i++) { hA[i] = sin(0.37 * i) + 0.01 * i; hB[i] = cos(0.11 * i * i) - 0.02 * i; }
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 16; k++) s += hA[i * 16 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 2048);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(hC, dC, 2048, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 256; i++) e = fmax(e, fabs(hC[i] - ref[i]));
  printf("max abs err vs host GEMM (asymmetric A,B): %.3e\n", e);
  return e < 1e-9 ? 0 : 1;
}
