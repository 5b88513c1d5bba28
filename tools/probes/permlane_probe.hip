// gfx950 lane swaps (v_permlane16_swap / v_permlane32_swap) as a sum over the four 16-lane rows of a wavefront: checks the
// semantics the back substitution relies on (result = p[c] + p[c+16] + p[c+32] + p[c+48] in every lane).
// hipcc --offload-arch=gfx950 -O3 tools/probes/permlane_probe.hip -o tools/probes/permlane_probe && tools/probes/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double sum_rows(double p) {
  unsigned lo = __double2loint(p), hi = __double2hiint(p);
  v2u a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  v2u b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  double q = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(q); hi = __double2hiint(q);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__global__ void k(double* o) { o[threadIdx.x] = sum_rows(o[threadIdx.x]); }
int main() {
  double h[64], *d;
  for (int i = 0; i < 64; i++) h[i] = 1.0 + i * 1.25 + (i % 7) * 1e-3;
  hipMalloc(&d, sizeof h);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double r[64];
  hipMemcpy(r, d, sizeof r, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; i++) {
    const int c = i & 15;
    const double e = (h[c] + h[c + 16]) + (h[c + 32] + h[c + 48]);
    if (r[i] != e) bad++;
  }
  std::printf("permlane row sum: %s (%d lanes differ); lane 5 = %.6f expected %.6f\n", bad ? "MISMATCH" : "ok", bad, r[5], (h[5] + h[21]) + (h[37] + h[53]));
  return bad != 0;
}
