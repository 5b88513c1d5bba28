// Accuracy of two 1/sqrt(d) refinements of the v_rsq_f64 seed against the correctly rounded value (host long double):
//   A: the coupled step of tile_chol.h rsqrt_sqrt (9 dependent-ish FP64 instructions behind the seed)
//   B: one step with the second-order term, y (1 + e/2 + 3 e^2/8), e = 1 - d y^2 (5 instructions)
// and of the raw seed.  hipcc --offload-arch=gfx950 -O2 -o tools/probes/rsqrt_probe tools/probes/rsqrt_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* d, double* seed, double* a, double* b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = d[i];
  const double y = __builtin_amdgcn_rsq(x);
  seed[i] = y;
  {
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    a[i] = fma(fma(-h, g, 0.5), h + h, h + h);
  }
  {
    const double t = x * y;
    const double e = fma(-t, y, 1.0);
    const double p = fma(0.375, e, 0.5);
    const double q = y * e;
    b[i] = fma(q, p, y);
  }
}
int main() {
  const int n = 1 << 22;
  std::vector<double> h(n);
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> u(-300.0, 300.0), m(1.0, 2.0);
  for (int i = 0; i < n; i++) h[i] = (i & 1) ? m(rng) * 4.0 : std::ldexp(m(rng), (int)u(rng));
  double *d, *s, *a, *b;
  hipMalloc(&d, 8 * n); hipMalloc(&s, 8 * n); hipMalloc(&a, 8 * n); hipMalloc(&b, 8 * n);
  hipMemcpy(d, h.data(), 8 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, s, a, b, n);
  std::vector<double> hs(n), ha(n), hb(n);
  hipMemcpy(hs.data(), s, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(ha.data(), a, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, 8 * n, hipMemcpyDeviceToHost);
  double ms = 0, ma = 0, mb = 0; long na = 0, nb = 0;
  for (int i = 0; i < n; i++) {
    const long double ref = 1.0L / sqrtl((long double)h[i]);
    const double r = (double)ref, ulp = std::nextafter(r, INFINITY) - r;
    ms = std::fmax(ms, std::fabs((double)((long double)hs[i] - ref)) / r);
    const double ea = std::fabs((double)((long double)ha[i] - ref)) / ulp, eb = std::fabs((double)((long double)hb[i] - ref)) / ulp;
    ma = std::fmax(ma, ea); mb = std::fmax(mb, eb); na += ha[i] != r; nb += hb[i] != r;
  }
  std::printf("v_rsq_f64 seed: max relative error %.3e (2^%.1f)\n", ms, std::log2(ms));
  std::printf("A (coupled step, 9 instructions): max error %.3f ulp, %.2f %% not correctly rounded\n", ma, 100.0 * na / n);
  std::printf("B (second-order step, 5 instructions): max error %.3f ulp, %.2f %% not correctly rounded\n", mb, 100.0 * nb / n);
  return 0;
}
