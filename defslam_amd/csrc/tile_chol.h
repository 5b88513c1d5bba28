// 16x16 FP64 tile primitives on MFMA accumulator registers (gfx950), shared by the SfT solver (sft_kernels.hip) and the
// Schwarp normal-equation solve (nrsfm_kernels.hip).
//
// Accumulator layout of v_mfma_f64_16x16x4_f64: lane l = (g = l >> 4, c = l & 15), register q holds element
// [row g + 4q][column c].  Register q of a tile in this layout is at the same time the k-chunk q of the B operand of
// that matrix and of the A operand of its transpose -- the blocked factorisation below never leaves the registers.
#pragma once
#include <hip/hip_runtime.h>

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Sum over the four 16-lane rows of a wavefront (lanes c, c+16, c+32, c+48), result (p[c] + p[c+16]) + (p[c+32] + p[c+48]) in every
// lane: the gfx950 lane swaps v_permlane16_swap / v_permlane32_swap on the vector ALU instead of two round trips through the LDS
// crossbar (ds_bpermute) -- same association as `p += shfl_xor(p, 16); p += shfl_xor(p, 32)` (tools/probes/permlane_probe.hip).
__device__ __forceinline__ double sum_rows(double p) {
  typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
  unsigned lo = __double2loint(p), hi = __double2hiint(p);
  v2u_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  v2u_t b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double q = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(q); hi = __double2hiint(q);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// 1/sqrt(d) to double precision: v_rsq_f64 seed (~2^-26 relative) + one coupled Goldschmidt/Newton step
// (quadratic: ~2^-52) + one residual correction of the square root.
__device__ __forceinline__ void rsqrt_sqrt(double d, double& inv, double& s) {
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double res = fma(-g, g, d);
  s = fma(res, h, g);
  inv = fma(fma(-h, g, 0.5), h + h, h + h);   // one more correction of 1/sqrt without lengthening the sqrt chain
}

// Cholesky A = L L^T of a symmetric 16x16 tile and W = L^-1, blocked by 4 (13 MFMAs, 4 dependent block steps), one wavefront.
// Block step J (rows/columns 4J..4J+3 live in register J of lane groups g = 0..3):
//   1. the 10 entries of the symmetric 4x4 diagonal block are broadcast to every lane; every lane factors it and inverts
//      the factor redundantly (uniform scalars): D = Ld Ld^T, M = Ld^-1
//   2. Z  = Mpad * A[4J..4J+3, :]   one MFMA, B operand = register J of `a` as it is; Z[g][c] = L[c][4J+g] comes out
//      in exactly the lane layout the rank-4 update needs for both of its operands
//      Zw = Mpad * W[4J..4J+3, :]   the new rows 4J..4J+3 of W = L^-1
//   3. a -= Z^T Z (rank 4),  W[rows below] -= L[rows below, 4J..4J+3] * Zw,  W[4J..4J+3, :] = Zw
// Returns false when a pivot is not positive.
__device__ __forceinline__ bool chol_inv_blocked(v4d& a, v4d& w) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  w = (v4d){(g == c) ? 1.0 : 0.0, (g + 4 == c) ? 1.0 : 0.0, (g + 8 == c) ? 1.0 : 0.0, (g + 12 == c) ? 1.0 : 0.0};
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
  double plast = 1.0;
#pragma unroll
  for (int J = 0; J < 4; J++) {
    const double aJ = a[J];
    const int b0 = 4 * J;
    const double d00 = bcast_lane(aJ, b0), d10 = bcast_lane(aJ, 16 + b0), d11 = bcast_lane(aJ, 16 + b0 + 1);
    const double d20 = bcast_lane(aJ, 32 + b0), d21 = bcast_lane(aJ, 32 + b0 + 1), d22 = bcast_lane(aJ, 32 + b0 + 2);
    const double d30 = bcast_lane(aJ, 48 + b0), d31 = bcast_lane(aJ, 48 + b0 + 1), d32 = bcast_lane(aJ, 48 + b0 + 2), d33 = bcast_lane(aJ, 48 + b0 + 3);
    double i0, i1, i2, i3, sq;
    rsqrt_sqrt(d00, i0, sq);
    const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
    const double p1 = fma(-l10, l10, d11);
    rsqrt_sqrt(p1, i1, sq);
    const double l21 = fma(-l20, l10, d21) * i1, l31 = fma(-l30, l10, d31) * i1;
    const double p2 = fma(-l21, l21, fma(-l20, l20, d22));
    rsqrt_sqrt(p2, i2, sq);
    const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * i2;
    const double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
    rsqrt_sqrt(p3, i3, sq);
    // A pivot that is not positive (or not a number) turns its 1/sqrt into NaN or infinity, and from there every later pivot of the
    // tile into NaN (through l = d * inv and through the rank-4 update): the LAST pivot tells whether all sixteen were positive.
    plast = p3;
    // M = Ld^-1 (lower triangular)
    const double m10 = -(l10 * i0) * i1;
    const double m21 = -(l21 * i1) * i2;
    const double m32 = -(l32 * i2) * i3;
    const double m20 = -fma(l21, m10, l20 * i0) * i2;
    const double m31 = -fma(l32, m21, l31 * i1) * i3;
    const double m30 = -fma(l32, m20, fma(l31, m10, l30 * i0)) * i3;
    // A operand of Z = Mpad * rows: lane (i = c, k = g) holds M[i][k] for i < 4, k <= i
    double sel = 0.0;
    sel = (c == 0 && g == 0) ? i0 : sel;
    sel = (c == 1) ? (g == 0 ? m10 : (g == 1 ? i1 : 0.0)) : sel;
    sel = (c == 2) ? (g == 0 ? m20 : (g == 1 ? m21 : (g == 2 ? i2 : 0.0))) : sel;
    sel = (c == 3) ? (g == 0 ? m30 : (g == 1 ? m31 : (g == 2 ? m32 : i3))) : sel;
    const v4d zw = __builtin_amdgcn_mfma_f64_16x16x4f64(sel, w[J], zero, 0, 0, 0);
    if (J < 3) {
      const v4d z = __builtin_amdgcn_mfma_f64_16x16x4f64(sel, aJ, zero, 0, 0, 0);
      const double lp = z[0];                       // L[c][4J+g]
      const double nlp = -lp;
      a = __builtin_amdgcn_mfma_f64_16x16x4f64(nlp, lp, a, 0, 0, 0);
      const double below = (c >= 4 * J + 4) ? nlp : 0.0;
      w = __builtin_amdgcn_mfma_f64_16x16x4f64(below, zw[0], w, 0, 0, 0);
    }
    w[J] = zw[0];
  }
  return plast > 0.0;
}

