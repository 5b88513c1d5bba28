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

// 1/sqrt(d) alone: the v_rsq_f64 seed (2^-24) and ONE step that carries the second-order term, y (1 + e/2 + 3 e^2/8) with e = 1 - d y^2
// exact through the fused multiply-add -- 5 dependent-ish instructions behind the seed, within 0.99 ulp of the correctly rounded value
// (rsqrt_sqrt above: 9 instructions, 1.61 ulp; tools/probes/rsqrt_probe.hip).
__device__ __forceinline__ double rsqrt_only(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double t = d * y;
  const double e = fma(-t, y, 1.0);
  const double p = fma(0.375, e, 0.5);
  const double q = y * e;
  return fma(q, p, y);
}

// Block step J of the 16 x 16 factorisation, the part every lane computes on broadcast scalars: Cholesky Ld Ld^T of the 4 x 4 diagonal block
// of `aJ` (register J of the tile) and the A operand of the products with M = Ld^-1 -- lane (g, c) gets M[c][g] for c < 4 and 0 otherwise (callers that
// multiply with the 4 x 4 x 4 MFMA pass c & 3: M replicated over the four blocks).  This is a CHAIN of
// dependent FP64 operations (four pivots: 1/sqrt, scale, update), the pipe mostly waits for its own results, so what counts is the length of
// the critical path, not the instruction count:
//   * every lane solves Ld x = e_g -- its own column of the inverse -- by four steps of forward substitution and keeps x_c (22 instructions;
//     the explicit formulas of the six off-diagonal entries + a ten-way select took 41);
//   * x3, the last value of the chain, enters the LAST select (r04: the tile Cholesky of the one-wavefront solver 3.09 k -> 2.32 k cycles).
// `plast`: the last pivot -- a pivot that is not positive (or not a number) turns its 1/sqrt into NaN or infinity and from there every later
// pivot of the tile into NaN (through l = d * inv and through the rank-4 update), so the LAST pivot tells whether all sixteen were positive.
__device__ __forceinline__ double chol4_inverse_operand(double aJ, int J, int g, int c, double& plast) {
  const int b0 = 4 * J;
  const double d00 = bcast_lane(aJ, b0), d10 = bcast_lane(aJ, 16 + b0), d11 = bcast_lane(aJ, 16 + b0 + 1);
  const double d20 = bcast_lane(aJ, 32 + b0), d21 = bcast_lane(aJ, 32 + b0 + 1), d22 = bcast_lane(aJ, 32 + b0 + 2);
  const double d30 = bcast_lane(aJ, 48 + b0), d31 = bcast_lane(aJ, 48 + b0 + 1), d32 = bcast_lane(aJ, 48 + b0 + 2), d33 = bcast_lane(aJ, 48 + b0 + 3);
  const double i0 = rsqrt_only(d00);
  const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
  const double p1 = fma(-l10, l10, d11);
  const double i1 = rsqrt_only(p1);
  const double l21 = fma(-l20, l10, d21) * i1, l31 = fma(-l30, l10, d31) * i1;
  const double p2 = fma(-l21, l21, fma(-l20, l20, d22));
  const double i2 = rsqrt_only(p2);
  const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * i2;
  const double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
  const double i3 = rsqrt_only(p3);
  plast = p3;
  const double x0 = ((g == 0) ? 1.0 : 0.0) * i0;
  const double x1 = fma(-l10, x0, (g == 1) ? 1.0 : 0.0) * i1;
  const double x2 = fma(-l21, x1, fma(-l20, x0, (g == 2) ? 1.0 : 0.0)) * i2;
  const double x3 = fma(-l32, x2, fma(-l31, x1, fma(-l30, x0, (g == 3) ? 1.0 : 0.0))) * i3;
  double sel = (c == 0) ? x0 : 0.0;
  sel = (c == 1) ? x1 : sel;
  sel = (c == 2) ? x2 : sel;
  sel = (c == 3) ? x3 : sel;
  return sel;
}

// Cholesky A = L L^T of a symmetric 16x16 tile and W = L^-1, blocked by 4 (13 MFMAs -- seven of them 4 x 4 x 4 --, 4 dependent block steps), one wavefront.
// Block step J (rows/columns 4J..4J+3 live in register J of lane groups g = 0..3):
//   1. the 10 entries of the symmetric 4x4 diagonal block are broadcast to every lane; every lane factors it and inverts
//      the factor redundantly (uniform scalars): D = Ld Ld^T, M = Ld^-1
//   2. Z  = M * A[4J..4J+3, :]   one MFMA, B operand = register J of `a` as it is; Z[g][c] = L[c][4J+g] comes out
//      in exactly the lane layout the rank-4 update needs for both of its operands
//      Zw = M * W[4J..4J+3, :]   the new rows 4J..4J+3 of W = L^-1
//      (r06: both are 4 x 16 results = the four 4 x 4 blocks of ONE v_mfma_f64_4x4x4_4b_f64 -- A_b[i][k] lane i + 4 b + 16 k = M[i][k] for
//      every block b, i.e. the operand of chol4_inverse_operand replicated over the four quads of a 16-lane row (c & 3 instead of c);
//      B_b[k][j] lane j + 4 b + 16 k = register J of the tile as it lies; D lane j + 4 b + 16 i = register 0 of the 16 x 16 result the padded
//      16 x 16 x 4 product used to deliver.  16 cycles instead of 64, and a quarter of the result latency on the chain of dependent operations)
//   3. a -= Z^T Z (rank 4),  W[rows below] -= L[rows below, 4J..4J+3] * Zw,  W[4J..4J+3, :] = Zw
// Returns false when a pivot is not positive.
__device__ __forceinline__ bool chol_inv_blocked(v4d& a, v4d& w) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  w = (v4d){(g == c) ? 1.0 : 0.0, (g + 4 == c) ? 1.0 : 0.0, (g + 8 == c) ? 1.0 : 0.0, (g + 12 == c) ? 1.0 : 0.0};
  double plast = 1.0;
#pragma unroll
  for (int J = 0; J < 4; J++) {
    const double aJ = a[J];
    const double sel = chol4_inverse_operand(aJ, J, g, c & 3, plast);
    const double zw = __builtin_amdgcn_mfma_f64_4x4x4f64(sel, w[J], 0.0, 0, 0, 0);
    if (J < 3) {
      const double lp = __builtin_amdgcn_mfma_f64_4x4x4f64(sel, aJ, 0.0, 0, 0, 0);   // L[c][4J+g]
      const double nlp = -lp;
      a = __builtin_amdgcn_mfma_f64_16x16x4f64(nlp, lp, a, 0, 0, 0);
      const double below = (c >= 4 * J + 4) ? nlp : 0.0;
      w = __builtin_amdgcn_mfma_f64_16x16x4f64(below, zw, w, 0, 0, 0);
    }
    w[J] = zw;
  }
  return plast > 0.0;
}

