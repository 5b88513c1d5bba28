// One wavefront = one factorisation: the banded-arrowhead Cholesky of H + lambda I and the back substitution of tile mode 1
// (half-bandwidth <= 128) written for ONE wave that owns a SIMD (512 registers), the throughput shape of the batched solver.
// Included by sft_kernels.hip (inside its anonymous namespace).
//
// Why (measured on MI355X, tools/probes/wave_mfma_probe.hip, agpr_tile_probe.hip): v_mfma_f64_16x16x4_f64 issues every 64 cycles and NO
// vector-ALU instruction of the same wave issues in its shadow (LDS, global memory and scalar instructions do), so a SIMD is a serial
// resource for MFMA + VALU work and what a factor step costs is the instructions it issues.  The multi-wavefront solver
// (factor_tiles_df) reaches 75 % of that bound: the waves of a problem wait for each other's tiles through LDS flags.  Here nothing
// waits: the whole 8 x 8 tile window (36 live tiles = 288 registers) belongs to one wave, four problems are resident per CU.
//
// Register plan.  The compiler cannot keep 36 MFMA accumulators in the accumulator file (it shuttles them through VGPRs every step:
// probe), so the window tiles are pinned by hand: EVERY v_mfma of this file is inline asm, tile T of the window is a[8T:8T+7], and
// the compiler never sees an accumulator register (its own code stays below 256 VGPRs; audit: `.vgpr_spill_count 0`, no v_accvgpr
// outside ASMSTART/ASMEND).  The window slides without moving a register: tile (I, J), d = I - J, of ring row r = I mod 8 lives in
// physical tile wv_phys(r, d) from the step it enters (I - 8) to the step its column is eliminated (J); tiles whose lifetimes add
// up to a full turn of the ring share a physical tile (d with 8 - d), 36 physical tiles in all: 32 in a[0:255], the four d = 4
// tiles and the eight border tiles in LDS (read - 4 MFMAs - written back once per step: LDS traffic is free beside MFMAs).
// The step loop is specialised for the eight ring phases (literal register numbers) around ONE copy of the tile Cholesky.
//
// Layout.  Tiles are kept TRANSPOSED: D(I,J) = H(I,J)^T in accumulator order (lane (g, c), register q: element [g + 4q][c]).
// Register q of a tile in that order is chunk q of the B operand of the tile and of the A operand of its transpose, so
//   TRSM    Y_i = X(k+i,k)^T = W D(k+i,k)            A = acc(W^T)[q]   B = acc(D)[q]
//   update  D(k+i,k+j) -= Y_j^T Y_i                  A = acc(Y_j)[q]   B = acc(Y_i)[q]     (NEG bit on A: no negation on the VALU)
// take every operand as it lies in registers -- no LDS round trip, no lane shuffle; only W^T is transposed through LDS once per step.
// The 7 camera / right-hand-side rows ride along as border tiles Bd(J)^T (16 x 7) with the same two formulas, the corner as Yb^T Yb.
#pragma once

// ---- physical tile of window tile (ring row r = I mod 8, d = I - J in 0..7) ------------------------------------------------------
__host__ __device__ constexpr int wv_phys(int r, int d) {
  return d == 0 ? r : (d <= 3 ? 8 * d + r : (d == 4 ? 32 + (r & 3) : 8 * (8 - d) + ((r + 8 - d) & 7)));
}
constexpr int WV_AGPR_TILES = 32;                 // physical tiles 0..31 = a[0:255]; 32..35 = LDS tiles 0..3 (the d = 4 tiles)
constexpr int WV_LDS_WIN = 4, WV_LDS_BORD = 8;    // LDS tiles 4..11: border tiles Bd(J)^T, ring slot J mod 8
constexpr int WV_LDS_DOUBLES = (WV_LDS_WIN + WV_LDS_BORD) * 256 + 16 * 17 + 64;   // + W transposition scratch + corner
typedef double v2d_w __attribute__((ext_vector_type(2)));
using lds_v2d = __attribute__((address_space(3))) v2d_w;

// MFMA result -> any reader other than the next MFMA that accumulates into it: 16 passes + write-back (the compiler pads nothing
// inside or behind an asm statement)
#define WV_NOP_MFMA_RESULT "s_nop 15\n\ts_nop 3"

// ---- MFMA on a hand-pinned accumulator tile -----------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ void wv_upd_agpr(const v4d& A, const v4d& B, int q0 = 0) {   // a[T] -= A^T-chunks x B-chunks
  static_assert(T >= 0 && T < WV_AGPR_TILES, "accumulator tile");
#define WV_MF(q) asm volatile("v_mfma_f64_16x16x4_f64 a[%c0:%c1], %2, %3, a[%c0:%c1] neg:[1,0,0]" ::"i"(8 * T), "i"(8 * T + 7), "v"(A[q]), "v"(B[q]))
  if (q0 <= 0) WV_MF(0);
  if (q0 <= 1) WV_MF(1);
  if (q0 <= 2) WV_MF(2);
  WV_MF(3);
#undef WV_MF
}
// the same into a VGPR tile (LDS-resident tiles, corner)
// (the compiler may copy C into the operand registers right in front of the statement -- it keeps the corner elsewhere between steps --:
// a VALU write needs two wait states before an MFMA reads it, and nothing pads an asm statement)
__device__ __forceinline__ void wv_upd_vgpr(v4d& C, const v4d& A, const v4d& B, int q0 = 0) {
#define WV_MF(q) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(C) : "v"(A[q]), "v"(B[q]))
  asm volatile("s_nop 1" : "+v"(C));
  if (q0 <= 0) WV_MF(0);
  if (q0 <= 1) WV_MF(1);
  if (q0 <= 2) WV_MF(2);
  WV_MF(3);
#undef WV_MF
}
// Y = W D with D in accumulator tile T (B operand straight from a[..]); A = acc(W^T)
template <int T>
__device__ __forceinline__ void wv_trsm_agpr(v4d& Y, const v4d& Wt) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, a[%c5:%c6], 0\n\t"
               "v_mfma_f64_16x16x4_f64 %0, %2, a[%c7:%c8], %0\n\t"
               "v_mfma_f64_16x16x4_f64 %0, %3, a[%c9:%c10], %0\n\t"
               "v_mfma_f64_16x16x4_f64 %0, %4, a[%c11:%c12], %0"
               : "=&v"(Y)
               : "v"(Wt[0]), "v"(Wt[1]), "v"(Wt[2]), "v"(Wt[3]), "i"(8 * T), "i"(8 * T + 1), "i"(8 * T + 2), "i"(8 * T + 3), "i"(8 * T + 4), "i"(8 * T + 5),
                 "i"(8 * T + 6), "i"(8 * T + 7)
               : "memory");
}
__device__ __forceinline__ void wv_trsm_vgpr(v4d& Y, const v4d& Wt, const v4d& D, int q0 = 0) {
  if (q0 <= 0) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %1, %5, 0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                 : "=&v"(Y)
                 : "v"(Wt[0]), "v"(Wt[1]), "v"(Wt[2]), "v"(Wt[3]), "v"(D[0]), "v"(D[1]), "v"(D[2]), "v"(D[3])
                 : "memory");
  } else {   // the leading q0 row chunks of D are structurally zero (the corner tile of the band)
    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(Y) : "v"(Wt[3]), "v"(D[3]) : "memory");
    if (q0 <= 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(Y) : "v"(Wt[2]), "v"(D[2]) : "memory");
    if (q0 <= 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(Y) : "v"(Wt[1]), "v"(D[1]) : "memory");
  }
}
// the values of MFMA-written VGPRs may be read by compiler code behind this statement
__device__ __forceinline__ void wv_mfma_fence() { asm volatile(WV_NOP_MFMA_RESULT ::: "memory"); }

// accumulator tile -> VGPRs (the diagonal tile on its way to the tile Cholesky)
template <int T>
__device__ __forceinline__ v4d wv_read_agpr() {
  int r[8];
  asm volatile(WV_NOP_MFMA_RESULT "\n\t"
               "v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c9]\n\tv_accvgpr_read_b32 %2, a[%c10]\n\tv_accvgpr_read_b32 %3, a[%c11]\n\t"
               "v_accvgpr_read_b32 %4, a[%c12]\n\tv_accvgpr_read_b32 %5, a[%c13]\n\tv_accvgpr_read_b32 %6, a[%c14]\n\tv_accvgpr_read_b32 %7, a[%c15]"
               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
               : "i"(8 * T), "i"(8 * T + 1), "i"(8 * T + 2), "i"(8 * T + 3), "i"(8 * T + 4), "i"(8 * T + 5), "i"(8 * T + 6), "i"(8 * T + 7));
  v4d v;
#pragma unroll
  for (int q = 0; q < 4; q++) v[q] = __hiloint2double(r[2 * q + 1], r[2 * q]);
  return v;
}
// four 8-byte elements of the compact H blocks straight into an accumulator tile (global memory can address the accumulator file):
// byte offsets off[q] relative to `base` (wave-uniform)
template <int T>
__device__ __forceinline__ void wv_gather_agpr(const SFT_G double* base, unsigned o0, unsigned o1, unsigned o2, unsigned o3) {
  asm volatile("global_load_dwordx2 a[%c5:%c6], %0, %4\n\t"
               "global_load_dwordx2 a[%c7:%c8], %1, %4\n\t"
               "global_load_dwordx2 a[%c9:%c10], %2, %4\n\t"
               "global_load_dwordx2 a[%c11:%c12], %3, %4"
               :
               : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "i"(8 * T), "i"(8 * T + 1), "i"(8 * T + 2), "i"(8 * T + 3), "i"(8 * T + 4), "i"(8 * T + 5),
                 "i"(8 * T + 6), "i"(8 * T + 7)
               : "memory");
}
__device__ __forceinline__ void wv_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- LDS-resident tiles: two planes of 16 bytes per lane (conflict-free 128-bit accesses) ---------------------------------------
__device__ __forceinline__ v4d wv_lds_load(const lds_double* tile, int lane) {
  const lds_v2d* p = reinterpret_cast<const lds_v2d*>(tile);
  const v2d_w a = p[lane], b = p[64 + lane];
  return (v4d){a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ void wv_lds_store(lds_double* tile, int lane, const v4d& v) {
  lds_v2d* p = reinterpret_cast<lds_v2d*>(tile);
  p[lane] = (v2d_w){v[0], v[1]};
  p[64 + lane] = (v2d_w){v[2], v[3]};
}

// ---- 16 x 16 Cholesky + inverse, every MFMA as asm on VGPR tiles (tile_chol.h: chol_inv_blocked is the compiler-scheduled original) --
__device__ __forceinline__ bool wv_chol_inv(v4d& a, v4d& w) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  w = (v4d){(g == c) ? 1.0 : 0.0, (g + 4 == c) ? 1.0 : 0.0, (g + 8 == c) ? 1.0 : 0.0, (g + 12 == c) ? 1.0 : 0.0};
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
    plast = p3;
    const double m10 = -(l10 * i0) * i1;
    const double m21 = -(l21 * i1) * i2;
    const double m32 = -(l32 * i2) * i3;
    const double m20 = -fma(l21, m10, l20 * i0) * i2;
    const double m31 = -fma(l32, m21, l31 * i1) * i3;
    const double m30 = -fma(l32, m20, fma(l31, m10, l30 * i0)) * i3;
    double sel = 0.0;
    sel = (c == 0 && g == 0) ? i0 : sel;
    sel = (c == 1) ? (g == 0 ? m10 : (g == 1 ? i1 : 0.0)) : sel;
    sel = (c == 2) ? (g == 0 ? m20 : (g == 1 ? m21 : (g == 2 ? i2 : 0.0))) : sel;
    sel = (c == 3) ? (g == 0 ? m30 : (g == 1 ? m31 : (g == 2 ? m32 : i3))) : sel;
    const double wJ = w[J];
    v4d zw, z;
    if (J < 3) {
      asm volatile("s_nop 1\n\t"
                   "v_mfma_f64_16x16x4_f64 %0, %2, %3, 0\n\t"
                   "v_mfma_f64_16x16x4_f64 %1, %2, %4, 0\n\t" WV_NOP_MFMA_RESULT
                   : "=&v"(zw), "=&v"(z)
                   : "v"(sel), "v"(wJ), "v"(aJ));
      const double lp = z[0];
      const double nlp = -lp;
      const double below = (c >= 4 * J + 4) ? nlp : 0.0;
      const double zw0 = zw[0];
      asm volatile("s_nop 1\n\t"
                   "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\t"
                   "v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\t" WV_NOP_MFMA_RESULT
                   : "+v"(a), "+v"(w)
                   : "v"(nlp), "v"(lp), "v"(below), "v"(zw0));
      w[J] = zw0;
    } else {
      asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, 0\n\t" WV_NOP_MFMA_RESULT : "=&v"(zw) : "v"(sel), "v"(wJ));
      w[J] = zw[0];
    }
  }
  return plast > 0.0;
}

// ---- per-problem constants of a factorisation, all wave-uniform --------------------------------------------------------------------
struct WvProb {
  int Dn, Dnp, nT, q8;                    // q8: leading row chunks of the d = 8 corner tile that are structurally zero (kd = 122: 1)
  double lambda, lam_corner;
  const SFT_G double* Hc;
  const SFT_G uint32_t* hgl;
  const SFT_G double* Hbord;
  SFT_G double *Lg, *Linv;
};

// gather list of tile (I, d) for the TRANSPOSED tile: lane (g, c), register q takes element [row c][column g + 4q] of H(I, I-d), i.e. the
// entry the packer wrote for lane (c & 3) * 16 + g + 4q, register c >> 2.  Rows behind the matrix take the all-zero list row nT.
__device__ __forceinline__ void wv_list(const WvProb& W, int I, int d, int lane, unsigned (&o)[4]) {
  const int g = lane >> 4, c = lane & 15;
  const SFT_G uint32_t* row = W.hgl + ((size_t)(I < W.nT ? I : W.nT) * (BT + 1) + d) * 256;
#pragma unroll
  for (int q = 0; q < 4; q++) o[q] = row[4 * ((c & 3) * 16 + g + 4 * q) + (c >> 2)];
}
__device__ __forceinline__ v4d wv_gather_vgpr(const WvProb& W, const unsigned (&o)[4]) {
  const auto b8 = reinterpret_cast<const SFT_G char*>(W.Hc);
  v4d v;
#pragma unroll
  for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const SFT_G double*>(b8 + o[q]);
  return v;
}
// border tile Bd(J)^T from the 8-row border of H (rows 0-5 camera, 6 right-hand side, 7 zero): lane (g, c), register q = Hbord[c][16 J + g + 4q]
__device__ __forceinline__ v4d wv_border_fresh(const WvProb& W, int J, int lane) {
  const int g = lane >> 4, c = lane & 15;
  v4d v = {0.0, 0.0, 0.0, 0.0};
  if (c < 8 && J < W.nT) {
    const SFT_G double* p = W.Hbord + (size_t)c * W.Dnp + TS * J + g;
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = p[4 * q];
  }
  return v;
}

// Lists of the eight tiles (I, d), d = 0..7, of a row that enters the window: 32 offsets per lane, requested a step ahead.
struct WvRowList { unsigned o[8][4]; };
__device__ __forceinline__ void wv_row_list(const WvProb& W, int I, int lane, WvRowList& L) {
#pragma unroll
  for (int d = 0; d < 8; d++) wv_list(W, I, d, lane, L.o[d]);
}
// The row's elements: accumulator tiles straight from memory, the d = 4 tile through VGPRs into its LDS tile.  PH = I mod 8; dmax: the
// prologue rows have fewer tiles (J >= 0).  The caller waits (wv_wait_vm) before the first MFMA on these tiles.
template <int PH, int D>
__device__ __forceinline__ void wv_row_fetch_d(const WvProb& W, const WvRowList& L, int dmax, lds_double* ldsw, int lane) {
  if constexpr (D < 8) {
    if (D <= dmax) {
      if constexpr (D == 4) {
        const v4d v = wv_gather_vgpr(W, L.o[D]);
        wv_lds_store(ldsw + 256 * (wv_phys(PH, 4) - WV_AGPR_TILES), lane, v);
      } else {
        wv_gather_agpr<wv_phys(PH, D)>(W.Hc, L.o[D][0], L.o[D][1], L.o[D][2], L.o[D][3]);
      }
    }
    wv_row_fetch_d<PH, D + 1>(W, L, dmax, ldsw, lane);
  }
}

// ---- one factor step, ring phase PH = k mod 8 ------------------------------------------------------------------------------------------
struct WvState {
  v4d Y[9];          // Y[i], i = 1..8: X(k+i,k)^T; Y[0]: the border tile Xb^T of column k
  v4d corner;        // 7 x 7 camera corner (+ right-hand side row), accumulator order, lower triangle meaningful
  v4d araw;          // raw tile (k+8, k)^T of the NEXT step's column (the d = 8 corner of the band)
  v4d bnext;         // raw border tile of column k+8 (enters the ring this step)
  WvRowList rl;      // gather lists of the row that enters the window this step (row k+8)
  unsigned al[4];    // gather list of tile (k+9, k+1): the next araw
  int ok;
};

template <int PH, int I>
__device__ __forceinline__ void wv_trsm_cols(WvState& S, const v4d& Wt, int k, int nT, const lds_double* ldsw, int lane) {
  if constexpr (I <= 7) {
    if (k + I < nT) {
      if constexpr (I == 4) {
        const v4d D = wv_lds_load(ldsw + 256 * (wv_phys((PH + 4) & 7, 4) - WV_AGPR_TILES), lane);
        wv_trsm_vgpr(S.Y[4], Wt, D);
      } else {
        wv_trsm_agpr<wv_phys((PH + I) & 7, I)>(S.Y[I], Wt);
      }
    } else {
      S.Y[I] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    wv_trsm_cols<PH, I + 1>(S, Wt, k, nT, ldsw, lane);
  }
}

// window tiles (k+I, k+J), 1 <= J <= I: rows 1..7 first, row 8 (fetched during this step) last
template <int PH, int I, int J>
__device__ __forceinline__ void wv_update_tiles(WvState& S, int k, int nT, int q8, lds_double* ldsw, int lane) {
  if constexpr (I <= 8) {
    if (k + I < nT) {
      constexpr int d = I - J, r = (PH + I) & 7;
      const int q0 = (I == 8) ? q8 : 0;
      if constexpr (d == 4) {
        lds_double* t = ldsw + 256 * (wv_phys(r, 4) - WV_AGPR_TILES);
        v4d C = wv_lds_load(t, lane);
        wv_upd_vgpr(C, S.Y[J], S.Y[I], q0);
        wv_mfma_fence();
        wv_lds_store(t, lane, C);
      } else {
        wv_upd_agpr<wv_phys(r, d)>(S.Y[J], S.Y[I], q0);
      }
    }
    if constexpr (J < I) wv_update_tiles<PH, I, J + 1>(S, k, nT, q8, ldsw, lane);
  }
}
template <int PH, int I0, int I1>
__device__ __forceinline__ void wv_update_rows(WvState& S, int k, int nT, int q8, lds_double* ldsw, int lane) {
  if constexpr (I0 <= I1) {
    wv_update_tiles<PH, I0, 1>(S, k, nT, q8, ldsw, lane);
    wv_update_rows<PH, I0 + 1, I1>(S, k, nT, q8, ldsw, lane);
  }
}
// border tiles Bd(k+J)^T -= Y_J^T Yb, J = J0..J1 (LDS ring slot (k+J) mod 8)
template <int PH, int J, int J1>
__device__ __forceinline__ void wv_update_border(WvState& S, int k, int nT, int q8, lds_double* ldsb, int lane) {
  if constexpr (J <= J1) {
    if (k + J < nT) {
      lds_double* t = ldsb + 256 * ((PH + J) & 7);
      v4d C = wv_lds_load(t, lane);
      wv_upd_vgpr(C, S.Y[J], S.Y[0], (J == 8) ? q8 : 0);
      wv_mfma_fence();
      wv_lds_store(t, lane, C);
    }
    wv_update_border<PH, J + 1, J1>(S, k, nT, q8, ldsb, lane);
  }
}

// Part A of a step (before the tile Cholesky): the diagonal tile of column k out of the accumulator file, damped.
template <int PH>
__device__ __forceinline__ v4d wv_step_diag(const WvProb& W, int k, int lane) {
  v4d d = wv_read_agpr<wv_phys(PH, 0)>();
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (g + 4 * q == c && TS * k + c < W.Dn) d[q] += W.lambda;
  return d;
}

// Part B (behind the tile Cholesky): TRSM of block column k, the row that enters the window, the trailing update.
template <int PH>
__device__ __forceinline__ void wv_step_rest(const WvProb& W, WvState& S, const v4d& Wt, int k, lds_double* lds, int lane) {
  lds_double* ldsw = lds;                       // LDS tiles 0..3: the d = 4 window tiles
  lds_double* ldsb = lds + 256 * WV_LDS_WIN;    // LDS tiles 4..11: border ring
  const int nT = W.nT;
  // ---- block column k: Y_i = W D(k+i, k), border Yb = W Bd(k)^T
  wv_trsm_cols<PH, 1>(S, Wt, k, nT, ldsw, lane);
  if (k + 8 < nT) wv_trsm_vgpr(S.Y[8], Wt, S.araw, W.q8);
  else S.Y[8] = (v4d){0.0, 0.0, 0.0, 0.0};
  {
    const v4d Bk = wv_lds_load(ldsb + 256 * PH, lane);
    wv_trsm_vgpr(S.Y[0], Wt, Bk);
  }
  wv_mfma_fence();
  // L leaves from the registers it was computed in: block column k = [Yb | Y_1 .. Y_8] at slots (k, 0..8), 2 KB each
  {
    SFT_G double* col = W.Lg + ((size_t)k * (BT + 1)) * 256 + 4 * lane;
#pragma unroll
    for (int i = 0; i <= 8; i++)
      if (i == 0 || k + i < nT) *reinterpret_cast<SFT_G v4d*>(col + 256 * i) = S.Y[i];
  }
  // ---- the ring row / border slot of column k are free: row k+8 enters (its lists came a step ahead), the lists of row k+9 are requested
  wv_row_fetch_d<PH, 0>(W, S.rl, 7, ldsw, lane);
  wv_lds_store(ldsb + 256 * PH, lane, S.bnext);
  const v4d araw_next = wv_gather_vgpr(W, S.al);          // tile (k+9, k+1)^T for the next step's TRSM
  wv_row_list(W, k + 9, lane, S.rl);
  wv_list(W, k + 10, 8, lane, S.al);
  S.bnext = wv_border_fresh(W, k + 9, lane);
  // ---- trailing update: rows 1..7, border 1..7, corner; then (the fetched row has landed) row 8 and border 8
  wv_upd_vgpr(S.corner, S.Y[0], S.Y[0]);
  wv_update_rows<PH, 1, 7>(S, k, nT, W.q8, ldsw, lane);
  wv_update_border<PH, 1, 7>(S, k, nT, W.q8, ldsb, lane);
  wv_wait_vm();
  wv_update_rows<PH, 8, 8>(S, k, nT, W.q8, ldsw, lane);
  wv_update_border<PH, 8, 8>(S, k, nT, W.q8, ldsb, lane);
  // The corner's MFMAs were issued at the head of the update: it is long complete here.  This empty statement takes the corner as an
  // operand, so any register copy the compiler makes of it (the merge of the eight phase bodies) sits BEHIND the whole update -- a copy
  // right behind the MFMA statement would read the registers before the matrix pipe has written them (no hazard padding around asm).
  asm volatile("" : "+v"(S.corner));
  S.araw = araw_next;
}

// prologue: tile rows 0..7 of H into the window
template <int PH>
__device__ __forceinline__ void wv_prologue_row(const WvProb& W, lds_double* lds, int lane) {
  WvRowList L;
  wv_row_list(W, PH, lane, L);
  wv_row_fetch_d<PH, 0>(W, L, PH, lds, lane);
  wv_lds_store(lds + 256 * (WV_LDS_WIN + PH), lane, wv_border_fresh(W, PH, lane));
}

// ---- the factorisation + back substitution of one problem by one wavefront --------------------------------------------------------------
// lds: WV_LDS_DOUBLES doubles of this wave.  Returns the "all pivots positive" flag; x (P.x) = solution in the natural ordering
// (node unknowns, then the 6 camera unknowns at Dnp).
__device__ __forceinline__ int wv_factor_solve(const SftDev& P, double lambda, double lam_corner, lds_double* lds) {
  asm volatile("" ::: "a0", "a255");   // the accumulator file is ours (the kernel descriptor allocates all of it)
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  WvProb W;
  W.Dn = uni(P.Dn);
  W.Dnp = ((W.Dn + NB - 1) / NB) * NB;
  W.nT = W.Dnp / TS;
  W.q8 = min(3, max(0, (TS * BT - uni(P.kd)) / 4));
  W.lambda = lambda;
  W.lam_corner = lam_corner;
  W.Hc = uni(P.Hc); W.hgl = uni(P.hgather); W.Hbord = uni(P.Hbord); W.Lg = uni(P.Lb); W.Linv = uni(P.Linv);
  lds_double* wscr = lds + 256 * (WV_LDS_WIN + WV_LDS_BORD);   // 16 x 17 transposition scratch of W
  lds_double* Cn = wscr + 16 * 17;                              // 7 x 7 corner

#ifdef DSH_LAB
  const long long wv_t0 = clock64();
#endif
  WvState S;
#pragma unroll
  for (int i = 0; i < 9; i++) S.Y[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  S.corner = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int r = g + 4 * q;
    if (r < SFT_BORDER && c < SFT_BORDER && c <= r) S.corner[q] = P.Hcorner[r * 7 + c] + ((r == c && r < 6) ? lam_corner : 0.0);
  }
  S.ok = 1;
  wv_prologue_row<0>(W, lds, lane); wv_prologue_row<1>(W, lds, lane); wv_prologue_row<2>(W, lds, lane); wv_prologue_row<3>(W, lds, lane);
  wv_prologue_row<4>(W, lds, lane); wv_prologue_row<5>(W, lds, lane); wv_prologue_row<6>(W, lds, lane); wv_prologue_row<7>(W, lds, lane);
  {
    unsigned a0[4];
    wv_list(W, 8, 8, lane, a0);
    S.araw = wv_gather_vgpr(W, a0);
  }
  wv_row_list(W, 8, lane, S.rl);
  wv_list(W, 9, 8, lane, S.al);
  S.bnext = wv_border_fresh(W, 8, lane);
  wv_wait_vm();
#ifdef DSH_LAB
  const long long wv_t1 = clock64();
#endif

#pragma unroll 1
  for (int k = 0; k < W.nT; k++) {
    const int ph = k & 7;
    v4d d;
    switch (ph) {
      case 0: d = wv_step_diag<0>(W, k, lane); break;
      case 1: d = wv_step_diag<1>(W, k, lane); break;
      case 2: d = wv_step_diag<2>(W, k, lane); break;
      case 3: d = wv_step_diag<3>(W, k, lane); break;
      case 4: d = wv_step_diag<4>(W, k, lane); break;
      case 5: d = wv_step_diag<5>(W, k, lane); break;
      case 6: d = wv_step_diag<6>(W, k, lane); break;
      default: d = wv_step_diag<7>(W, k, lane); break;
    }
    v4d w;
    if (!wv_chol_inv(d, w)) {
#ifdef DSH_LAB
      if (S.ok && lane == 0) P.dbg[3] = 1000.0 + k;   // first column whose diagonal tile had a non-positive pivot
#endif
      S.ok = 0;
    }
    *reinterpret_cast<SFT_G v4d*>(W.Linv + (size_t)k * 256 + 4 * lane) = w;
    // W^T in accumulator order: through LDS (element [row][col] at row * 17 + col)
    v4d Wt;
    {
#pragma unroll
      for (int q = 0; q < 4; q++) wscr[(g + 4 * q) * 17 + c] = w[q];
#pragma unroll
      for (int q = 0; q < 4; q++) Wt[q] = wscr[c * 17 + g + 4 * q];
    }
    switch (ph) {
      case 0: wv_step_rest<0>(W, S, Wt, k, lds, lane); break;
      case 1: wv_step_rest<1>(W, S, Wt, k, lds, lane); break;
      case 2: wv_step_rest<2>(W, S, Wt, k, lds, lane); break;
      case 3: wv_step_rest<3>(W, S, Wt, k, lds, lane); break;
      case 4: wv_step_rest<4>(W, S, Wt, k, lds, lane); break;
      case 5: wv_step_rest<5>(W, S, Wt, k, lds, lane); break;
      case 6: wv_step_rest<6>(W, S, Wt, k, lds, lane); break;
      default: wv_step_rest<7>(W, S, Wt, k, lds, lane); break;
    }
  }

#ifdef DSH_LAB
  const long long wv_t2 = clock64();
#endif
  // ---- camera corner: 6 x 6 Cholesky of the Schur complement, forward solve of its right-hand side, x_cam (one lane; 7 x 7 in LDS)
  wv_mfma_fence();
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int r = g + 4 * q;
    if (r < SFT_BORDER && c < SFT_BORDER) Cn[r * 7 + c] = S.corner[q];
  }
#ifdef DSH_LAB
  if (lane < 49) P.dbg[8 + lane] = Cn[lane];   // the Schur complement of the camera before its Cholesky
#endif
  double xcam = 0.0;
  {
    int okc = 1;
    if (lane == 0) {
      double xc[6] = {0, 0, 0, 0, 0, 0};
      bool bad = false;
      for (int kk = 0; kk < 6; kk++) {
        double dd = Cn[kk * 7 + kk];
        for (int j = 0; j < kk; j++) dd -= Cn[kk * 7 + j] * Cn[kk * 7 + j];
        if (!(dd > 0.0)) bad = true;
        const double piv = sqrt(dd);
        Cn[kk * 7 + kk] = piv;
        for (int r = kk + 1; r < 7; r++) {
          double v = Cn[r * 7 + kk];
          for (int j = 0; j < kk; j++) v -= Cn[r * 7 + j] * Cn[kk * 7 + j];
          Cn[r * 7 + kk] = v / piv;
        }
      }
      if (bad) okc = 0;
      for (int kk = 5; kk >= 0; kk--) {
        double v = Cn[6 * 7 + kk];
        for (int r = kk + 1; r < 6; r++) v -= Cn[r * 7 + kk] * xc[r];
        xc[kk] = v / Cn[kk * 7 + kk];
      }
      for (int kk = 0; kk < 6; kk++) Cn[49 + kk] = xc[kk];
    }
    okc = __builtin_amdgcn_readfirstlane(okc);
#ifdef DSH_LAB
    if (lane == 0) P.dbg[4] = okc;
#endif
    if (!okc) S.ok = 0;
    if (c < 6) xcam = Cn[49 + c];
  }
  const int ok = __builtin_amdgcn_readfirstlane(S.ok);
  if (!ok) return 0;     // like g2o, x keeps its previous content when the factorisation failed
  if (lane < 6) P.x[W.Dnp + lane] = xcam;

  // ---- back substitution.  Column J: S_q = sum_d Y_d[q] x_{J+d}[c] + Yb[q] xb[c] summed over the 16 lanes of a row = (L^T x)_tail + camera
  // term - y at index g + 4q;  z = -S;  x_J[c] = sum_r W[r][c] z[r].  The border tile is a ninth tile whose "x" is (x_cam, -1, 0..).
  const double xb = (c < 6) ? xcam : ((c == 6) ? -1.0 : 0.0);
  double xr[8];          // xr[s]: x of tile row with (row mod 8) == s, element c (every lane group holds the 16 values)
#pragma unroll
  for (int s = 0; s < 8; s++) xr[s] = 0.0;
  struct Col { v4d t[10]; };   // [0] Yb, [1..8] Y_d, [9] W
  auto fetch = [&](int J) -> Col {
    Col C;
    if (J >= 0) {
      const SFT_G double* col = W.Lg + ((size_t)J * (BT + 1)) * 256 + 4 * lane;
#pragma unroll
      for (int i = 0; i <= 8; i++) C.t[i] = (i == 0 || J + i < W.nT) ? *reinterpret_cast<const SFT_G v4d*>(col + 256 * i) : (v4d){0.0, 0.0, 0.0, 0.0};
      C.t[9] = *reinterpret_cast<const SFT_G v4d*>(W.Linv + (size_t)J * 256 + 4 * lane);
    } else {
#pragma unroll
      for (int i = 0; i < 10; i++) C.t[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    return C;
  };
  auto solve_col = [&](const Col& C, int J, auto phc) {
    constexpr int ph = decltype(phc)::value;   // J mod 8
    double s[4];
#pragma unroll
    for (int q = 0; q < 4; q++) s[q] = C.t[0][q] * xb;
#pragma unroll
    for (int dd = 1; dd <= 8; dd++)
#pragma unroll
      for (int q = 0; q < 4; q++) s[q] = fma(C.t[dd][q], xr[(ph + dd) & 7], s[q]);
    // all-reduce over the 16 lanes of a row (fixed butterfly: row_ror 8, 4, 2, 1)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      s[q] += dpp_mov<0x128>(s[q]);
      s[q] += dpp_mov<0x124>(s[q]);
      s[q] += dpp_mov<0x122>(s[q]);
      s[q] += dpp_mov<0x121>(s[q]);
    }
    double p = 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++) p = fma(C.t[9][q], -s[q], p);
    p = sum_rows(p);
    xr[ph] = p;
    if (lane < TS) P.x[TS * J + lane] = p;
  };
  {
    int J = W.nT - 1;
    Col c0 = fetch(J), c1 = fetch(J - 1);
#pragma unroll 1
    for (; J >= 0; J -= 2) {   // nT is even (Dnp is a multiple of 32)
      const Col n0 = fetch(J - 2), n1 = fetch(J - 3);
      switch (J & 7) {
        case 7: solve_col(c0, J, std::integral_constant<int, 7>{}); solve_col(c1, J - 1, std::integral_constant<int, 6>{}); break;
        case 5: solve_col(c0, J, std::integral_constant<int, 5>{}); solve_col(c1, J - 1, std::integral_constant<int, 4>{}); break;
        case 3: solve_col(c0, J, std::integral_constant<int, 3>{}); solve_col(c1, J - 1, std::integral_constant<int, 2>{}); break;
        default: solve_col(c0, J, std::integral_constant<int, 1>{}); solve_col(c1, J - 1, std::integral_constant<int, 0>{}); break;
      }
      c0 = n0; c1 = n1;
    }
  }
#ifdef DSH_LAB
  if (lane == 0) { const long long t3 = clock64(); P.dbg[5] = (double)(wv_t1 - wv_t0); P.dbg[6] = (double)(wv_t2 - wv_t1); P.dbg[7] = (double)(t3 - wv_t2); }
#endif
  return 1;
}
