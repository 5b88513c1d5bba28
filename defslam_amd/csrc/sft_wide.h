// Wide-band tile solver of the SfT normal equations (included by sft_kernels.hip inside its anonymous namespace):
// half-bandwidths 128 < kd <= 256 (e.g. the 2000-node stress template, kd = 248), where the trailing window of the
// right-looking tile factorisation (17 x 17 tiles) no longer fits the register file.
//
// LEFT-looking block Cholesky on 16x16 FP64 MFMA tiles, 8 wavefronts, one block column J per iteration:
//   accT(I,J) = H(I,J)^T - sum_{K = I-wb .. J-1} L(J,K) L(I,K)^T        rows I = J .. J+wb and the 7-row camera/rhs border
//   W_J = inverse Cholesky factor of acc(J,J)                            (chol_inv_blocked, tile_chol.h)
//   X(I,J)^T = W_J accT(I,J)                                             (TRSM as a GEMM)
// Every tile is produced once.  Both MFMA operands of the update are taken as they are stored: a register of a tile in
// accumulator layout is a 4x16 / 16x4 operand chunk that contracts over the tile's ROW index, so the factor keeps the
// TRANSPOSE of every L tile (Lt: tile (I,K) holds X(I,K)^T, slot (K, I-K)) and, in this mode, H holds H(I,J)^T in tile
// (I,J).  The tiles of tile row J (the A operands shared by all rows of the column) are staged in LDS one column ahead;
// the tile produced last, X(J+1,J)^T, goes there straight from the registers.  The back substitution contracts over the
// other index, so the TRSM also emits X(I,J) itself (4 more MFMAs per tile) into Lb, block-column layout like tile mode.
// Work per column: (16 + 15 + ... + 1) + 16 border products of 4 MFMAs each + 18 TRSMs of 8 + 13 = 765 MFMAs on 4 SIMDs (14 k
// cycles at 73 cycles per FP64 MFMA); measured 26 k cycles per column on the 2000-node template.
#pragma once

#define WB 16   // most sub-diagonal tiles per block column in wide mode (half-bandwidth <= 256)

__device__ __forceinline__ size_t wtile_off(int tpr, int I, int d) { return ((size_t)I * tpr + d) * (TS * TS); }

#ifdef SFT_WIDE_SC1_PROBE
// A/B probe: L tiles loaded at agent scope (what a tile written by ANOTHER workgroup, possibly on another XCD, would need)
__device__ __forceinline__ v4d wide_ltile_load(const SFT_G double* p) {
  v4d r;
#pragma unroll
  for (int k = 0; k < 4; k++) r[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return r;
}
#else
__device__ __forceinline__ v4d wide_ltile_load(const SFT_G double* p) { return *reinterpret_cast<const SFT_G v4d*>(p); }
#endif

// What one call factors / back-substitutes: the whole node block of a problem (which = -1), or -- two-sided factorisation, SftPart in
// sft_problem.h -- part 0 or 1 (a band matrix whose LAST rows are the separator: the elimination stops after the part's own nS tile
// columns and the rest of the loop only forms the part's Schur contribution to the separator block, the separator x camera border and
// the camera corner, written to the part's exchange buffer in the layout of the reduced problem), or the reduced problem itself
// (which = 2: the summed contributions, a dense band).  Every field is wave-uniform (scalar registers).
struct WideView {
  int nT, nS, tpr, wb;               // tile rows, eliminated tile columns (== nT unless a part), pitch of a tile row, most sub-diagonal tiles
  int lam_lo, lam_hi;                // the damping is added to the diagonal entries lam_lo <= index < lam_hi (identity padding keeps its 1)
  int bstride, b_base, b_sign, b_lo, b_hi;   // border column j of this matrix = Hbord[row * bstride + b_base + b_sign * j] for b_lo <= j < b_hi, else 0
  int xr_tpr, xr_nT;                 // parts: layout of the reduced problem
  bool reversed, corner_from_H, finish;
  const SFT_G double *Hb, *Hbord, *Hcorner;
  SFT_G double *Lb, *Lt, *LbT, *Lbord, *Linv, *x, *xchg;
  SFT_G double *Pf, *PfB;            // parts: the far sums of a block column formed by helper workgroups (factor_wide_helper)
  SFT_G int32_t* sync;               // parts: [0] the owner's progress word, [WIDE_SYNC_READY + J] column J's far sums are in Pf / PfB
};
template <bool UNIFORM = true>
__device__ __forceinline__ WideView wide_view(const SftDev& P, int which_) {
  // (an argument of a non-inlined function arrives in a vector register: without this every field below, selected by branches on it, counts as
  // divergent -- and every comparison with nT, nS, wb in the callers becomes a vector compare and an exec-mask branch.  factor_part and the
  // helpers gain 6 % of a C5 frame from it; factor_wide / backsub_wide, bound by their barriers and tile loads, measured 2-5 % SLOWER with scalar
  // branches -- the predicated code lets the compiler hoist loads across them -- and keep the vector form: UNIFORM = false)
  const int which = UNIFORM ? __builtin_amdgcn_readfirstlane(which_) : which_;
  WideView v;
  const int Dn = uni(P.Dn), Dnp = ((Dn + NB - 1) / NB) * NB;
  if (which < 0) {
    v.nT = Dnp / TS; v.nS = v.nT; v.tpr = uni(P.tpr); v.wb = uni(P.wbt);
    v.lam_lo = 0; v.lam_hi = Dn;
    v.bstride = Dnp; v.b_base = 0; v.b_sign = 1; v.b_lo = 0; v.b_hi = Dnp;
    v.xr_tpr = 0; v.xr_nT = 0; v.reversed = false; v.corner_from_H = true; v.finish = true;
    v.Hb = uni(P.Hb); v.Hbord = uni(P.Hbord); v.Hcorner = uni(P.Hcorner);
    v.Lb = uni(P.Lb); v.Lt = uni(P.Lt); v.LbT = uni(P.LbT); v.Lbord = uni(P.Lbord); v.Linv = uni(P.Linv); v.x = uni(P.x); v.xchg = nullptr;
    v.Pf = nullptr; v.PfB = nullptr; v.sync = nullptr;
    return v;
  }
  const auto& q = P.part[which];
  v.nT = uni(q.nT); v.nS = uni(q.nS); v.tpr = uni(q.tpr); v.wb = uni(q.wbt);
  v.Hb = uni(q.Hb); v.Lb = uni(q.Lb); v.Lt = uni(q.Lt); v.LbT = uni(q.LbT); v.Lbord = uni(q.Lbord); v.Linv = uni(q.Linv); v.x = uni(q.x); v.xchg = uni(q.xchg);
  v.Pf = uni(q.Pf); v.PfB = uni(q.PfB); v.sync = uni(q.sync);
  v.xr_tpr = uni(P.part[2].tpr); v.xr_nT = uni(P.part[2].nT);
  if (which >= 2) {   // the reduced problem (2, or its second copy 3): its input is the summed exchange buffer
    v.lam_lo = 0; v.lam_hi = uni(P.sp_s);
    v.bstride = TS * v.nT; v.b_base = 0; v.b_sign = 1; v.b_lo = 0; v.b_hi = TS * v.nT;
    v.reversed = false; v.corner_from_H = true; v.finish = true;
    v.Hbord = v.xchg + (size_t)v.nT * v.tpr * (TS * TS);
    v.Hcorner = v.Hbord + (size_t)8 * TS * v.nT;
    return v;
  }
  v.lam_lo = which == 1 ? uni(P.sp_pad) : 0; v.lam_hi = TS * v.nS;
  v.bstride = Dnp; v.b_base = uni(q.b_base); v.b_sign = uni(q.b_sign); v.b_lo = uni(q.b_lo); v.b_hi = uni(q.b_hi);
  v.reversed = which == 1; v.corner_from_H = which == 0; v.finish = false;
  v.Hbord = uni(P.Hbord); v.Hcorner = uni(P.Hcorner);
  return v;
}

// ---- Helper workgroups of a part (latency mode: CUs idle anyway).  The products of a block column J are cut at K = J - near:
//   far(I,J)  = sum_{K <  Ks} L(J,K) L(I,K)^T      Ks = min(nS, max(0, J - near)): block columns that were finished `near` columns ago
//   near(I,J) = sum_{K >= Ks} L(J,K) L(I,K)^T      the last `near` block columns: the critical path, always the owner's
//   acc(I,J)  = (H(I,J) - far) - near
// A helper workgroup forms the far sums of whole block columns (rows, diagonal tile, border; factor_wide_helper) on another CU from the
// finished L tiles and leaves them in Pf / PfB; the owner picks them up when they are there and otherwise forms them itself, in the same
// order -- so the result does not depend on whether, when or how many helpers ran (the owner never blocks on one for long; a helper only
// ever waits for the owner).  What crosses CUs (the owner's Lt / LbT tiles, the helpers' far sums, the flags) is stored and loaded at
// agent scope (sc1): the L2 of another XCD holds no stale line of it.
#ifndef SFT_WIDE_NEAR
#define SFT_WIDE_NEAR 4
#ifndef SFT_CHAIN_FIRST
#define SFT_CHAIN_FIRST 1   // the wave of the pivot chain takes its other rows behind the chain (A/B: 22.83 -> 22.69 ms per C5 frame)
#endif
#endif
__device__ __forceinline__ int wide_near(int wb) { return SFT_WIDE_NEAR; }
#define WIDE_SYNC_READY 16          // sync[WIDE_SYNC_READY + J] == epoch: column J's far sums are stored
#ifndef WIDE_OWNER_POLLS
#define WIDE_OWNER_POLLS 24         // how often the owner looks for a helper's column before it forms the far sums itself
#endif
#define WIDE_HELPER_POLLS 400       // a helper that sees no progress of its owner for this many looks (about half a millisecond) leaves: its CU may be what the owner waits for

__device__ __forceinline__ v4d tile_ld_agent(const SFT_G double* p) {
  v4d r;
#pragma unroll
  for (int k = 0; k < 4; k++) r[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return r;
}
__device__ __forceinline__ void tile_st_agent(SFT_G double* p, const v4d& v) {
#pragma unroll
  for (int k = 0; k < 4; k++) __hip_atomic_store(p + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 32 bytes per lane at agent scope through a buffer resource: two 16-byte loads the compiler keeps exact wait counts for
typedef int v4i_w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const SFT_G double* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)(const double*)p, 0, 0xffffffffu, 0x00020000);
}
__device__ __forceinline__ v4d tile_ld_rsrc(__amdgpu_buffer_rsrc_t r, unsigned lane32, unsigned tile_bytes) {
  union { v4i_w i[2]; v4d d; } u;
  u.i[0] = __builtin_amdgcn_raw_buffer_load_b128(r, lane32, tile_bytes, 16);        // aux 16 = sc1
  u.i[1] = __builtin_amdgcn_raw_buffer_load_b128(r, lane32 + 16, tile_bytes, 16);
  return u.d;
}

// a tile's 32 bytes per lane through a buffer resource: the address is (scalar resource, one shared lane offset, scalar tile offset) -- no
// 64-bit per-lane pointer lives in vector registers.  AUX 0: plain, 16: agent scope (sc1)
template <int AUX>
__device__ __forceinline__ v4d gb_ld(__amdgpu_buffer_rsrc_t r, unsigned lane32, unsigned tile_bytes) {
  union { v4i_w i[2]; v4d d; } u;
  u.i[0] = __builtin_amdgcn_raw_buffer_load_b128(r, lane32, tile_bytes, AUX);
  u.i[1] = __builtin_amdgcn_raw_buffer_load_b128(r, lane32 + 16, tile_bytes, AUX);
  return u.d;
}
template <int AUX>
__device__ __forceinline__ void gb_st(__amdgpu_buffer_rsrc_t r, unsigned lane32, unsigned tile_bytes, const v4d& v) {
  union { v4i_w i[2]; v4d d; } u;
  u.d = v;
  __builtin_amdgcn_raw_buffer_store_b128(u.i[0], r, lane32, tile_bytes, AUX);
  __builtin_amdgcn_raw_buffer_store_b128(u.i[1], r, lane32 + 16, tile_bytes, AUX);
}
typedef double v2d_w __attribute__((ext_vector_type(2)));
using lds_v2d_w = __attribute__((address_space(3))) v2d_w;
// Staged tiles live in LDS as two 1 KB planes (registers 0,1 of every lane, then registers 2,3): every ds_read_b128 /
// ds_write_b128 touches 64 consecutive 16-byte words.
__device__ __forceinline__ v4d wide_lds_read(const lds_double* tile, int lane) {
  const v2d_w lo = *reinterpret_cast<const lds_v2d_w*>(tile + 2 * lane);
  const v2d_w hi = *reinterpret_cast<const lds_v2d_w*>(tile + 128 + 2 * lane);
  return (v4d){lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void wide_lds_write(lds_double* tile, int lane, const v4d& v) {
  *reinterpret_cast<lds_v2d_w*>(tile + 2 * lane) = (v2d_w){v[0], v[1]};
  *reinterpret_cast<lds_v2d_w*>(tile + 128 + 2 * lane) = (v2d_w){v[2], v[3]};
}
__device__ __forceinline__ void wide_mfma4(const v4d& a, const v4d& b, v4d& s0, v4d& s1) {
  s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], s0, 0, 0, 0);
  s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], s1, 0, 0, 0);
  s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], s0, 0, 0, 0);
  s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], s1, 0, 0, 0);
}
#define WIDE_U 4
// sum_{K = K0 .. Kend-1} L(J,K) L(I,K)^T for one row: A operand = row-J tile (LDS, slot J-K-1), B operand = the stored tile of row I, ldB(i) =
// the tile of block column K0 + i.  The FP64 MFMA pipe bounds the products (two waves per SIMD: 573 cycles per product), so the tile loads
// must not add their latency to it: NCH chunks of WIDE_U products, fully unrolled (straight-line code keeps exact wait counts), the loads of
// chunk c+1 issued before the MFMAs of chunk c.  Every chunk issues WIDE_U loads (the last tile again beyond the row's end).  The order of
// the sum (two accumulators, K ascending) does not depend on NCH.
template <int NCH, class LD>
__device__ __forceinline__ v4d wide_products_n(int lane, int J, int K0, int Kend, const lds_double* rowJ, LD ldB) {
  constexpr int U = WIDE_U;
  v4d s0 = {0.0, 0.0, 0.0, 0.0}, s1 = {0.0, 0.0, 0.0, 0.0};
  v4d buf[2][U];
  const int last = max(Kend - 1 - K0, 0);
#pragma unroll
  for (int u = 0; u < U; u++) buf[0][u] = ldB(min(u, last));
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if (c + 1 < NCH) {
#pragma unroll
      for (int u = 0; u < U; u++) buf[(c + 1) & 1][u] = ldB(min(U * (c + 1) + u, last));
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int K = K0 + U * c + u;
      if (K < Kend) wide_mfma4(wide_lds_read(rowJ + (size_t)(J - K - 1) * TS * TS, lane), buf[c & 1][u], s0, s1);
    }
  }
  return s0 + s1;
}
template <class LD>
__device__ __forceinline__ v4d wide_products(int lane, int J, int K0, int Kend, const lds_double* rowJ, LD ldB) {
  const int n = Kend - K0;
  if (n <= 0) return (v4d){0.0, 0.0, 0.0, 0.0};
  if (n <= WIDE_U) return wide_products_n<1>(lane, J, K0, Kend, rowJ, ldB);
  if (n <= 2 * WIDE_U) return wide_products_n<2>(lane, J, K0, Kend, rowJ, ldB);
  return wide_products_n<4>(lane, J, K0, Kend, rowJ, ldB);
}
// sum_{i < n} T_i T_i^T with T_i = ld(i) as both operands: the diagonal tile's products, formed from the stored tiles
template <int NCH, class LD>
__device__ __forceinline__ v4d wide_squares_n(int n, LD ld) {
  constexpr int U = WIDE_U;
  v4d s0 = {0.0, 0.0, 0.0, 0.0}, s1 = {0.0, 0.0, 0.0, 0.0};
  v4d buf[2][U];
  const int last = max(n - 1, 0);
#pragma unroll
  for (int u = 0; u < U; u++) buf[0][u] = ld(min(u, last));
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if (c + 1 < NCH) {
#pragma unroll
      for (int u = 0; u < U; u++) buf[(c + 1) & 1][u] = ld(min(U * (c + 1) + u, last));
    }
#pragma unroll
    for (int u = 0; u < U; u++)
      if (U * c + u < n) wide_mfma4(buf[c & 1][u], buf[c & 1][u], s0, s1);
  }
  return s0 + s1;
}
template <class LD>
__device__ __forceinline__ v4d wide_squares(int n, LD ld) {
  if (n <= 0) return (v4d){0.0, 0.0, 0.0, 0.0};
  if (n <= WIDE_U) return wide_squares_n<1>(n, ld);
  return wide_squares_n<4>(n, ld);
}

#ifdef SFT_WIDE_TRACE   // A/B builds: where a block column's time goes, per role (100 MHz ticks summed over the columns of the last factorisation)
#define WT_T0() long long wt_t__ = wall_clock64()
#define WT_SEG(seg) do { const long long t1__ = wall_clock64(); if (lane == 0) wtrace[d * 8 + (seg)] += (double)(t1__ - wt_t__); wt_t__ = t1__; } while (0)
#else
#define WT_T0() do {} while (0)
#define WT_SEG(seg) do {} while (0)
#endif
// (A part sums far and near products separately, like factor_part with its helpers: the two functions agree bit for bit.)
__device__ __noinline__ void factor_wide(const SftDev& P, int which, Ctl* ctl, double* ws) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WideView V = wide_view<false>(P, which);
  const int nT = V.nT, nS = V.nS;
  const int tpr = V.tpr, wb = V.wb;
  const int Dnp = TS * nT;
  lds_double* Lrow = to_lds(ws);                       // 2 x WB tiles, native layout (lane, 4 doubles): row-J tiles, dist 1..wb
  lds_double* LinvK = Lrow + 2 * WB * TS * TS;         // W_J, k-major padded: LinvK[k*TP + j] = W[j][k]
  lds_double* Cn = LinvK + TILE_LDS;                   // 8 partial 7x7 corners, then the corner itself
  const double lambda = ctl->lambda;
  const int crow = lane >> 4, ccol = lane & 15;
  const auto Hg = V.Hb;
  const auto Hbord = V.Hbord;
  const auto Lg = V.Lb;
  const auto Ltg = V.Lt;
  const auto LbTg = V.LbT;
  const auto Lbord = V.Lbord;
  const auto Linv_g = V.Linv;
  const bool is_part = which == 0 || which == 1;
  const int near = wide_near(wb);
  // first block column of the near products of column J (the far ones end there); everything is "near" unless this is a part
  auto ksplit = [&](int J) -> int { return is_part ? min(nS, max(0, J - near)) : 0; };
  // border element (row, column j of this matrix) of H
  auto bord_h = [&](int row, int j) -> double {
    const int jj = (j >= V.b_lo && j < V.b_hi) ? j : V.b_lo;
    const double v = Hbord[(size_t)row * V.bstride + V.b_base + V.b_sign * jj];
    return (j >= V.b_lo && j < V.b_hi) ? v : 0.0;
  };
  v4d cacc = {0.0, 0.0, 0.0, 0.0};     // this wave's share of the corner updates; wave 0 starts from H_cc + lambda I
  if (wave == 0 && V.corner_from_H) {
    const double lam_c = which >= 2 ? 0.0 : lambda;   // the reduced corner already carries the damping (part 0 added it)
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER && ccol <= r) cacc[q] = V.Hcorner[r * 7 + ccol] + ((r == ccol && r < 6) ? lam_c : 0.0);
    }
  }
  if (tid == 0) ctl->fact_ok = 1;
#ifdef SFT_WIDE_TRACE
  lds_double* wtrace = Cn + 580;          // (the A/B build gets 64 doubles more of LDS: sft_lm_kernel_lds_bytes)
  if (tid < 64) wtrace[tid] = 0.0;
#endif
  __syncthreads();

  const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
  const long kstride = (long)(tpr - 1) * TS * TS;        // tile (I, K+1) sits (tpr - 1) tiles after tile (I, K)
  // tiles of this workgroup's own factor: plain loads
  auto own_row = [&](int I, int K0) { const SFT_G double* brow = Ltg + wtile_off(tpr, K0, I - K0) + 4 * lane; return [=](int i) -> v4d { return wide_ltile_load(brow + (long)i * kstride); }; };
  auto own_brd = [&](int K0) { const SFT_G double* brow = LbTg + (size_t)K0 * TS * TS + 4 * lane; return [=](int i) -> v4d { return wide_ltile_load(brow + (long)i * (TS * TS)); }; };
  // Schur contribution of a part: tile (I, J) of the separator block, transposed tile in accumulator layout like H's tiles, into the
  // exchange buffer (layout of the reduced problem's H).  Part 1 runs in reversed order: element (i, j) of its separator block is
  // element (s-1-j, s-1-i) of the natural one.
  auto schur_store = [&](int I, int J, const v4d& t) {
    const int Ir = I - nS, Jr = J - nS;
    if (!V.reversed) {
      *reinterpret_cast<SFT_G v4d*>(V.xchg + wtile_off(V.xr_tpr, Ir, Ir - Jr) + 4 * lane) = t;
      return;
    }
    // t[q] = T[b][a] with a = crow + 4 q (column index inside tile J), b = ccol (row index inside tile I)
    const int It = V.xr_nT - 1 - Jr, Jt = V.xr_nT - 1 - Ir;          // natural tile (It, Jt), It >= Jt
    const auto dst = V.xchg + wtile_off(V.xr_tpr, It, It - Jt);
#pragma unroll
    for (int q = 0; q < 4; q++) dst[tile_elem(15 - ccol, 15 - (crow + 4 * q))] = t[q];
  };
  // Look-ahead: the diagonal tile of column J+1 minus its products with block columns <= J-1 is formed during column J by the
  // wave that owns column J+1 next (roles rotate by one wave per column) and waits in its registers.
  v4d dlook = *reinterpret_cast<const SFT_G v4d*>(Hg + 4 * lane);     // column 0: H(0,0), no products

#pragma unroll 1
  for (int J = 0; J < nT; J++) {
    lds_double* rowJ = Lrow + (size_t)(J & 1) * WB * TS * TS;
    lds_double* rowN = Lrow + (size_t)((J + 1) & 1) * WB * TS * TS;
    const int Kend = min(J, nS), Ks = ksplit(J);
    // Roles rotate with the column.  The FP64 MFMA pipe is what the products are bound by (73 cycles per MFMA, two waves per
    // SIMD: tools/probes/lds_mfma_probe.hip), so the rows are dealt by their number of products, and the Cholesky is taken
    // off the path every wave waits for: role 0 (owner of column J) starts from the look-ahead tile, adds the one product
    // with block column J-1, factors, and then does two short rows; role 1 (owner of column J+1) forms the look-ahead tile of
    // the next column (15 products, tiles from memory) and two short rows; the others take about 20 products each:
    //   0: diag, J+10, J+12 (+ product-free J+16) | 1: look-ahead, J+13, J+14 | 2: J+1, J+11 | 3: J+2, J+9 | 4: J+3, J+8
    //   5: J+4, J+7 | 6: J+5, J+6 | 7: border, J+15
    const int d = (wave - J) & 7;
    WT_T0();
    int Irow[3];
    Irow[0] = J + (d == 0 ? 10 : d == 1 ? 13 : d == 7 ? 15 : d - 1);
    Irow[1] = J + (d == 0 ? 12 : d == 1 ? 14 : d == 2 ? 11 : d == 3 ? 9 : d == 4 ? 8 : d == 5 ? 7 : d == 6 ? 6 : 99);
    Irow[2] = d == 0 ? J + 16 : nT;
    if (d == 7) Irow[1] = nT;
    // stage tile row J+1 for the next column: dist 2 + wave and 10 + wave (dist 1 is produced by this column's TRSM)
    v4d stage[2];
    bool staged[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int dist = 2 + wave + 8 * h, K = J + 1 - dist;
      staged[h] = dist <= wb && K >= 0 && J + 1 < nT;
      if (staged[h]) stage[h] = wide_ltile_load(Ltg + wtile_off(tpr, K, dist) + 4 * lane);
    }
    const bool elim = J < nS;          // a separator column of a part is not eliminated: its tiles are the Schur contribution
    if (d == 0) {
      // the diagonal tile: look-ahead tile minus the product with block column J-1 (the staged tile (J, J-1) twice), Cholesky
      __builtin_amdgcn_s_setprio(3);
      v4d dt = dlook;
      if (J >= 1 && J - 1 < nS) {
        const v4d a = wide_lds_read(rowJ, lane);
        v4d s0 = zero4, s1 = zero4;
        wide_mfma4(a, a, s0, s1);
        dt -= s0 + s1;
      }
      if (elim) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (crow + 4 * q == ccol && TS * J + ccol >= V.lam_lo && TS * J + ccol < V.lam_hi) dt[q] += lambda;
        v4d w = dt;
        const bool ok = chol_inv_blocked(dt, w);
        if (!ok && lane == 0) ctl->fact_ok = 0;
        lds_double* dst = LinvK + ccol * TP + crow;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[4 * q] = w[q];
        *reinterpret_cast<SFT_G v4d*>(Linv_g + (size_t)J * TS * TS + 4 * lane) = w;
      } else {
        schur_store(J, J, dt);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    WT_SEG(0);
    v4d accT[3];
    bool have[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int I = Irow[t];
      have[t] = I < nT && I - J <= wb;
      if (have[t]) {
        const v4d h = *reinterpret_cast<const SFT_G v4d*>(Hg + wtile_off(tpr, I, I - J) + 4 * lane);
        const int K0 = max(0, I - wb), Kn = max(K0, Ks);
        v4d far = zero4;
        if (Ks > K0) far = wide_products(lane, J, K0, Ks, rowJ, own_row(I, K0));
        accT[t] = (h - far) - wide_products(lane, J, Kn, Kend, rowJ, own_row(I, Kn));
      }
    }
    WT_SEG(1);
    if (d == 1 && J + 1 < nT) {
      // look-ahead for column J+1: H(J+1,J+1) - sum_{K <= J-1} L(J+1,K) L(J+1,K)^T  (the product with block column J follows
      // in the next column, when tile (J+1, J) exists)
      const int I = J + 1, K0 = max(0, I - wb), Ks1 = ksplit(I), Kn = max(K0, Ks1);
      const v4d h = *reinterpret_cast<const SFT_G v4d*>(Hg + wtile_off(tpr, I, 0) + 4 * lane);
      v4d far = zero4;
      if (Ks1 > K0) far = wide_squares(Ks1 - K0, own_row(I, K0));
      dlook = (h - far) - wide_squares(Kend - Kn, own_row(I, Kn));
    }
    WT_SEG(2);
    // border: accTb[j][i] = Hbord[i][16 J + j] - sum_K (L(J,K) Lb(K)^T)[j][i], i < 7
    v4d accTb = zero4;
    const bool bwave = d == 7;
    if (bwave) {
      v4d h = zero4;
      if (ccol < SFT_BORDER) {
#pragma unroll
        for (int q = 0; q < 4; q++) h[q] = bord_h(ccol, TS * J + crow + 4 * q);
      }
      const int K0 = max(0, J - wb), Kn = max(K0, Ks);
      v4d far = zero4;
      if (Ks > K0) far = wide_products(lane, J, K0, Ks, rowJ, own_brd(K0));
      accTb = (h - far) - wide_products(lane, J, Kn, Kend, rowJ, own_brd(Kn));
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (staged[h]) wide_lds_write(rowN + (size_t)(1 + wave + 8 * h) * TS * TS, lane, stage[h]);
    WT_SEG(3);
    lds_barrier();                                       // W_J is published
    WT_SEG(4);
    // ---- TRSM: X^T = W accT (kept for the factor), X = acc W^T (kept for the back substitution) ----
    if (!elim) {
      // separator column of a part: the raw tiles are its Schur contribution (separator block, separator x camera/rhs border)
#pragma unroll
      for (int t = 0; t < 3; t++)
        if (have[t] && Irow[t] != J) schur_store(Irow[t], J, accT[t]);
      if (bwave && ccol < SFT_BORDER) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int jl = TS * (J - nS) + crow + 4 * q;                    // column inside the separator block, this part's order
          V.xchg[(size_t)V.xr_nT * V.xr_tpr * (TS * TS) + (size_t)ccol * (TS * V.xr_nT) + (V.reversed ? TS * V.xr_nT - 1 - jl : jl)] = accTb[q];
        }
      }
      __syncthreads();
      continue;
    }
    double wv[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) wv[kk] = LinvK[(4 * kk + crow) * TP + ccol];     // lane (x = ccol, k = crow): W[x][4kk + k]
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int I = Irow[t];
      if (!have[t] || I == J) continue;
      v4d xT = zero4, x = zero4;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        xT = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[kk], accT[t][kk], xT, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(accT[t][kk], wv[kk], x, 0, 0, 0);
      }
      *reinterpret_cast<SFT_G v4d*>(Ltg + wtile_off(tpr, J, I - J) + 4 * lane) = xT;
      *reinterpret_cast<SFT_G v4d*>(Lg + wtile_off(tpr, J, I - J) + 4 * lane) = x;
      if (I == J + 1) wide_lds_write(rowN, lane, xT);
    }
    if (bwave) {
      v4d xbT = zero4;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) xbT = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[kk], accTb[kk], xbT, 0, 0, 0);
      *reinterpret_cast<SFT_G v4d*>(LbTg + (size_t)J * TS * TS + 4 * lane) = xbT;
      if (ccol < SFT_BORDER) {
#pragma unroll
        for (int q = 0; q < 4; q++) Lbord[(size_t)ccol * Dnp + TS * J + crow + 4 * q] = xbT[q];
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(xbT[kk], -xbT[kk], cacc, 0, 0, 0);
    }
    WT_SEG(5);
    WT_SEG(6);
    __syncthreads();                                     // the column's tiles are in memory (and in rowN) for the next one
    WT_SEG(7);
  }
#ifdef SFT_WIDE_TRACE
  __syncthreads();
  if (is_part && tid < 64) P.dbg[64 * which + tid] = wtrace[tid];
#endif
  // ---- corner: partial sums of the eight waves in a fixed order, then the 6x6 Schur complement of the camera ----
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int r = crow + 4 * q;
    if (r < SFT_BORDER && ccol < SFT_BORDER) Cn[64 * wave + r * 7 + ccol] = cacc[q];
  }
  __syncthreads();
  if (tid < 49) {
    double s = 0.0;
    for (int w = 0; w < 8; w++) s += Cn[64 * w + tid];
    Cn[512 + tid] = s;
  }
  __syncthreads();
  if (!V.finish) {   // a part: its corner contribution (part 0 started from H_cc + lambda I) and whether its factorisation failed
    const auto xc = V.xchg + (size_t)V.xr_nT * V.xr_tpr * (TS * TS) + (size_t)8 * TS * V.xr_nT;
    if (tid < 49) xc[tid] = Cn[512 + tid];
    if (tid == 0) xc[56] = ctl->fact_ok ? 0.0 : 1.0;
    __syncthreads();
    return;
  }
  if (tid == 0) {
    lds_double* C = Cn + 512;
    bool bad = false;
    for (int k = 0; k < 6; k++) {
      double dd = C[k * 7 + k];
      for (int j = 0; j < k; j++) dd -= C[k * 7 + j] * C[k * 7 + j];
      if (!(dd > 0.0)) bad = true;
      const double piv = sqrt(dd);
      C[k * 7 + k] = piv;
      for (int r = k + 1; r < 7; r++) {
        double v = C[r * 7 + k];
        for (int j = 0; j < k; j++) v -= C[r * 7 + j] * C[k * 7 + j];
        C[r * 7 + k] = v / piv;
      }
    }
    if (bad) ctl->fact_ok = 0;
    if (ctl->fact_ok)
      for (int k = 5; k >= 0; k--) {
        double v = C[6 * 7 + k];
        for (int r = k + 1; r < 6; r++) v -= C[r * 7 + k] * V.x[Dnp + r];
        V.x[Dnp + k] = v / C[k * 7 + k];
      }
  }
  __syncthreads();
}

template <int NW> __device__ __noinline__ void wide_far_column_of(const SftDev& P, int which, int J, lds_double* rowJ);   // (the view is rebuilt there: the caller's stays in registers)

// ------------------------------------------------------------------------------------------------------------------------------------
// The owner of a part when helper workgroups deliver the far sums (latency mode; the arithmetic, order for order, of factor_wide(P, part)
// -- the two are interchangeable bit for bit).  What is left to the owner are the products with the last NEAR block columns, and their
// operands never leave the CU:
//   * rows are owned by waves (row I on wave I mod 8: two live rows per wave, three on the pivot wave), and a wave keeps the last NEAR
//     tiles of its rows -- its own TRSM results -- in registers (ring Lr, shifted by one tile per column): the B operands;
//   * the tiles of a row that is about to become the pivot row (distance <= NEAR) also go to an LDS ring (row mod 8, column mod NEAR): the A
//     operands of every wave NEAR columns later; the border tiles of the last NEAR columns likewise;
//   * the far sum arrives as ONE tile per row, H(I,J)^T - far(I,J), formed by the helper and requested a column ahead;
//   * the pivot chain runs inside one wave, a column ahead: the wave that owns row J+1 takes tile (J+1, J) out of its TRSM straight into the
//     diagonal tile of column J+1, factors it and publishes W_{J+1} while the other waves finish column J -- so a column is products, TRSM and
//     ONE barrier, and nobody waits for a Cholesky.
// LDS: 8 x NEAR + NEAR staged tiles, two W, the corner partials.
// The rare paths of factor_part, out of line (their per-lane address arithmetic would otherwise be hoisted out of the column loop and held --
// or spilled -- across it): a separator column's tiles into the exchange buffer, the border tile gathered from H's border rows.
__device__ __noinline__ void wide_schur_store_of(const SftDev& P, int which, int I, int J, v4d t) {
  const WideView V = wide_view(P, which);
  const int lane = threadIdx.x & 63, crow = lane >> 4, ccol = lane & 15;
  const int Ir = I - V.nS, Jr = J - V.nS;
  if (!V.reversed) {
    *reinterpret_cast<SFT_G v4d*>(V.xchg + wtile_off(V.xr_tpr, Ir, Ir - Jr) + 4 * lane) = t;
    return;
  }
  // t[q] = T[b][a] with a = crow + 4 q (column index inside tile J), b = ccol (row index inside tile I); part 1 runs in reversed order
  const int It = V.xr_nT - 1 - Jr, Jt = V.xr_nT - 1 - Ir;          // natural tile (It, Jt), It >= Jt
  const auto dst = V.xchg + wtile_off(V.xr_tpr, It, It - Jt);
#pragma unroll
  for (int q = 0; q < 4; q++) dst[tile_elem(15 - ccol, 15 - (crow + 4 * q))] = t[q];
}
__device__ __noinline__ void wide_schur_border_of(const SftDev& P, int which, int J, v4d t) {
  const WideView V = wide_view(P, which);
  const int lane = threadIdx.x & 63, crow = lane >> 4, ccol = lane & 15;
  if (ccol >= SFT_BORDER) return;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int jl = TS * (J - V.nS) + crow + 4 * q;                    // column inside the separator block, this part's order
    V.xchg[(size_t)V.xr_nT * V.xr_tpr * (TS * TS) + (size_t)ccol * (TS * V.xr_nT) + (V.reversed ? TS * V.xr_nT - 1 - jl : jl)] = t[q];
  }
}
__device__ __noinline__ v4d wide_bord_tile_of(const SftDev& P, int which, int J) {
  const WideView V = wide_view(P, which);
  const int lane = threadIdx.x & 63, crow = lane >> 4, ccol = lane & 15;
  v4d h = {0.0, 0.0, 0.0, 0.0};
  if (ccol < SFT_BORDER) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int j = TS * J + crow + 4 * q;
      const bool in = j >= V.b_lo && j < V.b_hi;
      const double v = V.Hbord[(size_t)ccol * V.bstride + V.b_base + V.b_sign * (in ? j : V.b_lo)];
      h[q] = in ? v : 0.0;
    }
  }
  return h;
}

// One tile, global memory -> LDS without passing registers (LDS-DMA): two requests of 64 x 16 bytes; the lane's 32 bytes of the tile land as the
// two planes wide_lds_read expects.  Not known to the compiler's wait counters: whoever reads the tile waits for vmcnt itself.
template <bool AGENT>
__device__ __forceinline__ void wide_dma_tile(lds_double* dst, const SFT_G double* tile_, unsigned lane32) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst);
  const SFT_G double* tile = uni(tile_);
  if (AGENT)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 sc1\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 sc1"
                 :: "s"(la), "s"(tile), "v"(lane32), "v"(lane32 + 16u) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1"
                 :: "s"(la), "s"(tile), "v"(lane32), "v"(lane32 + 16u) : "memory", "m0");
}

// NW: wavefronts of the workgroup.  8: the layout above.  16 (sft_part_factor_kernel, 128 registers per lane): row I on wave I mod 16 -- ONE live
// row per wave, the pivot wave holds none but the entering one -- so four wavefronts per SIMD interleave where two did: what bounds a column is
// the latency of its dependent steps, not what the SIMDs could issue.  Same arithmetic, same order: the two are interchangeable bit for bit
// (the corner's eight partial sums live in LDS there: the border role visits sixteen waves, the partials are numbered by column).
template <int NEAR, int NW = 8>
__device__ __noinline__ void factor_part(const SftDev& P, int which_, Ctl* ctl, double* ws, int epoch_, int nh_) {
  static_assert(NW == 8 || NW == 16, "factor_part: 8 or 16 wavefronts");
  constexpr int RPW = 16 / NW;                       // live rows of a wave (besides the entering one on the pivot wave)
  constexpr int SE = NW == 16 ? 0 : RPW;             // landing slot of the entering row (16 waves: the pivot wave's row slot is free)
  constexpr int SX = NW == 16 ? 1 : RPW + 1;         // ... of the diagonal / border tile
  constexpr int SLOTS = SX + 1;
  const int which = __builtin_amdgcn_readfirstlane(which_), epoch = __builtin_amdgcn_readfirstlane(epoch_), nh = __builtin_amdgcn_readfirstlane(nh_);   // (arguments arrive in vector registers)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WideView V = wide_view(P, which);
  const int nT = V.nT, nS = V.nS;
  const int tpr = V.tpr, wb = V.wb;
  const int Dnp = TS * nT;
  // [(I % (NEAR + 1)) * NEAR + K % NEAR]: X(I,K)^T of the rows within NEAR of becoming the pivot row (NEAR + 1 row slots: the border role reads
  // all NEAR tiles of the pivot row while the row NEAR below it already writes its first)
  lds_double* Aring = to_lds(ws);
  lds_double* Bring = Aring + (NEAR + 1) * NEAR * TS * TS;       // [K % NEAR]: the border tile of block column K
  lds_double* Wbuf = Bring + NEAR * TS * TS;               // [J & 1]: W_J, k-major padded: [k*TP + j] = W[j][k]
  lds_double* Cn = Wbuf + 2 * TILE_LDS;                    // 8 partial 7x7 corners, then the corner itself
  lds_int* hflag = (lds_int*)(Cn + 576);                   // [c & 3]: the far sums of block column c have been delivered by a helper
  lds_double* Land = Cn + 704;                             // [wave * 4 + slot]: where the start tiles of a wave's next column land (slot 3: border / diagonal tile)
  // the staging area of wide_far_column when the owner forms a column itself: far tiles sit in its slots NEAR .. wb-1, the area starts there
  lds_double* Fstage = Land + 32 * TS * TS - NEAR * TS * TS;
  // the diagonal tile of the next block column on its way from the wave that forms it to the wave that factors it (below), and its flag
  lds_double* Dt = Land + (32 + WB - NEAR) * TS * TS;
  lds_int* dflag = (lds_int*)(Cn + 648);
  lds_double* myland = Land + (size_t)wave * SLOTS * TS * TS;
  const double lambda = ctl->lambda;
  const int crow = lane >> 4, ccol = lane & 15;
  const auto Hg = V.Hb;
  const auto Hbord = V.Hbord;
  const auto Lg = V.Lb;
  const auto Ltg = V.Lt;
  const auto LbTg = V.LbT;
  const auto Lbord = V.Lbord;
  const auto Linv_g = V.Linv;
  const bool helped = nh > 0 && epoch > 0 && V.sync != nullptr;
  const __amdgpu_buffer_rsrc_t rLt = wide_rsrc(Ltg), rLb = wide_rsrc(Lg), rLbT = wide_rsrc(LbTg), rLinv = wide_rsrc(Linv_g);
  const unsigned lane32 = 32u * lane;
  auto toff = [&](int I, int dd) -> unsigned { return (unsigned)(((unsigned)I * (unsigned)tpr + (unsigned)dd) * (TS * TS * 8u)); };
  const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
  auto ksplit = [&](int J) -> int { return min(nS, max(0, J - NEAR)); };
  auto bord_tile = [&](int J) -> v4d { return wide_bord_tile_of(P, which, J); };
  auto schur_store = [&](int I, int J, const v4d& t) { wide_schur_store_of(P, which, I, J, t); };
  // The tile a row starts a column from -- H(I,J)^T - far(I,J) as a helper (or the fallback below) left it, or H(I,J)^T where no far product
  // exists -- is requested into the wave's landing slot a column ahead
  auto request_tile = [&](int slot, int I, int J) {
    if (I >= nT || I - J > wb) return;
    if (ksplit(J) > max(0, I - wb)) wide_dma_tile<true>(myland + slot * TS * TS, V.Pf + wtile_off(tpr, J, I - J), lane32);
    else wide_dma_tile<false>(myland + slot * TS * TS, Hg + wtile_off(tpr, I, I - J), lane32);
  };
  auto request_border = [&](int J) {
    if (ksplit(J) > max(0, J - wb)) wide_dma_tile<true>(myland + SX * TS * TS, V.PfB + (size_t)J * TS * TS, lane32);
    else wide_lds_write(myland + SX * TS * TS, lane, bord_tile(J));
  };
  v4d cacc = zero4;
  if (wave == 0 && V.corner_from_H) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER && ccol <= r) cacc[q] = V.Hcorner[r * 7 + ccol] + ((r == ccol && r < 6) ? lambda : 0.0);
    }
  }
  if (NW == 16 && wave == 0) {   // the eight partial corners, numbered like the eight waves of the other layout: column J adds to (J + 7) & 7
#pragma unroll
    for (int p = 0; p < 8; p++) Cn[64 * p + lane] = 0.0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER) Cn[r * 7 + ccol] = cacc[q];
    }
  }
  if (tid == 0) { ctl->fact_ok = 1; *dflag = 0; }
  if (tid < 4) hflag[tid] = 0;
#ifdef SFT_WIDE_TRACE
  lds_double* wtrace = NW == 16 ? Land + 32 * TS * TS : Cn + 580;   // (16 waves: 128 entries, in the fallback's staging area -- a trace build's numbers are void if the fallback runs)
  if (tid < 8 * NW) wtrace[tid] = 0.0;
  const long long wt_c0 = clock64(), wt_w0 = wall_clock64();
#endif
  __syncthreads();
  int misses = 0;
#ifdef DSH_LAB
  int st_got = 0, st_miss = 0, st_polls = 0;
  long long st_wait = 0;
#endif
  auto look_for_helper = [&](int J) {
    if (!helped || tid != 0) return;
    const int c = J + 2;
    int got = 0;
    if (c < nT && ksplit(c) - max(0, c - wb) > 0) {
#ifdef DSH_LAB
      const long long t0 = wall_clock64();
#endif
      const int polls = misses >= 2 ? 1 : WIDE_OWNER_POLLS;
      for (int i = 0; i < polls; i++) {
#ifdef DSH_LAB
        st_polls++;
#endif
        if (__hip_atomic_load(V.sync + WIDE_SYNC_READY + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) { got = 1; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      misses = got ? 0 : misses + 1;
#ifdef DSH_LAB
      st_wait += wall_clock64() - t0;
      if (got) st_got++; else st_miss++;
#endif
    }
    hflag[c & 3] = got;
  };
  // diagonal tile -> W (or, in a separator column, the part's Schur contribution): the pivot chain of block column Jp
  auto pivot = [&](int Jp, v4d dt) {
    if (Jp < nS) {
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (crow + 4 * q == ccol && TS * Jp + ccol >= V.lam_lo && TS * Jp + ccol < V.lam_hi) dt[q] += lambda;
      v4d w = dt;
      const bool ok = chol_inv_blocked(dt, w);
      if (!ok && lane == 0) ctl->fact_ok = 0;
      lds_double* dst = Wbuf + (size_t)(Jp & 1) * TILE_LDS + ccol * TP + crow;
#pragma unroll
      for (int q = 0; q < 4; q++) dst[4 * q] = w[q];
      gb_st<0>(rLinv, lane32, (unsigned)Jp * (TS * TS * 8u), w);
    } else {
      schur_store(Jp, Jp, dt);
    }
  };
  // ---- prologue: the tiles of column 0 and W_0
  v4d Lr[RPW][NEAR];                 // ring of the wave's rows: at column J entry k is X(I, J - NEAR + k)^T
#pragma unroll
  for (int s = 0; s < RPW; s++)
#pragma unroll
    for (int k = 0; k < NEAR; k++) Lr[s][k] = zero4;
  // Sums over the block columns that were finished before the last barrier are formed a column ahead (pre-products): behind the barrier
  // a row only waits for ONE product -- with the tile of row J received in the column before -- and its TRSM
  v4d pre0[RPW], pre1[RPW], preS = zero4;
#pragma unroll
  for (int s = 0; s < RPW; s++) { pre0[s] = zero4; pre1[s] = zero4; }
  constexpr int S0 = NW == 16 ? SX : 0;   // where the first diagonal tile lands (16 waves: slot 0 of wave 0 takes the entering row)
  if (NW == 16 && wave == 0) {
    request_tile(S0, 0, 0);
  } else {
#pragma unroll
    for (int s = 0; s < RPW; s++) request_tile(s, wave + NW * s, 0);
  }
  if (wave == 0) request_tile(SE, 16, 0);
  if (wave == NW - 1) request_border(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0) -- as a builtin: the compiler's own wait counters take note */
  if (wave == 0) pivot(0, wide_lds_read(myland + S0 * TS * TS, lane));
  __syncthreads();

#pragma unroll 1
  for (int J = 0; J < nT; J++) {
    const int d = (wave - J) & (NW - 1);
    const int Kend = min(J, nS), Ks = ksplit(J);
    const bool elim = J < nS;
    // rows of this column (the last entry: the row that enters the band, on the pivot wave) and of the next (slot by slot: behind the pivot row
    // a wave's rows move up one slot)
    int Irow[RPW + 1], nI[RPW + 1];
#pragma unroll
    for (int s = 0; s < RPW; s++) {
      Irow[s] = J + d + NW * s;
      nI[s] = d == 0 ? (s + 1 < RPW ? J + NW * (s + 1) : J + 16) : ((d == 1 && s == 0) ? nT : J + d + NW * s);
    }
    Irow[RPW] = d == 0 ? J + 16 : nT;
    nI[RPW] = d == 1 ? J + 17 : nT;
    auto slot_of = [&](int s) -> int { return s < RPW ? s : SE; };
    WT_T0();
    // (1) a column no helper has delivered (the verdict on column J+1 fell before the last barrier): the workgroup forms it itself, now
    if (J + 1 < nT && ksplit(J + 1) > max(0, J + 1 - wb) && !uni(hflag[(J + 1) & 3])) wide_far_column_of<NW>(P, which, J + 1, Fstage);
    // (2) the next pivot's diagonal tile: requested now (slot 3), used behind this column's first TRSM
    if (d == 1 && J + 1 < nT) request_tile(SX, J + 1, J + 1);
    WT_SEG(0);
    // (3) the last near product (block column J-1: its tile of row J arrived with the barrier), then the row's tile
    const lds_double* Arow = Aring + (size_t)(J % (NEAR + 1)) * NEAR * TS * TS;
    v4d cur[RPW + 1], curB = zero4;
#pragma unroll
    for (int s = 0; s <= RPW; s++) cur[s] = zero4;
#ifdef SFT_CHAIN_PRIO
    if (d <= 1) __builtin_amdgcn_s_setprio(SFT_CHAIN_PRIO);   // the two waves of the pivot chain go first on their SIMDs (A/B, priority 3: 22.95 against 22.83 ms -- not used)
#endif
    auto t3_row = [&](int s) {
      const int K = J - 1;
      const bool kin = K >= 0 && K < Kend;
      const int I = Irow[s];
      if ((s == 0 && d == 0) || I >= nT || I - J > wb) return;
      const v4d st = wide_lds_read(myland + slot_of(s) * TS * TS, lane);
      if (s < RPW) {
        if (kin && K >= max(max(0, I - wb), Ks)) wide_mfma4(wide_lds_read(Arow + (size_t)(K % NEAR) * TS * TS, lane), Lr[s][NEAR - 1], pre0[s], pre1[s]);
        cur[s] = st - (pre0[s] + pre1[s]);
      } else {
        cur[s] = st - zero4;
      }
    };
    {
      // (the wave of the pivot chain takes its other rows behind the chain: what the column waits for is ITS first row)
#pragma unroll
      for (int s = 0; s <= RPW; s++)
        if (SFT_CHAIN_FIRST == 0 || d != 1 || s == 0) t3_row(s);
      if (d == NW - 1) {   // the border: all of its near products here (the role moves from wave to wave)
        const int Klo = max(max(0, J - wb), Ks);
        v4d s0 = zero4, s1 = zero4;
#pragma unroll
        for (int k = 0; k < NEAR; k++) {
          const int Kb = J - NEAR + k;
          if (Kb >= Klo && Kb < Kend)
            wide_mfma4(wide_lds_read(Arow + (size_t)(Kb % NEAR) * TS * TS, lane), wide_lds_read(Bring + (size_t)(Kb % NEAR) * TS * TS, lane), s0, s1);
        }
        curB = wide_lds_read(myland + SX * TS * TS, lane) - (s0 + s1);
      }
    }
    // the start tiles of the next column (the landing slots are free again); the wave with the pivot chain asks behind the chain
    if (d != 1 && J + 1 < nT) {
#pragma unroll
      for (int s = 0; s <= RPW; s++) request_tile(slot_of(s), nI[s], J + 1);
      if (d == 0) request_border(J + 1);
    }
    WT_SEG(1);
    // (4) TRSM with W_J (published a column ago), nearest row first; the wave that owns the next pivot row runs the pivot chain of
    //     column J+1 right behind its first tile
    double wv[4];
    if (elim) {
      const lds_double* Wj = Wbuf + (size_t)(J & 1) * TILE_LDS;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) wv[kk] = Wj[(4 * kk + crow) * TP + ccol];
    }
    v4d newC = zero4;
#pragma unroll
    for (int s = 0; s <= RPW; s++) {
      const int I = Irow[s];
      const bool have = !(s == 0 && d == 0) && I < nT && I - J <= wb;
      v4d xT = zero4, x = zero4;
      if (have && elim) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          xT = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[kk], cur[s][kk], xT, 0, 0, 0);
          x = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[s][kk], wv[kk], x, 0, 0, 0);
        }
        if (I - J <= NEAR) wide_lds_write(Aring + ((size_t)(I % (NEAR + 1)) * NEAR + (J % NEAR)) * TS * TS, lane, xT);
      }
      if (s == 0 && d == 1 && J + 1 < nT) {
        // the pivot chain of column J+1: start tile - the squares of the row's ring (block columns <= J-1: summed a column ago), then - the
        // square of the tile that has just left the TRSM
        const int Ip = J + 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0) -- as a builtin: the compiler's own wait counters take note */      // the diagonal tile has landed
        v4d dt = wide_lds_read(myland + SX * TS * TS, lane);
        dt = dt - preS;
        if (elim) {
          v4d t0 = zero4, t1 = zero4;
          wide_mfma4(xT, xT, t0, t1);
          dt -= t0 + t1;
        }
        // The Cholesky of that tile (1.3 us of dependent operations) is not this wave's: it owns two full rows and was the one every other wave
        // waited for at the barrier.  The wave of the CURRENT pivot row has the least to do in this column (its first row is eliminated): it
        // takes the tile through LDS (flag behind the data) and publishes W_{J+1} in front of the barrier.
        wide_lds_write(Dt, lane, dt);
        flag_set(dflag, Ip);
#ifdef SFT_CHAIN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (SFT_CHAIN_FIRST) {
#pragma unroll
          for (int t = 1; t <= RPW; t++) t3_row(t);
        }
        if (J + 1 < nT) {
#pragma unroll
          for (int t = 0; t <= RPW; t++) request_tile(slot_of(t), nI[t], J + 1);
        }
      }
      if (have) {
        if (elim) {
          if (helped) gb_st<16>(rLt, lane32, toff(J, I - J), xT);
          else gb_st<0>(rLt, lane32, toff(J, I - J), xT);
          gb_st<0>(rLb, lane32, toff(J, I - J), x);
        } else {
          schur_store(I, J, cur[s]);
        }
      }
      if (s < RPW) {
#pragma unroll
        for (int k = 0; k + 1 < NEAR; k++) Lr[s][k] = Lr[s][k + 1];
        Lr[s][NEAR - 1] = xT;
      } else {
        newC = xT;
      }
    }
    WT_SEG(2);
    // (5) the border's TRSM, the corner's update
    if (d == NW - 1) {
      if (elim) {
        v4d xbT = zero4;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) xbT = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[kk], curB[kk], xbT, 0, 0, 0);
        wide_lds_write(Bring + (size_t)(J % NEAR) * TS * TS, lane, xbT);
        if (helped) gb_st<16>(rLbT, lane32, (unsigned)J * (TS * TS * 8u), xbT);
        else gb_st<0>(rLbT, lane32, (unsigned)J * (TS * TS * 8u), xbT);
        if (ccol < SFT_BORDER) {
#pragma unroll
          for (int q = 0; q < 4; q++) Lbord[(size_t)ccol * Dnp + TS * J + crow + 4 * q] = xbT[q];
        }
        lds_double* cp = Cn + 64 * ((J + 7) & 7);
        if (NW == 16) {   // (the partial this column adds to: last touched eight columns ago, barriers in between)
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const int r = crow + 4 * q;
            cacc[q] = (r < SFT_BORDER && ccol < SFT_BORDER) ? cp[r * 7 + ccol] : 0.0;
          }
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(xbT[kk], -xbT[kk], cacc, 0, 0, 0);
        if (NW == 16) {
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const int r = crow + 4 * q;
            if (r < SFT_BORDER && ccol < SFT_BORDER) cp[r * 7 + ccol] = cacc[q];
          }
        }
      } else {
        wide_schur_border_of(P, which, J, curB);
      }
    }
    WT_SEG(3);
    // (6) the rows move up a slot behind the pivot row
    if (d == 0) {
#pragma unroll
      for (int s = 0; s + 1 < RPW; s++)
#pragma unroll
        for (int k = 0; k < NEAR; k++) Lr[s][k] = Lr[s + 1][k];
#pragma unroll
      for (int k = 0; k < NEAR; k++) Lr[RPW - 1][k] = k + 1 < NEAR ? zero4 : newC;
    }
    // (7) the pre-products of column J+1: everything but its last block column (ring entry k is block column J + 1 - NEAR + k now)
    {
      const int J1 = J + 1, Kend1 = min(J1, nS), Ks1 = ksplit(J1);
      const lds_double* Arow1 = Aring + (size_t)(J1 % (NEAR + 1)) * NEAR * TS * TS;
#pragma unroll
      for (int s = 0; s < RPW; s++) {
        pre0[s] = zero4; pre1[s] = zero4;
        const int I = nI[s];
        if (I >= nT || I - J1 > wb) continue;
        const int Klo = max(max(0, I - wb), Ks1);
#pragma unroll
        for (int k = 0; k + 1 < NEAR; k++) {
          const int K = J1 - NEAR + k;
          if (K >= Klo && K < Kend1) wide_mfma4(wide_lds_read(Arow1 + (size_t)(K % NEAR) * TS * TS, lane), Lr[s][k], pre0[s], pre1[s]);
        }
      }
      if (d == 2) {   // the pivot chain of the next column: the squares of row J+2's ring (block columns <= J)
        const int Ip = J + 2, Klo = max(max(0, Ip - wb), ksplit(Ip));
        v4d s0 = zero4, s1 = zero4;
#pragma unroll
        for (int k = 0; k < NEAR; k++) {
          const int K = J1 - NEAR + k;
          if (K >= Klo && K < Kend1) wide_mfma4(Lr[0][k], Lr[0][k], s0, s1);
        }
        preS = s0 + s1;
      }
    }
    if (d == 0 && J + 1 < nT) {   // the pivot chain's last link (see above)
      flag_wait(dflag, J + 1);
      pivot(J + 1, wide_lds_read(Dt, lane));
    }
#ifdef SFT_CHAIN_PRIO
    if (d == 0) __builtin_amdgcn_s_setprio(0);
#endif
    WT_SEG(4);
    look_for_helper(J);
    WT_SEG(5);
    __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0) -- as a builtin: the compiler's own wait counters take note */     // this column's stores have arrived (a helper may read them), the next column's tiles have landed
    WT_SEG(6);
    __syncthreads();
    WT_SEG(7);
    if (helped && tid == 0) __hip_atomic_store(V.sync, (epoch << 16) | (J + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef SFT_WIDE_TRACE
  __syncthreads();
  if (NW == 16) {   // part 0 only: 16 roles x 8 segments fill the debug words
    if (which == 0 && tid < 128) P.dbg[tid] = wtrace[tid];
    if (which == 0 && tid == 0) { P.dbg[126] = (double)(clock64() - wt_c0); P.dbg[127] = (double)(wall_clock64() - wt_w0); }
  } else {
    if (tid < 64) P.dbg[64 * which + tid] = wtrace[tid];
    if (tid == 0) { P.dbg[64 * which + 62] = (double)(clock64() - wt_c0); P.dbg[64 * which + 63] = (double)(wall_clock64() - wt_w0); }   // shader clocks, 100 MHz ticks
  }
#endif
#ifdef DSH_LAB
  if (helped && tid == 0) {
    atomicAdd((int*)V.sync + 1, st_got); atomicAdd((int*)V.sync + 2, st_miss); atomicAdd((int*)V.sync + 3, st_polls); atomicAdd((int*)V.sync + 4, (int)st_wait);
  }
#endif
  // ---- corner: partial sums of the eight waves in a fixed order; the part's corner contribution and whether its factorisation failed
  if (NW == 8) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER) Cn[64 * wave + r * 7 + ccol] = cacc[q];
    }
  }
  __syncthreads();
  if (tid < 49) {
    double sum = 0.0;
    for (int w = 0; w < 8; w++) sum += Cn[64 * w + tid];
    Cn[512 + tid] = sum;
  }
  __syncthreads();
  const auto xc = V.xchg + (size_t)V.xr_nT * V.xr_tpr * (TS * TS) + (size_t)8 * TS * V.xr_nT;
  if (tid < 49) xc[tid] = Cn[512 + tid];
  if (tid == 0) xc[56] = ctl->fact_ok ? 0.0 : 1.0;
  __syncthreads();
}

// The far sums of block column J of a part, by one workgroup (a helper's -- or the owner's own, for a column no helper delivered in time:
// one routine, one order of summation): the far tiles of tile row J staged in LDS (rowJ, slot J - K - 1), then far(I,J) for every row of the
// column, the diagonal tile and the border.  Items by falling number of products: 0 border, 1 diagonal tile, i >= 2 row J + i - 1; wave w
// takes item w (pass 1: up to MAXN products) and item 15 - w (pass 2: at most MAXN2).  What is stored (agent scope) is the tile the owner
// starts the column from: H(I,J)^T - far(I,J).  Every operand tile of an item is requested before its first product -- and the steps are
// separate so that a helper can request the tiles of its NEXT column while it multiplies this one's (FarColumn::*, factor_wide_helper).
// How a helper loads the finished L tiles.  They were written -- at agent scope, through to memory -- by the owner's CU BEFORE the helper learnt
// (from the owner's progress word) that they exist, and a tile is written once per launch: no cache on the helper's side can hold an older
// version of it from this launch, and what earlier launches left was invalidated when this kernel started (that is how two kernels on different
// XCDs see each other's results at all).  So a plain load is as correct as an agent-scope one here, and unlike it, it leaves the tile in the
// helper's L2 -- each tile is an operand of up to twelve block columns, and a CU keeps only about 64 KB of misses in flight (32 GB/s at 2 us:
// 7 us for the 240 KB of a column, measured, against 1.3 us of MFMA time).  -DSFT_FAR_LD_AGENT restores the agent-scope loads (A/B).
#ifdef SFT_FAR_LD_AGENT
#define FAR_LD_AUX 16
#else
#define FAR_LD_AUX 0
#endif
struct FarColumn {
  static constexpr int MAXN = WB - SFT_WIDE_NEAR;       // most far products of a tile
  static constexpr int MAXN2 = 5;                       // ... of an item of the second pass (items 8 .. 15: rows J + 7 and beyond)
  static_assert(WB - SFT_WIDE_NEAR - 7 <= MAXN2 && WB - SFT_WIDE_NEAR >= 1, "FarColumn: the second pass holds at most MAXN2 operand tiles");
  v4d bt[MAXN], bt2[MAXN2], h1, h2, areg[2];
  int n1, n2, K01, K02, na;                             // products and first block column of the two items; tiles of row J this wave stages

  // which tiles an item multiplies: I (row), K0, n; false: nothing to do
  __device__ __forceinline__ static bool item_of(const WideView& V, int J, int item, int& I, int& K0, int& n) {
    const int Ks = min(V.nS, J - SFT_WIDE_NEAR), K0d = max(0, J - V.wb);
    I = item <= 1 ? J : J + item - 1;
    K0 = item == 0 ? K0d : max(0, I - V.wb);
    n = Ks - K0;
    if (I >= V.nT || I - J > V.wb || n <= 0) { n = 0; return false; }
    return true;
  }
  template <int N>
  __device__ __forceinline__ static void request_item(const WideView& V, int J, int item, int lane, v4d (&b)[N], v4d& h, int& n, int& K0) {
    int I;
    if (!item_of(V, J, item, I, K0, n)) return;
    n = min(n, N);
    const __amdgpu_buffer_rsrc_t r = wide_rsrc(item == 0 ? V.LbT : V.Lt);
    const unsigned lane32 = 32u * lane;
    const unsigned b0 = item == 0 ? (unsigned)K0 * (TS * TS * 8u) : (unsigned)(wtile_off(V.tpr, K0, I - K0) * 8);
    const unsigned bs = item == 0 ? (TS * TS * 8u) : (unsigned)(V.tpr - 1) * (TS * TS * 8u);
#pragma unroll
    for (int i = 0; i < N; i++) b[i] = gb_ld<FAR_LD_AUX>(r, lane32, b0 + (unsigned)min(i, n - 1) * bs);
    if (item == 0) {
      const int crow = lane >> 4, ccol = lane & 15;
      h = (v4d){0.0, 0.0, 0.0, 0.0};
      if (ccol < SFT_BORDER) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int j = TS * J + crow + 4 * q;
          const bool in = j >= V.b_lo && j < V.b_hi;
          const double v = V.Hbord[(size_t)ccol * V.bstride + V.b_base + V.b_sign * (in ? j : V.b_lo)];
          h[q] = in ? v : 0.0;
        }
      }
    } else {
      h = *reinterpret_cast<const SFT_G v4d*>(V.Hb + wtile_off(V.tpr, I, I - J) + 4 * lane);
    }
  }
  template <int N>
  __device__ __forceinline__ static void finish_item(const WideView& V, int J, int item, int lane, const lds_double* rowJ, const v4d (&b)[N], const v4d& h, int n, int K0) {
    if (n <= 0) return;
    const bool sq = item == 1;
    v4d s0 = {0.0, 0.0, 0.0, 0.0}, s1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < N; i++)
      if (i < n) wide_mfma4(sq ? b[i] : wide_lds_read(rowJ + (size_t)(J - (K0 + i) - 1) * TS * TS, lane), b[i], s0, s1);
    const v4d t = h - (s0 + s1);
    if (item == 0) tile_st_agent(V.PfB + (size_t)J * TS * TS + 4 * lane, t);
    else tile_st_agent(V.Pf + wtile_off(V.tpr, J, item == 1 ? 0 : item - 1) + 4 * lane, t);
  }
  __device__ __forceinline__ void request1(const WideView& V, int J, int wave, int lane) { request_item<MAXN>(V, J, wave, lane, bt, h1, n1, K01); }
  __device__ __forceinline__ void request2(const WideView& V, int J, int wave, int lane) { request_item<MAXN2>(V, J, 15 - wave, lane, bt2, h2, n2, K02); }
  // the far tiles of tile row J this wave brings to LDS (every eighth): requested into registers ...
  __device__ __forceinline__ void requestA(const WideView& V, int J, int wave, int lane) {
    const int Ks = min(V.nS, J - SFT_WIDE_NEAR), K0d = max(0, J - V.wb);
    const __amdgpu_buffer_rsrc_t rLt = wide_rsrc(V.Lt);
    na = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int K = K0d + wave + 8 * i;
      if (K < Ks) { areg[i] = gb_ld<FAR_LD_AUX>(rLt, 32u * lane, (unsigned)(wtile_off(V.tpr, K, J - K) * 8)); na = i + 1; }
    }
  }
  // ... and written to their slots J - K - 1
  __device__ __forceinline__ void stageA(const WideView& V, int J, int wave, int lane, lds_double* rowJ) {
    const int K0d = max(0, J - V.wb);
#pragma unroll
    for (int i = 0; i < 2; i++)
      if (i < na) wide_lds_write(rowJ + (size_t)(J - (K0d + wave + 8 * i) - 1) * TS * TS, lane, areg[i]);
  }
  __device__ __forceinline__ void finish1(const WideView& V, int J, int wave, int lane, const lds_double* rowJ) { finish_item<MAXN>(V, J, wave, lane, rowJ, bt, h1, n1, K01); }
  __device__ __forceinline__ void finish2(const WideView& V, int J, int wave, int lane, const lds_double* rowJ) { finish_item<MAXN2>(V, J, 15 - wave, lane, rowJ, bt2, h2, n2, K02); }
};

// The same column by a workgroup of SIXTEEN wavefronts (128 registers per lane): one item per wave, its operand tiles in chunks of U -- two
// chunks requested ahead, the third into the first's registers behind its products.  Product for product the order of FarColumn::finish_item.
struct FarColumn16 {
  static constexpr int U = 4, MAXN = WB - SFT_WIDE_NEAR;
  static_assert(MAXN <= 3 * U, "FarColumn16: three chunks of operand tiles");
  v4d b0[U], b1[U], h, areg;
  int n, K0, na;
  __device__ __forceinline__ static void tiles_of(const WideView& V, int item, int I, int K0, __amdgpu_buffer_rsrc_t& r, unsigned& base, unsigned& step) {
    r = wide_rsrc(item == 0 ? V.LbT : V.Lt);
    base = item == 0 ? (unsigned)K0 * (TS * TS * 8u) : (unsigned)(wtile_off(V.tpr, K0, I - K0) * 8);
    step = item == 0 ? (TS * TS * 8u) : (unsigned)(V.tpr - 1) * (TS * TS * 8u);
  }
  __device__ __forceinline__ void request(const WideView& V, int J, int wave, int lane) {
    int I;
    if (!FarColumn::item_of(V, J, wave, I, K0, n)) return;
    n = min(n, MAXN);
    __amdgpu_buffer_rsrc_t r; unsigned base, step;
    tiles_of(V, wave, I, K0, r, base, step);
    const unsigned lane32 = 32u * lane;
#pragma unroll
    for (int i = 0; i < U; i++) b0[i] = gb_ld<FAR_LD_AUX>(r, lane32, base + (unsigned)min(i, n - 1) * step);
    if (n > U) {
#pragma unroll
      for (int i = 0; i < U; i++) b1[i] = gb_ld<FAR_LD_AUX>(r, lane32, base + (unsigned)min(U + i, n - 1) * step);
    }
    if (wave == 0) {
      const int crow = lane >> 4, ccol = lane & 15;
      h = (v4d){0.0, 0.0, 0.0, 0.0};
      if (ccol < SFT_BORDER) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int j = TS * J + crow + 4 * q;
          const bool in = j >= V.b_lo && j < V.b_hi;
          const double v = V.Hbord[(size_t)ccol * V.bstride + V.b_base + V.b_sign * (in ? j : V.b_lo)];
          h[q] = in ? v : 0.0;
        }
      }
    } else {
      h = *reinterpret_cast<const SFT_G v4d*>(V.Hb + wtile_off(V.tpr, I, I - J) + 4 * lane);
    }
  }
  __device__ __forceinline__ void requestA(const WideView& V, int J, int wave, int lane) {
    const int Ks = min(V.nS, J - SFT_WIDE_NEAR), K = max(0, J - V.wb) + wave;
    na = 0;
    if (K < Ks) { areg = gb_ld<FAR_LD_AUX>(wide_rsrc(V.Lt), 32u * lane, (unsigned)(wtile_off(V.tpr, K, J - K) * 8)); na = 1; }
  }
  __device__ __forceinline__ void stageA(const WideView& V, int J, int wave, int lane, lds_double* rowJ) {
    if (na) wide_lds_write(rowJ + (size_t)(J - (max(0, J - V.wb) + wave) - 1) * TS * TS, lane, areg);
  }
  __device__ __forceinline__ void finish(const WideView& V, int J, int wave, int lane, const lds_double* rowJ) {
    if (n <= 0) return;
    const bool sq = wave == 1;
    const int I = wave <= 1 ? J : J + wave - 1;
    v4d s0 = {0.0, 0.0, 0.0, 0.0}, s1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < U; i++)
      if (i < n) wide_mfma4(sq ? b0[i] : wide_lds_read(rowJ + (size_t)(J - (K0 + i) - 1) * TS * TS, lane), b0[i], s0, s1);
    if (n > 2 * U) {
      __amdgpu_buffer_rsrc_t r; unsigned base, step;
      tiles_of(V, wave, I, K0, r, base, step);
#pragma unroll
      for (int i = 0; i < U; i++) b0[i] = gb_ld<FAR_LD_AUX>(r, 32u * lane, base + (unsigned)min(2 * U + i, n - 1) * step);
    }
#pragma unroll
    for (int i = 0; i < U; i++)
      if (U + i < n) wide_mfma4(sq ? b1[i] : wide_lds_read(rowJ + (size_t)(J - (K0 + U + i) - 1) * TS * TS, lane), b1[i], s0, s1);
#pragma unroll
    for (int i = 0; i < U; i++)
      if (2 * U + i < n) wide_mfma4(sq ? b0[i] : wide_lds_read(rowJ + (size_t)(J - (K0 + 2 * U + i) - 1) * TS * TS, lane), b0[i], s0, s1);
    const v4d t = h - (s0 + s1);
    if (wave == 0) tile_st_agent(V.PfB + (size_t)J * TS * TS + 4 * lane, t);
    else tile_st_agent(V.Pf + wtile_off(V.tpr, J, wave == 1 ? 0 : wave - 1) + 4 * lane, t);
  }
};

template <int NW>
__device__ __forceinline__ void wide_far_column(const WideView& V, int J, lds_double* rowJ) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (min(V.nS, J - SFT_WIDE_NEAR) <= max(0, J - V.wb)) return;
  if constexpr (NW == 16) {
    FarColumn16 F;
    F.n = F.na = 0;
    F.request(V, J, wave, lane);
    F.requestA(V, J, wave, lane);
    F.stageA(V, J, wave, lane, rowJ);
    lds_barrier();
    F.finish(V, J, wave, lane, rowJ);
  } else {
    FarColumn F;
    F.n1 = F.n2 = 0;
    F.request1(V, J, wave, lane);
    F.request2(V, J, wave, lane);
    F.requestA(V, J, wave, lane);
    F.stageA(V, J, wave, lane, rowJ);
    lds_barrier();
    F.finish1(V, J, wave, lane, rowJ);
    F.finish2(V, J, wave, lane, rowJ);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();                                        // every wave's tiles have arrived
}

template <int NW>
__device__ __noinline__ void wide_far_column_of(const SftDev& P, int which, int J_, lds_double* rowJ) {
  const WideView V = wide_view(P, which);
  const int J = __builtin_amdgcn_readfirstlane(J_);
  wide_far_column<NW>(V, J, rowJ);
}

// A helper workgroup of part `which` (see the comment in front of factor_wide): hidx of nh, block columns near + 1 + hidx, + nh, ...
// For its column J it waits until the owner has finished the block columns below Ks = min(nS, J - near), forms the column (FarColumn) and
// raises its flag.  While it multiplies, the operand tiles of its next column are already on their way when the owner's progress allows
// (a helper that cannot keep up with its owner is bound by what it multiplies then, not by one memory round trip per step).  It never makes
// the owner wait: a column the owner has already decided about is skipped, and a helper whose owner shows no progress (or is not there) leaves.
//
// What the owner <-> helper hand-over relies on (gfx950; NOT what the HIP memory model promises for relaxed atomics, so it is pinned to this
// target by the static_assert below instead of being paid for with a release / acquire pair per block column -- an agent-scope acquire is
// `buffer_inv sc1`, which would also throw the finished L tiles out of the helper's L2, each of them an operand of up to twelve columns):
//   1. data that crosses CUs (Pf / PfB tiles, L tiles the helper reads) is written with sc1 stores = written through to memory past the
//      writer's XCD L2, and the writer's `s_waitcnt vmcnt(0)` + workgroup barrier sit between those stores and the flag / progress store;
//      on gfx9 a store has left the L2 write path when vmcnt counts it done, and one wave's stores to one address space are not reordered
//      past a completed s_waitcnt;
//   2. flags and progress words are agent-scope atomics (performed at memory, never cached in a non-coherent L2 line);
//   3. a reader's XCD L2 holds no stale copy of a tile: every tile is written exactly once per launch, before any reader can learn (through
//      2.) that it exists, a kernel starts with its L2 invalidated for non-coherent lines, and nothing prefetches neighbouring lines
//      (tiles are 2 KB aligned and whole);
//   4. the compiler does not move the plain loads of a tile above the atomic load that allowed them: the poll loop ends in a branch on the
//      loaded value and the loads sit behind `__builtin_amdgcn_s_waitcnt` / barrier intrinsics, which hipcc treats as memory barriers for
//      scheduling.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "sft_wide.h: the owner / helper hand-over (relaxed agent-scope flags + sc1 write-through) is validated for gfx950 only -- see the comment above factor_wide_helper"
#endif
template <int NW = 8>
__device__ __noinline__ void factor_wide_helper(const SftDev& P, int which_, int hidx_, int nh_, int epoch_, Ctl* ctl, double* ws) {
  const int which = __builtin_amdgcn_readfirstlane(which_), hidx = __builtin_amdgcn_readfirstlane(hidx_), nh = __builtin_amdgcn_readfirstlane(nh_), epoch = __builtin_amdgcn_readfirstlane(epoch_);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WideView V = wide_view(P, which);
  if (V.sync == nullptr || nh <= 0) return;
  const int nT = V.nT, nS = V.nS, wb = V.wb;
  const int near = wide_near(wb);
  lds_double* rowJ = to_lds(ws);                        // far tiles of tile row J, slot J - K - 1 like the owner's
  lds_int* cmd = (lds_int*)(rowJ + 2 * WB * TS * TS);   // thread 0's verdict: [0] this column: 1 go, 0 skip, -1 leave; [1] where to skip to; [2] the next column may be requested
  int polls_left = WIDE_HELPER_POLLS, cols_seen = -1;
#ifdef DSH_LAB
  int st_done = 0, st_skip = 0;
  long long st_wait = 0, st_work = 0, st_seg[4] = {0, 0, 0, 0};
#define HT_SEG(i) do { const long long t__ = wall_clock64(); st_seg[i] += t__ - ht0; ht0 = t__; } while (0)
#else
#define HT_SEG(i) do {} while (0)
#endif
  using Far = typename std::conditional<NW == 16, FarColumn16, FarColumn>::type;
  Far F;
  if constexpr (NW == 16) { F.n = F.na = 0; } else { F.n1 = F.n2 = F.na = 0; }
  bool have = false;                                    // the tiles of column J have been requested (in the iteration before)
  int J = near + 1 + hidx;
#pragma unroll 1
  while (J < nT) {
    const int Ks = min(nS, J - near), K0d = max(0, J - wb);
    const int Jn = J + nh, Ksn = min(nS, Jn - near);
#ifdef DSH_LAB
    const long long t0 = wall_clock64();
#endif
    if (tid == 0) {
      int c = 0, nx = 0;
      while (true) {
        const int w = __hip_atomic_load(V.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int cols = (w >> 16) == epoch ? (w & 0xffff) : 0;
        if (cols >= nT) { c = -1; break; }                // the owner is through
        if (cols >= J - 1) {                              // the owner has decided about this column (or is past it): on to the first column it has not
          cmd[1] = J + ((cols + 2 - J + nh - 1) / nh) * nh;
          c = 0;
          break;
        }
        if (cols >= Ks) { c = 1; nx = Jn < nT && cols >= Ksn && cols < Jn - 1; break; }
        if (cols != cols_seen) { cols_seen = cols; polls_left = WIDE_HELPER_POLLS; }   // the owner moves: the patience starts over
        if (--polls_left <= 0) { c = -1; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      cmd[0] = c;
      cmd[2] = nx;
    }
    __syncthreads();
    const int go = uni(cmd[0]), Jskip = uni(cmd[1]), next_ok = uni(cmd[2]);
    __syncthreads();                                      // (cmd is rewritten for the next column)
#ifdef DSH_LAB
    const long long t1 = wall_clock64();
    st_wait += t1 - t0;
    if (go <= 0 && tid == 0) {
      if (go == 0) st_skip++;
      if (go < 0) { atomicAdd((int*)V.sync + 5, st_done); atomicAdd((int*)V.sync + 6, st_skip); atomicAdd((int*)V.sync + 7, (int)st_wait); atomicAdd((int*)V.sync + 8, (int)st_work); }
    }
#endif
    if (go < 0) return;
    if (go == 0) { J = Jskip; have = false; continue; }
    if (Ks <= K0d) { J += nh; have = false; continue; }
    if (!have) {
      if constexpr (NW == 16) {
        F.request(V, J, wave, lane);
      } else {
        F.request1(V, J, wave, lane);
        F.request2(V, J, wave, lane);
      }
      F.requestA(V, J, wave, lane);
    }
#ifdef DSH_LAB
    long long ht0 = t1;
#endif
    F.stageA(V, J, wave, lane, rowJ);
    lds_barrier();
    HT_SEG(0);
    if constexpr (NW == 16) {
      F.finish(V, J, wave, lane, rowJ);
      HT_SEG(1);
      if (next_ok) { F.request(V, Jn, wave, lane); F.requestA(V, Jn, wave, lane); }
    } else {
      F.finish1(V, J, wave, lane, rowJ);                  // (its result is stored at once: done long before the column's flag)
      if (next_ok) { F.request1(V, Jn, wave, lane); F.requestA(V, Jn, wave, lane); }
      HT_SEG(1);
      F.finish2(V, J, wave, lane, rowJ);
      if (next_ok) F.request2(V, Jn, wave, lane);
    }
    HT_SEG(2);
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // this column's tiles have arrived (and the next one's operands)
    __syncthreads();
    HT_SEG(3);
    if (tid == 0) __hip_atomic_store(V.sync + WIDE_SYNC_READY + J, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef DSH_LAB
    st_work += wall_clock64() - t1;
    st_done++;
#endif
    have = next_ok != 0;
    J = Jn;
  }
#ifdef DSH_LAB
  if (tid == 0) {
    atomicAdd((int*)V.sync + 5, st_done); atomicAdd((int*)V.sync + 6, st_skip); atomicAdd((int*)V.sync + 7, (int)st_wait); atomicAdd((int*)V.sync + 8, (int)st_work);
    for (int i = 0; i < 4; i++) atomicAdd((int*)V.sync + 9 + i, (int)st_seg[i]);
  }
#endif
}

// Back substitution for the wide band: x_J = W_J^T (y_J - sum_{I > J} X(I,J)^T x_I - L_cJ^T x_cam), block columns from the last to the first.
// The products of a block column with everything but x_{J+1} do not depend on the column before it: waves 1..7 form them ONE COLUMN AHEAD
// (tiles (J-1+d, J-1), d = 2..16, and the camera rows, while wave 0 finishes column J), wave 0 adds the one product that needs x_{J+1}
// (tile (J+1, J)), sums in the fixed order camera, d = 1, 2, ..., multiplies by W_J^T and publishes x_J: one barrier per column, and what a
// column waits for is one tile product, seventeen subtractions and the product with W -- not all of its products.
// which: see WideView.  A part (0 / 1) starts behind its eliminated columns: the separator rows of its band matrix take the solution of
// the reduced problem (part 1 in reversed order), the camera update comes from there as well.
// red: which copy of the reduced problem holds the separator / camera solution a part starts from (2 or 3)
__device__ __noinline__ void backsub_wide(const SftDev& P, int which, Ctl* ctl, double* ws, int red = 2) {
  constexpr int NW = 8, RING = 32;
  static_assert(WB == 16, "backsub_wide: the tiles d = 2..16 of a column are dealt to seven waves");
  if (!ctl->fact_ok) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef SFT_BS_UNIFORM
#define SFT_BS_UNIFORM true
#endif
  const WideView V = wide_view<SFT_BS_UNIFORM>(P, which);
  const int nT = V.nT, nS = V.nS;
  const int Dnp = TS * nT;
  const int tpr = V.tpr, wb = V.wb;
  lds_double* xw = to_lds(ws);                 // ring of RING x-tiles
  lds_double* part = xw + TS * RING;           // [column & 1][slot]: slot 0 camera rows, slots 2..WB sub-diagonal tiles (slot 1: wave 0's own, in registers)
  constexpr int PSZ = TS * (WB + 1);
  const int crow = lane >> 4, ccol = lane & 15;
  const bool is_part = which == 0 || which == 1;
  const auto xred = is_part ? uni(P.part[red].x) : V.x;        // where the camera update (and, for a part, the separator solution) is
  const int sred = is_part ? TS * V.xr_nT : Dnp;
  const double xc = (lane < 6) ? xred[sred + lane] : 0.0;
  double xcr[6];
#pragma unroll
  for (int r = 0; r < 6; r++) xcr[r] = bcast_lane(xc, r);
  const auto Lg = V.Lb;
  const auto Lbord = V.Lbord;
  const auto Linv_g = V.Linv;
  const auto xg = V.x;
  for (int i = tid; i < 2 * PSZ; i += 64 * NW) part[i] = 0.0;   // slots beyond wb stay zero
  if (is_part)
    for (int i = tid; i < TS * (nT - nS); i += 64 * NW) {             // separator rows nS .. nT-1 of the part
      const int I = nS + (i >> 4), l = i & 15;
      xw[(I & (RING - 1)) * TS + l] = xred[V.reversed ? sred - 1 - i : i];
    }
  __syncthreads();
  // Prefetch: every load of the loops below is UNCONDITIONAL (tiles that do not exist are read from the column's diagonal slot and never used,
  // columns below zero from column zero) and each role has its own loop: the compiler can then count the loads in flight and wait for the
  // oldest only -- a conditional load anywhere in the loop makes it wait for ALL of them (vmcnt(0)) once per column, which is the latency
  // of a trip to memory per column and was what the back substitution took (1 us per column; C5: 196 columns).
  auto tile_at = [&](int J, int d) -> v4d {
    const int Jc = uni(max(J, 0)), dd = uni((Jc + d < nT && d <= wb) ? d : 0);   // (scalar address arithmetic)
    return *reinterpret_cast<const SFT_G v4d*>(uni(Lg + wtile_off(tpr, Jc, dd)) + 4 * lane);
  };
  // sum over the rows of tile (J + d, J) times x_{J+d}: the partial product of one tile, in every lane of its column
  auto tile_product = [&](const v4d& t, int I) -> double {
    const lds_double* xi = xw + (I & (RING - 1)) * TS + crow;
    double p = 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++) p = fma(t[q], xi[4 * q], p);
    return sum_rows(p);
  };
#ifndef SFT_BS_PF
#define SFT_BS_PF 2   // columns of look-ahead of the loads (A/B: 2 < 3 < 4 -- the tiles are in L2, the registers are what is short)
#endif
  constexpr int PF = SFT_BS_PF;
  if (wave == 0) {
    // ---- wave 0 finishes column J: the product with x_{J+1}, the sum, W_J^T
    struct Pre0 { v4d t, li; double y; };
    auto fetch0 = [&](int J) -> Pre0 {
      Pre0 p;
      const int Jc = uni(max(J, 0));
      p.t = tile_at(J, 1);
      p.li = *reinterpret_cast<const SFT_G v4d*>(Linv_g + (size_t)Jc * TS * TS + 4 * lane);
      p.y = Lbord[(size_t)6 * Dnp + TS * Jc + ccol];
      return p;
    };
    Pre0 ring[PF];
#pragma unroll
    for (int j = 0; j < PF; j++) ring[j] = fetch0(nS - 1 - j);
    lds_barrier();
#pragma unroll 1
    for (int base = nS - 1; base >= 0; base -= PF) {
#pragma unroll
      for (int j = 0; j < PF; j++) {
        const int J = base - j;
        if (J < 0) break;
        const Pre0 cur = ring[j];
        ring[j] = fetch0(J - PF);
        const lds_double* pj = part + (J & 1) * PSZ;
        const double p1 = (J + 1 < nT && 1 <= wb) ? tile_product(cur.t, J + 1) : 0.0;
        double v = cur.y;
        v -= pj[ccol];
        v -= p1;
#pragma unroll
        for (int i = 2; i <= WB; i++) v -= pj[i * TS + ccol];   // (the order of the sum is the order it always had: subtracting the late product last gained nothing, A/B)
        double p = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) p = fma(cur.li[q], __shfl(v, crow + 4 * q, 64), p);
        p = sum_rows(p);
        if (lane < TS) { xw[(J & (RING - 1)) * TS + lane] = p; xg[TS * J + lane] = p; }
        lds_barrier();
      }
    }
  } else {
    // ---- waves 1..7, one column ahead of wave 0: the products of column J - 1 that do not need x_J -- tiles d = wave + 1, wave + 8 and (wave 1) 16;
    // wave 7 the camera rows as well.  How many of a wave's tiles lie inside the band does not change from column to column: one loop per
    // count (and with / without the camera rows), each with a fixed number of loads per column.
    const int d0 = wave + 1, d1 = wave + 8, d2 = wave == 1 ? WB : WB + 1;
    auto run = [&](auto ntile_c, auto cam_c) {
      constexpr int NTILE = decltype(ntile_c)::value;
      constexpr bool CAM = decltype(cam_c)::value;
      struct Pre1 { v4d t[NTILE > 0 ? NTILE : 1]; double aux[CAM ? 6 : 1]; };
      auto dof = [&](int t) -> int { return t == 0 ? d0 : (t == 1 ? d1 : d2); };
      auto fetch1 = [&](int J) -> Pre1 {
        Pre1 p;
#pragma unroll
        for (int t = 0; t < NTILE; t++) p.t[t] = tile_at(J, dof(t));
        if (CAM) {
          const int Jc = uni(max(J, 0));
#pragma unroll
          for (int r = 0; r < 6; r++) p.aux[r] = Lbord[(size_t)r * Dnp + TS * Jc + ccol];
        }
        return p;
      };
      auto ahead = [&](int J, const Pre1& cur) {
        lds_double* pj = part + (J & 1) * PSZ;
#pragma unroll
        for (int t = 0; t < NTILE; t++) {
          const int d = dof(t), I = J + d;
          const double p = (I < nT) ? tile_product(cur.t[t], I) : 0.0;
          if (lane < TS) pj[d * TS + lane] = p;
        }
        if (CAM && lane < TS) {
          double p = 0.0;
#pragma unroll
          for (int r = 0; r < 6; r++) p = fma(cur.aux[r], xcr[r], p);
          pj[lane] = p;
        }
      };
      Pre1 ring[PF];
#pragma unroll
      for (int j = 0; j < PF; j++) ring[j] = fetch1(nS - 1 - j);
      if (nS > 0) {                                // column nS-1 ahead of the loop
        const Pre1 cur = ring[0];
#pragma unroll
        for (int j = 0; j + 1 < PF; j++) ring[j] = ring[j + 1];
        ring[PF - 1] = fetch1(nS - 1 - PF);
        ahead(nS - 1, cur);
      }
      lds_barrier();
#pragma unroll 1
      for (int base = nS - 1; base >= 0; base -= PF) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
          const int J = base - j;                   // the column wave 0 finishes
          if (J < 0) break;
          const Pre1 cur = ring[j];
          ring[j] = fetch1(J - 1 - PF);
          if (J - 1 >= 0) ahead(J - 1, cur);
          lds_barrier();
        }
      }
    };
    using std::integral_constant;
    const int ntile = (d0 <= wb ? 1 : 0) + (d1 <= wb ? 1 : 0) + (d2 <= wb ? 1 : 0);   // (d0 < d1 < d2: the first ntile lie inside the band)
    if (wave == 7) {
      if (ntile >= 2) run(integral_constant<int, 2>{}, integral_constant<bool, true>{});
      else if (ntile == 1) run(integral_constant<int, 1>{}, integral_constant<bool, true>{});
      else run(integral_constant<int, 0>{}, integral_constant<bool, true>{});
    } else {
      if (ntile >= 3) run(integral_constant<int, 3>{}, integral_constant<bool, false>{});
      else if (ntile == 2) run(integral_constant<int, 2>{}, integral_constant<bool, false>{});
      else if (ntile == 1) run(integral_constant<int, 1>{}, integral_constant<bool, false>{});
      else run(integral_constant<int, 0>{}, integral_constant<bool, false>{});
    }
  }
  __syncthreads();
}
