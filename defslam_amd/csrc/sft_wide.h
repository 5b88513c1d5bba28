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
};
__device__ __forceinline__ WideView wide_view(const SftDev& P, int which) {
  WideView v;
  const int Dn = uni(P.Dn), Dnp = ((Dn + NB - 1) / NB) * NB;
  if (which < 0) {
    v.nT = Dnp / TS; v.nS = v.nT; v.tpr = uni(P.tpr); v.wb = uni(P.wbt);
    v.lam_lo = 0; v.lam_hi = Dn;
    v.bstride = Dnp; v.b_base = 0; v.b_sign = 1; v.b_lo = 0; v.b_hi = Dnp;
    v.xr_tpr = 0; v.xr_nT = 0; v.reversed = false; v.corner_from_H = true; v.finish = true;
    v.Hb = uni(P.Hb); v.Hbord = uni(P.Hbord); v.Hcorner = uni(P.Hcorner);
    v.Lb = uni(P.Lb); v.Lt = uni(P.Lt); v.LbT = uni(P.LbT); v.Lbord = uni(P.Lbord); v.Linv = uni(P.Linv); v.x = uni(P.x); v.xchg = nullptr;
    return v;
  }
  const auto& q = P.part[which];
  v.nT = uni(q.nT); v.nS = uni(q.nS); v.tpr = uni(q.tpr); v.wb = uni(q.wbt);
  v.Hb = uni(q.Hb); v.Lb = uni(q.Lb); v.Lt = uni(q.Lt); v.LbT = uni(q.LbT); v.Lbord = uni(q.Lbord); v.Linv = uni(q.Linv); v.x = uni(q.x); v.xchg = uni(q.xchg);
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

// lam_corner: damping of the 6x6 camera block (a part that does not start from H_cc adds none; the reduced problem's corner carries it already)
__device__ __noinline__ void factor_wide(const SftDev& P, int which, Ctl* ctl, double* ws) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WideView V = wide_view(P, which);
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
  __syncthreads();

  // sum_K L(J,K) L(I,K)^T for one row: A operand = row-J tile (LDS), B operand = the stored tile of row I (memory).
  // The FP64 MFMA pipe bounds the products (two waves per SIMD: 573 cycles per product), so the tile loads must not add
  // their latency to it: NCH chunks of U products, fully unrolled (straight-line code keeps exact wait counts), the loads of
  // chunk c+1 issued before the MFMAs of chunk c.  Every chunk issues U loads (the last tile again beyond the row's end).
  constexpr int U = 4;
  const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
  // Staged tiles live in LDS as two 1 KB planes (registers 0,1 of every lane, then registers 2,3): every ds_read_b128 /
  // ds_write_b128 touches 64 consecutive 16-byte words.
  typedef double v2d_ __attribute__((ext_vector_type(2)));
  using lds_v2d = __attribute__((address_space(3))) v2d_;
  auto lds_tile_read = [&](const lds_double* tile) -> v4d {
    const v2d_ lo = *reinterpret_cast<const lds_v2d*>(tile + 2 * lane);
    const v2d_ hi = *reinterpret_cast<const lds_v2d*>(tile + 128 + 2 * lane);
    return (v4d){lo.x, lo.y, hi.x, hi.y};
  };
  auto lds_tile_write = [&](lds_double* tile, const v4d& v) {
    *reinterpret_cast<lds_v2d*>(tile + 2 * lane) = (v2d_){v[0], v[1]};
    *reinterpret_cast<lds_v2d*>(tile + 128 + 2 * lane) = (v2d_){v[2], v[3]};
  };
  auto mfma4 = [&](const v4d& a, const v4d& b, v4d& s0, v4d& s1) {
    s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], s1, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], s1, 0, 0, 0);
  };
  // (Kend: the products run over block columns K0 .. Kend-1 -- Kend = J for an eliminated column, the part's nS for a separator column)
  auto products_n = [&](auto nch_c, int J, int Kend, const lds_double* rowJ, const SFT_G double* brow, long bstride, int K0) -> v4d {
    // brow: tile of block column K0 for this row; the tile of block column K0 + i sits i * bstride doubles further
    constexpr int NCH = decltype(nch_c)::value;
    v4d s0 = zero4, s1 = zero4;
    v4d buf[2][U];
    const int last = max(Kend - 1 - K0, 0);
#pragma unroll
    for (int u = 0; u < U; u++) buf[0][u] = wide_ltile_load(brow + (long)min(u, last) * bstride + 4 * lane);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      if (c + 1 < NCH) {
#pragma unroll
        for (int u = 0; u < U; u++)
          buf[(c + 1) & 1][u] = wide_ltile_load(brow + (long)min(U * (c + 1) + u, last) * bstride + 4 * lane);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int K = K0 + U * c + u;
        if (K < Kend) mfma4(lds_tile_read(rowJ + (size_t)(J - K - 1) * TS * TS), buf[c & 1][u], s0, s1);
      }
    }
    return s0 + s1;
  };
  // sum_{K = K0 .. Kend-1} T_K T_K^T with T_K = the stored tile (I, K) from memory as both operands: the diagonal tile of the
  // NEXT column, formed a column ahead (look-ahead) from the tiles that already exist.
  auto squares = [&](const SFT_G double* brow, long bstride, int n) -> v4d {
    v4d s0 = zero4, s1 = zero4;
    v4d buf[2][U];
    const int last = max(n - 1, 0);
#pragma unroll
    for (int u = 0; u < U; u++) buf[0][u] = wide_ltile_load(brow + (long)min(u, last) * bstride + 4 * lane);
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c + 1 < 4) {
#pragma unroll
        for (int u = 0; u < U; u++)
          buf[(c + 1) & 1][u] = wide_ltile_load(brow + (long)min(U * (c + 1) + u, last) * bstride + 4 * lane);
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (U * c + u < n) mfma4(buf[c & 1][u], buf[c & 1][u], s0, s1);
    }
    return s0 + s1;
  };
  auto products = [&](int J, int I, const lds_double* rowJ, const SFT_G double* brow, long bstride, int K0) -> v4d {
    const int Kend = min(J, nS);
    const int n = Kend - K0;
    if (n <= 0) return zero4;
    if (n <= 2 * U) return products_n(std::integral_constant<int, 2>{}, J, Kend, rowJ, brow, bstride, K0);
    return products_n(std::integral_constant<int, 4>{}, J, Kend, rowJ, brow, bstride, K0);
  };
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
  const long kstride = (long)(tpr - 1) * TS * TS;        // tile (I, K+1) sits (tpr - 1) tiles after tile (I, K)
  // Look-ahead: the diagonal tile of column J+1 minus its products with block columns <= J-1 is formed during column J by the
  // wave that owns column J+1 next (roles rotate by one wave per column) and waits in its registers.
  v4d dlook = *reinterpret_cast<const SFT_G v4d*>(Hg + 4 * lane);     // column 0: H(0,0), no products

#pragma unroll 1
  for (int J = 0; J < nT; J++) {
    lds_double* rowJ = Lrow + (size_t)(J & 1) * WB * TS * TS;
    lds_double* rowN = Lrow + (size_t)((J + 1) & 1) * WB * TS * TS;
    // Roles rotate with the column.  The FP64 MFMA pipe is what the products are bound by (73 cycles per MFMA, two waves per
    // SIMD: tools/probes/lds_mfma_probe.hip), so the rows are dealt by their number of products, and the Cholesky is taken
    // off the path every wave waits for: role 0 (owner of column J) starts from the look-ahead tile, adds the one product
    // with block column J-1, factors, and then does two short rows; role 1 (owner of column J+1) forms the look-ahead tile of
    // the next column (15 products, tiles from memory) and two short rows; the others take about 20 products each:
    //   0: diag, J+10, J+12 (+ product-free J+16) | 1: look-ahead, J+13, J+14 | 2: J+1, J+11 | 3: J+2, J+9 | 4: J+3, J+8
    //   5: J+4, J+7 | 6: J+5, J+6 | 7: border, J+15
    const int d = (wave - J) & 7;
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
        const v4d a = lds_tile_read(rowJ);
        v4d s0 = zero4, s1 = zero4;
        mfma4(a, a, s0, s1);
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
    v4d accT[3];
    bool have[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int I = Irow[t];
      have[t] = I < nT && I - J <= wb;
      if (have[t]) {
        const v4d h = *reinterpret_cast<const SFT_G v4d*>(Hg + wtile_off(tpr, I, I - J) + 4 * lane);
        const int K0 = max(0, I - wb);
        accT[t] = h - products(J, I, rowJ, Ltg + wtile_off(tpr, K0, I - K0), kstride, K0);
      }
    }
    if (d == 1 && J + 1 < nT) {
      // look-ahead for column J+1: H(J+1,J+1) - sum_{K <= J-1} L(J+1,K) L(J+1,K)^T  (the product with block column J follows
      // in the next column, when tile (J+1, J) exists)
      const int I = J + 1, K0 = max(0, I - wb);
      const v4d h = *reinterpret_cast<const SFT_G v4d*>(Hg + wtile_off(tpr, I, 0) + 4 * lane);
      dlook = h - squares(Ltg + wtile_off(tpr, K0, I - K0), kstride, min(J, nS) - K0);
    }
    // border: accTb[j][i] = Hbord[i][16 J + j] - sum_K (L(J,K) Lb(K)^T)[j][i], i < 7
    v4d accTb = zero4;
    const bool bwave = d == 7;
    if (bwave) {
      v4d h = zero4;
      if (ccol < SFT_BORDER) {
#pragma unroll
        for (int q = 0; q < 4; q++) h[q] = bord_h(ccol, TS * J + crow + 4 * q);
      }
      const int K0 = max(0, J - wb);
      accTb = h - products(J, nT, rowJ, LbTg + (size_t)K0 * TS * TS, (long)TS * TS, K0);
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (staged[h]) lds_tile_write(rowN + (size_t)(1 + wave + 8 * h) * TS * TS, stage[h]);
    lds_barrier();                                       // W_J is published
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
      if (I == J + 1) lds_tile_write(rowN, xT);
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
    __syncthreads();                                     // the column's tiles are in memory (and in rowN) for the next one
  }
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

// Back substitution for the wide band: x_J = W_J^T (y_J - sum_{I > J} X(I,J)^T x_I - L_cJ^T x_cam), block columns from the
// last to the first; wave w forms the partial products of tiles (J + d, J), d = w + 1 and w + 9; wave 0 finishes the block.
// which: see WideView.  A part (0 / 1) starts behind its eliminated columns: the separator rows of its band matrix take the solution of
// the reduced problem (part 1 in reversed order), the camera update comes from there as well.
// red: which copy of the reduced problem holds the separator / camera solution a part starts from (2 or 3)
__device__ __noinline__ void backsub_wide(const SftDev& P, int which, Ctl* ctl, double* ws, int red = 2) {
  constexpr int NW = 8, RPW = WB / NW, RING = 32;
  if (!ctl->fact_ok) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WideView V = wide_view(P, which);
  const int nT = V.nT, nS = V.nS;
  const int Dnp = TS * nT;
  const int tpr = V.tpr, wb = V.wb;
  lds_double* xw = to_lds(ws);                 // ring of RING x-tiles
  lds_double* part = xw + TS * RING;           // slot 0: camera rows, slots 1..WB: sub-diagonal tiles
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
  for (int i = tid; i < TS * (WB + 1); i += 64 * NW) part[i] = 0.0;   // slots beyond wb stay zero
  if (is_part)
    for (int i = tid; i < TS * (nT - nS); i += 64 * NW) {             // separator rows nS .. nT-1 of the part
      const int I = nS + (i >> 4), l = i & 15;
      xw[(I & (RING - 1)) * TS + l] = xred[V.reversed ? sred - 1 - i : i];
    }
  __syncthreads();
  struct Pre { v4d t[RPW]; double aux[6]; };
  auto fetch = [&](int J) -> Pre {
    Pre p;
#pragma unroll
    for (int t = 0; t < RPW; t++) p.t[t] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 6; r++) p.aux[r] = 0.0;
    if (J < 0) return p;
#pragma unroll
    for (int t = 0; t < RPW; t++) {
      const int d = wave + 1 + NW * t;
      if (J + d < nT && d <= wb) p.t[t] = *reinterpret_cast<const SFT_G v4d*>(Lg + wtile_off(tpr, J, d) + 4 * lane);
    }
    if (wave == 0) {
      const v4d li = *reinterpret_cast<const SFT_G v4d*>(Linv_g + (size_t)J * TS * TS + 4 * lane);
#pragma unroll
      for (int q = 0; q < 4; q++) p.aux[q] = li[q];
      p.aux[4] = Lbord[(size_t)6 * Dnp + TS * J + ccol];
    } else if (wave == 1) {
#pragma unroll
      for (int r = 0; r < 6; r++) p.aux[r] = Lbord[(size_t)r * Dnp + TS * J + ccol];
    }
    return p;
  };
  constexpr int PF = 4;
  Pre ring[PF];
#pragma unroll
  for (int j = 0; j < PF; j++) ring[j] = fetch(nS - 1 - j);
#pragma unroll 1
  for (int base = nS - 1; base >= 0; base -= PF) {
#pragma unroll
    for (int j = 0; j < PF; j++) {
      const int J = base - j;
      if (J < 0) break;
      const Pre cur = ring[j];
      ring[j] = fetch(J - PF);
#pragma unroll
      for (int t = 0; t < RPW; t++) {
        const int d = wave + 1 + NW * t;
        const int I = J + d;
        double p = 0.0;
        if (I < nT && d <= wb) {
          const lds_double* xi = xw + (I & (RING - 1)) * TS + crow;
#pragma unroll
          for (int q = 0; q < 4; q++) p = fma(cur.t[t][q], xi[4 * q], p);
          p = sum_rows(p);
        }
        if (lane < TS && d <= wb) part[d * TS + lane] = p;
      }
      if (wave == 1 && lane < TS) {
        double p = 0.0;
#pragma unroll
        for (int r = 0; r < 6; r++) p = fma(cur.aux[r], xcr[r], p);
        part[lane] = p;
      }
      lds_barrier();
      if (wave == 0) {
        double v = cur.aux[4];
#pragma unroll
        for (int i = 0; i <= WB; i++) v -= part[i * TS + ccol];
        double p = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) p = fma(cur.aux[q], __shfl(v, crow + 4 * q, 64), p);
        p = sum_rows(p);
        if (lane < TS) { xw[(J & (RING - 1)) * TS + lane] = p; xg[TS * J + lane] = p; }
      }
      lds_barrier();
    }
  }
  __syncthreads();
}
