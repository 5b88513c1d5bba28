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

__device__ __noinline__ void factor_wide(const SftDev& P, Ctl* ctl, double* ws) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Dn = uni(P.Dn);
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  const int nT = Dnp / TS;
  const int tpr = uni(P.tpr), wb = uni(P.wbt);
  lds_double* Lrow = to_lds(ws);                       // 2 x WB tiles, native layout (lane, 4 doubles): row-J tiles, dist 1..wb
  lds_double* LinvK = Lrow + 2 * WB * TS * TS;         // W_J, k-major padded: LinvK[k*TP + j] = W[j][k]
  lds_double* Cn = LinvK + TILE_LDS;                   // 8 partial 7x7 corners, then the corner itself
  const double lambda = ctl->lambda;
  const int crow = lane >> 4, ccol = lane & 15;
  const auto Hg = uni(P.Hb);
  const auto Hbord = uni(P.Hbord);
  const auto Lg = uni(P.Lb);
  const auto Ltg = uni(P.Lt);
  const auto LbTg = uni(P.LbT);
  const auto Lbord = uni(P.Lbord);
  const auto Linv_g = uni(P.Linv);
  v4d cacc = {0.0, 0.0, 0.0, 0.0};     // this wave's share of the corner updates; wave 0 starts from H_cc + lambda I
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER && ccol <= r) cacc[q] = P.Hcorner[r * 7 + ccol] + ((r == ccol && r < 6) ? lambda : 0.0);
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
  auto products_n = [&](auto nch_c, int J, const lds_double* rowJ, const SFT_G double* brow, long bstride, int K0) -> v4d {
    // brow: tile of block column K0 for this row; the tile of block column K0 + i sits i * bstride doubles further
    constexpr int NCH = decltype(nch_c)::value;
    v4d s0 = zero4, s1 = zero4;
    v4d buf[2][U];
    const int last = max(J - 1 - K0, 0);
#pragma unroll
    for (int u = 0; u < U; u++) buf[0][u] = *reinterpret_cast<const SFT_G v4d*>(brow + (long)min(u, last) * bstride + 4 * lane);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      if (c + 1 < NCH) {
#pragma unroll
        for (int u = 0; u < U; u++)
          buf[(c + 1) & 1][u] = *reinterpret_cast<const SFT_G v4d*>(brow + (long)min(U * (c + 1) + u, last) * bstride + 4 * lane);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int K = K0 + U * c + u;
        if (K < J) mfma4(lds_tile_read(rowJ + (size_t)(J - K - 1) * TS * TS), buf[c & 1][u], s0, s1);
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
    for (int u = 0; u < U; u++) buf[0][u] = *reinterpret_cast<const SFT_G v4d*>(brow + (long)min(u, last) * bstride + 4 * lane);
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c + 1 < 4) {
#pragma unroll
        for (int u = 0; u < U; u++)
          buf[(c + 1) & 1][u] = *reinterpret_cast<const SFT_G v4d*>(brow + (long)min(U * (c + 1) + u, last) * bstride + 4 * lane);
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (U * c + u < n) mfma4(buf[c & 1][u], buf[c & 1][u], s0, s1);
    }
    return s0 + s1;
  };
  auto products = [&](int J, int I, const lds_double* rowJ, const SFT_G double* brow, long bstride, int K0) -> v4d {
    const int n = J - K0;
    if (n <= 0) return zero4;
    if (n <= 2 * U) return products_n(std::integral_constant<int, 2>{}, J, rowJ, brow, bstride, K0);
    return products_n(std::integral_constant<int, 4>{}, J, rowJ, brow, bstride, K0);
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
      if (staged[h]) stage[h] = *reinterpret_cast<const SFT_G v4d*>(Ltg + wtile_off(tpr, K, dist) + 4 * lane);
    }
    if (d == 0) {
      // the diagonal tile: look-ahead tile minus the product with block column J-1 (the staged tile (J, J-1) twice), Cholesky
      __builtin_amdgcn_s_setprio(3);
      v4d dt = dlook;
      if (J >= 1) {
        const v4d a = lds_tile_read(rowJ);
        v4d s0 = zero4, s1 = zero4;
        mfma4(a, a, s0, s1);
        dt -= s0 + s1;
      }
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (crow + 4 * q == ccol && TS * J + ccol < Dn) dt[q] += lambda;
      v4d w = dt;
      const bool ok = chol_inv_blocked(dt, w);
      if (!ok && lane == 0) ctl->fact_ok = 0;
      lds_double* dst = LinvK + ccol * TP + crow;
#pragma unroll
      for (int q = 0; q < 4; q++) dst[4 * q] = w[q];
      *reinterpret_cast<SFT_G v4d*>(Linv_g + (size_t)J * TS * TS + 4 * lane) = w;
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
      dlook = h - squares(Ltg + wtile_off(tpr, K0, I - K0), kstride, J - K0);
    }
    // border: accTb[j][i] = Hbord[i][16 J + j] - sum_K (L(J,K) Lb(K)^T)[j][i], i < 7
    v4d accTb = zero4;
    const bool bwave = d == 7;
    if (bwave) {
      v4d h = zero4;
      if (ccol < SFT_BORDER) {
#pragma unroll
        for (int q = 0; q < 4; q++) h[q] = Hbord[(size_t)ccol * Dnp + TS * J + crow + 4 * q];
      }
      const int K0 = max(0, J - wb);
      accTb = h - products(J, nT, rowJ, LbTg + (size_t)K0 * TS * TS, (long)TS * TS, K0);
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (staged[h]) lds_tile_write(rowN + (size_t)(1 + wave + 8 * h) * TS * TS, stage[h]);
    lds_barrier();                                       // W_J is published
    // ---- TRSM: X^T = W accT (kept for the factor), X = acc W^T (kept for the back substitution) ----
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
        for (int r = k + 1; r < 6; r++) v -= C[r * 7 + k] * P.x[Dnp + r];
        P.x[Dnp + k] = v / C[k * 7 + k];
      }
  }
  __syncthreads();
}

// Back substitution for the wide band: x_J = W_J^T (y_J - sum_{I > J} X(I,J)^T x_I - L_cJ^T x_cam), block columns from the
// last to the first; wave w forms the partial products of tiles (J + d, J), d = w + 1 and w + 9; wave 0 finishes the block.
__device__ __noinline__ void backsub_wide(const SftDev& P, Ctl* ctl, double* ws) {
  constexpr int NW = 8, RPW = WB / NW, RING = 32;
  if (!ctl->fact_ok) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Dnp = ((uni(P.Dn) + NB - 1) / NB) * NB;
  const int nT = Dnp / TS;
  const int tpr = uni(P.tpr), wb = uni(P.wbt);
  lds_double* xw = to_lds(ws);                 // ring of RING x-tiles
  lds_double* part = xw + TS * RING;           // slot 0: camera rows, slots 1..WB: sub-diagonal tiles
  const int crow = lane >> 4, ccol = lane & 15;
  const double xc = (lane < 6) ? P.x[Dnp + lane] : 0.0;
  double xcr[6];
#pragma unroll
  for (int r = 0; r < 6; r++) xcr[r] = bcast_lane(xc, r);
  const auto Lg = uni(P.Lb);
  const auto Lbord = uni(P.Lbord);
  const auto Linv_g = uni(P.Linv);
  const auto xg = uni(P.x);
  for (int i = tid; i < TS * (WB + 1); i += 64 * NW) part[i] = 0.0;   // slots beyond wb stay zero
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
  for (int j = 0; j < PF; j++) ring[j] = fetch(nT - 1 - j);
#pragma unroll 1
  for (int base = nT - 1; base >= 0; base -= PF) {
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
