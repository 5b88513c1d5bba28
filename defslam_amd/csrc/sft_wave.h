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
// The 7 camera / right-hand-side rows ride along as border tiles Bd(J)^T (16 x 8) with the same two formulas, the corner as Yb^T Yb.
//
// The border on v_mfma_f64_4x4x4_4b_f64 (r06).  A 16 x 16 x 4 MFMA spends half its 64 cycles on the eight columns a border tile does not have.
// The 4 x 4 x 4 instruction (four independent 4 x 4 blocks, 16 cycles: the same flop rate -- tools/probes/mfma4x4_probe.hip) has the operand
// layout A_b[i][k] lane i + 4 b + 16 k, B_b[k][j] lane j + 4 b + 16 k, D_b[i][j] lane j + 4 b + 16 i, i.e. register q of a tile in accumulator
// order IS its A operand for the k-chunk q with block b = row block b of the tile.  So D = Y_j^T Yb (16 x 8) is, per k-chunk q and column block
// nb = 0, 1, one instruction with A = Y_j[q] as it lies and B = Yb's rows 4q..4q+3, columns 4nb..4nb+3 replicated over the four blocks:
//   QL (result / storage layout of a border tile T, 16 x 8): two registers, reg nb lane (g, c) = T[4 (c >> 2) + g][4 nb + (c & 3)]
//   OL (operand layout):  reg (q, nb) lane (g, c) = T[4 q + g][4 nb + (c & 3)]  =  QL reg nb of lane (g, 4 q + (c & 3))
// A border tile lives in LDS in QL (element (lane, nb) at 2 lane + nb: one 16-byte access per lane) and is read back in OL by four 16-byte
// reads per lane (q = 0..3; lanes c, c + 4, c + 8, c + 12 read the same address: a broadcast).  8 instead of 4 MFMAs per border product, at a
// quarter of the cycles each.
#pragma once

// Section timers of a factorisation (shader clock, accumulated over the steps of problem; lab builds with EXTRA=-DWV_STEP_TRACE): P.dbg[40 + e],
// e = 0 head requests + diagonal read, 1 tile Cholesky, 2 W transposition, 3 TRSM, 4 L stores + row fetch, 5 corner + rows 1-7 + first LDS run,
// 6 wait for memory, 7 deferred back substitution column, 8 row 8 + second LDS run
#if defined(WV_STEP_TRACE) && defined(DSH_LAB)
#define WV_T_DECL long long wv_tr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, wv_tl = clock64()
#define WV_T(e) do { const long long t__ = clock64(); wv_tr[e] += t__ - wv_tl; wv_tl = t__; } while (0)
#define WV_T_ARG , long long (&wv_tr)[9], long long& wv_tl
#define WV_T_PASS , wv_tr, wv_tl
#define WV_T_DUMP(P) do { if ((threadIdx.x & 63) == 0) for (int e__ = 0; e__ < 9; e__++) (P).dbg[40 + e__] = (double)wv_tr[e__]; } while (0)
#else
#define WV_T_DECL do {} while (0)
#define WV_T(e) do {} while (0)
#define WV_T_ARG
#define WV_T_PASS
#define WV_T_DUMP(P) do {} while (0)
#endif

// ---- physical tile of window tile (ring row r = I mod 8, d = I - J in 0..7) ------------------------------------------------------
__host__ __device__ constexpr int wv_phys(int r, int d) {
  return d == 0 ? r : (d <= 3 ? 8 * d + r : (d == 4 ? 32 + (r & 3) : 8 * (8 - d) + ((r + 8 - d) & 7)));
}
constexpr int WV_AGPR_TILES = 32;                 // physical tiles 0..31 = a[0:255]; 32..35 = LDS window tiles 0..3 (the d = 4 tiles)
// LDS of a wave, in doubles (40 KB: four waves per CU):
constexpr int WV_L_WIN = 0;                       // 4 window tiles x 256
constexpr int WV_L_BORD = 1024;                   // 8 border tiles Bd(J)^T, ring slot J mod 8, COMPACT: only the lanes c < 8 hold data (7 rows + a zero row), 128 doubles each
constexpr int WV_L_WSCR = 2048;                   // 16 x 17 transposition scratch of W (+ pad)
constexpr int WV_L_CN = 2336;                     // 7 x 7 corner + x_cam
constexpr int WV_L_XRING = 2400;                  // back substitution of the PREVIOUS problem: ring of the last eight x tiles
constexpr int WV_L_LAND = 2528;                   // ... and the landing buffer of one block column of its L: [Yb | Y_1 .. Y_8 | W], 10 x 256
constexpr int WV_LDS_DOUBLES = WV_L_LAND + 10 * 256;   // 5088 doubles = 40704 bytes
typedef double v2d_w __attribute__((ext_vector_type(2)));
#ifndef WV_DMA_AUX
#define WV_DMA_AUX 0     // cache policy bits of the LDS-DMA requests of the deferred back substitution (2 = non-temporal)
#endif
using lds_v2d = __attribute__((address_space(3))) v2d_w;

// MFMA result -> any reader other than the next MFMA that accumulates into it: 16 passes + write-back (the compiler pads nothing
// inside or behind an asm statement)
#define WV_NOP_MFMA_RESULT "s_nop 15\n\ts_nop 3"

// ---- MFMA on a hand-pinned accumulator tile -----------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ void wv_upd_agpr(const v4d& A, const v4d& B, int q0 = 0) {   // a[T] -= A^T-chunks x B-chunks
  static_assert(T >= 0 && T < WV_AGPR_TILES, "accumulator tile");
#define WV_MF(q) asm volatile("v_mfma_f64_16x16x4_f64 a[%c0:%c1], %2, %3, a[%c0:%c1] neg:[1,0,0]" ::"i"(8 * T), "i"(8 * T + 7), "v"(A[q]), "v"(B[q]) : "memory")
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
#define WV_MF(q) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(C) : "v"(A[q]), "v"(B[q]) : "memory")
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
// Y = W D with the 4 x 4 x 4 MFMA (r06).  W is LOWER TRIANGULAR: of the sixteen 4 x 4 blocks of the product only the ten with k <= q
// contribute, Y[row block q] = sum_{k <= q} W(q, k) D[row block k] -- ten instructions of 16 cycles where the 16 x 16 x 4 form spends four of
// 64 on a full matrix.  A_b[i][kk] = W[4q + i][4k + kk] is the same for the four blocks b (column blocks of D): `WvTri::a[q (q + 1) / 2 + k]`,
// read from the W scratch in LDS as lane (g, c) = W[4q + (c & 3)][4k + g]; B = register k of D as it lies; the result is register q of Y.
// Order of issue: the accumulations into one register are never adjacent (a dependent 4 x 4 x 4 pair needs four wait states; one instruction
// between them provides them): (3,0) (2,0) (3,1) (2,1) (3,2) (2,2) (3,3) (1,0) (0,0) (1,1).
struct WvTri { double a[10]; };
__device__ __forceinline__ void wv_tri_load(const lds_double* wscr, int lane, WvTri& A) {   // wscr: W as [row][col] at row * 17 + col
  const lds_double* p = wscr + (lane & 3) * 17 + (lane >> 4);
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int k = 0; k <= q; k++) A.a[q * (q + 1) / 2 + k] = p[68 * q + 4 * k];
}
#define WV_TRI_BODY(D0, D1, D2, D3)                          \
  "v_mfma_f64_4x4x4_4b_f64 %3, %10, " D0 ", 0\n\t"  /* (3,0) */ \
  "v_mfma_f64_4x4x4_4b_f64 %2, %7, " D0 ", 0\n\t"   /* (2,0) */ \
  "v_mfma_f64_4x4x4_4b_f64 %3, %11, " D1 ", %3\n\t" /* (3,1) */ \
  "v_mfma_f64_4x4x4_4b_f64 %2, %8, " D1 ", %2\n\t"  /* (2,1) */ \
  "v_mfma_f64_4x4x4_4b_f64 %3, %12, " D2 ", %3\n\t" /* (3,2) */ \
  "v_mfma_f64_4x4x4_4b_f64 %2, %9, " D2 ", %2\n\t"  /* (2,2) */ \
  "v_mfma_f64_4x4x4_4b_f64 %3, %13, " D3 ", %3\n\t" /* (3,3) */ \
  "v_mfma_f64_4x4x4_4b_f64 %1, %5, " D0 ", 0\n\t"   /* (1,0) */ \
  "v_mfma_f64_4x4x4_4b_f64 %0, %4, " D0 ", 0\n\t"   /* (0,0) */ \
  "v_mfma_f64_4x4x4_4b_f64 %1, %6, " D1 ", %1"        /* (1,1) */
template <int T>
__device__ __forceinline__ void wv_trsm4_agpr(v4d& Y, const WvTri& A) {   // D = accumulator tile T
  asm volatile(WV_TRI_BODY("a[%c14:%c15]", "a[%c16:%c17]", "a[%c18:%c19]", "a[%c20:%c21]")
               : "=&v"(Y[0]), "=&v"(Y[1]), "=&v"(Y[2]), "=&v"(Y[3])
               : "v"(A.a[0]), "v"(A.a[1]), "v"(A.a[2]), "v"(A.a[3]), "v"(A.a[4]), "v"(A.a[5]), "v"(A.a[6]), "v"(A.a[7]), "v"(A.a[8]), "v"(A.a[9]),
                 "i"(8 * T), "i"(8 * T + 1), "i"(8 * T + 2), "i"(8 * T + 3), "i"(8 * T + 4), "i"(8 * T + 5), "i"(8 * T + 6), "i"(8 * T + 7)
               : "memory");
}
__device__ __forceinline__ void wv_trsm4_vgpr(v4d& Y, const WvTri& A, const v4d& D) {
  asm volatile("s_nop 1\n\t" WV_TRI_BODY("%14", "%15", "%16", "%17")
               : "=&v"(Y[0]), "=&v"(Y[1]), "=&v"(Y[2]), "=&v"(Y[3])
               : "v"(A.a[0]), "v"(A.a[1]), "v"(A.a[2]), "v"(A.a[3]), "v"(A.a[4]), "v"(A.a[5]), "v"(A.a[6]), "v"(A.a[7]), "v"(A.a[8]), "v"(A.a[9]),
                 "v"(D[0]), "v"(D[1]), "v"(D[2]), "v"(D[3])
               : "memory");
}
// the d = 8 corner tile of a band whose leading row chunk is structurally zero (q8 == 1, e.g. kd = 122): D[0] = 0, so the terms k = 0 drop out and
// row block 0 of Y is zero -- six instructions: (3,1) (2,1) (3,2) (2,2) (3,3) (1,1)
__device__ __forceinline__ void wv_trsm4_vgpr_q1(v4d& Y, const WvTri& A, const v4d& D) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %2, %6, %9, 0\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %1, %4, %9, 0\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %2, %7, %10, %2\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %1, %5, %10, %1\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %2, %8, %11, %2\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %0, %3, %9, 0"
               : "=&v"(Y[1]), "=&v"(Y[2]), "=&v"(Y[3])
               : "v"(A.a[2]), "v"(A.a[4]), "v"(A.a[5]), "v"(A.a[7]), "v"(A.a[8]), "v"(A.a[9]), "v"(D[1]), "v"(D[2]), "v"(D[3])
               : "memory");
  Y[0] = 0.0;
}
#undef WV_TRI_BODY
// the values of MFMA-written VGPRs may be read by compiler code behind this statement
__device__ __forceinline__ void wv_mfma_fence() { asm volatile(WV_NOP_MFMA_RESULT ::: "memory"); }

// accumulator tile -> VGPRs (the diagonal tile on its way to the tile Cholesky)
template <int T>
__device__ __forceinline__ v4d wv_read_agpr() {
  int r[8];
  // (the diagonal tile was last written at the head of the previous step's update: thousands of cycles ago)
  asm volatile("v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c9]\n\tv_accvgpr_read_b32 %2, a[%c10]\n\tv_accvgpr_read_b32 %3, a[%c11]\n\t"
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
// one register (8 bytes per lane) of it: the gathers of a row are issued one at a time between the MFMAs of the update
template <int T, int Q>
__device__ __forceinline__ void wv_gather1_agpr(const SFT_G double* base, unsigned off) {
  asm volatile("global_load_dwordx2 a[%c2:%c3], %0, %1" : : "v"(off), "s"(base), "i"(8 * T + 2 * Q), "i"(8 * T + 2 * Q + 1) : "memory");
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
// border tiles (16 x 8: 7 rows of the border + a zero one) in the quad layout QL of the 4 x 4 x 4 MFMA (head of the file)
struct WvB2 { double n0, n1; };                 // QL: reg nb = columns 4 nb .. 4 nb + 3
struct WvBO { double v[4][2]; };                // OL: [q][nb]
__device__ __forceinline__ WvB2 wv_bq_load(const lds_double* tile, int lane) {
  const v2d_w a = reinterpret_cast<const lds_v2d*>(tile)[lane];
  return WvB2{a.x, a.y};
}
__device__ __forceinline__ void wv_bq_store(lds_double* tile, int lane, const WvB2& v) {
  reinterpret_cast<lds_v2d*>(tile)[lane] = (v2d_w){v.n0, v.n1};
}
// the lane's OL source lane for q = 0: (g, c & 3); q adds 4 lanes = 4 v2d
__device__ __forceinline__ int wv_bo_lane(int lane) { return (lane & 0x30) | (lane & 3); }
__device__ __forceinline__ void wv_bo_load(const lds_double* tile, int lane, WvBO& o) {
  const lds_v2d* p = reinterpret_cast<const lds_v2d*>(tile) + wv_bo_lane(lane);
#pragma unroll
  for (int q = 0; q < 4; q++) { const v2d_w a = p[4 * q]; o.v[q][0] = a.x; o.v[q][1] = a.y; }
}
// C (QL) -= A^T-chunks x B (OL): 8 MFMAs of 16 cycles, the two column blocks alternate (independent accumulators)
__device__ __forceinline__ void wv_upd_b4(WvB2& C, const v4d& A, const WvBO& B, int q0 = 0) {
#define WV_MF(q) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0 neg:[1,0,0]\n\tv_mfma_f64_4x4x4_4b_f64 %1, %2, %4, %1 neg:[1,0,0]" \
                              : "+v"(C.n0), "+v"(C.n1) : "v"(A[q]), "v"(B.v[q][0]), "v"(B.v[q][1]) : "memory")
  asm volatile("s_nop 1" : "+v"(C.n0), "+v"(C.n1));
  if (q0 <= 0) WV_MF(0);
  if (q0 <= 1) WV_MF(1);
  if (q0 <= 2) WV_MF(2);
  WV_MF(3);
#undef WV_MF
}
// Yb (QL) = W Bd(k)^T: A = acc(W^T)[q], B = Bd(k)^T in OL
__device__ __forceinline__ void wv_trsm_b4(WvB2& Y, const v4d& Wt, const WvBO& B) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %0, %2, %6, 0\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %1, %2, %7, 0\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %0, %3, %8, %0\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %1, %3, %9, %1\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %0, %4, %10, %0\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %1, %4, %11, %1\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %0, %5, %12, %0\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %1, %5, %13, %1"
               : "=&v"(Y.n0), "=&v"(Y.n1)
               : "v"(Wt[0]), "v"(Wt[1]), "v"(Wt[2]), "v"(Wt[3]), "v"(B.v[0][0]), "v"(B.v[0][1]), "v"(B.v[1][0]), "v"(B.v[1][1]), "v"(B.v[2][0]), "v"(B.v[2][1]),
                 "v"(B.v[3][0]), "v"(B.v[3][1])
               : "memory");
}
// a QL tile (LDS image or landing buffer) read as the tile in ACCUMULATOR order restricted to its 8 columns: lane (g, c), register q =
// T[g + 4 q][c & 7] (the lanes c >= 8 mirror c - 8: finite numbers everywhere): element at 2 (16 g + 4 q + (c & 3)) + ((c >> 2) & 1)
__device__ __forceinline__ int wv_bstd_index(int lane) { return 2 * wv_bo_lane(lane) + ((lane >> 2) & 1); }

// ---- 16 x 16 Cholesky + inverse, every MFMA as asm on VGPR tiles (tile_chol.h: chol_inv_blocked is the compiler-scheduled original) --
// A lane's value moved along its 16-lane row.  row_ror reads a valid lane for every lane, so the destination's previous content never shows --
// __builtin_amdgcn_mov_dpp leaves it undefined, where update_dpp(0, ...) (dpp_mov in sft_kernels.hip) makes the compiler write a zero into
// the destination in front of every move: 32 v_mov_b32 per factor step in the deferred back substitution's all-reduce.
template <int CTRL>
__device__ __forceinline__ double wv_dpp_mov(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ bool wv_chol_inv(v4d& a, v4d& w) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  w = (v4d){(g == c) ? 1.0 : 0.0, (g + 4 == c) ? 1.0 : 0.0, (g + 8 == c) ? 1.0 : 0.0, (g + 12 == c) ? 1.0 : 0.0};
  double plast = 1.0;
#pragma unroll
  for (int J = 0; J < 4; J++) {
    const double aJ = a[J];
    const double sel = chol4_inverse_operand(aJ, J, g, c & 3, plast);   // (tile_chol.h: the chain of dependent operations of a block step; M replicated over the quads)
    const double wJ = w[J];
    double zw, lp;
    if (J < 3) {
      // Zw = M W[4J..4J+3, :] and Z = M A[4J..4J+3, :]: 4 x 16 results, one 4 x 4 x 4 MFMA each (tile_chol.h); six wait states before a vector
      // instruction may read a 4-pass result
      asm volatile("s_nop 1\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, 0\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 %1, %2, %4, 0\n\t"
                   "s_nop 7"
                   : "=&v"(zw), "=&v"(lp)
                   : "v"(sel), "v"(wJ), "v"(aJ));
      const double nlp = -lp;
      const double below = (c >= 4 * J + 4) ? nlp : 0.0;
      asm volatile("s_nop 1\n\t"
                   "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\t"
                   "v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\t" WV_NOP_MFMA_RESULT
                   : "+v"(a), "+v"(w)
                   : "v"(nlp), "v"(lp), "v"(below), "v"(zw));
      w[J] = zw;
    } else {
      asm volatile("s_nop 1\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0\n\ts_nop 7" : "=&v"(zw) : "v"(sel), "v"(wJ));
      w[J] = zw;
    }
  }
  return plast > 0.0;
}

// ---- per-problem constants of a factorisation, all wave-uniform --------------------------------------------------------------------
struct WvProb {
  int Dn, Dnp, nT, q8;                    // q8: leading row chunks of the d = 8 corner tile that are structurally zero (kd = 122: 1)
  double lambda, lam_corner;
  unsigned bofs;                          // per lane: (c & 3) * Dnp + 4 (c >> 2) + g, the lane's QL element (nb = 0) of a border tile relative to column 16 J (doubles); nb = 1: + 4 Dnp
  const SFT_G double* Hc;
  const SFT_G uint32_t* hgl;              // the TRANSPOSED gather lists (SftDev::hgatherT): 16 bytes per lane and tile
  const SFT_G double* Hbord;
  SFT_G double *Lg, *Linv;
};

// gather list of tile (I, d) (transposed lists: lane (g, c), register q takes element [row c][column g + 4q] of H(I, I-d)): one 16-byte load
// per lane, wave-uniform base + lane * 16.  Rows behind the matrix take the all-zero list row nT.
typedef unsigned v4u_w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wv_list(const WvProb& W, int I, int d, int lane, unsigned (&o)[4]) {
  const SFT_G v4u_w* row = reinterpret_cast<const SFT_G v4u_w*>(W.hgl + ((size_t)(I < W.nT ? I : W.nT) * (BT + 1) + d) * 256);
  const v4u_w v = row[lane];
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ v4d wv_gather_vgpr(const WvProb& W, const unsigned (&o)[4]) {
  const auto b8 = reinterpret_cast<const SFT_G char*>(W.Hc);
  v4d v;
#pragma unroll
  for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const SFT_G double*>(b8 + o[q]);
  return v;
}
// border tile Bd(J)^T (QL) from the 8-row border of H (rows 0-5 camera, 6 right-hand side, 7 zero): lane (g, c), register nb =
// Hbord[4 nb + (c & 3)][16 J + 4 (c >> 2) + g]
__device__ __forceinline__ WvB2 wv_border_fresh(const WvProb& W, int J, int lane) {
  const SFT_G double* p = W.Hbord + TS * (J < W.nT ? J : W.nT);   // (behind the matrix, J >= nT: the row pitch is Dnp, so what is read there is finite data of the NEXT border row -- never consumed: those ring slots are never a pivot column and their Y operands are zero)
  (void)lane;
  return WvB2{p[W.bofs], p[W.bofs + 4 * (unsigned)W.Dnp]};
}

// Lists of the eight tiles (I, d), d = 0..7, of a row that enters the window: 32 offsets per lane, requested a step ahead.
struct WvRowList { unsigned o[8][4]; };
__device__ __forceinline__ void wv_row_list(const WvProb& W, int I, int lane, WvRowList& L) {
#pragma unroll
  for (int d = 0; d < 8; d++) wv_list(W, I, d, lane, L.o[d]);
}
// The row's elements: accumulator tiles straight from memory; the d = 4 tile (LDS-resident) comes back in `fresh4` and is stored by the
// caller once it has arrived (storing it here would wait for memory in the middle of a step).  PH = I mod 8; dmax: the prologue rows have
// fewer tiles (J >= 0).  The caller waits (wv_wait_vm) before the first MFMA on these tiles.
template <int PH, int D>
__device__ __forceinline__ void wv_row_fetch_d(const WvProb& W, const WvRowList& L, int dmax, v4d& fresh4) {
  if constexpr (D < 8) {
    if (D <= dmax) {
      if constexpr (D == 4) fresh4 = wv_gather_vgpr(W, L.o[D]);
      else wv_gather_agpr<wv_phys(PH, D)>(W.Hc, L.o[D][0], L.o[D][1], L.o[D][2], L.o[D][3]);
    }
    wv_row_fetch_d<PH, D + 1>(W, L, dmax, fresh4);
  }
}

// ---- the back substitution of the PREVIOUS problem of this wave, one block column per factor step of the current one ----------------------
// All waves of a launch reach their back substitutions at the same time, and the 1.9 MB of L each of them reads back then meet at the HBM
// limit (measured: 0.19 M cycles alone, 0.8 M with 1024 waves in step -- a quarter of a factorisation).  So a wave defers it: while it
// factors its next problem, block column J of the previous L arrives in an LDS landing buffer by LDS-DMA (no registers; memory
// instructions issue in the shadow of the MFMAs) and costs ~110 VALU instructions per step.
struct WvPrev {
  const SFT_G double* Lg;     // block columns [Yb | Y_1 .. Y_8], 9 x 2 KB each
  const SFT_G double* Linv;   // W tiles
  SFT_G double* x;
  int nT, active;
  double xb;                  // per lane: x_cam[c] for c < 6, -1 for c == 6 (the right-hand side column of the border), else 0
};
// (r06: the requests as asm -- scalar base + the lane's 32-bit offset + an immediate that advances the global AND the LDS address alike (the
// slots are 1 KB apart on both sides), M0 = the LDS slot of each group of four.  Through the builtin every request cost a 64-bit vector add for
// its address and a v_readlane for its LDS base out of the SGPR spill register: 38 vector instructions per step on the one FP64 pipe.)
__device__ __forceinline__ void wv_bs_request(const WvPrev& Q, int J, lds_double* land, int lane) {
  const unsigned voff = 16u * (unsigned)lane;
  const SFT_G char* col = reinterpret_cast<const SFT_G char*>(Q.Lg + ((size_t)J * (BT + 1)) * 256);    // wave-uniform
  const SFT_G char* w = reinterpret_cast<const SFT_G char*>(Q.Linv + (size_t)J * 256);
  const unsigned l0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)land);                        // LDS byte address of the landing buffer
  // slot 0: Yb (1 KB; the second KB of the slot is not used), slots 1..8: Y_1 .. Y_8 = KB 2..17
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:3072"
               :: "v"(voff), "s"(col), "s"(l0) : "memory", "m0");
#pragma unroll
  for (int grp = 1; grp < 5; grp++) {
    const SFT_G char* cg = col + 4096 * grp;
    const unsigned lg = l0 + 4096u * grp;
    if (grp < 4)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:3072"
                   :: "v"(voff), "s"(cg), "s"(lg) : "memory", "m0");
    else   // KB 16, 17
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:1024"
                   :: "v"(voff), "s"(cg), "s"(lg) : "memory", "m0");
  }
  // W: 2 KB behind the eighteen of the column
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:1024"
               :: "v"(voff), "s"(w), "s"(l0 + 18u * 1024u) : "memory", "m0");
}
// (Measured and dropped, r06: the nineteen requests issued one at a time behind the first tiles of the trailing update instead of as a burst
// here -- profiles/r06/ab/phases_ab_v7_dma_spread.log: FACTOR 264.4 against 261.2 ms per step; the 1.5-2.0 k cycles the deferred column costs
// per step under a full batch are the latency of its own chains of dependent operations, not the issue of its requests.)
// Column J from the landing buffer: S_q = sum_d Y_d[q] x_{J+d}[c] + Yb[q] xb[c], summed over the 16 lanes of a row, is (L^T x)_tail + the
// camera term - y at index g + 4q; x_J[c] = sum_r W[r][c] (-S[r]).  The border tile is a ninth tile whose "x" is (x_cam, -1, 0, ...).
__device__ __forceinline__ void wv_bs_column(const WvPrev& Q, int J, const lds_double* land, lds_double* xring, int lane) {
  const int c = lane & 15;
  const lds_v2d* t2 = reinterpret_cast<const lds_v2d*>(land) + 2 * lane;   // the lane's 32 bytes of tile 0; tile i: + 128 * i
  double s[4];
  {
    const lds_double* tb = land + wv_bstd_index(lane);   // Yb is stored in QL (1 KB): element [g + 4 q][c & 7]; the lanes c >= 8 mirror c - 8 (times xb = 0)
    s[0] = tb[0] * Q.xb; s[1] = tb[8] * Q.xb; s[2] = tb[16] * Q.xb; s[3] = tb[24] * Q.xb;
  }
#pragma unroll
  for (int d = 1; d <= 8; d++) {
    const double xr = xring[((J + d) & 7) * 16 + c];   // tile rows behind the matrix: the ring still holds its zeros
    const v2d_w a = t2[128 * d], b = t2[128 * d + 1];
    s[0] = fma(a.x, xr, s[0]); s[1] = fma(a.y, xr, s[1]); s[2] = fma(b.x, xr, s[2]); s[3] = fma(b.y, xr, s[3]);
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {   // all-reduce over the 16 lanes of a row (fixed butterfly: row_ror 8, 4, 2, 1)
    s[q] += wv_dpp_mov<0x128>(s[q]);
    s[q] += wv_dpp_mov<0x124>(s[q]);
    s[q] += wv_dpp_mov<0x122>(s[q]);
    s[q] += wv_dpp_mov<0x121>(s[q]);
  }
  const v2d_w wa = t2[128 * 9], wb = t2[128 * 9 + 1];
  double p = wa.x * -s[0];
  p = fma(wa.y, -s[1], p); p = fma(wb.x, -s[2], p); p = fma(wb.y, -s[3], p);
  p = sum_rows(p);
  if (lane < TS) { xring[(J & 7) * 16 + lane] = p; Q.x[TS * J + lane] = p; }
}
// one column of the deferred back substitution + the request of the next one (call behind a wv_wait_vm: the column has landed)
__device__ __forceinline__ void wv_bs_step(const WvPrev& Q, int J, lds_double* lds, int lane) {
  wv_bs_column(Q, J, lds + WV_L_LAND, lds + WV_L_XRING, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of the landing buffer has returned before the DMA overwrites it
  if (J >= 1) wv_bs_request(Q, J - 1, lds + WV_L_LAND, lane);
}
__device__ __forceinline__ void wv_bs_begin(const WvPrev& Q, lds_double* lds, int lane) {
  lds_double* xring = lds + WV_L_XRING;
  xring[lane] = 0.0; xring[64 + lane] = 0.0;
  wv_bs_request(Q, Q.nT - 1, lds + WV_L_LAND, lane);
}
// The whole back substitution at once (the last problem of a wave; lab A/B): block columns two ahead in registers.
__device__ __forceinline__ void wv_backsub_now(const WvPrev& Q, int lane) {
  const int c = lane & 15;
  (void)c;
  double xr[8];
#pragma unroll
  for (int s = 0; s < 8; s++) xr[s] = 0.0;
  struct Col { v4d t[10]; };   // [0] Yb, [1..8] Y_d, [9] W
  auto fetch = [&](int J) -> Col {
    Col C;
    if (J >= 0) {
      const SFT_G double* col = Q.Lg + ((size_t)J * (BT + 1)) * 256 + 4 * lane;
#pragma unroll
      for (int i = 1; i <= 8; i++) C.t[i] = (J + i < Q.nT) ? *reinterpret_cast<const SFT_G v4d*>(col + 256 * i) : (v4d){0.0, 0.0, 0.0, 0.0};
      {   // Yb: stored in QL (1 KB)
        const SFT_G double* tb = col - 4 * lane + wv_bstd_index(lane);
        C.t[0] = (v4d){tb[0], tb[8], tb[16], tb[24]};
      }
      C.t[9] = *reinterpret_cast<const SFT_G v4d*>(Q.Linv + (size_t)J * 256 + 4 * lane);
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
    for (int q = 0; q < 4; q++) s[q] = C.t[0][q] * Q.xb;
#pragma unroll
    for (int dd = 1; dd <= 8; dd++)
#pragma unroll
      for (int q = 0; q < 4; q++) s[q] = fma(C.t[dd][q], xr[(ph + dd) & 7], s[q]);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      s[q] += wv_dpp_mov<0x128>(s[q]);
      s[q] += wv_dpp_mov<0x124>(s[q]);
      s[q] += wv_dpp_mov<0x122>(s[q]);
      s[q] += wv_dpp_mov<0x121>(s[q]);
    }
    double p = C.t[9][0] * -s[0];
#pragma unroll
    for (int q = 1; q < 4; q++) p = fma(C.t[9][q], -s[q], p);
    p = sum_rows(p);
    xr[ph] = p;
    if (lane < TS) Q.x[TS * J + lane] = p;
  };
  int J = Q.nT - 1;
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

// ---- one factor step, ring phase PH = k mod 8 ------------------------------------------------------------------------------------------
struct WvState {
  v4d Y[9];          // Y[i], i = 1..8: X(k+i,k)^T ([0] is not used: the border tile Xb^T of column k is Yq / Yo)
  WvB2 Yq;           // Xb^T of column k in QL, as its TRSM leaves it: stored to L behind the first update tile
  WvBO Yo;           // ... and in OL (through the W scratch in LDS): the B operand of the border updates
  double corner;     // 7 x 7 camera corner (+ right-hand side row) as the four blocks of a 4 x 4 x 4 MFMA: lane (g, c) = [4 (c >> 3) + g][4 ((c >> 2) & 1) + (c & 3)]; lower triangle meaningful
  v4d araw;          // raw tile (k+8, k)^T of this step's column (the d = 8 corner of the band): requested at the head of the step
  WvB2 bnext;        // raw border tile of column k+8 (QL; enters the ring this step): requested at the head of the step
  WvRowList rl;      // gather lists of the row that enters the window this step (row k+8): requested at the head of the step
  unsigned al[4];    // gather list of tile (k+9, k+1), the next step's araw -- the only request that is carried across the update
  unsigned off_lane, off_bord;   // byte offsets of the lane inside a 2 KB tile slot of L (32 l) and inside the 1 KB border tile (QL: 16 l)
  int ok;
};

template <int PH, int I, int IEND>
__device__ __forceinline__ void wv_trsm_cols(WvState& S, const WvTri& Wt, int k, int nT, const v4d& D4, int lane) {
  if constexpr (I <= IEND) {
    // unconditional: behind the matrix the window tiles are the zeros they were gathered as (list row nT), W 0 = 0 -- a branch per tile made
    // the compiler write zeros into all nine Y tiles in front of it, every step (28 register moves), to save 36 tile products per factorisation
    (void)k; (void)nT;
    if constexpr (I == 4) {
      wv_trsm4_vgpr(S.Y[4], Wt, D4);   // (the LDS-resident d = 4 tile of the column: requested in front of the W transposition)
    } else {
      wv_trsm4_agpr<wv_phys((PH + I) & 7, I)>(S.Y[I], Wt);
    }
    wv_trsm_cols<PH, I + 1, IEND>(S, Wt, k, nT, D4, lane);
  }
}

// window tiles (k+I, k+J), 1 <= J <= I, that live in the accumulator file (the d = 4 tiles: wv_lds_pipe).  Behind the n-th tile of rows
// 1..7 one 1 KB half of block column k of L is stored (18 halves: Yb, Y_1 .. Y_8) -- a wave has 63 memory operations in flight at most,
// and stores retire slowly when HBM is busy: issued as one burst behind the TRSM (with the 36 gathers of the entering row) they stalled the
// in-order instruction stream, MFMAs included, for thousands of cycles per step; one at a time between MFMAs they cost nothing.
__host__ __device__ constexpr int wv_tile_index(int I, int J) {   // rank of (I, J) among the accumulator-file tiles of rows 1..7, row-major
  int n = 0;
  for (int i = 1; i <= 7; i++)
    for (int j = 1; j <= i; j++) {
      if (i == I && j == J) return n;
      if (i - j != 4) n++;
    }
  return n;
}
// request number SLOT (0..27) of the row that enters the window: register SLOT % 4 of its accumulator-file tile number SLOT / 4 (d = 0..3, 5..7)
template <int PH, int SLOT>
__device__ __forceinline__ void wv_fetch_slot(const WvProb& W, const WvRowList& L) {
  if constexpr (SLOT < 28) {
    constexpr int t = SLOT / 4, q = SLOT % 4, d = t < 4 ? t : t + 1;
    wv_gather1_agpr<wv_phys(PH, d), q>(W.Hc, L.o[d][q]);
  }
}
template <int PH, bool TAIL, int I, int J>
__device__ __forceinline__ void wv_update_tiles(const WvProb& W, WvState& S, int k, int nT, int q8, SFT_G double* col) {
  if constexpr (I <= 8) {
    constexpr int d = I - J, r = (PH + I) & 7;
    if constexpr (d != 4) {
      // (TAIL: the last eight steps of a factorisation, where rows of the window lie behind the matrix -- the others carry no branch)
      if (!TAIL || k + I < nT) wv_upd_agpr<wv_phys(r, d)>(S.Y[J], S.Y[I], (I == 8) ? q8 : 0);
      if constexpr (I <= 7) {
        constexpr int n = wv_tile_index(I, J);
        if constexpr (n < 18) {   // half (n & 1) of tile n / 2: lanes' registers 2 (n & 1), 2 (n & 1) + 1
          // scalar base (the slot of the tile, wave-uniform: SALU) + the lane's 32-bit offset + immediate: as a pointer expression every
          // store cost one or two 64-bit vector adds for its address (35 vector instructions per step on the one FP64 pipe)
          const SFT_G double* slot = col + 256 * (n >> 1);
          if constexpr (n == 0) {   // the border tile Yb in QL: 16 bytes per lane, 1 KB, one store (n == 1: its slot has no second half)
            const v2d_w yb = (v2d_w){S.Yq.n0, S.Yq.n1};
            asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(S.off_bord), "v"(yb), "s"(slot) : "memory");
          } else if constexpr (n >= 2) {
            const v4d& y = S.Y[n >> 1];
            const v2d_w half = (v2d_w){y[2 * (n & 1)], y[2 * (n & 1) + 1]};
            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%c3" :: "v"(S.off_lane), "v"(half), "s"(slot), "n"(16 * (n & 1)) : "memory");
          }
        }
        // ... and two of the 28 gathers of the row that enters the window (the tiles of ring row PH are free since the TRSM): like the
        // stores, one burst of them holds up the instruction stream (each scattered 8-byte load keeps the address unit busy ~100 cycles)
        wv_fetch_slot<PH, 2 * n>(W, S.rl);
        wv_fetch_slot<PH, 2 * n + 1>(W, S.rl);
      }
    }
    if constexpr (J < I) wv_update_tiles<PH, TAIL, I, J + 1>(W, S, k, nT, q8, col);
  }
}
template <int PH, bool TAIL, int I0, int I1>
__device__ __forceinline__ void wv_update_rows(const WvProb& W, WvState& S, int k, int nT, int q8, SFT_G double* col) {
  if constexpr (I0 <= I1) {
    wv_update_tiles<PH, TAIL, I0, 1>(W, S, k, nT, q8, col);
    wv_update_rows<PH, TAIL, I0 + 1, I1>(W, S, k, nT, q8, col);
  }
}
// The LDS-resident tiles of a step as a list of tasks: T = 0..6 border tiles Bd(k+T+1)^T, T = 7, 8, 9 the d = 4 window tiles of rows 5, 6, 7
// (first half of the update); T = 10 border tile Bd(k+8)^T, T = 11 window tile (k+8, k+4) (second half: behind the fetch of row k+8).
// Unconditional -- behind the matrix the Y operands are zero -- and software-pipelined: a tile is written back behind the four MFMAs of
// the NEXT one, by which time its own have long left the matrix pipe (no wait states spent per tile; one fence at the end of a run).
template <int PH, int T>
__device__ __forceinline__ lds_double* wv_task_tile(lds_double* lds) {
  if constexpr (T < 7) return lds + WV_L_BORD + 128 * ((PH + T + 1) & 7);
  else if constexpr (T < 10) return lds + WV_L_WIN + 256 * (wv_phys((PH + T - 2) & 7, 4) - WV_AGPR_TILES);   // rows I = 5, 6, 7
  else if constexpr (T == 10) return lds + WV_L_BORD + 128 * (PH & 7);                                        // column k+8: ring slot k mod 8
  else return lds + WV_L_WIN + 256 * (wv_phys(PH & 7, 4) - WV_AGPR_TILES);                                     // row 8: ring row k mod 8
}
template <int T> struct WvTaskTile { using type = typename std::conditional<(T < 7 || T == 10), WvB2, v4d>::type; };   // border tiles: QL, two registers
template <int T> using wv_task_t = typename WvTaskTile<T>::type;
template <int PH, int T>
__device__ __forceinline__ wv_task_t<T> wv_task_load(lds_double* lds, int lane) {
  if constexpr (T < 7 || T == 10) return wv_bq_load(wv_task_tile<PH, T>(lds), lane);
  else return wv_lds_load(wv_task_tile<PH, T>(lds), lane);
}
template <int PH, int T>
__device__ __forceinline__ void wv_task_store(lds_double* lds, int lane, const wv_task_t<T>& C) {
  if constexpr (T < 7 || T == 10) wv_bq_store(wv_task_tile<PH, T>(lds), lane, C);
  else wv_lds_store(wv_task_tile<PH, T>(lds), lane, C);
}
template <int T>
__device__ __forceinline__ void wv_task_mfma(wv_task_t<T>& C, const WvState& S, int q8) {
  if constexpr (T < 7) wv_upd_b4(C, S.Y[T + 1], S.Yo);
  else if constexpr (T < 10) wv_upd_vgpr(C, S.Y[T - 6], S.Y[T - 2]);   // (I, J) = (5, 1), (6, 2), (7, 3)
  else if constexpr (T == 10) wv_upd_b4(C, S.Y[8], S.Yo, q8);
  else wv_upd_vgpr(C, S.Y[4], S.Y[8], q8);
}
// C: tile T, already loaded; Cprev: tile T-1, its MFMAs issued.  The load of tile T+1 is requested in front of T's MFMAs.
template <int PH, int T, int TEND>
__device__ __forceinline__ void wv_lds_pipe_next(const WvState& S, int q8, lds_double* lds, int lane, wv_task_t<T>& C, const wv_task_t<T - 1>& Cprev) {
  if constexpr (T < TEND) {
    wv_task_t<T + 1> Cn = wv_task_load<PH, T + 1>(lds, lane);
    wv_task_mfma<T>(C, S, q8);
    wv_task_store<PH, T - 1>(lds, lane, Cprev);
    wv_lds_pipe_next<PH, T + 1, TEND>(S, q8, lds, lane, Cn, C);
  } else {
    wv_task_mfma<T>(C, S, q8);
    wv_task_store<PH, T - 1>(lds, lane, Cprev);
    wv_mfma_fence();
    wv_task_store<PH, T>(lds, lane, C);
  }
}
template <int PH, int T0, int TEND>
__device__ __forceinline__ void wv_lds_pipe(const WvState& S, int q8, lds_double* lds, int lane) {
  wv_task_t<T0> C = wv_task_load<PH, T0>(lds, lane);
  wv_task_t<T0 + 1> Cn = wv_task_load<PH, T0 + 1>(lds, lane);
  wv_task_mfma<T0>(C, S, q8);
  wv_lds_pipe_next<PH, T0 + 1, TEND>(S, q8, lds, lane, Cn, C);
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
template <int PH, bool TAIL>
__device__ __forceinline__ void wv_step_rest(const WvProb& W, WvState& S, const v4d& w, int k, lds_double* lds, int lane, const WvPrev& Q WV_T_ARG) {
  lds_double* ldsw = lds + WV_L_WIN;
  lds_double* ldsb = lds + WV_L_BORD;
  lds_double* wscr = lds + WV_L_WSCR;
  const int nT = W.nT;
  // The two LDS-resident operands of the TRSM -- the border tile of column k (OL) and the d = 4 window tile -- are requested in FRONT of the
  // W transposition: they travel while W goes through LDS, and the TRSM starts with everything in registers (requested right in front of
  // their first use they cost a full LDS round trip each, ~250 cycles per step, with the matrix pipe idle).
  WvBO Bk;
  wv_bo_load(ldsb + 128 * PH, lane, Bk);
  const v4d D4 = wv_lds_load(ldsw + 256 * (wv_phys((PH + 4) & 7, 4) - WV_AGPR_TILES), lane);
  // W^T in accumulator order: through LDS (element [row][col] at row * 17 + col)
  v4d Wt;
  {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int q = 0; q < 4; q++) wscr[(g + 4 * q) * 17 + c] = w[q];
#pragma unroll
    for (int q = 0; q < 4; q++) Wt[q] = wscr[c * 17 + g + 4 * q];
  }
  WvTri Wq;   // the ten non-zero 4 x 4 blocks of W as operands of the 4 x 4 x 4 TRSM
  wv_tri_load(wscr, lane, Wq);
  WV_T(2);
  // ---- block column k: border Yb = W Bd(k)^T (4 x 4 x 4 MFMAs; first, so that its trip through LDS into the operand layout lies behind
  // the window's TRSMs), Y_i = W D(k+i, k)
  wv_trsm_b4(S.Yq, Wt, Bk);
  wv_trsm_cols<PH, 1, 2>(S, Wq, k, nT, D4, lane);   // (W^T has been read from the scratch: LDS instructions of a wave complete in order)
  wv_bq_store(wscr, lane, S.Yq);        // (8 MFMAs behind its own: complete)
  wv_trsm_cols<PH, 3, 4>(S, Wq, k, nT, D4, lane);
  // ... and back in the operand layouts while the other sixteen MFMAs of the TRSM run: OL for the border updates; for the corner product
  // Yb^T Yb (8 x 8 = the four blocks of ONE 4 x 4 x 4 MFMA per k-chunk, block b = (row block b >> 1, column block b & 1)) the A operand
  // (columns 4 (c >> 3) + (c & 3)) and the B operand (columns c & 7: the tile in accumulator order)
  wv_bo_load(wscr, lane, S.Yo);
  v4d yba, ybs;
  {
    const lds_double* t = wscr + 2 * wv_bo_lane(lane);
    const lds_double* ta = t + (lane >> 3 & 1);
    const lds_double* tb = t + (lane >> 2 & 1);
#pragma unroll
    for (int q = 0; q < 4; q++) { yba[q] = ta[8 * q]; ybs[q] = tb[8 * q]; }
  }
  // (D4 was the B operand of the ten MFMAs right in front of these loads: named here, its registers are not the ones the loads land in --
  // tools/wave_audit.py rule 5)
  asm volatile("" :: "v"(D4[0]), "v"(D4[1]), "v"(D4[2]), "v"(D4[3]));
  wv_trsm_cols<PH, 5, 7>(S, Wq, k, nT, D4, lane);
  if (W.q8 == 1) wv_trsm4_vgpr_q1(S.Y[8], Wq, S.araw);   // (wave-uniform)
  else wv_trsm_vgpr(S.Y[8], Wt, S.araw, W.q8);
  wv_mfma_fence();
  WV_T(3);
  // L leaves from the registers it was computed in: block column k = [Yb | Y_1 .. Y_8] at slots (k, 0..8), 2 KB each (lane l: 32 bytes at
  // 32 l) -- stored half by half between the MFMAs of the update below (wv_update_tiles); tiles behind the matrix are stored as the
  // zeros they are
  SFT_G double* col = W.Lg + ((size_t)k * (BT + 1)) * 256;   // (wave-uniform; the lanes' offsets are S.off_lane / S.off_bord)
  // ---- the ring row / border slot of column k are free: row k+8 enters (its lists came a step ahead), the lists of row k+9 are requested
  v4d fresh4 = wv_gather_vgpr(W, S.rl.o[4]);   // (the accumulator-file tiles of the row: wv_update_tiles)
  // A value that is only loaded and stored may be allocated to the accumulator file by the compiler (memory instructions address it) --
  // onto a window tile.  Naming it as a VGPR operand where it is consumed keeps it out (tools/wave_audit.py checks the ISA for such accesses).
  asm volatile("" : "+v"(S.bnext.n0), "+v"(S.bnext.n1));
  wv_bq_store(ldsb + 128 * PH, lane, S.bnext);
  WV_T(4);
  // ---- trailing update: corner, rows 1..7, the LDS tiles of the first half; then (everything requested has landed) the deferred back
  // substitution's column, row 8 and the LDS tiles of the second half
  // (ONE accumulator: a 4 x 4 x 4 MFMA that accumulates into the result of the instruction right in front of it needs four wait states -- the
  // hardware does not hold it back long enough (tools/probes/mfma4x4_probe.hip, r06: without them three launches of the benched batch were not
  // bit-identical); the border tiles alternate two accumulators, which puts a whole instruction between the dependent ones)
  asm volatile("s_nop 1\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %0, %1, %5, %0 neg:[1,0,0]\n\ts_nop 3\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %0, %2, %6, %0 neg:[1,0,0]\n\ts_nop 3\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %0, %3, %7, %0 neg:[1,0,0]\n\ts_nop 3\n\t"
               "v_mfma_f64_4x4x4_4b_f64 %0, %4, %8, %0 neg:[1,0,0]"
               : "+v"(S.corner)
               : "v"(yba[0]), "v"(yba[1]), "v"(yba[2]), "v"(yba[3]), "v"(ybs[0]), "v"(ybs[1]), "v"(ybs[2]), "v"(ybs[3])
               : "memory");
  wv_update_rows<PH, TAIL, 1, 7>(W, S, k, nT, W.q8, col);
  wv_lds_pipe<PH, 0, 9>(S, W.q8, lds, lane);
  WV_T(5);
  wv_wait_vm();
  WV_T(6);
  {   // second half: border tile Bd(k+8)^T and window tile (k+8, k+4) -- the latter straight from the registers it was fetched into
    WvB2 C10 = wv_task_load<PH, 10>(lds, lane);
    wv_update_rows<PH, TAIL, 8, 8>(W, S, k, nT, W.q8, col);
    wv_task_mfma<10>(C10, S, W.q8);
    wv_task_mfma<11>(fresh4, S, W.q8);
    wv_task_store<PH, 10>(lds, lane, C10);
    wv_mfma_fence();
    wv_task_store<PH, 11>(lds, lane, fresh4);
  }
  // The corner's MFMAs were issued at the head of the update: it is long complete here.  This empty statement takes the corner as an
  // operand, so any register copy the compiler makes of it (the merge of the eight phase bodies) sits BEHIND the whole update -- a copy
  // right behind the MFMA statement would read the registers before the matrix pipe has written them (no hazard padding around asm).
  asm volatile("" : "+v"(S.corner));
  WV_T(8);
}

// prologue: tile rows 0..7 of H into the window.  The compiler does not see the accumulator gathers (asm), so any wait it inserts for a load
// it does see also waits for gathers issued in between -- a round trip of everything in flight.  The rows used to be fetched one after the
// other, lists, gathers, border and the LDS-resident d = 4 tile each (20 such drains: 33-48 k cycles per factorisation with the MFMA pipe idle).
// Now: the gather lists of all 36 tiles first (144 registers; the window is not live yet), ONE wait, every gather back to back, the border
// loads, and what goes to LDS is stored behind the last request.
struct WvTriList { unsigned o[36][4]; };   // tile (R, d), d <= R, at R (R + 1) / 2 + d
template <int R, int D>
__device__ __forceinline__ void wv_prologue_lists(const WvProb& W, int lane, WvTriList& T) {
  if constexpr (R < 8) {
    wv_list(W, R, D, lane, T.o[R * (R + 1) / 2 + D]);
    if constexpr (D < R) wv_prologue_lists<R, D + 1>(W, lane, T);
    else wv_prologue_lists<R + 1, 0>(W, lane, T);
  }
}
template <int R, int D>
__device__ __forceinline__ void wv_prologue_gathers(const WvProb& W, WvTriList& T, v4d (&f4)[8]) {
  if constexpr (R < 8) {
    unsigned (&o)[4] = T.o[R * (R + 1) / 2 + D];
    if constexpr (R == 0 && D == 0) {   // the lists have arrived -- said once, in front of the first gather
#pragma unroll
      for (int t = 0; t < 36; t++) asm volatile("" : "+v"(T.o[t][0]), "+v"(T.o[t][1]), "+v"(T.o[t][2]), "+v"(T.o[t][3]));
    }
    if constexpr (D == 4) f4[R] = wv_gather_vgpr(W, o);
    else wv_gather_agpr<wv_phys(R, D)>(W.Hc, o[0], o[1], o[2], o[3]);
    if constexpr (D < R) wv_prologue_gathers<R, D + 1>(W, T, f4);
    else wv_prologue_gathers<R + 1, 0>(W, T, f4);
  }
}
template <int PH>
__device__ __forceinline__ void wv_prologue_store(lds_double* lds, int lane, v4d& f4, WvB2& bd) {
  if constexpr (PH >= 4) {
    asm volatile("" : "+v"(f4));   // (VGPR operands where they are consumed: see S.bnext in wv_step_rest)
    wv_lds_store(lds + WV_L_WIN + 256 * (wv_phys(PH, 4) - WV_AGPR_TILES), lane, f4);
  }
  asm volatile("" : "+v"(bd.n0), "+v"(bd.n1));
  wv_bq_store(lds + WV_L_BORD + 128 * PH, lane, bd);
}

// ---- the factorisation of one problem by one wavefront (+ the deferred back substitution of the wave's previous problem) ----------------------
// lds: WV_LDS_DOUBLES doubles of this wave.  Returns the "all pivots positive" flag and, per lane c < 6, the camera solution x_cam[c];
// L = block columns [Yb (1 KB, QL) | Y_1 .. Y_8 (slots of 2 KB)] in P.Lb + W tiles in P.Linv.  When it returns, Q's back substitution is complete (Q.x written).
__device__ __forceinline__ int wv_factor(const SftDev& P, double lambda, double lam_corner, lds_double* lds, const WvPrev& Q, double& xcam_out) {
  asm volatile("" ::: "a0", "a255");   // the accumulator file is ours (the kernel descriptor allocates all of it)
  // The lane index is formed HERE, by a statement the compiler can neither hoist nor merge with another one: derived from threadIdx.x, the
  // lane-dependent offsets of this function are invariants of the persistent kernel's problem loop, and under register pressure the compiler
  // parked some of them in accumulator registers across the factorisations -- on a window tile (tools/wave_audit.py rule 1 found three builds
  // of r06 that did; the values came back as whatever the window had left there).
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  const int g = lane >> 4, c = lane & 15;
  WvProb W;
  W.Dn = uni(P.Dn);
  W.Dnp = ((W.Dn + NB - 1) / NB) * NB;
  W.nT = W.Dnp / TS;
  W.q8 = min(3, max(0, (TS * BT - uni(P.kd)) / 4));
  W.lambda = lambda;
  W.lam_corner = lam_corner;
  W.bofs = (unsigned)((c & 3) * W.Dnp + 4 * (c >> 2) + g);
  W.Hc = uni(P.Hc); W.hgl = uni(P.hgatherT); W.Hbord = uni(P.Hbord); W.Lg = uni(P.Lb); W.Linv = uni(P.Linv);
  lds_double* wscr = lds + WV_L_WSCR;
  lds_double* Cn = lds + WV_L_CN;
#ifdef DSH_LAB
  const long long wv_t0 = clock64();
#endif
  if (Q.active) wv_bs_begin(Q, lds, lane);
  WvState S;
#pragma unroll
  for (int i = 0; i < 9; i++) S.Y[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  const int cr = 4 * (c >> 3) + g, cc = 4 * ((c >> 2) & 1) + (c & 3);   // the lane's element of the corner
  S.corner = 0.0;
  if (cr < SFT_BORDER && cc < SFT_BORDER && cc <= cr) S.corner = P.Hcorner[cr * 7 + cc] + ((cr == cc && cr < 6) ? lam_corner : 0.0);
  S.ok = 1;
  {
    WvTriList T;
    v4d f4[8];
    WvB2 bd[8];
    wv_prologue_lists<0, 0>(W, lane, T);
    wv_prologue_gathers<0, 0>(W, T, f4);
#pragma unroll
    for (int r = 0; r < 8; r++) bd[r] = wv_border_fresh(W, r, lane);
    wv_list(W, 8, 8, lane, S.al);
    wv_prologue_store<0>(lds, lane, f4[0], bd[0]); wv_prologue_store<1>(lds, lane, f4[1], bd[1]); wv_prologue_store<2>(lds, lane, f4[2], bd[2]);
    wv_prologue_store<3>(lds, lane, f4[3], bd[3]); wv_prologue_store<4>(lds, lane, f4[4], bd[4]); wv_prologue_store<5>(lds, lane, f4[5], bd[5]);
    wv_prologue_store<6>(lds, lane, f4[6], bd[6]); wv_prologue_store<7>(lds, lane, f4[7], bd[7]);
  }
  wv_wait_vm();
  asm volatile("" : "+v"(S.al[0]), "+v"(S.al[1]), "+v"(S.al[2]), "+v"(S.al[3]));   // (consumed here as far as the compiler is concerned: see the step head)
#ifdef DSH_LAB
  const long long wv_t1 = clock64();
  const long long wv_w1 = wall_clock64();   // 100 MHz, constant: the shader clock the loop actually ran at = cycles / wall time
#endif

  WV_T_DECL;
#pragma unroll 1
  for (int k = 0; k < W.nT; k++) {
    const int ph = k & 7;
    // Requests of this step, issued in front of the tile Cholesky (2 us of vector-ALU work: they arrive behind it) and consumed behind the
    // TRSM -- nothing of them is alive during the trailing update, where the 9 Y tiles and the pipelined LDS tiles need the registers.
    // the lanes' byte offsets inside a 2 KB tile slot of L / inside the compact border tile: computed every step by statements the compiler
    // cannot hoist -- as loop invariants such registers were parked in the accumulator file, on a window tile (tools/wave_audit.py)
    asm volatile("v_lshlrev_b32 %0, 5, %1" : "=v"(S.off_lane) : "v"(lane));
    asm volatile("v_lshlrev_b32 %0, 4, %1" : "=v"(S.off_bord) : "v"(lane));
    S.araw = wv_gather_vgpr(W, S.al);            // tile (k+8, k)^T through the list that came a step ahead
    // The next list right behind it, in FRONT of the other requests: the wait that consumes them (behind the TRSM) covers it as well, so the
    // compiler's wait in front of the gather above is s_waitcnt vmcnt(14) instead of vmcnt(0) (the list was the youngest load crossing the back
    // edge).  Measured: no change of the head section (1.32 k cycles either way) -- what is in flight there has long landed.
    wv_list(W, k + 9, 8, lane, S.al);
    wv_row_list(W, k + 8, lane, S.rl);
    S.bnext = wv_border_fresh(W, k + 8, lane);
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
    WV_T(0);
    // The deferred back substitution's column sits between this step's requests and the tile Cholesky (it used to follow the update of rows
    // 1-7): its data was requested a step ago -- right here, behind the previous column -- and waited for in the middle of that step
    // (wv_wait_vm in wv_step_rest), so the whole update lies between request and use, and the requests of this step travel while it
    // computes.  Measured (A/B on one box): FACTOR 298.7 -> 297.0 ms; in front of the requests instead: 306.6.
    if (Q.active) {
      const int J = Q.nT - 1 - k;
      if (J >= 0) wv_bs_step(Q, J, lds, lane);
    }
    WV_T(7);
    v4d w;
    if (!wv_chol_inv(d, w)) {
#ifdef DSH_LAB
      if (S.ok && lane == 0) P.dbg[3] = 1000.0 + k;   // first column whose diagonal tile had a non-positive pivot
#endif
      S.ok = 0;
    }
    WV_T(1);
    {   // W of column k leaves like the tiles of L: scalar base + the lane's 32-bit offset
      const SFT_G double* slot = W.Linv + (size_t)k * 256;
      const v2d_w w01 = (v2d_w){w[0], w[1]}, w23 = (v2d_w){w[2], w[3]};
      asm volatile("global_store_dwordx4 %0, %1, %3\n\tglobal_store_dwordx4 %0, %2, %3 offset:16" :: "v"(S.off_lane), "v"(w01), "v"(w23), "s"(slot) : "memory");
    }
    // (W^T in accumulator order -- through LDS -- is formed inside wv_step_rest, behind the requests of the TRSM's LDS-resident operands)
    switch (ph + ((k + 8 >= W.nT) ? 8 : 0)) {
      case 0: wv_step_rest<0, false>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 1: wv_step_rest<1, false>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 2: wv_step_rest<2, false>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 3: wv_step_rest<3, false>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 4: wv_step_rest<4, false>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 5: wv_step_rest<5, false>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 6: wv_step_rest<6, false>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 7: wv_step_rest<7, false>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 8: wv_step_rest<0, true>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 9: wv_step_rest<1, true>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 10: wv_step_rest<2, true>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 11: wv_step_rest<3, true>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 12: wv_step_rest<4, true>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 13: wv_step_rest<5, true>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      case 14: wv_step_rest<6, true>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
      default: wv_step_rest<7, true>(W, S, w, k, lds, lane, Q WV_T_PASS); break;
    }
  }
  WV_T_DUMP(P);
  // columns of the previous problem's back substitution that are left (it had more tile rows than this one)
  if (Q.active) {
    for (int J = Q.nT - 1 - W.nT; J >= 0; J--) { wv_wait_vm(); wv_bs_step(Q, J, lds, lane); }
  }
#ifdef DSH_LAB
  const long long wv_t2 = clock64();
  const long long wv_w2 = wall_clock64();
#endif
  // ---- camera corner: 6 x 6 Cholesky of the Schur complement, forward solve of its right-hand side, x_cam (one lane; 7 x 7 in LDS)
  wv_mfma_fence();
  {   // (the lane's corner element from a lane index formed afresh: hoisted to the top of the kernel as a loop invariant, this address was parked in an
      // accumulator register -- on a window tile -- across the factorisation; tools/wave_audit.py rule 1)
    int l2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
    const int g2 = l2 >> 4, c2 = l2 & 15;
    const int cr2 = 4 * (c2 >> 3) + g2, cc2 = 4 * ((c2 >> 2) & 1) + (c2 & 3);
    if (cr2 < SFT_BORDER && cc2 < SFT_BORDER) Cn[cr2 * 7 + cc2] = S.corner;
  }
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
    if (!okc) S.ok = 0;
    if (c < 6) xcam = Cn[49 + c];
  }
  xcam_out = xcam;
#ifdef DSH_LAB
  if (lane == 0) { P.dbg[5] = (double)(wv_t1 - wv_t0); P.dbg[6] = (double)(wv_t2 - wv_t1); P.dbg[4] = (double)(wv_w2 - wv_w1); }
#endif
  return __builtin_amdgcn_readfirstlane(S.ok);
}

// what a finished factorisation hands to its (deferred or immediate) back substitution; like g2o, x keeps its previous content when the
// factorisation failed (ok == 0: nothing is written)
__device__ __forceinline__ WvPrev wv_prev_of(const SftDev& P, int ok, double xcam, int lane_) {
  (void)lane_;
  int lane;   // (formed afresh: see wv_factor)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  const int c = lane & 15;
  WvPrev Q;
  const int Dn = uni(P.Dn), Dnp = ((Dn + NB - 1) / NB) * NB;
  Q.Lg = uni(P.Lb); Q.Linv = uni(P.Linv); Q.x = uni(P.x);
  Q.nT = Dnp / TS;
  Q.active = ok;
  Q.xb = (c < 6) ? xcam : ((c == 6) ? -1.0 : 0.0);
  if (ok && lane < 6) P.x[Dnp + lane] = xcam;
  return Q;
}

// factorisation + back substitution of one problem right away (lab A/B kernel)
__device__ __forceinline__ int wv_factor_solve(const SftDev& P, double lambda, double lam_corner, lds_double* lds) {
  WvPrev none;
  none.Lg = nullptr; none.Linv = nullptr; none.x = nullptr; none.nT = 0; none.active = 0; none.xb = 0.0;
  double xcam;
  const int ok = wv_factor(P, lambda, lam_corner, lds, none, xcam);
  int lane;   // (formed behind the factorisation: see wv_factor)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  const WvPrev Q = wv_prev_of(P, ok, xcam, lane);
#ifdef DSH_LAB
  const long long t2 = clock64();
#endif
  if (ok) wv_backsub_now(Q, lane);
#ifdef DSH_LAB
  if (lane == 0) P.dbg[7] = (double)(clock64() - t2);
#endif
  return ok;
}
