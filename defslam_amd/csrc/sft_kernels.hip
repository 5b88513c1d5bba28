// Shape-from-Template Levenberg-Marquardt solve on gfx950 (MI355X), FP64.
//
// One workgroup (8 wavefronts = 512 threads for latency, or 4 wavefronts so that two problems share a CU) owns one problem and runs the
// whole optimisation of defSLAM::Optimizer::DefPoseOptimization on the device:
//
//   residuals + Jacobians ... sft_types.h:102-133,137-206 (EdgeNodesCamera), :257-311
//                             (EdgeMeanCurvature), :351-377 (EdgesStreching), :401-408 (EdgesReference)
//   robust weights .......... robust_kernel_impl.cpp:78-91, base_edge.h:96-102
//   normal equations ........ base_multi_edge.hpp:171-222, base_binary_edge.hpp:57-131,
//                             base_unary_edge.hpp:43-72, block_solver.hpp:502-560
//   damping / accept-reject . optimization_algorithm_levenberg.cpp:61-189
//   linear solve ............ replaces block_solver.hpp:356-365 + linear_solver_dense.h:65-113
//                             (dense Eigen::LDLT of the full (6+3n)^2 matrix) by a banded Cholesky
//                             of the node block with the camera as a dense border (arrowhead)
//   state update ............ sparse_optimizer.cpp:477-491, se3quat.h:223-257, types_sba.h:52-56
//
// Every sum is evaluated in a fixed order (no atomics), so a run is bit-reproducible.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <math.h>
#include <float.h>
#define SFT_KERNEL_SOURCE
#include "sft_problem.h"
#include "tile_chol.h"

#define NB 32  // panel width of the blocked band Cholesky

// Phase timers (thread 0, constant 100 MHz counter): dbg[1] residuals, [2] assembly, [3] H->L copy,
// [4] panel factorisation, [5] trailing update, [6] back substitution, [7] update + control
#if defined(SFT_PHASE_TIMERS) && defined(DSH_LAB)   // lab builds only: make lab EXTRA=-DSFT_PHASE_TIMERS
#define PH_T0() long long ph_t0__ = wall_clock64()
#define PH_ADD(slot)                                                          \
  do {                                                                        \
    const long long t1__ = wall_clock64();                                    \
    if (threadIdx.x == 0) P.dbg[slot] += (double)(t1__ - ph_t0__);            \
    ph_t0__ = t1__;                                                           \
  } while (0)
#define PH_RESET() ph_t0__ = wall_clock64()
// sections of the assembly (shader clock, wave 0): dbg[32] corner reduction, [33] diagonal gather, [34] butterfly + diagonal finish,
// [35] off-diagonal blocks, [36] round overhead (header prefetch, loop), [37] rounds
#define AS_T0() long long as_t0__ = clock64()
#define AS_ADD(slot) do { const long long t1__ = clock64(); if (threadIdx.x == 0) P.dbg[slot] += (double)(t1__ - as_t0__); as_t0__ = t1__; } while (0)
#else
#define AS_T0() do {} while (0)
#define AS_ADD(slot) do {} while (0)
#define PH_T0() do {} while (0)
#define PH_ADD(slot) do {} while (0)
#define PH_RESET() do {} while (0)
#endif

// Per-wave time stamps of one factorisation step (shader clock), for tuning: -DSFT_STEP_TRACE
#if defined(SFT_STEP_TRACE) && defined(DSH_LAB)
#define ST_MARK(ev) do { if (st_on && k == 40 && (threadIdx.x & 63) == 0) st_buf[(threadIdx.x >> 6) * 8 + (ev)] = (double)clock64(); } while (0)
#define ST_DONE() do {} while (0)
#define ST_BEGIN() lds_double* st_buf = to_lds(ws) + 5376; const bool st_on = P.dbg[8] < 0.5   /* LDS scratch: stamps must not add global traffic */
#define ST_END() do { __syncthreads(); if (st_on && threadIdx.x < 64) P.dbg[16 + threadIdx.x] = st_buf[threadIdx.x]; if (threadIdx.x == 0) P.dbg[8] = 1.0; } while (0)
#else
#define ST_MARK(ev) do {} while (0)
#define ST_DONE() do {} while (0)
#define ST_BEGIN() do {} while (0)
#define ST_END() do {} while (0)
#endif

#define TS 16   // MFMA tile edge (v_mfma_f64_16x16x4_f64)
#define TP 17   // padded leading dimension of a k-major tile in LDS
#define BT 8    // sub-diagonal tiles per block column in tile mode (half-bandwidth <= 16*BT)
#define TILE_LDS (TS * TP)


namespace {

// Offset of tile (I, I-d) in the tile-band storage: (BT+1) row-major 16x16 tiles per tile row.
__device__ __forceinline__ size_t tile_off(int I, int d) { return ((size_t)I * (BT + 1) + d) * (TS * TS); }

// Address of H(r, c), c <= r, in either storage mode.
// Inside a tile the element (row, col) sits at ((row & 3) * 16 + col) * 4 + (row >> 2): the four doubles a lane holds in the
// MFMA accumulator layout (rows (lane>>4) + 4q, column lane & 15) are contiguous -> two 16-byte accesses per lane and tile.
__device__ __forceinline__ int tile_elem(int row, int col) { return (((row & 3) << 4) + col) * 4 + (row >> 2); }

template <class PT>
__device__ __forceinline__ size_t h_index(const PT& P, int r, int c) {
  if (P.tile_mode == 1) return tile_off(r >> 4, (r >> 4) - (c >> 4)) + tile_elem(r & 15, c & 15);
  if (P.tile_mode == 2) return ((size_t)(r >> 4) * P.tpr + ((r >> 4) - (c >> 4))) * (TS * TS) + tile_elem(c & 15, r & 15);   // wide mode: tiles hold H(I,J)^T
  return (size_t)r * P.ldh + (c - r + P.kd);
}
// Two-sided factorisation (SftPart in sft_problem.h): where element (r, c), c <= r, of the natural band ordering lives -- part 0 keeps
// its own rows, the separator rows and the separator block; part 1 is stored in reversed order behind its identity padding, its
// separator rows (reversed as well) behind its own.  (rr, cc), cc <= rr: the element's position in that part's band matrix.
struct SplitMap {
  int on, c0, s, n1p, pad, Dn, tpr0, tpr1;
  SFT_G double *H0, *H1;
};
template <class PT>
__device__ __forceinline__ SplitMap split_map(const PT& P) {
  return SplitMap{P.split, P.sp_c0, P.sp_s, P.sp_n1p, P.sp_pad, P.Dn, P.part[0].tpr, P.part[1].tpr, P.part[0].Hb, P.part[1].Hb};
}
__device__ __forceinline__ void split_target(const SplitMap& S, int r, int c, SFT_G double*& H, int& tpr, int& rr, int& cc) {
  if (r < S.c0 + S.s) { H = S.H0; tpr = S.tpr0; rr = r; cc = c; return; }
  const int rp = S.pad + (S.Dn - 1 - r);
  H = S.H1; tpr = S.tpr1; cc = rp;
  rr = (c >= S.c0 + S.s) ? S.pad + (S.Dn - 1 - c) : S.n1p + (S.s - 1 - (c - S.c0));
}
__device__ __forceinline__ size_t wide_elem(int tpr, int rr, int cc) {   // wide tile layout: tile (I,J) holds H(I,J)^T
  return ((size_t)(rr >> 4) * tpr + ((rr >> 4) - (cc >> 4))) * (TS * TS) + tile_elem(cc & 15, rr & 15);
}

// diagonal element r of H (tile mode 1: from the compact 3x3 blocks)
template <class PT>
__device__ __forceinline__ double h_diag(const PT& P, int r) {
  if (P.tile_mode == 1) { const int a = r / 3, e = r - 3 * a; return P.Hc[9 * (size_t)(a + P.off_ptr[a]) + 4 * e]; }
  if (P.tile_mode == 2 && P.split) {
    SFT_G double* H; int tpr, rr, cc;
    split_target(split_map(P), r, r, H, tpr, rr, cc);
    return H[wide_elem(tpr, rr, cc)];
  }
  return P.Hb[h_index(P, r, r)];
}

struct Ctl {       // LDS-resident control block, written by thread 0
  double R[9], t[3];
  double lambda, ni, chi_cur, chi_tmp, chi_ini, rho, scale;
  int fact_ok, qmax, nbad, stop, accepted, it;
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

__device__ void quat_from_R(const double* R, double* q) {
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}

__device__ __forceinline__ void quat_unit_pos(double* q) {
  if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= nrm; q[1] /= nrm; q[2] /= nrm; q[3] /= nrm;
}

// pose <- exp(delta) * pose, delta = [omega, upsilon] (types_six_dof_expmap.h:73-76)
__device__ void pose_oplus(double* pose, const double* d) {
  const double om0 = d[0], om1 = d[1], om2 = d[2];
  const double theta = sqrt(om0 * om0 + om1 * om1 + om2 * om2);
  const double Om[9] = {0, -om2, om1, om2, 0, -om0, -om1, om0, 0};
  double Om2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Om[i * 3 + k] * Om[k * 3 + j];
      Om2[i * 3 + j] = s;
    }
  double Rd[9], V[9];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { Rd[i] = ((i % 4 == 0 ? 1.0 : 0.0) + Om[i]) + Om2[i]; V[i] = Rd[i]; }
  } else {
    const double a = sin(theta) / theta;
    const double b = (1 - cos(theta)) / (theta * theta);
    const double c = (theta - sin(theta)) / (theta * theta * theta);
    for (int i = 0; i < 9; i++) {
      const double id = (i % 4 == 0) ? 1.0 : 0.0;
      Rd[i] = (id + a * Om[i]) + b * Om2[i];
      V[i] = (id + b * Om[i]) + c * Om2[i];
    }
  }
  double qd[4], td[3];
  quat_from_R(Rd, qd);
  quat_unit_pos(qd);
  for (int i = 0; i < 3; i++) td[i] = V[i * 3] * d[3] + V[i * 3 + 1] * d[4] + V[i * 3 + 2] * d[5];
  // t <- td + qd (x) t
  const double* q = pose + 3;
  const double v0 = pose[0], v1 = pose[1], v2 = pose[2];
  double uv0 = qd[1] * v2 - qd[2] * v1, uv1 = qd[2] * v0 - qd[0] * v2, uv2 = qd[0] * v1 - qd[1] * v0;
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  const double c0 = qd[1] * uv2 - qd[2] * uv1, c1 = qd[2] * uv0 - qd[0] * uv2, c2 = qd[0] * uv1 - qd[1] * uv0;
  const double nt0 = td[0] + (v0 + qd[3] * uv0 + c0);
  const double nt1 = td[1] + (v1 + qd[3] * uv1 + c1);
  const double nt2 = td[2] + (v2 + qd[3] * uv2 + c2);
  double nq[4];
  nq[3] = qd[3] * q[3] - qd[0] * q[0] - qd[1] * q[1] - qd[2] * q[2];
  nq[0] = qd[3] * q[0] + qd[0] * q[3] + qd[1] * q[2] - qd[2] * q[1];
  nq[1] = qd[3] * q[1] + qd[1] * q[3] + qd[2] * q[0] - qd[0] * q[2];
  nq[2] = qd[3] * q[2] + qd[2] * q[3] + qd[0] * q[1] - qd[1] * q[0];
  quat_unit_pos(nq);
  pose[0] = nt0; pose[1] = nt1; pose[2] = nt2;
  pose[3] = nq[0]; pose[4] = nq[1]; pose[5] = nq[2]; pose[6] = nq[3];
}

// Deterministic block-wide sum of NV values per thread. red: LDS scratch of 16*NV doubles.
// Result valid in out[0..NV) (LDS) after the call for every thread.
template <int NV>
__device__ void block_sum(double* v, double* red, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double s = v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave * NV + i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += red[w * NV + threadIdx.x];
    out[threadIdx.x] = s;
  }
  __syncthreads();
}

__device__ double block_max(double v, double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double m = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); w++) m = fmax(m, red[w]);
  __syncthreads();
  return m;
}

__device__ __forceinline__ void huber(const SftDev& P, double e2, double& rho0, double& rho1) {
  if (e2 <= P.hub_dsqr) { rho0 = e2; rho1 = 1.0; }
  else { const double sq = sqrt(e2); rho0 = 2 * sq * P.hub_delta - P.hub_dsqr; rho1 = P.hub_delta / sq; }
}

// ------------------------------------------------------------------------------------------
// Residuals (+ assembly records when WANT_J): returns the robust chi2 in ctl-independent LDS out[0]
// ------------------------------------------------------------------------------------------
// Solver workspace and assembly records share the LDS panel.  Explicit address space: see to_lds below.
using lds_double = __attribute__((address_space(3))) double;
__device__ __forceinline__ lds_double* to_lds(double* p) { return (lds_double*)p; }

// Records of one linearisation that the assembly gathers from.  Each array lives in LDS when the host packer found room for it
// (P.lds_flags, in the order of how often a gather touches it) and in the problem's workspace otherwise:
//   wt[m]      rho'_m w_m of observation m (robust weight x information)
//   A[6 a]     the 2x3 matrix A_a of active node a: the reference linearises an observation at every node's OWN camera-frame
//              depth (sft_types.h:176-205), so J_node(m, s) = b_ms A_node -- the Jacobian of an observation with respect to a node
//              depends on the observation only through its barycentric coordinate.  H_ij(obs) = (sum_m wt_m b_mi b_mj) A_i^T A_j.
//   star[4 s]  unit vector u and residual r of curvature star s;  str[4 e]  gradient g and residual of stretch edge e
// Placement class of the records (P.lds_class, chosen by the host packer from the LDS budget of the launch shape):
//   0: everything in the workspace (global memory)   1: observation weights + curvature records in LDS
//   2: all four arrays in LDS
// The class is a template parameter: every access names its address space at compile time.  (A run-time choice per access
// would put a branch and a wait of its own around every load -- the gathers then cannot batch their loads -- and a generic
// pointer compiles to flat_* instructions, which count on the LDS counter AND the memory counter.)
using gdouble = SFT_G double;
using lds_int = __attribute__((address_space(3))) int;
typedef double v2d __attribute__((ext_vector_type(2)));

// The 2x3 matrix A of a node at world position p: -(1/z) [[fx, 0, -x/z fx], [0, fy, -y/z fy]] R at the node's own camera-frame position
// (sft_types.h:176-205).  ONE definition: the residual pass (which stores it where there is room) and the assembly (which recomputes it
// from the LDS copy of the positions where there is not) must produce the same bits.
__device__ __forceinline__ void node_A(const double* R, const double* t, double fx, double fy, double p0, double p1, double p2, double* A) {
  const double xs = (R[0] * p0 + R[1] * p1 + R[2] * p2) + t[0];
  const double ys = (R[3] * p0 + R[4] * p1 + R[5] * p2) + t[1];
  const double zs = (R[6] * p0 + R[7] * p1 + R[8] * p2) + t[2];
  const double sc = -1. / zs;
  const double t00 = sc * fx, t02 = sc * (-xs / zs * fx), t11 = sc * fy, t12 = sc * (-ys / zs * fy);
#pragma unroll
  for (int cc = 0; cc < 3; cc++) {
    A[cc] = (t00 * R[cc] + 0.0 * R[3 + cc]) + t02 * R[6 + cc];
    A[3 + cc] = (0.0 * R[cc] + t11 * R[3 + cc]) + t12 * R[6 + cc];
  }
}
// Stretching record of mesh edge (a, b) with rest length L0: unit gradient / L0 and the residual (sft_types.h:351-377); returns the residual.
__device__ __forceinline__ void stretch_record(double d0, double d1, double d2, double L0, double* o) {
  const double nrm = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
  const double er = nrm * (1.0 / L0) - 1.0;
  const double ddo = 1.0 / (nrm * L0);
  o[0] = d0 * ddo; o[1] = d1 * ddo; o[2] = d2 * ddo; o[3] = er;
}

// Placement class of the records of one linearisation (P.lds_class, chosen by the host from the LDS budget of the launch shape):
//   0: everything in the workspace (global memory)
//   1: node positions (every residual gathers three to seven of them), observation weights and curvature records in LDS
//   2: also the node matrices and the stretching records
//   3: also the camera records, as five doubles per observation (x/z, y/z, 1/z, e0, e1) from which cam_jac rebuilds the ten Jacobian entries --
//      the 128-byte record in the workspace is the last thing the diagonal blocks gathered from memory (rounds of phase kernels: sft_batch.h)
template <int CLS>
struct AsmRec {
  static constexpr bool XYZ_L = CLS >= 1, WT_L = CLS >= 1, STAR_L = CLS >= 1, A_L = CLS >= 2, STR_L = CLS >= 2, CAM_L = CLS >= 3;
  lds_double *xyz_l, *wt_l, *A_l, *star_l, *str_l, *cam_l;
  gdouble *xyz_g, *wt_g, *A_g, *star_g, *str_g;
  __device__ __forceinline__ double xyz(size_t i) const { if constexpr (XYZ_L) return xyz_l[i]; else return xyz_g[i]; }
  __device__ __forceinline__ double wt(int m) const { if constexpr (WT_L) return wt_l[m]; else return wt_g[m]; }
  __device__ __forceinline__ void set_wt(int m, double v) const { if constexpr (WT_L) wt_l[m] = v; else wt_g[m] = v; }
  __device__ __forceinline__ double A(size_t i) const { if constexpr (A_L) return A_l[i]; else return A_g[i]; }
  __device__ __forceinline__ void set_A(size_t i, double v) const { if constexpr (A_L) A_l[i] = v; else A_g[i] = v; }
  // curvature (star = true) or stretch record e: four doubles (direction, residual).  Both candidates are loaded and one is
  // selected: two independent loads, no branch.
  __device__ __forceinline__ void rec4(bool star, size_t e, bool valid, double* o) const {
    double a[4], b[4];
    const size_t ea = (valid && star) ? e : 0, eb = (valid && !star) ? e : 0;
    if constexpr (STAR_L) { const lds_double* r = star_l + 4 * ea; a[0] = r[0]; a[1] = r[1]; a[2] = r[2]; a[3] = r[3]; }
    else { const gdouble* r = star_g + 4 * ea; a[0] = r[0]; a[1] = r[1]; a[2] = r[2]; a[3] = r[3]; }
    if constexpr (STR_L) { const lds_double* r = str_l + 4 * eb; b[0] = r[0]; b[1] = r[1]; b[2] = r[2]; b[3] = r[3]; }
    else { const gdouble* r = str_g + 4 * eb; b[0] = r[0]; b[1] = r[1]; b[2] = r[2]; b[3] = r[3]; }
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = star ? a[k] : b[k];
  }
  __device__ __forceinline__ void set_star(size_t e, double a, double b, double c, double d) const {
    if constexpr (STAR_L) { lds_double* r = star_l + 4 * e; r[0] = a; r[1] = b; r[2] = c; r[3] = d; }
    else { gdouble* r = star_g + 4 * e; r[0] = a; r[1] = b; r[2] = c; r[3] = d; }
  }
  __device__ __forceinline__ void set_str(size_t e, double a, double b, double c, double d) const {
    if constexpr (STR_L) { lds_double* r = str_l + 4 * e; r[0] = a; r[1] = b; r[2] = c; r[3] = d; }
    else { gdouble* r = str_g + 4 * e; r[0] = a; r[1] = b; r[2] = c; r[3] = d; }
  }
};

template <int NW, int CLS>
__device__ __forceinline__ AsmRec<CLS> asm_records(const SftDev& P, double* lds) {
  AsmRec<CLS> r;
  lds_double* base = to_lds(lds);
  size_t off = 0;
  r.xyz_l = base + off; if (AsmRec<CLS>::XYZ_L) off += ((size_t)3 * P.n + 1) & ~(size_t)1;
  r.wt_l = base + off; if (AsmRec<CLS>::WT_L) off += (size_t)(P.M + 1) & ~(size_t)1;
  r.star_l = base + off; if (AsmRec<CLS>::STAR_L) off += 4 * (size_t)P.S;
  r.A_l = base + off; if (AsmRec<CLS>::A_L) off += 6 * (size_t)P.nA;
  r.str_l = base + off; if (AsmRec<CLS>::STR_L) off += 4 * (size_t)P.Es;
  r.cam_l = base + off; if (AsmRec<CLS>::CAM_L) off += 5 * (size_t)P.M;
  r.xyz_g = P.xyz; r.wt_g = P.wtv; r.A_g = P.Anode; r.star_g = P.Jstar; r.str_g = P.Jstr;
  return r;
}

// The ten non-zero entries of the camera Jacobian of an observation (sft_types.h:162-174, columns [omega, upsilon]; row 0: columns 0, 1, 2,
// 3, 5, row 1: columns 0, 1, 2, 4, 5) from xz = x/z, yz = y/z, iz = 1/z -- placement class 3 keeps these three per observation in LDS and both
// the residual pass (camera corner) and the assembly (camera x node blocks) call this ONE function, so they work with the same bits.
__device__ __forceinline__ void cam_jac(double xz, double yz, double iz, double fx, double fy, double* j0, double* j1) {
  const double xy = xz * yz;
  j0[0] = xy * fx; j0[1] = -(1.0 + xz * xz) * fx; j0[2] = yz * fx; j0[3] = -iz * fx; j0[4] = (xz * iz) * fx;
  j1[0] = (1.0 + yz * yz) * fy; j1[1] = -xy * fy; j1[2] = -xz * fy; j1[3] = -iz * fy; j1[4] = (yz * iz) * fy;
}

// (force-inlined like assemble: inside the kernel `P` is a kernel-argument-derived reference whose fields are scalar loads; as a
// separate function it would arrive as a generic pointer in vector registers and every P.field would be a flat vector load)
template <bool WANT_J, int CLS>
__device__ __forceinline__ double eval_edges(const SftDev& P, Ctl* ctl, double* red, double* out, const AsmRec<CLS>& ar, const SFT_G double* step = nullptr) {
  if (threadIdx.x == 0) {
    quat_to_R(P.pose + 3, ctl->R);
    ctl->t[0] = P.pose[0]; ctl->t[1] = P.pose[1]; ctl->t[2] = P.pose[2];
  }
  if constexpr (AsmRec<CLS>::XYZ_L) {   // the node positions once, coalesced; every gather below is an LDS read
    const auto src = P.xyz;
    if (step) {   // a damping trial of the phase rounds: the trial state xyz + x exists in LDS only (sftb_trial_kernel writes it back if accepted)
      for (int i = threadIdx.x; i < 3 * P.n; i += blockDim.x) {
        const double v = src[i];
        const int a = P.act[i / 3];
        ar.xyz_l[i] = a >= 0 ? v + step[3 * a + (i % 3)] : v;
      }
    } else {
      for (int i = threadIdx.x; i < 3 * P.n; i += blockDim.x) ar.xyz_l[i] = src[i];
    }
  }
  __syncthreads();
  double R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; i++) R[i] = ctl->R[i];
  t[0] = ctl->t[0]; t[1] = ctl->t[1]; t[2] = ctl->t[2];
  double chi = 0.0;
  // camera corner of the normal equations, H_cc (lower 21) and b_c (6): accumulated while the camera Jacobian of an observation is
  // in registers, reduced over the block in a fixed tree below (the assembly used to re-read every record for it)
  double cc_acc[WANT_J ? 27 : 1];
#pragma unroll
  for (int i = 0; i < (WANT_J ? 27 : 1); i++) cc_acc[i] = 0.0;
  const int total = P.M + P.n + P.S + P.Es;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    if (idx < P.M) {
      const int m = idx;
      const int n0 = P.obs_nodes[3 * m], n1 = P.obs_nodes[3 * m + 1], n2 = P.obs_nodes[3 * m + 2];
      const double b0 = P.obs_bary[3 * m], b1 = P.obs_bary[3 * m + 1], b2 = P.obs_bary[3 * m + 2];
      double p[3][3];
#pragma unroll
      for (int k = 0; k < 3; k++) { p[0][k] = ar.xyz(3 * n0 + k); p[1][k] = ar.xyz(3 * n1 + k); p[2][k] = ar.xyz(3 * n2 + k); }
      double pw[3], pc[3];
#pragma unroll
      for (int k = 0; k < 3; k++) pw[k] = (b0 * p[0][k] + b1 * p[1][k]) + b2 * p[2][k];
#pragma unroll
      for (int k = 0; k < 3; k++) pc[k] = (R[3 * k] * pw[0] + R[3 * k + 1] * pw[1] + R[3 * k + 2] * pw[2]) + t[k];
      const double e0 = P.obs_uv[2 * m] - ((pc[0] / pc[2]) * P.fx + P.cx);
      const double e1 = P.obs_uv[2 * m + 1] - ((pc[1] / pc[2]) * P.fy + P.cy);
      const double w = P.obs_w[m];
      const double c2 = e0 * (w * e0) + e1 * (w * e1);
      double rho0, rho1;
      huber(P, c2, rho0, rho1);
      chi += rho0;
      P.chi2_obs[m] = c2;
      if (WANT_J) {
        // camera record: robust information, error, the ten non-zero entries of J_cam (sft_types.h:162-174, columns [omega, upsilon]);
        // the point is the barycentric combination of the nodes' camera-frame positions, like the reference
        double c[3][3];
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
          for (int k = 0; k < 3; k++) c[s][k] = (R[3 * k] * p[s][0] + R[3 * k + 1] * p[s][1] + R[3 * k + 2] * p[s][2]) + t[k];
        const double x = (c[0][0] * b0 + c[1][0] * b1) + c[2][0] * b2;
        const double y = (c[0][1] * b0 + c[1][1] * b1) + c[2][1] * b2;
        const double z = (c[0][2] * b0 + c[1][2] * b1) + c[2][2] * b2;
        const double z2 = z * z, fx = P.fx, fy = P.fy;
        const double wt = rho1 * w;
        // row 0: columns 0, 1, 2, 3, 5 (column 4 is zero); row 1: columns 0, 1, 2, 4, 5 (column 3 is zero)
        double j0[6], j1[6];
        if constexpr (AsmRec<CLS>::CAM_L) {
          const double iz = 1.0 / z, xz = x * iz, yz = y * iz;
          double a0[5], a1[5];
          cam_jac(xz, yz, iz, fx, fy, a0, a1);
          j0[0] = a0[0]; j0[1] = a0[1]; j0[2] = a0[2]; j0[3] = a0[3]; j0[4] = 0.0; j0[5] = a0[4];
          j1[0] = a1[0]; j1[1] = a1[1]; j1[2] = a1[2]; j1[3] = 0.0; j1[4] = a1[3]; j1[5] = a1[4];
          lds_double* cr = ar.cam_l + 5 * (size_t)m;
          cr[0] = xz; cr[1] = yz; cr[2] = iz; cr[3] = e0; cr[4] = e1;
        } else {
          j0[0] = x * y / z2 * fx; j0[1] = -(1 + (x * x / z2)) * fx; j0[2] = y / z * fx; j0[3] = -1. / z * fx; j0[4] = 0.0; j0[5] = x / z2 * fx;
          j1[0] = (1 + y * y / z2) * fy; j1[1] = -x * y / z2 * fy; j1[2] = -x / z * fy; j1[3] = 0.0; j1[4] = -1. / z * fy; j1[5] = y / z2 * fy;
          // the record is one 128-byte line (SFT_CAM_STRIDE = 16): eight 16-byte stores here, one line per gather in the assembly
          const auto rec = reinterpret_cast<SFT_G v2d*>(P.camrec + (size_t)m * SFT_CAM_STRIDE);
          rec[0] = (v2d){wt, e0}; rec[1] = (v2d){e1, j0[0]}; rec[2] = (v2d){j0[1], j0[2]}; rec[3] = (v2d){j0[3], j0[5]};
          rec[4] = (v2d){j1[0], j1[1]}; rec[5] = (v2d){j1[2], j1[4]}; rec[6] = (v2d){j1[5], c2}; rec[7] = (v2d){0.0, 0.0};
        }
        ar.set_wt(m, wt);
        if constexpr (WANT_J) {
          int q = 0;
#pragma unroll
          for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c <= r; c++) cc_acc[q++] += wt * (j0[r] * j0[c] + j1[r] * j1[c]);
#pragma unroll
          for (int r = 0; r < 6; r++) cc_acc[21 + r] -= wt * (j0[r] * e0 + j1[r] * e1);
        }
      }
    } else if (idx < P.M + P.n) {
      const int nd = idx - P.M;
      const int a = P.act[nd];
      if (a >= 0) {
        const double p0 = ar.xyz(3 * nd), p1 = ar.xyz(3 * nd + 1), p2 = ar.xyz(3 * nd + 2);
        if (P.viewed[a]) {   // reference (temporal) edge, EdgesReference: e = v - v_ref (sft_types.h:401-408)
          const double e0 = p0 - P.xyz0[3 * nd], e1 = p1 - P.xyz0[3 * nd + 1], e2 = p2 - P.xyz0[3 * nd + 2];
          chi += (e0 * (P.w_ref * e0) + e1 * (P.w_ref * e1)) + e2 * (P.w_ref * e2);
        }
        if constexpr (WANT_J) {
          double A6[6];
          node_A(R, t, P.fx, P.fy, p0, p1, p2, A6);
#pragma unroll
          for (int k = 0; k < 6; k++) ar.set_A(6 * (size_t)a + k, A6[k]);
        }
      }
    } else if (idx < P.M + P.n + P.S) {
      const int s = idx - P.M - P.n;
      const int nd = P.star_node[s];
      double a0 = 0, a1 = 0, a2 = 0;
      {   // weighted mean of the 1-ring, neighbours in list order; the loads of up to eight neighbours travel together (one round trip
          // for the indices and weights, one for the positions) instead of two dependent round trips per neighbour
        constexpr int NCH = 8;
        const int q0 = P.nbr_ptr[nd], q1 = P.nbr_ptr[nd + 1];
        for (int qb = q0; qb < q1; qb += NCH) {
          int jj[NCH];
          double ww[NCH], xx[NCH][3];
#pragma unroll
          for (int i = 0; i < NCH; i++) {
            const bool in = qb + i < q1;
            jj[i] = in ? P.nbr_idx[qb + i] : nd;
            ww[i] = in ? P.nbr_w[qb + i] : 0.0;
          }
#pragma unroll
          for (int i = 0; i < NCH; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) xx[i][k] = ar.xyz(3 * jj[i] + k);
#pragma unroll
          for (int i = 0; i < NCH; i++)
            if (qb + i < q1) { a0 = a0 + ww[i] * xx[i][0]; a1 = a1 + ww[i] * xx[i][1]; a2 = a2 + ww[i] * xx[i][2]; }
        }
      }
      const double sw = P.nbr_sumw[nd];
      const double m0 = ar.xyz(3 * nd) - a0 / sw, m1 = ar.xyz(3 * nd + 1) - a1 / sw, m2 = ar.xyz(3 * nd + 2) - a2 / sw;
      const double nrm = sqrt(m0 * m0 + m1 * m1 + m2 * m2);
      const double r = nrm - P.k0[nd];
      chi += (P.w_curv * P.star_sL[s]) * (r * r);
      if (WANT_J) {
        if (nrm < 1E-15) ar.set_star(s, 0.0, 0.0, 0.0, r);
        else ar.set_star(s, m0 / nrm, m1 / nrm, m2 / nrm, r);
      }
    } else {
      const int e = idx - P.M - P.n - P.S;
      const int a = P.str_nodes[2 * e], b = P.str_nodes[2 * e + 1];
      const double d0 = ar.xyz(3 * a) - ar.xyz(3 * b), d1 = ar.xyz(3 * a + 1) - ar.xyz(3 * b + 1), d2 = ar.xyz(3 * a + 2) - ar.xyz(3 * b + 2);
      double sr[4];
      stretch_record(d0, d1, d2, P.str_L0[e], sr);
      const double er = sr[3];
      chi += er * (P.w_str * er);
      if constexpr (WANT_J) ar.set_str(e, sr[0], sr[1], sr[2], sr[3]);
    }
  }
  if constexpr (WANT_J) {
    block_sum<27>(cc_acc, red, out);
    if (threadIdx.x == 0) {
      int q = 0;
      for (int r = 0; r < 6; r++)
        for (int c = 0; c <= r; c++) P.Hcorner[r * 7 + c] = out[q++];
      for (int r = 0; r < 6; r++) P.Hcorner[6 * 7 + r] = out[21 + r];
      P.Hcorner[48] = 0.0;
    }
    __syncthreads();
  }
  block_sum<1>(&chi, red, out);
  return out[0];
}

// ------------------------------------------------------------------------------------------
// Normal equations: one gather per 3x3 block, fixed contribution order (no atomics: a run is bit-reproducible).
//
// A wavefront takes seven block rows (nodes) per round: 8 lanes per diagonal block (observations, curvature, stretching, reference
// edge, the 6x3 camera block and b of that node), one lane per off-diagonal block of those rows.  In tile mode 1 a finished block
// is stored as the nine doubles it consists of into the compact array P.Hc -- the factorisation gathers its 16x16 tiles from
// there, nothing is padded to tiles in memory; the other storage modes (wide tiles, row-major band) store the elements into their
// layout.  No workgroup barrier inside: the wavefronts of a problem drift apart and cover each other's gather latency.
// ------------------------------------------------------------------------------------------

// Sum over the 8 lanes of a group into its first lane with data-parallel-primitive moves (row_shl: lane i reads lane i + n of its
// 16-lane row; no LDS crossbar, no wait): only the lanes that feed lane 0 of a group matter, and they read inside the group.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  // (mov_dpp: the destination of a lane whose source is outside its row stays undefined instead of 0 -- update_dpp(0, ...) costs a v_mov of
  // the zero in front of every move; the lanes that read outside never feed lane 0 of a group, see group8_sum)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group8_sum(double v) {
  v += dpp_mov<0x104>(v);   // row_shl:4
  v += dpp_mov<0x102>(v);   // row_shl:2
  v += dpp_mov<0x101>(v);   // row_shl:1
  return v;
}

template <int NW, int CLS, int PARTS = 3>
__device__ __forceinline__ void assemble(const SftDev& P_, double* red, double* out, const AsmRec<CLS>& ar, int part = 0, int nparts = 1) {
  // The pointers and scalars the gathers use, read once: wave-uniform values stay in scalar registers instead of being re-read
  // from the problem record (a scalar load + a wait that also drains the LDS counter) inside the loops.
  struct {
    int Dn, nA, M, tile_mode, tpr, kd, ldh;
    double w_ref, w_curv, w_str, fx, fy;
    decltype(P_.camrec) camrec; decltype(P_.ob_ptr) ob_ptr, sh_ptr, ob_m, off_ptr, off_rc, tmask, actnode; decltype(P_.ob_c) ob_c, sh_cf, xyz0;
    decltype(P_.sh_rec) sh_rec; decltype(P_.viewed) viewed; decltype(P_.Hb) Hb, Hc, Hbord, Hcorner, xyz, dbg;
  } P{P_.Dn, P_.nA, P_.M, P_.tile_mode, P_.tpr, P_.kd, P_.ldh, P_.w_ref, P_.w_curv, P_.w_str, P_.fx, P_.fy, P_.camrec, P_.ob_ptr, P_.sh_ptr, P_.ob_m, P_.off_ptr,
      P_.off_rc, P_.tmask, P_.actnode, P_.ob_c, P_.sh_cf, P_.xyz0, P_.sh_rec, P_.viewed, P_.Hb, P_.Hc, P_.Hbord, P_.Hcorner, P_.xyz, P_.dbg};
  const int Dnp = ((P.Dn + NB - 1) / NB) * NB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const SplitMap SM = split_map(P_);
  AS_T0();
  AS_ADD(32);
  // Tile mode 1 keeps H as compact 3x3 blocks (P.Hc; the factorisation gathers its tiles from them): a lane that finishes a
  // block stores its nine doubles, nothing is padded to tiles.  The other modes store into their band / tile layout element by element.
  const bool compact = P.tile_mode == 1;
#ifndef SFT_ASM_GN
#define SFT_ASM_GN 7
#endif
  constexpr int GN = SFT_ASM_GN;             // block rows (nodes) per round of a wavefront: 8 lanes each for the diagonal blocks
  const int ngroups = (P.nA + GN - 1) / GN;
  const auto Hg = P.Hb;
  const auto Hc = P.Hc;
  // node range of group I and the headers of a lane's first off-diagonal block: fetched one group ahead
  auto group_nodes = [&](int I, int& a_lo, int& a_hi) {
    a_lo = GN * I;
    a_hi = min(GN * I + GN - 1, P.nA - 1);
  };
  struct Hdr { int q, qe, bi, bj, ob0, ob1, sh0, sh1, dob0, dob1, dsh0, dsh1, doff; };
  auto load_hdr = [&](int I) -> Hdr {
    Hdr h{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (I >= ngroups) return h;
    int a_lo, a_hi;
    group_nodes(I, a_lo, a_hi);
    {   // list bounds of the lane's diagonal block
      const int a = a_lo + (lane >> 3);
      if ((lane >> 3) < GN && a <= a_hi) { h.dob0 = P.ob_ptr[a]; h.dob1 = P.ob_ptr[a + 1]; h.dsh0 = P.sh_ptr[a]; h.dsh1 = P.sh_ptr[a + 1]; h.doff = P.off_ptr[a]; }
    }
    const int qb = P.off_ptr[a_lo];
    h.qe = P.off_ptr[a_hi + 1];
    h.q = qb + lane;
    if (h.q < h.qe) {
      h.bi = P.off_rc[2 * h.q]; h.bj = P.off_rc[2 * h.q + 1];
      const int blk = P.nA + h.q;
      h.ob0 = P.ob_ptr[blk]; h.ob1 = P.ob_ptr[blk + 1]; h.sh0 = P.sh_ptr[blk]; h.sh1 = P.sh_ptr[blk + 1];
    }
    return h;
  };
  // First chunk of the two lists of the lane's diagonal block (8 lanes per node: 2 observations and 3 curvature / stretch contributions per
  // lane).  Fetched ONE ROUND AHEAD as well, from the header that came a round ahead of that: a round then starts with the loads of its
  // records instead of with the loads of the list entries that say which records -- one dependent round trip per round instead of two.
  constexpr int DCH = 2, HCH = 3;   // per lane and round trip: 8 lanes x 2 = 16 observations, 8 x 3 = 24 curvature / stretch contributions
  struct DL { int mm[DCH]; double bb[DCH]; uint32_t rc[HCH]; double c0[HCH], c1[HCH]; };
  auto load_dlists = [&](const Hdr& h) -> DL {
    DL l;
    const int sub = lane & 7;
#pragma unroll
    for (int i = 0; i < DCH; i++) {
      const int p = h.dob0 + sub + 8 * i;
      const bool in = p < h.dob1;
      l.mm[i] = in ? P.ob_m[p] : 0;
      l.bb[i] = in ? P.ob_c[p] : 0.0;      // a zero coefficient switches a padding entry off
    }
#pragma unroll
    for (int i = 0; i < HCH; i++) {
      const int p = h.dsh0 + sub + 8 * i;
      const bool in = p < h.dsh1;
      l.rc[i] = in ? P.sh_rec[p] : 0xFFFFFFFFu;
      l.c0[i] = in ? P.sh_cf[2 * p] : 0.0;
      l.c1[i] = in ? P.sh_cf[2 * p + 1] : 0.0;
    }
    return l;
  };
  // (part, nparts): this workgroup takes every nparts-th round of its wavefronts -- the latency mode splits one assembly over the
  // workgroups of a problem, which all write into the same H (sft_spec_kernel); 0, 1 = everything
  const int I0 = wave + NW * part, dI = NW * nparts;
  Hdr nxt = load_hdr(I0);
  DL nl{};
  if constexpr (PARTS & 1) nl = load_dlists(nxt);

#pragma unroll 1
  for (int I = I0; I < ngroups; I += dI) {
    int a_lo, a_hi;
    group_nodes(I, a_lo, a_hi);
    Hdr cur = nxt;
    const DL cl = nl;
    nxt = load_hdr(I + dI);
    // element (r, c), c <= r, of H in the band / wide-tile layouts (and its mirror inside a diagonal tile, which is stored symmetric)
    auto put = [&](int r, int c, double v) {
      if (P.tile_mode == 2 && SM.on) {   // two-sided factorisation: the element goes into the band matrix of its part
        SFT_G double* H; int tpr, rr, cc;
        split_target(SM, r, c, H, tpr, rr, cc);
        const size_t idx = wide_elem(tpr, rr, cc);
        H[idx] = v;
        if (cc < rr && (rr >> 4) == (cc >> 4)) H[idx + (tile_elem(rr & 15, cc & 15) - tile_elem(cc & 15, rr & 15))] = v;
        return;
      }
      const size_t idx = h_index(P, r, c);
      Hg[idx] = v;
      if (P.tile_mode && c < r && (r >> 4) == (c >> 4)) {
        const int e1 = tile_elem(r & 15, c & 15), e2 = tile_elem(c & 15, r & 15);
        Hg[idx + (P.tile_mode == 2 ? e1 - e2 : e2 - e1)] = v;
      }
    };
    // first chunk of the lists of the lane's (first) off-diagonal block: issued here, with the diagonal lists below, used after the
    // diagonal section -- one memory round trip less per round
    constexpr int OCH = 4, SCH = 6;
    int om0[OCH] = {0, 0, 0, 0};
    double oc0[OCH] = {0.0, 0.0, 0.0, 0.0};
    uint32_t orc0[SCH] = {0, 0, 0, 0, 0, 0};
    double occ0[SCH] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if constexpr (PARTS & 2) {
#pragma unroll
    for (int i = 0; i < OCH; i++) {
      const bool in = cur.ob0 + i < cur.ob1;
      om0[i] = in ? P.ob_m[cur.ob0 + i] : 0;
      oc0[i] = in ? P.ob_c[cur.ob0 + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < SCH; i++) {
      const bool in = cur.sh0 + i < cur.sh1;
      orc0[i] = in ? P.sh_rec[cur.sh0 + i] : 0xFFFFFFFFu;
      occ0[i] = in ? P.sh_cf[2 * (cur.sh0 + i)] : 0.0;
    }
    }
    // ---- diagonal blocks: 8 lanes per node, contributions dealt round-robin, partial sums combined by a fixed xor butterfly.
    // Every level of the gather (list entries -> records) is issued for up to DCH contributions at once.
    if constexpr (PARTS & 1) {
      const int sub = lane & 7, a = a_lo + (lane >> 3);
      const bool on = (lane >> 3) < GN && a <= a_hi;
      double sii = 0.0, G0[5], G1[5], g0 = 0.0, g1 = 0.0, Hs[6], bn[3];
#pragma unroll
      for (int k = 0; k < 5; k++) { G0[k] = 0.0; G1[k] = 0.0; }
#pragma unroll
      for (int k = 0; k < 6; k++) Hs[k] = 0.0;
      bn[0] = bn[1] = bn[2] = 0.0;
      // what the finishing lane of the node needs besides the sums: issued now, used after the gather
      double A[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, eref[3] = {0.0, 0.0, 0.0};
      bool vw = false;
      if (on && sub == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) A[k] = ar.A(6 * (size_t)a + k);
        vw = P.viewed[a] != 0;
        const int nd = P.actnode[a];
#pragma unroll
        for (int c = 0; c < 3; c++) eref[c] = P.xyz[3 * nd + c] - P.xyz0[3 * nd + c];
      }
      if (on) {
        const int ob0 = cur.dob0, ob1 = cur.dob1, sh0 = cur.dsh0, sh1 = cur.dsh1;
        // camera record of observation m as [wt, e0, e1, j0 (5), j1 (5), -]: one 128-byte line of the workspace (seven 16-byte loads), or -- placement
        // class 3 -- five doubles from LDS and the Jacobian rebuilt from them
        auto cam_record = [&](int m, double (&rr)[14]) {
          if constexpr (AsmRec<CLS>::CAM_L) {
            const lds_double* cr = ar.cam_l + 5 * (size_t)m;
            const double xz = cr[0], yz = cr[1], iz = cr[2];
            rr[0] = ar.wt(m); rr[1] = cr[3]; rr[2] = cr[4];
            cam_jac(xz, yz, iz, P.fx, P.fy, &rr[3], &rr[8]);
            rr[13] = 0.0;
          } else {
            const auto rec = reinterpret_cast<const SFT_G v2d*>(P.camrec + (size_t)m * SFT_CAM_STRIDE);
#pragma unroll
            for (int k = 0; k < 7; k++) { const v2d v = rec[k]; rr[2 * k] = v.x; rr[2 * k + 1] = v.y; }
          }
        };
        auto add_obs = [&](const double (&rr)[14], double b) {
          const double om = rr[0] * b;
          sii += om * b;
#pragma unroll
          for (int k = 0; k < 5; k++) { G0[k] += om * rr[3 + k]; G1[k] += om * rr[8 + k]; }
          g0 += om * rr[1];
          g1 += om * rr[2];
        };
        auto add_sh = [&](uint32_t rcv, double c0v, double c1v, const double (&r)[4]) {
          const double wgt = (rcv >> 30) == SFT_KIND_STAR ? P.w_curv : P.w_str;
          const double f = wgt * c0v, g = (wgt * c1v) * r[3];
          const double u0 = r[0], u1 = r[1], u2 = r[2];
          Hs[0] += f * (u0 * u0); Hs[1] += f * (u1 * u0); Hs[2] += f * (u1 * u1);
          Hs[3] += f * (u2 * u0); Hs[4] += f * (u2 * u1); Hs[5] += f * (u2 * u2);
          bn[0] -= g * u0; bn[1] -= g * u1; bn[2] -= g * u2;
        };
        // the first chunk of both lists came a round ahead (cl); their records now; the sums stay in list order per lane
        const auto& mm = cl.mm; const auto& bb = cl.bb; const auto& rc = cl.rc; const auto& c0 = cl.c0; const auto& c1 = cl.c1;
        {
          double rr[DCH][14], r[HCH][4];
#pragma unroll
          for (int i = 0; i < DCH; i++) cam_record(mm[i], rr[i]);
#pragma unroll
          for (int i = 0; i < HCH; i++) ar.rec4((rc[i] >> 30) == SFT_KIND_STAR, rc[i] & 0x3FFFFFu, rc[i] != 0xFFFFFFFFu, r[i]);
#pragma unroll
          for (int i = 0; i < DCH; i++)
            if (ob0 + sub + 8 * i < ob1) add_obs(rr[i], bb[i]);
#pragma unroll
          for (int i = 0; i < HCH; i++)
            if (rc[i] != 0xFFFFFFFFu) add_sh(rc[i], c0[i], c1[i], r[i]);
        }
        for (int p = ob0 + sub + 8 * DCH; p < ob1; p += 8) {   // nodes seen by more than 16 observations
          double rr[14];
          cam_record(P.ob_m[p], rr);
          add_obs(rr, P.ob_c[p]);
        }
        for (int p = sh0 + sub + 8 * HCH; p < sh1; p += 8) {   // more than 24 curvature / stretch contributions
          const uint32_t rcv = P.sh_rec[p];
          double r1[4];
          ar.rec4((rcv >> 30) == SFT_KIND_STAR, rcv & 0x3FFFFFu, true, r1);
          add_sh(rcv, P.sh_cf[2 * p], P.sh_cf[2 * p + 1], r1);
        }
      }
      AS_ADD(33);
      sii = group8_sum(sii); g0 = group8_sum(g0); g1 = group8_sum(g1);   // fixed tree: ((0+4)+(2+6)) + ((1+5)+(3+7))
#pragma unroll
      for (int k = 0; k < 5; k++) { G0[k] = group8_sum(G0[k]); G1[k] = group8_sum(G1[k]); }
#pragma unroll
      for (int k = 0; k < 6; k++) Hs[k] = group8_sum(Hs[k]);
#pragma unroll
      for (int k = 0; k < 3; k++) bn[k] = group8_sum(bn[k]);
      if (on && sub == 0) {
        const double wr = vw ? P.w_ref : 0.0;
        // lower triangle of the 3x3 block: observations (s_ii A^T A), curvature + stretching, reference edge (J = I)
        const double h00 = (sii * (A[0] * A[0] + A[3] * A[3]) + Hs[0]) + wr, h10 = sii * (A[1] * A[0] + A[4] * A[3]) + Hs[1];
        const double h11 = (sii * (A[1] * A[1] + A[4] * A[4]) + Hs[2]) + wr, h20 = sii * (A[2] * A[0] + A[5] * A[3]) + Hs[3];
        const double h21 = sii * (A[2] * A[1] + A[5] * A[4]) + Hs[4], h22 = (sii * (A[2] * A[2] + A[5] * A[5]) + Hs[5]) + wr;
        if (compact) {
          const auto dst = Hc + 9 * (size_t)(a + cur.doff);
          dst[0] = h00; dst[1] = h10; dst[2] = h20; dst[3] = h10; dst[4] = h11; dst[5] = h21; dst[6] = h20; dst[7] = h21; dst[8] = h22;
        } else {
          put(3 * a, 3 * a, h00); put(3 * a + 1, 3 * a, h10); put(3 * a + 1, 3 * a + 1, h11);
          put(3 * a + 2, 3 * a, h20); put(3 * a + 2, 3 * a + 1, h21); put(3 * a + 2, 3 * a + 2, h22);
        }
        {   // camera x node block and b of the node
          const double G0f[6] = {G0[0], G0[1], G0[2], G0[3], 0.0, G0[4]};
          const double G1f[6] = {G1[0], G1[1], G1[2], 0.0, G1[3], G1[4]};
#pragma unroll
          for (int k = 0; k < 6; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) P.Hbord[(size_t)k * Dnp + 3 * a + c] = G0f[k] * A[c] + G1f[k] * A[3 + c];
#pragma unroll
          for (int c = 0; c < 3; c++) P.Hbord[(size_t)6 * Dnp + 3 * a + c] = (bn[c] - (A[c] * g0 + A[3 + c] * g1)) - wr * eref[c];
        }
      }
    }
    AS_ADD(34);
    if constexpr (PARTS & 1) nl = load_dlists(nxt);   // the next round's list entries travel while this round's off-diagonal blocks are summed
    // ---- off-diagonal blocks of the block rows a_lo .. a_hi: one lane per block; the headers of the first 64 came a group ahead
    if constexpr (PARTS & 2)
    for (int q = cur.q; q < cur.qe; q += 64) {
      Hdr h = cur;
      if (q != cur.q) {   // more than 64 off-diagonal blocks in the group (rare): headers on demand
        h.bi = P.off_rc[2 * q]; h.bj = P.off_rc[2 * q + 1];
        const int blk = P.nA + q;
        h.ob0 = P.ob_ptr[blk]; h.ob1 = P.ob_ptr[blk + 1]; h.sh0 = P.sh_ptr[blk]; h.sh1 = P.sh_ptr[blk + 1];
      }
      const int bi = h.bi, bj = h.bj;
      int om[OCH];
      double oc[OCH];
      uint32_t rc[SCH];
      double c0[SCH];
      if (q == cur.q) {
#pragma unroll
        for (int i = 0; i < OCH; i++) { om[i] = om0[i]; oc[i] = oc0[i]; }
#pragma unroll
        for (int i = 0; i < SCH; i++) { rc[i] = orc0[i]; c0[i] = occ0[i]; }
      } else {
#pragma unroll
        for (int i = 0; i < OCH; i++) {
          const bool in = h.ob0 + i < h.ob1;
          om[i] = in ? P.ob_m[h.ob0 + i] : 0;
          oc[i] = in ? P.ob_c[h.ob0 + i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < SCH; i++) {
          const bool in = h.sh0 + i < h.sh1;
          rc[i] = in ? P.sh_rec[h.sh0 + i] : 0xFFFFFFFFu;
          c0[i] = in ? P.sh_cf[2 * (h.sh0 + i)] : 0.0;
        }
      }
      double Ai[6], Aj[6];
#pragma unroll
      for (int k = 0; k < 6; k++) { Ai[k] = ar.A(6 * (size_t)bi + k); Aj[k] = ar.A(6 * (size_t)bj + k); }
      double s = 0.0;
      {
        double wv[OCH];
#pragma unroll
        for (int i = 0; i < OCH; i++) wv[i] = ar.wt(om[i]);
#pragma unroll
        for (int i = 0; i < OCH; i++) s += wv[i] * oc[i];
        for (int p = h.ob0 + OCH; p < h.ob1; p++) s += ar.wt(P.ob_m[p]) * P.ob_c[p];   // more than OCH observations on one mesh edge
      }
      double H[9];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) H[3 * a + b] = s * (Ai[a] * Aj[b] + Ai[3 + a] * Aj[3 + b]);
      auto add_shared = [&](uint32_t rcv, double cf, const double* r) {
        const double f = ((rcv >> 30) == SFT_KIND_STAR ? P.w_curv : P.w_str) * cf;
        const double u0 = r[0], u1 = r[1], u2 = r[2];
        H[0] += f * (u0 * u0); H[1] += f * (u0 * u1); H[2] += f * (u0 * u2);
        H[3] += f * (u1 * u0); H[4] += f * (u1 * u1); H[5] += f * (u1 * u2);
        H[6] += f * (u2 * u0); H[7] += f * (u2 * u1); H[8] += f * (u2 * u2);
      };
      {
        double r[SCH][4];
#pragma unroll
        for (int i = 0; i < SCH; i++) ar.rec4((rc[i] >> 30) == SFT_KIND_STAR, rc[i] & 0x3FFFFFu, rc[i] != 0xFFFFFFFFu, r[i]);
#pragma unroll
        for (int i = 0; i < SCH; i++)
          if (rc[i] != 0xFFFFFFFFu) add_shared(rc[i], c0[i], r[i]);
        for (int p = h.sh0 + SCH; p < h.sh1; p++) {
          const uint32_t rcv = P.sh_rec[p];
          double r1[4];
          ar.rec4((rcv >> 30) == SFT_KIND_STAR, rcv & 0x3FFFFFu, true, r1);
          add_shared(rcv, P.sh_cf[2 * p], r1);
        }
      }
      if (compact) {
        const auto dst = Hc + 9 * (size_t)(bi + 1 + q);
#pragma unroll
        for (int e = 0; e < 9; e++) dst[e] = H[e];
      } else {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b = 0; b < 3; b++) put(3 * bi + a, 3 * bj + b, H[3 * a + b]);
      }
    }
    AS_ADD(35);
    AS_ADD(36);
#if defined(SFT_PHASE_TIMERS) && defined(DSH_LAB)
    if (threadIdx.x == 0) P.dbg[37] += 1.0;
#endif
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Banded Cholesky of (H + lambda I) with the 7-row border, right-looking, NB-wide panels.
//   Lb    : lower band factor (row-major band), Lbord rows 0-5: L_cam,node ; row 6: forward-solved b
//   panel : LDS, k-major: panel[k*LDP + row], row = 0..nb+m+7
// Returns (in ctl->fact_ok) 0 when a pivot is not positive (== Eigen LDLT !isPositive()).
// ------------------------------------------------------------------------------------------
__device__ void factor_and_solve(const SftDev& P, Ctl* ctl, double* panel, double* red) {
  const int Dn = P.Dn, kd = P.kd, ldh = P.ldh;
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  const double lambda = ctl->lambda;
  const int tid = threadIdx.x;
  PH_T0();
  // working copy L <- H + lambda I
  for (size_t i = tid; i < (size_t)Dnp * ldh; i += SFT_NT) {
    double v = P.Hb[i];
    const int k = (int)(i % ldh);
    const int r = (int)(i / ldh);
    if (k == kd && r < Dn) v += lambda;
    P.Lb[i] = v;
  }
  for (size_t i = tid; i < (size_t)SFT_BORDER * Dnp; i += SFT_NT) P.Lbord[i] = P.Hbord[i];
  if (tid < 49) {
    double v = P.Hcorner[tid];
    if (tid % 8 == 0 && tid < 48) v += lambda;  // diagonal of the 6x6 camera block
    P.Lcorner[tid] = v;
  }
  if (tid == 0) ctl->fact_ok = 1;
  __syncthreads();
  PH_ADD(3);

  for (int j0 = 0; j0 < Dnp; j0 += NB) {
    const int m = min(kd, Dnp - j0 - NB);   // band rows below the diagonal block
    const int rows = NB + m + SFT_BORDER;   // panel rows
    const int LDP = rows | 1;               // odd leading dimension
    // ---- load + factor the panel, one thread per row, row kept in registers ----------------
    // raw diagonal block (lower) goes to LDS first so every thread can rebuild pivots
    double* diagraw = panel + (size_t)NB * LDP;          // NB*NB, [r*NB + c]
    double* lrow = diagraw + NB * NB;                    // finalised rows of the diagonal block [r*NB + c]
    for (int e = tid; e < NB * NB; e += SFT_NT) {
      const int r = e / NB, c = e % NB;
      double v = 0.0;
      if (c <= r) v = P.Lb[(size_t)(j0 + r) * ldh + (c - r + kd)];
      diagraw[e] = v;
    }
    __syncthreads();
    {
      const bool active = tid < rows;
      const bool wave_active = (tid & ~63) < rows;   // wave-uniform: idle waves only take part in the barriers
      const int pr = tid;                            // panel row
      double a[NB];
#pragma unroll
      for (int k = 0; k < NB; k++) a[k] = 0.0;
      if (active) {
        if (pr < NB + m) {
          const int r = j0 + pr;
#pragma unroll
          for (int k = 0; k < NB; k++) {
            const int c = j0 + k;
            if (c <= r && r - c <= kd) a[k] = P.Lb[(size_t)r * ldh + (c - r + kd)];
          }
        } else {
          const int br = pr - NB - m;
#pragma unroll
          for (int k = 0; k < NB; k++) a[k] = P.Lbord[(size_t)br * Dnp + j0 + k];
        }
      }
      bool bad = false;
#pragma unroll
      for (int k = 0; k < NB; k++) {
        if (wave_active) {
          // d = A[k][k] - sum_j L[k][j]^2 ; v = a[k] - sum_j a[j] L[k][j]   (j < k), two chains each
          double d0 = diagraw[k * NB + k], d1 = 0.0, v0 = a[k], v1 = 0.0;
#pragma unroll
          for (int j = 0; j + 1 < k; j += 2) {
            const double l0 = lrow[k * NB + j], l1 = lrow[k * NB + j + 1];
            d0 -= l0 * l0; d1 -= l1 * l1;
            v0 -= a[j] * l0; v1 -= a[j + 1] * l1;
          }
          if (k & 1) { const double l0 = lrow[k * NB + k - 1]; d0 -= l0 * l0; v0 -= a[k - 1] * l0; }
          const double d = d0 + d1;
          if (!(d > 0.0)) bad = true;
          const double piv = sqrt(d);
          const double v = v0 + v1;
          a[k] = (pr == k) ? piv : ((pr < k) ? 0.0 : v / piv);
          if (pr < NB && pr >= k) lrow[pr * NB + k] = a[k];
        }
        if (k + 1 < NB) __syncthreads();   // row k+1 of the diagonal block must be complete before step k+1
      }
      if (bad && tid == 0) ctl->fact_ok = 0;
      if (active) {
#pragma unroll
        for (int k = 0; k < NB; k++) panel[(size_t)k * LDP + pr] = a[k];
        if (pr < NB + m) {
          const int r = j0 + pr;
#pragma unroll
          for (int k = 0; k < NB; k++) {
            const int c = j0 + k;
            if (c <= r && r - c <= kd) P.Lb[(size_t)r * ldh + (c - r + kd)] = a[k];
          }
        } else {
          const int br = pr - NB - m;
#pragma unroll
          for (int k = 0; k < NB; k++) P.Lbord[(size_t)br * Dnp + j0 + k] = a[k];
        }
      }
    }
    __syncthreads();
    PH_ADD(4);
    // ---- trailing update: window rows/cols [0, m+7) relative to j0+NB ----------------------
    const int W = m + SFT_BORDER;
    const int T = (W + 3) >> 2;
    const int ntiles = T * (T + 1) / 2;
    for (int q = tid; q < ntiles; q += SFT_NT) {
      int ti = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
      while ((ti + 1) * (ti + 2) / 2 <= q) ti++;
      while (ti * (ti + 1) / 2 > q) ti--;
      const int tj = q - ti * (ti + 1) / 2;
      const int r0 = 4 * ti, c0 = 4 * tj;
      double acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
#pragma unroll 4
      for (int k = 0; k < NB; k++) {
        const double* col = panel + (size_t)k * LDP + NB;
        double ar[4], ac[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { ar[i] = (r0 + i < W) ? col[r0 + i] : 0.0; ac[i] = (c0 + i < W) ? col[c0 + i] : 0.0; }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] += ar[i] * ac[j];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int rr = r0 + i;
        if (rr >= W) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int cc = c0 + j;
          if (cc > rr || cc >= W) continue;
          double* dst;
          if (rr < m) {
            const int Rg = j0 + NB + rr, Cg = j0 + NB + cc;
            dst = &P.Lb[(size_t)Rg * ldh + (Cg - Rg + kd)];
          } else if (cc < m) {
            dst = &P.Lbord[(size_t)(rr - m) * Dnp + j0 + NB + cc];
          } else {
            dst = &P.Lcorner[(rr - m) * 7 + (cc - m)];
          }
          *dst -= acc[i][j];
        }
      }
    }
    __syncthreads();
    PH_ADD(5);
  }
  // ---- corner: Cholesky of the 6x6 Schur complement, forward-solve its right-hand side, x_cam ---
  double* x = P.x;
  if (tid == 0) {
    double* C = P.Lcorner;
    bool bad = false;
    for (int k = 0; k < 6; k++) {
      double d = C[k * 7 + k];
      for (int j = 0; j < k; j++) d -= C[k * 7 + j] * C[k * 7 + j];
      if (!(d > 0.0)) bad = true;
      const double piv = sqrt(d);
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
      for (int r = k + 1; r < 6; r++) v -= C[r * 7 + k] * x[Dnp + r];
      x[Dnp + k] = v / C[k * 7 + k];
    }
  }
  __syncthreads();
  // ---- back substitution over the node blocks (L^T x = y - Lcn^T x_cam) ----------------------
  // (skipped when the factorisation failed: like g2o, x then keeps its previous content)
  if (ctl->fact_ok) {
    double* tvec = panel;            // NB
    double* part = panel + NB;       // 32 parts x NB
    constexpr int NPART = SFT_NT / NB;
    double* dblk = part + NPART * NB;   // NB x NB diagonal block of L
    const int c = tid & (NB - 1), pidx = tid / NB;  // NB columns x NPART parts
    double xc[6];
#pragma unroll
    for (int i = 0; i < 6; i++) xc[i] = x[Dnp + i];
    for (int j0 = Dnp - NB; j0 >= 0; j0 -= NB) {
      const int C = j0 + c;
      double s = 0.0;
      const int rend = min(Dnp - 1, C + kd);
      for (int Rr = j0 + NB + pidx; Rr <= rend; Rr += NPART) s += P.Lb[(size_t)Rr * ldh + (C - Rr + kd)] * x[Rr];
      if (pidx < 6) s += P.Lbord[(size_t)pidx * Dnp + C] * xc[pidx];
      part[pidx * NB + c] = s;
      for (int e = tid; e < NB * NB; e += SFT_NT) {
        const int r = e / NB, cc2 = e % NB;
        dblk[e] = (cc2 <= r) ? P.Lb[(size_t)(j0 + r) * ldh + (cc2 - r + kd)] : 0.0;
      }
      __syncthreads();
      if (tid < NB) {
        double s2 = 0.0;
        for (int p2 = 0; p2 < NPART; p2++) s2 += part[p2 * NB + tid];
        tvec[tid] = P.Lbord[(size_t)6 * Dnp + j0 + tid] - s2;
      }
      __syncthreads();
      if (tid < 64) {
        double tv = (tid < NB) ? tvec[tid] : 0.0;
        for (int k = NB - 1; k >= 0; k--) {
          const double xk = __shfl(tv, k, 64) / dblk[k * NB + k];
          if (tid == k) tv = xk;
          else if (tid < k) tv -= dblk[k * NB + tid] * xk;
        }
        if (tid < NB) x[j0 + tid] = tv;
      }
      __syncthreads();
    }
  }
  PH_ADD(6);
}


// ------------------------------------------------------------------------------------------
// Tile mode (half-bandwidth <= 16*BT): right-looking Cholesky on 16x16 tiles.
//   * the trailing window (BT x BT tiles, lower triangle live) lives in MFMA accumulator registers:
//     wave w owns ring slots (a = w>>1, b = 4*(w&1)+t), t = 0..3; slot (a,b) holds tile (I,J) with
//     I = a, J = b (mod BT) inside the current window [k+1, k+BT]
//   * every step publishes block column k to LDS, wave 0 factors the diagonal tile and inverts it,
//     the sub-diagonal tiles become X = A * Linv^T (MFMA), the window gets C -= X_I X_J^T (MFMA)
//   * the 6 camera rows + the right-hand side ride along as a 7 x 128 ring in LDS (VALU)
// H is read once (fresh tiles enter the window as it slides), L is written once.
// ------------------------------------------------------------------------------------------


// Barrier that orders LDS traffic only: outstanding global loads/stores stay in flight across it
// (prefetched H tiles must not be drained at every step; L tiles are first re-read after the loop).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The solver workspace lives in LDS.  A non-inlined function only sees a generic pointer; the explicit address space keeps
// its accesses on ds_* instructions (generic = flat_* instructions, which count on both the LDS and the memory counter).

// Wave-uniform values that reach a (non-inlined) device function through memory or VGPR arguments: moving them to
// scalar registers keeps ring bookkeeping, tile addresses and branches on the scalar unit.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* uni(T* p) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

#ifdef DSH_LAB   // the barrier version of the factor steps: A/B reference of the dataflow version below (dsh_lab_set_option "dataflow" 0)
// Tile-mode factorisation on NW wavefronts (NW = 8: one ring row per wave, lowest latency; NW = 4: two ring rows per
// wave so that two problems share a CU).  Wave w owns ring rows a = w + NW*t (t < RPW) of the BT x BT accumulator window
// (slots b = 0..BT-1; tile (I,J) of the window [k+1, k+BT] sits in slot (I mod BT, J mod BT)), the border tiles of the
// same ring columns (7 camera/rhs rows x 16 columns) and, for wave 0, the 7x7 corner.  One step:
//   C(k): X_i = A_i Linv_k^T for the published block column k (MFMA GEMM), border panel on 16 lanes per border row
//   D(k): block column k+1 of the window is updated first and published; the owner of tile row k+1 factors the next
//         diagonal tile while the other waves update the rest of the window (look-ahead); border and corner follow
template <int NW>
__device__ __noinline__ void factor_tiles(const SftDev& P, Ctl* ctl, double* ws) {
  constexpr int RPW = BT / NW;
  constexpr int NT = 64 * NW;
  static_assert(RPW * NW == BT && NW >= 2, "ring rows must divide evenly over the wavefronts");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps ring bookkeeping and branches on the scalar unit
  const int Dn = uni(P.Dn);
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  const int nT = Dnp / TS;
  lds_double* Araw = to_lds(ws);                        // (BT+1) tiles, k-major padded (slot 0 unused)
  lds_double* Xp = Araw + (BT + 1) * TILE_LDS;   // (BT+1) tiles; slot 0: border panel as a tile (rows 7..15 zero)
  lds_double* LinvK = Xp + (BT + 1) * TILE_LDS;  // Linv^T, k-major padded: LinvK[k*TP + j] = Linv[j][k]
  lds_double* Abord = LinvK + TILE_LDS;          // 7 x 16 border block of the published column, row-major
  lds_double* Cn = Abord + SFT_BORDER * TS;      // 7 x 7 corner (written once at the end)
  const double lambda = ctl->lambda;
  const int crow = lane >> 4, ccol = lane & 15;   // accumulator layout: rows crow + 4q, column ccol
  const auto Hbord = uni(P.Hbord);
  const auto Lg = uni(P.Lb);
  const auto Lbord = uni(P.Lbord);
  const auto Linv_g = uni(P.Linv);
#ifdef SFT_EXPERIMENTS   // tuning builds only (EXTRA=-DSFT_EXPERIMENTS): DSH_EXPERIMENT bits switch phases off, results are then invalid
  const int mode = uni(P.mode);
#define SFT_EXP(bit) ((mode & (bit)) != 0)
#else
#define SFT_EXP(bit) false
#endif
  ST_BEGIN();
  v4d acc[RPW][BT];
  v4d bacc[RPW];                        // border tiles of ring columns (wave + NW*t + BOFF) mod BT: never on the wave that factors that column
  constexpr int BOFF = 2;

  // Raw tile (I, I-d) of H in accumulator layout, gathered from the compact 3x3 blocks through the graph's element list: loads
  // are unconditional (exact wait counts).  Structurally zero tiles (no 3x3 block touches them: 30 % of the C2 band) take the
  // all-zero list row, which stays in cache.  mrow: the mask word of tile row I (P.tmask).
  const auto tmask = uni(P.tmask);
  const auto hgl = uni(P.hgather);
  const auto Hc = uni(P.Hc);
  // two phases so that no wait sits between the tiles of a row: first the byte offsets of every tile (16 bytes per lane; the all-zero
  // list row nT, which stays in cache, for structurally zero tiles), then the elements of the tiles the mask has (wave-uniform branch)
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  auto fresh_idx = [&](int I, int d, int mrow) -> v4u_t {
    return *reinterpret_cast<const SFT_G v4u_t*>(hgl + (((size_t)(((mrow >> d) & 1) ? I : nT) * (BT + 1) + d) * 64 + lane) * 4);
  };
  auto fresh_data = [&](const v4u_t& ix, int d, int mrow) -> v4d {
    v4d v = {0.0, 0.0, 0.0, 0.0};
    if ((mrow >> d) & 1) {
      const auto Hb8 = reinterpret_cast<const SFT_G char*>(Hc);
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const SFT_G double*>(Hb8 + ix[q]);
    }
    return v;
  };
  // border block of tile column J: rows crow, crow+4 of the 8-row border (row 7 is zero), accumulator layout
  auto fresh_border = [&](int J) -> v4d {
    v4d v = {0.0, 0.0, 0.0, 0.0};
    v[0] = Hbord[(size_t)crow * Dnp + TS * J + ccol];
    v[1] = Hbord[(size_t)(crow + 4) * Dnp + TS * J + ccol];
    return v;
  };
#pragma unroll
  for (int t = 0; t < RPW; t++) {       // rows 0..BT-1
    const int I = wave + NW * t;
    const int m0 = uni(tmask[I]);
    { v4u_t ix[BT];
#pragma unroll
      for (int b = 0; b < BT; b++) ix[b] = fresh_idx(I, (I - b) & (BT - 1), m0);
#pragma unroll
      for (int b = 0; b < BT; b++) acc[t][b] = fresh_data(ix[b], (I - b) & (BT - 1), m0); }
    bacc[t] = fresh_border((I + BOFF) & (BT - 1));
  }
  v4d cacc = {0.0, 0.0, 0.0, 0.0};      // wave 0: corner
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER && ccol <= r) cacc[q] = P.Hcorner[r * 7 + ccol] + ((r == ccol && r < 6) ? lambda : 0.0);
    }
  }
  for (int i = tid; i < TILE_LDS; i += NT) Xp[i] = 0.0;   // rows 7..15 of the border panel tile stay zero
  if (tid == 0) ctl->fact_ok = 1;
  __syncthreads();
  PH_T0();

#pragma unroll 1
  for (int k = -1; k < nT; k++) {
    const int kc = k + 1;                  // block column published / factored in this D phase
    const int kslot = kc & (BT - 1);       // its ring slot
    ST_DONE();
    ST_MARK(0);
    // ---- global-memory duties of this step, all on the wave that factored column k (its ring row is free again and it
    // sits at the far end of the window, off the critical path): recycle the ring row with tile row k+BT, fetch tile
    // (kc+BT, kc) for the published column, and at the end of D store block column k of L from the LDS panel.
    // No other wave has vector-memory traffic in flight inside the loop, so the vmcnt waits the compiler places in
    // front of the accumulator MFMAs cost the other waves nothing (the owner's Linv store precedes its turn here).
    bool memwave = false;
    v4d fr8 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < RPW; t++)
      if (wave + NW * t == (k & (BT - 1))) {
        memwave = true;
        if (k >= 0) {
          const int I = k + BT;
          const int mI = uni(tmask[I]);
#pragma unroll
          for (int b = 0; b < BT; b++) acc[t][b] = fresh_data(fresh_idx(I, (I - b) & (BT - 1), mI), (I - b) & (BT - 1), mI);
        }
      }
    if (memwave) fr8 = fresh_data(fresh_idx(kc + BT, BT, uni(tmask[kc + BT])), BT, uni(tmask[kc + BT]));
    if (k >= 0) {
#pragma unroll
      for (int t = 0; t < RPW; t++)   // the holder of ring column k mod BT (consumed by now) fetches the border block of column k+BT
        if (((wave + NW * t + BOFF) & (BT - 1)) == (k & (BT - 1))) bacc[t] = fresh_border(k + BT);
    }
    if (k >= 0 && !SFT_EXP(16)) {
      // ---- C(k): X_i = A_i Linv^T (RPW sub-diagonal tiles per wave), border panel on 16 lanes per border row ------
      v4d x[RPW], x2[RPW];
      double bv[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) bv[kk] = LinvK[(4 * kk + crow) * TP + ccol];
#pragma unroll
      for (int t = 0; t < RPW; t++) {
        const int i = wave + 1 + NW * t;
        double av[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) av[kk] = Araw[i * TILE_LDS + (4 * kk + crow) * TP + ccol];
        x[t] = (v4d){0.0, 0.0, 0.0, 0.0}; x2[t] = x[t];
        x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], x[t], 0, 0, 0);     // two independent accumulation chains
        x2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], x2[t], 0, 0, 0);
        x[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], x[t], 0, 0, 0);
        x2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], x2[t], 0, 0, 0);
      }
      if (wave == NW - 1) {   // border panel (7 camera/rhs rows) as one more MFMA tile: rows 7..15 of the operand are zero
        double av[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) av[kk] = (ccol < SFT_BORDER) ? Abord[ccol * TS + 4 * kk + crow] : 0.0;
        v4d xb = {0.0, 0.0, 0.0, 0.0}, xb2 = xb;
        xb = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], xb, 0, 0, 0);
        xb2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], xb2, 0, 0, 0);
        xb = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], xb, 0, 0, 0);
        xb2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], xb2, 0, 0, 0);
        xb += xb2;
        Xp[ccol * TP + crow] = xb[0];                                   // slot 0 of Xp, k-major like the other panel tiles
        if (crow + 4 < SFT_BORDER) Xp[ccol * TP + crow + 4] = xb[1];
      }
#pragma unroll
      for (int t = 0; t < RPW; t++) {
        x[t] += x2[t];
        lds_double* dst = Xp + (wave + 1 + NW * t) * TILE_LDS + ccol * TP + crow;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[4 * q] = x[t][q];
      }
      ST_MARK(1);
      lds_barrier();
      PH_ADD(0);
    }
    ST_MARK(2);
    // ---- D(k): trailing update of window / border / corner, look-ahead factorisation of column kc --
    int I[RPW];                                     // tile rows held by this wave inside the window [kc, kc+BT-1]
#pragma unroll
    for (int t = 0; t < RPW; t++) I[t] = kc + ((wave + NW * t - kc) & (BT - 1));
    double an[RPW][4], bn[4];   // one ring row per wave: kept for the rest of the window; two rows: re-read where needed (registers)
    if constexpr (RPW == 1) {
      if (k >= 0) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          an[0][kk] = -Xp[(I[0] - k) * TILE_LDS + (4 * kk + crow) * TP + ccol];
          bn[kk] = -Xp[(4 * kk + crow) * TP + ccol];                 // border panel
        }
      }
    }
    // block column kc first: it is what the next step needs published.  Update, then publish / hand to the factorisation
    // straight from the accumulator (no copies of the tiles are kept).
    v4d dtile = {0.0, 0.0, 0.0, 0.0};
    bool owner = false;
    {
      double bc[4] = {0.0, 0.0, 0.0, 0.0};
      const bool upd = k >= 0 && !SFT_EXP(4);
      if (upd) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) bc[kk] = Xp[TILE_LDS + (4 * kk + crow) * TP + ccol];
      }
#pragma unroll
      for (int c = 0; c < BT; c++)
        if (c == kslot) {
#pragma unroll
          for (int t = 0; t < RPW; t++) {
            if (upd && I[t] < nT) {   // two accumulation chains halve the dependent-MFMA latency of the critical tile
              double a4[4];
#pragma unroll
              for (int kk = 0; kk < 4; kk++) a4[kk] = (RPW == 1) ? an[t][kk] : -Xp[(I[t] - k) * TILE_LDS + (4 * kk + crow) * TP + ccol];
              v4d side = {0.0, 0.0, 0.0, 0.0};
              acc[t][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[0], bc[0], acc[t][c], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);   // keep the two chains interleaved (the scheduler would serialise them)
              side = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[1], bc[1], side, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              acc[t][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[2], bc[2], acc[t][c], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              side = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[3], bc[3], side, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              acc[t][c] += side;
            }
            if (kc < nT) {
              if (I[t] == kc) {
                owner = true;
                dtile = acc[t][c];
              } else {   // publish tile (I, kc) of block column kc
                lds_double* dst = Araw + (I[t] - kc) * TILE_LDS + ccol * TP + crow;
#pragma unroll
                for (int q = 0; q < 4; q++) dst[4 * q] = acc[t][c][q];
              }
            }
          }
        }
      ST_MARK(3);
#pragma unroll
      for (int t = 0; t < RPW; t++)
        if (((wave + NW * t + BOFF) & (BT - 1)) == kslot) {   // holder of the border tile of column kc: update and publish for the next C phase
          if (upd) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
              const double b4 = (RPW == 1) ? bn[kk] : -Xp[(4 * kk + crow) * TP + ccol];
              bacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b4, bc[kk], bacc[t], 0, 0, 0);
            }
          }
          if (kc < nT) {
            Abord[crow * TS + ccol] = bacc[t][0];
            if (crow + 4 < SFT_BORDER) Abord[(crow + 4) * TS + ccol] = bacc[t][1];
          }
        }
    }
    if (kc < nT) {
      if (memwave) {
        lds_double* dst = Araw + BT * TILE_LDS + ccol * TP + crow;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[4 * q] = fr8[q];
      }
      if (owner) {
        // owner of tile row kc: factor the diagonal tile (critical path)
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (crow + 4 * q == ccol && TS * kc + ccol < Dn) dtile[q] += lambda;
        v4d w = dtile;
        ST_MARK(4);
        const bool ok = SFT_EXP(8) ? true : chol_inv_blocked(dtile, w);
        ST_MARK(5);
        if (!ok && lane == 0) ctl->fact_ok = 0;
        lds_double* dst = LinvK + ccol * TP + crow;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[4 * q] = w[q];
        *reinterpret_cast<v4d*>(Linv_g + (size_t)kc * TS * TS + 4 * lane) = w;
        __builtin_amdgcn_s_setprio(0);
      }
    }
    if (k >= 0) {
      // rest of the window (columns kc+1 ..), remaining border tiles, corner
      bool more = false;    // does any ring row of this wave reach beyond block column kc?
#pragma unroll
      for (int t = 0; t < RPW; t++) more = more || (I[t] < nT && I[t] > kc);
      if constexpr (RPW > 1) {   // two ring rows per wave: registers are scarce across the factorisation, re-read the operands
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
#pragma unroll
          for (int t = 0; t < RPW; t++) an[t][kk] = -Xp[(I[t] - k) * TILE_LDS + (4 * kk + crow) * TP + ccol];
          bn[kk] = -Xp[(4 * kk + crow) * TP + ccol];
        }
      }
      if (more && !SFT_EXP(4)) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          // one batch of LDS reads per k-chunk (a single wait), then MFMAs on independent accumulators back to back
          double bb[BT];
#pragma unroll
          for (int b = 0; b < BT; b++) bb[b] = Xp[(kc + ((b - kc) & (BT - 1)) - k) * TILE_LDS + (4 * kk + crow) * TP + ccol];
#pragma unroll
          for (int b = 0; b < BT; b++) {
            const int J = kc + ((b - kc) & (BT - 1));
#pragma unroll
            for (int t = 0; t < RPW; t++)
              if (I[t] < nT && J <= I[t] && J != kc) acc[t][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[t][kk], bb[b], acc[t][b], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < RPW; t++) {
        const int Jb = kc + ((wave + NW * t + BOFF - kc) & (BT - 1));   // global tile column of this border tile
        if (Jb != kc) {
          double bbord[4];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) bbord[kk] = Xp[(Jb - k) * TILE_LDS + (4 * kk + crow) * TP + ccol];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) bacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[kk], bbord[kk], bacc[t], 0, 0, 0);
        }
      }
      if (wave == 0) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[kk], -bn[kk], cacc, 0, 0, 0);
      }
    }
    if (memwave && k >= 0 && !SFT_EXP(16) && !SFT_EXP(32)) {
      // block column k of L goes to global memory from the LDS panel (native tile layout), border rows included
#pragma unroll
      for (int i = 1; i <= BT; i++)
        if (k + i < nT) {
          const lds_double* src = Xp + i * TILE_LDS + ccol * TP + crow;
          v4d x;
#pragma unroll
          for (int q = 0; q < 4; q++) x[q] = src[4 * q];
          *reinterpret_cast<v4d*>(Lg + tile_off(k, i) + 4 * lane) = x;   // L is stored by block COLUMN: tile (k+i, k) at slot (k, i)
        }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int e = lane + 64 * h, r = e >> 4, j = e & 15;
        if (r < SFT_BORDER) Lbord[(size_t)r * Dnp + TS * k + j] = Xp[j * TP + r];
      }
    }
    ST_MARK(6);
    lds_barrier();
    ST_MARK(7);
    PH_ADD(5);
  }
  ST_END();
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER) Cn[r * 7 + ccol] = cacc[q];
    }
  }
  __syncthreads();
  // ---- corner: 6x6 Schur complement of the camera, its right-hand side, x_cam -------------------
  if (tid == 0) {
    bool bad = false;
    for (int k = 0; k < 6; k++) {
      double d = Cn[k * 7 + k];
      for (int j = 0; j < k; j++) d -= Cn[k * 7 + j] * Cn[k * 7 + j];
      if (!(d > 0.0)) bad = true;
      const double piv = sqrt(d);
      Cn[k * 7 + k] = piv;
      for (int r = k + 1; r < 7; r++) {
        double v = Cn[r * 7 + k];
        for (int j = 0; j < k; j++) v -= Cn[r * 7 + j] * Cn[k * 7 + j];
        Cn[r * 7 + k] = v / piv;
      }
    }
    if (bad) ctl->fact_ok = 0;
    if (ctl->fact_ok)
      for (int k = 5; k >= 0; k--) {
        double v = Cn[6 * 7 + k];
        for (int r = k + 1; r < 6; r++) v -= Cn[r * 7 + k] * P.x[Dnp + r];
        P.x[Dnp + k] = v / Cn[k * 7 + k];
      }
  }
  __syncthreads();
}

#endif  // DSH_LAB

// ------------------------------------------------------------------------------------------
// Dataflow variant of the tile-mode factorisation (8 wavefronts): no workgroup barriers inside the step loop.
// Every wave TRSMs the tile of its OWN ring row (X_i = A_i W_k^T: the raw tile is its own publication, private LDS
// slot), so the critical chain of a step runs inside one wave -- the owner of row k+1 waits for W_k, forms its X,
// updates its diagonal tile and factors it -- while the other waves consume X tiles of step k as they appear
// (LDS flags written after the data, polled with s_sleep) and may lag behind by up to one step (Xp / LinvK / Abord
// are double buffered by step parity; a wave re-enters a buffer only after every wave has finished with it).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void flag_set(lds_int* f, int v) {      // publish: data first, then the flag
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) *(volatile lds_int*)f = v;
}
__device__ __forceinline__ void flag_wait(lds_int* f, int v) {     // all lanes poll the same word
  while (__builtin_amdgcn_readfirstlane(*(volatile lds_int*)f) < v) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
// wait until flags[j] >= v for every j in [lo, hi] (at most 16 flags, one per lane group position)
__device__ __forceinline__ void flags_wait(lds_int* flags, int lo, int hi, int v) {
  const int j = threadIdx.x & 15;
  while (true) {
    const int x = *(volatile lds_int*)(flags + j);
    const bool ok = (j < lo) || (j > hi) || (x >= v);
    if (__all(ok)) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

// 8-wavefront specialisation (one ring row per wave): the same algorithm as factor_tiles_df<NW> below written without the
// per-row loops -- the compiler allocates registers noticeably better for it (10.8 vs 12.0 ms per C2 problem).
// Cholesky of the 6x6 Schur complement of the camera (lower triangle of Cn, row 6 = right-hand side), forward solve of its
// right-hand side, x_cam into P.x[Dnp ..]; one thread.
__device__ __forceinline__ void corner_finish(const SftDev& P, Ctl* ctl, lds_double* Cn, int Dnp) {
  bool bad = false;
  for (int k = 0; k < 6; k++) {
    double d = Cn[k * 7 + k];
    for (int j = 0; j < k; j++) d -= Cn[k * 7 + j] * Cn[k * 7 + j];
    if (!(d > 0.0)) bad = true;
    const double piv = sqrt(d);
    Cn[k * 7 + k] = piv;
    for (int r = k + 1; r < 7; r++) {
      double v = Cn[r * 7 + k];
      for (int j = 0; j < k; j++) v -= Cn[r * 7 + j] * Cn[k * 7 + j];
      Cn[r * 7 + k] = v / piv;
    }
  }
  if (bad) ctl->fact_ok = 0;
  if (ctl->fact_ok)
    for (int k = 5; k >= 0; k--) {
      double v = Cn[6 * 7 + k];
      for (int r = k + 1; r < 6; r++) v -= Cn[r * 7 + k] * P.x[Dnp + r];
      P.x[Dnp + k] = v / Cn[k * 7 + k];
    }
}

// lam_corner: damping added to the 6x6 camera block (the shared-camera mode adds it on one rank only); finish_corner = false leaves the
// 7x7 Schur complement of the camera (lower triangle + right-hand side row) in P.Lcorner instead of solving for the camera update.
// (INST: the tail kernel of the batched rounds takes an instance of its own -- a call of the shared one from there sends hipcc 7.2 into
// "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base")
template <int INST = 0>
__device__ __noinline__ void factor_tiles_df8(const SftDev& P, Ctl* ctl, double* ws, double lam_corner, bool finish_corner) {
  constexpr int NW = 8, NT = 64 * NW, BOFF = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Dn = uni(P.Dn);
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  const int nT = Dnp / TS;
  lds_double* Aself = to_lds(ws) + wave * TILE_LDS;            // this wave's own raw tile of the current column, operand layout
  lds_double* XpB = to_lds(ws) + NW * TILE_LDS;                // 2 x (BT+1) X tiles, slot 0 = border panel
  lds_double* LinvB = XpB + 2 * (BT + 1) * TILE_LDS;           // 3 x Linv^T (column mod 3: a lagging wave may still read W of step k-1 while W of k+1 is written)
  lds_double* AbordB = LinvB + 3 * TILE_LDS;                   // 3 x 7x16 border block
  lds_double* Cn = AbordB + 3 * SFT_BORDER * TS;               // 7x7 corner
  lds_int* F = (lds_int*)(Cn + 64);                            // flags: [0] W ready (column), [1] border block ready, [2..10] X tile j ready (step+1), [16..23] wave done (step+1)
  lds_int* wflag = F, *bflag = F + 1, *dflag = F + 24;
  const double lambda = ctl->lambda;
  const int crow = lane >> 4, ccol = lane & 15;
  const auto Hbord = uni(P.Hbord);
  const auto Lg = uni(P.Lb);
  const auto Lbord = uni(P.Lbord);
  const auto Linv_g = uni(P.Linv);
  v4d acc[BT];
  v4d bacc;
#if defined(SFT_STEP_TRACE) && defined(DSH_LAB)
  long long wt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt0;   // cycles spent waiting: [0] buffers free [1] W [2] X1 [3] X2..i [4] border [5] store flags [6] chol [7] whole loop
  const long long wloop0 = clock64();
#define WT_BEGIN() wt0 = clock64()
#define WT_END(e) wt[e] += clock64() - wt0
#else
#define WT_BEGIN() do {} while (0)
#define WT_END(e) do {} while (0)
#endif
  // Tiles of H are gathered from the compact 3x3 blocks (unconditional loads, exact wait counts); structurally zero tiles (30 % of
  // the C2 band) take the all-zero list row, which stays in cache.  mrow: the mask word of tile row I (P.tmask).
  const auto tmask = uni(P.tmask);
  const auto hgl = uni(P.hgather);
  const auto Hc = uni(P.Hc);
  // two phases so that no wait sits between the tiles of a row: first the byte offsets of every tile (16 bytes per lane; the all-zero
  // list row nT, which stays in cache, for structurally zero tiles), then the elements of the tiles the mask has (wave-uniform branch)
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  auto fresh_idx = [&](int I, int d, int mrow) -> v4u_t {
    return *reinterpret_cast<const SFT_G v4u_t*>(hgl + (((size_t)(((mrow >> d) & 1) ? I : nT) * (BT + 1) + d) * 64 + lane) * 4);
  };
  auto fresh_data = [&](const v4u_t& ix, int d, int mrow) -> v4d {
    v4d v = {0.0, 0.0, 0.0, 0.0};
    if ((mrow >> d) & 1) {
      const auto Hb8 = reinterpret_cast<const SFT_G char*>(Hc);
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const SFT_G double*>(Hb8 + ix[q]);
    }
    return v;
  };
  auto fresh_border = [&](int J) -> v4d {
    v4d v = {0.0, 0.0, 0.0, 0.0};
    v[0] = Hbord[(size_t)crow * Dnp + TS * J + ccol];
    v[1] = Hbord[(size_t)(crow + 4) * Dnp + TS * J + ccol];
    return v;
  };
  const int m0 = uni(tmask[wave]);
#pragma unroll
  for (int b = 0; b < BT; b++) acc[b] = fresh_data(fresh_idx(wave, (wave - b) & (BT - 1), m0), (wave - b) & (BT - 1), m0);
  bacc = fresh_border((wave + BOFF) & (BT - 1));
  v4d cacc = {0.0, 0.0, 0.0, 0.0};
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER && ccol <= r) cacc[q] = P.Hcorner[r * 7 + ccol] + ((r == ccol && r < 6) ? lam_corner : 0.0);
    }
  }
  for (int i = tid; i < 2 * (BT + 1) * TILE_LDS; i += NT) XpB[i] = 0.0;   // rows 7..15 of both border panel tiles stay zero
  if (tid < 48) F[tid] = (tid == 0 || tid == 1) ? -1 : 0;
  if (tid == 0) ctl->fact_ok = 1;
  __syncthreads();

  v4d araw_n = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
  for (int k = -1; k < nT; k++) {
    const int kc = k + 1, kslot = kc & (BT - 1), par = k & 1, p3 = (k + 3) % 3, p3c = kc % 3;
    const int I = kc + ((wave - kc) & (BT - 1));     // ring row held during this step: window [kc, kc+BT-1]
    const int i = I - k;                             // its tile of block column k is (I, k): X slot i in 1..BT
    lds_double* Xp = XpB + par * (BT + 1) * TILE_LDS;
    lds_int* xflag = F + 2 + 10 * par;
    const bool memwave = wave == (k & (BT - 1));     // factored column k: its ring row holds tile row k+BT now, tile (k+BT, k) waits in araw_n
    // the wave that factored column k recycled its ring row with tile row k+BT right behind its Cholesky of the previous step
    // (below): tile (k+BT, k) is raw H and waits in araw_n
    if (k >= 0) {
      if (((wave + BOFF) & (BT - 1)) == (k & (BT - 1))) bacc = fresh_border(k + BT);
      // nobody may still be reading the buffers of step k-2 (same parity)
      WT_BEGIN();
      if (k >= 2) flags_wait(dflag, 0, NW - 1, k - 1);
      WT_END(0);
      // ---- C: X_i = A_i W_k^T for the tile of the own ring row ----
      WT_BEGIN();
      flag_wait(wflag, k);
      WT_END(1);
      lds_double* LinvK = LinvB + p3 * TILE_LDS;
      if (memwave) {                                 // its raw tile arrives from HBM: through the private LDS slot into operand layout
        lds_double* dst = Aself + ccol * TP + crow;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[4 * q] = araw_n[q];
      }
      {
        double av[4], bv[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          av[kk] = Aself[(4 * kk + crow) * TP + ccol];
          bv[kk] = LinvK[(4 * kk + crow) * TP + ccol];
        }
        v4d x = {0.0, 0.0, 0.0, 0.0}, x2 = x;
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], x, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], x2, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], x, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], x2, 0, 0, 0);
        x += x2;
        lds_double* dst = Xp + i * TILE_LDS + ccol * TP + crow;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[4 * q] = x[q];
        flag_set(xflag + i, k + 1);
        // the tile of L leaves from the registers it was computed in (accumulator order = storage order): nobody re-reads the panel
        if (I < nT) *reinterpret_cast<v4d*>(Lg + tile_off(k, i) + 4 * lane) = x;   // L is stored by block COLUMN: tile (k+i, k) at slot (k, i)
        if (i == 4) {                                // border panel (7 camera/rhs rows): one more MFMA tile, off the critical path
          flag_wait(bflag, k);
          lds_double* Abord = AbordB + p3 * SFT_BORDER * TS;
          double ab[4];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) ab[kk] = (ccol < SFT_BORDER) ? Abord[ccol * TS + 4 * kk + crow] : 0.0;
          v4d xb = {0.0, 0.0, 0.0, 0.0}, xb2 = xb;
          xb = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[0], bv[0], xb, 0, 0, 0);
          xb2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[1], bv[1], xb2, 0, 0, 0);
          xb = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[2], bv[2], xb, 0, 0, 0);
          xb2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[3], bv[3], xb2, 0, 0, 0);
          xb += xb2;
          Xp[ccol * TP + crow] = xb[0];
          if (crow + 4 < SFT_BORDER) Xp[ccol * TP + crow + 4] = xb[1];
          flag_set(xflag, k + 1);
          Lbord[(size_t)crow * Dnp + TS * k + ccol] = xb[0];
          if (crow + 4 < SFT_BORDER) Lbord[(size_t)(crow + 4) * Dnp + TS * k + ccol] = xb[1];
        }
      }
    }
    // ---- D1: block column kc of the own row, then publish it (private slot) or factor it (owner) ----
    double an[4] = {0.0, 0.0, 0.0, 0.0};
    v4d dtile = acc[0];        // k == -1: column 0 sits in slot 0
    if (k >= 0) {
      WT_BEGIN();
      flag_wait(xflag + 1, k + 1);                   // X of tile (kc, k); the owner produced it itself
      WT_END(2);
      double bc[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        an[kk] = -Xp[i * TILE_LDS + (4 * kk + crow) * TP + ccol];
        bc[kk] = Xp[TILE_LDS + (4 * kk + crow) * TP + ccol];
      }
      if (I < nT) {
#pragma unroll
        for (int c = 0; c < BT; c++)
          if (c == kslot) {
            v4d side = {0.0, 0.0, 0.0, 0.0};
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[0], bc[0], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            side = __builtin_amdgcn_mfma_f64_16x16x4f64(an[1], bc[1], side, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[2], bc[2], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            side = __builtin_amdgcn_mfma_f64_16x16x4f64(an[3], bc[3], side, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[c] += side;
            dtile = acc[c];      // picked out of the ring inside the wave-uniform branch (a run-time register index costs 56 v_cndmask)
          }
      }
    }
    if (kc < nT) {
      if (I == kc) {
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (crow + 4 * q == ccol && TS * kc + ccol < Dn) dtile[q] += lambda;
        v4d w = dtile;
        WT_BEGIN();
        const bool ok = chol_inv_blocked(dtile, w);
        WT_END(6);
        if (!ok && lane == 0) ctl->fact_ok = 0;
        lds_double* dst = LinvB + p3c * TILE_LDS + ccol * TP + crow;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[4 * q] = w[q];
        flag_set(wflag, kc);
        *reinterpret_cast<v4d*>(Linv_g + (size_t)kc * TS * TS + 4 * lane) = w;
        __builtin_amdgcn_s_setprio(0);
        // The ring row of column kc is free from here on: tile row kc+BT is gathered into it now, a step ahead of its first use, so
        // that the two dependent round trips (element list, then the elements) overlap with the rest of this step.  The border
        // registers are waited for first -- otherwise the in-order memory counter would make their next use wait for the gathers.
        asm volatile("" ::"v"(bacc));
        {
          const int mk = uni(tmask[kc + BT]);
          v4u_t ix[BT + 1];
          ix[BT] = fresh_idx(kc + BT, BT, mk);
#pragma unroll
          for (int b = 0; b < BT; b++) ix[b] = fresh_idx(kc + BT, (kc + BT - b) & (BT - 1), mk);
          araw_n = fresh_data(ix[BT], BT, mk);
#pragma unroll
          for (int b = 0; b < BT; b++) acc[b] = fresh_data(ix[b], (kc + BT - b) & (BT - 1), mk);
        }
      } else {
        lds_double* dst = Aself + ccol * TP + crow;  // raw tile (I, kc) for the next step's TRSM: private slot
#pragma unroll
        for (int q = 0; q < 4; q++) dst[4 * q] = dtile[q];
      }
    }
    if (k >= 0) {
      // ---- D2: rest of the own ring row, tiles (I, J), kc < J <= I: needs X tiles 2..i of this step ----
      if (I < nT && I > kc) {
        WT_BEGIN();
        flags_wait(xflag, 2, i, k + 1);
        WT_END(3);
#pragma unroll
        for (int b = 0; b < BT; b++) {                 // one wave-uniform branch per tile; only the X tiles the row needs are read
          const int J = kc + ((b - kc) & (BT - 1));
          if (J <= I && J != kc) {
            double bb[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) bb[kk] = Xp[(J - k) * TILE_LDS + (4 * kk + crow) * TP + ccol];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[kk], bb[kk], acc[b], 0, 0, 0);
          }
        }
      }
      // ---- border tile of ring column (wave + BOFF), corner ----
      {
        const int Jb = kc + ((wave + BOFF - kc) & (BT - 1));
        WT_BEGIN();
        flag_wait(xflag, k + 1);
        flag_wait(xflag + (Jb - k), k + 1);
        WT_END(4);
        double bn[4], bbord[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          bn[kk] = -Xp[(4 * kk + crow) * TP + ccol];
          bbord[kk] = Xp[(Jb - k) * TILE_LDS + (4 * kk + crow) * TP + ccol];
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) bacc = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[kk], bbord[kk], bacc, 0, 0, 0);
        if (wave == 0) {
#pragma unroll
          for (int kk = 0; kk < 4; kk++) cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[kk], -bn[kk], cacc, 0, 0, 0);
        }
        if (Jb == kc && kc < nT) {                   // border block of column kc for the next TRSM
          lds_double* Abord = AbordB + p3c * SFT_BORDER * TS;
          Abord[crow * TS + ccol] = bacc[0];
          if (crow + 4 < SFT_BORDER) Abord[(crow + 4) * TS + ccol] = bacc[1];
          flag_set(bflag, kc);
        }
      }
      flag_set(dflag + wave, k + 1);                 // done with the buffers of step k
    } else {
      // k == -1: the border block of column 0 comes straight from H
      const int Jb = (wave + BOFF) & (BT - 1);
      if (Jb == 0) {
        lds_double* Abord = AbordB + p3c * SFT_BORDER * TS;
        Abord[crow * TS + ccol] = bacc[0];
        if (crow + 4 < SFT_BORDER) Abord[(crow + 4) * TS + ccol] = bacc[1];
        flag_set(bflag, 0);
      }
    }
  }
#if defined(SFT_STEP_TRACE) && defined(DSH_LAB)
  wt[7] = clock64() - wloop0;
  if (lane == 0 && P.dbg[8] < 0.5) { for (int e = 0; e < 8; e++) P.dbg[16 + 8 * wave + e] = (double)wt[e] + 1.0; }
  __syncthreads();
  if (tid == 0) P.dbg[8] = 1.0;
#endif
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER) Cn[r * 7 + ccol] = cacc[q];
    }
  }
  __syncthreads();
  if (!finish_corner) {   // shared-camera mode: the local Schur complement goes out for the all-reduce
    if (tid < 49) P.Lcorner[tid] = Cn[tid];
    __syncthreads();
    return;
  }
  if (tid == 0) corner_finish(P, ctl, Cn, Dnp);
  __syncthreads();
}

template <int NW>
__device__ __noinline__ void factor_tiles_df(const SftDev& P, Ctl* ctl, double* ws) {
  constexpr int RPW = BT / NW, NT = 64 * NW, BOFF = 2;
  static_assert(RPW == 1 || RPW == 2, "one or two ring rows per wave");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Dn = uni(P.Dn);
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  const int nT = Dnp / TS;
  lds_double* AselfB = to_lds(ws);                             // BT private raw tiles (one per ring slot), operand layout
  lds_double* XpB = to_lds(ws) + BT * TILE_LDS;                // 2 x (BT+1) X tiles, slot 0 = border panel
  lds_double* LinvB = XpB + 2 * (BT + 1) * TILE_LDS;           // 3 x Linv^T (column mod 3: a lagging wave may still read W of step k-1 while W of k+1 is written)
  lds_double* AbordB = LinvB + 3 * TILE_LDS;                   // 3 x 7x16 border block
  lds_double* Cn = AbordB + 3 * SFT_BORDER * TS;               // 7x7 corner
  // flags: [0] W ready (column), [1] border block ready, [2..10] / [12..20] X tile j of an even / odd step ready (step+1),
  // [24..31] wave done (step+1).  The X flags are per buffer parity: a wave that runs one step ahead must not satisfy a
  // reader that still waits for the same tile slot of the previous step.
  lds_int* F = (lds_int*)(Cn + 64);
  lds_int* wflag = F, *bflag = F + 1, *dflag = F + 24;
  const double lambda = ctl->lambda;
  const int crow = lane >> 4, ccol = lane & 15;
  const auto Hbord = uni(P.Hbord);
  const auto Lg = uni(P.Lb);
  const auto Lbord = uni(P.Lbord);
  const auto Linv_g = uni(P.Linv);
  v4d acc[RPW][BT];
  v4d bacc[RPW];
#if defined(SFT_STEP_TRACE) && defined(DSH_LAB)
  long long wt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt0;   // cycles spent waiting: [0] buffers free [1] W [2] X1 [3] X2..i [4] border [5] store flags [6] chol [7] whole loop
  const long long wloop0 = clock64();
#define WT_BEGIN() wt0 = clock64()
#define WT_END(e) wt[e] += clock64() - wt0
#else
#define WT_BEGIN() do {} while (0)
#define WT_END(e) do {} while (0)
#endif
  // Tiles of H are gathered from the compact 3x3 blocks (unconditional loads, exact wait counts); structurally zero tiles (30 % of
  // the C2 band) take the all-zero list row, which stays in cache.  mrow: the mask word of tile row I (P.tmask).
  const auto tmask = uni(P.tmask);
  const auto hgl = uni(P.hgather);
  const auto Hc = uni(P.Hc);
  // two phases so that no wait sits between the tiles of a row: first the byte offsets of every tile (16 bytes per lane; the all-zero
  // list row nT, which stays in cache, for structurally zero tiles), then the elements of the tiles the mask has (wave-uniform branch)
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  auto fresh_idx = [&](int I, int d, int mrow) -> v4u_t {
    return *reinterpret_cast<const SFT_G v4u_t*>(hgl + (((size_t)(((mrow >> d) & 1) ? I : nT) * (BT + 1) + d) * 64 + lane) * 4);
  };
  auto fresh_data = [&](const v4u_t& ix, int d, int mrow) -> v4d {
    v4d v = {0.0, 0.0, 0.0, 0.0};
    if ((mrow >> d) & 1) {
      const auto Hb8 = reinterpret_cast<const SFT_G char*>(Hc);
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const SFT_G double*>(Hb8 + ix[q]);
    }
    return v;
  };
  auto fresh_border = [&](int J) -> v4d {
    v4d v = {0.0, 0.0, 0.0, 0.0};
    v[0] = Hbord[(size_t)crow * Dnp + TS * J + ccol];
    v[1] = Hbord[(size_t)(crow + 4) * Dnp + TS * J + ccol];
    return v;
  };
#pragma unroll
  for (int t = 0; t < RPW; t++) {
    const int a = wave + NW * t;
#pragma unroll
    for (int b = 0; b < BT; b++) acc[t][b] = fresh_data(fresh_idx(a, (a - b) & (BT - 1), uni(tmask[a])), (a - b) & (BT - 1), uni(tmask[a]));
    bacc[t] = fresh_border((a + BOFF) & (BT - 1));
  }
  v4d cacc = {0.0, 0.0, 0.0, 0.0};
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER && ccol <= r) cacc[q] = P.Hcorner[r * 7 + ccol] + ((r == ccol && r < 6) ? lambda : 0.0);
    }
  }
  for (int i = tid; i < 2 * (BT + 1) * TILE_LDS; i += NT) XpB[i] = 0.0;   // rows 7..15 of both border panel tiles stay zero
  if (tid < 48) F[tid] = (tid == 0 || tid == 1) ? -1 : 0;
  if (tid == 0) ctl->fact_ok = 1;
  __syncthreads();

  v4d araw_n = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
  for (int k = -1; k < nT; k++) {
    const int kc = k + 1, kslot = kc & (BT - 1), par = k & 1, p3 = (k + 3) % 3, p3c = kc % 3;
    lds_double* Xp = XpB + par * (BT + 1) * TILE_LDS;
    lds_int* xflag = F + 2 + 10 * par;
    int I[RPW];                                       // ring rows held during this step: window [kc, kc+BT-1]
#pragma unroll
    for (int t = 0; t < RPW; t++) I[t] = kc + ((wave + NW * t - kc) & (BT - 1));
    // the row closer to the diagonal goes first everywhere (it is the one other waves -- and the factorisation -- wait for)
    const bool swap = RPW == 2 && I[RPW - 1] < I[0];
    // the wave that factored column k recycled that ring row with tile row k+BT right behind its Cholesky of the previous step
    // (below): tile (k+BT, k) is raw H and waits in araw_n
    if (k >= 0) {
#pragma unroll
      for (int t = 0; t < RPW; t++) {
        if (((wave + NW * t + BOFF) & (BT - 1)) == (k & (BT - 1))) bacc[t] = fresh_border(k + BT);
      }
      // nobody may still be reading the buffers of step k-2 (same parity)
      WT_BEGIN();
      if (k >= 2) flags_wait(dflag, 0, NW - 1, k - 1);
      WT_END(0);
      // ---- C: X_i = A_i W_k^T for the tiles of the own ring rows ----
      WT_BEGIN();
      flag_wait(wflag, k);
      WT_END(1);
      lds_double* LinvK = LinvB + p3 * TILE_LDS;
      double bv[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) bv[kk] = LinvK[(4 * kk + crow) * TP + ccol];
#pragma unroll
      for (int o = 0; o < RPW; o++)
#pragma unroll
      for (int t = 0; t < RPW; t++)
      if (t == (swap ? RPW - 1 - o : o)) {   // rows in order of their distance to the diagonal, t stays a compile-time index
        const int i = I[t] - k;
        lds_double* Aself = AselfB + (wave + NW * t) * TILE_LDS;
        if (i == BT) {                                // the recycled row: its raw tile arrives from HBM, through the private LDS slot into operand layout
          lds_double* dst = Aself + ccol * TP + crow;
#pragma unroll
          for (int q = 0; q < 4; q++) dst[4 * q] = araw_n[q];
        }
        double av[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) av[kk] = Aself[(4 * kk + crow) * TP + ccol];
        v4d x = {0.0, 0.0, 0.0, 0.0}, x2 = x;
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], x, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], x2, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], x, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], x2, 0, 0, 0);
        x += x2;
        lds_double* dst = Xp + i * TILE_LDS + ccol * TP + crow;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[4 * q] = x[q];
        flag_set(xflag + i, k + 1);
        // the tile of L leaves from the registers it was computed in (accumulator order = storage order): nobody re-reads the panel
        if (I[t] < nT) *reinterpret_cast<v4d*>(Lg + tile_off(k, i) + 4 * lane) = x;   // L is stored by block COLUMN: tile (k+i, k) at slot (k, i)
      }
      bool border_trsm = false;
#pragma unroll
      for (int t = 0; t < RPW; t++) border_trsm = border_trsm || (I[t] - k == 4);
      if (border_trsm) {                              // border panel (7 camera/rhs rows): one more MFMA tile, off the critical path
        WT_BEGIN();
        flag_wait(bflag, k);
        WT_END(4);
        lds_double* Abord = AbordB + p3 * SFT_BORDER * TS;
        double ab[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) ab[kk] = (ccol < SFT_BORDER) ? Abord[ccol * TS + 4 * kk + crow] : 0.0;
        v4d xb = {0.0, 0.0, 0.0, 0.0}, xb2 = xb;
        xb = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[0], bv[0], xb, 0, 0, 0);
        xb2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[1], bv[1], xb2, 0, 0, 0);
        xb = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[2], bv[2], xb, 0, 0, 0);
        xb2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[3], bv[3], xb2, 0, 0, 0);
        xb += xb2;
        Xp[ccol * TP + crow] = xb[0];
        if (crow + 4 < SFT_BORDER) Xp[ccol * TP + crow + 4] = xb[1];
        flag_set(xflag, k + 1);
        Lbord[(size_t)crow * Dnp + TS * k + ccol] = xb[0];
        if (crow + 4 < SFT_BORDER) Lbord[(size_t)(crow + 4) * Dnp + TS * k + ccol] = xb[1];
      }
    }
    // ---- D1: block column kc of the own rows, then publish (private slot) or factor (owner); D2: rest of the row ----
#pragma unroll
    for (int o = 0; o < RPW; o++)
#pragma unroll
    for (int t = 0; t < RPW; t++)
    if (t == (swap ? RPW - 1 - o : o)) {
      const int i = I[t] - k;
      double an[4] = {0.0, 0.0, 0.0, 0.0};
      // the tile of block column kc is picked out of the ring inside the (wave-uniform) branch that updates it: a run-time index into
      // the register array costs 56 v_cndmask per row and step.  k == -1: column 0 sits in slot 0; rows behind the matrix never use it.
      v4d dtile = acc[t][0];
      if (k >= 0) {
        WT_BEGIN();
        flag_wait(xflag + 1, k + 1);                  // X of tile (kc, k); the owner produced it itself
        WT_END(2);
        double bc[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          an[kk] = -Xp[i * TILE_LDS + (4 * kk + crow) * TP + ccol];
          bc[kk] = Xp[TILE_LDS + (4 * kk + crow) * TP + ccol];
        }
        if (I[t] < nT) {
#pragma unroll
          for (int c = 0; c < BT; c++)
            if (c == kslot) {
              v4d side = {0.0, 0.0, 0.0, 0.0};
              acc[t][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[0], bc[0], acc[t][c], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              side = __builtin_amdgcn_mfma_f64_16x16x4f64(an[1], bc[1], side, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              acc[t][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[2], bc[2], acc[t][c], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              side = __builtin_amdgcn_mfma_f64_16x16x4f64(an[3], bc[3], side, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              acc[t][c] += side;
              dtile = acc[t][c];
            }
        }
      }
      if (kc < nT) {
        if (I[t] == kc) {
          __builtin_amdgcn_s_setprio(3);
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (crow + 4 * q == ccol && TS * kc + ccol < Dn) dtile[q] += lambda;
          v4d w = dtile;
          WT_BEGIN();
          const bool ok = chol_inv_blocked(dtile, w);
          WT_END(6);
          if (!ok && lane == 0) ctl->fact_ok = 0;
          lds_double* dst = LinvB + p3c * TILE_LDS + ccol * TP + crow;
#pragma unroll
          for (int q = 0; q < 4; q++) dst[4 * q] = w[q];
          flag_set(wflag, kc);
          *reinterpret_cast<v4d*>(Linv_g + (size_t)kc * TS * TS + 4 * lane) = w;
          __builtin_amdgcn_s_setprio(0);
          // The ring row of column kc is free from here on: tile row kc+BT is gathered into it now, a step ahead of its first use, so
          // that the two dependent round trips (element list, then the elements) overlap with the rest of this step.  The border
          // registers are waited for first -- otherwise the in-order memory counter would make their next use wait for the gathers.
#pragma unroll
          for (int u = 0; u < RPW; u++) asm volatile("" ::"v"(bacc[u]));
          {
            const int mk = uni(tmask[kc + BT]);
            v4u_t ix[BT + 1];
            ix[BT] = fresh_idx(kc + BT, BT, mk);
#pragma unroll
            for (int b = 0; b < BT; b++) ix[b] = fresh_idx(kc + BT, (kc + BT - b) & (BT - 1), mk);
            araw_n = fresh_data(ix[BT], BT, mk);
#pragma unroll
            for (int b = 0; b < BT; b++) acc[t][b] = fresh_data(ix[b], (kc + BT - b) & (BT - 1), mk);
          }
        } else {
          lds_double* dst = AselfB + (wave + NW * t) * TILE_LDS + ccol * TP + crow;   // raw tile (I, kc) for the next step's TRSM: private slot
#pragma unroll
          for (int q = 0; q < 4; q++) dst[4 * q] = dtile[q];
        }
      }
      if (k >= 0 && I[t] < nT && I[t] > kc) {         // tiles (I, J), kc < J <= I: need X tiles 2..i of this step
        WT_BEGIN();
        flags_wait(xflag, 2, i, k + 1);
        WT_END(3);
#pragma unroll
        for (int b = 0; b < BT; b++) {                // one wave-uniform branch per tile; only the X tiles the row needs are read
          const int J = kc + ((b - kc) & (BT - 1));
          if (J <= I[t] && J != kc) {
            double bb[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) bb[kk] = Xp[(J - k) * TILE_LDS + (4 * kk + crow) * TP + ccol];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc[t][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[kk], bb[kk], acc[t][b], 0, 0, 0);
          }
        }
      }
    }
    if (k >= 0) {
      // ---- border tiles of ring columns (wave + NW t + BOFF), corner ----
      WT_BEGIN();
      flag_wait(xflag, k + 1);
      WT_END(4);
      double bn[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) bn[kk] = -Xp[(4 * kk + crow) * TP + ccol];
#pragma unroll
      for (int t = 0; t < RPW; t++) {
        const int Jb = kc + ((wave + NW * t + BOFF - kc) & (BT - 1));
        WT_BEGIN();
        flag_wait(xflag + (Jb - k), k + 1);
        WT_END(4);
        double bbord[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) bbord[kk] = Xp[(Jb - k) * TILE_LDS + (4 * kk + crow) * TP + ccol];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) bacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[kk], bbord[kk], bacc[t], 0, 0, 0);
        if (Jb == kc && kc < nT) {                    // border block of column kc for the next TRSM
          lds_double* Abord = AbordB + p3c * SFT_BORDER * TS;
          Abord[crow * TS + ccol] = bacc[t][0];
          if (crow + 4 < SFT_BORDER) Abord[(crow + 4) * TS + ccol] = bacc[t][1];
          flag_set(bflag, kc);
        }
      }
      if (wave == 0) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[kk], -bn[kk], cacc, 0, 0, 0);
      }
      flag_set(dflag + wave, k + 1);                  // done with the buffers of step k
    } else {
      // k == -1: the border block of column 0 comes straight from H
#pragma unroll
      for (int t = 0; t < RPW; t++)
        if (((wave + NW * t + BOFF) & (BT - 1)) == 0) {
          lds_double* Abord = AbordB + p3c * SFT_BORDER * TS;
          Abord[crow * TS + ccol] = bacc[t][0];
          if (crow + 4 < SFT_BORDER) Abord[(crow + 4) * TS + ccol] = bacc[t][1];
          flag_set(bflag, 0);
        }
    }
  }
#if defined(SFT_STEP_TRACE) && defined(DSH_LAB)
  wt[7] = clock64() - wloop0;
  if (lane == 0 && P.dbg[8] < 0.5) { for (int e = 0; e < 8; e++) P.dbg[16 + 8 * wave + e] = (double)wt[e] + 1.0; }
  __syncthreads();
  if (tid == 0) P.dbg[8] = 1.0;
#endif
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int r = crow + 4 * q;
      if (r < SFT_BORDER && ccol < SFT_BORDER) Cn[r * 7 + ccol] = cacc[q];
    }
  }
  __syncthreads();
  if (tid == 0) {
    bool bad = false;
    for (int k = 0; k < 6; k++) {
      double d = Cn[k * 7 + k];
      for (int j = 0; j < k; j++) d -= Cn[k * 7 + j] * Cn[k * 7 + j];
      if (!(d > 0.0)) bad = true;
      const double piv = sqrt(d);
      Cn[k * 7 + k] = piv;
      for (int r = k + 1; r < 7; r++) {
        double v = Cn[r * 7 + k];
        for (int j = 0; j < k; j++) v -= Cn[r * 7 + j] * Cn[k * 7 + j];
        Cn[r * 7 + k] = v / piv;
      }
    }
    if (bad) ctl->fact_ok = 0;
    if (ctl->fact_ok)
      for (int k = 5; k >= 0; k--) {
        double v = Cn[6 * 7 + k];
        for (int r = k + 1; r < 6; r++) v -= Cn[r * 7 + k] * P.x[Dnp + r];
        P.x[Dnp + k] = v / Cn[k * 7 + k];
      }
  }
  __syncthreads();
}

// Back substitution in tile mode: x_J = Linv_J^T (y_J - sum_{I>J} X_{I,J}^T x_I - Lcn_J^T x_cam).
// Roles: the product of tile (J+d, J) with x_{J+d} belongs to wave (d-1) mod NW; wave 1 also forms the camera term; wave 0
// finishes the block.  Only the d = 1 product needs the block that was finished last, and it belongs to wave 0 itself -- so the
// chain x_{J+1} -> X_{J+1,J}^T x_{J+1} -> sum -> Linv_J^T -> x_J stays inside wave 0, and everything else of block J (d >= 2, the
// camera term) only needs x_{J+2} and older: the other products run one block AHEAD of wave 0.  One LDS barrier per block, none
// on the chain (the version before: two barriers and a partial-sum hand-over on the chain of every block).
// The L tiles of the next PF blocks are in flight in a register ring (LDS-only barriers keep the loads in flight).
#ifndef SFT_BT_PF
#define SFT_BT_PF 6
#endif
template <int NW>
__device__ __noinline__ void backsub_tiles(const SftDev& P, Ctl* ctl, double* ws) {
  constexpr int RPW = BT / NW;
  if (!ctl->fact_ok) return;   // like g2o, x keeps its previous content when the factorisation failed
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps ring bookkeeping and branches on the scalar unit
  const int Dnp = ((uni(P.Dn) + NB - 1) / NB) * NB;
  const int nT = Dnp / TS;
  lds_double* xw = to_lds(ws);                // ring of BT x-tiles
  lds_double* part = xw + TS * BT;            // 2 x (BT+1) partial vectors, double buffered by block parity
  const int crow = lane >> 4, ccol = lane & 15;
  const double xc = (lane < 6) ? P.x[Dnp + lane] : 0.0;
  double xcr[6];
#pragma unroll
  for (int r = 0; r < 6; r++) xcr[r] = bcast_lane(xc, r);
  const auto Lg = uni(P.Lb);
  const auto Lbord = uni(P.Lbord);
  const auto Linv_g = uni(P.Linv);
  const auto xg = uni(P.x);
  // Slot s (s = nT ... 0): wave 0 finishes block s (its own d = 1 product + the partials the others wrote in slot s+1);
  // every wave writes its d >= 2 products (wave 1: + the camera term) of block s-1.  Ring position j holds block (base - j) and
  // is refilled with block (base - j - PF) as soon as it has been taken: PF blocks of look-ahead all the time.
  // Every load is UNCONDITIONAL (a block outside the matrix is read from its nearest column, a tile behind the matrix from its slot of L --
  // neither is used) and each role has its own loop -- wave 0 (Linv_J, y_J), wave 1 (the six camera rows of the border), the others -- so that
  // a trip has a fixed number of loads and the compiler waits for the oldest ones only: a load under a condition made it wait for ALL loads in
  // flight (vmcnt(0)) once per block, the look-ahead was worth nothing (sft_wide.h: backsub_wide, where this was found).
  auto run = [&](auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;
    struct Pre { v4d t[RPW]; double aux[ROLE == 2 ? 1 : 6]; };
    auto fetch = [&](int J) -> Pre {
      Pre p;
      const int Jc = uni(min(max(J, 0), nT - 1));
#pragma unroll
      for (int t = 0; t < RPW; t++) p.t[t] = *reinterpret_cast<const v4d*>(Lg + tile_off(Jc, wave + 1 + NW * t) + 4 * lane);   // column-major L: the BT tiles of block column J are contiguous (16 KB)
      if constexpr (ROLE == 0) {
        const v4d li = *reinterpret_cast<const v4d*>(Linv_g + (size_t)Jc * TS * TS + 4 * lane);
#pragma unroll
        for (int q = 0; q < 4; q++) p.aux[q] = li[q];
        p.aux[4] = Lbord[(size_t)6 * Dnp + TS * Jc + ccol];
        p.aux[5] = 0.0;
      } else if constexpr (ROLE == 1) {
#pragma unroll
        for (int r = 0; r < 6; r++) p.aux[r] = Lbord[(size_t)r * Dnp + TS * Jc + ccol];
      } else {
        p.aux[0] = 0.0;
      }
      return p;
    };
    // product of the wave's tile t of block J with x_{J+d}: 16 partial sums (replicated in the four 16-lane groups after the shuffles)
    auto product = [&](const Pre& e, int t, int J) -> double {
      const int d = wave + 1 + NW * t, I = J + d;
      double p = 0.0;
      if (I < nT) {
        const lds_double* xi = xw + (I & (BT - 1)) * TS + crow;
#pragma unroll
        for (int q = 0; q < 4; q++) p = fma(e.t[t][q], xi[4 * q], p);
        p = sum_rows(p);
      }
      return p;
    };
    constexpr int PF = SFT_BT_PF;
    Pre ring[PF];
#pragma unroll
    for (int j = 0; j < PF; j++) ring[j] = fetch(nT - j);
#pragma unroll 1
    for (int base = nT; base >= 0; base -= PF) {
#pragma unroll
      for (int j = 0; j < PF; j++) {
        const int s = base - j;
        if (s < 0) break;
        const Pre cur = ring[j];            // block s
        ring[j] = fetch(s - PF);
        const Pre& nxt = ring[(j + 1) % PF];   // block s - 1 (position 0 was refilled with block base - PF at the start of this chunk)
        lds_double* part_s = part + (s & 1) * (BT + 1) * TS;
        lds_double* part_n = part + ((s - 1) & 1) * (BT + 1) * TS;
        // ---- ahead of wave 0: block s-1, tiles at distance >= 2 and the camera term (x_{s+1} and older are final)
        if (s >= 1) {
#pragma unroll
          for (int t = 0; t < RPW; t++) {
            const int d = wave + 1 + NW * t;
            if (d >= 2) { const double p = product(nxt, t, s - 1); if (lane < TS) part_n[d * TS + lane] = p; }
          }
          if (ROLE == 1 && lane < TS) {
            double p = 0.0;
#pragma unroll
            for (int r = 0; r < 6; r++) p = fma(nxt.aux[r], xcr[r], p);
            part_n[lane] = p;
          }
        }
        // ---- wave 0: block s
        if (ROLE == 0 && s < nT) {
          double v = cur.aux[4] - product(cur, 0, s);      // y_s - X_{s+1,s}^T x_{s+1}
          v -= part_s[ccol];                               // camera term
#pragma unroll
          for (int i = 2; i <= BT; i++) v -= part_s[i * TS + ccol];
          // x[c] = sum_r Linv[r][c] v[r]   (v is replicated in every 16-lane group)
          double p = 0.0;
#pragma unroll
          for (int q = 0; q < 4; q++) p = fma(cur.aux[q], __shfl(v, crow + 4 * q, 64), p);
          p = sum_rows(p);
          if (lane < TS) { xw[(s & (BT - 1)) * TS + lane] = p; xg[TS * s + lane] = p; }
        }
        lds_barrier();
      }
    }
  };
  if (wave == 0) run(std::integral_constant<int, 0>{});
  else if (wave == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 2>{});
  __syncthreads();
}

#include "sft_wide.h"
#include "sft_wave.h"

// The register-window factorisation for either launch shape.  (The ONLY call site of factor_tiles_df<NW>: the host pass of hipcc rejects a
// second instantiation request of that template from another function -- "substitution failure" -- so every user goes through here.)
template <int NW>
__device__ __forceinline__ void tile_factor(const SftDev& P, Ctl* ctl, double* panel) {
  if constexpr (NW == 8) factor_tiles_df8(P, ctl, panel, ctl->lambda, true);
  else factor_tiles_df<NW>(P, ctl, panel);
}

// One damping trial at ctl->lambda from the current state: factorisation of H + lambda I, back substitution, state update
// (sparse_optimizer.cpp:477-491), scale = sum_j x_j (lambda x_j + b_j) (optimization_algorithm_levenberg.cpp:166-176) and the robust
// chi2 at the trial state.  The caller has pushed the state and pops it on rejection.
template <int NW>
__device__ __forceinline__ double damping_trial(const SftDev& P, Ctl* ctl, double* red, double* out, double* panel, int& ok, double& scale, bool prefactored = false) {
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x;
  const int Dn = P.Dn;
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  PH_T0();
  if (P.tile_mode == 2) {
    if constexpr (NW == 8) {   // wide band: always the 512-thread kernel
      if (P.split && prefactored) {
        // two-sided factorisation: the parts were factored (SFT_SPEC_FACTOR) and the system solved (SFT_SPEC_SOLVE: P.x holds the update
        // in the natural ordering) by the two workgroups of the lane; what they report is all this launch needs
        if (tid == 0) ctl->fact_ok = (P.part[0].x[TS * P.part[0].nT + 7] == 1.0 && P.part[1].x[TS * P.part[1].nT + 7] == 1.0) ? 1 : 0;
        __syncthreads();
        PH_ADD(6);
      } else {
        factor_wide(P, -1, ctl, panel);
        PH_ADD(5);
        backsub_wide(P, -1, ctl, panel);
        PH_ADD(6);
      }
    }
  } else if (P.tile_mode) {
#ifdef DSH_LAB
    if (!(P.mode & 2)) factor_tiles<NW>(P, ctl, panel);
    else
#endif
    {
      PH_RESET();
      tile_factor<NW>(P, ctl, panel);
      PH_ADD(5);
    }
    PH_RESET();
    backsub_tiles<NW>(P, ctl, panel);
    PH_ADD(6);
  } else {
    if constexpr (NW == 8) factor_and_solve(P, ctl, panel, red);   // band mode always runs the 512-thread kernel
  }
  PH_RESET();
  ok = ctl->fact_ok;
  // update
  for (int i = tid; i < 3 * P.n; i += NT) {
    const int a = P.act[i / 3];
    if (a >= 0) P.xyz[i] += P.x[3 * a + (i % 3)];
  }
  if (tid == 0) pose_oplus(P.pose, P.x + Dnp);
  // scale = sum_j x_j (lambda x_j + b_j)
  double sc = 0.0;
  const double lam = ctl->lambda;
  for (int r = tid; r < Dn; r += NT) { const double xv = P.x[r]; sc += xv * (lam * xv + P.Hbord[(size_t)6 * Dnp + r]); }
  if (tid < 6) { const double xv = P.x[Dnp + tid]; sc += xv * (lam * xv + P.Hcorner[42 + tid]); }
  __syncthreads();
  block_sum<1>(&sc, red, out);
  scale = out[0];
  __syncthreads();
  PH_ADD(7);
  const double chi_new = eval_edges<false, 0>(P, ctl, red, out, asm_records<NW, 0>(P, panel));
  PH_ADD(1);
  return chi_new;
}

// ------------------------------------------------------------------------------------------
// The persistent per-problem kernel
// ------------------------------------------------------------------------------------------
// Initial state of a run (restored from the uploaded frame) and the zeroed system with its identity padding.
template <int NT>
__device__ __forceinline__ void init_state(const SftDev& P) {
  const int tid = threadIdx.x;
  const int Dn = P.Dn, ldh = P.ldh, kd = P.kd;
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  for (int i = tid; i < 3 * P.n; i += NT) P.xyz[i] = P.xyz_init[i];
  if (tid < 7) P.pose[tid] = P.pose_init[tid];
  if (P.tile_mode == 1) {
    // compact blocks: the assembly rewrites every block of every linearisation; behind them the 0.0 and the 1.0 the gather lists point
    // structurally empty elements and the identity padding at
    if (tid == 0) { const size_t z = 9 * (size_t)(P.nA + P.noff); P.Hc[z] = 0.0; P.Hc[z + 1] = 1.0; }
  } else if (P.tile_mode == 2 && P.split) {
    // two-sided factorisation: the band matrices of the two parts (the assembly writes the structural non-zeros of every linearisation),
    // identity on the padding scalars in front of part 1
    for (int g = 0; g < 2; g++) {
      const auto H = P.part[g].Hb;
      const size_t nel = (size_t)P.part[g].nT * P.part[g].tpr * TS * TS;
      for (size_t i = tid; i < nel; i += NT) H[i] = 0.0;
    }
    __syncthreads();
    for (int j = tid; j < P.sp_pad; j += NT) P.part[1].Hb[wide_elem(P.part[1].tpr, j, j)] = 1.0;
  } else if (P.tile_mode) {
    // zero tiles + identity padding
    const int tpr = P.tpr;
    const size_t nel = (size_t)(Dnp / TS + SFT_H_PAD_TILE_ROWS) * tpr * TS * TS;   // incl. the zero tile rows below the matrix
    for (size_t i = tid; i < nel; i += NT) {
      const int e = (int)(i % (TS * TS)), td = (int)((i / (TS * TS)) % tpr), I = (int)(i / ((size_t)tpr * TS * TS));
      const int el = e >> 2, erow = (el >> 4) + 4 * (e & 3), ecol = el & 15;   // native tile order: lane, register
      const bool pad_diag = td == 0 && erow == ecol && TS * I + ecol >= Dn && I < Dnp / TS;
      P.Hb[i] = pad_diag ? 1.0 : 0.0;
    }
  } else {
    for (size_t i = tid; i < (size_t)Dnp * ldh; i += NT) {
      const int k = (int)(i % ldh), r = (int)(i / ldh);
      P.Hb[i] = (k == kd && r >= Dn) ? 1.0 : 0.0;
    }
  }
  for (size_t i = tid; i < (size_t)(SFT_BORDER + 1) * Dnp + SFT_H_PAD_BORDER; i += NT) P.Hbord[i] = 0.0;   // 8th row + padding stay zero
  for (int i = tid; i < Dnp + 6; i += NT) P.x[i] = 0.0;
  // (max_iters = 0: no evaluation ever writes them, and the classification reads them -- like the oracle's edges that never computed an error)
  for (int m = tid; m < P.M; m += NT) P.chi2_obs[m] = 0.0;
  if (tid == 0) {
    P.res->iters = 0; P.res->trials = 0; P.res->status = 0; P.res->inliers = 0;
    for (int i = 0; i < 96; i++) P.dbg[i] = 0.0;
  }
}

// One linearisation: residuals + assembly records, then the normal equations.  The records alias the solver workspace in LDS (dead
// once H is assembled); their placement class is a template parameter (AsmRec).  (part, nparts): see assemble.
template <int NW, class F, bool WITH_CLASS3 = false>
__device__ __forceinline__ double linearise(const SftDev& P, Ctl* ctl, double* red, double* out, double* panel, F ph_residuals, int part = 0, int nparts = 1) {
  double chi = 0.0;
  if constexpr (WITH_CLASS3) {   // (only the LIN kernel of the phase rounds is compiled with it: the host sets class 3 for that launch shape alone)
    if (P.lds_class == 3) {
      const auto jp = asm_records<NW, 3>(P, panel); chi = eval_edges<true, 3>(P, ctl, red, out, jp); ph_residuals(); assemble<NW, 3>(P, red, out, jp, part, nparts);
      return chi;
    }
  }
  switch (P.lds_class) {
    case 2: { const auto jp = asm_records<NW, 2>(P, panel); chi = eval_edges<true, 2>(P, ctl, red, out, jp); ph_residuals(); assemble<NW, 2>(P, red, out, jp, part, nparts); break; }
    case 1: { const auto jp = asm_records<NW, 1>(P, panel); chi = eval_edges<true, 1>(P, ctl, red, out, jp); ph_residuals(); assemble<NW, 1>(P, red, out, jp, part, nparts); break; }
    default: { const auto jp = asm_records<NW, 0>(P, panel); chi = eval_edges<true, 0>(P, ctl, red, out, jp); ph_residuals(); assemble<NW, 0>(P, red, out, jp, part, nparts); break; }
  }
  return chi;
}

// Classification and statistics (DefOptimizer.cc:515-559), map-point write-back (DefMapPoint.cc:129-147), counters.
//   outlier[m] = (float)chi2 > 5.991 with the chi2 of the observation's LAST evaluation (stale when the last damping
//   trial was rejected: the reference reads e->chi2() without recomputing the error of inliers);
//   repError = sum over the inliers, in index order, of the reprojection error norm at the final estimate, / count.
template <int NT>
__device__ __forceinline__ void classify(const SftDev& P, Ctl* ctl, double* panel, int iters, int total_trials) {
  const int tid = threadIdx.x;
  if (tid == 0) { quat_to_R(P.pose + 3, ctl->R); ctl->t[0] = P.pose[0]; ctl->t[1] = P.pose[1]; ctl->t[2] = P.pose[2]; ctl->nbad = 0; ctl->chi_tmp = 0.0; }
  __syncthreads();
  {
    constexpr int CHK = 2048;                 // observations per pass: their error norms wait in LDS for the ordered sum
    lds_double* ers = to_lds(panel);
    int nbad = 0;
    for (int m0 = 0; m0 < P.M; m0 += CHK) {
      const int mend = min(P.M, m0 + CHK);
      for (int m = m0 + tid; m < mend; m += NT) {
        const int n0 = P.obs_nodes[3 * m], n1 = P.obs_nodes[3 * m + 1], n2 = P.obs_nodes[3 * m + 2];
        const double b0 = P.obs_bary[3 * m], b1 = P.obs_bary[3 * m + 1], b2 = P.obs_bary[3 * m + 2];
        double pw[3], pc[3];
        for (int k = 0; k < 3; k++) pw[k] = (b0 * P.xyz[3 * n0 + k] + b1 * P.xyz[3 * n1 + k]) + b2 * P.xyz[3 * n2 + k];
        for (int k = 0; k < 3; k++) pc[k] = (ctl->R[3 * k] * pw[0] + ctl->R[3 * k + 1] * pw[1] + ctl->R[3 * k + 2] * pw[2]) + ctl->t[k];
        const double e0 = P.obs_uv[2 * m] - ((pc[0] / pc[2]) * P.fx + P.cx);
        const double e1 = P.obs_uv[2 * m + 1] - ((pc[1] / pc[2]) * P.fy + P.cy);
        const double er = sqrt(e0 * e0 + e1 * e1);
        const bool bad = (double)(float)P.chi2_obs[m] > 5.991;
        P.outlier[m] = bad ? 1 : 0;
        nbad += bad ? 1 : 0;
        ers[m - m0] = bad ? -1.0 : er;          // error norms are >= 0 (or NaN): a negative entry marks an outlier
        // float32 world position, plain IEEE products and sums like the host code of the reference (no contraction)
        for (int k = 0; k < 3; k++)
          P.mappoint[3 * m + k] = (float)__dadd_rn(__dadd_rn(__dmul_rn(b0, P.xyz[3 * n0 + k]), __dmul_rn(b1, P.xyz[3 * n1 + k])), __dmul_rn(b2, P.xyz[3 * n2 + k]));
      }
      __syncthreads();
      if (tid == 0) {
        double sum = ctl->chi_tmp;
        const int cnt = mend - m0;
#pragma unroll 8
        for (int i = 0; i < cnt; i++) { const double v = ers[i]; if (!(v < 0.0)) sum += v; }
        ctl->chi_tmp = sum;
      }
      __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nbad += __shfl_down(nbad, off, 64);
    if ((tid & 63) == 0 && nbad) atomicAdd(&ctl->nbad, nbad);   // integer: order independent
    __syncthreads();
  }
  if (tid == 0) {
    const int inl = P.M - ctl->nbad;
    P.res->iters = iters; P.res->trials = total_trials; P.res->inliers = inl;
    P.res->rep_error = ctl->chi_tmp / (double)(unsigned)inl;
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW, SFT_WAVES_PER_EU) void sft_lm_kernel(const SftDev* __restrict__ probs) {
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SftDev& P = probs[blockIdx.x];
  Ctl* ctl = reinterpret_cast<Ctl*>(smem);
  double* red = reinterpret_cast<double*>(smem + 512);   // 16*27 doubles
  double* out = red + 16 * 27 + 5;                        // 27 doubles
  double* panel = out + 32;
  const int tid = threadIdx.x;
  const int Dn = P.Dn;

  init_state<NT>(P);
  if (tid == 0) { ctl->lambda = -1.0; ctl->ni = 2.0; ctl->nbad = 0; ctl->stop = 0; ctl->it = 0; }
  __syncthreads();
  PH_T0();

#ifdef DSH_LAB
  if (P.mode == 1) {  // lab hook (dsh_lab_sft_system): one assembly at the initial state
    const double chi = linearise<NW>(P, ctl, red, out, panel, [] {});
    if (tid == 0) P.dbg[0] = chi;
    return;
  }
#endif

  int total_trials = 0, iters = 0;
  for (int it = 0; it < P.max_iters; it++) {
    const double chi0 = linearise<NW>(P, ctl, red, out, panel, [&] { PH_ADD(1); });
    PH_ADD(2);
    if (it == 0) {
      double mx = 0.0;
      for (int r = tid; r < Dn; r += NT) mx = fmax(mx, fabs(h_diag(P, r)));
      if (tid < 6) mx = fmax(mx, fabs(P.Hcorner[tid * 8]));
      mx = block_max(mx, red);
      if (tid == 0) { ctl->lambda = 1e-5 * mx; ctl->ni = 2.0; ctl->nbad = 0; }
    }
    if (tid == 0) { ctl->chi_cur = chi0; ctl->chi_ini = chi0; ctl->qmax = 0; ctl->rho = 0.0; ctl->accepted = 0; }
    __syncthreads();
    const double lambda_start = ctl->lambda;
    int all_ok = 1;
    bool again;
    do {
      // push
      for (int i = tid; i < 3 * P.n; i += NT) P.xyz_bak[i] = P.xyz[i];
      double pose_bak = (tid < 7) ? P.pose[tid] : 0.0;
      PH_ADD(7);
      int ok;
      double scale;
      const double chi_new = damping_trial<NW>(P, ctl, red, out, panel, ok, scale);
      PH_RESET();
      all_ok &= ok;
      if (tid == 0) {
        double tempChi = ok ? chi_new : DBL_MAX;
        double rho = (ctl->chi_cur - tempChi);
        rho /= (scale + 1e-3);
        ctl->rho = rho;
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - pow((2 * rho - 1), 3);
          alpha = fmin(alpha, 2. / 3.);
          const double sf = fmax(1. / 3., alpha);
          ctl->lambda *= sf; ctl->ni = 2.0; ctl->chi_cur = tempChi; ctl->accepted = 1;
          ctl->stop = 0;
        } else {
          ctl->lambda *= ctl->ni; ctl->ni *= 2.0;
          ctl->stop = 1;  // reused as "restore" flag below
        }
        ctl->qmax++;
      }
      __syncthreads();
      if (ctl->stop) {  // pop
        for (int i = tid; i < 3 * P.n; i += NT) P.xyz[i] = P.xyz_bak[i];
        if (tid < 7) P.pose[tid] = pose_bak;
      }
      again = (ctl->rho < 0) && (ctl->qmax < 10);
      __syncthreads();
      PH_ADD(7);
    } while (again);
    total_trials += ctl->qmax;
    iters++;
    if (tid == 0 && P.trace) {
      double* t = P.trace + it * 8;
      t[0] = ctl->chi_ini; t[1] = lambda_start; t[2] = ctl->qmax; t[3] = ctl->chi_cur; t[4] = ctl->lambda; t[5] = ctl->rho;
      t[6] = ctl->accepted; t[7] = all_ok;
    }
    if (tid == 0 && !all_ok) P.res->status |= 1;
    bool term = (ctl->qmax == 10) || (ctl->rho == 0);
    if (!term) {
      if (tid == 0) {
        if ((ctl->chi_ini - ctl->chi_cur) * 1e3 < ctl->chi_ini) ctl->nbad++; else ctl->nbad = 0;
      }
      __syncthreads();
      term = ctl->nbad >= 3;
    }
    __syncthreads();
    if (term) break;
  }
  classify<NT>(P, ctl, panel, iters, total_trials);
}

// ------------------------------------------------------------------------------------------
// Latency mode: speculative damping trials.  One tracked frame is ONE problem and 255 of the 256 CUs idle; a Levenberg-Marquardt
// iteration, however, usually needs two or three damping trials (after an accepted step the damping shrinks to a third, the next
// first trial is too bold, is rejected, and lambda nu is accepted: golden traces 1,3,1,3,2,2,2...), and the dampings of the
// rejection chain are known in advance: lambda, lambda nu, lambda nu (2 nu), ...  K workgroups ("lanes", on K CUs) linearise the
// same state redundantly -- identical arithmetic, so identical H -- and lane j runs trial j of the chain; the next launch reads the
// K results and replays the controller in trial order: the first accepted trial wins, exactly the sequence the one-workgroup
// kernel would have walked through, at one trial per iteration instead of two.  The kernel boundary is the only synchronisation
// (no device-scope fences, no flags); results, state and decisions are bit-identical to sft_lm_kernel.
// ------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, SFT_WAVES_PER_EU) void sft_spec_kernel(const SftDev* __restrict__ probs, SftSpec* __restrict__ specs, int K, int phase, int nh) {
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // SFT_SPEC_FACTOR / SFT_SPEC_SOLVE run two workgroups per (problem, lane): one per part of the two-sided factorisation; a FACTOR launch
  // carries nh helper workgroups per part behind them (role 1 .. nh, role-major: an owner and its helpers are a multiple of 8 workgroups
  // apart -- the same XCD under the round-robin dispatch, though nothing depends on it)
  const int roles = phase == SFT_SPEC_FACTOR ? 1 + nh : 1, per_role = gridDim.x / roles;
  const int role = blockIdx.x / per_role, bid = blockIdx.x % per_role;
  const int wgs = (phase == SFT_SPEC_FACTOR || phase == SFT_SPEC_SOLVE) ? 2 : 1, fpart = bid % wgs;
  const int bx = bid / wgs;
  const int B = per_role / (K * wgs), b = bx / K, j = bx % K;   // tables are lane-major: entry (lane j, problem b) at j * B + b
  const SftDev& P = probs[(size_t)j * B + b];
  SftSpec& S = specs[(size_t)j * B + b];
  auto peer = [&](int jj) -> const SftSpec& { return specs[(size_t)jj * B + b]; };
  Ctl* ctl = reinterpret_cast<Ctl*>(smem);
  double* red = reinterpret_cast<double*>(smem + 512);
  double* out = red + 16 * 27 + 5;
  double* panel = out + 32;
  const int tid = threadIdx.x;
  const int Dn = P.Dn;
  // Three kinds of launches (the kernel boundary is the only synchronisation between the lanes of a problem):
  //   SFT_SPEC_INIT   state and workspace (once);
  //   SFT_SPEC_LIN    the controller over the trials of the last round, then -- on a new iteration -- the linearisation: every
  //                   lane evaluates all edges (the records the gathers need, the same chi2 everywhere) and assembles ITS share
  //                   of the block rows into the H the lanes of a tile-mode-1 problem share (the other storage modes: each its own H);
  //   SFT_SPEC_TRIAL  the lane's trial of the rejection chain on the complete H.
  if (S.done) return;
  const int L = S.launches, par = L & 1, prev = par ^ 1;   // L: completed trial rounds
  if (phase == SFT_SPEC_INIT) {
    init_state<NT>(P);
    if (tid == 0) { S.lambda = -1.0; S.ni = 2.0; S.nbad = 0; S.it = 0; S.qbase = 0; S.iters = 0; S.trials = 0; S.need_lin = 1; S.last_lane = 0; S.all_ok = 1; S.pad = 0; }
    return;
  }
  if (phase == SFT_SPEC_LIN && S.pad) {   // S.pad: a trial round is waiting for its verdict
    // ---- the controller over the K trials of the previous launch, in trial order (optimization_algorithm_levenberg.cpp:102-164)
    if (tid == 0) {
      double lam = S.lambda, ni = S.ni, chi_cur = S.chi_cur, rho = 0.0;
      int q = S.qbase, accepted = 0, winner = -1, ended = 0, last = 0, all_ok = S.all_ok;
      for (int jj = 0; jj < K && !ended; jj++) {
        const SftSpecRes& R = peer(jj).res[prev];
        if (!R.valid) break;
        q++;
        last = jj;
        all_ok &= R.ok;
        const double tempChi = R.ok ? R.chi_new : DBL_MAX;
        rho = (chi_cur - tempChi);
        rho /= (R.scale + 1e-3);
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - pow((2 * rho - 1), 3);
          alpha = fmin(alpha, 2. / 3.);
          const double sf = fmax(1. / 3., alpha);
          lam = R.lambda * sf; ni = 2.0; chi_cur = tempChi; accepted = 1; winner = jj; ended = 1;
        } else {
          lam = R.lambda * R.ni; ni = R.ni * 2.0;
          if (!(rho < 0) || q >= 10) ended = 1;
        }
      }
      S.lambda = lam; S.ni = ni; S.chi_cur = chi_cur; S.rho = rho; S.all_ok = all_ok; S.last_lane = last;
      if (accepted) S.accepted = 1;
      ctl->accepted = accepted; ctl->qmax = q; ctl->it = winner; ctl->stop = ended;
    }
    __syncthreads();
    const int accepted = ctl->accepted, qmax = ctl->qmax, winner = ctl->it, ended = ctl->stop;
    if (accepted) {   // every lane continues from the winner's state
      const auto src = probs[(size_t)winner * B + b].spec_xyz[prev];
      for (int i = tid; i < 3 * P.n; i += NT) P.xyz[i] = src[i];
      if (tid < 7) P.pose[tid] = peer(winner).res[prev].pose[tid];
    } else {          // pop the lane's own trial
      for (int i = tid; i < 3 * P.n; i += NT) P.xyz[i] = P.xyz_bak[i];
      if (tid < 7) P.pose[tid] = S.pose_bak[tid];
    }
    __syncthreads();
    if (ended) {      // the outer iteration is over (sparse_optimizer.cpp:403-475 + the reference's stop rule)
      if (tid == 0) {
        S.trials += qmax;
        S.iters++;
        if (j == 0 && P.trace) {
          double* t = P.trace + S.it * 8;
          t[0] = S.chi_ini; t[1] = S.lambda_start; t[2] = qmax; t[3] = S.chi_cur; t[4] = S.lambda; t[5] = S.rho; t[6] = S.accepted; t[7] = S.all_ok;
        }
        if (j == 0 && !S.all_ok) P.res->status |= 1;
        bool term = (qmax == 10) || (S.rho == 0);
        if (!term) {
          if ((S.chi_ini - S.chi_cur) * 1e3 < S.chi_ini) S.nbad++; else S.nbad = 0;
          term = S.nbad >= 3;
        }
        S.it++;
        if (S.it >= P.max_iters) term = true;
        S.need_lin = 1; S.qbase = 0;
        ctl->nbad = term ? 1 : 0;
      }
      __syncthreads();
      if (ctl->nbad) {   // finished: lane 0 owns the results
        if (j == 0) {
          const int last = S.last_lane;
          if (last != 0) {   // the errors of the LAST evaluated trial are what the classification reads (DefOptimizer.cc:515-537)
            const auto src = probs[(size_t)last * B + b].chi2_obs;
            for (int m = tid; m < P.M; m += NT) P.chi2_obs[m] = src[m];
            __syncthreads();
          }
          classify<NT>(P, ctl, panel, S.iters, S.trials);
        }
        __syncthreads();
        if (tid == 0) S.done = 1;
        return;
      }
    } else if (tid == 0) {
      S.qbase = qmax; S.need_lin = 0;
    }
    if (tid == 0) S.pad = 0;
    __syncthreads();
  }
  if (phase == SFT_SPEC_LIN) {
    if (S.need_lin == 1) {   // a new iteration: linearise (this lane's share of the assembly)
      const bool shared_h = P.tile_mode == 1 || (P.tile_mode == 2 && P.split);
      const double chi0 = linearise<NW>(P, ctl, red, out, panel, [] {}, shared_h ? j : 0, shared_h ? K : 1);
      if (tid == 0) { S.chi_cur = chi0; S.chi_ini = chi0; S.rho = 0.0; S.accepted = 0; S.all_ok = 1; S.need_lin = 2; }
    }
    return;
  }
  if (phase == SFT_SPEC_FACTOR || phase == SFT_SPEC_SOLVE) {
    // ---- two-sided factorisation of the lane's trial: this workgroup factors part `fpart` at the lane's damping.  The controller state is
    // only READ here (the trial launch behind this one updates it): on a fresh linearisation of the first iteration the initial damping is
    // recomputed from H -- the same number the trial launch stores.
    if (!(P.tile_mode == 2 && P.split) || S.need_lin == 1) return;
    if (S.qbase + j >= 10) return;
    const int epoch = S.launches + 1;     // what the cross-workgroup flags of this launch hold (the sync words are cleared once per run)
    if (role > 0) {   // a helper of the part: the far products of its block columns (sft_wide.h)
      factor_wide_helper(P, fpart, role - 1, nh, epoch, ctl, panel);
      return;
    }
    double lam = S.lambda, ni = S.ni;
    if (S.need_lin == 2 && S.it == 0) {
      double mx = 0.0;
      for (int r = tid; r < Dn; r += NT) mx = fmax(mx, fabs(h_diag(P, r)));
      if (tid < 6) mx = fmax(mx, fabs(P.Hcorner[tid * 8]));
      mx = block_max(mx, red);
      lam = 1e-5 * mx; ni = 2.0;
    }
    for (int t = 0; t < j; t++) { lam *= ni; ni *= 2.0; }
    if (tid == 0) ctl->lambda = lam;
    __syncthreads();
    if (phase == SFT_SPEC_FACTOR) {
      // with helpers (and for any band that fits the near window whole): the owner that works from registers and LDS
      if (nh > 0 || P.part[fpart].wbt <= SFT_WIDE_NEAR) factor_part<SFT_WIDE_NEAR>(P, fpart, ctl, panel, epoch, nh);
      else factor_wide(P, fpart, ctl, panel);
      return;
    }
    // ---- SFT_SPEC_SOLVE: the Schur contributions of the two parts are summed into the reduced (separator + camera) problem (which adds the
    // damping of the separator's diagonal); BOTH workgroups of the lane solve it, each in its own workspace (part[2 + fpart]), and then
    // back-substitute their own part and scatter it into the natural ordering
    const int red = 2 + fpart;
    const int xl = P.sp_xl;
    const auto x0 = P.part[0].xchg, x1 = P.part[1].xchg, xs = P.part[red].xchg;
#ifdef SFT_SOLVE_TRACE
    long long st_t[6]; st_t[0] = wall_clock64(); const long long st_c0 = clock64();
#define ST_MARK(i) st_t[i] = wall_clock64()
#else
#define ST_MARK(i) do {} while (0)
#endif
    {   // (four independent pairs of loads per thread and trip: the loop is bound by the latency of its loads)
      int i = tid;
      for (; i + 3 * NT < xl; i += 4 * NT) {
        const double a0 = x0[i], a1 = x0[i + NT], a2 = x0[i + 2 * NT], a3 = x0[i + 3 * NT];
        const double b0 = x1[i], b1 = x1[i + NT], b2 = x1[i + 2 * NT], b3 = x1[i + 3 * NT];
        xs[i] = a0 + b0; xs[i + NT] = a1 + b1; xs[i + 2 * NT] = a2 + b2; xs[i + 3 * NT] = a3 + b3;
      }
      for (; i < xl; i += NT) xs[i] = x0[i] + x1[i];
    }
    __syncthreads();
    ST_MARK(1);
    const int nTr = P.part[2].nT;
    const bool parts_ok = xs[(size_t)nTr * P.part[2].tpr * (TS * TS) + (size_t)8 * TS * nTr + 56] == 0.0;
    __syncthreads();
    if (parts_ok) factor_wide(P, red, ctl, panel);
    else if (tid == 0) ctl->fact_ok = 0;
    __syncthreads();
    ST_MARK(2);
    backsub_wide(P, red, ctl, panel);
    ST_MARK(3);
    backsub_wide(P, fpart, ctl, panel, red);
    ST_MARK(4);
#ifdef SFT_SOLVE_TRACE
    if (tid == 0 && j == 0) for (int i = 0; i < 4; i++) P.dbg[96 + 8 * fpart + i] = (double)(st_t[i + 1] - st_t[i]);
    if (tid == 0 && j == 0) P.dbg[96 + 8 * fpart + 4] = (double)(clock64() - st_c0) / (double)(st_t[4] - st_t[0]);   // shader clocks per 10 ns
#endif
    const int okf = ctl->fact_ok;
    if (okf) {   // this workgroup's share of the solution in the natural ordering
      const int c0 = P.sp_c0, sp = P.sp_s, pad = P.sp_pad, n1p = P.sp_n1p;
      const int Dnp = ((Dn + NB - 1) / NB) * NB;
      const auto xp = P.part[fpart].x, xr = P.part[red].x;
      if (fpart == 0) {
        for (int q = tid; q < c0; q += NT) P.x[q] = xp[q];
        for (int k = tid; k < sp; k += NT) P.x[c0 + k] = xr[k];
        if (tid < 6) P.x[Dnp + tid] = xr[TS * nTr + tid];
      } else {
        for (int q = pad + tid; q < n1p; q += NT) P.x[Dn - 1 - (q - pad)] = xp[q];
      }
    }
    if (tid == 0) P.part[fpart].x[TS * P.part[fpart].nT + 7] = okf ? 1.0 : 0.0;
    return;
  }
  // ---- SFT_SPEC_TRIAL
  if (S.need_lin == 1) return;   // (never: a linearisation launch always precedes)
  if (S.need_lin == 2) {         // first trial round on a fresh linearisation: H is complete now
    if (S.it == 0) {
      double mx = 0.0;
      for (int r = tid; r < Dn; r += NT) mx = fmax(mx, fabs(h_diag(P, r)));
      if (tid < 6) mx = fmax(mx, fabs(P.Hcorner[tid * 8]));
      mx = block_max(mx, red);
      if (tid == 0) { S.lambda = 1e-5 * mx; S.ni = 2.0; S.nbad = 0; }
    }
    __syncthreads();
    if (tid == 0) { S.lambda_start = S.lambda; S.need_lin = 0; }
  }
  __syncthreads();
  double lam = S.lambda, ni = S.ni;
  for (int t = 0; t < j; t++) { lam *= ni; ni *= 2.0; }   // trial j of the chain: the damping after j rejections
  const bool in_range = S.qbase + j < 10;                 // the reference stops after ten trials
  if (in_range) {
    for (int i = tid; i < 3 * P.n; i += NT) P.xyz_bak[i] = P.xyz[i];   // push
    if (tid < 7) S.pose_bak[tid] = P.pose[tid];
    if (tid == 0) ctl->lambda = lam;
    __syncthreads();
    int ok;
    double scale;
    const double chi_new = damping_trial<NW>(P, ctl, red, out, panel, ok, scale, P.tile_mode == 2 && P.split);
    const auto dst = P.spec_xyz[par];
    for (int i = tid; i < 3 * P.n; i += NT) dst[i] = P.xyz[i];
    if (tid == 0) {
      SftSpecRes& R = S.res[par];
      R.chi_new = chi_new; R.scale = scale; R.lambda = lam; R.ni = ni; R.ok = ok; R.valid = 1;
      for (int k = 0; k < 7; k++) R.pose[k] = P.pose[k];
    }
  } else if (tid == 0) {
    S.res[par].valid = 0;
    for (int k = 0; k < 7; k++) S.pose_bak[k] = P.pose[k];
  }
  if (!in_range) for (int i = tid; i < 3 * P.n; i += NT) P.xyz_bak[i] = P.xyz[i];
  if (tid == 0) { S.launches = L + 1; S.pad = 1; }
}

// ------------------------------------------------------------------------------------------
// Shared-camera mode across GPUs (BASELINE.json north star / configs[3]: patches or keyframes sharded over the GPUs of a node,
// "RCCL all-reduce of the shared camera-pose normal equations").  Every rank owns the nodes and observations of its own
// patch; the only coupling is the 6-dof camera.  With the camera last (arrowhead) the local factorisation ends in the
// rank's Schur complement of the camera S_g = H_cc,g - H_cn,g H_nn,g^-1 H_nc,g and its right-hand side: the ranks all-reduce
// those 27 numbers (sum), every rank solves the same 6x6 system and back-substitutes its own nodes -- the exact solution of
// the joint normal equations.  Levenberg-Marquardt control is replicated: it only reads all-reduced scalars, so every rank
// takes the same decisions.  The persistent kernel is cut at the collectives into four phases sequenced by the host
// (dsh_multi.cpp): LIN linearise, FAC factorise, SOL camera solve + back substitution + update + trial evaluation, CTL
// accept / reject.  State between the phases lives in SftSc (global memory).
// ------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, SFT_WAVES_PER_EU) void sft_sc_kernel(const SftDev* __restrict__ probs, SftSc* __restrict__ scs, int phase) {
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SftDev& P = probs[blockIdx.x];
  SftSc& S = scs[blockIdx.x];
  Ctl* ctl = reinterpret_cast<Ctl*>(smem);
  double* red = reinterpret_cast<double*>(smem + 512);
  double* out = red + 16 * 27 + 5;
  double* panel = out + 32;
  const int tid = threadIdx.x;
  const int Dn = P.Dn;
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  if (phase == SFT_SC_LIN) {
    if (S.it == 0) {
      init_state<NT>(P);
      if (tid == 0) { S.lambda = -1.0; S.ni = 2.0; S.nbad = 0; S.iters = 0; S.trials = 0; S.done = 0; }
      __syncthreads();
    }
    const double chi0 = linearise<NW>(P, ctl, red, out, panel, [] {});
    double mx = 0.0;
    for (int r = tid; r < Dn; r += NT) mx = fmax(mx, fabs(h_diag(P, r)));
    mx = block_max(mx, red);
    if (tid == 0) {
      // send: robust chi2, diagonal of the local H_cc, local b_c, the largest node diagonal in this rank's slot (a sum over
      // the ranks then carries every rank's maximum: no second collective for the max of computeLambdaInit)
      for (int i = 0; i < SFT_SC_XCHG; i++) S.send[i] = 0.0;
      S.send[0] = chi0;
      for (int k = 0; k < 6; k++) { S.send[1 + k] = P.Hcorner[k * 8]; S.send[7 + k] = P.Hcorner[42 + k]; }
      S.send[13 + S.rank] = mx;
      S.qmax = 0; S.rho = 0.0; S.accepted = 0; S.all_ok = 1;
    }
  } else if (phase == SFT_SC_FAC) {
    if (tid == 0 && S.qmax == 0) {   // first trial of the iteration: take over the all-reduced linearisation
      S.chi_cur = S.chi_ini = S.recv[0];
      for (int k = 0; k < 6; k++) S.bc[k] = S.recv[7 + k];
      if (S.it == 0) {
        double mxa = 0.0;
        for (int k = 0; k < 6; k++) mxa = fmax(mxa, fabs(S.recv[1 + k]));
        for (int r = 0; r < S.nranks; r++) mxa = fmax(mxa, S.recv[13 + r]);
        S.lambda = 1e-5 * mxa; S.ni = 2.0; S.nbad = 0;
      }
      S.lambda_start = S.lambda;
    }
    __syncthreads();
    // push
    for (int i = tid; i < 3 * P.n; i += NT) P.xyz_bak[i] = P.xyz[i];
    if (tid < 7) S.pose_bak[tid] = P.pose[tid];
    if (tid == 0) ctl->lambda = S.lambda;
    __syncthreads();
    if constexpr (NW == 8) factor_tiles_df8(P, ctl, panel, S.rank == 0 ? S.lambda : 0.0, false);
    if (tid == 0) {
      for (int i = 0; i < SFT_SC_XCHG; i++) S.send[i] = 0.0;
      int q = 0;
      for (int r = 0; r < 7; r++)
        for (int c = 0; c <= r && c < 6; c++) S.send[q++] = P.Lcorner[r * 7 + c];   // 21 + 6 entries: lower 6x6, then the right-hand side row
      S.send[27] = ctl->fact_ok ? 0.0 : 1.0;
    }
  } else if (phase == SFT_SC_SOL) {
    lds_double* Cn = to_lds(panel);
    if (tid == 0) {
      int q = 0;
      for (int r = 0; r < 7; r++)
        for (int c = 0; c <= r && c < 6; c++) Cn[r * 7 + c] = S.recv[q++];
      ctl->fact_ok = S.recv[27] == 0.0 ? 1 : 0;
      ctl->lambda = S.lambda;
      corner_finish(P, ctl, Cn, Dnp);
      S.fact_ok = ctl->fact_ok;
    }
    __syncthreads();
    backsub_tiles<NW>(P, ctl, panel);
    // update
    for (int i = tid; i < 3 * P.n; i += NT) {
      const int a = P.act[i / 3];
      if (a >= 0) P.xyz[i] += P.x[3 * a + (i % 3)];
    }
    if (tid == 0) pose_oplus(P.pose, P.x + Dnp);
    double sc = 0.0;
    const double lam = S.lambda;
    for (int r = tid; r < Dn; r += NT) { const double xv = P.x[r]; sc += xv * (lam * xv + P.Hbord[(size_t)6 * Dnp + r]); }
    __syncthreads();
    block_sum<1>(&sc, red, out);
    const double scale_nodes = out[0];
    __syncthreads();
    const double chi_new = eval_edges<false, 0>(P, ctl, red, out, asm_records<NW, 0>(P, panel));
    if (tid == 0) {
      for (int i = 0; i < SFT_SC_XCHG; i++) S.send[i] = 0.0;
      S.send[0] = chi_new; S.send[1] = scale_nodes;
    }
  } else {   // SFT_SC_CTL
    if (tid == 0) {
      const int ok = S.fact_ok;
      S.all_ok &= ok;
      double scale = S.recv[1];   // node parts of every rank; the camera part once, from the all-reduced b_c
      for (int k = 0; k < 6; k++) { const double xv = P.x[Dnp + k]; scale += xv * (S.lambda * xv + S.bc[k]); }
      double tempChi = ok ? S.recv[0] : DBL_MAX;
      double rho = (S.chi_cur - tempChi);
      rho /= (scale + 1e-3);
      S.rho = rho;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        const double sf = fmax(1. / 3., alpha);
        S.lambda *= sf; S.ni = 2.0; S.chi_cur = tempChi; S.accepted = 1;
        ctl->stop = 0;
      } else {
        S.lambda *= S.ni; S.ni *= 2.0;
        ctl->stop = 1;
      }
      S.qmax++;
    }
    __syncthreads();
    if (ctl->stop) {  // pop
      for (int i = tid; i < 3 * P.n; i += NT) P.xyz[i] = P.xyz_bak[i];
      if (tid < 7) P.pose[tid] = S.pose_bak[tid];
    }
    __syncthreads();
    const bool again = (S.rho < 0) && (S.qmax < 10);
    bool finished = false;
    if (!again) {   // the outer iteration is over: trace, termination tests (sparse_optimizer.cpp / DefOptimizer's stop rule)
      if (tid == 0) {
        S.trials += S.qmax;
        S.iters++;
        if (P.trace) {
          double* t = P.trace + S.it * 8;
          t[0] = S.chi_ini; t[1] = S.lambda_start; t[2] = S.qmax; t[3] = S.chi_cur; t[4] = S.lambda; t[5] = S.rho; t[6] = S.accepted; t[7] = S.all_ok;
        }
        if (!S.all_ok) P.res->status |= 1;
        bool term = (S.qmax == 10) || (S.rho == 0);
        if (!term) {
          if ((S.chi_ini - S.chi_cur) * 1e3 < S.chi_ini) S.nbad++; else S.nbad = 0;
          term = S.nbad >= 3;
        }
        S.it++;
        if (S.it >= P.max_iters) term = true;
        S.done = term ? 1 : 0;
        ctl->qmax = term ? 1 : 0;
      }
      __syncthreads();
      finished = ctl->qmax != 0;
    }
    if (tid == 0) S.again = again ? 1 : 0;
    if (finished) classify<NT>(P, ctl, panel, S.iters, S.trials);
  }
}

// ------------------------------------------------------------------------------------------
// Connected-mesh mode across two GPUs: ONE problem on ONE connected template, cut like the two-sided factorisation (SftPart): rank g
// factors part g of the band, the ranks all-reduce their Schur contributions to the separator + camera system (the "halo": the
// separator is one bandwidth of unknowns, so every curvature / stretching / observation edge that crosses the cut lives in it or in
// its coupling rows), every rank solves that reduced system, back-substitutes its own part, and a second all-reduce assembles the
// update.  Residuals, Jacobians and the Levenberg-Marquardt control are replicated (every rank holds the whole state and evaluates all
// edges: identical numbers, identical decisions, no collective for the control); what is partitioned is the factorisation, i.e. where
// the time goes.  Phases, sequenced by the host between the collectives (dsh_api.cpp: cn_solve):
//   LIN  linearise (both parts' band matrices), initial damping        FAC  push, factor the rank's part -> its exchange buffer
//   -- all-reduce of the exchange buffers into the reduced problem --
//   SOL  reduced factor + solve, back substitution of the rank's part, its piece of the update in the natural ordering (zeros elsewhere)
//   -- all-reduce of the update --
//   CTL  state update, trial evaluation, accept / reject, termination, classification
// ------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, SFT_WAVES_PER_EU) void sft_cn_kernel(const SftDev* __restrict__ probs, SftSc* __restrict__ scs, int phase) {
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SftDev& P = probs[blockIdx.x];
  SftSc& S = scs[blockIdx.x];
  Ctl* ctl = reinterpret_cast<Ctl*>(smem);
  double* red = reinterpret_cast<double*>(smem + 512);
  double* out = red + 16 * 27 + 5;
  double* panel = out + 32;
  const int tid = threadIdx.x;
  const int Dn = P.Dn;
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  const int part = S.rank;
  if (phase == SFT_CN_LIN) {
    if (S.it == 0) {
      init_state<NT>(P);
      if (tid == 0) { S.lambda = -1.0; S.ni = 2.0; S.nbad = 0; S.iters = 0; S.trials = 0; S.done = 0; }
      __syncthreads();
    }
    const double chi0 = linearise<NW>(P, ctl, red, out, panel, [] {});
    if (S.it == 0) {
      double mx = 0.0;
      for (int r = tid; r < Dn; r += NT) mx = fmax(mx, fabs(h_diag(P, r)));
      if (tid < 6) mx = fmax(mx, fabs(P.Hcorner[tid * 8]));
      mx = block_max(mx, red);
      if (tid == 0) { S.lambda = 1e-5 * mx; S.ni = 2.0; S.nbad = 0; }
    }
    __syncthreads();
    if (tid == 0) { S.chi_cur = chi0; S.chi_ini = chi0; S.qmax = 0; S.rho = 0.0; S.accepted = 0; S.all_ok = 1; S.lambda_start = S.lambda; }
  } else if (phase == SFT_CN_FAC) {
    for (int i = tid; i < 3 * P.n; i += NT) P.xyz_bak[i] = P.xyz[i];   // push
    if (tid < 7) S.pose_bak[tid] = P.pose[tid];
    if (tid == 0) ctl->lambda = S.lambda;
    __syncthreads();
    if constexpr (NW == 8) factor_wide(P, part, ctl, panel);
  } else if (phase == SFT_CN_SOL) {
    // P.part[2].xchg holds the all-reduced sum of the two exchange buffers
    const int nTr = P.part[2].nT;
    const bool parts_ok = P.part[2].xchg[(size_t)nTr * P.part[2].tpr * (TS * TS) + (size_t)8 * TS * nTr + 56] == 0.0;
    if (tid == 0) ctl->lambda = S.lambda;
    __syncthreads();
    if constexpr (NW == 8) {
      if (parts_ok) factor_wide(P, 2, ctl, panel);
      else if (tid == 0) ctl->fact_ok = 0;
      __syncthreads();
      backsub_wide(P, 2, ctl, panel);
      backsub_wide(P, part, ctl, panel);
    }
    for (int i = tid; i < Dnp + 6; i += NT) P.x[i] = 0.0;
    __syncthreads();
    if (ctl->fact_ok) {   // this rank's piece of the update, natural ordering (rank 0 also carries the separator and the camera)
      const int c0 = P.sp_c0, sp = P.sp_s, pad = P.sp_pad, n1p = P.sp_n1p;
      const auto xa = P.part[part].x, xr = P.part[2].x;
      if (part == 0) {
        for (int j = tid; j < c0; j += NT) P.x[j] = xa[j];
        for (int k = tid; k < sp; k += NT) P.x[c0 + k] = xr[k];
        if (tid < 6) P.x[Dnp + tid] = xr[TS * nTr + tid];
      } else {
        for (int j = pad + tid; j < n1p; j += NT) P.x[Dn - 1 - (j - pad)] = xa[j];
      }
    }
    if (tid == 0) S.fact_ok = ctl->fact_ok;
  } else {   // SFT_CN_CTL: P.x holds the complete update on every rank
    for (int i = tid; i < 3 * P.n; i += NT) {
      const int a = P.act[i / 3];
      if (a >= 0) P.xyz[i] += P.x[3 * a + (i % 3)];
    }
    if (tid == 0) pose_oplus(P.pose, P.x + Dnp);
    double sc = 0.0;
    const double lam = S.lambda;
    for (int r = tid; r < Dn; r += NT) { const double xv = P.x[r]; sc += xv * (lam * xv + P.Hbord[(size_t)6 * Dnp + r]); }
    if (tid < 6) { const double xv = P.x[Dnp + tid]; sc += xv * (lam * xv + P.Hcorner[42 + tid]); }
    __syncthreads();
    block_sum<1>(&sc, red, out);
    const double scale = out[0];
    __syncthreads();
    const double chi_new = eval_edges<false, 0>(P, ctl, red, out, asm_records<NW, 0>(P, panel));
    if (tid == 0) {
      const int ok = S.fact_ok;
      S.all_ok &= ok;
      const double tempChi = ok ? chi_new : DBL_MAX;
      double rho = (S.chi_cur - tempChi);
      rho /= (scale + 1e-3);
      S.rho = rho;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        const double sf = fmax(1. / 3., alpha);
        S.lambda *= sf; S.ni = 2.0; S.chi_cur = tempChi; S.accepted = 1;
        ctl->stop = 0;
      } else {
        S.lambda *= S.ni; S.ni *= 2.0;
        ctl->stop = 1;
      }
      S.qmax++;
    }
    __syncthreads();
    if (ctl->stop) {  // pop
      for (int i = tid; i < 3 * P.n; i += NT) P.xyz[i] = P.xyz_bak[i];
      if (tid < 7) P.pose[tid] = S.pose_bak[tid];
    }
    __syncthreads();
    const bool again = (S.rho < 0) && (S.qmax < 10);
    bool finished = false;
    if (!again) {
      if (tid == 0) {
        S.trials += S.qmax;
        S.iters++;
        if (P.trace) {
          double* t = P.trace + S.it * 8;
          t[0] = S.chi_ini; t[1] = S.lambda_start; t[2] = S.qmax; t[3] = S.chi_cur; t[4] = S.lambda; t[5] = S.rho; t[6] = S.accepted; t[7] = S.all_ok;
        }
        if (!S.all_ok) P.res->status |= 1;
        bool term = (S.qmax == 10) || (S.rho == 0);
        if (!term) {
          if ((S.chi_ini - S.chi_cur) * 1e3 < S.chi_ini) S.nbad++; else S.nbad = 0;
          term = S.nbad >= 3;
        }
        S.it++;
        if (S.it >= P.max_iters) term = true;
        S.done = term ? 1 : 0;
        ctl->qmax = term ? 1 : 0;
      }
      __syncthreads();
      finished = ctl->qmax != 0;
    }
    if (tid == 0) S.again = again ? 1 : 0;
    if (finished) classify<NT>(P, ctl, panel, S.iters, S.trials);
  }
}

#include "sft_batch.h"

#ifdef DSH_LAB
// Measurement kernel of the Jacobian-assembly roofline: one linearisation (residuals + Jacobian records) and one
// normal-equation assembly per problem at its uploaded initial state, nothing else.  H and the border keep the zero pattern
// of the last full run of the batch (the assembly overwrites every structural non-zero).  Same launch shape and LDS layout
// as sft_lm_kernel; its own name keeps the profiler statistics of the two apart.
template <int NW>
__global__ __launch_bounds__(64 * NW, SFT_WAVES_PER_EU) void sft_assembly_kernel(const SftDev* __restrict__ probs) {
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SftDev& P = probs[blockIdx.x];
  Ctl* ctl = reinterpret_cast<Ctl*>(smem);
  double* red = reinterpret_cast<double*>(smem + 512);
  double* out = red + 16 * 27 + 5;
  double* panel = out + 32;
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * P.n; i += NT) P.xyz[i] = P.xyz_init[i];
  if (tid < 7) P.pose[tid] = P.pose_init[tid];
  __syncthreads();
  double chi = 0.0;
  switch (P.lds_class) {
    case 3: { const auto jp = asm_records<NW, 3>(P, panel); chi = eval_edges<true, 3>(P, ctl, red, out, jp); assemble<NW, 3>(P, red, out, jp); break; }
    case 2: { const auto jp = asm_records<NW, 2>(P, panel); chi = eval_edges<true, 2>(P, ctl, red, out, jp); assemble<NW, 2>(P, red, out, jp); break; }
    case 1: { const auto jp = asm_records<NW, 1>(P, panel); chi = eval_edges<true, 1>(P, ctl, red, out, jp); assemble<NW, 1>(P, red, out, jp); break; }
    default: { const auto jp = asm_records<NW, 0>(P, panel); chi = eval_edges<true, 0>(P, ctl, red, out, jp); assemble<NW, 0>(P, red, out, jp); break; }
  }
  if (tid == 0) P.dbg[0] = chi;
}

// A/B pair for the one-wavefront solver (sft_wave.h): both solve (H + lambda I) x = b on the H that sft_assembly_kernel left in memory,
// lambda = rel * 1e-5 * max |diag H| (the first damping of the LM loop scaled by rel); x in P.x, lambda in P.dbg[1], "ok" in P.dbg[2].
template <int NW>
__device__ __forceinline__ void ref_factor_solve(const SftDev& P, Ctl* ctl, double* panel) {
  tile_factor<NW>(P, ctl, panel);
  backsub_tiles<NW>(P, ctl, panel);
}
template <int NW>
__global__ __launch_bounds__(64 * NW, SFT_WAVES_PER_EU) void sft_ref_solve_kernel(const SftDev* __restrict__ probs, double rel) {
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SftDev& P = probs[blockIdx.x];
  Ctl* ctl = reinterpret_cast<Ctl*>(smem);
  double* red = reinterpret_cast<double*>(smem + 512);
  double* out = red + 16 * 27 + 5;
  double* panel = out + 32;
  const int tid = threadIdx.x;
  double mx = 0.0;
  for (int r = tid; r < P.Dn; r += NT) mx = fmax(mx, fabs(h_diag(P, r)));
  if (tid < 6) mx = fmax(mx, fabs(P.Hcorner[tid * 8]));
  mx = block_max(mx, red);
  if (tid == 0) { ctl->lambda = rel * 1e-5 * mx; P.dbg[1] = ctl->lambda; }
  __syncthreads();
  ref_factor_solve<NW>(P, ctl, panel);
  if (tid == 0) P.dbg[2] = ctl->fact_ok;
}
__global__ __launch_bounds__(64, 1) void sft_wave_solve_kernel(const SftDev* __restrict__ probs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SftDev& P = probs[blockIdx.x];
  const double lambda = P.dbg[1];
  const int ok = wv_factor_solve(P, lambda, lambda, to_lds(reinterpret_cast<double*>(smem)));
  int lane_now;   // (not threadIdx.x: a value that lives across the factorisation may be parked on a window tile -- sft_wave.h, wv_factor)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_now));
  if (lane_now == 0) P.dbg[2] = ok;
}
#endif  // DSH_LAB

}  // namespace


// LDS bytes the kernel needs for a problem with half-bandwidth kd
extern "C" size_t sft_lm_kernel_lds_bytes(int kd, size_t jl_doubles) {
  const size_t rows = NB + kd + SFT_BORDER;
  const size_t LDP = rows | 1;
  size_t panel = (size_t)NB * LDP + 2 * NB * NB;   // panel + diagraw + lrow
  const size_t backsub = NB + (SFT_NT / NB) * NB + NB * NB;
  if (backsub > panel) panel = backsub;
  const size_t tiles = (size_t)(BT + 2 * (BT + 1) + 3) * TILE_LDS + 3 * SFT_BORDER * TS + 64 + 48 + 128 + 128;  // dataflow layout (the larger one) + step-trace stamps
  if (kd <= TS * BT) panel = tiles;
  else if (kd <= TS * WB) panel = std::max(panel, std::max((size_t)2 * WB * TS * TS + TILE_LDS + 704, (size_t)(SFT_WIDE_NEAR + 2) * SFT_WIDE_NEAR * TS * TS + 2 * TILE_LDS + 704 + (size_t)32 * TS * TS + (size_t)(WB - SFT_WIDE_NEAR + 1) * TS * TS));   // wide mode: two staged tile rows, W, corners (or the band panel)
  if (jl_doubles > panel) panel = jl_doubles;
  return 512 + (16 * 27 + 5 + 32 + panel) * sizeof(double) + 64;
}

#ifdef DSH_LAB
extern "C" hipError_t sft_assembly_launch(const SftDev* d_probs, int B, int max_kd, size_t jl_doubles, int nw, hipStream_t stream) {
  const size_t lds = sft_lm_kernel_lds_bytes(max_kd, jl_doubles);
  const void* fn = nw == 4 ? reinterpret_cast<const void*>(sft_assembly_kernel<4>) : reinterpret_cast<const void*>(sft_assembly_kernel<8>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if (nw == 4) hipLaunchKernelGGL(sft_assembly_kernel<4>, dim3(B), dim3(256), lds, stream, d_probs);
  else hipLaunchKernelGGL(sft_assembly_kernel<8>, dim3(B), dim3(512), lds, stream, d_probs);
  return hipGetLastError();
}
#endif

extern "C" hipError_t sft_sc_launch(const SftDev* d_probs, SftSc* d_sc, int B, int phase, int max_kd, size_t jl_doubles, size_t* configured, hipStream_t stream) {
  const size_t lds = sft_lm_kernel_lds_bytes(max_kd, jl_doubles);
  if (lds > *configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sft_sc_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    *configured = lds;
  }
  hipLaunchKernelGGL(sft_sc_kernel<8>, dim3(B), dim3(512), lds, stream, d_probs, d_sc, phase);
  return hipGetLastError();
}

extern "C" hipError_t sft_cn_launch(const SftDev* d_probs, SftSc* d_sc, int phase, int max_kd, size_t jl_doubles, size_t* configured, hipStream_t stream) {
  const size_t lds = sft_lm_kernel_lds_bytes(max_kd, jl_doubles);
  if (lds > *configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sft_cn_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    *configured = lds;
  }
  hipLaunchKernelGGL(sft_cn_kernel<8>, dim3(1), dim3(512), lds, stream, d_probs, d_sc, phase);
  return hipGetLastError();
}

extern "C" hipError_t sftb_tail_launch(const SftDev* d_probs, SftRun* d_runs, int* d_counters, int B, int max_kd, size_t jl_doubles, size_t* configured, int num_cus, int tail_below, hipStream_t stream) {
  const size_t lds = sft_lm_kernel_lds_bytes(max_kd, jl_doubles);   // the larger of the linearisation's records and the solver's workspace
  if (lds > *configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sftb_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    *configured = lds;
  }
  hipLaunchKernelGGL(sftb_tail_kernel, dim3(std::min(B, num_cus)), dim3(64 * SFTB_NW), lds, stream, d_probs, d_runs, d_counters, B, tail_below);
  return hipGetLastError();
}

#ifdef DSH_LAB   // an A/B variant (lab option owner_waves 16; measured slower than eight wavefronts: DESIGN 4.2)
// The FACTOR launch of sft_spec_kernel for parts with helper workgroups, SIXTEEN wavefronts per workgroup (128 registers per lane): the owner
// of a part keeps one live row per wave (factor_part<NEAR, 16>), a helper one item per wave (FarColumn16).  Same grid layout (role-major), same
// early exits, same damping as the FACTOR branch of sft_spec_kernel; results bit-identical to it.
__global__ __launch_bounds__(1024) void sft_part_factor_kernel(const SftDev* __restrict__ probs, SftSpec* __restrict__ specs, int K, int nh) {
  constexpr int NT = 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int roles = 1 + nh, per_role = gridDim.x / roles;
  const int role = blockIdx.x / per_role, bid = blockIdx.x % per_role;
  const int fpart = bid % 2, bx = bid / 2;
  const int B = per_role / (K * 2), b = bx / K, j = bx % K;
  const SftDev& P = probs[(size_t)j * B + b];
  const SftSpec& S = specs[(size_t)j * B + b];
  Ctl* ctl = reinterpret_cast<Ctl*>(smem);
  double* red = reinterpret_cast<double*>(smem + 512);
  double* panel = red + 16 * 27 + 5 + 32;
  const int tid = threadIdx.x;
  if (S.done) return;
  if (!(P.tile_mode == 2 && P.split) || S.need_lin == 1) return;
  if (S.qbase + j >= 10) return;
  const int epoch = S.launches + 1;
  if (role > 0) {
    factor_wide_helper<16>(P, fpart, role - 1, nh, epoch, ctl, panel);
    return;
  }
  double lam = S.lambda, ni = S.ni;
  if (S.need_lin == 2 && S.it == 0) {
    double mx = 0.0;
    for (int r = tid; r < P.Dn; r += NT) mx = fmax(mx, fabs(h_diag(P, r)));
    if (tid < 6) mx = fmax(mx, fabs(P.Hcorner[tid * 8]));
    mx = block_max(mx, red);
    lam = 1e-5 * mx; ni = 2.0;
  }
  for (int t = 0; t < j; t++) { lam *= ni; ni *= 2.0; }
  if (tid == 0) ctl->lambda = lam;
  __syncthreads();
  factor_part<SFT_WIDE_NEAR, 16>(P, fpart, ctl, panel, epoch, nh);
}
#endif  // DSH_LAB

// out_a[i] = out_b[i] = a[i] + b[i]: the in-process stand-in of a two-rank all-reduce (dsh_sft_connected_solve_group); in place is fine
__global__ void sft_vec_sum2_kernel(const double* a, const double* b, double* out_a, double* out_b, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double v = a[i] + b[i];
    out_a[i] = v;
    out_b[i] = v;
  }
}
extern "C" hipError_t sft_vec_sum2(const double* a, const double* b, double* out_a, double* out_b, int n, hipStream_t stream) {
  hipLaunchKernelGGL(sft_vec_sum2_kernel, dim3(64), dim3(256), 0, stream, a, b, out_a, out_b, n);
  return hipGetLastError();
}

// configured: two marks (sft_spec_kernel<8>, sft_part_factor_kernel).  owner_waves 16 (lab builds): a FACTOR launch with helpers runs sft_part_factor_kernel
extern "C" hipError_t sft_spec_launch(const SftDev* d_probs, SftSpec* d_spec, int B, int K, int phase, int nh, int owner_waves, int max_kd, size_t jl_doubles, size_t* configured, hipStream_t stream) {
  const size_t lds = sft_lm_kernel_lds_bytes(max_kd, jl_doubles);
#ifdef DSH_LAB
  if (phase == SFT_SPEC_FACTOR && nh > 0 && owner_waves == 16) {
    if (lds > configured[1]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sft_part_factor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      configured[1] = lds;
    }
    hipLaunchKernelGGL(sft_part_factor_kernel, dim3(B * K * 2 * (1 + nh)), dim3(1024), lds, stream, d_probs, d_spec, K, nh);
    return hipGetLastError();
  }
#endif
  if (lds > *configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sft_spec_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    *configured = lds;
  }
  const int wgs = (phase == SFT_SPEC_FACTOR || phase == SFT_SPEC_SOLVE) ? 2 : 1, roles = phase == SFT_SPEC_FACTOR ? 1 + nh : 1;
  hipLaunchKernelGGL(sft_spec_kernel<8>, dim3(B * K * wgs * roles), dim3(512), lds, stream, d_probs, d_spec, K, phase, nh);
  return hipGetLastError();
}

// sum of G exchange vectors (the in-process stand-in of the all-reduce: dsh_comm_create_local)
__global__ void sft_sc_local_reduce_kernel(SftSc* const* scs, int G) {
  const int i = threadIdx.x;
  if (i >= SFT_SC_XCHG) return;
  double s = 0.0;
  for (int g = 0; g < G; g++) s += scs[g]->send[i];
  for (int g = 0; g < G; g++) scs[g]->recv[i] = s;
}
extern "C" hipError_t sft_sc_local_reduce(SftSc* const* d_ptrs, int G, hipStream_t stream) {
  hipLaunchKernelGGL(sft_sc_local_reduce_kernel, dim3(1), dim3(64), 0, stream, d_ptrs, G);
  return hipGetLastError();
}

// `configured` (two slots of the caller's per-DEVICE table, dsh_api.cpp: LdsMarks): the largest dynamic LDS size the two kernels have been
// enabled for on that device by anybody in this process -- the function attribute belongs to (device, kernel): it is only ever raised,
// under the caller's lock, so that two contexts on one GPU cannot lower each other's setting.
extern "C" hipError_t sft_lm_launch(const SftDev* d_probs, int B, int max_kd, size_t jl_doubles, int nw, size_t* configured, hipStream_t stream) {
  const size_t lds = sft_lm_kernel_lds_bytes(max_kd, jl_doubles);
  const int slot = nw == 4 ? 0 : 1;
  if (lds > configured[slot]) {
    const void* fn = nw == 4 ? reinterpret_cast<const void*>(sft_lm_kernel<4>) : reinterpret_cast<const void*>(sft_lm_kernel<8>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured[slot] = lds;
  }
  if (nw == 4) hipLaunchKernelGGL(sft_lm_kernel<4>, dim3(B), dim3(256), lds, stream, d_probs);
  else hipLaunchKernelGGL(sft_lm_kernel<8>, dim3(B), dim3(512), lds, stream, d_probs);
  return hipGetLastError();
}

extern "C" size_t sft_lm_kernel_lds_bytes(int kd, size_t jl_doubles);
#ifdef DSH_LAB
extern "C" hipError_t sft_wave_lab_launch(const SftDev* d_probs, int B, int which, double rel, int max_kd, size_t jl_doubles, hipStream_t stream) {
  if (which == 0) {
    (void)jl_doubles;   // (the solver workspace only: this kernel does not assemble)
    const size_t lds = sft_lm_kernel_lds_bytes(max_kd, 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sft_ref_solve_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sft_ref_solve_kernel<4>, dim3(B), dim3(256), lds, stream, d_probs, rel);
  } else {
    hipLaunchKernelGGL(sft_wave_solve_kernel, dim3(B), dim3(64), WV_LDS_DOUBLES * sizeof(double), stream, d_probs);
  }
  return hipGetLastError();
}
#endif

// Phase launches of the batched throughput shape (sft_batch.h).  `configured` (two slots of the per-device table): the largest dynamic LDS
// sizes the LIN and TRIAL kernels have been enabled for on that device (see sft_lm_launch).
extern "C" hipError_t sftb_launch(const SftDev* d_probs, SftRun* d_runs, int* d_counters, int* d_list, int B, int phase, size_t jl_doubles, size_t xyz_doubles, size_t* configured, int num_cus, int tail_below, hipStream_t stream) {
  const size_t head = 512 + (16 * 27 + 5 + 32) * sizeof(double) + 64;
  if (phase == SFTB_PH_INIT) {
    hipError_t e = hipMemsetAsync(d_counters, 0, 16 * sizeof(int), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sftb_init_kernel, dim3(B), dim3(64 * SFTB_NW), 0, stream, d_probs, d_runs, d_counters, d_list, tail_below);
  } else if (phase == SFTB_PH_LIN) {
    const size_t lds = head + jl_doubles * sizeof(double);
    if (lds > configured[0]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sftb_lin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      configured[0] = lds;
    }
    const int per_cu = lds > 80 * 1024 ? 1 : 2;   // persistent workgroups: as many as are resident at once
    hipLaunchKernelGGL(sftb_lin_kernel, dim3(std::min(B, per_cu * num_cus)), dim3(64 * SFTB_LIN_NW), lds, stream, d_probs, d_runs, d_counters, d_list, B, tail_below);
  } else if (phase == SFTB_PH_FACTOR) {
    hipLaunchKernelGGL(sftb_factor_kernel, dim3(std::min(B, 4 * num_cus)), dim3(64), WV_LDS_DOUBLES * sizeof(double), stream, d_probs, d_runs, d_counters, B);   // one wave per SIMD
  } else {
    const size_t lds = head + std::max<size_t>(2048, xyz_doubles) * sizeof(double);   // classify: the error norms of 2048 observations per pass; the trial's residual pass: the node positions
    if (lds > configured[1]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sftb_trial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      configured[1] = lds;
    }
    hipLaunchKernelGGL(sftb_trial_kernel, dim3(B), dim3(64 * SFTB_NW), lds, stream, d_probs, d_runs, d_counters, d_list);
  }
  return hipGetLastError();
}
