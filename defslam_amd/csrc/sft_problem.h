// Device-side description of one packed Shape-from-Template problem.
//
// One problem = one call of defSLAM::Optimizer::DefPoseOptimization
// (Modules/Tracking/DefOptimizer.cc:251-578).  The host packer (sft_pack.cpp) turns the
// reference's pointer graph into the flat arrays below; the persistent kernel
// (sft_kernels.hip) runs the whole Levenberg-Marquardt loop on them.
//
// Unknown ordering on the device: active nodes first (compact index a, scalars 3a..3a+2),
// the 6 camera unknowns LAST (arrowhead border) -- g2o puts the camera first
// (sparse_optimizer.cpp:181-211); the two orderings are a symmetric permutation of the
// same linear system.
#pragma once
#include <stdint.h>

#define SFT_NT 512             // threads of the band-mode workgroup / upper bound of the tile-mode one (8 wavefronts)
#define SFT_CAM_STRIDE 16      // doubles per observation camera record: rho' w, e0, e1, five non-zero entries of each row of J_cam, chi2, 2 pad = ONE 128-byte line
#define SFT_BORDER 7           // 6 camera rows + the right-hand side carried through the factorisation
// tile mode keeps zero padding around H so that the sliding window loads every tile unconditionally:
#define SFT_H_PAD_TILE_ROWS 9  // zero tile rows below the band matrix (window height BT + the look-ahead tile)
#define SFT_H_PAD_BORDER 160   // doubles after the 8th (zero) border row: one window of tile columns

// contribution record: kind(2) | slot_row(4) | slot_col(4) | edge(22)
#define SFT_KIND_OBS 0u
#define SFT_KIND_REF 1u
#define SFT_KIND_STAR 2u
#define SFT_KIND_STR 3u
#define SFT_REC(kind, s, t, e) (((uint32_t)(kind) << 30) | ((uint32_t)(s) << 26) | ((uint32_t)(t) << 22) | (uint32_t)(e))

// Every pointer below addresses global (HBM) memory.  Saying so in the device pass makes the compiler emit global_*
// instead of flat_* memory instructions: flat loads count on the LDS counter too, so every LDS wait would also wait
// for all outstanding HBM traffic.  The host pass sees plain pointers; the layout is identical.
#if defined(__HIP_DEVICE_COMPILE__) && defined(SFT_KERNEL_SOURCE)   // sft_kernels.hip only; host sources see plain pointers
#define SFT_G __attribute__((address_space(1)))
#else
#define SFT_G
#endif

// Per-problem result header.  The headers of a batch are contiguous (one copy brings every counter back) and are followed
// by the result bodies (vertices, per-observation chi2, trace, map points, outlier flags): dsh_sft_batch_download moves the
// whole region with ONE hipMemcpyAsync.  Classification and statistics are computed by the kernel (DefOptimizer.cc:515-559).
struct SftResHdr {
  int32_t iters, trials, status, inliers;   // outer LM iterations, damping trials, bit 0: a factorisation failed, M - nBad
  double rep_error;                         // sumError / n over the inliers, summed in index order (pFrame->repError)
  double pose[8];                           // t(3), q(x,y,z,w); [7] unused
  double pad[5];
};                                          // 128 bytes

// State of the shared-camera mode between its phase kernels (sft_kernels.hip: sft_sc_kernel; host loop: dsh_multi.cpp).
#define SFT_SC_XCHG 32          // doubles per exchange (all-reduce) vector
#define SFT_SC_LIN 0
#define SFT_SC_FAC 1
#define SFT_SC_SOL 2
#define SFT_SC_CTL 3
// Connected-mesh mode across two ranks (sft_kernels.hip: sft_cn_kernel): the same phase structure; the controller state lives in SftSc.
#define SFT_CN_LIN 0
#define SFT_CN_FAC 1
#define SFT_CN_SOL 2
#define SFT_CN_CTL 3
struct SftSc {
  double send[SFT_SC_XCHG], recv[SFT_SC_XCHG];   // local partials / their sum over the ranks
  double lambda, ni, chi_cur, chi_ini, rho, lambda_start;
  double bc[6];                 // all-reduced b_c of the current linearisation
  double pose_bak[8];
  int32_t it, qmax, nbad, accepted, again, done, all_ok, fact_ok, iters, trials, rank, nranks;
};

// Speculative damping trials (latency mode, sft_kernels.hip: sft_spec_kernel): K workgroups ("lanes") per problem try the next K
// dampings lambda, lambda nu, lambda nu 2nu, ... of the Levenberg-Marquardt rejection chain at the same time.  Every lane keeps
// the controller state (identical by construction) and publishes its trial, double buffered by launch parity.
#define SFT_SPEC_MAXK 4
#define SFT_SPEC_INIT 0
#define SFT_SPEC_LIN 1
#define SFT_SPEC_TRIAL 2
// waves per SIMD the persistent kernels are compiled for (A/B builds override it: tools/ab_build.sh)
#ifndef SFT_WAVES_PER_EU
#define SFT_WAVES_PER_EU 2      // 256 VGPRs per wave: the trailing window of the factorisation lives in accumulator registers
#endif
#define SFT_SPEC_FACTOR 3       // split problems: the two parts of the factorisation, one workgroup each, in front of SFT_SPEC_TRIAL
#define SFT_SPEC_SOLVE 4        // split problems: reduced solve (by both workgroups, each in its own copy) + the workgroup's part back-substituted
struct SftSpecRes { double chi_new, scale, lambda, ni, pose[8]; int32_t ok, valid; };
struct SftSpec {
  double lambda, ni, chi_cur, chi_ini, lambda_start, rho;
  double pose_bak[8];
  int32_t it, qbase, nbad, accepted, all_ok, iters, trials, done, launches, need_lin, last_lane, pad;   // launches: completed trial rounds; need_lin: 1 linearise,
                                  // 2 linearised, lambda of the first iteration still to come; pad: a trial round waits for its verdict
  SftSpecRes res[2];
};

// Batched throughput shape (sft_batch.h): the LM controller of a problem between the phase kernels of a round.
#define SFTB_LIN 0      // the next launch it takes part in is a linearisation (start of an outer iteration)
#define SFTB_TRIAL 1    // H is assembled: factor with R.lambda, then the trial
#define SFTB_DONE 2
#define SFTB_FINISH 3   // max_iters == 0: classification only
#define SFTB_PH_INIT 0
#define SFTB_PH_LIN 1
#define SFTB_PH_FACTOR 2
#define SFTB_PH_TRIAL 3
#define SFTB_PH_TAIL 4     // the last problems of a step: one workgroup runs each to its end (sft_batch.h: sftb_tail_kernel)
struct SftRun {
  double lambda, ni, chi_cur, chi_ini, rho, lambda_start;
  double pose_bak[8];
  int32_t it, qmax, nbad, accepted, all_ok, iters, trials, state, fact_ok, pad[3];
};

// Two-sided factorisation of a wide-band problem (sft_wide.h, latency mode; the connected-mesh mode across two GPUs uses the same cut).
// The band ordering is cut into  [ part 0 : scalars 0 .. c0 ) [ separator : c0 .. c0+s ) [ part 1 : c0+s .. Dn ).  With s >= the scalar
// half-bandwidth no element of H joins the two parts, so both can be eliminated at the same time -- part 0 top-down, part 1 in
// REVERSED order (bottom-up), each as a band factorisation that simply continues into the separator rows but stops after its own
// columns: what it leaves in the separator block is its Schur contribution.  The separator (+ camera) system is the sum of the two
// contributions, is solved once, and the parts back-substitute independently.
//   part g as a band matrix: nS eliminated tile columns (part 1: pad identity scalars first, so that the separator starts on a tile
//   boundary), then the sT separator tile rows; its border columns are read from the natural border rows through (base, sign).
//   part[2] = the reduced (separator) problem: dense band of sT tile columns, border = camera + right-hand side; part[3] = a second
//   workspace of the same shape: in SFT_SPEC_SOLVE both workgroups of a lane solve the reduced problem (cheaper than handing its
//   solution from one to the other through another launch), workgroup g in part[2 + g].
struct SftPart {
  int32_t nT, nS, tpr, wbt;            // tile rows (eliminated + separator), eliminated tile columns, tile pitch of a row, most sub-diagonal tiles
  int32_t b_base, b_sign, b_lo, b_hi;  // column j of the part (b_lo <= j < b_hi) is natural border column b_base + b_sign * j; others are zero
  SFT_G double* Hb;                    // H of the part, wide tile layout (tile (I,J) at (I*tpr + I-J)*256, transposed tiles); shared by the lanes of a problem
  SFT_G double *Lb, *Lt, *LbT, *Lbord, *Linv, *x;
  SFT_G double* xchg;                  // parts 0/1: the part's Schur contribution in the layout of the reduced problem's input
                                       //   [H tiles sT*tpr_r*256 | border 8 x 16 sT | corner 56 | failed flag 8]; part[2]: the sum (its Hb points into it)
  // parts 0/1, latency mode with helper workgroups (sft_wide.h: factor_wide_helper): the far sums of a block column formed on another CU
  SFT_G double *Pf, *PfB;              // Pf: tile (J, d) = far(J + d, J), layout of Lt; PfB: tile J = the border's far sum
  SFT_G int32_t* sync;                 // [0] the owner's progress (epoch << 16 | finished block columns), [16 + J] == epoch: column J's far sums are stored
};

struct SftDev {
  // sizes
  int32_t n, nA, Dn, kd, ldh, M, V, S, Es, noff, max_iters, mode;
  int32_t tile_mode;          // 1: 16x16-tile band storage + register-window MFMA factorisation (kd <= 128); 2: wide tile band, left-looking
                              // MFMA factorisation (kd <= 256, sft_wide.h); 0: row-major band (general)
  int32_t pad1;
  int32_t lds_class;          // assembly records kept in LDS instead of the workspace: 0 none, 1 observation weights + curvature records, 2 also node matrices + stretch records, 3 also the camera records (phase rounds only)
  int32_t tpr, wbt;           // tile modes: pitch of a tile row of the band storage in tiles (wbt + 1), sub-diagonal tiles per block
                              // column (mode 1: 8; mode 2: ceil(kd/16); an even pitch, i.e. an odd tile distance between (I,K) and
                              // (I,K+1), was tried against L2 channel aliasing: no effect)
  int32_t pad0;
  double fx, fy, cx, cy;
  double w_ref, w_curv, w_str, hub_delta, hub_dsqr;
  // template (shared by every problem of a batch)
  const SFT_G double* xyz0;
  const SFT_G int32_t* nbr_ptr;
  const SFT_G int32_t* nbr_idx;
  const SFT_G double* nbr_w;
  const SFT_G double* nbr_sumw;   // per node
  const SFT_G double* k0;
  // graph: structure of the normal equations for this template and active set (sft_pack.h: SftGraph), device-resident and
  // shared by every problem with the same active set.  Blocks: q < nA diagonal block of active node q, q >= nA off-diagonal block q - nA.
  const SFT_G int32_t* act;       // n: compact index or -1
  const SFT_G int32_t* actnode;   // nA: node of compact index a
  const SFT_G int32_t* star_node; // S
  const SFT_G double* star_sL;    // S  sum over incident mesh edges of 1/L^2
  const SFT_G int32_t* str_nodes; // Es*2
  const SFT_G double* str_L0;     // Es
  const SFT_G int32_t* off_ptr;   // nA+1: off-diagonal blocks of block row a (columns ascending)
  const SFT_G int32_t* off_rc;    // noff*2 (block row, block col)
  const SFT_G int32_t* sh_ptr;    // nblk+1: curvature / stretch contributions of block q
  const SFT_G uint32_t* sh_rec;   // SFT_REC
  const SFT_G double* sh_cf;      // 2 per contribution: H and b factors without the regulariser weight
  const SFT_G int32_t* tmask;     // tile mode 1: per tile row I (nT + SFT_H_PAD_TILE_ROWS entries) bit d set if tile (I, I-d) holds any element of H
  const SFT_G uint32_t* hgatherT; // tile mode 1: the same lists for the transposed tile, lane (g, c), register q: element [row c][column g + 4q] (sft_wave.h)
  const SFT_G uint32_t* hgather;  // tile mode 1: the element of Hc every (lane, register) of tile (I, I-d) takes: ((I*9 + d)*64 + lane)*4 + q (sft_pack.h)
  // frame
  const SFT_G int32_t* obs_nodes; // M*3
  const SFT_G double* obs_bary;   // M*3
  const SFT_G double* obs_uv;     // M*2
  const SFT_G double* obs_w;      // M  invSigma2 / N_frame
  const SFT_G int32_t* ob_ptr;    // nblk+1: observation contributions of block q, observation order
  const SFT_G int32_t* ob_m;      // observation index
  const SFT_G double* ob_c;       // diagonal block: b_s; off-diagonal block: b_s b_t
  const SFT_G uint8_t* viewed;    // nA: the node carries a reference (temporal) edge
  // initial state (restored at the start of every run)
  const SFT_G double* xyz_init;   // n*3
  const SFT_G double* pose_init;  // 7: t, q(x,y,z,w)
  // state + workspace
  SFT_G double* xyz;              // n*3
  SFT_G double* xyz_bak;          // n*3
  SFT_G double* pose;             // 7 (inside *res)
  SFT_G double* camrec;           // M*SFT_CAM_STRIDE: rho' w, e, the non-zero entries of J_cam (sft_types.h:162-174)
  SFT_G double* wtv;              // M     rho' w            (lds_class < 1)
  SFT_G double* Anode;            // nA*6  node matrices A_i (when not in LDS): J_node of an observation = b_s A_node (sft_types.h:176-205)
  SFT_G double* Jstar;            // S*4  (u, r)           (when not in LDS)
  SFT_G double* Jstr;             // Es*4 (g, e)           (when not in LDS)
  SFT_G double* Hc;               // tile mode 1: H as compact 3x3 blocks, 9 * (nA + noff) doubles + one 0.0 + one 1.0; block rows interleaved:
                            //            diagonal block of node a at 9 * (a + off_ptr[a]), off-diagonal block q of block row a at 9 * (a + 1 + q)
  SFT_G double* Hb;               // band mode: Dnp*ldh lower band, row-major: (r,c) at r*ldh + c-r+kd
                            // tile mode 2: nT*tpr 16x16 tiles, tile (I,J) at (I*tpr + I-J)*256, element (row,col) at
                            //            ((row&3)*16 + col)*4 + (row>>2)  (= MFMA accumulator order: lane, register),
                            //            off-diagonal tiles transposed (element (col,row)); L uses the tile layout in modes 1 and 2
  SFT_G double* Hbord;            // 7*Dn     rows 0-5: camera x node, row 6: b_node
  SFT_G double* Hcorner;          // 7*7      camera x camera (lower) + b_cam in row 6
  SFT_G double* Lb;               // Dn*ldh
  SFT_G double* Lbord;            // 7*Dn
  SFT_G double* Lcorner;          // 7*7
  SFT_G double* Linv;             // tile mode: nT inverse diagonal tiles (16x16 row-major)
  SFT_G double* Lt;               // mode 2: the transposed L tiles (tile (I,K) holds X(I,K)^T at slot (K, I-K)), operands of the left-looking update
  SFT_G double* LbT;              // mode 2: nT tiles, tile K = transposed 7x16 border block of block column K
  SFT_G double* x;                // Dn+6
  // outputs
  SFT_G double* chi2_obs;         // M
  SFT_G double* trace;            // max_iters*8
  SFT_G SftResHdr* res;           // counters, statistics and the final pose (pose points into it)
  SFT_G uint8_t* outlier;         // M  (float)chi2 > 5.991 (DefOptimizer.cc:515-537)
  SFT_G float* mappoint;          // M*3 DefMapPoint::RecalculatePosition of every observation's point (DefMapPoint.cc:129-147)
  SFT_G double* dbg;              // lab builds: [0] robust chi2 of dsh_lab_sft_system, phase timers, step stamps
  SFT_G double* spec_xyz[2];      // speculative trials: n*3 each, the lane's state after its trial (by launch parity)
  // two-sided factorisation (tile mode 2, latency mode): see SftPart
  int32_t split;                  // 0: one factorisation of the whole band; 1: parts 0 / 1 + the separator problem
  int32_t sp_c0, sp_s, sp_n1p, sp_pad;   // cut: part 0 = scalars [0, c0), separator [c0, c0+s), part 1 = the rest, reversed behind sp_pad identity scalars (n1p = pad + count)
  int32_t sp_xl;                  // doubles of one exchange buffer
  int32_t pad2[2];
  SftPart part[4];
};
