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

#define SFT_NT 512             // threads of the per-problem workgroup (8 wavefronts: 2 per SIMD, 256-VGPR budget)
#define SFT_JOBS_STRIDE 36     // doubles per observation Jacobian record
#define SFT_BORDER 7           // 6 camera rows + the right-hand side carried through the factorisation

// contribution record: kind(2) | slot_row(4) | slot_col(4) | edge(22)
#define SFT_KIND_OBS 0u
#define SFT_KIND_REF 1u
#define SFT_KIND_STAR 2u
#define SFT_KIND_STR 3u
#define SFT_REC(kind, s, t, e) (((uint32_t)(kind) << 30) | ((uint32_t)(s) << 26) | ((uint32_t)(t) << 22) | (uint32_t)(e))

struct SftDev {
  // sizes
  int32_t n, nA, Dn, kd, ldh, M, V, S, Es, nblk, max_iters, mode;
  int32_t tile_mode, jl_lds;  // 1: 16x16-tile band storage + MFMA factorisation (kd <= 128); 0: row-major band (general)
  double fx, fy, cx, cy;
  double w_ref, w_curv, w_str, hub_delta, hub_dsqr;
  // template (shared by every problem of a batch)
  const double* xyz0;
  const int32_t* nbr_ptr;
  const int32_t* nbr_idx;
  const double* nbr_w;
  const double* nbr_c;      // -(w_j / sum_j w_j)
  const double* nbr_sumw;   // per node
  const double* k0;
  // frame / graph
  const int32_t* act;       // n: compact index or -1
  const int32_t* obs_nodes; // M*3
  const double* obs_bary;   // M*3
  const double* obs_uv;     // M*2
  const double* obs_w;      // M  invSigma2 / N_frame
  const int32_t* ref_node;  // V
  const int32_t* star_node; // S
  const double* star_sL;    // S  sum over incident mesh edges of 1/L^2
  const int32_t* str_nodes; // Es*2
  const double* str_L0;     // Es
  const int32_t* blk_rc;    // nblk*2 (block row, block col), lower, sorted
  const int32_t* blk_ptr;   // nblk+1
  const int32_t* diag_blk;  // nA: block index of every diagonal block
  const int32_t* off_blk;   // nblk-nA: block indices of the off-diagonal blocks
  const uint32_t* contrib;
  // initial state (restored at the start of every run)
  const double* xyz_init;   // n*3
  const double* pose_init;  // 7: t, q(x,y,z,w)
  // state + workspace
  double* xyz;              // n*3
  double* xyz_bak;          // n*3
  double* pose;             // 7
  double* Jobs;             // M*SFT_JOBS_STRIDE
  double* Jstar;            // S*4  (u, r)
  double* Jstr;             // Es*4 (g, e)
  double* Jref;             // V*4  (e)
  double* Hb;               // band mode: Dnp*ldh lower band, row-major: (r,c) at r*ldh + c-r+kd
                            // tile mode: nT*(BT+1) 16x16 tiles, tile (I,J) at (I*(BT+1) + I-J)*256, element (row,col) at
                            //            ((row&3)*16 + col)*4 + (row>>2)  (= MFMA accumulator order: lane, register)
  double* Hbord;            // 7*Dn     rows 0-5: camera x node, row 6: b_node
  double* Hcorner;          // 7*7      camera x camera (lower) + b_cam in row 6
  double* Lb;               // Dn*ldh
  double* Lbord;            // 7*Dn
  double* Lcorner;          // 7*7
  double* Linv;             // tile mode: nT inverse diagonal tiles (16x16 row-major)
  double* x;                // Dn+6
  // outputs
  double* chi2_obs;         // M
  double* final_err;        // M  reprojection error norm at the final estimate
  double* trace;            // max_iters*8
  int32_t* info;            // [0] iters [1] trials [2] status
  double* dbg;              // [0] robust chi2 of the debug assembly
};
