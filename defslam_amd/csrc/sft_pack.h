// Host-side packing of Shape-from-Template problems (the graph of DefOptimizer.cc:293-507 as flat arrays).
//
// The structure of the normal equations depends on the template and on WHICH nodes are optimised, not on the frame:
//   * SftGraph  -- per (template, active set): compact numbering, curvature stars, stretch edges, the 3x3 block pattern
//                  of H (one diagonal block per active node + the off-diagonal blocks of its 2-ring, lower triangle), the
//                  curvature / stretch contribution list of every block with its state-independent factors, the 16x16
//                  tile mask.  Built once, cached by the context, device-resident and SHARED by every problem of a batch
//                  (one copy stays hot in L2 for 8192 problems).
//   * SftFramePack -- per frame: observation arrays, the observation contribution list of every block, which active nodes
//                  carry a temporal (reference) edge, the initial state.  This is all the per-frame packing there is.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/defslam_hip.h"
#include "dsh_template.h"
#include "sft_problem.h"

namespace dsh {

struct SftGraph {
  std::vector<uint8_t> opt;          // key: which nodes are optimised (viewed + 1-ring, DefOptimizer.cc:388-406)
  uint64_t opt_hash = 0;
  int n = 0, nA = 0, S = 0, Es = 0, noff = 0, n_curv_ref = 0, kd = 0, max_slots = 0;
  std::vector<int32_t> act;          // n: compact index or -1
  std::vector<int32_t> actnode;      // nA
  std::vector<int32_t> star_node;    // S: optimised interior nodes (one fused curvature record per node)
  std::vector<double> star_sL;       // S: sum over incident mesh edges of 1/L^2 (DefOptimizer.cc:427-461 adds deg copies / L)
  std::vector<int32_t> str_nodes;    // Es*2
  std::vector<double> str_L0;        // Es
  std::vector<int32_t> off_ptr;      // nA+1: off-diagonal blocks of block row a, columns ascending
  std::vector<int32_t> off_rc;       // noff*2: (row, col) compact
  // Block numbering everywhere below: q < nA is the diagonal block of active node q, q >= nA the off-diagonal block q - nA.
  std::vector<int32_t> sh_ptr;       // nblk+1: curvature / stretch contributions of block q
  std::vector<uint32_t> sh_rec;      // SFT_REC(kind, slot row, slot col, edge)
  std::vector<double> sh_cf;         // 2 per contribution: H factor and b factor WITHOUT the regulariser weight
                                     //   curvature: sL * c_s * c_t, sL * c_s ; stretch: +-1, +-1
  std::vector<int32_t> tmask;        // tile mode 1: bit d of entry I = tile (I, I-d) holds an element of some block
  // tile mode 1 (kd <= 128): H lives in HBM as compact 3x3 blocks, Hc[9 * hpos(block) + 3 * row + col] (row-major, diagonal blocks
  // full), block rows interleaved: hpos(diagonal a) = a + off_ptr[a], hpos(off-diagonal q of block row a) = a + 1 + q; behind the
  // blocks one 0.0 and one 1.0.  hgather: for every 16x16 tile (I, I-d), d = 0..8, I = 0..nT (row nT: all zero), the BYTE offset into Hc
  // each (lane, register) of the MFMA accumulator layout takes -- 256 entries, entry 4 * lane + q = element (row (lane >> 4) + 4 q,
  // column lane & 15); the factorisation gathers its tiles through it, the padded tile form never exists in memory.
  std::vector<uint32_t> hgather;
  std::vector<uint32_t> hgatherT;   // the same lists for the TRANSPOSED tile (one-wavefront solver, sft_wave.h): lane (g, c), register q takes element [row c][column g + 4q]
  size_t hc_elems() const { return 9 * (size_t)(nA + noff) + 2; }
  int nblk() const { return nA + noff; }
  uint64_t last_use = 0;             // serial of the last upload that used the graph (cache eviction)
  // device copy (owned by the context)
  char* d_base = nullptr;
  size_t d_bytes = 0;
  struct { size_t act, actnode, star_node, star_sL, str_nodes, str_L0, off_ptr, off_rc, sh_ptr, sh_rec, sh_cf, tmask, hgather, hgatherT; } o{};
};

// Node degree limit of the device kernels (slot fields of SFT_REC are 4 bits: centre + 14 neighbours).
constexpr int kMaxDegree = 14;

int build_graph(const TemplateHost& t, const std::vector<uint8_t>& opt, SftGraph& g, std::string& err);

struct SftFramePack {
  int M = 0, V = 0, max_iters = 0;
  std::vector<int32_t> obs_nodes;    // M*3
  std::vector<double> obs_bary, obs_uv, obs_w;   // obs_w = invSigma2 / N_frame (DefOptimizer.cc:340)
  std::vector<int32_t> ob_ptr;       // nblk+1: observation contributions of block q (observation order)
  std::vector<int32_t> ob_m;         // observation index
  std::vector<double> ob_c;          // diagonal block: b_s ; off-diagonal block: b_s * b_t
  std::vector<uint8_t> viewed;       // nA: the node carries a reference (temporal) edge
  std::vector<double> xyz_init;      // n*3
  double pose_init[8] = {0, 0, 0, 0, 0, 0, 1, 0};
};

// Which nodes the frame optimises (viewed: a facet node of some observation; optimised: viewed + their 1-ring).
int frame_active_set(const TemplateHost& t, const dsh_sft_frame& f, std::vector<uint8_t>& viewed, std::vector<uint8_t>& opt, std::string& err);
int pack_frame(const TemplateHost& t, const SftGraph& g, const dsh_sft_frame& f, const std::vector<uint8_t>& viewed, SftFramePack& P, std::string& err);

void pose7_from_Tcw(const float* T, double* p);
void Tcw_from_pose7(const double* p, float* T);

}  // namespace dsh
