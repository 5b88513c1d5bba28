// Host-side packing of Shape-from-Template problems: see sft_pack.h.
#include "sft_pack.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace dsh {

namespace {
constexpr int kNB = 32, kTS = 16;   // must match NB / TS in sft_kernels.hip
}

// ---- pose conversions at the float32 boundary (Converter.cc:35-66, se3quat.h:58-64,269-285) -------
void pose7_from_Tcw(const float* T, double* p) {
  double R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = (double)T[4 * i + j];
  double q[4];
  double tr = R[0] + R[4] + R[8];
  if (tr > 0.0) {
    double t = std::sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
  if (q[3] < 0)
    for (double& c : q) c = -c;
  const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (double& c : q) c /= nrm;
  p[0] = (double)T[3];
  p[1] = (double)T[7];
  p[2] = (double)T[11];
  p[3] = q[0];
  p[4] = q[1];
  p[5] = q[2];
  p[6] = q[3];
}

void Tcw_from_pose7(const double* p, float* T) {
  const double x = p[3], y = p[4], z = p[5], w = p[6];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[4 * i + j] = (float)R[3 * i + j];
    T[4 * i + 3] = (float)p[i];
  }
  T[12] = T[13] = T[14] = 0.f;
  T[15] = 1.f;
}

int frame_active_set(const TemplateHost& t, const dsh_sft_frame& f, std::vector<uint8_t>& viewed, std::vector<uint8_t>& opt, std::string& err) {
  const int n = t.n, M = f.M;
  if (M <= 0 || !f.obs_nodes || !f.obs_bary || !f.obs_uv || !f.obs_invsig2 || !f.xyz || !f.Tcw) { err = "empty or null frame"; return DSH_ERR_ARG; }
  if (f.n_frame <= 0 || f.max_iters < 0 || f.max_iters > DSH_MAX_ITERS) { err = "bad n_frame/max_iters"; return DSH_ERR_ARG; }
  viewed.assign(n, 0);
  for (int i = 0; i < 3 * M; i++) {
    const int32_t nd = f.obs_nodes[i];
    if (nd < 0 || nd >= n) { err = "observation node id out of range"; return DSH_ERR_ARG; }
    viewed[nd] = 1;
  }
  opt = viewed;
  if (f.neighbour_layers >= 1)  // always the 1-ring of the viewed set (DefOptimizer.cc:388-406)
    for (int i = 0; i < n; i++)
      if (viewed[i])
        for (int p = t.nbr_ptr[i]; p < t.nbr_ptr[i + 1]; p++) opt[t.nbr_idx[p]] = 1;
  return DSH_OK;
}

// The block pattern and the curvature / stretch gather lists of one (template, active set).
int build_graph(const TemplateHost& t, const std::vector<uint8_t>& opt, SftGraph& g, std::string& err) {
  const int n = t.n;
  g.opt = opt;
  g.n = n;
  g.act.assign(n, -1);
  g.actnode.clear();
  for (int i = 0; i < n; i++)
    if (opt[i]) { g.act[i] = (int)g.actnode.size(); g.actnode.push_back(i); }
  const int nA = g.nA = (int)g.actnode.size();
  // curvature "stars": the reference adds deg(i) copies of the same residual divided by the incident
  // edge lengths (DefOptimizer.cc:427-461); they are fused here into one record with sum(1/L^2).
  g.star_node.clear();
  g.star_sL.clear();
  g.n_curv_ref = 0;
  for (int i = 0; i < n; i++)
    if (opt[i] && !t.boundary[i]) {
      double s = 0.0;
      for (int p = t.inc_ptr[i]; p < t.inc_ptr[i + 1]; p++) { const double il = 1.0 / t.edge_L0[t.inc_edge[p]]; s += il * il; g.n_curv_ref++; }
      if (t.nbr_ptr[i + 1] - t.nbr_ptr[i] > kMaxDegree) { err = "node degree > 14 unsupported"; return DSH_ERR_ARG; }
      g.star_node.push_back(i);
      g.star_sL.push_back(s);
    }
  // stretch edges: mesh edges incident to an optimised node, creation order (DefOptimizer.cc:468-507)
  g.str_nodes.clear();
  g.str_L0.clear();
  for (int e = 0; e < t.E; e++) {
    const int a = t.edge_nodes[2 * e], b = t.edge_nodes[2 * e + 1];
    if (opt[a] || opt[b]) { g.str_nodes.push_back(a); g.str_nodes.push_back(b); g.str_L0.push_back(t.edge_L0[e]); }
  }
  const int S = g.S = (int)g.star_node.size(), Es = g.Es = (int)g.str_L0.size();
  if ((size_t)std::max(S, Es) >= (1u << 22)) { err = "edge index exceeds 22 bits"; return DSH_ERR_ARG; }

  // ---- block pattern: per block row the sorted block columns (< row) with their contribution counts ------------------
  struct Col { int c; int cnt; int off; };
  std::vector<std::vector<Col>> rows(nA);
  std::vector<int> dcnt(nA, 0), doff(nA, 0);
  auto touch = [&](int bi, int bj) -> Col& {
    auto& r = rows[bi];
    for (auto& cc : r)
      if (cc.c == bj) return cc;
    r.push_back({bj, 0, 0});
    return r.back();
  };
  // every contribution of the state-independent kinds, in the reference's edge order (curvature, then stretching)
  auto for_each_contrib = [&](auto&& emit) {
    for (int s = 0; s < S; s++) {
      const int nd = g.star_node[s];
      const int deg = t.nbr_ptr[nd + 1] - t.nbr_ptr[nd];
      int a[kMaxDegree + 2];
      a[0] = g.act[nd];
      for (int j = 0; j < deg; j++) a[1 + j] = g.act[t.nbr_idx[t.nbr_ptr[nd] + j]];
      for (int p = 0; p <= deg; p++)
        for (int q = 0; q <= deg; q++)
          if (a[p] >= 0 && a[q] >= 0 && (a[p] > a[q] || p == q)) emit(a[p], a[q], SFT_REC(SFT_KIND_STAR, p, q, s));
    }
    for (int e = 0; e < Es; e++) {
      const int a[2] = {g.act[g.str_nodes[2 * e]], g.act[g.str_nodes[2 * e + 1]]};
      for (int p = 0; p < 2; p++)
        for (int q = 0; q < 2; q++)
          if (a[p] >= 0 && a[q] >= 0 && (a[p] > a[q] || p == q)) emit(a[p], a[q], SFT_REC(SFT_KIND_STR, p, q, e));
    }
  };
  for_each_contrib([&](int bi, int bj, uint32_t) { if (bi == bj) dcnt[bi]++; else touch(bi, bj).cnt++; });
  int noff = 0, bwn = 0;
  size_t total = 0;
  for (int a = 0; a < nA; a++) { doff[a] = (int)total; total += dcnt[a]; dcnt[a] = 0; }
  g.off_ptr.assign(nA + 1, 0);
  for (int a = 0; a < nA; a++) {
    std::sort(rows[a].begin(), rows[a].end(), [](const Col& x, const Col& y) { return x.c < y.c; });
    g.off_ptr[a] = noff;
    for (auto& cc : rows[a]) { cc.off = (int)total; total += cc.cnt; cc.cnt = 0; noff++; bwn = std::max(bwn, a - cc.c); }
  }
  g.off_ptr[nA] = noff;
  g.noff = noff;
  if (total >= (1u << 22) * 64ull) { err = "too many contributions"; return DSH_ERR_ARG; }
  g.sh_rec.assign(total, 0u);
  g.sh_cf.assign(2 * total, 0.0);
  auto slot_of = [&](int bi, int bj) -> int {   // position of the next contribution of block (bi, bj)
    if (bi == bj) return doff[bi] + dcnt[bi]++;
    auto& r = rows[bi];
    int lo = 0, hi = (int)r.size() - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (r[mid].c < bj) lo = mid + 1; else hi = mid; }
    return r[lo].off + r[lo].cnt++;
  };
  for_each_contrib([&](int bi, int bj, uint32_t rec) {
    const int p = slot_of(bi, bj);
    g.sh_rec[p] = rec;
    const uint32_t kind = rec >> 30, s = (rec >> 26) & 15u, u = (rec >> 22) & 15u, e = rec & 0x3FFFFFu;
    if (kind == SFT_KIND_STAR) {
      const int base = t.nbr_ptr[g.star_node[e]];
      const double cs = (s == 0) ? 1.0 : t.nbr_c[base + s - 1];
      const double ct = (u == 0) ? 1.0 : t.nbr_c[base + u - 1];
      g.sh_cf[2 * p] = g.star_sL[e] * (cs * ct);
      g.sh_cf[2 * p + 1] = g.star_sL[e] * cs;
    } else {
      g.sh_cf[2 * p] = ((s == 0) == (u == 0)) ? 1.0 : -1.0;
      g.sh_cf[2 * p + 1] = (s == 0) ? 1.0 : -1.0;
    }
  });
  g.off_rc.resize(2 * (size_t)noff);
  g.sh_ptr.assign((size_t)nA + noff + 1, 0);
  for (int a = 0; a < nA; a++) g.sh_ptr[a] = doff[a];
  {
    int q = 0;
    for (int a = 0; a < nA; a++)
      for (auto& cc : rows[a]) { g.off_rc[2 * q] = a; g.off_rc[2 * q + 1] = cc.c; g.sh_ptr[nA + q] = cc.off; q++; }
  }
  g.sh_ptr[(size_t)nA + noff] = (int)total;
  // diagonal lists were laid out first, the off-diagonal ones behind them in row order: sh_ptr is monotone
  g.kd = 3 * bwn + 2;
  {  // which 16x16 tiles of the band hold an element of some 3x3 block (30 % of the C2 band is structurally zero)
    const int Dn_ = 3 * nA, nT_ = ((Dn_ + kNB - 1) / kNB) * kNB / kTS;
    g.tmask.assign((size_t)nT_ + SFT_H_PAD_TILE_ROWS, 0);
    for (int I = 0; I < nT_; I++) g.tmask[I] = 1;   // diagonal tiles (incl. the identity padding of the last one)
    auto mark = [&](int bi, int bj) {
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          const int r = 3 * bi + a, cc = 3 * bj + b;
          if (cc > r) continue;
          const int d = (r >> 4) - (cc >> 4);
          if (d < 31) g.tmask[r >> 4] |= 1 << d;
        }
    };
    for (int a = 0; a < nA; a++) mark(a, a);
    for (int q = 0; q < noff; q++) mark(g.off_rc[2 * q], g.off_rc[2 * q + 1]);
    g.max_slots = 0;
    for (int I = 0; I < nT_; I++) g.max_slots = std::max(g.max_slots, __builtin_popcount((unsigned)g.tmask[I]));
    g.hgather.clear();
    g.hgatherT.clear();
    if (g.kd <= kTS * 8) {   // register-window solver: the gather lists of its tiles (sft_pack.h)
      if (9 * (size_t)(nA + noff) + 2 >= (1u << 28)) { err = "template too large for the 32-bit gather offsets"; return DSH_ERR_ARG; }
      const uint32_t ZERO = (uint32_t)(8 * 9 * (size_t)(nA + noff)), ONE = ZERO + 8;   // byte offsets into Hc
      g.hgather.assign((size_t)(nT_ + 1) * 9 * 256, ZERO);
      g.hgatherT.assign((size_t)(nT_ + 1) * 9 * 256, ZERO);
      auto elem = [&](int r, int c) -> uint32_t {   // element (r, c) of the symmetric matrix, r, c < Dn_
        int bi = r / 3, bj = c / 3, er = r % 3, ec = c % 3;
        if (bj > bi) { std::swap(bi, bj); std::swap(er, ec); }
        if (bi == bj) return (uint32_t)(8 * (9 * (size_t)(bi + g.off_ptr[bi]) + 3 * er + ec));
        int lo = g.off_ptr[bi], hi = g.off_ptr[bi + 1] - 1;
        while (lo <= hi) {
          const int mid = (lo + hi) >> 1, cm = g.off_rc[2 * mid + 1];
          if (cm == bj) return (uint32_t)(8 * (9 * (size_t)(bi + 1 + mid) + 3 * er + ec));
          if (cm < bj) lo = mid + 1; else hi = mid - 1;
        }
        return ZERO;
      };
      for (int I = 0; I < nT_; I++)
        for (int d = 0; d <= 8 && d <= I; d++) {
          if (!((g.tmask[I] >> d) & 1)) continue;
          uint32_t* tl = g.hgather.data() + ((size_t)I * 9 + d) * 256;
          for (int l = 0; l < 64; l++)
            for (int q = 0; q < 4; q++) {
              const int r = kTS * I + (l >> 4) + 4 * q, c = kTS * (I - d) + (l & 15);
              tl[4 * l + q] = (r >= Dn_ || c >= Dn_) ? (r == c ? ONE : ZERO) : elem(r, c);
            }
          uint32_t* tt = g.hgatherT.data() + ((size_t)I * 9 + d) * 256;
          for (int l = 0; l < 64; l++)
            for (int q = 0; q < 4; q++) {
              const int r = kTS * I + (l & 15), c = kTS * (I - d) + (l >> 4) + 4 * q;
              tt[4 * l + q] = (r >= Dn_ || c >= Dn_) ? (r == c ? ONE : ZERO) : elem(r, c);
            }
        }
    }
  }
  uint64_t h = 1469598103934665603ull;
  for (uint8_t b : opt) { h ^= b; h *= 1099511628211ull; }
  g.opt_hash = h;
  return DSH_OK;
}

int pack_frame(const TemplateHost& t, const SftGraph& g, const dsh_sft_frame& f, const std::vector<uint8_t>& viewed, SftFramePack& P, std::string& err) {
  const int n = t.n, M = f.M, nA = g.nA, nblk = g.nblk();
  P.M = M;
  P.max_iters = f.max_iters;
  P.obs_nodes.assign(f.obs_nodes, f.obs_nodes + 3 * (size_t)M);
  P.obs_bary.assign(f.obs_bary, f.obs_bary + 3 * (size_t)M);
  P.obs_uv.assign(f.obs_uv, f.obs_uv + 2 * (size_t)M);
  P.obs_w.resize(M);
  for (int i = 0; i < M; i++) P.obs_w[i] = f.obs_invsig2[i] / (double)f.n_frame;  // DefOptimizer.cc:340
  P.viewed.assign(nA, 0);
  P.V = 0;
  for (int a = 0; a < nA; a++)
    if (viewed[g.actnode[a]]) { P.viewed[a] = 1; P.V++; }
  // ---- observation contributions per block: count, prefix, fill in observation order -------------------------------
  P.ob_ptr.assign((size_t)nblk + 1, 0);
  std::vector<int32_t> qid(6 * (size_t)M);   // block of each of the 6 (slot, slot) pairs of an observation, -1: none
  for (int m = 0; m < M; m++) {
    int a[3];
    for (int s = 0; s < 3; s++) a[s] = g.act[P.obs_nodes[3 * (size_t)m + s]];
    int k = 0;
    for (int s = 0; s < 3; s++)
      for (int u = 0; u <= s; u++, k++) {
        // (s, s): the diagonal block of node s; (s, u): the block (max, min) -- node ids are ascending inside an observation,
        // so are the compact indices; a repeated node id contributes its diagonal block only
        int q = -1;
        if (s == u) q = a[s];
        else if (a[s] != a[u]) {
          const int bi = std::max(a[s], a[u]), bj = std::min(a[s], a[u]);
          int lo = g.off_ptr[bi], hi = g.off_ptr[bi + 1] - 1;
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (g.off_rc[2 * mid + 1] < bj) lo = mid + 1; else hi = mid; }
          if (lo > hi || g.off_rc[2 * lo + 1] != bj) { err = "observation " + std::to_string(m) + ": its nodes are not joined by a mesh edge of the template"; return DSH_ERR_ARG; }
          q = nA + lo;
        }
        qid[6 * (size_t)m + k] = q;
        if (q >= 0) P.ob_ptr[q + 1]++;
      }
  }
  for (int q = 0; q < nblk; q++) P.ob_ptr[q + 1] += P.ob_ptr[q];
  const size_t total = (size_t)P.ob_ptr[nblk];
  P.ob_m.resize(total);
  P.ob_c.resize(total);
  std::vector<int32_t> fill(P.ob_ptr.begin(), P.ob_ptr.end() - 1);
  for (int m = 0; m < M; m++) {
    const double* bb = &P.obs_bary[3 * (size_t)m];
    int k = 0;
    for (int s = 0; s < 3; s++)
      for (int u = 0; u <= s; u++, k++) {
        const int q = qid[6 * (size_t)m + k];
        if (q < 0) continue;
        const int p = fill[q]++;
        P.ob_m[p] = m;
        P.ob_c[p] = (s == u) ? bb[s] : bb[s] * bb[u];
      }
  }
  P.xyz_init.assign(f.xyz, f.xyz + 3 * (size_t)n);
  pose7_from_Tcw(f.Tcw, P.pose_init);
  return DSH_OK;
}

}  // namespace dsh
