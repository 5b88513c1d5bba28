// Host-side template constants: C++ counterpart of what Modules/Template computes once per
// template in the reference.  Containers the reference keeps in
// pointer-ordered std::set<T*> are kept in index / creation order here.
#include "dsh_template.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <utility>

namespace dsh {

namespace {
struct V3 {
  double x, y, z;
};
inline V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
}  // namespace

void TemplateHost::build(int n_, const double* xyz, int F_, const int32_t* fac) {
  n = n_;
  F = F_;
  xyz0.assign(xyz, xyz + 3 * (size_t)n);
  auto P = [&](int i) { return V3{xyz0[3 * i], xyz0[3 * i + 1], xyz0[3 * i + 2]}; };

  // --- mesh edges in creation order (Facet.cc:45-56: (v1,v2),(v2,v3),(v1,v3) unless already present)
  std::map<std::pair<int, int>, int> edge_id;
  edge_nodes.clear();
  edge_L0.clear();
  facets.resize(3 * (size_t)F);
  for (int f = 0; f < F; f++) {
    const int v[3] = {fac[3 * f], fac[3 * f + 1], fac[3 * f + 2]};
    const int pairs[3][2] = {{v[0], v[1]}, {v[1], v[2]}, {v[0], v[2]}};
    for (auto& pr : pairs) {
      const int lo = std::min(pr[0], pr[1]), hi = std::max(pr[0], pr[1]);
      if (edge_id.count({lo, hi})) continue;
      edge_id[{lo, hi}] = (int)edge_L0.size();
      edge_nodes.push_back(lo);
      edge_nodes.push_back(hi);
      // Node::distanceto: sqrt(pow(dx,2)+pow(dy,2)+pow(dz,2)) evaluated from the first node of the pair
      const V3 a = P(pr[0]), b = P(pr[1]);
      const double d = std::pow(a.x - b.x, 2) + std::pow(a.y - b.y, 2) + std::pow(a.z - b.z, 2);
      edge_L0.push_back(std::sqrt(d));
    }
    int s[3] = {v[0], v[1], v[2]};
    std::sort(s, s + 3);
    facets[3 * f] = s[0];
    facets[3 * f + 1] = s[1];
    facets[3 * f + 2] = s[2];
  }
  E = (int)edge_L0.size();

  // --- incidence + 1-ring
  std::vector<std::vector<int>> inc(n), ring(n);
  for (int e = 0; e < E; e++) {
    inc[edge_nodes[2 * e]].push_back(e);
    inc[edge_nodes[2 * e + 1]].push_back(e);
    ring[edge_nodes[2 * e]].push_back(edge_nodes[2 * e + 1]);
    ring[edge_nodes[2 * e + 1]].push_back(edge_nodes[2 * e]);
  }
  inc_ptr.assign(n + 1, 0);
  nbr_ptr.assign(n + 1, 0);
  inc_edge.clear();
  nbr_idx.clear();
  for (int i = 0; i < n; i++) {
    std::sort(ring[i].begin(), ring[i].end());
    inc_edge.insert(inc_edge.end(), inc[i].begin(), inc[i].end());
    nbr_idx.insert(nbr_idx.end(), ring[i].begin(), ring[i].end());
    inc_ptr[i + 1] = (int)inc_edge.size();
    nbr_ptr[i + 1] = (int)nbr_idx.size();
  }

  // --- Laplacian weights and boundary flags (LaplacianMesh.cc:55-121)
  boundary.assign(n, 0);
  nbr_w.assign(nbr_idx.size(), 0.0);
  for (int i = 0; i < n; i++) {
    const V3 Ni = P(i);
    for (int p = nbr_ptr[i]; p < nbr_ptr[i + 1]; p++) {
      const int j = nbr_idx[p];
      // neighbours of j that are also neighbours of i, in index order
      std::vector<int> common;
      std::set_intersection(ring[j].begin(), ring[j].end(), ring[i].begin(), ring[i].end(), std::back_inserter(common));
      if (common.size() == 1) {
        boundary[j] = 1;  // the reference flags the NEIGHBOUR (LaplacianMesh.cc:90-93)
      } else if (common.size() >= 2) {
        const V3 Nj = P(j), Nj1 = P(common[0]), Nj_1 = P(common[1]);
        const V3 eij = sub(Nj, Ni);
        const V3 a = sub(Nj_1, Ni), c = sub(Nj1, Ni);
        const double t1 = norm(cross(a, eij)) / dot(a, eij);
        const double t2 = norm(cross(c, eij)) / dot(c, eij);
        nbr_w[p] = (std::tan(std::fabs(std::atan(t1)) / 2) + std::tan(std::fabs(std::atan(t2)) / 2)) / norm(sub(Ni, Nj));
      }
      // no common neighbour: the reference deletes node j; we keep it with weight 0
    }
  }
  // --- initial mean curvature (LaplacianMesh.cc:123-147,157-162)
  k0.assign(n, 0.0);
  for (int i = 0; i < n; i++) {
    if (boundary[i] || nbr_ptr[i + 1] - nbr_ptr[i] <= 1) continue;
    double L[3] = {0, 0, 0}, sw = 0.0;
    for (int p = nbr_ptr[i]; p < nbr_ptr[i + 1]; p++) {
      const int j = nbr_idx[p];
      for (int k = 0; k < 3; k++) L[k] = L[k] + nbr_w[p] * xyz0[3 * j + k];
      sw = sw + nbr_w[p];
    }
    double lc[3];
    for (int k = 0; k < 3; k++) lc[k] = xyz0[3 * i + k] - (L[k] / sw);
    k0[i] = std::sqrt(lc[0] * lc[0] + lc[1] * lc[1] + lc[2] * lc[2]);
  }
  // --- Template::getEdgeMeanSize returns the median (Template.cc:158-175)
  if (E > 0) {
    std::vector<double> d(edge_L0);
    std::sort(d.begin(), d.end());
    median_L = d[d.size() / 2];
  } else {
    median_L = 0.10;
  }
  finish_derived();
}

void TemplateHost::set(int n_, const double* xyz, const uint8_t* bnd, const int32_t* rowptr, const int32_t* col, const double* w,
                       const double* k0_, int E_, const int32_t* en, const double* eL, double med) {
  n = n_;
  E = E_;
  F = 0;
  facets.clear();
  xyz0.assign(xyz, xyz + 3 * (size_t)n);
  boundary.assign(bnd, bnd + n);
  nbr_ptr.assign(rowptr, rowptr + n + 1);
  nbr_idx.assign(col, col + rowptr[n]);
  nbr_w.assign(w, w + rowptr[n]);
  k0.assign(k0_, k0_ + n);
  edge_nodes.assign(en, en + 2 * (size_t)E);
  edge_L0.assign(eL, eL + E);
  median_L = med;
  std::vector<std::vector<int>> inc(n);
  for (int e = 0; e < E; e++) {
    inc[edge_nodes[2 * e]].push_back(e);
    inc[edge_nodes[2 * e + 1]].push_back(e);
  }
  inc_ptr.assign(n + 1, 0);
  inc_edge.clear();
  for (int i = 0; i < n; i++) {
    inc_edge.insert(inc_edge.end(), inc[i].begin(), inc[i].end());
    inc_ptr[i + 1] = (int)inc_edge.size();
  }
  finish_derived();
}

void TemplateHost::finish_derived() {
  nbr_sumw.assign(n, 0.0);
  nbr_c.assign(nbr_idx.size(), 0.0);
  for (int i = 0; i < n; i++) {
    double sw = 0.0;  // same left-to-right sum EdgeMeanCurvature::computeError forms (sft_types.h:275-281)
    for (int p = nbr_ptr[i]; p < nbr_ptr[i + 1]; p++) sw = sw + nbr_w[p];
    nbr_sumw[i] = sw;
    for (int p = nbr_ptr[i]; p < nbr_ptr[i + 1]; p++) nbr_c[p] = -(nbr_w[p] / sw);
  }
  nf_ptr.assign(n + 1, 0);
  nf_idx.clear();
  if (F > 0) {
    std::vector<std::vector<int>> nf(n);
    for (int f = 0; f < F; f++)
      for (int s = 0; s < 3; s++) nf[facets[3 * f + s]].push_back(f);
    for (int i = 0; i < n; i++) {
      nf_idx.insert(nf_idx.end(), nf[i].begin(), nf[i].end());
      nf_ptr[i + 1] = (int)nf_idx.size();
    }
  }
  valid = true;
}

namespace {
// TriangularMesh::pointInTriangle (TriangularMesh.cc:207-236), float32 arithmetic.
bool point_in_triangle(const float q[3], const float v0[3], const float v1[3], const float v2[3], float bary[3]) {
  float u[3], v[3], w[3];
  for (int k = 0; k < 3; k++) {
    u[k] = v1[k] - v0[k];
    v[k] = v2[k] - v0[k];
    w[k] = q[k] - v0[k];
  }
  const float nx = u[1] * v[2] - u[2] * v[1], ny = u[2] * v[0] - u[0] * v[2], nz = u[0] * v[1] - u[1] * v[0];
  const float ax = u[1] * w[2] - u[2] * w[1], ay = u[2] * w[0] - u[0] * w[2], az = u[0] * w[1] - u[1] * w[0];
  const float bx = w[1] * v[2] - w[2] * v[1], by = w[2] * v[0] - w[0] * v[2], bz = w[0] * v[1] - w[1] * v[0];
  const float n2 = nx * nx + ny * ny + nz * nz;
  const float gamma = (ax * nx + ay * ny + az * nz) / n2;
  const float beta = (bx * nx + by * ny + bz * nz) / n2;
  const float alpha = 1 - gamma - beta;
  bary[0] = alpha;
  bary[1] = beta;
  bary[2] = gamma;
  float d2 = 0;
  for (int k = 0; k < 3; k++) {
    const float proj = v0[k] * alpha + v1[k] * beta + v2[k] * gamma;
    const float df = proj - q[k];
    d2 += df * df;
  }
  if (d2 > 1E-1) return false;
  return (0 <= alpha) && (alpha <= 1) && (0 <= beta) && (beta <= 1) && (0 <= gamma) && (gamma <= 1);
}
}  // namespace

void TemplateHost::embed(int P, const float* pts, int32_t* facet_id, int32_t* nodes, float* bary) const {
  for (int p = 0; p < P; p++) {
    facet_id[p] = -1;
    for (int k = 0; k < 3; k++) {
      nodes[3 * p + k] = -1;
      bary[3 * p + k] = 0.f;
    }
    const float* mp = pts + 3 * p;
    int closest = -1;
    double best = 100;  // TriangularMesh.cc:152
    for (int i = 0; i < n; i++) {
      const double dist = std::sqrt(std::pow(xyz0[3 * i] - mp[0], 2) + std::pow(xyz0[3 * i + 1] - mp[1], 2) + std::pow(xyz0[3 * i + 2] - mp[2], 2));
      if (dist < best) {
        closest = i;
        best = dist;
      }
    }
    if (closest < 0) continue;
    for (int q = nf_ptr[closest]; q < nf_ptr[closest + 1]; q++) {
      const int f = nf_idx[q];
      float v[3][3], b[3];
      for (int s = 0; s < 3; s++)
        for (int k = 0; k < 3; k++) v[s][k] = (float)xyz0[3 * facets[3 * f + s] + k];
      if (point_in_triangle(mp, v[0], v[1], v[2], b)) {
        facet_id[p] = f;
        for (int k = 0; k < 3; k++) {
          nodes[3 * p + k] = facets[3 * f + k];
          bary[3 * p + k] = b[k];
        }
        break;
      }
    }
  }
}

}  // namespace dsh
