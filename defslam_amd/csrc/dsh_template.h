// Host-side template constants (the numbers the reference's Modules/Template classes
// hand to the SfT solve -- SURVEY.md section 8a, row A7).
#pragma once
#include <cstdint>
#include <vector>

namespace dsh {

struct TemplateHost {
  int n = 0, E = 0, F = 0;
  std::vector<double> xyz0;          // n*3 rest positions (Node::xO,yO,zO)
  std::vector<uint8_t> boundary;     // n
  std::vector<int32_t> nbr_ptr;      // n+1, 1-ring in ascending node index
  std::vector<int32_t> nbr_idx;
  std::vector<double> nbr_w;         // Laplacian weights w_ij
  std::vector<double> nbr_c;         // -(w_ij / sum_j w_ij)
  std::vector<double> nbr_sumw;      // n
  std::vector<double> k0;            // n initial mean-curvature norm
  std::vector<int32_t> edge_nodes;   // E*2 (lo, hi), creation order
  std::vector<double> edge_L0;       // E rest lengths
  std::vector<int32_t> inc_ptr;      // n+1: mesh edges incident to a node, creation order
  std::vector<int32_t> inc_edge;
  std::vector<int32_t> facets;       // F*3 ascending node ids (std::set<Node*> order); empty if set directly
  std::vector<int32_t> nf_ptr, nf_idx;  // facets incident to a node (index order)
  double median_L = 0.10;
  bool valid = false;

  // Derive everything from vertices + facets (LaplacianMesh.cc:53-162, Facet.cc:32-56, Edge.cc:29-59,
  // Node.cc:114-129, Template.cc:158-175).
  void build(int n_, const double* xyz, int F_, const int32_t* fac);
  // Adopt constants computed elsewhere (a shim that still owns the reference's Template object).
  void set(int n_, const double* xyz, const uint8_t* bnd, const int32_t* rowptr, const int32_t* col, const double* w,
           const double* k0_, int E_, const int32_t* en, const double* eL, double med);
  // Barycentric embedding, float32 like TriangularMesh.cc:133-236.
  void embed(int P, const float* pts, int32_t* facet_id, int32_t* nodes, float* bary) const;

 private:
  void finish_derived();
};

}  // namespace dsh
