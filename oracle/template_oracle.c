/*
 * template_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never on the product path).
 *
 * Restates the numbers the reference's template classes hand to the SfT solve
 * (SURVEY.md section 8, row A7):
 *
 *   regular triangulation ..... Modules/Template/TriangularMesh.cc:92-107
 *   edges + rest length ....... Modules/Template/Facet.cc:32-56, Edge.cc:29-59, Node.cc:70-75
 *   1-ring neighbours ......... Modules/Template/Node.cc:114-129
 *   Laplacian weights, boundary flags, initial mean curvature
 *                               Modules/Template/LaplacianMesh.cc:53-162
 *   "mean" (median) edge ...... Modules/Template/Template.cc:158-175
 *   barycentric embedding ..... Modules/Template/TriangularMesh.cc:133-236 (float32)
 *
 * PARITY UNPINNED: no reference test vectors exist; Eigen/OpenCV absent so the
 * reference classes cannot be compiled here.  Pointer-ordered std::set<T*>
 * containers are replaced by index / creation order (documented in DESIGN.md).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* TriangularMesh.cc:92-107 generalised to rows x cols (the reference hard-codes 10x10 and its
 * index arithmetic is only valid for square grids): node id = col + cols*row. */
int tmpl_oracle_regular_triangulation(int rows, int cols, int32_t* facets /* (rows-1)*(cols-1)*2*3 */) {
  int f = 0;
  for (int j = 0; j < rows - 1; j++)
    for (int i = 0; i < cols - 1; i++) {
      facets[3 * f + 0] = i + cols * j; facets[3 * f + 1] = i + cols * j + 1; facets[3 * f + 2] = cols * (j + 1) + i; f++;
      facets[3 * f + 0] = i + cols * j + 1; facets[3 * f + 1] = cols * (j + 1) + i; facets[3 * f + 2] = cols * (j + 1) + i + 1; f++;
    }
  return f;
}

static double node_dist(const double* a, const double* b) { /* Node.cc:70-75 */
  double d = pow(a[0] - b[0], 2) + pow(a[1] - b[1], 2) + pow(a[2] - b[2], 2);
  return sqrt(d);
}

static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

static int has_nbr(const int32_t* nbr_ptr, const int32_t* nbr_idx, int i, int j) {
  for (int p = nbr_ptr[i]; p < nbr_ptr[i + 1]; p++) if (nbr_idx[p] == j) return 1;
  return 0;
}

/*
 * Build every template constant from vertices + facets.
 * Outputs (caller allocates): edge_nodes[3F*2], edge_L0[3F], nbr_ptr[n+1], nbr_idx[6F], nbr_w[6F],
 * inc_ptr[n+1], inc_edge[6F], boundary[n], k0[n], facet_sorted[F*3], lap0[n*3] (may be NULL).
 * Returns E (number of edges); *median_L receives Template::getEdgeMeanSize().
 */
int tmpl_oracle_build(int n, const double* xyz, int F, const int32_t* facets,
                      int32_t* edge_nodes, double* edge_L0,
                      int32_t* nbr_ptr, int32_t* nbr_idx, double* nbr_w,
                      int32_t* inc_ptr, int32_t* inc_edge,
                      uint8_t* boundary, double* k0, int32_t* facet_sorted, double* lap0,
                      double* median_L) {
  int E = 0;
  /* edges in creation order: Facet.cc:45-56 checks all three pairs first, then creates (v1,v2),(v2,v3),(v1,v3) */
  for (int f = 0; f < F; f++) {
    int v[3] = {facets[3 * f], facets[3 * f + 1], facets[3 * f + 2]};
    int pr[3][2] = {{v[0], v[1]}, {v[1], v[2]}, {v[0], v[2]}};
    int rep[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++)
      for (int e = 0; e < E; e++) {
        int a = edge_nodes[2 * e], b = edge_nodes[2 * e + 1];
        if ((a == pr[k][0] && b == pr[k][1]) || (a == pr[k][1] && b == pr[k][0])) { rep[k] = 1; break; }
      }
    for (int k = 0; k < 3; k++)
      if (!rep[k]) {
        int a = pr[k][0], b = pr[k][1], dup = 0;
        for (int e = 0; e < E; e++) { /* Edge.cc:36-46 re-checks */
          int ea = edge_nodes[2 * e], eb = edge_nodes[2 * e + 1];
          if ((ea == a && eb == b) || (ea == b && eb == a)) { dup = 1; break; }
        }
        if (dup) continue;
        edge_L0[E] = node_dist(&xyz[3 * a], &xyz[3 * b]); /* distanceto(v1 -> v2) */
        edge_nodes[2 * E] = a < b ? a : b;                /* std::set<Node*> order == index order */
        edge_nodes[2 * E + 1] = a < b ? b : a;
        E++;
      }
    /* facet nodes as std::set<Node*>: ascending */
    int s0 = v[0], s1 = v[1], s2 = v[2], t;
    if (s0 > s1) { t = s0; s0 = s1; s1 = t; }
    if (s1 > s2) { t = s1; s1 = s2; s2 = t; }
    if (s0 > s1) { t = s0; s0 = s1; s1 = t; }
    facet_sorted[3 * f] = s0; facet_sorted[3 * f + 1] = s1; facet_sorted[3 * f + 2] = s2;
  }
  /* incident edges per node (creation order) */
  memset(inc_ptr, 0, sizeof(int32_t) * (n + 1));
  for (int e = 0; e < E; e++) { inc_ptr[edge_nodes[2 * e] + 1]++; inc_ptr[edge_nodes[2 * e + 1] + 1]++; }
  for (int i = 0; i < n; i++) inc_ptr[i + 1] += inc_ptr[i];
  {
    int32_t* fill = (int32_t*)calloc(n, sizeof(int32_t));
    for (int e = 0; e < E; e++)
      for (int s = 0; s < 2; s++) { int v = edge_nodes[2 * e + s]; inc_edge[inc_ptr[v] + fill[v]++] = e; }
    free(fill);
  }
  /* neighbours ascending (Node.cc:114-129 returns a std::set<Node*>) */
  nbr_ptr[0] = 0;
  for (int i = 0; i < n; i++) {
    int cnt = 0;
    int32_t* dst = &nbr_idx[nbr_ptr[i]];
    for (int p = inc_ptr[i]; p < inc_ptr[i + 1]; p++) {
      int e = inc_edge[p];
      int o = edge_nodes[2 * e] == i ? edge_nodes[2 * e + 1] : edge_nodes[2 * e];
      int q = cnt;
      while (q > 0 && dst[q - 1] > o) { dst[q] = dst[q - 1]; q--; }
      dst[q] = o; cnt++;
    }
    nbr_ptr[i + 1] = nbr_ptr[i] + cnt;
  }
  /* weights + boundary flags, LaplacianMesh.cc:55-121 */
  memset(boundary, 0, n);
  for (int p = 0; p < nbr_ptr[n]; p++) nbr_w[p] = 0.0; /* std::map::operator[] default */
  for (int i = 0; i < n; i++) {
    const double* Ni = &xyz[3 * i];
    for (int p = nbr_ptr[i]; p < nbr_ptr[i + 1]; p++) {
      int j = nbr_idx[p];
      const double* Nj = &xyz[3 * j];
      int common[2], nc = 0, ncount = 0;
      for (int q = nbr_ptr[j]; q < nbr_ptr[j + 1]; q++)
        if (has_nbr(nbr_ptr, nbr_idx, i, nbr_idx[q])) { if (nc < 2) common[nc++] = nbr_idx[q]; ncount++; }
      if (ncount == 0) {
        /* reference deletes the node (setBadFlag); not representable here: leave weight 0 */
      } else if (ncount == 1) {
        boundary[j] = 1;
      } else {
        const double* Nj1 = &xyz[3 * common[0]];
        const double* Nj_1 = &xyz[3 * common[1]];
        double a[3], b[3], c[3], cr[3];
        for (int k = 0; k < 3; k++) { a[k] = Nj_1[k] - Ni[k]; b[k] = Nj[k] - Ni[k]; c[k] = Nj1[k] - Ni[k]; }
        cr[0] = a[1] * b[2] - a[2] * b[1]; cr[1] = a[2] * b[0] - a[0] * b[2]; cr[2] = a[0] * b[1] - a[1] * b[0];
        double t1 = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) / (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
        cr[0] = c[1] * b[2] - c[2] * b[1]; cr[1] = c[2] * b[0] - c[0] * b[2]; cr[2] = c[0] * b[1] - c[1] * b[0];
        double t2 = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) / (c[0] * b[0] + c[1] * b[1] + c[2] * b[2]);
        double d[3] = {Ni[0] - Nj[0], Ni[1] - Nj[1], Ni[2] - Nj[2]};
        double wij = (tan(fabs(atan(t1)) / 2) + tan(fabs(atan(t2)) / 2)) / sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        nbr_w[p] = wij;
      }
    }
  }
  /* initial Laplacian coordinates and their norm, LaplacianMesh.cc:123-147,157-162 */
  for (int i = 0; i < n; i++) {
    k0[i] = 0.0;
    if (lap0) lap0[3 * i] = lap0[3 * i + 1] = lap0[3 * i + 2] = 0.0;
    if (boundary[i]) continue;
    if (nbr_ptr[i + 1] - nbr_ptr[i] <= 1) continue;
    double L[3] = {0, 0, 0}, sw = 0.0;
    for (int p = nbr_ptr[i]; p < nbr_ptr[i + 1]; p++) {
      const double* Nj = &xyz[3 * nbr_idx[p]];
      for (int k = 0; k < 3; k++) L[k] = L[k] + nbr_w[p] * Nj[k];
      sw = sw + nbr_w[p];
    }
    double lc[3];
    for (int k = 0; k < 3; k++) lc[k] = xyz[3 * i + k] - (L[k] / sw);
    if (lap0) { lap0[3 * i] = lc[0]; lap0[3 * i + 1] = lc[1]; lap0[3 * i + 2] = lc[2]; }
    k0[i] = sqrt(lc[0] * lc[0] + lc[1] * lc[1] + lc[2] * lc[2]);
  }
  /* Template.cc:158-175 */
  if (median_L) {
    if (E > 0) {
      double* d = (double*)malloc(sizeof(double) * E);
      memcpy(d, edge_L0, sizeof(double) * E);
      qsort(d, E, sizeof(double), cmp_double);
      *median_L = d[E / 2];
      free(d);
    } else *median_L = 0.10;
  }
  return E;
}

/* TriangularMesh.cc:207-236, float32 throughout. */
static int point_in_triangle_f32(const float q[3], const float v0[3], const float v1[3], const float v2[3], float bary[3]) {
  float u[3], v[3], nn[3], w[3], uw[3], wv[3];
  for (int k = 0; k < 3; k++) { u[k] = v1[k] - v0[k]; v[k] = v2[k] - v0[k]; w[k] = q[k] - v0[k]; }
  nn[0] = u[1] * v[2] - u[2] * v[1]; nn[1] = u[2] * v[0] - u[0] * v[2]; nn[2] = u[0] * v[1] - u[1] * v[0];
  uw[0] = u[1] * w[2] - u[2] * w[1]; uw[1] = u[2] * w[0] - u[0] * w[2]; uw[2] = u[0] * w[1] - u[1] * w[0];
  wv[0] = w[1] * v[2] - w[2] * v[1]; wv[1] = w[2] * v[0] - w[0] * v[2]; wv[2] = w[0] * v[1] - w[1] * v[0];
  float n2 = nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2];
  float gamma = (uw[0] * nn[0] + uw[1] * nn[1] + uw[2] * nn[2]) / n2;
  float beta = (wv[0] * nn[0] + wv[1] * nn[1] + wv[2] * nn[2]) / n2;
  float alpha = 1 - gamma - beta;
  bary[0] = alpha; bary[1] = beta; bary[2] = gamma;
  float d2 = 0;
  for (int k = 0; k < 3; k++) {
    float np = v0[k] * alpha + v1[k] * beta + v2[k] * gamma;
    float df = np - q[k];
    d2 += df * df;
  }
  if (d2 > 1E-1) return 0;
  return ((0 <= alpha) && (alpha <= 1) && (0 <= beta) && (beta <= 1) && (0 <= gamma) && (gamma <= 1));
}

/*
 * Barycentric embedding of P points (float32 positions) -- TriangularMesh.cc:133-200.
 * node_facet_ptr/node_facet_idx: facets incident to a node in creation (index) order.
 * Outputs: facet_id[P] (-1 if not embedded), bary[P*3] (float32 values).
 */
void tmpl_oracle_embed(int n, const double* xyz, int F, const int32_t* facet_sorted,
                       int P, const float* pts, int32_t* facet_id, float* bary) {
  (void)F;
  for (int p = 0; p < P; p++) {
    facet_id[p] = -1; bary[3 * p] = bary[3 * p + 1] = bary[3 * p + 2] = 0.f;
    const float* mp = &pts[3 * p];
    int closest = -1; double best = 100;
    for (int i = 0; i < n; i++) {
      double dist = sqrt(pow(xyz[3 * i] - mp[0], 2) + pow(xyz[3 * i + 1] - mp[1], 2) + pow(xyz[3 * i + 2] - mp[2], 2));
      if (dist < best) { closest = i; best = dist; }
    }
    if (closest < 0) continue;
    for (int f = 0; f < F; f++) {
      const int32_t* fn = &facet_sorted[3 * f];
      if (fn[0] != closest && fn[1] != closest && fn[2] != closest) continue;
      float v[3][3], b[3];
      for (int s = 0; s < 3; s++) for (int k = 0; k < 3; k++) v[s][k] = (float)xyz[3 * fn[s] + k];
      if (point_in_triangle_f32(mp, v[0], v[1], v[2], b)) {
        facet_id[p] = f; bary[3 * p] = b[0]; bary[3 * p + 1] = b[1]; bary[3 * p + 2] = b[2];
        break;
      }
    }
  }
}
