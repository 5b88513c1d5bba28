/*
 * sft_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never on the product path).
 *
 * Plain-C, single-thread, FP64 restatement of the reference's Shape-from-Template
 * solve, i.e. what defSLAM::Optimizer::DefPoseOptimization does between building
 * the g2o graph and writing the result back:
 *
 *   graph construction ....... Modules/Tracking/DefOptimizer.cc:251-513
 *   the four residual types .. Thirdparty/g2o/g2o/types/sft_types.h:75-411
 *   quadratic forms .......... Thirdparty/g2o/g2o/core/base_multi_edge.hpp:36-48,171-222
 *                              base_binary_edge.hpp:57-131, base_unary_edge.hpp:43-72
 *   Huber kernel ............. Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:78-91
 *   Levenberg-Marquardt ...... Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189
 *   outer loop ............... Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:403-475
 *   index mapping ............ Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:181-211
 *   dense solve .............. Thirdparty/g2o/g2o/solvers/linear_solver_dense.h:65-113
 *   SE3 ...................... Thirdparty/g2o/g2o/types/se3quat.h:58-64,104-121,217-285
 *   vertex updates ........... types_six_dof_expmap.h:73-76, types_sba.h:52-56
 *   float32 pose boundary .... Thirdparty/ORBSLAM_2/src/Converter.cc:35-66
 *   classification/stats ..... Modules/Tracking/DefOptimizer.cc:515-577
 *
 * PARITY UNPINNED: the reference ships no test vectors for this path and g2o
 * cannot be compiled in this image (Eigen is absent).  Eigen (version unpinned
 * by the reference, >=3.3) supplies Quaterniond(Matrix3d), Quaternion*vector,
 * toRotationMatrix and LDLT<MatrixXd>; their published algorithms are restated
 * below.  This file is cross-checked against an independent NumPy restatement
 * (oracle/sft_oracle_np.py) and against closed-form known answers in tests/.
 *
 * Canonical ordering (the reference orders by std::set<T*> pointer value, which
 * is allocation order in practice): nodes by index, facets' nodes ascending,
 * mesh edges in creation order (facet order, (v1,v2),(v2,v3),(v1,v3)).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define SFT_TRACE_STRIDE 8 /* per outer iteration: chi_start, lambda_start, trials, chi_end, lambda_end, rho, accepted, reserved */

/* ------------------------------------------------------------------ */
/* Small fixed-size algebra                                            */
/* ------------------------------------------------------------------ */
#include "small_algebra.h"

/* ------------------------------------------------------------------ */
/* Dense LDLT                                                          */
/* ------------------------------------------------------------------ */
/* mode 0: Eigen::LDLT restated (unblocked, diagonal pivoting, column-major,
 *         in place, lower).  Returns 1 if "isPositive()" (no negative pivot).
 * mode 1: unpivoted blocked LDL^T (same maths for SPD input, different
 *         rounding) -- used for large sizes / the CPU baseline. */
/* Unpivoted right-looking blocked LDL^T, column-major lower. Returns 1 if all pivots > 0. */
static int ldlt_blocked(double* A, int n, double* work /* n*NBK */) {
  enum { NBK = 48 };
  int ok = 1;
  for (int j0 = 0; j0 < n; j0 += NBK) {
    int nb = n - j0 < NBK ? n - j0 : NBK;
    /* panel factorisation (unblocked, left-looking inside the panel) */
    for (int k = j0; k < j0 + nb; k++) {
      for (int j = j0; j < k; j++) {
        double t = A[j + (size_t)j * n] * A[k + (size_t)j * n];
        const double* cj = &A[(size_t)j * n];
        double* ck = &A[(size_t)k * n];
        for (int i = k; i < n; i++) ck[i] -= cj[i] * t;
      }
      double d = A[k + (size_t)k * n];
      if (!(d > 0)) ok = 0;
      double* ck = &A[(size_t)k * n];
      for (int i = k + 1; i < n; i++) ck[i] /= d;
    }
    /* trailing update: A22 -= L21 * D * L21^T ; work = L21*D (rows j0+nb.., cols nb) */
    int r0 = j0 + nb, m = n - r0;
    if (m <= 0) break;
    for (int j = 0; j < nb; j++) {
      double d = A[(j0 + j) + (size_t)(j0 + j) * n];
      const double* c = &A[r0 + (size_t)(j0 + j) * n];
      double* w = &work[(size_t)j * m];
      for (int i = 0; i < m; i++) w[i] = c[i] * d;
    }
    for (int c = 0; c < m; c++) {
      double* dst = &A[(r0 + c) + (size_t)(r0 + c) * n];
      int len = m - c;
      for (int j = 0; j < nb; j++) {
        double lcj = A[(r0 + c) + (size_t)(j0 + j) * n];
        const double* w = &work[(size_t)j * m + c];
        for (int i = 0; i < len; i++) dst[i] -= w[i] * lcj;
      }
    }
  }
  return ok;
}

static void ldlt_blocked_solve(const double* A, int n, const double* b, double* x) {
  for (int i = 0; i < n; i++) x[i] = b[i];
  for (int j = 0; j < n; j++) {
    double xj = x[j];
    const double* col = &A[(size_t)j * n];
    for (int i = j + 1; i < n; i++) x[i] -= col[i] * xj;
  }
  for (int i = 0; i < n; i++) x[i] /= A[i + (size_t)i * n];
  for (int j = n - 1; j >= 0; j--) {
    const double* col = &A[(size_t)j * n];
    double s = x[j];
    for (int i = j + 1; i < n; i++) s -= col[i] * x[i];
    x[j] = s;
  }
}

/* exported for unit tests of the factorisations */
int sft_oracle_ldlt_solve(int mode, int n, const double* A_colmajor, const double* b, double* x) {
  double* M = (double*)malloc(sizeof(double) * (size_t)n * n);
  memcpy(M, A_colmajor, sizeof(double) * (size_t)n * n);
  int ok;
  if (mode == 0) {
    int* perm = (int*)malloc(sizeof(int) * n);
    double* tmp = (double*)malloc(sizeof(double) * n);
    ok = ldlt_pivoted(M, n, perm, tmp);
    if (ok) ldlt_pivoted_solve(M, n, perm, b, x);
    free(perm); free(tmp);
  } else {
    double* work = (double*)malloc(sizeof(double) * (size_t)n * 48);
    ok = ldlt_blocked(M, n, work);
    if (ok) ldlt_blocked_solve(M, n, b, x);
    free(work);
  }
  free(M);
  return ok;
}

/* ------------------------------------------------------------------ */
/* Graph                                                               */
/* ------------------------------------------------------------------ */
enum { EK_OBS = 0, EK_REF = 1, EK_CURV = 2, EK_STRETCH = 3 };

typedef struct {
  int kind;
  int nv;          /* number of vertices (g2o sense) */
  int v[8];        /* vertex ids: 0 = camera, 1+i = node i; up to 1+7 */
  int* vext;       /* used when nv > 8 (curvature with many neighbours) */
  double info;     /* scalar information (all Omegas here are scalar * I) */
  double meas[3];
  double bary[3];
  const double* w; /* curvature weights (nv-1) */
  double L;        /* curvature: incident edge length; */
  double err[3];   /* last computed error (g2o _error) */
  /* cached by computeError of the curvature edge (sft_types.h:257-291) */
  double mc[3], mcn, sumw;
} edge_t;

typedef struct {
  int n;
  const double* xyz0;
  se3_t cam;
  double* xyz;     /* current estimates n*3 */
  double fx, fy, cx, cy;
  int ne;
  edge_t* e;
  int* hidx;       /* per vertex id (0..n): hessian scalar offset or -1 if fixed */
  int D;
  double huber_delta, huber_dsqr;
} graph_t;

static const int* edge_verts(const edge_t* e) { return e->vext ? e->vext : e->v; }

/* computeError for every kind */
static void edge_error(graph_t* g, edge_t* e) {
  const int* v = edge_verts(e);
  switch (e->kind) {
    case EK_OBS: { /* sft_types.h:102-133 */
      const double* p1 = &g->xyz[3 * (v[1] - 1)];
      const double* p2 = &g->xyz[3 * (v[2] - 1)];
      const double* p3 = &g->xyz[3 * (v[3] - 1)];
      double pw[3], pc[3];
      for (int k = 0; k < 3; k++) pw[k] = (e->bary[0] * p1[k] + e->bary[1] * p2[k]) + e->bary[2] * p3[k];
      se3_map(&g->cam, pw, pc);
      double r0 = pc[0] / pc[2], r1 = pc[1] / pc[2];
      e->err[0] = e->meas[0] - (r0 * g->fx + g->cx);
      e->err[1] = e->meas[1] - (r1 * g->fy + g->cy);
    } break;
    case EK_REF: { /* sft_types.h:401-406 */
      const double* p = &g->xyz[3 * (v[0] - 1)];
      for (int k = 0; k < 3; k++) e->err[k] = p[k] - e->meas[k];
    } break;
    case EK_CURV: { /* sft_types.h:257-291 */
      const double* ni = &g->xyz[3 * (v[0] - 1)];
      double acc[3] = {0, 0, 0}, sw = 0.0;
      for (int j = 1; j < e->nv; j++) {
        const double* nj = &g->xyz[3 * (v[j] - 1)];
        double wj = e->w[j - 1];
        for (int k = 0; k < 3; k++) acc[k] = acc[k] + wj * nj[k];
        sw = sw + wj;
      }
      e->sumw = sw;
      for (int k = 0; k < 3; k++) e->mc[k] = ni[k] - acc[k] / sw;
      e->mcn = sqrt(e->mc[0] * e->mc[0] + e->mc[1] * e->mc[1] + e->mc[2] * e->mc[2]);
      e->err[0] = (e->mcn - e->meas[0]) / e->L;
    } break;
    case EK_STRETCH: { /* sft_types.h:351-361 */
      const double* a = &g->xyz[3 * (v[0] - 1)];
      const double* b = &g->xyz[3 * (v[1] - 1)];
      double d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
      double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      e->err[0] = nrm * (1.0 / e->meas[0]) - 1.0;
    } break;
  }
}

static int edge_dim(const edge_t* e) { return e->kind == EK_OBS ? 2 : (e->kind == EK_REF ? 3 : 1); }

/* base_edge.h:58-61 with Omega = info * I */
static double edge_chi2(const edge_t* e) {
  int d = edge_dim(e);
  double s = 0;
  for (int k = 0; k < d; k++) s += e->err[k] * (e->info * e->err[k]);
  return s;
}

/* robust_kernel_impl.cpp:78-91 */
static void huber(const graph_t* g, double e2, double rho[3]) {
  if (e2 <= g->huber_dsqr) { rho[0] = e2; rho[1] = 1.; rho[2] = 0.; }
  else {
    double sq = sqrt(e2);
    rho[0] = 2 * sq * g->huber_delta - g->huber_dsqr;
    rho[1] = g->huber_delta / sq;
    rho[2] = -0.5 * rho[1] / e2;
  }
}

/* sparse_optimizer.cpp:61-91,104-120 */
static void compute_active_errors(graph_t* g) { for (int i = 0; i < g->ne; i++) edge_error(g, &g->e[i]); }

static double active_robust_chi2(const graph_t* g) {
  double chi = 0.0, rho[3];
  for (int i = 0; i < g->ne; i++) {
    const edge_t* e = &g->e[i];
    if (e->kind == EK_OBS) { huber(g, edge_chi2(e), rho); chi += rho[0]; }
    else chi += edge_chi2(e);
  }
  return chi;
}

/* Jacobian blocks of one edge: J[vertex slot] is dim x vdim, row-major, max 2x6. */
static void edge_linearize(graph_t* g, edge_t* e, double (*J)[12]) {
  const int* v = edge_verts(e);
  switch (e->kind) {
    case EK_OBS: { /* sft_types.h:137-206 */
      double c[3][3], xyz[3], R[9];
      for (int s = 0; s < 3; s++) se3_map(&g->cam, &g->xyz[3 * (v[1 + s] - 1)], c[s]);
      for (int k = 0; k < 3; k++) xyz[k] = (c[0][k] * e->bary[0] + c[1][k] * e->bary[1]) + c[2][k] * e->bary[2];
      quat_to_R(&g->cam.r, R);
      double x = xyz[0], y = xyz[1], z = xyz[2], z2 = z * z;
      double fx = g->fx, fy = g->fy;
      double* Jc = J[0]; /* 2x6 row-major */
      Jc[0] = x * y / z2 * fx;
      Jc[1] = -(1 + (x * x / z2)) * fx;
      Jc[2] = y / z * fx;
      Jc[3] = -1. / z * fx;
      Jc[4] = 0;
      Jc[5] = x / z2 * fx;
      Jc[6] = (1 + y * y / z2) * fy;
      Jc[7] = -x * y / z2 * fy;
      Jc[8] = -x / z * fy;
      Jc[9] = 0;
      Jc[10] = -1. / z * fy;
      Jc[11] = y / z2 * fy;
      for (int s = 0; s < 3; s++) {
        x = c[s][0]; y = c[s][1]; z = c[s][2];
        double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
        double sc = -1. / z;
        /* ((-1/z * tmp) * Trot) * bary  -- Eigen evaluates left to right */
        double st[6];
        for (int i = 0; i < 6; i++) st[i] = sc * tmp[i];
        for (int r = 0; r < 2; r++)
          for (int cc = 0; cc < 3; cc++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += st[r * 3 + k] * R[k * 3 + cc];
            J[1 + s][r * 3 + cc] = acc * e->bary[s];
          }
      }
    } break;
    case EK_REF: { /* sft_types.h:408 */
      static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      memcpy(J[0], I3, sizeof(I3));
    } break;
    case EK_CURV: { /* sft_types.h:293-311 */
      for (int i = 0; i < e->nv; i++) {
        if (e->mcn < 1E-15) { J[i][0] = J[i][1] = J[i][2] = 0.0; continue; }
        double base[3] = {e->mc[0], e->mc[1], e->mc[2]};
        if (i > 0) {
          double wa = -(e->w[i - 1] / e->sumw);
          for (int k = 0; k < 3; k++) base[k] = wa * e->mc[k];
        }
        double den = e->mcn * e->L;
        for (int k = 0; k < 3; k++) J[i][k] = base[k] / den;
      }
    } break;
    case EK_STRETCH: { /* sft_types.h:362-377 */
      const double* a = &g->xyz[3 * (v[0] - 1)];
      const double* b = &g->xyz[3 * (v[1] - 1)];
      double d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
      double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      double ddo = 1.0 / (nrm * e->meas[0]);
      for (int k = 0; k < 3; k++) { J[0][k] = d[k] * ddo; J[1][k] = -(d[k] * ddo); }
    } break;
  }
}

/* buildSystem: block_solver.hpp:502-560 + constructQuadraticForm of each edge type.
 * H dense column-major D x D (symmetric, both triangles filled), b length D. */
static void build_system(graph_t* g, double* H, double* b) {
  const int D = g->D;
  memset(H, 0, sizeof(double) * (size_t)D * D);
  memset(b, 0, sizeof(double) * D);
  double (*J)[12] = (double (*)[12])malloc(sizeof(double) * 12 * 64);
  for (int ei = 0; ei < g->ne; ei++) {
    edge_t* e = &g->e[ei];
    const int* v = edge_verts(e);
    int dim = edge_dim(e);
    edge_linearize(g, e, J);
    double wr = 1.0; /* rho[1] */
    if (e->kind == EK_OBS) { double rho[3]; huber(g, edge_chi2(e), rho); wr = rho[1]; }
    double om = wr * e->info; /* robustInformation: rho[1]*Omega (base_edge.h:96-102) */
    double omr[3];
    for (int k = 0; k < dim; k++) { omr[k] = -(e->info * e->err[k]); omr[k] *= wr; }
    for (int i = 0; i < e->nv; i++) {
      int hi = g->hidx[v[i]];
      if (hi < 0) continue;
      int di = (v[i] == 0) ? 6 : 3;
      const double* A = J[i]; /* dim x di */
      /* ii block and b */
      for (int r = 0; r < di; r++) {
        for (int c = 0; c < di; c++) {
          double s = 0;
          for (int k = 0; k < dim; k++) s += (A[k * di + r] * om) * A[k * di + c];
          H[(hi + r) + (size_t)(hi + c) * D] += s;
        }
        double sb = 0;
        for (int k = 0; k < dim; k++) sb += A[k * di + r] * omr[k];
        b[hi + r] += sb;
      }
      for (int j = i + 1; j < e->nv; j++) {
        int hj = g->hidx[v[j]];
        if (hj < 0) continue;
        int dj = (v[j] == 0) ? 6 : 3;
        const double* B = J[j];
        for (int r = 0; r < di; r++)
          for (int c = 0; c < dj; c++) {
            double s = 0;
            for (int k = 0; k < dim; k++) s += (A[k * di + r] * om) * B[k * dj + c];
            H[(hi + r) + (size_t)(hj + c) * D] += s;
            H[(hj + c) + (size_t)(hi + r) * D] += s;
          }
      }
    }
  }
  free(J);
}

/* ------------------------------------------------------------------ */
/* Public entry: one DefPoseOptimization-equivalent                    */
/* ------------------------------------------------------------------ */
/*
 * Template inputs (see oracle/template_oracle.c for how the reference derives them):
 *   n, xyz0[n*3] rest ("initial") positions, boundary[n], nbr_ptr[n+1]/nbr_idx/nbr_w
 *   (1-ring, ascending node index, weights w_ij), k0[n], E, edge_nodes[E*2] (lo,hi),
 *   edge_L0[E], inc_ptr[n+1]/inc_edge (edges incident to a node, creation order), median_L.
 * Frame inputs: Tcw_in 4x4 float32 row-major, K = fx,fy,cx,cy, N_frame, M observations
 *   (node triplet ascending, barycentrics, undistorted keypoint uv, invSigma2), xyz_in[n*3].
 * Outputs: any pointer may be NULL.
 * ldlt_mode: 0 = Eigen-style pivoted LDLT, 1 = blocked unpivoted.
 * Returns nInitialCorrespondences - nBad (DefOptimizer.cc:577), or -1 on bad input.
 */
/* Graph construction shared by the solve and the "one assembly" test hook. */
typedef struct { uint8_t *viewed, *optlap, *eact; int nOptLap, nViewed, nCurv, nStretch; } graph_aux_t;

static int graph_build(graph_t* gp, graph_aux_t* aux,
    int n, const double* xyz0, const uint8_t* boundary,
    const int32_t* nbr_ptr, const int32_t* nbr_idx, const double* nbr_w, const double* k0,
    int E, const int32_t* edge_nodes, const double* edge_L0,
    const int32_t* inc_ptr, const int32_t* inc_edge, double median_L,
    const float* Tcw_in, const double* K, int N_frame, int M,
    const int32_t* obs_nodes, const double* obs_bary, const double* obs_uv, const double* obs_invsig2,
    const double* xyz_in, double regLap, double regInex, double regTemp, int neighbours_layers, int32_t* dims_out) {
#define g (*gp)
  memset(&g, 0, sizeof(g));
  g.n = n; g.xyz0 = xyz0;
  g.cam = se3_from_f32(Tcw_in);
  g.xyz = (double*)malloc(sizeof(double) * 3 * n);
  memcpy(g.xyz, xyz_in, sizeof(double) * 3 * n);
  g.fx = K[0]; g.fy = K[1]; g.cx = K[2]; g.cy = K[3];
  { /* DefOptimizer.cc:286,342-344: const float deltaMono = sqrt(5.991); rk->setDelta(deltaMono) */
    const float deltaMono = (float)sqrt(5.991);
    g.huber_delta = (double)deltaMono;
    g.huber_dsqr = g.huber_delta * g.huber_delta;
  }

  uint8_t* viewed = (uint8_t*)calloc(n, 1);
  uint8_t* optlap = (uint8_t*)calloc(n, 1);
  for (int i = 0; i < M; i++)
    for (int s = 0; s < 3; s++) viewed[obs_nodes[3 * i + s]] = 1;
  /* DefOptimizer.cc:384-406 -- the loop always expands from ViewedNodes: 1-ring for any layers>=1 */
  for (int i = 0; i < n; i++) optlap[i] = viewed[i];
  if (neighbours_layers >= 1)
    for (int i = 0; i < n; i++)
      if (viewed[i])
        for (int p = nbr_ptr[i]; p < nbr_ptr[i + 1]; p++) optlap[nbr_idx[p]] = 1;
  int nOptLap = 0, nViewed = 0;
  for (int i = 0; i < n; i++) { nOptLap += optlap[i]; nViewed += viewed[i]; }

  /* stretch edge set: edges incident to OptLap nodes, creation order (DefOptimizer.cc:468-478) */
  uint8_t* eact = (uint8_t*)calloc(E > 0 ? E : 1, 1);
  int nStretch = 0;
  for (int i = 0; i < n; i++)
    if (optlap[i])
      for (int p = inc_ptr[i]; p < inc_ptr[i + 1]; p++)
        if (!eact[inc_edge[p]]) { eact[inc_edge[p]] = 1; nStretch++; }

  int nCurv = 0;
  for (int i = 0; i < n; i++)
    if (optlap[i] && !boundary[i]) nCurv += inc_ptr[i + 1] - inc_ptr[i];

  g.ne = M + nViewed + nCurv + nStretch;
  g.e = (edge_t*)calloc(g.ne > 0 ? g.ne : 1, sizeof(edge_t));
  int ne = 0;
  /* observation edges, DefOptimizer.cc:293-361 */
  for (int i = 0; i < M; i++) {
    edge_t* e = &g.e[ne++];
    e->kind = EK_OBS; e->nv = 4;
    e->v[0] = 0;
    for (int s = 0; s < 3; s++) { e->v[1 + s] = 1 + obs_nodes[3 * i + s]; e->bary[s] = obs_bary[3 * i + s]; }
    e->meas[0] = obs_uv[2 * i]; e->meas[1] = obs_uv[2 * i + 1];
    e->info = obs_invsig2[i] / (double)N_frame;
  }
  /* temporal edges, DefOptimizer.cc:363-382 */
  for (int i = 0; i < n; i++)
    if (viewed[i]) {
      edge_t* e = &g.e[ne++];
      e->kind = EK_REF; e->nv = 1; e->v[0] = 1 + i;
      for (int k = 0; k < 3; k++) e->meas[k] = xyz0[3 * i + k];
      e->info = regTemp / pow(median_L, 2);
    }
  /* curvature edges, DefOptimizer.cc:408-463: one per incident mesh edge of every interior OptLap node */
  for (int i = 0; i < n; i++)
    if (optlap[i] && !boundary[i]) {
      int deg = nbr_ptr[i + 1] - nbr_ptr[i];
      for (int p = inc_ptr[i]; p < inc_ptr[i + 1]; p++) {
        edge_t* e = &g.e[ne++];
        e->kind = EK_CURV; e->nv = 1 + deg;
        int* vv = e->v;
        if (e->nv > 8) { e->vext = (int*)malloc(sizeof(int) * e->nv); vv = e->vext; }
        vv[0] = 1 + i;
        for (int j = 0; j < deg; j++) vv[1 + j] = 1 + nbr_idx[nbr_ptr[i] + j];
        e->w = &nbr_w[nbr_ptr[i]];
        e->L = edge_L0[inc_edge[p]];
        e->meas[0] = k0[i];
        e->info = regLap / (double)nOptLap;
      }
    }
  /* stretching edges, DefOptimizer.cc:480-507 */
  for (int k = 0; k < E; k++)
    if (eact[k]) {
      edge_t* e = &g.e[ne++];
      e->kind = EK_STRETCH; e->nv = 2;
      e->v[0] = 1 + edge_nodes[2 * k]; e->v[1] = 1 + edge_nodes[2 * k + 1];
      e->meas[0] = edge_L0[k];
      e->info = regInex / (double)nStretch;
    }

  /* index mapping: camera first, then non-fixed nodes ascending (sparse_optimizer.cpp:181-211) */
  g.hidx = (int*)malloc(sizeof(int) * (n + 1));
  g.hidx[0] = 0;
  int D = 6;
  for (int i = 0; i < n; i++) {
    if (optlap[i]) { g.hidx[1 + i] = D; D += 3; } else g.hidx[1 + i] = -1;
  }
  g.D = D;
  if (dims_out) { dims_out[0] = D; dims_out[1] = nOptLap; dims_out[2] = nViewed; dims_out[3] = nCurv; dims_out[4] = nStretch; dims_out[5] = g.ne; }

  aux->viewed = viewed; aux->optlap = optlap; aux->eact = eact;
  aux->nOptLap = nOptLap; aux->nViewed = nViewed; aux->nCurv = nCurv; aux->nStretch = nStretch;
  return D;
#undef g
}

static void graph_free(graph_t* gp, graph_aux_t* aux) {
  for (int i = 0; i < gp->ne; i++) if (gp->e[i].vext) free(gp->e[i].vext);
  free(gp->e); free(gp->hidx); free(gp->xyz); free(aux->viewed); free(aux->optlap); free(aux->eact);
}

/* Test hook: residuals + Jacobians + normal equations once, at the given state.
 * H is column-major D x D in the reference's index order (camera, then active nodes ascending). */
int sft_oracle_system(
    int n, const double* xyz0, const uint8_t* boundary,
    const int32_t* nbr_ptr, const int32_t* nbr_idx, const double* nbr_w, const double* k0,
    int E, const int32_t* edge_nodes, const double* edge_L0,
    const int32_t* inc_ptr, const int32_t* inc_edge, double median_L,
    const float* Tcw_in, const double* K, int N_frame, int M,
    const int32_t* obs_nodes, const double* obs_bary, const double* obs_uv, const double* obs_invsig2,
    const double* xyz_in, double regLap, double regInex, double regTemp, int neighbours_layers,
    int32_t D_expected, double* H, double* b, double* chi2) {
  graph_t g; graph_aux_t aux;
  int D = graph_build(&g, &aux, n, xyz0, boundary, nbr_ptr, nbr_idx, nbr_w, k0, E, edge_nodes, edge_L0, inc_ptr, inc_edge, median_L,
                      Tcw_in, K, N_frame, M, obs_nodes, obs_bary, obs_uv, obs_invsig2, xyz_in, regLap, regInex, regTemp, neighbours_layers, NULL);
  if (H == NULL || D != D_expected) { graph_free(&g, &aux); return D; }
  compute_active_errors(&g);
  if (chi2) *chi2 = active_robust_chi2(&g);
  build_system(&g, H, b);
  graph_free(&g, &aux);
  return D;
}

int sft_oracle_solve(
    int n, const double* xyz0, const uint8_t* boundary,
    const int32_t* nbr_ptr, const int32_t* nbr_idx, const double* nbr_w, const double* k0,
    int E, const int32_t* edge_nodes, const double* edge_L0,
    const int32_t* inc_ptr, const int32_t* inc_edge, double median_L,
    const float* Tcw_in, const double* K, int N_frame, int M,
    const int32_t* obs_nodes, const double* obs_bary, const double* obs_uv, const double* obs_invsig2,
    const double* xyz_in,
    double regLap, double regInex, double regTemp, int neighbours_layers, int max_iters, int ldlt_mode,
    float* Tcw_out, double* pose7_out, double* xyz_out, double* chi2_obs, uint8_t* outlier,
    double* rep_error, int32_t* iters_done, int32_t* trials_done, double* trace, int32_t* dims_out) {
  graph_t g; graph_aux_t aux;
  int D = graph_build(&g, &aux, n, xyz0, boundary, nbr_ptr, nbr_idx, nbr_w, k0, E, edge_nodes, edge_L0, inc_ptr, inc_edge, median_L,
                      Tcw_in, K, N_frame, M, obs_nodes, obs_bary, obs_uv, obs_invsig2, xyz_in, regLap, regInex, regTemp, neighbours_layers, dims_out);
  double* H = (double*)malloc(sizeof(double) * (size_t)D * D);
  double* Hs = (double*)malloc(sizeof(double) * (size_t)D * D);
  double* b = (double*)malloc(sizeof(double) * D);
  double* x = (double*)calloc(D, sizeof(double));
  int* perm = (int*)malloc(sizeof(int) * D);
  double* tmp = (double*)malloc(sizeof(double) * (size_t)D * 48);
  double* xyz_bak = (double*)malloc(sizeof(double) * 3 * n);

  /* LM state, optimization_algorithm_levenberg.cpp:42-55 */
  double lambda = -1., ni = 2.;
  int nBad = 0;
  const double tau = 1e-5, goodUp = 2. / 3., goodLo = 1. / 3.;
  const int maxTrials = 10;
  int it_count = 0, total_trials = 0;

  if (g.ne > 0) {
    for (int it = 0; it < max_iters; it++) {
      /* solve(), optimization_algorithm_levenberg.cpp:61-164 */
      compute_active_errors(&g);
      double currentChi = active_robust_chi2(&g);
      double tempChi = currentChi;
      double iniChi = currentChi;
      build_system(&g, H, b);
      if (it == 0) {
        double maxDiag = 0.;
        for (int j = 0; j < D; j++) { double v = fabs(H[j + (size_t)j * D]); if (v > maxDiag) maxDiag = v; }
        lambda = tau * maxDiag; ni = 2; nBad = 0;
      }
      double lambda_start = lambda;
      double rho = 0;
      int qmax = 0, accepted = 0;
      do {
        /* push */
        se3_t cam_bak = g.cam;
        memcpy(xyz_bak, g.xyz, sizeof(double) * 3 * n);
        /* setLambda + dense copy + LDLT */
        memcpy(Hs, H, sizeof(double) * (size_t)D * D);
        for (int j = 0; j < D; j++) Hs[j + (size_t)j * D] += lambda;
        int ok2;
        if (ldlt_mode == 0) { ok2 = ldlt_pivoted(Hs, D, perm, tmp); if (ok2) ldlt_pivoted_solve(Hs, D, perm, b, x); }
        else { ok2 = ldlt_blocked(Hs, D, tmp); if (ok2) ldlt_blocked_solve(Hs, D, b, x); }
        /* update (x keeps its previous content if the factorisation failed, as in g2o) */
        {
          se3_t dT = se3_exp(&x[0]);
          g.cam = se3_mul(&dT, &g.cam);
          for (int i = 0; i < n; i++) {
            int h = g.hidx[1 + i];
            if (h >= 0) for (int k = 0; k < 3; k++) g.xyz[3 * i + k] += x[h + k];
          }
        }
        compute_active_errors(&g);
        tempChi = active_robust_chi2(&g);
        if (!ok2) tempChi = DBL_MAX;
        rho = (currentChi - tempChi);
        double scale = 0.;
        for (int j = 0; j < D; j++) scale += x[j] * (lambda * x[j] + b[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - pow((2 * rho - 1), 3);
          alpha = alpha < goodUp ? alpha : goodUp;
          double sf = goodLo > alpha ? goodLo : alpha;
          lambda *= sf; ni = 2; currentChi = tempChi; accepted = 1;
        } else {
          lambda *= ni; ni *= 2;
          g.cam = cam_bak; memcpy(g.xyz, xyz_bak, sizeof(double) * 3 * n);
        }
        qmax++;
      } while (rho < 0 && qmax < maxTrials);
      total_trials += qmax;
      if (trace) {
        double* t = &trace[it * SFT_TRACE_STRIDE];
        t[0] = iniChi; t[1] = lambda_start; t[2] = qmax; t[3] = currentChi; t[4] = lambda; t[5] = rho; t[6] = accepted; t[7] = 0;
      }
      it_count++;
      if (qmax == maxTrials || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) break;
    }
  }
  if (iters_done) *iters_done = it_count;
  if (trials_done) *trials_done = total_trials;

  /* classification, DefOptimizer.cc:515-537 (uses the edges' last computed error; float chi2) */
  int nBadObs = 0;
  for (int i = 0; i < M; i++) {
    const float chi2 = (float)edge_chi2(&g.e[i]);
    int bad = chi2 > 5.991;
    if (chi2_obs) chi2_obs[i] = edge_chi2(&g.e[i]);
    if (outlier) outlier[i] = (uint8_t)bad;
    nBadObs += bad;
    g.e[i].meas[2] = bad; /* scratch */
  }
  /* mean reprojection error over inliers at the final estimate, :538-559 */
  double sumError = 0.0; unsigned cnt = 0;
  for (int i = 0; i < M; i++)
    if (!(int)g.e[i].meas[2]) {
      edge_error(&g, &g.e[i]);
      double er = sqrt(pow(g.e[i].err[0], 2) + pow(g.e[i].err[1], 2));
      sumError += er; cnt++;
    }
  if (rep_error) *rep_error = sumError / cnt;
  if (Tcw_out) se3_to_f32(&g.cam, Tcw_out);
  if (pose7_out) {
    pose7_out[0] = g.cam.t[0]; pose7_out[1] = g.cam.t[1]; pose7_out[2] = g.cam.t[2];
    pose7_out[3] = g.cam.r.x; pose7_out[4] = g.cam.r.y; pose7_out[5] = g.cam.r.z; pose7_out[6] = g.cam.r.w;
  }
  if (xyz_out) memcpy(xyz_out, g.xyz, sizeof(double) * 3 * n);

  graph_free(&g, &aux);
  free(H); free(Hs); free(b); free(x); free(perm); free(tmp); free(xyz_bak);
  return M - nBadObs;
}

/* ------------------------------------------------------------------ */
/* Unit-test hooks                                                     */
/* ------------------------------------------------------------------ */
void sft_oracle_se3_exp(const double u[6], double pose7[7]) {
  se3_t T = se3_exp(u);
  pose7[0] = T.t[0]; pose7[1] = T.t[1]; pose7[2] = T.t[2];
  pose7[3] = T.r.x; pose7[4] = T.r.y; pose7[5] = T.r.z; pose7[6] = T.r.w;
}

void sft_oracle_pose_from_f32(const float* Tcw, double pose7[7]) {
  se3_t T = se3_from_f32(Tcw);
  pose7[0] = T.t[0]; pose7[1] = T.t[1]; pose7[2] = T.t[2];
  pose7[3] = T.r.x; pose7[4] = T.r.y; pose7[5] = T.r.z; pose7[6] = T.r.w;
}

void sft_oracle_huber(double delta, double e2, double rho[3]) {
  graph_t g; g.huber_delta = delta; g.huber_dsqr = delta * delta;
  huber(&g, e2, rho);
}

/* Map-point write back, DefMapPoint.cc:129-147 (double products, float32 store). */
void sft_oracle_recalc_points(int M, const int32_t* obs_nodes, const double* obs_bary, const double* xyz, float* out) {
  for (int i = 0; i < M; i++)
    for (int k = 0; k < 3; k++)
      out[3 * i + k] = (float)(obs_bary[3 * i] * xyz[3 * obs_nodes[3 * i] + k] + obs_bary[3 * i + 1] * xyz[3 * obs_nodes[3 * i + 1] + k] +
                               obs_bary[3 * i + 2] * xyz[3 * obs_nodes[3 * i + 2] + k]);
}
