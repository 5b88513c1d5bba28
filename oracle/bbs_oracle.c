/*
 * bbs_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never on the product path).
 *
 * Restatement of the uniform bicubic B-spline primitives the mapping side uses
 * (SURVEY.md section 8a row B1d):
 *   interval / normalised coordinate .... Thirdparty/BBS/bbs.cc:70-92
 *   cubic basis B, B', B'' ............... Thirdparty/BBS/bbs.cc:95-121
 *   derivative scale ..................... Thirdparty/BBS/bbs.cc:140-145
 *   tensor-product evaluation ............ Thirdparty/BBS/bbs.cc:155-195
 *   colocation weights (16 taps / site) .. Thirdparty/BBS/bbs.cc:214-355
 *
 * PARITY PINNED: tests/test_oracle_nrsfm.py (test_bbs_oracle_*) checks this file bit-for-bit against the
 * reference's own bbs.cc compiled into oracle/_ref/libbbs_ref.so (built by oracle/Makefile
 * from /root/reference, never copied) and against the golden vectors generated from it
 * (tests/golden/bbs_*.npz, tests/golden/make_golden_bbs.py).
 */
#include <math.h>
#include <stdint.h>

typedef struct { double umin, umax; int nptsu; double vmin, vmax; int nptsv; int valdim; } bbs_par;

static void norm_inter(double xmin, double xmax, int npts, double x, double* nx, int* inter) {
  int ninter = npts - 3;
  double width = (xmax - xmin) / ninter;
  if (x == xmax) { *nx = 1.0; *inter = ninter - 1; }
  else if (x < xmin) { *nx = (x - xmin) / width; *inter = -1; }
  else if (x > xmax) { *nx = (x - xmin) / width - ninter; *inter = ninter; }
  else { double s = (x - xmin) / width; *inter = (int)floor(s); *nx = s - *inter; }
}

static void basis(int order, double t, double* b) {
  double t2 = t * t, t3 = t2 * t;
  switch (order) {
    case 0:
      b[0] = (-t3 + 3.0 * t2 - 3.0 * t + 1.0) / 6.0;
      b[1] = (3.0 * t3 - 6.0 * t2 + 4.0) / 6.0;
      b[2] = (-3.0 * t3 + 3.0 * t2 + 3.0 * t + 1.0) / 6.0;
      b[3] = t3 / 6.0;
      break;
    case 1:
      b[0] = (-t2 + 2 * t - 1) / 2.0;
      b[1] = (3.0 * t2 - 4.0 * t) / 2.0;
      b[2] = (-3 * t2 + 2 * t + 1) / 2.0;
      b[3] = t2 / 2.0;
      break;
    default:
      b[0] = -t + 1.0;
      b[1] = 3.0 * t - 2.0;
      b[2] = -3.0 * t + 1.0;
      b[3] = t;
      break;
  }
}

static double deriv_fact(const bbs_par* p, int du, int dv) {
  double su = (p->umax - p->umin) / (p->nptsu - 3);
  double sv = (p->vmax - p->vmin) / (p->nptsv - 3);
  return 1.0 / (pow(su, du) * pow(sv, dv));
}

void bbs_oracle_basis(int order, double t, double* b4) { basis(order, t, b4); }

/* bbs.cc:155-195.  status[k] = 1 when the site is outside the definition domain (the reference then
 * indexes out of bounds; here the value is left at 0). */
void bbs_oracle_eval(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, int valdim,
                     const double* ctrl, const double* u, const double* v, int n, int du, int dv, double* val, uint8_t* status) {
  bbs_par p = {umin, umax, nptsu, vmin, vmax, nptsv, valdim};
  double fact = deriv_fact(&p, du, dv);
  for (int k = 0; k < n; k++) {
    double nu, nv, bu[4], bv[4];
    int Iu, Iv;
    norm_inter(umin, umax, nptsu, u[k], &nu, &Iu);
    norm_inter(vmin, vmax, nptsv, v[k], &nv, &Iv);
    basis(du, nu, bu);
    basis(dv, nv, bv);
    for (int d = 0; d < valdim; d++) val[valdim * k + d] = 0.0;
    int bad = (Iu < 0 || Iu > nptsu - 4 || Iv < 0 || Iv > nptsv - 4);
    if (status) status[k] = (uint8_t)bad;
    if (bad) continue;
    for (int iu = 0; iu < 4; iu++)
      for (int iv = 0; iv < 4; iv++) {
        double bas = bu[iu] * bv[iv];
        int ind = valdim * ((iu + Iu) * nptsv + iv + Iv);
        for (int d = 0; d < valdim; d++) val[valdim * k + d] += ctrl[ind++] * bas;
      }
    for (int d = 0; d < valdim; d++) val[valdim * k + d] *= fact;
  }
}

/* Row view of the colocation matrix (bbs.cc:214-355 builds the same numbers in CSC form): for site k the 16
 * (column, weight) pairs in (iu, iv) order; weight = fact * Bu[iu] * Bv[iv] (fact == 1 for du = dv = 0, where the
 * reference's coloc() does not multiply at all).  Returns 1 if a site is outside the domain (reference error code). */
int bbs_oracle_coloc(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv,
                     const double* u, const double* v, int n, int du, int dv, int32_t* cols, double* w) {
  bbs_par p = {umin, umax, nptsu, vmin, vmax, nptsv, 1};
  double fact = deriv_fact(&p, du, dv);
  int ret = 0;
  for (int k = 0; k < n; k++) {
    double nu, nv, bu[4], bv[4];
    int Iu, Iv;
    norm_inter(umin, umax, nptsu, u[k], &nu, &Iu);
    norm_inter(vmin, vmax, nptsv, v[k], &nv, &Iv);
    if (Iu < 0 || Iu > nptsu - 4 || Iv < 0 || Iv > nptsv - 4) {
      ret = 1;
      for (int t = 0; t < 16; t++) { cols[16 * k + t] = -1; w[16 * k + t] = 0.0; }
      continue;
    }
    basis(du, nu, bu);
    basis(dv, nv, bv);
    for (int iu = 0; iu < 4; iu++)
      for (int iv = 0; iv < 4; iv++) {
        cols[16 * k + 4 * iu + iv] = (iu + Iu) * nptsv + iv + Iv;
        w[16 * k + 4 * iu + iv] = (du == 0 && dv == 0) ? bu[iu] * bv[iv] : fact * bu[iu] * bv[iv];
      }
  }
  return ret;
}
