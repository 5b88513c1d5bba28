// C ABI of the Schwarzian warp fit (include/defslam_hip.h: dsh_schwarp_fit, dsh_schwarp_eval).
// The trust-region loop (3 iterations in the reference, SchwarpDatabase.cc:211-222) is sequenced on the host;
// residuals, Jacobian, normal equations, the 2N x 2N Cholesky solve and the DiffProp extraction run on the GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/defslam_hip.h"
#include "dsh_ctx.h"
#include "dsh_diffdb.h"

extern "C" hipError_t ddb_append(int, const uint8_t*, const float*, const int32_t*, const int32_t*, const int32_t*, int32_t*, int32_t*, void*, size_t, long long, long long,
                                 float*, int32_t*, int32_t*, int32_t*, hipStream_t);
extern "C" size_t ddb_scan_tmp_bytes(int);

extern "C" hipError_t nrsfm_swp_eval(double, double, int, double, double, int, int, double, double, double, const float*, const float*, const float*,
                                     const double*, double*, double*, int, hipStream_t);
extern "C" hipError_t nrsfm_swp_loss(int, int, const double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_swp_normal(int, int, int, double*, double*, const double*, const double*, double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_swp_colscale(int, const double*, double*, hipStream_t);
extern "C" hipError_t nrsfm_swp_solve(int, const double*, const double*, double, double*, double*, double*, double*, int, int, hipStream_t);
extern "C" int nrsfm_swp_solve_np(int);
extern "C" hipError_t nrsfm_swp_step(int, const double*, const double*, const double*, const double*, double*, double*, hipStream_t);
extern "C" size_t nrsfm_swp_fit_bytes();
extern "C" void nrsfm_swp_fit_fill(void*, double, double, int, double, double, int, int, double, double, double, float, float, int, const float*, const float*, const float*,
                                   double*, double*, double*, double*, double*, double*, double*, double*, double*, double*, double*, float*, uint8_t*, int32_t*, double*,
                                   const double*, void*);
extern "C" size_t nrsfm_swp_compact_bytes(int, int, int);
extern "C" hipError_t nrsfm_swp_fit_batch(void*, int, int, int, int, int, hipStream_t);
namespace dsh { void bbs_bending_dense(const dsh_bbs* b, double lambda, double* Bm); }
extern "C" hipError_t nrsfm_swp_diffprop(double, double, int, double, double, int, int, const float*, const float*, const double*, float, float, float*,
                                         uint8_t*, hipStream_t);

namespace {
#define HIPCHK(c, call)                                                                                        \
  do {                                                                                                         \
    hipError_t e__ = (call);                                                                                   \
    if (e__ != hipSuccess) {   /* copies from local host buffers may be in flight: drain the stream before they go away */    \
      (void)hipStreamSynchronize((c)->stream);                                                                  \
      return dsh_fail(c, DSH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));                      \
    }                                                                                                           \
  } while (0)

struct DevBuf {   // a slice of the context's scratch (dsh_ctx.h); nothing to free
  void* p = nullptr;
  hipError_t alloc(dsh_ctx_base* c, size_t bytes) { return c->scratch.take(bytes, &p); }
  template <class T> T* as() { return static_cast<T*>(p); }
};

struct Fit {
  dsh_ctx_base* c;
  const dsh_bbs* b;
  int P, N, n2, m;
  double fxs, fys, lambda;
  DevBuf kp1, kp2, isg, x, xn, cs, g, dx, r, J, A, M, W, scal;
  hipStream_t st;

  int eval(const double* xdev, bool with_j) {
    HIPCHK(c, nrsfm_swp_eval(b->umin, b->umax, b->nptsu, b->vmin, b->vmax, b->nptsv, P, fxs, fys, lambda, kp1.as<float>(), kp2.as<float>(), isg.as<float>(),
                             xdev, r.as<double>(), J.as<double>(), with_j ? 1 : 0, st));
    HIPCHK(c, nrsfm_swp_loss(2 * P, m, r.as<double>(), scal.as<double>(), st));
    if (with_j) HIPCHK(c, nrsfm_swp_normal(2 * P, m, n2, J.as<double>(), r.as<double>(), cs.as<double>(), scal.as<double>(), A.as<double>(), g.as<double>(), st));
    return DSH_OK;
  }
  int scalars(double* out8) {
    HIPCHK(c, hipMemcpyAsync(out8, scal.p, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return DSH_OK;
  }
};

int setup(Fit& f, dsh_ctx_base* c, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, const float* invsig, double fxs, double fys, double lambda,
          const double* x) {
  f.c = c; f.b = bbs; f.P = P; f.N = bbs->nptsu * bbs->nptsv; f.n2 = 2 * f.N; f.m = 2 * P + 4 * f.N;
  f.fxs = fxs; f.fys = fys; f.lambda = lambda; f.st = c->stream;
  c->scratch.reset();
  HIPCHK(c, f.kp1.alloc(c, 8 * (size_t)P)); HIPCHK(c, f.kp2.alloc(c, 8 * (size_t)P)); HIPCHK(c, f.isg.alloc(c, 4 * (size_t)P));
  HIPCHK(c, f.x.alloc(c, 8 * (size_t)f.n2)); HIPCHK(c, f.xn.alloc(c, 8 * (size_t)f.n2)); HIPCHK(c, f.cs.alloc(c, 8 * (size_t)f.n2)); HIPCHK(c, f.g.alloc(c, 8 * (size_t)f.n2));
  HIPCHK(c, f.dx.alloc(c, 8 * (size_t)f.n2)); HIPCHK(c, f.r.alloc(c, 8 * (size_t)f.m)); HIPCHK(c, f.J.alloc(c, 8 * (size_t)f.m * f.n2));
  HIPCHK(c, f.A.alloc(c, 8 * (size_t)f.n2 * f.n2)); {
    const size_t np = (size_t)nrsfm_swp_solve_np(f.n2);
    HIPCHK(c, f.M.alloc(c, 8 * np * np)); HIPCHK(c, f.W.alloc(c, 8 * np * 16));
  }
  HIPCHK(c, f.scal.alloc(c, 64));
  HIPCHK(c, hipMemcpyAsync(f.kp1.p, kp1, 8 * (size_t)P, hipMemcpyHostToDevice, f.st));
  HIPCHK(c, hipMemcpyAsync(f.kp2.p, kp2, 8 * (size_t)P, hipMemcpyHostToDevice, f.st));
  HIPCHK(c, hipMemcpyAsync(f.isg.p, invsig, 4 * (size_t)P, hipMemcpyHostToDevice, f.st));
  HIPCHK(c, hipMemcpyAsync(f.x.p, x, 8 * (size_t)f.n2, hipMemcpyHostToDevice, f.st));
  std::vector<double> ones(f.n2, 1.0);
  HIPCHK(c, hipMemcpyAsync(f.cs.p, ones.data(), 8 * (size_t)f.n2, hipMemcpyHostToDevice, f.st));
  HIPCHK(c, hipMemsetAsync(f.scal.p, 0, 64, f.st));
  HIPCHK(c, hipMemsetAsync(f.dx.p, 0, 8 * (size_t)f.n2, f.st));
  HIPCHK(c, hipStreamSynchronize(f.st));
  return DSH_OK;
}

bool args_ok(const dsh_bbs* b, int P, const float* kp1, const float* kp2, const float* invsig, const double* x) {
  return b && b->nptsu >= 4 && b->nptsv >= 4 && b->umax > b->umin && b->vmax > b->vmin && P > 0 && kp1 && kp2 && invsig && x && b->nptsu * b->nptsv <= 4096;
}
}  // namespace

extern "C" {

int dsh_schwarp_eval(dsh_ctx* ctx, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, const float* invsig, double fx_slot, double fy_slot,
                     double lambda, const double* x, double* residuals, double* jacobian) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_schwarp_eval: host-only context, no GPU (there is no CPU fallback)");
  if (!args_ok(bbs, P, kp1, kp2, invsig, x) || !residuals) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_eval: bad argument");
  (void)hipSetDevice(c->device);
  Fit f;
  int rc = setup(f, c, bbs, P, kp1, kp2, invsig, fx_slot, fy_slot, lambda, x);
  if (rc != DSH_OK) return rc;
  HIPCHK(c, nrsfm_swp_eval(bbs->umin, bbs->umax, bbs->nptsu, bbs->vmin, bbs->vmax, bbs->nptsv, P, fx_slot, fy_slot, lambda, f.kp1.as<float>(), f.kp2.as<float>(),
                           f.isg.as<float>(), f.x.as<double>(), f.r.as<double>(), f.J.as<double>(), jacobian ? 1 : 0, f.st));
  HIPCHK(c, hipMemcpyAsync(residuals, f.r.p, 8 * (size_t)f.m, hipMemcpyDeviceToHost, f.st));
  if (jacobian) HIPCHK(c, hipMemcpyAsync(jacobian, f.J.p, 8 * (size_t)f.m * f.n2, hipMemcpyDeviceToHost, f.st));
  HIPCHK(c, hipStreamSynchronize(f.st));
  return DSH_OK;
}

}  // extern "C"

// The batched fit: every problem's inputs go up in ONE copy, the fits advance together through a fixed sequence of launches
// with the trust-region control on the device (nrsfm_kernels.hip: nrsfm_swp_fit_batch), every result comes back in ONE copy.
// stores / db (both or neither): the DiffProp records of the matches that are kept go into the device-resident database instead of
// (or besides) the host -- dsh_schwarp_fit_batch_store.
static int fit_batch(dsh_ctx* ctx, int B, dsh_schwarp_problem* probs, const dsh_schwarp_store* stores, dsh_diffdb* db) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_schwarp_fit_batch: host-only context, no GPU (there is no CPU fallback)");
  if (B <= 0 || !probs) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_fit_batch: bad argument");
  if (db && (db->ctx != c || !stores)) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_fit_batch_store: the database belongs to another context / no store descriptors");
  if (db) {   // room for every record this call can add, BEFORE anything is launched: a call stores all of its records or fails untouched
    long long worst = 0;
    for (int b = 0; b < B; b++) worst += std::max(probs[b].P, 0);
    if (ddb_reserve(db, db->count + worst) != 0) return dsh_fail(c, DSH_ERR_HIP, "dsh_schwarp_fit_batch_store: out of device memory while growing the database");
  }
  int maxP = 0, maxN = 0, max_it = 0;
  for (int b = 0; b < B; b++) {
    const dsh_schwarp_problem& q = probs[b];
    if (!args_ok(&q.bbs, q.P, q.kp1, q.kp2, q.invsig, q.x) || q.max_iters < 0 || q.max_iters > 1000) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_fit_batch: bad argument in problem " + std::to_string(b));
    if (q.bbs.nptsu * q.bbs.nptsv > 256)
      return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_fit: more than 256 control points (the one-workgroup solve handles 2N <= 512 unknowns; the reference uses 13 x 15 = 195)");
    if (!db && (q.diff == nullptr) != (q.drop == nullptr)) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_fit_batch: diff and drop go together");
    if (db && !stores[b].point_id) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_fit_batch_store: point_id missing in problem " + std::to_string(b));
    maxP = std::max(maxP, q.P); maxN = std::max(maxN, q.bbs.nptsu * q.bbs.nptsv); max_it = std::max(max_it, q.max_iters);
  }
  (void)hipSetDevice(c->device);
  hipStream_t st = c->stream;
  c->scratch.reset();
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  // ---- layout of the input block (host-staged) and of the output block
  const size_t fit_bytes = nrsfm_swp_fit_bytes();
  size_t in_bytes = al(fit_bytes * (size_t)B), out_bytes = 0;
  int with_init = 0;
  struct Off { size_t kp1, kp2, isg, x0, cs, xo, diff, drop, info, costs, bend; };
  std::vector<Off> off(B);
  // x lives at the head of the output block (in/out): only that part is uploaded with the start values
  for (int b = 0; b < B; b++) { off[b].xo = out_bytes; out_bytes += al(8 * 2 * (size_t)probs[b].bbs.nptsu * probs[b].bbs.nptsv); }
  const size_t x_bytes = out_bytes;
  for (int b = 0; b < B; b++) {
    const dsh_schwarp_problem& q = probs[b];
    const size_t n2 = 2 * (size_t)q.bbs.nptsu * q.bbs.nptsv;
    Off& o = off[b];
    o.kp1 = in_bytes; in_bytes += al(8 * (size_t)q.P);
    o.kp2 = in_bytes; in_bytes += al(8 * (size_t)q.P);
    o.isg = in_bytes; in_bytes += al(4 * (size_t)q.P);
    o.cs = in_bytes; in_bytes += al(8 * n2);
    // bending matrix of the Warp::initialize stage: one per run of problems with the same grid and weight
    o.bend = 0;
    if (q.init_lambda > 0.0) {
      const bool same = b > 0 && probs[b - 1].init_lambda == q.init_lambda && std::memcmp(&probs[b - 1].bbs, &q.bbs, sizeof(dsh_bbs)) == 0 && off[b - 1].bend;
      if (same) o.bend = off[b - 1].bend;
      else { o.bend = in_bytes; in_bytes += al(8 * (n2 / 2) * (n2 / 2)); }
      with_init = 1;
    }
    o.diff = out_bytes; out_bytes += al(!db && q.diff ? 72 * (size_t)q.P : 0);     // store mode: records and flags live in one strided block (below)
    o.drop = out_bytes; out_bytes += al(!db && q.drop ? (size_t)q.P : 0);
    o.info = out_bytes; out_bytes += 256;
    o.costs = out_bytes; out_bytes += 256;
  }
  // store mode: DiffProp records / drop flags / point ids / tags / second-keyframe indices of all fits, problem b at stride maxP
  DevBuf sdiff, sdrop, skeep, spos, stmp;
  const size_t nall = (size_t)B * maxP;
  size_t o_pid = 0, o_tag = 0, o_idx2 = 0;
  if (db) {
    o_pid = in_bytes; in_bytes += al(4 * nall);
    o_tag = in_bytes; in_bytes += al(4 * nall);
    o_idx2 = in_bytes; in_bytes += al(4 * nall);
    HIPCHK(c, sdiff.alloc(c, 72 * nall)); HIPCHK(c, sdrop.alloc(c, nall)); HIPCHK(c, skeep.alloc(c, 4 * nall)); HIPCHK(c, spos.alloc(c, 4 * nall));
    HIPCHK(c, stmp.alloc(c, ddb_scan_tmp_bytes((int)nall)));
  }
  DevBuf din, dout;
  HIPCHK(c, din.alloc(c, in_bytes)); HIPCHK(c, dout.alloc(c, out_bytes));
  HIPCHK(c, c->pin_in.ensure(in_bytes + x_bytes)); HIPCHK(c, c->pin_out.ensure(out_bytes));
  char* hin = c->pin_in.p;
  char* hx = c->pin_in.p + in_bytes;   // start values of x, laid out like the head of the output block
  std::memset(hx, 0, x_bytes);
  char* dib = din.as<char>();
  char* dob = dout.as<char>();
  // what has to start at zero (scalars of the controller, the step vector) lies in one block: one memset for the whole batch
  DevBuf dzero;
  size_t zero_bytes = 0;
  for (int b = 0; b < B; b++) zero_bytes += 128 + al(8 * 2 * (size_t)probs[b].bbs.nptsu * probs[b].bbs.nptsv);
  HIPCHK(c, dzero.alloc(c, zero_bytes));
  HIPCHK(c, hipMemsetAsync(dzero.p, 0, zero_bytes, st));
  size_t zoff = 0;
  for (int b = 0; b < B; b++) {
    const dsh_schwarp_problem& q = probs[b];
    const Off& o = off[b];
    const int N = q.bbs.nptsu * q.bbs.nptsv, n2 = 2 * N, m = 2 * q.P + 4 * N;
    std::memcpy(hin + o.kp1, q.kp1, 8 * (size_t)q.P); std::memcpy(hin + o.kp2, q.kp2, 8 * (size_t)q.P); std::memcpy(hin + o.isg, q.invsig, 4 * (size_t)q.P);
    double* cs = reinterpret_cast<double*>(hin + o.cs);
    for (int j = 0; j < n2; j++) cs[j] = 1.0;
    if (q.init_lambda > 0.0) {
      if (b == 0 || off[b - 1].bend != o.bend) dsh::bbs_bending_dense(&q.bbs, q.init_lambda, reinterpret_cast<double*>(hin + o.bend));
    } else {
      std::memcpy(hx + o.xo, q.x, 8 * (size_t)n2);
    }
    DevBuf xn, g, r, J, A, M, W, compact;
    double* scal = reinterpret_cast<double*>(dzero.as<char>() + zoff);
    double* dx = reinterpret_cast<double*>(dzero.as<char>() + zoff + 128);
    zoff += 128 + al(8 * (size_t)n2);
    const size_t np = (size_t)nrsfm_swp_solve_np(n2);
    HIPCHK(c, xn.alloc(c, 8 * (size_t)n2)); HIPCHK(c, g.alloc(c, 8 * (size_t)n2)); HIPCHK(c, r.alloc(c, 8 * (size_t)m));
    // the dense (2P+4N) x 2N buffer only serves the Warp::initialize stage (its colocation matrix); the fit keeps its Jacobian structured
    HIPCHK(c, J.alloc(c, q.init_lambda > 0.0 ? 8 * (size_t)m * n2 : 256)); HIPCHK(c, compact.alloc(c, nrsfm_swp_compact_bytes(q.P, q.bbs.nptsu, q.bbs.nptsv)));
    HIPCHK(c, A.alloc(c, 8 * (size_t)n2 * n2)); HIPCHK(c, M.alloc(c, 8 * np * np)); HIPCHK(c, W.alloc(c, 8 * np * 16));
    nrsfm_swp_fit_fill(hin + fit_bytes * (size_t)b, q.bbs.umin, q.bbs.umax, q.bbs.nptsu, q.bbs.vmin, q.bbs.vmax, q.bbs.nptsv, q.P, q.fx_slot, q.fy_slot, q.lambda, q.fx, q.fy,
                       q.max_iters, reinterpret_cast<const float*>(dib + o.kp1), reinterpret_cast<const float*>(dib + o.kp2), reinterpret_cast<const float*>(dib + o.isg),
                       reinterpret_cast<double*>(dob + o.xo), xn.as<double>(), reinterpret_cast<double*>(dib + o.cs), g.as<double>(), dx, r.as<double>(),
                       J.as<double>(), A.as<double>(), M.as<double>(), W.as<double>(), scal,
                       db ? sdiff.as<float>() + 18 * (size_t)b * maxP : (q.diff ? reinterpret_cast<float*>(dob + o.diff) : nullptr),
                       db ? sdrop.as<uint8_t>() + (size_t)b * maxP : (q.drop ? reinterpret_cast<uint8_t*>(dob + o.drop) : nullptr),
                       reinterpret_cast<int32_t*>(dob + o.info), reinterpret_cast<double*>(dob + o.costs),
                       q.init_lambda > 0.0 ? reinterpret_cast<const double*>(dib + o.bend) : nullptr, compact.p);
  }
  int32_t max_pid = -1;
  if (db) {
    int32_t* hp = reinterpret_cast<int32_t*>(hin + o_pid);
    int32_t* ht = reinterpret_cast<int32_t*>(hin + o_tag);
    int32_t* hi = reinterpret_cast<int32_t*>(hin + o_idx2);
    for (int b = 0; b < B; b++)
      for (int i = 0; i < maxP; i++) {
        const bool in = i < probs[b].P;
        const int32_t id = in ? stores[b].point_id[i] : -1;
        hp[(size_t)b * maxP + i] = id;
        ht[(size_t)b * maxP + i] = stores[b].tag;
        hi[(size_t)b * maxP + i] = (in && stores[b].idx2) ? stores[b].idx2[i] : (in ? i : -1);
        max_pid = std::max(max_pid, id);
      }
  }
  HIPCHK(c, hipMemcpyAsync(dib, hin, in_bytes, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dob, hx, x_bytes, hipMemcpyHostToDevice, st));
  if (out_bytes > x_bytes) HIPCHK(c, hipMemsetAsync(dob + x_bytes, 0, out_bytes - x_bytes, st));
  HIPCHK(c, nrsfm_swp_fit_batch(dib, B, maxP, maxN, max_it, with_init, st));
  HIPCHK(c, hipMemcpyAsync(c->pin_out.p, dob, out_bytes, hipMemcpyDeviceToHost, st));
  std::vector<uint8_t> hdrop;
  std::vector<float> hdiff;
  int32_t added = 0;
  if (db) {   // kept records -> the database, in (fit, match) order; only the drop flags (and, if asked for, the records) travel to the host
    HIPCHK(c, ddb_append((int)nall, sdrop.as<uint8_t>(), sdiff.as<float>(), reinterpret_cast<const int32_t*>(dib + o_pid), reinterpret_cast<const int32_t*>(dib + o_tag),
                         reinterpret_cast<const int32_t*>(dib + o_idx2), skeep.as<int32_t>(), spos.as<int32_t>(), stmp.p, ddb_scan_tmp_bytes((int)nall), db->count,
                         db->cap, db->rec, db->pid, db->tag, db->idx2, st));
    hdrop.resize(nall);
    HIPCHK(c, hipMemcpyAsync(hdrop.data(), sdrop.p, nall, hipMemcpyDeviceToHost, st));
    bool want_diff = false;
    for (int b = 0; b < B; b++) want_diff = want_diff || probs[b].diff != nullptr;
    if (want_diff) { hdiff.resize(18 * nall); HIPCHK(c, hipMemcpyAsync(hdiff.data(), sdiff.p, 72 * nall, hipMemcpyDeviceToHost, st)); }
    int32_t last[2] = {0, 0};   // records added = exclusive scan position + keep flag of the last entry
    HIPCHK(c, hipMemcpyAsync(&last[0], spos.as<int32_t>() + nall - 1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(&last[1], skeep.as<int32_t>() + nall - 1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    added = last[0] + last[1];
  }
  HIPCHK(c, hipStreamSynchronize(st));
  if (db) {
    if (db->count + added > db->cap) return dsh_fail(c, DSH_ERR_STATE, "dsh_schwarp_fit_batch_store: internal error, the reserved capacity was exceeded");
    db->count += added;
    db->max_pid = std::max(db->max_pid, max_pid);
  }
  const char* ho = c->pin_out.p;
  for (int b = 0; b < B; b++) {
    dsh_schwarp_problem& q = probs[b];
    const Off& o = off[b];
    const size_t n2 = 2 * (size_t)q.bbs.nptsu * q.bbs.nptsv;
    std::memcpy(q.x, ho + o.xo, 8 * n2);
    if (db) {
      if (q.drop) std::memcpy(q.drop, hdrop.data() + (size_t)b * maxP, (size_t)q.P);
      if (q.diff) std::memcpy(q.diff, hdiff.data() + 18 * (size_t)b * maxP, 72 * (size_t)q.P);
    } else if (q.diff) { std::memcpy(q.diff, ho + o.diff, 72 * (size_t)q.P); std::memcpy(q.drop, ho + o.drop, (size_t)q.P); }
    std::memcpy(q.info, ho + o.info, sizeof q.info);
    std::memcpy(&q.init_ok, ho + o.info + sizeof q.info, sizeof q.init_ok);
    std::memcpy(q.costs, ho + o.costs, sizeof q.costs);
  }
  return DSH_OK;
}

extern "C" {

int dsh_schwarp_fit_batch(dsh_ctx* ctx, int B, dsh_schwarp_problem* probs) { return fit_batch(ctx, B, probs, nullptr, nullptr); }

int dsh_schwarp_fit_batch_store(dsh_ctx* ctx, int B, dsh_schwarp_problem* probs, const dsh_schwarp_store* stores, dsh_diffdb* db) {
  if (!db || !stores) return dsh_fail(reinterpret_cast<dsh_ctx_base*>(ctx), DSH_ERR_ARG, "dsh_schwarp_fit_batch_store: bad argument");
  return fit_batch(ctx, B, probs, stores, db);
}

int dsh_schwarp_fit(dsh_ctx* ctx, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, const float* invsig, double fx_slot, double fy_slot,
                    double lambda, float fx, float fy, int max_iters, double* x, dsh_diffprop* diff, uint8_t* drop, int32_t* info, double* costs) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_schwarp_fit: host-only context, no GPU (there is no CPU fallback)");
  if (!args_ok(bbs, P, kp1, kp2, invsig, x) || max_iters < 0) return dsh_fail(c, DSH_ERR_ARG, "dsh_schwarp_fit: bad argument");
  dsh_schwarp_problem q{};
  q.bbs = *bbs; q.P = P; q.kp1 = kp1; q.kp2 = kp2; q.invsig = invsig; q.fx_slot = fx_slot; q.fy_slot = fy_slot; q.lambda = lambda; q.fx = fx; q.fy = fy;
  q.max_iters = max_iters; q.x = x;
  q.diff = (diff && drop) ? diff : nullptr; q.drop = (diff && drop) ? drop : nullptr;
  const int rc = dsh_schwarp_fit_batch(ctx, 1, &q);
  if (rc != DSH_OK) return rc;
  if (info) { info[0] = q.info[0]; info[1] = q.info[1]; }
  if (costs) { costs[0] = q.costs[0]; costs[1] = q.costs[1]; }
  return DSH_OK;
}

}  // extern "C"
