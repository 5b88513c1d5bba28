// C ABI of the device-resident DiffProp database (include/defslam_hip.h: dsh_diffdb_*, dsh_normals_estimate_db): the records of
// SchwarpDatabase::calculateSchwarps stay in HBM between the Schwarp fits and NormalEstimator::ObtainK1K2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/defslam_hip.h"
#include "dsh_ctx.h"
#include "dsh_diffdb.h"

extern "C" hipError_t ddb_group(long long, const int32_t*, int, const int32_t*, int, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, void*, int32_t*, hipStream_t);
extern "C" hipError_t ddb_gather(int, const int32_t*, const float*, const int32_t*, const int32_t*, float*, int32_t*, int32_t*, hipStream_t);
extern "C" size_t ddb_group_tmp_bytes(int);
extern "C" hipError_t nrsfm_launch_normals(int, int, const int32_t*, const int32_t*, const float*, const uint8_t*, const float*, const uint8_t*, const float*,
                                           const uint8_t*, const float*, double*, double*, double*, int32_t*, float*, float*, uint8_t*, int32_t*, hipStream_t);

namespace {
#define HIPCHK(c, call)                                                                                        \
  do {                                                                                                         \
    hipError_t e__ = (call);                                                                                   \
    if (e__ != hipSuccess) {                                                                                   \
      (void)hipStreamSynchronize((c)->stream);                                                                  \
      return dsh_fail(c, DSH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));                      \
    }                                                                                                           \
  } while (0)
struct DevBuf {
  void* p = nullptr;
  hipError_t alloc(dsh_ctx_base* c, size_t bytes) { return c->scratch.take(bytes, &p); }
  template <class T> T* as() { return static_cast<T*>(p); }
};
}  // namespace

extern "C" {

int dsh_diffdb_create(dsh_ctx* ctx, int64_t capacity, dsh_diffdb** out) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c || !out || capacity <= 0 || capacity > (int64_t)1 << 30) return dsh_fail(c, DSH_ERR_ARG, "dsh_diffdb_create: bad argument");
  *out = nullptr;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_diffdb_create: host-only context, no GPU (there is no CPU fallback)");
  (void)hipSetDevice(c->device);
  dsh_diffdb* db = new dsh_diffdb();
  db->ctx = c;
  db->device = c->device;
  db->cap = capacity;
  char* base = nullptr;
  const size_t bytes = (size_t)capacity * (72 + 12);
  if (hipMalloc((void**)&base, bytes) != hipSuccess) { delete db; return dsh_fail(c, DSH_ERR_HIP, "dsh_diffdb_create: out of device memory"); }
  db->rec = reinterpret_cast<float*>(base);
  db->pid = reinterpret_cast<int32_t*>(base + (size_t)capacity * 72);
  db->tag = db->pid + capacity;
  db->idx2 = db->tag + capacity;
  c->diffdbs.push_back(db);
  *out = db;
  return DSH_OK;
}

// Works in either order with dsh_destroy of the context: the database remembers its device, waits for the whole device (the context's
// stream may be gone) and never dereferences a context that has been destroyed (dsh_destroy detaches its databases).
int dsh_diffdb_destroy(dsh_diffdb* db) {
  if (!db) return DSH_ERR_ARG;
  (void)hipSetDevice(db->device);
  (void)hipDeviceSynchronize();
  if (db->ctx) {
    auto& v = db->ctx->diffdbs;
    v.erase(std::remove(v.begin(), v.end(), db), v.end());
  }
  if (db->rec) (void)hipFree(db->rec);
  if (db->last_normals) (void)hipFree(db->last_normals);
  delete db;
  return DSH_OK;
}

}  // extern "C"

void ddb_detach_all(dsh_ctx_base* c) {
  for (dsh_diffdb* db : c->diffdbs) db->ctx = nullptr;
  c->diffdbs.clear();
}

int ddb_reserve(dsh_diffdb* db, long long need) {
  if (need <= db->cap) return 0;
  if (need > (1ll << 30)) return (int)hipErrorOutOfMemory;
  const long long ncap = std::min<long long>(std::max(need, 2 * db->cap), 1ll << 30);
  (void)hipSetDevice(db->device);
  char* base = nullptr;
  hipError_t e = hipMalloc((void**)&base, (size_t)ncap * (72 + 12));
  if (e != hipSuccess) return (int)e;
  float* rec = reinterpret_cast<float*>(base);
  int32_t* pid = reinterpret_cast<int32_t*>(base + (size_t)ncap * 72);
  int32_t *tag = pid + ncap, *idx2 = tag + ncap;
  (void)hipDeviceSynchronize();   // nothing of this device may still read or write the old arrays
  const size_t n = (size_t)db->count;
  if (n) {
    e = hipMemcpy(rec, db->rec, 72 * n, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemcpy(pid, db->pid, 4 * n, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemcpy(tag, db->tag, 4 * n, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemcpy(idx2, db->idx2, 4 * n, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { (void)hipFree(base); return (int)e; }
  }
  (void)hipFree(db->rec);
  db->rec = rec; db->pid = pid; db->tag = tag; db->idx2 = idx2; db->cap = ncap;
  return 0;
}

extern "C" {

int dsh_diffdb_clear(dsh_diffdb* db) {
  if (!db) return DSH_ERR_ARG;
  db->count = 0;
  db->max_pid = -1;
  db->last_P = db->last_R = 0;
  return DSH_OK;
}

int64_t dsh_diffdb_count(const dsh_diffdb* db) { return db ? (int64_t)db->count : -1; }

int dsh_diffdb_append(dsh_diffdb* db, int n, const dsh_diffprop* recs, const int32_t* point_id, const int32_t* tag, const int32_t* idx2) {
  if (!db || !db->ctx) return DSH_ERR_ARG;
  dsh_ctx_base* c = db->ctx;
  if (n < 0 || (n > 0 && (!recs || !point_id))) return dsh_fail(c, DSH_ERR_ARG, "dsh_diffdb_append: bad argument");
  if (n == 0) return DSH_OK;
  (void)hipSetDevice(c->device);
  if (ddb_reserve(db, db->count + n) != 0) return dsh_fail(c, DSH_ERR_HIP, "dsh_diffdb_append: out of device memory while growing the database");
  hipStream_t st = c->stream;
  std::vector<int32_t> fill;
  const std::vector<int32_t> zeros(tag ? 0 : (size_t)n, 0);   // lives until the stream synchronisation below
  HIPCHK(c, hipMemcpyAsync(db->rec + 18 * (size_t)db->count, recs, 72 * (size_t)n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(db->pid + db->count, point_id, 4 * (size_t)n, hipMemcpyHostToDevice, st));
  if (!tag || !idx2) { fill.assign(n, 0); if (!idx2) for (int i = 0; i < n; i++) fill[i] = i; }
  HIPCHK(c, hipMemcpyAsync(db->tag + db->count, tag ? tag : zeros.data(), 4 * (size_t)n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(db->idx2 + db->count, idx2 ? idx2 : fill.data(), 4 * (size_t)n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipStreamSynchronize(st));
  for (int i = 0; i < n; i++) db->max_pid = std::max(db->max_pid, point_id[i]);
  db->count += n;
  return DSH_OK;
}

int dsh_normals_estimate_db(dsh_ctx* ctx, dsh_diffdb* db, int P, const int32_t* point_ids, const float* x0, const uint8_t* has_x0, const float* ref_uv,
                            double* k1k2, double* cov, int32_t* status, float* normal_ref, int32_t* iters, int32_t max_rec, int32_t* n_rec, int32_t* rec_point,
                            int32_t* rec_tag, int32_t* rec_idx2, float* normal_rec, uint8_t* rec_written) {
  dsh_ctx_base* c = reinterpret_cast<dsh_ctx_base*>(ctx);
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return dsh_fail(c, DSH_ERR_NO_DEVICE, "dsh_normals_estimate_db: host-only context, no GPU (there is no CPU fallback)");
  if (!db || db->ctx != c || P < 0 || (P > 0 && (!point_ids || !x0 || !has_x0 || !ref_uv || !k1k2 || !status)) || max_rec < 0)
    return dsh_fail(c, DSH_ERR_ARG, "dsh_normals_estimate_db: bad argument");
  if (n_rec) *n_rec = 0;
  db->last_P = db->last_R = 0;
  if (P == 0) return DSH_OK;
  if (hipSetDevice(c->device) != hipSuccess) return dsh_fail(c, DSH_ERR_HIP, "dsh_normals_estimate_db: hipSetDevice failed");
  c->scratch.reset();
  hipStream_t st = c->stream;
  const long long n = db->count;
  const size_t nn = n > 0 ? (size_t)n : 1;
  if (db->max_pid == INT32_MAX) return dsh_fail(c, DSH_ERR_ARG, "dsh_normals_estimate_db: point id 2^31 - 1 is not supported");
  const int nlook = std::max(db->max_pid + 1, 1);
  DevBuf d_ids, d_look, d_key, d_count, d_cursor, d_perm, d_owner, d_tmp, d_ptr, d_x0, d_hx0, d_uv;
  HIPCHK(c, d_ids.alloc(c, 4 * (size_t)P)); HIPCHK(c, d_look.alloc(c, 4 * (size_t)nlook)); HIPCHK(c, d_key.alloc(c, 4 * nn)); HIPCHK(c, d_count.alloc(c, 4 * (size_t)(P + 1)));
  HIPCHK(c, d_cursor.alloc(c, 4 * (size_t)(P + 1))); HIPCHK(c, d_perm.alloc(c, 4 * nn)); HIPCHK(c, d_owner.alloc(c, 4 * nn)); HIPCHK(c, d_tmp.alloc(c, ddb_group_tmp_bytes(P)));
  HIPCHK(c, d_ptr.alloc(c, 4 * (size_t)(P + 2)));
  HIPCHK(c, d_x0.alloc(c, 8 * (size_t)P)); HIPCHK(c, d_hx0.alloc(c, P)); HIPCHK(c, d_uv.alloc(c, 8 * (size_t)P));
  HIPCHK(c, hipMemcpyAsync(d_ids.p, point_ids, 4 * (size_t)P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_x0.p, x0, 8 * (size_t)P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_hx0.p, has_x0, P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_uv.p, ref_uv, 8 * (size_t)P, hipMemcpyHostToDevice, st));
  // group the database by the requested points (a point's records keep their insertion order)
  HIPCHK(c, ddb_group(n, db->pid, P, d_ids.as<int32_t>(), nlook, d_look.as<int32_t>(), d_key.as<int32_t>(), d_count.as<int32_t>(), d_cursor.as<int32_t>(),
                      d_perm.as<int32_t>(), d_owner.as<int32_t>(), d_tmp.p, d_ptr.as<int32_t>(), st));
  int32_t R = 0;   // records that belong to a requested point: the one number the host needs before it can size the launches
  HIPCHK(c, hipMemcpyAsync(&R, d_ptr.as<int32_t>() + P, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (n_rec) *n_rec = R;
  if ((rec_point || rec_tag || rec_idx2 || normal_rec || rec_written) && R > max_rec)
    return dsh_fail(c, DSH_ERR_ARG, "dsh_normals_estimate_db: " + std::to_string(R) + " records, the per-record buffers hold " + std::to_string(max_rec));
  const size_t Rn = R > 0 ? (size_t)R : 1;
  DevBuf d_soa, d_tag, d_idx2, d_isref, d_fn, d_hfn, d_Q, d_k, d_cov, d_st, d_nref, d_nrec, d_wr, d_it;
  HIPCHK(c, d_soa.alloc(c, 72 * Rn)); HIPCHK(c, d_tag.alloc(c, 4 * Rn)); HIPCHK(c, d_idx2.alloc(c, 4 * Rn));
  HIPCHK(c, d_isref.alloc(c, Rn)); HIPCHK(c, d_fn.alloc(c, 8 * Rn)); HIPCHK(c, d_hfn.alloc(c, Rn)); HIPCHK(c, d_Q.alloc(c, 8 * 20 * Rn));
  HIPCHK(c, d_k.alloc(c, 16 * (size_t)P)); HIPCHK(c, d_cov.alloc(c, 32 * (size_t)P)); HIPCHK(c, d_st.alloc(c, 4 * (size_t)P)); HIPCHK(c, d_nref.alloc(c, 12 * (size_t)P));
  HIPCHK(c, d_nrec.alloc(c, 12 * Rn)); HIPCHK(c, d_wr.alloc(c, Rn)); HIPCHK(c, d_it.alloc(c, 4 * (size_t)P));
  HIPCHK(c, ddb_gather(R, d_perm.as<int32_t>(), db->rec, db->tag, db->idx2, d_soa.as<float>(), d_tag.as<int32_t>(), d_idx2.as<int32_t>(), st));
  // every stored record is anchored in its point's reference keyframe (SchwarpDatabase.cc:297: records of other points are not saved)
  HIPCHK(c, hipMemsetAsync(d_isref.p, 1, Rn, st));
  HIPCHK(c, hipMemsetAsync(d_fn.p, 0, 8 * Rn, st));
  HIPCHK(c, hipMemsetAsync(d_hfn.p, 0, Rn, st));
  HIPCHK(c, hipMemsetAsync(d_cov.p, 0, 32 * (size_t)P, st));
  HIPCHK(c, hipMemsetAsync(d_nref.p, 0, 12 * (size_t)P, st));
  HIPCHK(c, hipMemsetAsync(d_nrec.p, 0, 12 * Rn, st));
  HIPCHK(c, nrsfm_launch_normals(P, R, d_ptr.as<int32_t>(), d_owner.as<int32_t>(), d_soa.as<float>(), d_isref.as<uint8_t>(), d_fn.as<float>(), d_hfn.as<uint8_t>(),
                                 d_x0.as<float>(), d_hx0.as<uint8_t>(), d_uv.as<float>(), d_Q.as<double>(), d_k.as<double>(), d_cov.as<double>(), d_st.as<int32_t>(),
                                 d_nref.as<float>(), d_nrec.as<float>(), d_wr.as<uint8_t>(), d_it.as<int32_t>(), st));
  // the normals stay behind for dsh_sfn_estimate_db
  const long long need = 3ll * P + 3ll * R;
  if (need > db->last_cap) {
    HIPCHK(c, hipStreamSynchronize(st));
    if (db->last_normals) (void)hipFree(db->last_normals);
    db->last_normals = nullptr; db->last_cap = 0;
    if (hipMalloc((void**)&db->last_normals, 4 * (size_t)(need + need / 2)) != hipSuccess) return dsh_fail(c, DSH_ERR_HIP, "dsh_normals_estimate_db: out of device memory");
    db->last_cap = need + need / 2;
  }
  HIPCHK(c, hipMemcpyAsync(db->last_normals, d_nref.p, 12 * (size_t)P, hipMemcpyDeviceToDevice, st));
  if (R > 0) HIPCHK(c, hipMemcpyAsync(db->last_normals + 3 * (size_t)P, d_nrec.p, 12 * (size_t)R, hipMemcpyDeviceToDevice, st));
  HIPCHK(c, hipMemcpyAsync(k1k2, d_k.p, 16 * (size_t)P, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(status, d_st.p, 4 * (size_t)P, hipMemcpyDeviceToHost, st));
  if (cov) HIPCHK(c, hipMemcpyAsync(cov, d_cov.p, 32 * (size_t)P, hipMemcpyDeviceToHost, st));
  if (normal_ref) HIPCHK(c, hipMemcpyAsync(normal_ref, d_nref.p, 12 * (size_t)P, hipMemcpyDeviceToHost, st));
  if (iters) HIPCHK(c, hipMemcpyAsync(iters, d_it.p, 4 * (size_t)P, hipMemcpyDeviceToHost, st));
  if (R > 0) {
    if (rec_point) HIPCHK(c, hipMemcpyAsync(rec_point, d_owner.p, 4 * Rn, hipMemcpyDeviceToHost, st));
    if (rec_tag) HIPCHK(c, hipMemcpyAsync(rec_tag, d_tag.p, 4 * Rn, hipMemcpyDeviceToHost, st));
    if (rec_idx2) HIPCHK(c, hipMemcpyAsync(rec_idx2, d_idx2.p, 4 * Rn, hipMemcpyDeviceToHost, st));
    if (normal_rec) HIPCHK(c, hipMemcpyAsync(normal_rec, d_nrec.p, 12 * Rn, hipMemcpyDeviceToHost, st));
    if (rec_written) HIPCHK(c, hipMemcpyAsync(rec_written, d_wr.p, Rn, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipStreamSynchronize(st));
  db->last_P = P; db->last_R = R;
  return DSH_OK;
}

}  // extern "C"
