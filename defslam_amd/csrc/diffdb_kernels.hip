// Device-resident DiffProp database (the GPU side of defSLAM::WarpDatabase::mapPointsDB_, Modules/Mapping/WarpDatabase.h:61): the
// records SchwarpDatabase::calculateSchwarps produces (SchwarpDatabase.cc:299-345) stay in HBM between the fit and
// NormalEstimator::ObtainK1K2 (NormalEstimator.cc:38-229) instead of travelling to the host map and back.
//   append ........ records of the fits of one call, in (fit, match) order: positions by an exclusive scan of the keep flags
//                   (deterministic: the insertion order is what orders a point's residual blocks)
//   group ......... the records of the requested map points, a counting sort on the request index of the record's point: histogram,
//                   exclusive scan (= the row pointers), scatter through per-point cursors, then every point's record indices sorted
//                   ascending -- the index in the database IS the insertion order, so that restores it whatever order the atomics gave;
//                   records gathered + transposed into the SoA layout of the normals kernels
// The scan is written out here (three launches, 1024 items per block): no library call, nothing that reads the environment.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

__global__ void ddb_keep_kernel(int n, const uint8_t* __restrict__ drop, const int32_t* __restrict__ pid, int32_t* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keep[i] = (!drop[i] && pid[i] >= 0) ? 1 : 0;
}
// record i of the call (if kept) -> slot base + pos[i]
__global__ void ddb_store_kernel(int n, const int32_t* __restrict__ keep, const int32_t* __restrict__ pos, const float* __restrict__ diff,
                                 const int32_t* __restrict__ pid, const int32_t* __restrict__ tag, const int32_t* __restrict__ idx2, long long base,
                                 long long cap, float* __restrict__ rec, int32_t* __restrict__ dpid, int32_t* __restrict__ dtag, int32_t* __restrict__ didx2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !keep[i]) return;
  const long long j = base + pos[i];
  if (j >= cap) return;
#pragma unroll
  for (int k = 0; k < 18; k++) rec[18 * j + k] = diff[18 * (size_t)i + k];
  dpid[j] = pid[i]; dtag[j] = tag[i]; didx2[j] = idx2[i];
}
__global__ void ddb_fill_kernel(int n, int32_t v, int32_t* __restrict__ a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}
__global__ void ddb_scatter_lookup_kernel(int P, const int32_t* __restrict__ point_ids, int nlook, int32_t* __restrict__ lookup) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P && point_ids[p] >= 0 && point_ids[p] < nlook) lookup[point_ids[p]] = p;
}
// histogram of the request index q of every record's point (records of points that were not asked for: nowhere); key[j] = q or P
__global__ void ddb_keys_kernel(long long n, const int32_t* __restrict__ dpid, int nlook, const int32_t* __restrict__ lookup, int P, int32_t* __restrict__ key,
                                int32_t* __restrict__ count) {
  const long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int id = dpid[j];
  const int q = (id >= 0 && id < nlook) ? lookup[id] : -1;
  key[j] = q >= 0 ? q : P;
  if (q >= 0) atomicAdd(&count[q], 1);
}
__global__ void ddb_scatter_kernel(long long n, const int32_t* __restrict__ key, int P, const int32_t* __restrict__ rec_ptr, int32_t* __restrict__ cursor,
                                   int32_t* __restrict__ perm, int32_t* __restrict__ owner) {
  const long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int q = key[j];
  if (q >= P) return;
  const int slot = rec_ptr[q] + atomicAdd(&cursor[q], 1);
  perm[slot] = (int32_t)j;
  owner[slot] = q;
}
// one thread per requested point: its record indices ascending (insertion sort for the usual handful, heap sort beyond 32)
__global__ void ddb_order_kernel(int P, const int32_t* __restrict__ rec_ptr, int32_t* __restrict__ perm) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int32_t* a = perm + rec_ptr[p];
  const int m = rec_ptr[p + 1] - rec_ptr[p];
  if (m <= 32) {
    for (int i = 1; i < m; i++) {
      const int32_t x = a[i];
      int k = i - 1;
      while (k >= 0 && a[k] > x) { a[k + 1] = a[k]; k--; }
      a[k + 1] = x;
    }
    return;
  }
  auto sift = [&](int root, int end) {
    for (;;) {
      int child = 2 * root + 1;
      if (child >= end) return;
      if (child + 1 < end && a[child + 1] > a[child]) child++;
      if (a[root] >= a[child]) return;
      const int32_t t = a[root]; a[root] = a[child]; a[child] = t;
      root = child;
    }
  };
  for (int i = m / 2 - 1; i >= 0; i--) sift(i, m);
  for (int e = m - 1; e > 0; e--) { const int32_t t = a[0]; a[0] = a[e]; a[e] = t; sift(0, e); }
}
// sorted position j < R: record perm[j] -> SoA (field k at soa[k * R + j]), tag, idx2
__global__ void ddb_gather_kernel(int R, const int32_t* __restrict__ perm, const float* __restrict__ rec, const int32_t* __restrict__ dtag,
                                  const int32_t* __restrict__ didx2, float* __restrict__ soa, int32_t* __restrict__ otag, int32_t* __restrict__ oidx2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R) return;
  const size_t r = (size_t)perm[j];
#pragma unroll
  for (int k = 0; k < 18; k++) soa[(size_t)k * R + j] = rec[18 * r + k];
  otag[j] = dtag[r];
  oidx2[j] = didx2[r];
}

// ---- exclusive scan of int32 (n items; out[n] = total when with_total): 1024 items per block of 256 threads
constexpr int SCAN_T = 256, SCAN_I = 4, SCAN_B = SCAN_T * SCAN_I;
__device__ inline int block_exclusive(int v, int* lds_w, int& total) {   // exclusive prefix of v over the 256 threads
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
  if (lane == 63) lds_w[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < SCAN_T / 64; k++) { const int s = lds_w[k]; if (k < w) base += s; tot += s; }
  __syncthreads();
  total = tot;
  return base + inc - v;
}
__global__ __launch_bounds__(SCAN_T) void scan_sums_kernel(long long n, const int32_t* __restrict__ in, int32_t* __restrict__ bsum) {
  __shared__ int lds_w[SCAN_T / 64];
  const long long i0 = (long long)blockIdx.x * SCAN_B + (long long)threadIdx.x * SCAN_I;
  int v = 0;
#pragma unroll
  for (int k = 0; k < SCAN_I; k++) if (i0 + k < n) v += in[i0 + k];
  int total;
  (void)block_exclusive(v, lds_w, total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
__global__ __launch_bounds__(SCAN_T) void scan_blocks_kernel(int nb, int32_t* __restrict__ bsum) {   // one block: bsum -> its exclusive scan, in place
  __shared__ int lds_w[SCAN_T / 64];
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += SCAN_T) {
    const int i = b0 + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    int total;
    const int ex = block_exclusive(v, lds_w, total);
    if (i < nb) bsum[i] = carry + ex;
    carry += total;
  }
}
__global__ __launch_bounds__(SCAN_T) void scan_final_kernel(long long n, const int32_t* __restrict__ in, const int32_t* __restrict__ bsum, int32_t* __restrict__ out,
                                                            int with_total) {
  __shared__ int lds_w[SCAN_T / 64];
  const long long i0 = (long long)blockIdx.x * SCAN_B + (long long)threadIdx.x * SCAN_I;
  int x[SCAN_I], v = 0;
#pragma unroll
  for (int k = 0; k < SCAN_I; k++) { x[k] = i0 + k < n ? in[i0 + k] : 0; v += x[k]; }
  int total;
  int run = bsum[blockIdx.x] + block_exclusive(v, lds_w, total);
#pragma unroll
  for (int k = 0; k < SCAN_I; k++) {
    if (i0 + k < n) out[i0 + k] = run;
    run += x[k];
    if (with_total && i0 + k == n - 1) out[n] = run;
  }
}
hipError_t scan_exclusive(long long n, const int32_t* in, int32_t* out, int with_total, int32_t* bsum, hipStream_t st) {
  if (n <= 0) { if (with_total) return hipMemsetAsync(out, 0, 4, st); return hipSuccess; }
  const int nb = (int)((n + SCAN_B - 1) / SCAN_B);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(nb), dim3(SCAN_T), 0, st, n, in, bsum);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(SCAN_T), 0, st, nb, bsum);
  hipLaunchKernelGGL(scan_final_kernel, dim3(nb), dim3(SCAN_T), 0, st, n, in, bsum, out, with_total);
  return hipGetLastError();
}
size_t scan_tmp_bytes(long long n) { return 4 * (size_t)((n > 0 ? n : 1) + SCAN_B - 1) / SCAN_B * 1 + 256; }

// normals for Shape from Normals: sel >= 0 -> the normal of requested point sel, sel < 0 -> the normal propagated by record -1 - sel
__global__ void ddb_pick_normals_kernel(int n, const int32_t* __restrict__ sel, const float* __restrict__ nref, const float* __restrict__ nrec, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = sel[i];
  const float* src = s >= 0 ? nref + 3 * (size_t)s : nrec + 3 * (size_t)(-1 - s);
  out[3 * i] = src[0]; out[3 * i + 1] = src[1]; out[3 * i + 2] = src[2];
}

}  // namespace

extern "C" hipError_t ddb_pick_normals(int n, const int32_t* sel, const float* nref, const float* nrec, float* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(ddb_pick_normals_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, sel, nref, nrec, out);
  return hipGetLastError();
}

extern "C" hipError_t ddb_append(int n, const uint8_t* drop, const float* diff, const int32_t* pid, const int32_t* tag, const int32_t* idx2, int32_t* keep, int32_t* pos,
                                 void* tmp, size_t tmp_bytes, long long base, long long cap, float* rec, int32_t* dpid, int32_t* dtag, int32_t* didx2, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  (void)tmp_bytes;
  const int bl = 256, gr = (n + bl - 1) / bl;
  hipLaunchKernelGGL(ddb_keep_kernel, dim3(gr), dim3(bl), 0, st, n, drop, pid, keep);
  hipError_t e = scan_exclusive(n, keep, pos, 0, static_cast<int32_t*>(tmp), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ddb_store_kernel, dim3(gr), dim3(bl), 0, st, n, keep, pos, diff, pid, tag, idx2, base, cap, rec, dpid, dtag, didx2);
  return hipGetLastError();
}
extern "C" size_t ddb_scan_tmp_bytes(int n) { return scan_tmp_bytes(n); }
extern "C" size_t ddb_group_tmp_bytes(int P) { return scan_tmp_bytes((long long)P + 1); }
// lookup (nlook ints), key (n), count / cursor (P + 1 each), perm / owner (n each), rec_ptr (P + 2): the grouping of the database by the
// requested points; rec_ptr[P] = number of records that belong to a requested point
extern "C" hipError_t ddb_group(long long n, const int32_t* dpid, int P, const int32_t* point_ids, int nlook, int32_t* lookup, int32_t* key, int32_t* count,
                                int32_t* cursor, int32_t* perm, int32_t* owner, void* tmp, int32_t* rec_ptr, hipStream_t st) {
  const int bl = 256;
  hipLaunchKernelGGL(ddb_fill_kernel, dim3((nlook + bl - 1) / bl), dim3(bl), 0, st, nlook, -1, lookup);
  hipLaunchKernelGGL(ddb_scatter_lookup_kernel, dim3((P + bl - 1) / bl), dim3(bl), 0, st, P, point_ids, nlook, lookup);
  hipError_t e = hipMemsetAsync(count, 0, 4 * (size_t)(P + 1), st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(cursor, 0, 4 * (size_t)(P + 1), st);
  if (e != hipSuccess) return e;
  if (n > 0) hipLaunchKernelGGL(ddb_keys_kernel, dim3((unsigned)((n + bl - 1) / bl)), dim3(bl), 0, st, n, dpid, nlook, lookup, P, key, count);
  e = scan_exclusive(P, count, rec_ptr, 1, static_cast<int32_t*>(tmp), st);
  if (e != hipSuccess) return e;
  if (n > 0) {
    hipLaunchKernelGGL(ddb_scatter_kernel, dim3((unsigned)((n + bl - 1) / bl)), dim3(bl), 0, st, n, key, P, rec_ptr, cursor, perm, owner);
    hipLaunchKernelGGL(ddb_order_kernel, dim3((P + bl - 1) / bl), dim3(bl), 0, st, P, rec_ptr, perm);
  }
  return hipGetLastError();
}
extern "C" hipError_t ddb_gather(int R, const int32_t* perm, const float* rec, const int32_t* dtag, const int32_t* didx2, float* soa, int32_t* otag, int32_t* oidx2,
                                 hipStream_t st) {
  if (R <= 0) return hipSuccess;
  hipLaunchKernelGGL(ddb_gather_kernel, dim3((R + 255) / 256), dim3(256), 0, st, R, perm, rec, dtag, didx2, soa, otag, oidx2);
  return hipGetLastError();
}
