// Internal definition of the opaque dsh_diffdb (device-resident DiffProp database, diffdb_kernels.hip).
#pragma once
#include <cstdint>

struct dsh_ctx_base;
struct dsh_diffdb {
  dsh_ctx_base* ctx = nullptr;   // the owning context; set to null by dsh_destroy of that context (the database can then only be destroyed)
  int device = 0;                // HIP device of the allocation: dsh_diffdb_destroy needs nothing else
  long long cap = 0, count = 0;
  int32_t max_pid = -1;          // largest point id stored so far (size of the lookup table of a grouping)
  float* rec = nullptr;          // cap x 18 float32: the DiffProp fields in the order of dsh_diffprop
  int32_t *pid = nullptr, *tag = nullptr, *idx2 = nullptr;   // map point, caller's pair tag, key point index in the second keyframe
  // normals of the last dsh_normals_estimate_db, kept for dsh_sfn_estimate_db: per requested point (reference keyframe) and per record
  float* last_normals = nullptr;   // [3 last_P | 3 last_R] float32
  long long last_cap = 0;          // floats allocated
  int last_P = 0, last_R = 0;
};

// Room for `need` records in total: the database grows on demand (the reference's mapPointsDB_ is unbounded) -- a new allocation of at
// least twice the capacity, the stored records copied device to device.  Returns a HIP error code (0 = ok).
int ddb_reserve(dsh_diffdb* db, long long need);
