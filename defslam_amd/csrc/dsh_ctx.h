// Internal definition of the opaque dsh_ctx (shared by the translation units of libdefslam_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "dsh_template.h"
#include "sft_problem.h"

namespace dsh {
struct PackedSft;   // defined in dsh_api.cpp
}

// Grow-only device scratch of a context: the one-shot calls (BBS, normals, Schwarp) carve their temporaries out of it
// instead of paying a dozen hipMalloc/hipFree per call.  reset() at the start of a call, release() in dsh_destroy.
// A context serves one host thread at a time (it also owns one stream).
struct dsh_scratch {
  std::vector<std::pair<char*, size_t>> chunks;
  size_t chunk = 0, off = 0;
  void reset() { chunk = 0; off = 0; }
  hipError_t take(size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    for (; chunk < chunks.size(); chunk++, off = 0)
      if (off + bytes <= chunks[chunk].second) {
        *out = chunks[chunk].first + off;
        off += bytes;
        return hipSuccess;
      }
    const size_t cap = bytes > ((size_t)8 << 20) ? bytes : ((size_t)8 << 20);
    char* p = nullptr;
    const hipError_t e = hipMalloc((void**)&p, cap);
    if (e != hipSuccess) return e;
    chunks.emplace_back(p, cap);
    chunk = chunks.size() - 1;
    *out = p;
    off = bytes;
    return hipSuccess;
  }
  void release() {
    for (auto& c : chunks) (void)hipFree(c.first);
    chunks.clear();
    reset();
  }
};

// Page-locked host buffer for the one-copy-in / one-copy-out calls (grow-only; plain memory for a host-only context).
struct dsh_pinned {
  char* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    const size_t want = bytes + bytes / 4 + 4096;
    const hipError_t e = hipHostMalloc((void**)&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct dsh_ctx_base {
  dsh_scratch scratch;
  dsh_pinned pin_in, pin_out;
  int device = 0;
  bool host_only = false;   // device == -1: template + packer only (CPU tests of the host logic)
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<struct dsh_diffdb*> diffdbs;   // databases created on this context: dsh_destroy detaches them (dsh_diffdb.cpp: ddb_detach_all)
};
void ddb_detach_all(dsh_ctx_base* c);

// error helper usable from every translation unit
inline int dsh_fail(dsh_ctx_base* c, int code, const std::string& m) {
  if (c) c->err = m;
  return code;
}
