// Internal definition of the opaque dsh_ctx (shared by the translation units of libdefslam_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "dsh_template.h"
#include "sft_problem.h"

namespace dsh {
struct PackedSft;   // defined in dsh_api.cpp
}

struct dsh_ctx_base {
  int device = 0;
  bool host_only = false;   // device == -1: template + packer only (CPU tests of the host logic)
  hipStream_t stream = nullptr;
  std::string err;
};

// error helper usable from every translation unit
inline int dsh_fail(dsh_ctx_base* c, int code, const std::string& m) {
  if (c) c->err = m;
  return code;
}
