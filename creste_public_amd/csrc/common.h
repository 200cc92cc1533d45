// Shared helpers for the gfx950 kernels of libcreste_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/creste_hip.h"

namespace creste {

void set_error(const char* fmt, ...);

#define CRESTE_REQUIRE(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      ::creste::set_error(__VA_ARGS__);      \
      return CRESTE_ERR_ARG;                 \
    }                                        \
  } while (0)

#define CRESTE_CHECK_LAUNCH(name)                                                   \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      ::creste::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
      return CRESTE_ERR_HIP;                                                        \
    }                                                                               \
  } while (0)

#define CRESTE_HIP(call)                                                            \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      ::creste::set_error("%s failed: %s", #call, hipGetErrorString(e_));           \
      return CRESTE_ERR_HIP;                                                        \
    }                                                                               \
  } while (0)

constexpr int kNumXcd = 8;  // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

// Bijective remap of a 1-D block id so that each XCD (private L2) receives a CONTIGUOUS range of
// logical tile ids: neighbouring tiles (shared halos / shared weight panels) hit the same L2.
// Speed only -- correctness never depends on placement.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk / kNumXcd, r = nblk % kNumXcd;
  const int xcd = bid % kNumXcd, idx = bid / kNumXcd;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == CRESTE_ACT_RELU) return fmaxf(v, 0.f);
  if (act == CRESTE_ACT_SWISH) return v / (1.f + expf(-v));
  return v;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace creste
