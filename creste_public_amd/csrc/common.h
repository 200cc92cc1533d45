// Shared helpers for the gfx950 kernels of libcreste_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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

// max over the 64 lanes of a wave (result in every lane)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Raise the device-wide running maximum *amax (non-negative float, zero-initialised) to the block's
// max of `v`.  `scratch` = LDS floats, one per wave, free to overwrite; every thread of the block
// must call.  Positive floats order like their bit patterns, so the update is an integer atomicMax; a
// relaxed agent-scope read first makes the atomic rare (the maximum only rises O(log blocks) times).
__device__ __forceinline__ void block_amax_update(float v, float* amax, float* scratch) {
  v = wave_max(v);
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < nw; ++i) v = fmaxf(v, scratch[i]);
    const unsigned bits = __float_as_uint(v);
    unsigned* slot = reinterpret_cast<unsigned*>(amax);
    if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
  }
}

// F16X3 operand scale: the power of two that maps an upper bound `amax` of |x| into [2^14, 2^15)
// (fp16 max 65504); *inv gets its reciprocal.  amax == 0 / denormal -> 1.
__device__ __forceinline__ float f16_operand_scale(float amax, float* inv) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xff);      // biased exponent: amax in [2^(e-127), 2^(e-126))
  if (e == 0) { *inv = 1.f; return 1.f; }
  int se = 127 + 14 - (e - 127);                                   // biased exponent of the scale
  se = se > 253 ? 253 : (se < 1 ? 1 : se);
  *inv = __uint_as_float((unsigned)(254 - se) << 23);
  return __uint_as_float((unsigned)se << 23);
}

// hipFuncAttributeMaxDynamicSharedMemorySize applies to the CURRENT device: remember per device (bit i of `mask`,
// one static mask per kernel instantiation) whether the limit was raised there.  Thread-safe; devices >= 64 set it
// on every launch.
inline hipError_t ensure_dyn_smem(const void* fn, int bytes, std::atomic<uint64_t>& mask) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0ull;
  if (bit && (mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && bit) mask.fetch_or(bit, std::memory_order_release);
  return e;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace creste
