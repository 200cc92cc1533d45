// Experiment (profiles/r05_coresidency.md): a bandwidth-bound kernel small enough to share a CU with the persistent
// F(4x4) GEMM workgroup (2 x 200 of 512 VGPRs per SIMD and 120 of 160 KiB of LDS taken): <= 112 VGPRs, <= 40 KiB LDS.
// How fast does it stream beside the GEMM, and what does the GEMM lose?  Built by scripts/coresidency_probe.py:
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC scripts/micro/stream_probe.hip -o scripts/micro/libstream_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n16) {
  extern __shared__ char lds[];
  const size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  f32x4 v[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const size_t i = base + (size_t)u * 256;
    v[u] = i < n16 ? __builtin_nontemporal_load(src + i) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (n16 == 0) lds[threadIdx.x] = 1;      // (keeps the dynamic LDS allocation referenced)
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const size_t i = base + (size_t)u * 256;
    if (i < n16) dst[i] = v[u];
  }
}

extern "C" int stream_probe(const void* src, void* dst, size_t bytes, int lds_bytes, int unroll, void* stream) {
  const size_t n16 = bytes / 16;
  hipStream_t s = (hipStream_t)stream;
  if (unroll == 4) {
    const unsigned grid = (unsigned)((n16 + 256 * 4 - 1) / (256 * 4));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stream_copy_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    stream_copy_kernel<4><<<grid, 256, lds_bytes, s>>>((const f32x4*)src, (f32x4*)dst, n16);
  } else {
    const unsigned grid = (unsigned)((n16 + 256 * 8 - 1) / (256 * 8));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stream_copy_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    stream_copy_kernel<8><<<grid, 256, lds_bytes, s>>>((const f32x4*)src, (f32x4*)dst, n16);
  }
  return (int)hipGetLastError();
}
