// Stand-alone reproducer attempt for the packed-fp32 corruption found by the two-stream inference forward
// (profiles/r04_pipeline_notes.md section 4): a VALU-only kernel built on v_pk_fma_f32 (A) runs on one
// stream while an MFMA kernel small enough to share its CUs (B) runs on another; A's output is compared bit for bit with
// what A produced alone.  Build twice:   hipcc -O3 --offload-arch=gfx950 pk_hazard.hip -o pk_hazard
//                                        hipcc -O3 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops ... -o pk_hazard_nopk
// usage: pk_hazard [rounds] [B workgroups] [B iterations] [order]
// RESULT SO FAR: does NOT reproduce (0 wrong of 40 runs in every order / grid tried, packed and scalar builds alike): the
// synthetic pair is missing whatever the real pair has (the fused MBConv kernel beside conv_patch_kernel<1,...>:
// scripts/concurrency_bisect.py reproduces it with the library built WITH packed fp32).  Kept as the starting point.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// A: the shape of the fused MBConv kernel's expand phase -- per thread a quad of output channels, CIN broadcast inputs from
// LDS, weights in registers, FMAs as f32x4 (two v_pk_fma_f32 each), swish on the hardware transcendentals; 48 KB of LDS
// per 256-thread workgroup so that two or three of them and one workgroup of B fit a CU together.
constexpr int CIN = 16;
__global__ __launch_bounds__(256, 2) void valu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ out, int pixels_per_wg, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  f32x4 wq[CIN];
#pragma unroll
  for (int k = 0; k < CIN; ++k) wq[k] = *reinterpret_cast<const f32x4*>(w + (k * 64 + (tid & 63)) * 4);
  const float* xb = x + (size_t)blockIdx.x * pixels_per_wg * CIN;
  for (int i = tid; i < pixels_per_wg * CIN; i += 256) lds[i] = xb[i];
  __syncthreads();
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    for (int p = tid >> 6; p < pixels_per_wg; p += 4) {
      f32x4 acc = {0.1f, 0.2f, 0.3f, 0.4f};
      const float* xp = lds + p * CIN;
#pragma unroll
      for (int kq = 0; kq < CIN / 4; ++kq) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_elementwise_fma(wq[4 * kq + j], f32x4{xv[j], xv[j], xv[j], xv[j]}, acc);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = acc[j] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(acc[j] * -1.44269504088896341f));
      sum += acc;
      lds[pixels_per_wg * CIN + (p * 64 + (tid & 63)) % 4096] = acc[0];     // keep an LDS write stream going, as the ring does
    }
    __syncthreads();
  }
  *reinterpret_cast<f32x4*>(out + ((size_t)blockIdx.x * 256 + tid) * 4) = sum;
}

// B: an MFMA loop in the shape of the 1x1 conv kernel (512 threads, ~61 KB of LDS, operands re-read from LDS every step,
// weight tiles refilled by LDS-DMA, one barrier per step): shares CUs with A.
__global__ __launch_bounds__(512, 2) void mfma_kernel(const float* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 60 * 1024 / 16; i += 512)
    reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.001f * (i & 31), 0.002f, 0.003f, 0.004f};
  __syncthreads();
  f32x16 acc0 = {}, acc1 = {};
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)(it & 63) * 512 + tid) * 4),
                                     (__attribute__((address_space(3))) void*)(smem + 32 * 1024 + wave * 1024), 16, 0, 0);
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(smem + ((u * 64 + lane) * 16 + wave * 6144) % (32 * 1024));
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + 32 * 1024 + ((u * 64 + lane) * 16) % (24 * 1024));
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float sacc = 0.f;
  for (int i = 0; i < 16; ++i) sacc += acc0[i] + acc1[i];
  out[(size_t)blockIdx.x * 512 + tid] = sacc;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 40;
  const int nwgA = 4096, ppw = 528, itersA = 6;              // ~48 KB of LDS: 528 * 16 * 4 + 16 KB scratch
  const size_t ldsA = (size_t)(ppw * CIN + 4096) * 4, ldsB = 61 * 1024;
  const int nwgB = argc > 2 ? atoi(argv[2]) : 1024, itersB = argc > 3 ? atoi(argv[3]) : 300;
  const int order = argc > 4 ? atoi(argv[4]) : 0;            // 0: B, A, B   1: A, B   2: B only before
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
  std::vector<float> hx((size_t)nwgA * ppw * CIN), hw(CIN * 64 * 4);
  unsigned r = 12345u;
  auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd();
  float *dx, *dw, *dout, *dref, *dB, *dsrc;
  CHECK(hipMalloc(&dsrc, 64 * 512 * 16)); CHECK(hipMemset(dsrc, 0, 64 * 512 * 16));
  CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dw, hw.size() * 4));
  const size_t nout = (size_t)nwgA * 256 * 4;
  CHECK(hipMalloc(&dout, nout * 4)); CHECK(hipMalloc(&dref, nout * 4)); CHECK(hipMalloc(&dB, (size_t)nwgB * 512 * 4));
  CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s[9];
  for (auto& st : s) CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  // reference: A alone
  valu_kernel<<<nwgA, 256, ldsA, s[0]>>>(dx, dw, dref, ppw, itersA);
  CHECK(hipDeviceSynchronize());
  std::vector<float> ref(nout), got(nout);
  CHECK(hipMemcpy(ref.data(), dref, nout * 4, hipMemcpyDeviceToHost));
  int alone_bad = 0, beside_bad = 0;
  long wrong_values = 0;
  for (int mode = 0; mode < 2; ++mode) {            // 0: A alone again (control), 1: A beside B (B on each of the other streams in turn)
    for (int it = 0; it < rounds; ++it) {
      CHECK(hipMemsetAsync(dout, 0, nout * 4, s[0]));
      if (mode && order != 1) mfma_kernel<<<nwgB, 512, ldsB, s[1 + it % 8]>>>(dsrc, dB, itersB);
      valu_kernel<<<nwgA, 256, ldsA, s[0]>>>(dx, dw, dout, ppw, itersA);
      if (mode && order != 2) mfma_kernel<<<nwgB, 512, ldsB, s[1 + (it + 3) % 8]>>>(dsrc, dB, itersB);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(got.data(), dout, nout * 4, hipMemcpyDeviceToHost));
      long bad = 0;
      for (size_t i = 0; i < nout; ++i) bad += memcmp(&got[i], &ref[i], 4) != 0;
      if (bad) { (mode ? beside_bad : alone_bad)++; wrong_values += bad; }
    }
  }
  printf("A alone: %d of %d runs differ from the reference; A beside the MFMA kernel: %d of %d runs differ (%ld wrong values in all)\n",
         alone_bad, rounds, beside_bad, rounds, wrong_values);
  return 0;
}
