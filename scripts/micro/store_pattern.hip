// Store-pattern microbenchmark (GPU box): write a [P][C] fp32 tensor with SEG lanes covering one contiguous run.
// build+run: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// lanes per contiguous segment = SEG (each lane 16 B); a wave writes 64/SEG segments at stride `cs` floats
template <int SEG>
__global__ __launch_bounds__(256) void k(float* out, long P, int C, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % SEG, row = lane / SEG;          // row within the wave's group of pixels
  constexpr int RPW = 64 / SEG;                          // pixels per wave-instruction
  const int cq = C / 4;
  const long nwave = (long)gridDim.x * 4;
  f4 v = {1.f, 2.f, 3.f, (float)lane};
  for (long p0 = ((long)blockIdx.x * 4 + wave) * RPW; p0 < P; p0 += nwave * RPW) {
    const long p = p0 + row;
    if (p >= P) continue;
    for (int q = sub; q < cq; q += SEG) *reinterpret_cast<f4*>(out + p * C + q * 4) = v;
  }
}
template <int SEG> void run(float* d, long P, int C) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<SEG><<<4096, 256>>>(d, P, C, 1); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) k<SEG><<<4096, 256>>>(d, P, C, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("C=%d SEG=%2d (%4d B runs): %.3f ms  %.2f TB/s\n", C, SEG, SEG * 16, ms, P * C * 4.0 / ms / 1e9);
}
int main() {
  for (int C : {96, 144, 496}) {
    const long P = 16L * 304 * 608 * 96 / C;
    float* d; hipMalloc(&d, P * C * 4);
    run<1>(d, P, C); run<8>(d, P, C); run<32>(d, P, C); run<64>(d, P, C);
    hipFree(d);
  }
  return 0;
}
