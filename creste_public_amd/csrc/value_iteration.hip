// Value iteration on the 8-connected stochastic grid MDP (reference creste/models/blocks/vin.py:36-80).
//
//   v <- 0 ; repeat { x = r + gamma*v ; q_a = sum_taps w_a[tap]*x[neighbour] (zero outside the grid) ;
//   v' = max_a q_a ; delta = max_{batch,grid} |v' - v| ; v <- v' } until !(delta > threshold) ;
//   q = eval_q(r, v) ; policy = softmax_a(q)          (Jacobi sweeps, hard max, batch-global test)
//
// HBM-bound in principle (12 B/cell/sweep) but the whole state (r, v ping-pong: 12 B/cell, 0.8 MB for
// 8x64x128, 6 MB for 8x256x256) lives in L2/Infinity Cache, so the cost is launch/sync latency.  One
// launch per sweep; the batch-global convergence test is a device-side max (atomicMax on the float
// bits, deltas are >= 0) that the NEXT sweep's launch reads after the kernel boundary: once
// !(delta > threshold) every later launch is a no-op, so no host round trip per sweep (the reference
// does one `.item()` per sweep).  The host peeks at the state every kSweepsPerPeek launches.
#include <string.h>

#include "common.h"

namespace creste {

// Action a moves by DYN[a] = (drow, dcol); its 0.8 tap sits at window position 1+DYN[a]; the two
// 0.1 taps are the ring neighbours of that position (vin.py:36-46).  Taps are listed in row-major
// window order -- the accumulation order of a direct 3x3 cross-correlation.
struct Tap { int dy, dx; float w; };
constexpr Tap kTaps[8][3] = {
    {{-1, -1, 0.8f}, {-1, 0, 0.1f}, {0, -1, 0.1f}},   // a0 NW
    {{-1, -1, 0.1f}, {-1, 0, 0.8f}, {-1, 1, 0.1f}},   // a1 N
    {{-1, 0, 0.1f}, {-1, 1, 0.8f}, {0, 1, 0.1f}},     // a2 NE
    {{-1, -1, 0.1f}, {0, -1, 0.8f}, {1, -1, 0.1f}},   // a3 W
    {{-1, 1, 0.1f}, {0, 1, 0.8f}, {1, 1, 0.1f}},      // a4 E
    {{0, -1, 0.1f}, {1, -1, 0.8f}, {1, 0, 0.1f}},     // a5 SW
    {{1, -1, 0.1f}, {1, 0, 0.8f}, {1, 1, 0.1f}},      // a6 S
    {{0, 1, 0.1f}, {1, 0, 0.1f}, {1, 1, 0.8f}},       // a7 SE
};

struct ViState {
  int converged_at;   // sweeps run when the test first failed, -1 while still iterating
  int pad[3];
  // followed by unsigned delta_bits[max_sweeps]: max |v'-v| of sweep k (float bits), zero-initialised
};
__device__ __host__ __forceinline__ unsigned* vi_delta(ViState* st) {
  return reinterpret_cast<unsigned*>(st + 1);
}

constexpr int TW = 64, TH = 16;   // tile of cells per 256-thread workgroup (4 rows per thread)

__device__ __forceinline__ void q_values(const float (*xs)[TW + 2], int ly, int lx, float q[8]) {
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
      s = __fmaf_rn(kTaps[a][t].w, xs[ly + 1 + kTaps[a][t].dy][lx + 1 + kTaps[a][t].dx], s);
    q[a] = s;
  }
}

__device__ __forceinline__ void load_x_tile(float (*xs)[TW + 2], const float* __restrict__ r,
                                            const float* __restrict__ v, int H, int W, int y0, int x0,
                                            float gamma) {
  for (int i = threadIdx.x; i < (TH + 2) * (TW + 2); i += 256) {
    const int ly = i / (TW + 2), lx = i % (TW + 2);
    const int y = y0 + ly - 1, x = x0 + lx - 1;
    float val = 0.f;                                    // zero padding of the conv
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
      val = __fadd_rn(r[(long)y * W + x], __fmul_rn(v[(long)y * W + x], gamma));
    xs[ly][lx] = val;
  }
}

__global__ __launch_bounds__(256) void vi_sweep_kernel(const float* __restrict__ r,
                                                       const float* __restrict__ vin,
                                                       float* __restrict__ vout, ViState* st, int k,
                                                       int H, int W, float gamma, float threshold) {
  if (k > 0) {
    const float prev = __uint_as_float(vi_delta(st)[k - 1]);
    if (!(prev > threshold)) {          // converged (or never ran): this and all later sweeps are no-ops
      if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 &&
          st->converged_at < 0)
        st->converged_at = k;
      return;
    }
  }
  __shared__ float xs[TH + 2][TW + 2];
  __shared__ float red[4];
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const long plane = (long)b * H * W;
  load_x_tile(xs, r + plane, vin + plane, H, W, y0, x0, gamma);
  __syncthreads();
  const int lx = threadIdx.x & 63, lyb = threadIdx.x >> 6;
  float dmax = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ly = lyb * 4 + j, y = y0 + ly, x = x0 + lx;
    if (y < H && x < W) {
      float q[8];
      q_values(xs, ly, lx, q);
      float m = q[0];
#pragma unroll
      for (int a = 1; a < 8; ++a) m = fmaxf(m, q[a]);
      const float old = vin[plane + (long)y * W + x];
      vout[plane + (long)y * W + x] = m;
      dmax = fmaxf(dmax, fabsf(__fsub_rn(m, old)));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(&vi_delta(st)[k], __float_as_uint(m));   // non-negative floats order like uints
  }
}

__global__ __launch_bounds__(256) void vi_final_kernel(const float* __restrict__ r,
                                                       const float* __restrict__ buf0,
                                                       const float* __restrict__ buf1, ViState* st,
                                                       int max_sweeps, int H, int W, float gamma,
                                                       float threshold, float* __restrict__ v_out, float* __restrict__ q_out,
                                                       float* __restrict__ pi_out, int32_t* sweeps_out) {
  __shared__ float xs[TH + 2][TW + 2];
  int conv = st->converged_at;
  if (conv < 0 && !(__uint_as_float(vi_delta(st)[max_sweeps - 1]) > threshold)) conv = max_sweeps;
  const int nsweep = conv < 0 ? max_sweeps : conv;
  const float* v = (nsweep & 1) ? buf1 : buf0;
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const long plane = (long)b * H * W;
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    *sweeps_out = conv < 0 ? -max_sweeps : conv;
  load_x_tile(xs, r + plane, v + plane, H, W, y0, x0, gamma);
  __syncthreads();
  const int lx = threadIdx.x & 63, lyb = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ly = lyb * 4 + j, y = y0 + ly, x = x0 + lx;
    if (y < H && x < W) {
      float q[8];
      q_values(xs, ly, lx, q);
      float m = q[0];
#pragma unroll
      for (int a = 1; a < 8; ++a) m = fmaxf(m, q[a]);
      float e[8], s = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) { e[a] = expf(__fsub_rn(q[a], m)); s = __fadd_rn(s, e[a]); }
      const long c = (long)y * W + x;
      v_out[plane + c] = v[plane + c];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        q_out[((long)b * 8 + a) * H * W + c] = q[a];
        pi_out[((long)b * 8 + a) * H * W + c] = __fdiv_rn(e[a], s);
      }
    }
  }
}

constexpr int kSweepsPerPeek = 128;

}  // namespace creste

using namespace creste;

static inline size_t vi_align(size_t x) { return (x + 255) / 256 * 256; }

extern "C" int64_t creste_value_iteration_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return -1;
  // two v buffers + state header + one delta word per sweep (bounded by 1<<20 sweeps)
  return (int64_t)(2 * vi_align((size_t)B * H * W * 4) + vi_align(sizeof(ViState) + 4u * (1u << 20)));
}

extern "C" int creste_value_iteration_f32(const float* r, int B, int H, int W, float discount,
                                          float threshold, int max_sweeps, float* v, float* q,
                                          float* policy, int32_t* sweeps_out, void* work, void* stream) {
  CRESTE_REQUIRE(r && v && q && policy && sweeps_out && work, "value_iteration: null pointer");
  CRESTE_REQUIRE(B > 0 && H > 0 && W > 0, "value_iteration: bad dims");
  CRESTE_REQUIRE(max_sweeps > 0 && max_sweeps <= (1 << 20), "value_iteration: max_sweeps out of range");
  hipStream_t s = (hipStream_t)stream;
  char* wp = (char*)work;
  float* buf0 = (float*)wp;
  float* buf1 = (float*)(wp + vi_align((size_t)B * H * W * 4));
  ViState* st = (ViState*)(wp + 2 * vi_align((size_t)B * H * W * 4));
  CRESTE_HIP(hipMemsetAsync(buf0, 0, (size_t)B * H * W * 4, s));
  CRESTE_HIP(hipMemsetAsync(st, 0, sizeof(ViState) + 4u * (size_t)max_sweeps, s));
  const int neg1 = -1;
  CRESTE_HIP(hipMemcpyAsync(&st->converged_at, &neg1, 4, hipMemcpyHostToDevice, s));
  const dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
  int k = 0, conv = -1;
  while (k < max_sweeps && conv < 0) {
    const int end = (k + kSweepsPerPeek < max_sweeps) ? k + kSweepsPerPeek : max_sweeps;
    for (; k < end; ++k) {
      const float* vin = (k & 1) ? buf1 : buf0;
      float* vout = (k & 1) ? buf0 : buf1;
      vi_sweep_kernel<<<grid, 256, 0, s>>>(r, vin, vout, st, k, H, W, discount, threshold);
    }
    CRESTE_CHECK_LAUNCH("vi_sweep");
    // one extra no-op launch settles `converged_at` when the very last sweep of the chunk converged
    if (k < max_sweeps) {
      vi_sweep_kernel<<<dim3(1, 1, 1), 256, 0, s>>>(r, buf0, buf1, st, k, 0, 0, discount, threshold);
    }
    CRESTE_HIP(hipMemcpyAsync(&conv, &st->converged_at, 4, hipMemcpyDeviceToHost, s));
    CRESTE_HIP(hipStreamSynchronize(s));
  }
  vi_final_kernel<<<grid, 256, 0, s>>>(r, buf0, buf1, st, max_sweeps, H, W, discount, threshold, v, q, policy, sweeps_out);
  CRESTE_CHECK_LAUNCH("vi_final");
  if (conv < 0) {
    // last sweep may have converged exactly at max_sweeps; check its delta on the host
    unsigned bits = 0;
    CRESTE_HIP(hipMemcpyAsync(&bits, vi_delta(st) + (max_sweeps - 1), 4, hipMemcpyDeviceToHost, s));
    CRESTE_HIP(hipStreamSynchronize(s));
    float d; memcpy(&d, &bits, 4);
    if (d > threshold) {
      set_error("value_iteration: no convergence in %d sweeps (last delta %g)", max_sweeps, (double)d);
      return CRESTE_ERR_NOCONV;
    }
  }
  return CRESTE_OK;
}
