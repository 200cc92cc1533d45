// Expected state-visitation frequency under the sharpened policy + greedy rollout
// (reference creste/models/lfd.py:156-277, creste/utils/train_utils.py:765-803).
//
//   S_t   = clamp(floor_div(expert_xy_t, ds))                       expert poses on the IRL grid
//   S0    = first S_t inside the fov mask, else (H-1, W/2)
//   pi'   = softmax_a((pi - max_a pi) / temperature)                 ("sharpen")
//   mu_0  = onehot(S0);  mu_t[s'] = sum_a pi'_a[s' - d_a] * mu_{t-1}[s' - d_a]   (mass leaving the grid is dropped)
//   exp_svf = sum_t mu_t ;  greedy rollout s <- clamp(s + d_{argmax_a pi[s]}) for T steps
//
// The reference runs 49 python iterations of (clone, mul, depthwise conv over the WHOLE [B,8,H,W]
// grid, sum).  Mass starts in one cell and moves one cell per step, so mu_t is supported on a
// (2t+1)^2 window around S0: one workgroup per sample keeps mu_{t-1}, mu_t and the running sum of a
// (2T-1)^2 window in LDS (117 KB for T=50, of the CU's 160 KB) and never touches HBM between
// steps; only pi' (L2-resident) is re-read.  HBM traffic = policy once + exp_svf once.
#include "common.h"

namespace creste {

constexpr int kDyn[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};

// torch floor-division of floats (c10::div_floor_floating)
__device__ __forceinline__ float floor_div(float a, float b) {
  if (b == 0.f) return a / b;
  const float mod = fmodf(a, b);
  float div = __fdiv_rn(__fsub_rn(a, mod), b);
  if (mod != 0.f && ((b < 0.f) != (mod < 0.f))) div = __fsub_rn(div, 1.f);
  if (div != 0.f) {
    float f = floorf(div);
    if (__fsub_rn(div, f) > 0.5f) f = __fadd_rn(f, 1.f);
    return f;
  }
  return copysignf(0.f, a / b);
}

__global__ __launch_bounds__(256) void sharpen_policy_kernel(const float* __restrict__ pi,
                                                             float* __restrict__ out, long B, long HW,
                                                             float temperature, int sharpen) {
  const long total = B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, c = i % HW;
    float p[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) p[a] = pi[(b * 8 + a) * HW + c];
    if (sharpen) {
      float m = p[0];
#pragma unroll
      for (int a = 1; a < 8; ++a) m = fmaxf(m, p[a]);
      float l[8], lm;
#pragma unroll
      for (int a = 0; a < 8; ++a) l[a] = __fdiv_rn(__fsub_rn(p[a], m), temperature);
      lm = l[0];
#pragma unroll
      for (int a = 1; a < 8; ++a) lm = fmaxf(lm, l[a]);
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) { p[a] = expf(__fsub_rn(l[a], lm)); s = __fadd_rn(s, p[a]); }
#pragma unroll
      for (int a = 0; a < 8; ++a) p[a] = __fdiv_rn(p[a], s);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) out[(b * 8 + a) * HW + c] = p[a];
  }
}

__global__ __launch_bounds__(1024) void svf_kernel(const float* __restrict__ pi,
                                                   const float* __restrict__ spi,
                                                   const float* __restrict__ expert_xy,
                                                   const uint8_t* __restrict__ fov, int H, int W, int T,
                                                   int Te, float ds, int zero_terminal,
                                                   float* __restrict__ exp_svf,
                                                   int64_t* __restrict__ state_preds,
                                                   float* __restrict__ state_grid) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int s_start[2], s_term[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const long HW = (long)H * W;
  const float* exb = expert_xy + (long)b * Te * 2;

  if (tid == 0) {
    int r0 = H - 1, c0 = W / 2;
    bool found = false;
    int lr = 0, lc = 0;
    for (int t = 0; t < Te; ++t) {
      long r = (long)floor_div(exb[t * 2 + 0], ds), c = (long)floor_div(exb[t * 2 + 1], ds);
      r = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
      c = c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
      if (!found && fov[r * W + c] == 1) { found = true; r0 = (int)r; c0 = (int)c; }
      lr = (int)r; lc = (int)c;
    }
    s_start[0] = r0; s_start[1] = c0; s_term[0] = lr; s_term[1] = lc;
  }
  __syncthreads();
  const int r0 = s_start[0], c0 = s_start[1];
  const int R = T - 1;                     // window radius
  const int wy0 = max(0, r0 - R), wy1 = min(H - 1, r0 + R);
  const int wx0 = max(0, c0 - R), wx1 = min(W - 1, c0 + R);
  const int wh = wy1 - wy0 + 1, ww = wx1 - wx0 + 1, wn = wh * ww;
  float* mu_a = sm;
  float* mu_b = sm + wn;
  float* acc = sm + 2 * wn;
  for (int i = tid; i < wn; i += 1024) { mu_a[i] = 0.f; mu_b[i] = 0.f; acc[i] = 0.f; }
  __syncthreads();
  if (tid == 0) { mu_a[(r0 - wy0) * ww + (c0 - wx0)] = 1.f; }
  __syncthreads();
  const float* sp = spi + (long)b * 8 * HW;
  const int ty = s_term[0] - wy0, tx = s_term[1] - wx0;
  const bool term_in = (unsigned)ty < (unsigned)wh && (unsigned)tx < (unsigned)ww;

  // acc accumulates mu_0 .. mu_{T-1} in that order; mu_{t-1} is added at the top of step t, AFTER the
  // optional terminal-state zeroing, because the reference zeroes the stored mu_{t-1} that it later sums.
  float* prev = mu_a;
  float* next = mu_b;
  for (int t = 1; t < T; ++t) {
    __syncthreads();
    if (zero_terminal && term_in && tid == 0) prev[ty * ww + tx] = 0.f;
    __syncthreads();
    // mu_t is supported within radius t of S0 (and mu_{t-1} within radius t-1)
    const int y_lo = max(wy0, r0 - t), y_hi = min(wy1, r0 + t);
    const int x_lo = max(wx0, c0 - t), x_hi = min(wx1, c0 + t);
    const int bw = x_hi - x_lo + 1, bn = (y_hi - y_lo + 1) * bw;
    for (int i = tid; i < bn; i += 1024) {
      const int y = y_lo + i / bw, x = x_lo + i % bw;
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int sy = y - kDyn[a][0], sx = x - kDyn[a][1];     // source cell that moves onto (y,x) with a
        if (sy >= wy0 && sy <= wy1 && sx >= wx0 && sx <= wx1) {
          const float m = prev[(sy - wy0) * ww + (sx - wx0)];
          if (m != 0.f) s = __fadd_rn(s, __fmul_rn(sp[(long)a * HW + (long)sy * W + sx], m));
        }
      }
      const int li = (y - wy0) * ww + (x - wx0);
      acc[li] = __fadd_rn(acc[li], prev[li]);
      next[li] = s;
    }
    float* tmp = prev; prev = next; next = tmp;
  }
  __syncthreads();
  for (int i = tid; i < wn; i += 1024) acc[i] = __fadd_rn(acc[i], prev[i]);
  __syncthreads();
  float* out = exp_svf + (long)b * HW;
  float* grid = state_grid + (long)b * HW;
  for (long i = tid; i < HW; i += 1024) {
    const int y = (int)(i / W), x = (int)(i % W);
    float v = 0.f;
    if (y >= wy0 && y <= wy1 && x >= wx0 && x <= wx1) v = acc[(y - wy0) * ww + (x - wx0)];
    out[i] = v;
    grid[i] = 0.f;
  }
  __syncthreads();
  __threadfence_block();
  if (tid == 0) {
    const float* pb = pi + (long)b * 8 * HW;
    int64_t* sp_out = state_preds + (long)b * T * 2;
    int y = r0, x = c0;
    sp_out[0] = y; sp_out[1] = x;
    grid[(long)y * W + x] += 1.f;
    for (int t = 1; t < T; ++t) {
      const long c = (long)y * W + x;
      int best = 0; float bv = pb[c];
      for (int a = 1; a < 8; ++a) { const float v = pb[(long)a * HW + c]; if (v > bv) { bv = v; best = a; } }
      y += kDyn[best][0]; x += kDyn[best][1];
      y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
      x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
      sp_out[t * 2 + 0] = y; sp_out[t * 2 + 1] = x;
      grid[(long)y * W + x] += 1.f;
    }
  }
}

}  // namespace creste

using namespace creste;

extern "C" int creste_expected_svf_f32(const float* policy, const float* expert_xy, const uint8_t* fov,
                                       int B, int H, int W, int T, int T_expert, float ds, float temperature,
                                       int sharpen, int zero_terminal, float* sharp_policy,
                                       float* exp_svf, int64_t* state_preds, float* state_grid,
                                       void* stream) {
  CRESTE_REQUIRE(policy && expert_xy && fov && sharp_policy && exp_svf && state_preds && state_grid,
                 "expected_svf: null pointer");
  CRESTE_REQUIRE(B > 0 && H > 0 && W > 0 && T > 0 && T_expert > 0 && ds > 0.f, "expected_svf: bad dims");
  CRESTE_REQUIRE(!sharpen || temperature > 0.f, "expected_svf: temperature must be positive");
  const long win = (long)(2 * T - 1);
  const long wn = (win < H ? win : H) * (win < W ? win : W);
  const size_t smem = (size_t)wn * 3 * sizeof(float);
  CRESTE_REQUIRE(smem <= 160 * 1024 - 64, "expected_svf: window of horizon %d does not fit LDS", T);
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)B * H * W;
  sharpen_policy_kernel<<<(int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256), 256, 0, s>>>(
      policy, sharp_policy, B, (long)H * W, temperature, sharpen);
  CRESTE_CHECK_LAUNCH("sharpen_policy");
  if (smem > 64 * 1024)     // the size depends on T: set per call (per-device attribute, cheap)
    CRESTE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(svf_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  svf_kernel<<<B, 1024, smem, s>>>(policy, sharp_policy, expert_xy, fov, H, W, T, T_expert, ds, zero_terminal,
                                   exp_svf, state_preds, state_grid);
  CRESTE_CHECK_LAUNCH("svf");
  return CRESTE_OK;
}
