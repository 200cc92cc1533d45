// Trajectory rasterisation and scoring against a costmap -- the step AFTER the hot path.
//
// reference: creste/utils/loss_utils.py:1054-1116 (`compute_expert_visitation`: every polyline segment sampled at
// max_steps = max over the call of ceil(|segment|) points of torch.linspace(0, 1, max_steps), the last pose appended,
// coordinates clamped to the grid and truncated, every visited cell counted once) and :1197-1258 (the reward a
// trajectory collects = sum over its visited cells of the costmap: `(svf * reward).sum(dim=(1, 2))`, the quantity the
// counterfactual IRL loss compares between candidate trajectories).  The candidates come from the Ackermann sampler of
// scripts/traversability/planner_utils/control.py:12-118 (host RNG, mirrored in creste_public_amd/planner.py); the
// local planner that consumes the scores lives in the C++ sister repository.
//
// One workgroup per trajectory: the visited set is a bitmap in LDS (one bit per cell), so a cell counts once however
// many samples land on it; the score is the sum of the costmap over the set bits in row-major order (deterministic).
#include "common.h"

namespace creste {

__device__ __forceinline__ float seg_len(const float* a, const float* b, float ds) {
  const float dx = __fsub_rn(__fdiv_rn(b[0], ds), __fdiv_rn(a[0], ds));
  const float dy = __fsub_rn(__fdiv_rn(b[1], ds), __fdiv_rn(a[1], ds));
  return sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
}

// *max_steps = max over all segments of ceil(length) (device int, zero-initialised by the launcher)
// `group` (may be NULL = one group): trajectories rasterised by ONE reference call share max_steps; a batch of calls
// (the expert set and every sample's counterfactual set of MaxEntIRLLoss) is one launch with a group id per trajectory.
__global__ __launch_bounds__(256) void traj_max_steps_kernel(const float* __restrict__ xy, long nseg_total, int T,
                                                             float ds, const int* __restrict__ group,
                                                             int* __restrict__ max_steps) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nseg_total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / (T - 1), s = i - n * (T - 1);
    const float* a = xy + (n * T + s) * 2;
    const float len = ceilf(seg_len(a, a + 2, ds));
    if (len > 0.f) atomicMax(&max_steps[group ? group[n] : 0], len >= 2147483000.f ? 2147483000 : (int)len);
  }
}

__device__ __forceinline__ int clamp_cell(float v, int n) {
  v = fminf(fmaxf(v, 0.f), (float)(n - 1));          // .clamp(0, n-1).long(): truncation of a non-negative value
  return (int)v;
}

__global__ __launch_bounds__(256) void traj_score_kernel(const float* __restrict__ xy, int T, float ds, int H, int W,
                                                         const float* __restrict__ costmap,
                                                         const int* __restrict__ map_index, long map_stride,
                                                         const int* __restrict__ group,
                                                         const int* __restrict__ max_steps_p,
                                                         float* __restrict__ scores, float* __restrict__ visit,
                                                         int* __restrict__ n_cells) {
  extern __shared__ unsigned s_bits[];               // ceil(H*W/32) words
  __shared__ float s_part[4];
  __shared__ int s_cnt[4];
  const int n = blockIdx.x, t = threadIdx.x;
  const int HW = H * W, nw = (HW + 31) / 32;
  for (int i = t; i < nw; i += 256) s_bits[i] = 0u;
  __syncthreads();
  const int steps = max_steps_p[group ? group[n] : 0];
  const float* p = xy + (long)n * T * 2;
  const float step = steps > 1 ? __fdiv_rn(1.f, (float)(steps - 1)) : 0.f;       // torch.linspace(0, 1, steps)
  const long total = (long)(T - 1) * steps;
  for (long i = t; i < total; i += 256) {
    const int s = (int)(i / steps), j = (int)(i - (long)s * steps);
    const float ax = __fdiv_rn(p[s * 2 + 0], ds), ay = __fdiv_rn(p[s * 2 + 1], ds);
    const float bx = __fdiv_rn(p[s * 2 + 2], ds), by = __fdiv_rn(p[s * 2 + 3], ds);
    // linspace: start + step*j in the first half, end - step*(steps-1-j) in the second (ATen's symmetric form)
    const float lam = steps == 1 ? 0.f : (j < steps / 2 ? __fmul_rn(step, (float)j)
                                                         : __fsub_rn(1.f, __fmul_rn(step, (float)(steps - 1 - j))));
    const float px = __fadd_rn(ax, __fmul_rn(lam, __fsub_rn(bx, ax)));
    const float py = __fadd_rn(ay, __fmul_rn(lam, __fsub_rn(by, ay)));
    const int c = clamp_cell(px, H) * W + clamp_cell(py, W);
    atomicOr(&s_bits[c >> 5], 1u << (c & 31));
  }
  if (t == 0) {                                      // the last pose
    const int c = clamp_cell(__fdiv_rn(p[(T - 1) * 2 + 0], ds), H) * W + clamp_cell(__fdiv_rn(p[(T - 1) * 2 + 1], ds), W);
    atomicOr(&s_bits[c >> 5], 1u << (c & 31));
  }
  __syncthreads();
  const float* cm = costmap ? costmap + (map_index ? (long)map_index[n] : (long)n) * map_stride : nullptr;
  float acc = 0.f;
  int cnt = 0;
  // thread t owns the contiguous word range [t*wpt, (t+1)*wpt): a row-major walk; partial sums combine in thread order
  const int wpt = (nw + 255) / 256;
  for (int w = t * wpt; w < min(nw, (t + 1) * wpt); ++w) {
    unsigned bits = s_bits[w];
    if (visit)
      for (int k = 0; k < 32 && w * 32 + k < HW; ++k) visit[(long)n * HW + w * 32 + k] = (bits >> k) & 1u ? 1.f : 0.f;
    while (bits) {
      const int k = __ffs(bits) - 1;
      bits &= bits - 1;
      if (cm) acc = __fadd_rn(acc, cm[w * 32 + k]);
      ++cnt;
    }
  }
  // deterministic tree over the 256 partials
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    acc = __fadd_rn(acc, __shfl_xor(acc, o));
    cnt += __shfl_xor(cnt, o);
  }
  if ((t & 63) == 0) { s_part[t >> 6] = acc; s_cnt[t >> 6] = cnt; }
  __syncthreads();
  if (t == 0) {
    scores[n] = __fadd_rn(__fadd_rn(s_part[0], s_part[1]), __fadd_rn(s_part[2], s_part[3]));
    if (n_cells) n_cells[n] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  }
}

}  // namespace creste

using namespace creste;

extern "C" int creste_trajectory_scores_grouped_f32(const float* xy, int N, int T, float map_ds, int H, int W,
                                                    const float* costmap, const int* map_index, int64_t map_stride,
                                                    const int* group, int n_groups, float* scores, float* visit,
                                                    int* n_cells, int* work, void* stream) {
  CRESTE_REQUIRE(xy && scores && work && (costmap || visit), "trajectory_scores: null pointer");
  CRESTE_REQUIRE(n_groups >= 1 && (group || n_groups == 1), "trajectory_scores: bad grouping");
  CRESTE_REQUIRE(N > 0 && T >= 1 && H > 0 && W > 0 && map_ds > 0.f, "trajectory_scores: bad dims");
  // one bit per cell in LDS: bounded by what THIS device grants a workgroup (160 KiB on gfx950)
  int dev = 0, lds_max = 0;
  CRESTE_HIP(hipGetDevice(&dev));
  CRESTE_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
  const long max_cells = (long)(lds_max < 147456 ? lds_max : 147456) * 8;
  CRESTE_REQUIRE((long)H * W <= max_cells, "trajectory_scores: grid %dx%d does not fit the %d-byte LDS bitmap of this device", H,
                 W, (int)(max_cells / 8));
  hipStream_t s = (hipStream_t)stream;
  CRESTE_HIP(hipMemsetAsync(work, 0, sizeof(int) * (size_t)n_groups, s));
  if (T > 1) {
    const long nseg = (long)N * (T - 1);
    traj_max_steps_kernel<<<(int)((nseg + 255) / 256 > 1024 ? 1024 : (nseg + 255) / 256), 256, 0, s>>>(xy, nseg, T, map_ds, group, work);
    CRESTE_CHECK_LAUNCH("traj_max_steps");
  }
  const size_t smem = (size_t)(((long)H * W + 31) / 32) * sizeof(unsigned);
  if (smem > 64 * 1024)
    CRESTE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(traj_score_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  traj_score_kernel<<<N, 256, smem, s>>>(xy, T, map_ds, H, W, costmap, map_index, map_stride, group, work, scores, visit, n_cells);
  CRESTE_CHECK_LAUNCH("traj_score");
  return CRESTE_OK;
}

extern "C" int creste_trajectory_scores_f32(const float* xy, int N, int T, float map_ds, int H, int W,
                                            const float* costmap, const int* map_index, int64_t map_stride,
                                            float* scores, float* visit, int* n_cells, int* work, void* stream) {
  return creste_trajectory_scores_grouped_f32(xy, N, T, map_ds, H, W, costmap, map_index, map_stride, nullptr, 1, scores,
                                              visit, n_cells, work, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Visitation bookkeeping of the MaxEnt / counterfactual IRL objective in ONE launch (reference
// creste/utils/loss_utils.py:1139-1186): field-of-view masking, L1 normalisation (+1e-5) of the expert visitation and
// of the policy's expected visitation, and per sample with counterfactuals the mix
//   exp_svf[i] <- alpha * cf_svf + (1 - alpha) * exp_svf[i],  cf_svf = normalise(sum of the sample's rasterised
// sub-optimal trajectories).  The reference does this with ~12 tensor ops per sample in a Python loop.
namespace creste {

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void irl_visitation_mix_kernel(const float* __restrict__ exp_raw,
                                                                 const uint8_t* __restrict__ fov,
                                                                 const float* __restrict__ visit_expert,
                                                                 const float* __restrict__ visit_cf,
                                                                 const int* __restrict__ cf_ptr, float alpha, long HW,
                                                                 float* __restrict__ svf_out, float* __restrict__ exp_out,
                                                                 float* __restrict__ cf_out, float* __restrict__ policy_out) {
  __shared__ float sm[4];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* er = exp_raw + (long)b * HW;
  const float* ve = visit_expert + (long)b * HW;
  const uint8_t* fv = fov ? fov + (long)b * HW : nullptr;
  const int c0 = cf_ptr ? cf_ptr[b] : 0, c1 = cf_ptr ? cf_ptr[b + 1] : 0;
  float s_svf = 0.f, s_exp = 0.f, s_cf = 0.f;
  for (long i = t; i < HW; i += 256) {
    const float m = fv ? (fv[i] ? 1.f : 0.f) : 1.f;
    s_svf += ve[i] * m;
    s_exp += er[i] * m;
    float c = 0.f;
    for (int k = c0; k < c1; ++k) c += visit_cf[(long)k * HW + i];
    s_cf += c;
  }
  const float d_svf = block_sum_256(s_svf, sm) + 1e-5f;
  const float d_exp = block_sum_256(s_exp, sm) + 1e-5f;
  const float d_cf = block_sum_256(s_cf, sm) + 1e-5f;
  const float beta = 1.f - alpha;
  for (long i = t; i < HW; i += 256) {
    const float m = fv ? (fv[i] ? 1.f : 0.f) : 1.f;
    const float sv = __fdiv_rn(ve[i] * m, d_svf), ex = __fdiv_rn(er[i] * m, d_exp);
    float c = 0.f;
    for (int k = c0; k < c1; ++k) c += visit_cf[(long)k * HW + i];
    const float cf = c1 > c0 ? __fdiv_rn(c, d_cf) : 0.f;
    svf_out[(long)b * HW + i] = sv;
    policy_out[(long)b * HW + i] = ex;
    cf_out[(long)b * HW + i] = cf;
    exp_out[(long)b * HW + i] = c1 > c0 ? __fadd_rn(__fmul_rn(alpha, cf), __fmul_rn(beta, ex)) : ex;
  }
}

}  // namespace creste

extern "C" int creste_irl_visitation_mix_f32(const float* exp_svf_raw, const uint8_t* fov, const float* visit_expert,
                                             const float* visit_cf, const int* cf_ptr, float alpha, int B, int64_t HW,
                                             float* svf, float* exp_svf, float* cf_total, float* policy_svf, void* stream) {
  CRESTE_REQUIRE(exp_svf_raw && visit_expert && svf && exp_svf && cf_total && policy_svf && B > 0 && HW > 0,
                 "irl_visitation_mix: bad args");
  CRESTE_REQUIRE(!cf_ptr || visit_cf, "irl_visitation_mix: cf_ptr without visitation maps");
  creste::irl_visitation_mix_kernel<<<B, 256, 0, (hipStream_t)stream>>>(exp_svf_raw, fov, visit_expert, visit_cf, cf_ptr, alpha,
                                                                        HW, svf, exp_svf, cf_total, policy_svf);
  CRESTE_CHECK_LAUNCH("irl_visitation_mix");
  return CRESTE_OK;
}
