// Multi-positive contrastive loss (reference creste/models/losses/supcon_loss.py:56-115, used by SupPixelConLoss,
// loss_utils.py:203-286) without the N x M similarity matrix.
//
//   z_ij  = <f_i, a_j> / T                      f: local (L2-normalised) features [N][D], a: all-gathered features [M][D]
//   self pairs (j == i + self_off) are excluded; positives P_i = { j != self : label_j == label_i }
//   loss_i = w_i * ( lse_i - (1/|P_i|) sum_{j in P_i} z_ij )   if |P_i| > 0, else 0 ;  loss = mean_i loss_i
//   lse_i  = log sum_{j != self} exp(z_ij)
//
// The reference materialises mask, logits_mask, logits, p and log_softmax as N x M fp32 tensors (5 x 4.3 GB at the
// N = 32k samples one SSC batch yields) -- here three streaming sweeps over column tiles staged in LDS:
//   rows   : per-row max, sum-exp, positive count and positive logit sum (online softmax)          -> loss
//   g_rows : g_ij = (w_i/N) * (softmax_ij * [|P_i|>0] - [j in P_i]/|P_i|) ;  grad_f[i] = sum_j g_ij a_j / T
//   g_cols : the same g_ij walked column-wise: grad_a[j] = sum_i g_ij f_i / T      (no atomics, deterministic)
// One thread owns one row (its D <= 64 features in registers); a tile's features are broadcast reads from LDS.
#include "common.h"

namespace creste {

constexpr int MPC_TILE = 64;

template <int D>
__device__ __forceinline__ float mpc_dot(const float (&r)[D], const float* __restrict__ c) {
  // four independent partial sums (two packed-fp32 FMA chains) instead of one D-long dependent chain; the forward
  // statistics and the gradient sweeps both come through here, so they see bit-identical logits
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
  for (int d = 0; d + 3 < D; d += 4) {
    s0 = __builtin_elementwise_fma(f2{r[d], r[d + 1]}, f2{c[d], c[d + 1]}, s0);
    s1 = __builtin_elementwise_fma(f2{r[d + 2], r[d + 3]}, f2{c[d + 2], c[d + 3]}, s1);
  }
  float s = (s0[0] + s0[1]) + (s1[0] + s1[1]);
#pragma unroll
  for (int d = D & ~3; d < D; ++d) s = __fmaf_rn(r[d], c[d], s);
  return s;
}

// exp on the hardware transcendental (v_exp_f32, 1 ulp; the scaled argument adds <= |x| * 6e-8): the library expf is
// ~15 VALU instructions and every pair of the N x M sweep pays one
__device__ __forceinline__ float mpc_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// stats[i] = (max, sumexp, count, possum)
template <int D>
__global__ __launch_bounds__(256) void mpc_rows_kernel(const float* __restrict__ f, const float* __restrict__ a,
                                                       const int64_t* __restrict__ lab_f, const int64_t* __restrict__ lab_a,
                                                       int N, int M, int self_off, float inv_t, float4* __restrict__ stats,
                                                       int per_split) {
  __shared__ float tile[MPC_TILE][D];
  __shared__ int64_t tlab[MPC_TILE];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int c_lo = blockIdx.y * per_split, c_hi = min(M, c_lo + per_split);      // this block's column range
  float r[D];
  const bool ok = i < N;
#pragma unroll
  for (int d = 0; d < D; ++d) r[d] = ok ? f[(size_t)i * D + d] : 0.f;
  const int64_t li = ok ? lab_f[i] : -1;
  float mx = -3.0e38f, se = 0.f, cnt = 0.f, ps = 0.f;
  for (int j0 = c_lo; j0 < c_hi; j0 += MPC_TILE) {
    __syncthreads();
    for (int e = threadIdx.x; e < MPC_TILE * D; e += 256) {
      const int j = j0 + e / D;
      tile[e / D][e % D] = j < c_hi ? a[(size_t)j * D + e % D] : 0.f;
    }
    if (threadIdx.x < MPC_TILE) tlab[threadIdx.x] = j0 + threadIdx.x < c_hi ? lab_a[j0 + threadIdx.x] : -2;
    __syncthreads();
    const int jn = min(MPC_TILE, c_hi - j0);
    for (int jj = 0; jj < jn; ++jj) {
      const int j = j0 + jj;
      if (j == i + self_off) continue;
      const float z = mpc_dot<D>(r, tile[jj]) * inv_t;
      if (z > mx) { se = se * mpc_exp(mx - z) + 1.f; mx = z; }
      else se += mpc_exp(z - mx);
      if (tlab[jj] == li) { cnt += 1.f; ps += z; }
    }
  }
  if (ok) stats[(size_t)blockIdx.y * N + i] = make_float4(mx, se, cnt, ps);
}

// merge the per-split online-softmax partials of every row (fixed split order)
__global__ void mpc_merge_kernel(const float4* __restrict__ part, int N, int nsplit, float4* __restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float mx = -3.0e38f, se = 0.f, cnt = 0.f, ps = 0.f;
  for (int k = 0; k < nsplit; ++k) {
    const float4 p = part[(size_t)k * N + i];
    if (p.y > 0.f) {
      if (p.x > mx) { se = se * mpc_exp(mx - p.x) + p.y; mx = p.x; }
      else se += p.y * mpc_exp(p.x - mx);
    }
    cnt += p.z; ps += p.w;
  }
  stats[i] = make_float4(mx, se, cnt, ps);
}

// loss = mean_i w_i * (lse_i - ps_i / cnt_i) [cnt_i > 0]  : block partials, ordered final sum on the host side kernel
__global__ __launch_bounds__(256) void mpc_loss_kernel(const float4* __restrict__ stats, const float* __restrict__ rw,
                                                       int N, float* __restrict__ partial) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const float4 st = stats[i];
    if (st.z > 0.f) {
      const float li = (st.x + logf(st.y)) - st.w / st.z;
      s += rw ? li * rw[i] : li;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void mpc_loss_final_kernel(const float* __restrict__ partial, int nb, int N, float* __restrict__ loss) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += partial[b];
    loss[0] = s / (float)N;
  }
}

// ROWS = true : thread = local row i, sweeps the gathered columns  -> grad_f[i]
// ROWS = false: thread = gathered column j, sweeps the local rows   -> grad_a[j]
template <int D, bool ROWS>
__global__ __launch_bounds__(256) void mpc_grad_kernel(const float* __restrict__ f, const float* __restrict__ a,
                                                       const int64_t* __restrict__ lab_f, const int64_t* __restrict__ lab_a,
                                                       const float4* __restrict__ stats, const float* __restrict__ rw,
                                                       int N, int M, int self_off, float inv_t, float gscale,
                                                       float* __restrict__ grad, int per_split) {
  __shared__ float tile[MPC_TILE][D];
  __shared__ int64_t tlab[MPC_TILE];
  __shared__ float4 tst[MPC_TILE];
  __shared__ float trw[MPC_TILE];
  const int own_n = ROWS ? N : M, oth_n = ROWS ? M : N;
  const float* own = ROWS ? f : a;
  const float* oth = ROWS ? a : f;
  const int64_t* own_lab = ROWS ? lab_f : lab_a;
  const int64_t* oth_lab = ROWS ? lab_a : lab_f;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool ok = t < own_n;
  float r[D], g[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { r[d] = ok ? own[(size_t)t * D + d] : 0.f; g[d] = 0.f; }
  const int64_t lt = ok ? own_lab[t] : -1;
  float4 st = make_float4(0.f, 1.f, 0.f, 0.f);
  float wi = 0.f;
  if (ROWS && ok) { st = stats[t]; wi = (rw ? rw[t] : 1.f) * gscale; }
  // (max, 1/sumexp, count, 1/count): the pair loop multiplies instead of dividing twice per pair
  auto recip = [](float4 v) { return make_float4(v.x, 1.f / v.y, v.z, v.z > 0.f ? 1.f / v.z : 0.f); };
  st = recip(st);
  const int o_lo = blockIdx.y * per_split, o_hi = min(oth_n, o_lo + per_split);
  for (int k0 = o_lo; k0 < o_hi; k0 += MPC_TILE) {
    __syncthreads();
    for (int e = threadIdx.x; e < MPC_TILE * D; e += 256) {
      const int k = k0 + e / D;
      tile[e / D][e % D] = k < o_hi ? oth[(size_t)k * D + e % D] : 0.f;
    }
    if (threadIdx.x < MPC_TILE) {
      const int k = k0 + threadIdx.x;
      tlab[threadIdx.x] = k < o_hi ? oth_lab[k] : -2;
      if (!ROWS) {
        tst[threadIdx.x] = recip(k < o_hi ? stats[k] : make_float4(0.f, 1.f, 0.f, 0.f));
        trw[threadIdx.x] = (k < o_hi && rw) ? rw[k] : 1.f;
      }
    }
    __syncthreads();
    const int kn = min(MPC_TILE, o_hi - k0);
    for (int kk = 0; kk < kn; ++kk) {
      const int k = k0 + kk;
      const int i = ROWS ? t : k, j = ROWS ? k : t;               // (row, column) of this pair
      if (j == i + self_off) continue;
      const float4 s = ROWS ? st : tst[kk];
      if (s.z <= 0.f) continue;                                   // a row without positives contributes nothing
      const float z = mpc_dot<D>(r, tile[kk]) * inv_t;
      const float w = ROWS ? wi : trw[kk] * gscale;
      float gij = mpc_exp(z - s.x) * s.y;
      if (tlab[kk] == lt) gij -= s.w;
      gij *= w * inv_t;
#pragma unroll
      for (int d = 0; d < D; ++d) g[d] = __fmaf_rn(gij, tile[kk][d], g[d]);
    }
  }
  if (ok) {
    float* dst = grad + ((size_t)blockIdx.y * own_n + t) * D;
#pragma unroll
    for (int d = 0; d < D; ++d) dst[d] = g[d];
  }
}

__global__ void mpc_sum_splits_kernel(const float* __restrict__ part, long n, int nsplit, float* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * n + i];
    out[i] = s;
  }
}

}  // namespace creste

namespace creste {     // csrc/supcon_mfma.hip: the same sweeps on the matrix cores (D = 16, 32, 64)
bool mpc_mfma_supported(int D);
int64_t mpc_mfma_workspace_bytes(int N, int M, int D);
int mpc_mfma_rows(const float* f, const float* a, const int64_t* lab_f, const int64_t* lab_a, const float* rw, int N, int M,
                  int D, int self_off, float inv_t, float4* part, int ns, int per_split, void* work, hipStream_t s);
int mpc_mfma_grad(bool rows, const float4* stats, const float* rw, int N, int M, int D, int self_off, float inv_t,
                  float gscale, float* gpart, int ns, int per_split, void* work, hipStream_t s);
}  // namespace creste

using namespace creste;

static inline int mpc_splits(int rows) {
  const int rb = (rows + 255) / 256;
  int n = (2048 + rb - 1) / rb;
  return n < 1 ? 1 : (n > 16 ? 16 : n);
}

// workspace: stats [N] | split stats [16][N] | loss partials [4096] | split gradients [16][max(N,M)][D] | the packed
// fp16 operands of the matrix-core path (D = 16, 32, 64)
static inline int64_t mpc_base_bytes(int N, int M, int D) {
  const int64_t big = N > M ? N : M;
  return (int64_t)N * 16 * 17 + 4096 * 4 + 16 * big * D * 4;
}

extern "C" int64_t creste_multipos_con_workspace_bytes(int N, int M, int D) {
  if (N <= 0 || M <= 0 || D <= 0) return -1;
  return mpc_base_bytes(N, M, D) + (mpc_mfma_supported(D) ? mpc_mfma_workspace_bytes(N, M, D) : 0);
}

#define CRESTE_MPC_DISPATCH(D_, CALL)            \
  if (D_ == 8) { CALL(8); }                      \
  else if (D_ == 16) { CALL(16); }               \
  else if (D_ == 32) { CALL(32); }               \
  else if (D_ == 64) { CALL(64); }               \
  else { set_error("multipos_con: feature dimension %d not built (8, 16, 32, 64)", D_); return CRESTE_ERR_ARG; }

extern "C" int creste_multipos_con_forward_f32(const float* feats, const float* all_feats, const int64_t* labels,
                                               const int64_t* all_labels, const float* row_weights, int N, int M,
                                               int D, int self_offset, float temperature, float* loss, void* work,
                                               void* stream) {
  CRESTE_REQUIRE(feats && all_feats && labels && all_labels && loss && work && N > 0 && M > 0 && temperature > 0.f,
                 "multipos_con_forward: bad args");
  hipStream_t s = (hipStream_t)stream;
  float4* stats = (float4*)work;
  float4* part = stats + N;
  float* partial = (float*)((char*)work + (size_t)N * 16 * 17);
  const int nb = (N + 255) / 256, ns = mpc_splits(N);
  const int per = ((M + ns - 1) / ns + MPC_TILE - 1) / MPC_TILE * MPC_TILE;
  if (mpc_mfma_supported(D)) {
    const int rc = mpc_mfma_rows(feats, all_feats, labels, all_labels, row_weights, N, M, D, self_offset, 1.f / temperature,
                                 part, ns, per, (char*)work + mpc_base_bytes(N, M, D), s);
    if (rc != CRESTE_OK) return rc;
  } else {
#define CALL(DD) mpc_rows_kernel<DD><<<dim3(nb, ns), 256, 0, s>>>(feats, all_feats, labels, all_labels, N, M, self_offset, 1.f / temperature, part, per)
    CRESTE_MPC_DISPATCH(D, CALL)
#undef CALL
    CRESTE_CHECK_LAUNCH("mpc_rows");
  }
  mpc_merge_kernel<<<nb, 256, 0, s>>>(part, N, ns, stats);
  const int lb = nb < 1024 ? nb : 1024;
  mpc_loss_kernel<<<lb, 256, 0, s>>>(stats, row_weights, N, partial);
  mpc_loss_final_kernel<<<1, 64, 0, s>>>(partial, lb, N, loss);
  CRESTE_CHECK_LAUNCH("mpc_loss");
  return CRESTE_OK;
}

extern "C" int creste_multipos_con_backward_f32(const float* feats, const float* all_feats, const int64_t* labels,
                                                const int64_t* all_labels, const float* row_weights, int N, int M,
                                                int D, int self_offset, float temperature, float grad_scale, void* work,
                                                float* g_feats, float* g_all, void* stream) {
  CRESTE_REQUIRE(feats && all_feats && labels && all_labels && work && g_feats && g_all && N > 0 && M > 0,
                 "multipos_con_backward: bad args");
  hipStream_t s = (hipStream_t)stream;
  const float4* stats = (const float4*)work;
  float* gpart = (float*)((char*)work + (size_t)N * 16 * 17 + 4096 * 4);
  const float gs = grad_scale / (float)N;
  const int ns_r = mpc_splits(N), ns_c = mpc_splits(M);
  const int per_r = ((M + ns_r - 1) / ns_r + MPC_TILE - 1) / MPC_TILE * MPC_TILE;
  const int per_c = ((N + ns_c - 1) / ns_c + MPC_TILE - 1) / MPC_TILE * MPC_TILE;
  if (mpc_mfma_supported(D)) {      // (forward packed the operands into the same workspace)
    void* mw = (char*)work + mpc_base_bytes(N, M, D);
    int rc = mpc_mfma_grad(true, stats, row_weights, N, M, D, self_offset, 1.f / temperature, gs, gpart, ns_r, per_r, mw, s);
    if (rc != CRESTE_OK) return rc;
    mpc_sum_splits_kernel<<<1024, 256, 0, s>>>(gpart, (long)N * D, ns_r, g_feats);
    rc = mpc_mfma_grad(false, stats, row_weights, N, M, D, self_offset, 1.f / temperature, gs, gpart, ns_c, per_c, mw, s);
    if (rc != CRESTE_OK) return rc;
    mpc_sum_splits_kernel<<<1024, 256, 0, s>>>(gpart, (long)M * D, ns_c, g_all);
    CRESTE_CHECK_LAUNCH("mpc_grad");
    return CRESTE_OK;
  }
#define CALL(DD)                                                                                                          \
  mpc_grad_kernel<DD, true><<<dim3((N + 255) / 256, ns_r), 256, 0, s>>>(feats, all_feats, labels, all_labels, stats,    \
                                                                         row_weights, N, M, self_offset, 1.f / temperature, gs, gpart, per_r); \
  mpc_sum_splits_kernel<<<1024, 256, 0, s>>>(gpart, (long)N * DD, ns_r, g_feats);                                         \
  mpc_grad_kernel<DD, false><<<dim3((M + 255) / 256, ns_c), 256, 0, s>>>(feats, all_feats, labels, all_labels, stats,   \
                                                                          row_weights, N, M, self_offset, 1.f / temperature, gs, gpart, per_c); \
  mpc_sum_splits_kernel<<<1024, 256, 0, s>>>(gpart, (long)M * DD, ns_c, g_all)
  CRESTE_MPC_DISPATCH(D, CALL)
#undef CALL
  CRESTE_CHECK_LAUNCH("mpc_grad");
  return CRESTE_OK;
}

// =====================================================================================================================
// BEV classification and regression objectives of the SSC stage, fused (reference creste/utils/loss_utils.py:379-474
// `CrossEntropy`, :576-603 `SmoothL1`, :530-573 `SmoothL1Depth`): masking, label extraction, the loss, its metric and the
// gradient w.r.t. the prediction in streaming passes over the NHWC prediction -- the reference runs boolean-mask gathers
// (`pred.permute(0,2,3,1)[fov, :]`), torch.nn.CrossEntropyLoss / SmoothL1Loss and a second softmax for the metric.
// Deterministic: per-workgroup partial sums combined in index order.
namespace creste {

constexpr int BCE_MAXC = 64;

__device__ __forceinline__ void block_sum4(float v0, float v1, float v2, float v3, float* __restrict__ partial) {
  __shared__ float sm[4][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    v0 += __shfl_xor(v0, o, 64); v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); v3 += __shfl_xor(v3, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[w][0] = v0; sm[w][1] = v1; sm[w][2] = v2; sm[w][3] = v3; }
  __syncthreads();
  if (threadIdx.x < 4)
    partial[(size_t)blockIdx.x * 4 + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// label of pixel p: class_dim >= 0 -> (long) gt[b, class_dim, y, x]; else argmax_c gt[b,c,y,x] / (sum_c gt + eps)
__device__ __forceinline__ int bce_label(const float* __restrict__ gt, int Cg, long HW, long p, int class_dim, float eps) {
  const long b = p / HW, q = p - b * HW;
  const float* g = gt + b * Cg * HW + q;
  if (class_dim >= 0) return (int)(long)g[(long)class_dim * HW];
  float s = 0.f;
  for (int c = 0; c < Cg; ++c) s = __fadd_rn(s, g[(long)c * HW]);
  s = __fadd_rn(s, eps);
  int best = 0;
  float bv = __fdiv_rn(g[0], s);
  for (int c = 1; c < Cg; ++c) {
    const float v = __fdiv_rn(g[(long)c * HW], s);
    if (v > bv) { bv = v; best = c; }
  }
  return best;
}

// MODE 0: partial sums (w*nll, w, correct, labelled);  MODE 1: gradient (needs out4[2] = sum of weights)
// A label outside [0, C) that is not ignore_index (a 255 "unlabelled" value without ignore_index, a negative / NaN float
// label, distribution labels with more channels than classes) never indexes pred / class_weights: the pixel is skipped
// in the gradient and the LOSS becomes NaN -- torch.nn.CrossEntropyLoss raises a device assert for the same input; a
// silent finite loss would hide the bad label (tests/test_train_terrain_gpu.py::test_bev_ce_rejects_out_of_range_labels)
template <int MODE>
__global__ __launch_bounds__(256) void bev_ce_kernel(const float* __restrict__ pred, int cs, int C,
                                                     const float* __restrict__ gt, int Cg, long HW, long P,
                                                     const uint8_t* __restrict__ fov, const float* __restrict__ cw,
                                                     int class_dim, int ignore_index, float eps,
                                                     const float* __restrict__ out4, float gscale,
                                                     float* __restrict__ g, int g_cs, float* __restrict__ partial) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const float inv_w = MODE == 1 ? gscale / out4[2] : 0.f;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    const bool in = fov[p] != 0;
    const float* x = pred + p * cs;
    int y = 0;
    bool use = false;
    float mx = -3.0e38f, se = 0.f;
    int am = 0;
    if (in) {
      y = bce_label(gt, Cg, HW, p, class_dim, eps);
      use = y != ignore_index;
      if (use && (y < 0 || y >= C)) {
        use = false;
        if (MODE == 0) a0 = __int_as_float(0x7fc00000);
      }
      for (int c = 0; c < C; ++c) { const float v = x[c]; if (v > mx) { mx = v; am = c; } }
      for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
    }
    if (MODE == 0) {
      if (in) {
        if (use) {
          const float w = cw ? cw[y] : 1.f;
          a0 += w * (mx + logf(se) - x[y]);
          a1 += w;
        }
        if (y != 0) { a3 += 1.f; a2 += am == y ? 1.f : 0.f; }            // metric: class 0 taken as unlabelled
      }
    } else {
      float* go = g + p * g_cs;
      if (in && use) {
        const float w = (cw ? cw[y] : 1.f) * inv_w;
        const float inv = 1.f / se;
        for (int c = 0; c < C; ++c) go[c] = w * (expf(x[c] - mx) * inv - (c == y ? 1.f : 0.f));
      } else {
        for (int c = 0; c < C; ++c) go[c] = 0.f;
      }
    }
  }
  if (MODE == 0) block_sum4(a0, a1, a2, a3, partial);
}

__global__ void bev_ce_final_kernel(const float* __restrict__ partial, int nb, float eps, float* __restrict__ out4) {
  if (threadIdx.x == 0) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < nb; ++b)
      for (int k = 0; k < 4; ++k) s[k] += partial[(size_t)b * 4 + k];
    out4[0] = s[0] / s[1];                 // weighted mean (torch.nn.CrossEntropyLoss, reduction='mean')
    out4[1] = s[2] / (s[3] + eps);         // accuracy over labelled pixels (loss_utils.py:470-472)
    out4[2] = s[1];
    out4[3] = s[3];
  }
}

// smooth-L1 (Huber with threshold beta): MODE 0 partial (sum, count), MODE 1 gradient.
// KIND 0: elevation (loss_utils.py:576-603): pred NHWC [P][2] (pixel stride cs), gt NCHW [B][2][HW]; channel 1 of the label
//         relative to channel 0 unless `absolute`; nan / inf labels masked per element.
// KIND 1: metric depth (:530-573): pred [P] metres, gt [P] millimetres; valid where the label falls into a bin.
template <int MODE, int KIND>
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ pred, int cs, const float* __restrict__ gt,
                                                        long HW, long P, int absolute, float beta, float dmin, float bin_size,
                                                        int num_bins, const float* __restrict__ out2, float gscale,
                                                        float* __restrict__ g, int g_cs, float* __restrict__ partial) {
  float a0 = 0.f, a1 = 0.f;
  const float inv_n = MODE == 1 ? gscale / out2[1] : 0.f;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    float d[2], t[2];
    bool ok[2] = {false, false};
    int nch;
    if (KIND == 0) {
      nch = 2;
      const long b = p / HW, q = p - b * HW;
      const float g0 = gt[(b * 2) * HW + q], g1 = gt[(b * 2 + 1) * HW + q];
      t[0] = g0;
      t[1] = absolute ? g1 : __fsub_rn(g1, g0);
      for (int c = 0; c < 2; ++c) { ok[c] = isfinite(t[c]); d[c] = pred[p * cs + c] - t[c]; }
    } else {
      nch = 1;
      const float gm = gt[p];
      const float idx = __fdiv_rn(__fsub_rn(gm, dmin), bin_size);
      ok[0] = isfinite(idx) && idx >= 0.f && idx < (float)num_bins;        // the label's bin exists (depth_utils.bin_depths)
      t[0] = __fdiv_rn(gm, 1000.f);
      d[0] = pred[p] - t[0];
    }
    for (int c = 0; c < nch; ++c) {
      const float ad = fabsf(d[c]);
      if (MODE == 0) {
        if (ok[c]) { a0 += ad < beta ? 0.5f * d[c] * d[c] / beta : ad - 0.5f * beta; a1 += 1.f; }
      } else {
        const float gv = ok[c] ? (ad < beta ? d[c] / beta : (d[c] > 0.f ? 1.f : -1.f)) * inv_n : 0.f;
        if (KIND == 0) g[p * g_cs + c] = gv; else g[p] = gv;
      }
    }
  }
  if (MODE == 0) block_sum4(a0, a1, 0.f, 0.f, partial);
}

__global__ void smooth_l1_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out2) {
  if (threadIdx.x == 0) {
    float s = 0.f, n = 0.f;
    for (int b = 0; b < nb; ++b) { s += partial[(size_t)b * 4]; n += partial[(size_t)b * 4 + 1]; }
    out2[0] = s / n;
    out2[1] = n;
  }
}

}  // namespace creste

using namespace creste;

static inline int loss_grid(long P) { long b = (P + 255) / 256; return (int)(b < 1 ? 1 : (b > 512 ? 512 : b)); }

extern "C" int creste_bev_ce_loss_f32(const float* pred, int cs, int C, const float* gt, int Cg, int64_t HW, int64_t P,
                                      const uint8_t* fov, const float* class_weights, int class_dim, int ignore_index,
                                      float eps, float grad_scale, float* g_pred, int g_cs, float* out4, void* work,
                                      void* stream) {
  CRESTE_REQUIRE(pred && gt && fov && g_pred && out4 && work, "bev_ce_loss: null pointer");
  CRESTE_REQUIRE(P > 0 && HW > 0 && P % HW == 0 && C > 0 && C <= BCE_MAXC && cs >= C && g_cs >= C && Cg > 0 && class_dim < Cg,
                 "bev_ce_loss: bad dims");
  hipStream_t s = (hipStream_t)stream;
  const int nb = loss_grid(P);
  bev_ce_kernel<0><<<nb, 256, 0, s>>>(pred, cs, C, gt, Cg, HW, P, fov, class_weights, class_dim, ignore_index, eps, nullptr,
                                       0.f, nullptr, 0, (float*)work);
  bev_ce_final_kernel<<<1, 64, 0, s>>>((const float*)work, nb, eps, out4);
  bev_ce_kernel<1><<<nb, 256, 0, s>>>(pred, cs, C, gt, Cg, HW, P, fov, class_weights, class_dim, ignore_index, eps, out4,
                                       grad_scale, g_pred, g_cs, nullptr);
  CRESTE_CHECK_LAUNCH("bev_ce_loss");
  return CRESTE_OK;
}

extern "C" int creste_smooth_l1_loss_f32(int kind, const float* pred, int cs, const float* gt, int64_t HW, int64_t P,
                                         int absolute, float beta, float depth_min, float depth_max, int num_bins,
                                         float grad_scale, float* g_pred, int g_cs, float* out2, void* work, void* stream) {
  CRESTE_REQUIRE(pred && gt && g_pred && out2 && work && P > 0 && beta > 0.f, "smooth_l1_loss: bad args");
  CRESTE_REQUIRE(kind == 0 || kind == 1, "smooth_l1_loss: kind 0 (elevation) or 1 (metric depth)");
  CRESTE_REQUIRE(kind == 1 || (HW > 0 && P % HW == 0 && cs >= 2 && g_cs >= 2), "smooth_l1_loss: bad elevation dims");
  hipStream_t s = (hipStream_t)stream;
  const int nb = loss_grid(P);
  const float bin = num_bins > 0 ? (depth_max - depth_min) / (float)num_bins : 1.f;
  float* wk = (float*)work;
  if (kind == 0) {
    smooth_l1_kernel<0, 0><<<nb, 256, 0, s>>>(pred, cs, gt, HW, P, absolute, beta, 0.f, 1.f, 0, nullptr, 0.f, nullptr, 0, wk);
    smooth_l1_final_kernel<<<1, 64, 0, s>>>(wk, nb, out2);
    smooth_l1_kernel<1, 0><<<nb, 256, 0, s>>>(pred, cs, gt, HW, P, absolute, beta, 0.f, 1.f, 0, out2, grad_scale, g_pred, g_cs, nullptr);
  } else {
    smooth_l1_kernel<0, 1><<<nb, 256, 0, s>>>(pred, 1, gt, 1, P, 0, beta, depth_min, bin, num_bins, nullptr, 0.f, nullptr, 0, wk);
    smooth_l1_final_kernel<<<1, 64, 0, s>>>(wk, nb, out2);
    smooth_l1_kernel<1, 1><<<nb, 256, 0, s>>>(pred, 1, gt, 1, P, 0, beta, depth_min, bin, num_bins, out2, grad_scale, g_pred, 1, nullptr);
  }
  CRESTE_CHECK_LAUNCH("smooth_l1_loss");
  return CRESTE_OK;
}
