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
      if (z > mx) { se = se * expf(mx - z) + 1.f; mx = z; }
      else se += expf(z - mx);
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
      if (p.x > mx) { se = se * expf(mx - p.x) + p.y; mx = p.x; }
      else se += p.y * expf(p.x - mx);
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
      float gij = expf(z - s.x) * s.y;
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

using namespace creste;

static inline int mpc_splits(int rows) {
  const int rb = (rows + 255) / 256;
  int n = (2048 + rb - 1) / rb;
  return n < 1 ? 1 : (n > 16 ? 16 : n);
}

// workspace: stats [N] | split stats [16][N] | loss partials [4096] | split gradients [16][max(N,M)][D]
extern "C" int64_t creste_multipos_con_workspace_bytes(int N, int M, int D) {
  if (N <= 0 || M <= 0 || D <= 0) return -1;
  const int64_t big = N > M ? N : M;
  return (int64_t)N * 16 * 17 + 4096 * 4 + 16 * big * D * 4;
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
#define CALL(DD) mpc_rows_kernel<DD><<<dim3(nb, ns), 256, 0, s>>>(feats, all_feats, labels, all_labels, N, M, self_offset, 1.f / temperature, part, per)
  CRESTE_MPC_DISPATCH(D, CALL)
#undef CALL
  CRESTE_CHECK_LAUNCH("mpc_rows");
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
