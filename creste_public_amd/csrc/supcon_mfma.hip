// Multi-positive contrastive loss on the matrix cores (reference creste/models/losses/supcon_loss.py:56-115; the
// definition and the streaming formulation are in losses.hip).  The N x M logits z = F A^T / T and the two gradient
// products G A and G^T (w F) are GEMMs: here each 32 x 32 block of pairs is ONE accumulator tile of
// v_mfma_f32_32x32x16_f16, the softmax / positive-mask arithmetic runs on the tile in registers, and the tile of
// gradients G goes straight back into the matrix cores as the next product's operand.
//
// Operands are fp32-grade: every feature vector is rescaled by an exact power of two (from a device |max|, as the
// conv kernels do) into fp16 range and split into fp16 hi + lo (22 significand bits); a product is hi*hi + hi*lo +
// lo*hi with fp32 accumulation (error <= 2^-21 relative).  A pack pass writes, per side, the split rows [item][D]
// (contraction over D: the logits) and a transposed image [D][item] permuted to the accumulator's register order
// (contraction over items: the gradients).
//
// Register layout that makes this free of shuffles: the OTHER side's tile is the first MFMA operand and the wave's 32
// OWN items the second, so in D = Y X^T lane (li, lh) owns item li and its 16 accumulator registers are the pairs with
// other items (r&3) + 8*(r>>2) + 4*lh.  Row reductions (max, sum-exp, positive count / sum) are then per-lane loops over
// registers plus one cross-half shuffle at the very end; and the gradient tile G[own][other], converted to fp16 in
// register order, IS the first operand (rows = own, k = other in the permuted order pi) of the second product, whose
// other operand is read from the pi-permuted transposed image.  No LDS, no barriers: the four waves of a workgroup are
// independent.
#include "common.h"

namespace creste {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.44269504088896341f, kLn2 = 0.693147180559945309f;
constexpr float kGScale = 4096.f, kGInv = 1.f / 4096.f;       // gradients |G| <= 1 enter the matrix cores as G * 2^12

struct MpcPacked {                 // device pointers into the workspace
  const _Float16 *f_hi, *f_lo, *a_hi, *a_lo;                  // [rows32][D]
  const _Float16 *aT_hi, *aT_lo, *fwT_hi, *fwT_lo;            // [tiles][D][32] in pi order
  const int64_t *lab_f, *lab_a;                               // padded to a multiple of 32
  const float* amax;                                          // [3]: max|f|, max|a|, max|w f|
  const float4* rstat;                                        // [N]: (max * log2 e, [cnt>0]/sumexp, [cnt>0]/cnt, 0)
};

__device__ __forceinline__ int mpc_jmap(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// max |x| (and max |w x| when w is given) of a [rows][D] matrix
__global__ __launch_bounds__(256) void mpc_amax_kernel(const float* __restrict__ x, const float* __restrict__ w, long rows,
                                                       int D, float* __restrict__ amax_x, float* __restrict__ amax_wx) {
  __shared__ float scratch[4];
  float m = 0.f, mw = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < rows * D; i += (long)gridDim.x * 256) {
    const float v = fabsf(x[i]);
    m = fmaxf(m, v);
    if (amax_wx) mw = fmaxf(mw, v * fabsf(w ? w[i / D] : 1.f));
  }
  block_amax_update(m, amax_x, scratch);
  if (amax_wx) {
    __syncthreads();
    block_amax_update(mw, amax_wx, scratch);
  }
}

// split rows + the permuted transposed image of one side; rows32 = rows rounded up to 32 (padding: zeros)
__global__ __launch_bounds__(256) void mpc_pack_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const int64_t* __restrict__ lab, long rows, long rows32, int D,
                                                       const float* __restrict__ amax_row, const float* __restrict__ amax_t,
                                                       _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                       _Float16* __restrict__ t_hi, _Float16* __restrict__ t_lo,
                                                       int64_t* __restrict__ lab_out) {
  float inv;
  const float s_row = f16_operand_scale(*amax_row, &inv), s_t = f16_operand_scale(*amax_t, &inv);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < rows32 * D; i += (long)gridDim.x * 256) {
    const long row = i / D;
    const int d = (int)(i - row * D);
    const float v = row < rows ? x[i] : 0.f;
    const float a = v * s_row;
    const _Float16 ah = (_Float16)a;
    hi[i] = ah;
    lo[i] = (_Float16)(a - (float)ah);
    const float b = v * (w && row < rows ? w[row] : 1.f) * s_t;
    const _Float16 bh = (_Float16)b;
    const int jj = (int)(row & 31);
    const int slot = ((jj >> 4) * 2 + ((jj >> 2) & 1)) * 8 + ((jj & 3) | (((jj >> 3) & 1) << 2));
    const long o = ((row >> 5) * D + d) * 32 + slot;
    t_hi[o] = bh;
    t_lo[o] = (_Float16)(b - (float)bh);
    if (d == 0) lab_out[row] = row < rows ? lab[row] : (int64_t)0x8000000000000000LL;
  }
}

__global__ void mpc_rstat_kernel(const float4* __restrict__ stats, int N, float4* __restrict__ rstat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float4 s = stats[i];
  const bool on = s.z > 0.f;
  rstat[i] = make_float4(s.x * kLog2e, on ? 1.f / s.y : 0.f, on ? 1.f / s.z : 0.f, 0.f);
}

__device__ __forceinline__ f16v mfma(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// z tile of 32 other items x the wave's 32 own items (smallest products first)
template <int KS>
__device__ __forceinline__ f16v mpc_ztile(const _Float16* __restrict__ y_hi, const _Float16* __restrict__ y_lo, long y0, int D,
                                          int li, int lh, const h8 (&xh)[KS], const h8 (&xl)[KS]) {
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const long o = (y0 + li) * D + 16 * ks + 8 * lh;
    const h8 yh = *reinterpret_cast<const h8*>(y_hi + o), yl = *reinterpret_cast<const h8*>(y_lo + o);
    acc = mfma(yl, xh[ks], acc);
    acc = mfma(yh, xl[ks], acc);
    acc = mfma(yh, xh[ks], acc);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------- forward statistics
// part[split][i] = (max, sumexp, count, possum) of row i over the split's column range
template <int D>
__global__ __launch_bounds__(256) void mpc_rows_mfma_kernel(MpcPacked p, int N, int M, int self_off, float inv_t,
                                                            float4* __restrict__ part, int per_split) {
  constexpr int KS = D / 16;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int ow0 = blockIdx.x * 128 + (threadIdx.x >> 6) * 32;
  if (ow0 >= N) return;
  const int i = ow0 + li;
  float inv_f, inv_a;
  f16_operand_scale(p.amax[0], &inv_f);
  f16_operand_scale(p.amax[1], &inv_a);
  const float zc = inv_f * inv_a * inv_t, zcl = zc * kLog2e;
  h8 xh[KS], xl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const long o = (long)i * D + 16 * ks + 8 * lh;
    xh[ks] = *reinterpret_cast<const h8*>(p.f_hi + o);
    xl[ks] = *reinterpret_cast<const h8*>(p.f_lo + o);
  }
  const int64_t mylab = p.lab_f[i];
  const int self_j = i + self_off;
  const int c_lo = blockIdx.y * per_split, c_hi = min(M, c_lo + per_split);
  float mx2 = -3.0e38f, se = 0.f, cnt = 0.f, psa = 0.f;
  for (int y0 = c_lo; y0 < c_hi; y0 += 32) {
    const f16v acc = mpc_ztile<KS>(p.a_hi, p.a_lo, y0, D, li, lh, xh, xl);
    const bool edge = y0 + 32 > c_hi || (y0 < ow0 + self_off + 32 && y0 + 32 > ow0 + self_off);   // wave-uniform
    float t[16];
    float tmax = -3.0e38f;
    unsigned ok = 0xffffu;
    if (edge) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = y0 + mpc_jmap(r, lh);
        if (j >= c_hi || j == self_j) ok &= ~(1u << r);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      t[r] = (ok >> r & 1) ? acc[r] * zcl : -INFINITY;
      tmax = fmaxf(tmax, t[r]);
    }
    const float m_new = fmaxf(mx2, tmax);
    se *= __builtin_amdgcn_exp2f(mx2 - m_new);
#pragma unroll
    for (int r = 0; r < 16; ++r) se += __builtin_amdgcn_exp2f(t[r] - m_new);
    mx2 = m_new;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t* lp = p.lab_a + y0 + 4 * lh + 8 * g;
      const longlong2 l01 = *reinterpret_cast<const longlong2*>(lp), l23 = *reinterpret_cast<const longlong2*>(lp + 2);
      const int64_t l4[4] = {l01.x, l01.y, l23.x, l23.y};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const bool pos = l4[e] == mylab && (ok >> r & 1);
        cnt += pos ? 1.f : 0.f;
        psa += pos ? acc[r] : 0.f;
      }
    }
  }
  // the two halves of the wave hold the same rows over disjoint column subsets
  const float mx_o = __shfl_xor(mx2, 32, 64), se_o = __shfl_xor(se, 32, 64);
  cnt += __shfl_xor(cnt, 32, 64);
  psa += __shfl_xor(psa, 32, 64);
  const float m = fmaxf(mx2, mx_o);
  se = se * __builtin_amdgcn_exp2f(mx2 - m) + se_o * __builtin_amdgcn_exp2f(mx_o - m);
  if (lh == 0 && i < N) part[(size_t)blockIdx.y * N + i] = make_float4(m * kLn2, se, cnt, psa * zc);
}

// ------------------------------------------------------------------------------------------------------ gradients
// ROWS = true : own = local rows i (grad_f), other = gathered columns;  ROWS = false: own = columns j (grad_a), other =
// rows (whose weights w_i are folded into the transposed image fwT).  gpart[split][own][D].
template <int D, bool ROWS>
__global__ __launch_bounds__(256) void mpc_grad_mfma_kernel(MpcPacked p, const float* __restrict__ rw, int N, int M,
                                                            int self_off, float inv_t, float gscale,
                                                            float* __restrict__ gpart, int per_split) {
  constexpr int KS = D / 16, NT = D > 32 ? D / 32 : 1;
  const int own_n = ROWS ? N : M, oth_n = ROWS ? M : N;
  const _Float16 *x_hi = ROWS ? p.f_hi : p.a_hi, *x_lo = ROWS ? p.f_lo : p.a_lo;
  const _Float16 *y_hi = ROWS ? p.a_hi : p.f_hi, *y_lo = ROWS ? p.a_lo : p.f_lo;
  const _Float16 *yt_hi = ROWS ? p.aT_hi : p.fwT_hi, *yt_lo = ROWS ? p.aT_lo : p.fwT_lo;
  const int64_t *own_lab = ROWS ? p.lab_f : p.lab_a, *oth_lab = ROWS ? p.lab_a : p.lab_f;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int ow0 = blockIdx.x * 128 + (threadIdx.x >> 6) * 32;
  if (ow0 >= own_n) return;
  const int own = ow0 + li;
  float inv_f, inv_a, inv_w;
  f16_operand_scale(p.amax[0], &inv_f);
  f16_operand_scale(p.amax[1], &inv_a);
  f16_operand_scale(p.amax[2], &inv_w);
  const float zcl = inv_f * inv_a * inv_t * kLog2e;
  h8 xh[KS], xl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const long o = (long)own * D + 16 * ks + 8 * lh;
    xh[ks] = *reinterpret_cast<const h8*>(x_hi + o);
    xl[ks] = *reinterpret_cast<const h8*>(x_lo + o);
  }
  const int64_t mylab = own_lab[own];
  float4 st = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ROWS && own < N) st = p.rstat[own];
  // the self pair of own item `own`: column own + self_off (ROWS) / row own - self_off (columns)
  const int self_o = ROWS ? own + self_off : own - self_off;
  const int self_lo = ROWS ? ow0 + self_off : ow0 - self_off;
  f16v out[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[nt][r] = 0.f;
  const int o_lo = blockIdx.y * per_split, o_hi = min(oth_n, o_lo + per_split);
  for (int y0 = o_lo; y0 < o_hi; y0 += 32) {
    const f16v acc = mpc_ztile<KS>(y_hi, y_lo, y0, D, li, lh, xh, xl);
    const bool edge = y0 + 32 > o_hi || (y0 < self_lo + 32 && y0 + 32 > self_lo);
    float g[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t* lp = oth_lab + y0 + 4 * lh + 8 * q;
      const longlong2 l01 = *reinterpret_cast<const longlong2*>(lp), l23 = *reinterpret_cast<const longlong2*>(lp + 2);
      const int64_t l4[4] = {l01.x, l01.y, l23.x, l23.y};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e, k = y0 + mpc_jmap(r, lh);
        float4 s = st;
        if (!ROWS) s = p.rstat[k < N ? k : N - 1];            // per-row statistics of the other side's row k
        float v = __builtin_amdgcn_exp2f(fmaf(acc[r], zcl, -s.x)) * s.y;
        v -= l4[e] == mylab ? s.z : 0.f;
        if (edge && (k >= o_hi || k == self_o)) v = 0.f;
        g[r] = v * kGScale;
      }
    }
    // G as the first operand of the gradient product: rows = own items (this lane), k = other items in pi order
    h8 gh[2], gl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)g[8 * ks + e];
        gh[ks][e] = h;
        gl[ks][e] = (_Float16)(g[8 * ks + e] - (float)h);
      }
    const long tbase = (long)(y0 >> 5) * D;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int d = nt * 32 + (D >= 32 ? li : (li & (D - 1)));
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const long o = (tbase + d) * 32 + (2 * ks + lh) * 8;
        const h8 th = *reinterpret_cast<const h8*>(yt_hi + o), tl = *reinterpret_cast<const h8*>(yt_lo + o);
        out[nt] = mfma(gl[ks], th, out[nt]);
        out[nt] = mfma(gh[ks], tl, out[nt]);
        out[nt] = mfma(gh[ks], th, out[nt]);
      }
    }
  }
  // out[nt][r]: lane li = feature d, register r = own item ow0 + jmap(r, lh)
  const float fin = gscale * inv_t * kGInv * (ROWS ? inv_a : inv_w);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = ow0 + mpc_jmap(r, lh);
    if (m >= own_n) continue;
    const float sc = ROWS ? fin * (rw ? rw[m] : 1.f) : fin;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int d = nt * 32 + li;
      if (d < D) gpart[((size_t)blockIdx.y * own_n + m) * D + d] = out[nt][r] * sc;
    }
  }
}

static inline long r32(long n) { return (n + 31) / 32 * 32; }

int64_t mpc_mfma_workspace_bytes(int N, int M, int D) {
  const long n32 = r32(N), m32 = r32(M);
  return 64 + 16L * N + 8L * (n32 + m32) + 8L * D * (n32 + m32);
}

static MpcPacked mpc_layout(void* base, int N, int M, int D) {
  const long n32 = r32(N), m32 = r32(M);
  char* c = (char*)base;
  MpcPacked p;
  p.amax = (const float*)c; c += 64;
  p.rstat = (const float4*)c; c += 16L * N;
  p.lab_f = (const int64_t*)c; c += 8L * n32;
  p.lab_a = (const int64_t*)c; c += 8L * m32;
  auto take = [&](long rows) { const _Float16* q = (const _Float16*)c; c += 2L * rows * D; return q; };
  p.f_hi = take(n32); p.f_lo = take(n32); p.fwT_hi = take(n32); p.fwT_lo = take(n32);
  p.a_hi = take(m32); p.a_lo = take(m32); p.aT_hi = take(m32); p.aT_lo = take(m32);
  return p;
}

bool mpc_mfma_supported(int D) { return D == 16 || D == 32 || D == 64; }

// forward: |max| + pack + row statistics -> part[ns][N] (merged by the caller's mpc_merge_kernel)
int mpc_mfma_rows(const float* f, const float* a, const int64_t* lab_f, const int64_t* lab_a, const float* rw, int N, int M,
                  int D, int self_off, float inv_t, float4* part, int ns, int per_split, void* work, hipStream_t s) {
  MpcPacked p = mpc_layout(work, N, M, D);
  float* amax = const_cast<float*>(p.amax);
  CRESTE_HIP(hipMemsetAsync(amax, 0, 64, s));
  mpc_amax_kernel<<<256, 256, 0, s>>>(f, rw, N, D, amax + 0, amax + 2);
  mpc_amax_kernel<<<256, 256, 0, s>>>(a, nullptr, M, D, amax + 1, nullptr);
  const long n32 = r32(N), m32 = r32(M);
  mpc_pack_kernel<<<1024, 256, 0, s>>>(f, rw, lab_f, N, n32, D, amax + 0, amax + 2, const_cast<_Float16*>(p.f_hi),
                                       const_cast<_Float16*>(p.f_lo), const_cast<_Float16*>(p.fwT_hi),
                                       const_cast<_Float16*>(p.fwT_lo), const_cast<int64_t*>(p.lab_f));
  mpc_pack_kernel<<<1024, 256, 0, s>>>(a, nullptr, lab_a, M, m32, D, amax + 1, amax + 1, const_cast<_Float16*>(p.a_hi),
                                       const_cast<_Float16*>(p.a_lo), const_cast<_Float16*>(p.aT_hi),
                                       const_cast<_Float16*>(p.aT_lo), const_cast<int64_t*>(p.lab_a));
  const dim3 grid((N + 127) / 128, ns);
  if (D == 16) mpc_rows_mfma_kernel<16><<<grid, 256, 0, s>>>(p, N, M, self_off, inv_t, part, per_split);
  else if (D == 32) mpc_rows_mfma_kernel<32><<<grid, 256, 0, s>>>(p, N, M, self_off, inv_t, part, per_split);
  else mpc_rows_mfma_kernel<64><<<grid, 256, 0, s>>>(p, N, M, self_off, inv_t, part, per_split);
  CRESTE_CHECK_LAUNCH("mpc_rows_mfma");
  return CRESTE_OK;
}

// backward, one side: gpart[ns][own][D] (summed by the caller's mpc_sum_splits_kernel); stats = merged forward stats
int mpc_mfma_grad(bool rows, const float4* stats, const float* rw, int N, int M, int D, int self_off, float inv_t,
                  float gscale, float* gpart, int ns, int per_split, void* work, hipStream_t s) {
  MpcPacked p = mpc_layout(work, N, M, D);
  if (rows) mpc_rstat_kernel<<<(N + 255) / 256, 256, 0, s>>>(stats, N, const_cast<float4*>(p.rstat));
  const int own_n = rows ? N : M;
  const dim3 grid((own_n + 127) / 128, ns);
#define CRESTE_MPC_G(DD)                                                                                         \
  if (rows) mpc_grad_mfma_kernel<DD, true><<<grid, 256, 0, s>>>(p, rw, N, M, self_off, inv_t, gscale, gpart, per_split); \
  else mpc_grad_mfma_kernel<DD, false><<<grid, 256, 0, s>>>(p, rw, N, M, self_off, inv_t, gscale, gpart, per_split)
  if (D == 16) { CRESTE_MPC_G(16); }
  else if (D == 32) { CRESTE_MPC_G(32); }
  else { CRESTE_MPC_G(64); }
#undef CRESTE_MPC_G
  CRESTE_CHECK_LAUNCH("mpc_grad_mfma");
  return CRESTE_OK;
}

}  // namespace creste
