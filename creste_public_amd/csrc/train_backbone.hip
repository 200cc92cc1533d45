// Backward primitives of the RGB-D encoder (stage-1 distillation, reference train_pefree.py:71-99 through
// creste/models/distillation.py:145-207, blocks/effnet.py, efficientnet_pytorch MBConv blocks) and its
// training objectives (reference loss_utils.py: CrossEntropyDepth :477-527, MSELoss :606-647).
//   * general conv weight gradient (stride, asymmetric static padding, any K) on the fp32 MFMA;
//   * depthwise conv: input gradient and per-tap weight gradient;
//   * swish, per-sample scaling (squeeze-excite gate, drop-connect), per-sample channel reductions;
//   * the squeeze-excite bottleneck (two tiny FCs) forward / backward;
//   * fused depth-classification loss (bin the metric label, softmax cross-entropy over the valid pixels,
//     accuracy, gradient w.r.t. the logits) and the masked MSE of the distilled features.
// NHWC fp32 with explicit pixel strides; fixed-order reductions (block partials + ordered finalisation).
#include "common.h"
#include <stdlib.h>

namespace creste {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

static inline int grid1d(long work, int cap = 8192) {
  long b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------------------------ general conv wgrad
// partial[chunk][tap][co][ci] = sum over the chunk's OUTPUT pixels q of gy[q][co] * x[q*stride + tap - pad][ci]
constexpr int WGS_PIX = 16;
__global__ __launch_bounds__(256) void wgrad_strided_partial_kernel(
    const float* __restrict__ x, int x_cs, const float* __restrict__ gy, int gy_cs, float* __restrict__ partial,
    int N, int H, int W, int Ho, int Wo, int Cin, int Cout, int K, int stride, int pad_t, int pad_l, int chunk_px) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int chunk = blockIdx.x, tap = blockIdx.y;
  const int ky = tap / K - pad_t, kx = tap % K - pad_l;
  const int tco = (Cout + 31) / 32, tci = (Cin + 31) / 32;
  const int M = N * Ho * Wo;
  const int p0 = chunk * chunk_px, p1 = min(M, p0 + chunk_px);
  for (int t = wave; t < tco * tci; t += 4) {
    const int co = (t / tci) * 32 + li, ci = (t % tci) * 32 + li;
    const bool co_ok = co < Cout, ci_ok = ci < Cin;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int p = p0; p < p1; p += WGS_PIX) {
      float a[WGS_PIX / 2], b[WGS_PIX / 2];
#pragma unroll
      for (int j = 0; j < WGS_PIX / 2; ++j) {
        const int q = p + 2 * j + lh;
        a[j] = 0.f; b[j] = 0.f;
        if (q < p1) {
          const int rowi = q / Wo;
          const int ox = q - rowi * Wo;
          const int n = rowi / Ho;
          const int oy = rowi - n * Ho;
          const int iy = oy * stride + ky, ix = ox * stride + kx;
          if (co_ok) a[j] = gy[(long)q * gy_cs + co];
          if (ci_ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            b[j] = x[(((long)n * H + iy) * W + ix) * x_cs + ci];
        }
      }
#pragma unroll
      for (int j = 0; j < WGS_PIX / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
    }
    float* out = partial + ((size_t)chunk * K * K + tap) * Cout * Cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (t / tci) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < Cout && ci_ok) out[(size_t)row * Cin + ci] = acc[r];
    }
  }
}

// convs with a handful of input channels (the 4-channel RGB-D stem, 3x3/2): the K*K*Cin patch elements (36) play the
// input-channel role of ONE matrix product per pixel quad -- v_mfma_f32_16x16x4_f32 with A = gy (co x 4 pixels) and
// B = the im2col patch gathered straight from x (patch element x 4 pixels), 2 x 3 tiles for 32 x 48.  The per-tap kernel
// above filled 4 of 32 operand lanes and read gy nine times (1.1 ms per step); a VALU outer product was tried and is
// bound by the L1 (one FMA per loaded dword): here a wave issues 5 loads for 6 MFMAs per 4 pixels.
typedef float f32x4w __attribute__((ext_vector_type(4)));
constexpr int WSC_TCO = 2, WSC_TPE = 3;
__global__ __launch_bounds__(256) void wgrad_smallcin_kernel(
    const float* __restrict__ x, int x_cs, const float* __restrict__ gy, int gy_cs, float* __restrict__ partial,
    int N, int H, int W, int Ho, int Wo, int Cin, int Cout, int K, int stride, int pad_t, int pad_l, int wave_px) {
  __shared__ float red[4][WSC_TCO * 16][WSC_TPE * 16 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lj = lane >> 4;
  const int PE = K * K * Cin, M = N * Ho * Wo;
  // this lane's patch elements (one per pe tile): offsets of the tap relative to the window origin
  int eky[WSC_TPE], ekx[WSC_TPE], eoff[WSC_TPE];
  bool ev[WSC_TPE];
#pragma unroll
  for (int t = 0; t < WSC_TPE; ++t) {
    const int pe = t * 16 + lr;
    ev[t] = pe < PE;
    const int tap = pe / Cin, ci = pe - tap * Cin;
    eky[t] = tap / K - pad_t; ekx[t] = tap % K - pad_l;
    eoff[t] = (eky[t] * W + ekx[t]) * x_cs + ci;
  }
  int coc[WSC_TCO];
  bool cv[WSC_TCO];
#pragma unroll
  for (int c = 0; c < WSC_TCO; ++c) { cv[c] = c * 16 + lr < Cout; coc[c] = cv[c] ? c * 16 + lr : 0; }
  f32x4w acc[WSC_TCO][WSC_TPE];
#pragma unroll
  for (int c = 0; c < WSC_TCO; ++c)
#pragma unroll
    for (int t = 0; t < WSC_TPE; ++t) acc[c][t] = f32x4w{0.f, 0.f, 0.f, 0.f};
  const long p0l = ((long)blockIdx.x * 4 + wave) * wave_px;
  const int p0 = (int)min((long)M, p0l), p1 = (int)min((long)M, p0l + wave_px);
  const int nsteps = (p1 - p0 + 3) >> 2;
  // branch-free loads: invalid lanes read element 0 and are zeroed (a predicated load compiles to a branch + wait)
  for (int st = 0; st < nsteps; ++st) {
    const int q = p0 + 4 * st + lj;
    const bool qv = q < p1;
    const int qq = qv ? q : 0;
    const int rowi = qq / Wo, ox = qq - rowi * Wo, n = rowi / Ho, oy = rowi - n * Ho;
    const int iy0 = oy * stride, ix0 = ox * stride;
    const long base = (((long)n * H + iy0) * W + ix0) * x_cs;
    float a[WSC_TCO], b[WSC_TPE];
#pragma unroll
    for (int c = 0; c < WSC_TCO; ++c) {
      const float v = gy[(long)qq * gy_cs + coc[c]];
      a[c] = (qv && cv[c]) ? v : 0.f;
    }
#pragma unroll
    for (int t = 0; t < WSC_TPE; ++t) {
      const bool ok = qv && ev[t] && (unsigned)(iy0 + eky[t]) < (unsigned)H && (unsigned)(ix0 + ekx[t]) < (unsigned)W;
      const float v = x[ok ? base + eoff[t] : 0];
      b[t] = ok ? v : 0.f;
    }
#pragma unroll
    for (int c = 0; c < WSC_TCO; ++c)
#pragma unroll
      for (int t = 0; t < WSC_TPE; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[t], acc[c][t], 0, 0, 0);
  }
  // D layout: col (pe) = lane & 15, row (co) = 4 * (lane >> 4) + r; the four waves' tiles are summed in a fixed order
#pragma unroll
  for (int c = 0; c < WSC_TCO; ++c)
#pragma unroll
    for (int t = 0; t < WSC_TPE; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][c * 16 + 4 * lj + r][t * 16 + lr] = acc[c][t][r];
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * K * K * Cout * Cin;
  for (int e = threadIdx.x; e < Cout * PE; e += 256) {
    const int co = e / PE, pe = e - co * PE;
    const float v = (red[0][co][pe] + red[1][co][pe]) + (red[2][co][pe] + red[3][co][pe]);
    const int tap = pe / Cin, ci = pe - tap * Cin;
    out[((size_t)tap * Cout + co) * Cin + ci] = v;
  }
}

static inline bool wgrad_smallcin_ok(int Cin, int Cout, int K) {
  return Cin < 8 && Cout <= WSC_TCO * 16 && K * K * Cin <= WSC_TPE * 16;
}
static inline int wgrad_smallcin_chunks(long M) {
  const long cap = (M + 1023) / 1024;                 // at least 256 pixels per wave
  return (int)(cap < 512 ? (cap < 1 ? 1 : cap) : 512);
}

// LDS-tiled variant for the wide layers: a workgroup (2 x 2 waves) owns a (64*WM) x (64*WN) block of one tap's
// [Cout x Cin] gradient and walks its pixel chunk in steps of 16; per step the gy rows (A^T, 16 x BM) and the
// shifted x rows (B, 16 x BN) are staged ONCE in LDS (float4 global loads, zero fill outside the image) and feed
// 8 * WM * WN fp32 MFMAs per wave -- the direct kernel above re-reads both operands from L2 for every 32x32 tile
// (24 TFLOP/s on 496->496); register-staged double buffering, one barrier per step.
constexpr int WT_KS = 16;
template <int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_tiled_kernel(
    const float* __restrict__ x, int x_cs, const float* __restrict__ gy, int gy_cs, float* __restrict__ partial,
    int N, int H, int W, int Ho, int Wo, int Cin, int Cout, int K, int stride, int pad_t, int pad_l, int chunk_px,
    int tiles_co, int tiles_ci) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int A_F4 = WT_KS * BM / 4 / 256, B_F4 = WT_KS * BN / 4 / 256;     // float4 loads per thread per step
  static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small for 256 loader threads");
  __shared__ __attribute__((aligned(16))) float As[2][WT_KS][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][WT_KS][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  int id = blockIdx.x;
  const int tci = id % tiles_ci; id /= tiles_ci;
  const int tco = id % tiles_co;
  const int chunk = id / tiles_co;
  const int tap = blockIdx.y;
  const int ky = tap / K - pad_t, kx = tap % K - pad_l;
  const int co0 = tco * BM, ci0 = tci * BN;
  const int M = N * Ho * Wo;
  const int p0 = chunk * chunk_px, p1 = min(M, p0 + chunk_px);

  f32x16 c[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;

  f32x4 ra[A_F4], rb[B_F4];
  auto load = [&](int p) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A_F4; ++u) {
      const int e = tid + 256 * u;                   // float4 index inside the 16 x BM tile
      const int k = e / (BM / 4), cq = e % (BM / 4);
      const int q = p + k, co = co0 + cq * 4;
      ra[u] = (q < p1 && co < Cout) ? ld4(gy + (long)q * gy_cs + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < B_F4; ++u) {
      const int e = tid + 256 * u;
      const int k = e / (BN / 4), cq = e % (BN / 4);
      const int q = p + k, ci = ci0 + cq * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < p1 && ci < Cin) {
        const int rowi = q / Wo;
        const int ox = q - rowi * Wo;
        const int n = rowi / Ho;
        const int oy = rowi - n * Ho;
        const int iy = oy * stride + ky, ix = ox * stride + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = ld4(x + (((long)n * H + iy) * W + ix) * x_cs + ci);
      }
      rb[u] = v;
    }
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A_F4; ++u) {
      const int e = tid + 256 * u;
      st4(&As[buf][e / (BM / 4)][(e % (BM / 4)) * 4], ra[u]);
    }
#pragma unroll
    for (int u = 0; u < B_F4; ++u) {
      const int e = tid + 256 * u;
      st4(&Bs[buf][e / (BN / 4)][(e % (BN / 4)) * 4], rb[u]);
    }
  };

  load(p0);
  store(0);
  __syncthreads();
  int buf = 0;
  for (int p = p0; p < p1; p += WT_KS) {
    const bool more = p + WT_KS < p1;
    if (more) load(p + WT_KS);
#pragma unroll
    for (int kk = 0; kk < WT_KS; kk += 2) {
      float a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = As[buf][kk + lh][wm * 32 * WM + i * 32 + li];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = Bs[buf][kk + lh][wn * 32 * WN + j * 32 + li];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], c[i][j], 0, 0, 0);
    }
    if (more) store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  float* out = partial + ((size_t)chunk * K * K + tap) * Cout * Cin;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = ci0 + wn * 32 * WN + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = co0 + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < Cout && col < Cin) out[(size_t)row * Cin + col] = c[i][j][r];
      }
    }
}

// f16x3 variant (same operand scheme as the forward patch engine, csrc/conv_patch.hip): both operands are
// rescaled by exact powers of two from device-side |max| bounds, split into fp16 hi+lo while they are staged,
// and multiplied as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate) with fp32
// accumulation over the pixel chunk.  The MFMA's K runs over PIXELS, which are the strided dimension of the
// NHWC tensors: a loader thread owns one channel and reads its 8 consecutive pixels with 8 coalesced-across-
// lanes dword loads, so the transpose happens in registers and the LDS image is the conflict-free
// [piece][k-octet][channel][8 x fp16] layout of the forward engine (one 16-byte write, one ds_read_b128 fragment).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void wgrad_f16_kernel(
    const float* __restrict__ x, int x_cs, const float* __restrict__ gy, int gy_cs, float* __restrict__ partial,
    const float* __restrict__ x_amax, const float* __restrict__ gy_amax, int N, int H, int W, int Ho, int Wo, int Cin,
    int Cout, int K, int stride, int pad_t, int pad_l, int chunk_px, int tiles_co, int tiles_ci) {
  constexpr int BM = 128, KS = 32, NOCT = KS / 8;
  constexpr int OCT = BM * 16, PLANE = NOCT * OCT, TILE = 2 * PLANE;   // bytes: [piece][octet][row][8 fp16]
  __shared__ __attribute__((aligned(16))) char As[2][TILE];
  __shared__ __attribute__((aligned(16))) char Bs[2][TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // block order: tap fastest, then the channel tiles, then the pixel chunk, and contiguous ranges per XCD -- the
  // K*K*tiles workgroups that stream the SAME pixel range run together on ONE L2 (each operand element is needed
  // by K*K*tiles of them; with the tap in grid.y they were dispatched far apart and every one went to HBM)
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tap = id % (K * K); id /= K * K;
  const int tci = id % tiles_ci; id /= tiles_ci;
  const int tco = id % tiles_co;
  const int chunk = id / tiles_co;
  const int ky = tap / K - pad_t, kx = tap % K - pad_l;
  const int co0 = tco * BM, ci0 = tci * BM;
  const int M = N * Ho * Wo;
  const int p0 = chunk * chunk_px, p1 = min(M, p0 + chunk_px);
  float a_inv, b_inv;
  const float a_mul = f16_operand_scale(*gy_amax, &a_inv), b_mul = f16_operand_scale(*x_amax, &b_inv);

  // loader: threads 0..127 stage gy (A), 128..255 stage x (B); a thread owns one channel QUAD and one pixel octet:
  // 8 float4 loads (consecutive lanes = consecutive quads: coalesced rows), transposed in registers into four
  // 8-pixel fp16 rows
  const bool isB = tid >= 128;
  const int u = tid & 127, quad = u & 31, loct = u >> 5;
  const int ch = (isB ? ci0 : co0) + quad * 4;
  const bool ch_ok = ch < (isB ? Cin : Cout);
  // BRANCH-FREE loads: an out-of-range pixel reads the tensor's first element and is zeroed at conversion time (a
  // predicated load compiles to a branch with an s_waitcnt behind every load, which serialises the eight latencies)
  f32x4 rv[8];
  unsigned okmask = 0;
  auto load = [&](int p) __attribute__((always_inline)) {
    int q = p + loct * 8;
    int rowi = q / Wo;
    int ox = q - rowi * Wo;
    int n = rowi / Ho;
    int oy = rowi - n * Ho;
    okmask = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j, ++q) {
      const int iy = oy * stride + ky, ix = ox * stride + kx;
      const bool ok = q < p1 && ch_ok && (!isB || ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W));
      const float* ptr = isB ? x + (((long)n * H + iy) * W + ix) * x_cs + ch : gy + (long)q * gy_cs + ch;
      rv[j] = ld4(ok ? ptr : (isB ? x : gy));
      okmask |= (unsigned)ok << j;
      if (++ox == Wo) { ox = 0; if (++oy == Ho) { oy = 0; ++n; } }
    }
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
    char* base = (isB ? Bs[buf] : As[buf]) + loct * OCT + quad * 4 * 16;
    const float mul = isB ? b_mul : a_mul;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      h8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = rv[j][c] * ((okmask >> j & 1) ? mul : 0.f);
        hi[j] = (_Float16)v;
        lo[j] = (_Float16)(v - (float)hi[j]);
      }
      *reinterpret_cast<h8*>(base + c * 16) = hi;
      *reinterpret_cast<h8*>(base + PLANE + c * 16) = lo;
    }
  };

  f32x16 c[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;

  load(p0);
  store(0);
  __syncthreads();
  int buf = 0;
  for (int p = p0; p < p1; p += KS) {
    const bool more = p + KS < p1;
    if (more) load(p + KS);
#pragma unroll
    for (int ks = 0; ks < KS / 16; ++ks) {
      h8 a[2][2], b[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const int o = pl * PLANE + (ks * 2 + lh) * OCT;
          a[i][pl] = *reinterpret_cast<const h8*>(&As[buf][o + (wm * 64 + i * 32 + li) * 16]);
          b[i][pl] = *reinterpret_cast<const h8*>(&Bs[buf][o + (wn * 64 + i * 32 + li) * 16]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], c[i][j], 0, 0, 0);
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], c[i][j], 0, 0, 0);
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], c[i][j], 0, 0, 0);
        }
    }
    if (more) store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  const float sc = a_inv * b_inv;
  float* out = partial + ((size_t)chunk * K * K + tap) * Cout * Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = ci0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < Cout && col < Cin) out[(size_t)row * Cin + col] = c[i][j][r] * sc;
      }
    }
}

// 3x3 stride-1 "same" convs (the bulk of the network): one workgroup owns a whole KERNEL ROW -- the three taps
// kx = -1, 0, +1 of one ky -- for its 128 x 128 (co x ci) block and walks a band of 32 image rows column by column.
// The MFMA's K dimension runs DOWN a column (32 pixels at fixed x), so a horizontal tap shift is simply a
// neighbouring column image in LDS: the gy column and ONE new x column per step serve all three taps (a ring of four
// x-column images; the per-tap kernel above streams every operand 9 x tiles times through L2: 55 GB for the
// 496 -> 496 layer), every fragment stays an aligned ds_read_b128, and both loader halves do the same amount of
// conversion work.  36 MFMAs and 20 operand reads per wave and step (two waves per SIMD), one barrier per column.
constexpr int W3_KS = 32, W3_NOCT = W3_KS / 8, W3_OCT = 128 * 16, W3_PLANE = W3_NOCT * W3_OCT;
constexpr int w3_smem(int pieces) { return 6 * pieces * W3_PLANE; }   // gy column double buffer + ring of four x columns
// channel row r of an octet image sits at 16-byte slot r ^ ((r >> 4) & 3): the loader's rows 4q + c (fixed c) and the
// MFMA fragment reads' consecutive rows both fall on 16 distinct slots per ds_*_b128 lane group
__device__ __forceinline__ int w3_swz(int r) { return r ^ ((r >> 4) & 3); }
// F16 = true: the f16x3 scheme above (two fp16 pieces, operands rescaled from their |max| bounds, 3 products).
// F16 = false: bf16x6 -- three bf16 pieces = 24 significand bits, no rescaling, 6 products (fp32-equivalent, the mode
// the forward's bf16x6 convs run in): 24 KiB column images, 144 KiB of LDS.
typedef __bf16 w3b8 __attribute__((ext_vector_type(8)));
template <bool F16> struct W3Elt { typedef h8 vec; typedef _Float16 elt; };
template <> struct W3Elt<false> { typedef w3b8 vec; typedef __bf16 elt; };
template <bool F16>
__global__ __launch_bounds__(512, 1) void wgrad_col3_kernel(
    const float* __restrict__ x, int x_cs, const float* __restrict__ gy, int gy_cs, float* __restrict__ partial,
    const float* __restrict__ x_amax, const float* __restrict__ gy_amax, int N, int H, int W, int Cin, int Cout,
    int pad_t, int pad_l, int nbands, int nseg, int seg_w, int tiles_co, int tiles_ci) {
  constexpr int P = F16 ? 2 : 3;
  constexpr int W3_IMG = P * W3_PLANE;
  typedef typename W3Elt<F16>::vec ev8;
  typedef typename W3Elt<F16>::elt elt;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Aimg = smem;                                   // [2][W3_IMG]
  char* const Bimg = smem + 2 * W3_IMG;                      // [4][W3_IMG], slot = (source column + 1) & 3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // 8 waves: 2 (co halves of 64) x 4 (ci strips of 32).  A wave's gy fragments (2 sub-strips) serve all three taps and
  // each x fragment serves both sub-strips: 20 ds_read_b128 per 36 MFMAs (the 4 x 2 split read 28 -- 229 KB per
  // workgroup and step, 78 % of what the LDS delivers in the time the MFMAs take)
  const int wm = wave & 1, wn = wave >> 1;
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int kyi = id % 3; id /= 3;
  const int tci = id % tiles_ci; id /= tiles_ci;
  const int tco = id % tiles_co; id /= tiles_co;
  const int chunk = id;                                      // (n * nbands + band) * nseg + seg
  const int seg = id % nseg; id /= nseg;
  const int band = id % nbands;
  const int n = id / nbands;
  const int ky = kyi - pad_t;
  const int co0 = tco * 128, ci0 = tci * 128;
  const int y0 = band * W3_KS;
  const int xs = seg * seg_w, xe = min(W, xs + seg_w);
  float a_inv = 1.f, b_inv = 1.f, a_mul = 1.f, b_mul = 1.f;
  if (F16) { a_mul = f16_operand_scale(*gy_amax, &a_inv); b_mul = f16_operand_scale(*x_amax, &b_inv); }

  // loader: waves 0..3 (one per SIMD; the other wave of each SIMD only issues MFMAs, so conversion VALU work and
  // matrix work overlap): threads 0..127 stage the next gy column, 128..255 the next x column; a thread owns one
  // channel quad and 8 consecutive rows (a k-octet): 8 float4 loads (lanes = consecutive quads: coalesced),
  // transposed in registers
  const bool loader = tid < 256;
  const bool isB = tid >= 128;
  const int u = tid & 127, quad = u & 31, loct = u >> 5;
  const int ch = (isB ? ci0 : co0) + quad * 4;
  const bool ch_ok = ch < (isB ? Cin : Cout);
  const float* src = isB ? x : gy;
  const int cs = isB ? x_cs : gy_cs;
  const float mul = isB ? b_mul : a_mul;
  const int yk = y0 + loct * 8 + (isB ? ky : 0);             // first source row of this thread's octet
  f32x4 rv[8];                                               // branch-free loads, see wgrad_f16_kernel
  unsigned okmask = 0;
  auto load = [&](int col) __attribute__((always_inline)) {  // col: gy column (A) or x SOURCE column (B)
    const bool col_ok = ch_ok && (unsigned)col < (unsigned)W;
    okmask = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int yy = yk + j;
      const bool ok = col_ok && (unsigned)yy < (unsigned)H;
      rv[j] = ld4(ok ? src + (((long)n * H + yy) * W + col) * cs + ch : src);
      okmask |= (unsigned)ok << j;
    }
  };
  auto store = [&](char* img) __attribute__((always_inline)) {
    char* base = img + loct * W3_OCT;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ev8 pc[P];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = rv[j][c] * ((okmask >> j & 1) ? mul : 0.f);
#pragma unroll
        for (int pl = 0; pl < P; ++pl) {          // hi, then the rounding of what is left, ...
          pc[pl][j] = (elt)v;
          if (pl + 1 < P) v -= (float)pc[pl][j];
        }
      }
      const int pos = w3_swz(quad * 4 + c) * 16;
#pragma unroll
      for (int pl = 0; pl < P; ++pl) *reinterpret_cast<ev8*>(base + pl * W3_PLANE + pos) = pc[pl];
    }
  };
  auto bslot = [&](int srccol) { return Bimg + ((srccol + 1) & 3) * W3_IMG; };

  f32x16 c[3][2];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[k][j][r] = 0.f;

  // prologue: gy column xs; x source columns xs - pad_l + {0, 1, 2} (kx = 0..2 of output column xs)
  if (loader) {
    if (!isB) { load(xs); store(Aimg); }
    else {
      for (int t = 0; t < 3; ++t) { load(xs - pad_l + t); store(bslot(xs - pad_l + t)); }
    }
    if (xs + 1 < xe) load(isB ? xs + 1 - pad_l + 2 : xs + 1);   // registers: the data of the second column
  }
  __syncthreads();
  for (int col = xs; col < xe; ++col) {
    // loader waves FIRST convert and store the column fetched during the previous step (into the buffers the next
    // step reads), then issue the loads for the step after -- their conversion VALU work runs while the other wave of
    // the SIMD issues its MFMAs, and a load has a whole step to land
    if (loader && col + 1 < xe) {
      store(isB ? bslot(col + 1 - pad_l + 2) : Aimg + ((col + 1 - xs) & 1) * W3_IMG);
      if (col + 2 < xe) load(isB ? col + 2 - pad_l + 2 : col + 2);
    }
    const char* A = Aimg + ((col - xs) & 1) * W3_IMG;
#pragma unroll
    for (int ks = 0; ks < W3_KS / 16; ++ks) {
      ev8 a[2][P];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < P; ++pl)
          a[i][pl] = *reinterpret_cast<const ev8*>(A + pl * W3_PLANE + (ks * 2 + lh) * W3_OCT + w3_swz(wm * 64 + i * 32 + li) * 16);
#pragma unroll
      for (int kxi = 0; kxi < 3; ++kxi) {
        const char* B = bslot(col - pad_l + kxi);
        ev8 b[P];
#pragma unroll
        for (int pl = 0; pl < P; ++pl)
          b[pl] = *reinterpret_cast<const ev8*>(B + pl * W3_PLANE + (ks * 2 + lh) * W3_OCT + w3_swz(wn * 32 + li) * 16);
        // piece products, smallest first: (1,0) (0,1) (0,0) for two pieces; (2,0) (1,1) (0,2) (1,0) (0,1) (0,0) for three
#pragma unroll
        for (int order = P - 1; order >= 0; --order)
#pragma unroll
          for (int pa = order; pa >= 0; --pa)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              if constexpr (F16) c[kxi][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][pa], b[order - pa], c[kxi][i], 0, 0, 0);
              else c[kxi][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[order - pa], c[kxi][i], 0, 0, 0);
            }
      }
    }
    __syncthreads();
  }
  const float sc = a_inv * b_inv;
#pragma unroll
  for (int kxi = 0; kxi < 3; ++kxi) {
    float* out = partial + ((size_t)chunk * 9 + kyi * 3 + kxi) * Cout * Cin;
    const int colc = ci0 + wn * 32 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < Cout && colc < Cin) out[(size_t)row * Cin + colc] = c[kxi][i][r] * sc;
      }
    }
  }
}

// 64 elements x 4 chunk groups per block (two running sums per thread), the group sums added in a fixed order: a thread
// of the one-element-per-thread form walked all nchunk partial sets serially
__global__ __launch_bounds__(256) void wgrad_strided_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                                   int nchunk, int Cout, int Cin, int KK, int accumulate) {
  __shared__ float red[4][64];
  const int total = Cout * Cin * KK;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + e;
  const int per = (nchunk + 3) / 4;
  const int c0 = g * per, c1 = min(nchunk, c0 + per);
  float s0 = 0.f, s1 = 0.f;
  if (j < total) {
    int c = c0;
    for (; c + 1 < c1; c += 2) {
      s0 += partial[(size_t)c * total + j];
      s1 += partial[(size_t)(c + 1) * total + j];
    }
    if (c < c1) s0 += partial[(size_t)c * total + j];
  }
  red[g][e] = s0 + s1;
  __syncthreads();
  if (g == 0 && j < total) {
    const float s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    const int ci = j % Cin, co = (j / Cin) % Cout, tap = j / (Cin * Cout);
    const int i = (co * Cin + ci) * KK + tap;
    gw[i] = accumulate ? gw[i] + s : s;
  }
}

static inline int wgrad_chunks(long M, int K) {
  long n = 2048 / (K * K);
  n = n < 32 ? 32 : (n > 512 ? 512 : n);
  const long cap = (M + 255) / 256;
  return (int)(n < cap ? n : cap);
}

// ------------------------------------------------------------------------------------ depthwise conv backward
// gx[n,iy,ix,c] = sum_taps w[tap][c] * gy[n,(iy+pad_t-ky)/s,(ix+pad_l-kx)/s,c]  (where divisible and inside)
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                           float* __restrict__ gx, int N, int H, int W, int C, int Ho,
                                                           int Wo, int K, int stride, int pad_t, int pad_l) {
  const int cq = C >> 2;
  const long total = (long)N * H * W * cq;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % cq) * 4;
    long t = i / cq;
    const int ix = (int)(t % W); t /= W;
    const int iy = (int)(t % H);
    const int n = (int)(t / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < K; ++ky) {
      const int ty = iy + pad_t - ky;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < K; ++kx) {
        const int tx = ix + pad_l - kx;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= Wo) continue;
        acc = __builtin_elementwise_fma(ld4(gy + (((long)n * Ho + oy) * Wo + ox) * C + c), ld4(w + (ky * K + kx) * C + c), acc);
      }
    }
    st4(gx + i * 4, acc);
  }
}

// partial[blk][tap][c] = sum over the block's output pixels of gy[q][c] * x[q*s + tap - pad][c]
constexpr int DW_BLOCKS = 2048;           // upper bound; see dw_blocks()
static inline int dw_blocks(int C, int K) { const int b = 4194304 / (C * K * K); return b < 256 ? 256 : (b > DW_BLOCKS ? DW_BLOCKS : b); }
constexpr int DW_G = 4;                   // output pixels per step of the wgrad walk
template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           float* __restrict__ partial, int N, int H, int W, int C,
                                                           int Ho, int Wo, int pad_t, int pad_l) {
  extern __shared__ float sm[];   // [rows][Ct] per tap, reused tap by tap
  constexpr int WC = (DW_G - 1) * S + K;          // input columns under DW_G adjacent outputs
  constexpr int SH = DW_G * S, KEEP = WC > SH ? WC - SH : 0;
  {                                // blockIdx.y walks channel tiles of 256
    const int c0 = blockIdx.y * 256;
    const int Ct = min(256, C - c0), rows = 256 / Ct;
    const int c = c0 + threadIdx.x % Ct, row = threadIdx.x / Ct;
    const bool active = row < rows;
    float s[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) s[t] = 0.f;
    if (active) {
      // every (block, row slot) owns a CONTIGUOUS range of units = (image row, group of DW_G adjacent outputs) and keeps
      // the K x WC input window of the group in registers: along a row the KEEP columns shared with the next group
      // slide over and only DW_G * S new columns are loaded -- all of a step's loads (DW_G cotangents + K * DW_G * S
      // inputs) are independent and in flight together.  (One output per step with K dependent-free loads left the
      // walk latency-bound at 2-3 waves per SIMD: 200-400 us for tensors that stream in 40-60.)
      const int gpr = (Wo + DW_G - 1) / DW_G;
      const long U = (long)N * Ho * gpr;
      const long nslot = (long)gridDim.x * rows;
      const long per = (U + nslot - 1) / nslot;
      const long u0 = ((long)blockIdx.x * rows + row) * per, u1 = min(U, u0 + per);
      float win[K][WC];
      for (long u = u0; u < u1; ++u) {
        const long r = u / gpr;
        const int gi = (int)(u - r * gpr), n = (int)(r / Ho), oy = (int)(r - (long)n * Ho);
        const int ox0 = gi * DW_G;
        const bool cont = u > u0 && gi != 0;             // the previous unit was the group to the left
        float g[DW_G];
#pragma unroll
        for (int j = 0; j < DW_G; ++j) {
          const bool ok = ox0 + j < Wo;
          const float v = gy[(r * Wo + (ok ? ox0 + j : 0)) * C + c];
          g[j] = ok ? v : 0.f;
        }
        const float* xn = x + (long)n * H * W * C + c;
        const int ix0 = ox0 * S - pad_l, iy0 = oy * S - pad_t;
        auto load_cols = [&](int from) __attribute__((always_inline)) {
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            const int iy = iy0 + ky;
            const bool yok = (unsigned)iy < (unsigned)H;
            const long rowoff = (long)(yok ? iy : 0) * W;
#pragma unroll
            for (int j = 0; j < WC; ++j) {
              if (j >= from) {
                const int ix = ix0 + j;
                const bool ok = yok && (unsigned)ix < (unsigned)W;
                const float xv = xn[(rowoff + (ok ? ix : 0)) * C];   // branch-free: invalid taps read a valid address
                win[ky][j] = ok ? xv : 0.f;
              }
            }
          }
        };
        if (cont && KEEP > 0) {
#pragma unroll
          for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int j = 0; j < KEEP; ++j) win[ky][j] = win[ky][j + SH];
          load_cols(KEEP);
        } else {
          load_cols(0);
        }
#pragma unroll
        for (int j = 0; j < DW_G; ++j)
#pragma unroll
          for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
              s[ky * K + kx] = __builtin_fmaf(g[j], win[ky][j * S + kx], s[ky * K + kx]);   // (-ffp-contract=off build)
      }
    }
    const int cw = Ct;                           // channels handled by this block
    for (int t = 0; t < K * K; ++t) {
      __syncthreads();
      if (active) sm[row * cw + (c - c0)] = s[t];
      __syncthreads();
      if (active && row == 0) {
        float tot = 0.f;
        for (int r = 0; r < rows; ++r) tot += sm[r * cw + (c - c0)];
        partial[((size_t)blockIdx.x * K * K + t) * C + c] = tot;
      }
    }
  }
}

// out[i] (+)= sum_b partial[b][i] ; one wave per output (ordered lanes + xor tree)
__global__ __launch_bounds__(64) void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                          int nblocks, int n, float scale, int accumulate) {
  const int i = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[(size_t)b * n + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[i] = (accumulate ? out[i] : 0.f) + s * scale;
}

// ------------------------------------------------------------------------------------ elementwise
//   op 0 swish             o = a * sigmoid(a)
//   op 1 swish backward    o = b * (sg + a*sg*(1-sg)),  sg = sigmoid(a)      (a = pre-activation, b = gy)
//   op 2 gate              o = a * g[n][c]
//   op 3 gate + residual   o = b + a * g[n][c]        (g may have C == 1 stride: per-sample scalar, drop-connect)
//   op 4 gate backward     o = b * g[n][c] + r[n][c] / HW    (b = gy, r = gradient of the pooled mean; r may be null)
__global__ __launch_bounds__(256) void train_pointwise_kernel(int op, const float* __restrict__ a, int a_cs,
                                                              const float* __restrict__ b, int b_cs,
                                                              const float* __restrict__ g, int g_c,
                                                              const float* __restrict__ r, float* __restrict__ o,
                                                              int o_cs, long HW, long P, int C, float* out_amax) {
  __shared__ float amax_scratch[4];
  float vmax = 0.f;
  const long total = P * C;
  const float inv_hw = 1.f / (float)HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / C;
    const int c = (int)(i - p * C);
    const long n = p / HW;
    float v;
    if (op == 0) { const float x = a[p * a_cs + c]; v = x / (1.f + expf(-x)); }
    else if (op == 1) {
      const float x = a[p * a_cs + c], sg = 1.f / (1.f + expf(-x));
      v = b[p * b_cs + c] * (sg + x * sg * (1.f - sg));
    } else if (op == 2) v = a[p * a_cs + c] * g[n * g_c + (g_c == 1 ? 0 : c)];
    else if (op == 3) v = b[p * b_cs + c] + a[p * a_cs + c] * g[n * g_c + (g_c == 1 ? 0 : c)];
    else v = b[p * b_cs + c] * g[n * g_c + (g_c == 1 ? 0 : c)] + (r ? r[n * C + c] * inv_hw : 0.f);
    vmax = fmaxf(vmax, fabsf(v));
    o[p * o_cs + c] = v;
  }
  if (out_amax) block_amax_update(vmax, out_amax, amax_scratch);
}

// float4 variant (C, strides multiples of 4; 16-byte aligned; < 2^31 quads), op as a template parameter
template <int OP>
__global__ __launch_bounds__(256) void train_pointwise4_kernel(const float* __restrict__ a, int a_cs,
                                                               const float* __restrict__ b, int b_cs,
                                                               const float* __restrict__ g, int g_c,
                                                               const float* __restrict__ r, float* __restrict__ o,
                                                               int o_cs, unsigned HW, unsigned P, int C, float* out_amax) {
  __shared__ float amax_scratch[4];
  float vmax = 0.f;
  const unsigned Cq = (unsigned)C >> 2, total = P * Cq;
  const float inv_hw = 1.f / (float)HW;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned p = i / Cq;
    const int c = (int)(i - p * Cq) << 2;
    const unsigned n = p / HW;
    const f32x4 av = ld4(a + (long)p * a_cs + c);
    f32x4 v;
    if (OP == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = av[j] / (1.f + expf(-av[j]));
    } else if (OP == 1) {
      const f32x4 bv = ld4(b + (long)p * b_cs + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sg = 1.f / (1.f + expf(-av[j]));
        v[j] = bv[j] * (sg + av[j] * sg * (1.f - sg));
      }
    } else {
      f32x4 gv;
      if (g_c == 1) { const float g1 = g[n]; gv = f32x4{g1, g1, g1, g1}; }
      else gv = ld4(g + (long)n * g_c + c);
      if (OP == 2) v = av * gv;
      else if (OP == 3) v = ld4(b + (long)p * b_cs + c) + av * gv;
      else {
        v = ld4(b + (long)p * b_cs + c) * gv;
        if (r) v += ld4(r + (long)n * C + c) * inv_hw;
      }
    }
    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    *reinterpret_cast<f32x4*>(o + (long)p * o_cs + c) = v;
  }
  if (out_amax) block_amax_update(vmax, out_amax, amax_scratch);
}

// per-sample channel sums: out[n][c] = scale * sum_hw a[n,p,c] * (b ? b[n,p,c] : 1); grid (chunks, N)
constexpr int SR_CHUNKS = 256;
__global__ __launch_bounds__(256) void sample_reduce_kernel(const float* __restrict__ a, int a_cs,
                                                            const float* __restrict__ b, int b_cs,
                                                            float* __restrict__ partial, long HW, int C) {
  extern __shared__ float sm[];
  const int n = blockIdx.y;
  const int rows = 256 / C > 0 ? 256 / C : 1;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + (C >= 256 ? threadIdx.x : threadIdx.x % C);
    const int row = C >= 256 ? 0 : threadIdx.x / C;
    const bool active = c < C && row < rows;
    float s = 0.f;
    if (active)
#pragma unroll 4
      for (long p = (long)blockIdx.x * rows + row; p < HW; p += (long)gridDim.x * rows) {
        const long q = (long)n * HW + p;
        s += a[q * a_cs + c] * (b ? b[q * b_cs + c] : 1.f);
      }
    const int cw = C >= 256 ? 256 : C;
    __syncthreads();
    if (active) sm[row * cw + (c - c0)] = s;
    __syncthreads();
    if (active && row == 0) {
      float tot = 0.f;
      for (int r = 0; r < rows; ++r) tot += sm[r * cw + (c - c0)];
      partial[((size_t)n * gridDim.x + blockIdx.x) * C + c] = tot;
    }
  }
}

// one wave per output: lane l sums chunks l, l+64, ... in order, then a fixed xor tree (a thread per output walked
// the 256 chunks as 256 dependent loads: 60 us for a 9 KB result)
__global__ __launch_bounds__(256) void sample_reduce_finalize_kernel(const float* __restrict__ partial,
                                                                     float* __restrict__ out, int chunks, int NC, int C,
                                                                     float scale) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= NC) return;
  const int n = i / C, c = i % C;
  float s = 0.f;
  for (int k = lane; k < chunks; k += 64) s += partial[((size_t)n * chunks + k) * C + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[i] = s * scale;
}

// ------------------------------------------------------------------------------------ squeeze-excite FCs
// Grid (sample, 128-channel slice), 1024 threads: every workgroup computes the Cse hidden units itself (a few microseconds of L2
// reads) and then only ITS slice of the C outputs -- one 256-thread workgroup per sample walked 12 rounds of the first FC and
// C / 256 passes of Cse dependent loads of the second (36 / 31 us per launch, 32 launches per step, 8 workgroups on the chip).
constexpr int SEFC_T = 1024, SEFC_SLICE = 128;
// forward: hpre = W1 s + b1 ; h = swish(hpre) ; z = W2 h + b2 ; gate = sigmoid(z)
__global__ __launch_bounds__(SEFC_T) void se_fc_forward_kernel(const float* __restrict__ s, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, float* __restrict__ hpre,
                                                               float* __restrict__ hact, float* __restrict__ gate, int C,
                                                               int Cse) {
  extern __shared__ float sm[];   // s[C] | h[Cse]
  float* ss = sm; float* hh = sm + C;
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += SEFC_T) ss[c] = s[(size_t)n * C + c];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = wave; j < Cse; j += SEFC_T / 64) {
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc += w1[(size_t)j * C + c] * ss[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      const float hp = acc + b1[j];
      hh[j] = hp / (1.f + expf(-hp));
      if (blockIdx.y == 0) { hpre[(size_t)n * Cse + j] = hp; hact[(size_t)n * Cse + j] = hh[j]; }
    }
  }
  __syncthreads();
  const int c_lo = blockIdx.y * SEFC_SLICE, c_hi = c_lo + SEFC_SLICE < C ? c_lo + SEFC_SLICE : C;
  for (int c = c_lo + threadIdx.x; c < c_hi; c += SEFC_T) {
    float z = b2[c];
    for (int j = 0; j < Cse; ++j) z += w2[(size_t)c * Cse + j] * hh[j];
    gate[(size_t)n * C + c] = 1.f / (1.f + expf(-z));
  }
}

// backward: gg = d loss / d gate ->
//   gz = gg*gate*(1-gate) ; gh = W2^T gz ; ghpre = gh * swish'(hpre) ; gs = W1^T ghpre
// gz and ghpre are kept for the weight gradients (summed over the batch by se_fc_wgrad_kernel).
__global__ __launch_bounds__(SEFC_T) void se_fc_backward_kernel(const float* __restrict__ gg, const float* __restrict__ gate,
                                                                const float* __restrict__ hpre,
                                                                const float* __restrict__ w1, const float* __restrict__ w2,
                                                                float* __restrict__ gz, float* __restrict__ ghpre,
                                                                float* __restrict__ gs, int C, int Cse) {
  extern __shared__ float sm[];   // gz[C] | ghpre[Cse]
  float* sz = sm; float* sh = sm + C;
  const int n = blockIdx.x;
  const bool first = blockIdx.y == 0;
  for (int c = threadIdx.x; c < C; c += SEFC_T) {
    const float g = gate[(size_t)n * C + c];
    const float v = gg[(size_t)n * C + c] * g * (1.f - g);
    sz[c] = v;
    if (first) gz[(size_t)n * C + c] = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = wave; j < Cse; j += SEFC_T / 64) {
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc += w2[(size_t)c * Cse + j] * sz[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      const float x = hpre[(size_t)n * Cse + j], sg = 1.f / (1.f + expf(-x));
      const float v = acc * (sg + x * sg * (1.f - sg));
      sh[j] = v;
      if (first) ghpre[(size_t)n * Cse + j] = v;
    }
  }
  __syncthreads();
  const int c_lo = blockIdx.y * SEFC_SLICE, c_hi = c_lo + SEFC_SLICE < C ? c_lo + SEFC_SLICE : C;
  for (int c = c_lo + threadIdx.x; c < c_hi; c += SEFC_T) {
    float acc = 0.f;
    for (int j = 0; j < Cse; ++j) acc += w1[(size_t)j * C + c] * sh[j];
    gs[(size_t)n * C + c] = acc;
  }
}

// gw[o][i] (+)= sum_n go[n][o] * in[n][i] ; gb[o] (+)= sum_n go[n][o]
__global__ void fc_wgrad_kernel(const float* __restrict__ go, const float* __restrict__ in, float* __restrict__ gw,
                                float* __restrict__ gb, int N, int O, int I, int accumulate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < O * I) {
    const int o = idx / I, i = idx % I;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += go[(size_t)n * O + o] * in[(size_t)n * I + i];
    gw[idx] = accumulate ? gw[idx] + s : s;
  }
  if (idx < O) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += go[(size_t)n * O + idx];
    gb[idx] = accumulate ? gb[idx] + s : s;
  }
}

// ------------------------------------------------------------------------------------ losses
// Depth classification (reference CrossEntropyDepth): label bin = trunc((d - dmin) / bin_size) (UD), invalid when
// outside [0, num_bins) or non-finite (index num_bins); loss = mean over valid pixels of -log softmax[bin].
// Pass 1 (count_valid) counts the valid pixels; pass 2 evaluates loss / accuracy partial sums and writes
// g_logits = weight * (softmax - onehot) / n_valid (zero rows for invalid pixels).  32 lanes per pixel, C = 128.
__device__ __forceinline__ int depth_bin(float d, float dmin, float bin_size, int nb) {
  const float idx = (d - dmin) / bin_size;
  if (!(idx >= 0.f) || idx > (float)nb || !isfinite(idx)) return nb;
  return (int)idx;
}

__global__ __launch_bounds__(256) void depth_count_valid_kernel(const float* __restrict__ gt, long P, float dmin,
                                                                float bin_size, int nb, float* __restrict__ partial) {
  __shared__ float sm[4];
  float cnt = 0.f;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < P; p += (long)gridDim.x * 256)
    cnt += depth_bin(gt[p], dmin, bin_size, nb) != nb ? 1.f : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void depth_ce_kernel(const float* __restrict__ logits, int cs, const float* __restrict__ gt,
                                                       long P, float dmin, float bin_size, int nb,
                                                       const float* __restrict__ n_valid, float weight,
                                                       float* __restrict__ g_logits, int g_cs,
                                                       float* __restrict__ partial /* [blocks][2] */) {
  __shared__ float sm[8][2];
  const int sub = threadIdx.x & 31;
  const long half = (blockIdx.x * 256L + threadIdx.x) >> 5;
  const long nhalf = ((long)gridDim.x * 256) >> 5;
  const float inv_n = 1.f / fmaxf(*n_valid, 1.f);
  float loss = 0.f, hit = 0.f;
  for (long p = half; p < P; p += nhalf) {
    const int bin = depth_bin(gt[p], dmin, bin_size, nb);
    f32x4 gout = {0.f, 0.f, 0.f, 0.f};
    if (bin != nb) {
      const f32x4 x = ld4(logits + p * cs + sub * 4);
      float mx = x[0]; int am = sub * 4;
#pragma unroll
      for (int j = 1; j < 4; ++j) if (x[j] > mx) { mx = x[j]; am = sub * 4 + j; }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o);
        const int oa = __shfl_xor(am, o);
        if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
      }
      f32x4 e;
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { e[j] = expf(x[j] - mx); se += e[j]; }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) se += __shfl_xor(se, o);
      const float inv = 1.f / se;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = sub * 4 + j;
        gout[j] = (e[j] * inv - (k == bin ? 1.f : 0.f)) * inv_n * weight;
        if (k == bin) loss += -(x[j] - mx - logf(se));
      }
      if (sub == 0 && am == bin) hit += 1.f;
    }
    if (g_logits) st4(g_logits + p * g_cs + sub * 4, gout);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { loss += __shfl_xor(loss, o, 64); hit += __shfl_xor(hit, o, 64); }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = loss; sm[threadIdx.x >> 6][1] = hit; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
    partial[2 * blockIdx.x + 1] = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
  }
}

// masked MSE (reference MSELoss: mean over the elements whose label is not +-inf)
//   pass 1: partial[blk] = (sum (pred-gt)^2, count) ; pass 2: g = weight * 2 (pred-gt) / count (0 where masked)
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ pred, int p_cs,
                                                          const float* __restrict__ gt, int g_cs, long P, int C,
                                                          float* __restrict__ partial) {
  __shared__ float sm[4][2];
  float s = 0.f, n = 0.f;
  const long total = P * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / C; const int c = (int)(i - p * C);
    const float g = gt[p * g_cs + c];
    if (!isinf(g)) { const float d = pred[p * p_cs + c] - g; s += d * d; n += 1.f; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); n += __shfl_xor(n, o, 64); }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = s; sm[threadIdx.x >> 6][1] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
    partial[2 * blockIdx.x + 1] = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
  }
}

__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ pred, int p_cs,
                                                       const float* __restrict__ gt, int g_cs, long P, int C,
                                                       const float* __restrict__ sums /* [2]: sq, count */, float weight,
                                                       float* __restrict__ g, int o_cs) {
  const float sc = 2.f * weight / fmaxf(sums[1], 1.f);
  const long total = P * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / C; const int c = (int)(i - p * C);
    const float t = gt[p * g_cs + c];
    g[p * o_cs + c] = isinf(t) ? 0.f : sc * (pred[p * p_cs + c] - t);
  }
}

__global__ void depth_ce_finish_kernel(const float* __restrict__ sums, float* __restrict__ out3) {
  const float n = fmaxf(out3[2], 1.f);
  out3[0] = sums[0] / n;
  out3[1] = sums[1] / n;
}

__global__ void mse_finish_kernel(const float* __restrict__ sums, float* __restrict__ out2) {
  out2[0] = sums[0] / fmaxf(sums[1], 1.f);
  out2[1] = sums[1];
}

// out[k] = sum_b partial[b][k], k < nk <= 4: one wave per output (launch <<<1, 256>>>), lane l adds blocks l, l + 64, ... in order,
// then a fixed xor tree (one thread per output walked up to 1024 dependent loads: 68 us for a scalar)
__global__ void reduce_small_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblocks, int nk) {
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (k >= nk) return;
  float s = 0.f;
  for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * nk + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[k] = s;
}

}  // namespace creste

using namespace creste;

extern "C" int64_t creste_conv_wgrad_strided_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int K) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || K <= 0) return -1;
  const long M = (long)N * Ho * Wo;
  const int chunks = wgrad_smallcin_ok(Cin, Cout, K) ? wgrad_smallcin_chunks(M) : wgrad_chunks(M, K);
  return (int64_t)chunks * K * K * Cout * Cin * 4;
}

extern "C" int creste_conv_wgrad_strided_f32(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, int N,
                                             int H, int W, int Ho, int Wo, int Cin, int Cout, int K, int stride,
                                             int pad_t, int pad_l, int accumulate, void* work, void* stream) {
  CRESTE_REQUIRE(x && gy && gw && work, "conv_wgrad_strided: null pointer");
  CRESTE_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cout > 0 && K > 0 && stride > 0 &&
                     K * K <= 65535, "conv_wgrad_strided: bad dims");
  const long M = (long)N * Ho * Wo;
  CRESTE_REQUIRE(M < (1L << 31), "conv_wgrad_strided: N*Ho*Wo overflows int32");
  // thin 1x1 projections (2 / 6 classes): the streaming reduction of csrc/train.hip (same workspace bound)
  if (K == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && Ho == H && Wo == W && Cout < 8 && Cin % 4 == 0 && Cin <= 1024 &&
      x_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    return creste_conv_wgrad_f32(x, x_cs, gy, gy_cs, gw, N, H, W, Cin, Cout, 1, 0, accumulate, work, stream);
  hipStream_t s = (hipStream_t)stream;
  int nchunk;
  const bool vec = Cin % 4 == 0 && Cout % 4 == 0 && x_cs % 4 == 0 && gy_cs % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0;
  if (wgrad_smallcin_ok(Cin, Cout, K)) {
    nchunk = wgrad_smallcin_chunks(M);
    long wave_px = (M + 4L * nchunk - 1) / (4L * nchunk);
    wave_px = (wave_px + 3) / 4 * 4;
    nchunk = (int)((M + 4 * wave_px - 1) / (4 * wave_px));
    wgrad_smallcin_kernel<<<nchunk, 256, 0, s>>>(x, x_cs, gy, gy_cs, (float*)work, N, H, W, Ho, Wo, Cin, Cout, K, stride,
                                                 pad_t, pad_l, (int)wave_px);
    CRESTE_CHECK_LAUNCH("wgrad_smallcin");
  } else if (vec && Cin >= 32 && Cout >= 32) {
    const bool big = Cin > 64 && Cout > 64;
    const int bm = big ? 128 : 64;
    const int tiles_co = (Cout + bm - 1) / bm, tiles_ci = (Cin + bm - 1) / bm;
    const long tiles = (long)tiles_co * tiles_ci * K * K;
    long want = (2048 + tiles - 1) / tiles;                       // ~2048 workgroups in flight
    const long cap = wgrad_chunks(M, K);                          // the workspace was sized for at most this many
    nchunk = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    long chunk_px = (M + nchunk - 1) / nchunk;
    chunk_px = (chunk_px + WT_KS - 1) / WT_KS * WT_KS;
    nchunk = (int)((M + chunk_px - 1) / chunk_px);
    const dim3 grid(nchunk * tiles_co * tiles_ci, K * K);
    if (big)
      wgrad_tiled_kernel<2, 2><<<grid, 256, 0, s>>>(x, x_cs, gy, gy_cs, (float*)work, N, H, W, Ho, Wo, Cin, Cout, K, stride,
                                                   pad_t, pad_l, (int)chunk_px, tiles_co, tiles_ci);
    else
      wgrad_tiled_kernel<1, 1><<<grid, 256, 0, s>>>(x, x_cs, gy, gy_cs, (float*)work, N, H, W, Ho, Wo, Cin, Cout, K, stride,
                                                   pad_t, pad_l, (int)chunk_px, tiles_co, tiles_ci);
    CRESTE_CHECK_LAUNCH("wgrad_tiled");
  } else {
    nchunk = wgrad_chunks(M, K);
    long chunk_px = (M + nchunk - 1) / nchunk;
    chunk_px = (chunk_px + WGS_PIX - 1) / WGS_PIX * WGS_PIX;
    nchunk = (int)((M + chunk_px - 1) / chunk_px);
    wgrad_strided_partial_kernel<<<dim3(nchunk, K * K), 256, 0, s>>>(x, x_cs, gy, gy_cs, (float*)work, N, H, W, Ho, Wo,
                                                                    Cin, Cout, K, stride, pad_t, pad_l, (int)chunk_px);
    CRESTE_CHECK_LAUNCH("wgrad_strided_partial");
  }
  wgrad_strided_reduce_kernel<<<(unsigned)(((long)Cout * Cin * K * K + 63) / 64), 256, 0, s>>>((const float*)work, gw, nchunk,
                                                                                    Cout, Cin, K * K, accumulate);
  CRESTE_CHECK_LAUNCH("wgrad_strided_reduce");
  return CRESTE_OK;
}

extern "C" int creste_conv_wgrad_f16x3(const float* x, int x_cs, const float* gy, int gy_cs, float* gw,
                                       const float* x_amax, const float* gy_amax, int N, int H, int W, int Ho, int Wo,
                                       int Cin, int Cout, int K, int stride, int pad_t, int pad_l, int accumulate,
                                       void* work, void* stream) {
  CRESTE_REQUIRE(x && gy && gw && work && x_amax && gy_amax, "conv_wgrad_f16x3: null pointer");
  CRESTE_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cout > 0 && K > 0 && stride > 0 &&
                     K * K <= 65535, "conv_wgrad_f16x3: bad dims");
  CRESTE_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && x_cs % 4 == 0 && gy_cs % 4 == 0 && ((uintptr_t)x & 15) == 0 &&
                     ((uintptr_t)gy & 15) == 0,
                 "conv_wgrad_f16x3: channel counts and pixel strides must be multiples of 4 (16-byte operand quads)");
  const long M = (long)N * Ho * Wo;
  CRESTE_REQUIRE(M < (1L << 31), "conv_wgrad_f16x3: N*Ho*Wo overflows int32");
  const int tiles_co = (Cout + 127) / 128, tiles_ci = (Cin + 127) / 128;
  const long tiles = (long)tiles_co * tiles_ci * K * K;
  long want = (2048 + tiles - 1) / tiles;
  const long cap = wgrad_chunks(M, K);
  int nchunk = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  long chunk_px = (M + nchunk - 1) / nchunk;
  chunk_px = (chunk_px + 31) / 32 * 32;
  nchunk = (int)((M + chunk_px - 1) / chunk_px);
  hipStream_t s = (hipStream_t)stream;
  CRESTE_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && x_cs % 4 == 0 && gy_cs % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0,
                 "conv_wgrad_f16x3: channel counts / strides must be multiples of 4, tensors 16-byte aligned");
  if (K == 3 && stride == 1 && Ho == H && Wo == W && Cin >= 64 && Cout >= 64) {          // kernel-row / column-walk variant
    static std::atomic<uint64_t> attr_devs{0};
    CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wgrad_col3_kernel<true>), w3_smem(2), attr_devs));
    const int nbands = (H + W3_KS - 1) / W3_KS;
    const long base = (long)N * nbands;                            // chunks before column segmentation
    const long tiles3 = (long)tiles_co * tiles_ci * 3;
    long nseg = (512 + tiles3 * base - 1) / (tiles3 * base);       // >= 512 workgroups (A/B on one box: 1024 -> 512: SSC step 125.5 -> 124.4 ms; fewer partial sets to reduce)
    const long seg_cap = cap / base > 0 ? cap / base : 1;          // the workspace holds `cap` partial sets
    nseg = nseg < 1 ? 1 : (nseg > seg_cap ? seg_cap : nseg);
    if (nseg > W) nseg = W;
    CRESTE_REQUIRE(base * nseg <= cap || nseg == 1, "conv_wgrad_f16x3: workspace too small");
    if (base <= cap) {
      const int seg_w = (int)((W + nseg - 1) / nseg);
      nseg = (W + seg_w - 1) / seg_w;
      const int nchunk3 = (int)(base * nseg);
      wgrad_col3_kernel<true><<<(unsigned)(nchunk3 * tiles3), 512, w3_smem(2), s>>>(x, x_cs, gy, gy_cs, (float*)work, x_amax, gy_amax,
                                                                             N, H, W, Cin, Cout, pad_t, pad_l, nbands,
                                                                             (int)nseg, seg_w, tiles_co, tiles_ci);
      CRESTE_CHECK_LAUNCH("wgrad_f16_col3");
      wgrad_strided_reduce_kernel<<<(unsigned)(((long)Cout * Cin * 9 + 63) / 64), 256, 0, s>>>((const float*)work, gw, nchunk3, Cout,
                                                                                 Cin, 9, accumulate);
      CRESTE_CHECK_LAUNCH("wgrad_strided_reduce");
      return CRESTE_OK;
    }
  }
  wgrad_f16_kernel<<<nchunk * tiles_co * tiles_ci * K * K, 256, 0, s>>>(
      x, x_cs, gy, gy_cs, (float*)work, x_amax, gy_amax, N, H, W, Ho, Wo, Cin, Cout, K, stride, pad_t, pad_l,
      (int)chunk_px, tiles_co, tiles_ci);
  CRESTE_CHECK_LAUNCH("wgrad_f16");
  wgrad_strided_reduce_kernel<<<(unsigned)(((long)Cout * Cin * K * K + 63) / 64), 256, 0, s>>>((const float*)work, gw, nchunk,
                                                                                    Cout, Cin, K * K, accumulate);
  CRESTE_CHECK_LAUNCH("wgrad_strided_reduce");
  return CRESTE_OK;
}

// Weight gradient at the fp32-equivalent bf16x6 operand grade (three bf16 pieces per operand, six piece products on the
// bf16 MFMA, fp32 accumulation): the stride-1 "same" 3x3 convs with >= 64 channels on both sides run the column-walk
// kernel above; every other shape is the exact-fp32 MFMA path (creste_conv_wgrad_strided_f32), same arguments.
extern "C" int creste_conv_wgrad_bf16x6(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, int N, int H, int W,
                                        int Ho, int Wo, int Cin, int Cout, int K, int stride, int pad_t, int pad_l,
                                        int accumulate, void* work, void* stream) {
  CRESTE_REQUIRE(x && gy && gw && work, "conv_wgrad_bf16x6: null pointer");
  CRESTE_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cout > 0 && K > 0 && stride > 0, "conv_wgrad_bf16x6: bad dims");
  const bool quads = Cin % 4 == 0 && Cout % 4 == 0 && x_cs % 4 == 0 && gy_cs % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0;
  const long M = (long)N * Ho * Wo;
  if (quads && K == 3 && stride == 1 && Ho == H && Wo == W && Cin >= 64 && Cout >= 64 && M < (1L << 31)) {
    const int tiles_co = (Cout + 127) / 128, tiles_ci = (Cin + 127) / 128;
    const long cap = wgrad_chunks(M, K);
    const int nbands = (H + W3_KS - 1) / W3_KS;
    const long base = (long)N * nbands;
    const long tiles3 = (long)tiles_co * tiles_ci * 3;
    long nseg = (512 + tiles3 * base - 1) / (tiles3 * base);
    const long seg_cap = cap / base > 0 ? cap / base : 1;
    nseg = nseg < 1 ? 1 : (nseg > seg_cap ? seg_cap : nseg);
    if (nseg > W) nseg = W;
    if (base <= cap && base * nseg <= cap) {
      hipStream_t s = (hipStream_t)stream;
      static std::atomic<uint64_t> attr_devs{0};
      CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wgrad_col3_kernel<false>), w3_smem(3), attr_devs));
      const int seg_w = (int)((W + nseg - 1) / nseg);
      nseg = (W + seg_w - 1) / seg_w;
      const int nchunk3 = (int)(base * nseg);
      wgrad_col3_kernel<false><<<(unsigned)(nchunk3 * tiles3), 512, w3_smem(3), s>>>(x, x_cs, gy, gy_cs, (float*)work, nullptr, nullptr,
                                                                                    N, H, W, Cin, Cout, pad_t, pad_l, nbands,
                                                                                    (int)nseg, seg_w, tiles_co, tiles_ci);
      CRESTE_CHECK_LAUNCH("wgrad_bf16_col3");
      wgrad_strided_reduce_kernel<<<(unsigned)(((long)Cout * Cin * 9 + 63) / 64), 256, 0, s>>>((const float*)work, gw, nchunk3, Cout,
                                                                                 Cin, 9, accumulate);
      CRESTE_CHECK_LAUNCH("wgrad_strided_reduce");
      return CRESTE_OK;
    }
  }
  return creste_conv_wgrad_strided_f32(x, x_cs, gy, gy_cs, gw, N, H, W, Ho, Wo, Cin, Cout, K, stride, pad_t, pad_l, accumulate,
                                       work, stream);
}

extern "C" int creste_dwconv_dgrad_f32(const float* gy, const float* w, float* gx, int N, int H, int W, int C, int Ho,
                                       int Wo, int K, int stride, int pad_t, int pad_l, void* stream) {
  CRESTE_REQUIRE(gy && w && gx && C % 4 == 0 && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && K > 0 && stride > 0,
                 "dwconv_dgrad: bad args");
  dwconv_dgrad_kernel<<<grid1d((long)N * H * W * (C / 4)), 256, 0, (hipStream_t)stream>>>(gy, w, gx, N, H, W, C, Ho, Wo,
                                                                                         K, stride, pad_t, pad_l);
  CRESTE_CHECK_LAUNCH("dwconv_dgrad");
  return CRESTE_OK;
}

extern "C" int64_t creste_dwconv_wgrad_workspace_bytes(int C, int K) {
  return C > 0 && K > 0 ? (int64_t)dw_blocks(C, K) * K * K * C * 4 : -1;
}

extern "C" int creste_dwconv_wgrad_f32(const float* x, const float* gy, float* gw_taps, int N, int H, int W, int C,
                                       int Ho, int Wo, int K, int stride, int pad_t, int pad_l, int accumulate,
                                       void* work, void* stream) {
  CRESTE_REQUIRE(x && gy && gw_taps && work && C > 0 && N > 0, "dwconv_wgrad: bad args");
  CRESTE_REQUIRE(K == 3 || K == 5, "dwconv_wgrad: kernel size %d not built (3 or 5)", K);
  CRESTE_REQUIRE(stride == 1 || stride == 2, "dwconv_wgrad: stride %d not built (1 or 2: the window slides by it)", stride);
  const int rows = C >= 256 ? 1 : 256 / C;
  const long U = (long)N * Ho * ((Wo + DW_G - 1) / DW_G);            // (row, group of DW_G outputs) units
  const long per = (U + rows - 1) / rows;
  const int blocks = (int)(per < dw_blocks(C, K) ? per : dw_blocks(C, K));
  const size_t smem = 256 * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(blocks, (C + 255) / 256);
#define CRESTE_DWW(K_, S_) dwconv_wgrad_kernel<K_, S_><<<grid, 256, smem, s>>>(x, gy, (float*)work, N, H, W, C, Ho, Wo, pad_t, pad_l)
  if (K == 3 && stride == 1) CRESTE_DWW(3, 1);
  else if (K == 3) CRESTE_DWW(3, 2);
  else if (stride == 1) CRESTE_DWW(5, 1);
  else CRESTE_DWW(5, 2);
#undef CRESTE_DWW
  CRESTE_CHECK_LAUNCH("dwconv_wgrad");
  sum_partials_kernel<<<K * K * C, 64, 0, s>>>((const float*)work, gw_taps, blocks, K * K * C, 1.f, accumulate);
  CRESTE_CHECK_LAUNCH("dwconv_wgrad_sum");
  return CRESTE_OK;
}

extern "C" int creste_train_pointwise_f32(int op, const float* a, int a_cs, const float* b, int b_cs, const float* g,
                                          int g_c, const float* r, float* o, int o_cs, int64_t HW, int64_t P, int C,
                                          float* out_amax, void* stream) {
  CRESTE_REQUIRE(a && o && op >= 0 && op <= 4 && P > 0 && C > 0 && HW > 0, "train_pointwise: bad args");
  CRESTE_REQUIRE((op != 1 && op != 3 && op != 4) || b, "train_pointwise: op %d needs b", op);
  CRESTE_REQUIRE(op < 2 || g, "train_pointwise: op %d needs the gate", op);
  const bool vec = C % 4 == 0 && a_cs % 4 == 0 && o_cs % 4 == 0 && (!b || b_cs % 4 == 0) && (!g || g_c == 1 || g_c % 4 == 0) &&
                   P * (C / 4) < 2147483647L && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)o | (uintptr_t)r) & 15) == 0 &&
                   (!g || g_c == 1 || ((uintptr_t)g & 15) == 0);
  if (vec) {
    hipStream_t s = (hipStream_t)stream;
    const int g4 = grid1d(P * C / 4, 1024);
#define CRESTE_TP4(OP) train_pointwise4_kernel<OP><<<g4, 256, 0, s>>>(a, a_cs, b, b_cs, g, g_c, r, o, o_cs, (unsigned)HW, (unsigned)P, C, out_amax)
    if (op == 0) CRESTE_TP4(0); else if (op == 1) CRESTE_TP4(1); else if (op == 2) CRESTE_TP4(2);
    else if (op == 3) CRESTE_TP4(3); else CRESTE_TP4(4);
#undef CRESTE_TP4
    CRESTE_CHECK_LAUNCH("train_pointwise4");
    return CRESTE_OK;
  }
  train_pointwise_kernel<<<grid1d(P * C), 256, 0, (hipStream_t)stream>>>(op, a, a_cs, b, b_cs, g, g_c, r, o, o_cs, HW, P, C, out_amax);
  CRESTE_CHECK_LAUNCH("train_pointwise");
  return CRESTE_OK;
}

// chunks per sample: SR_CHUNKS, or more when there are few samples (N = 1: the whole batch as one sample -- 256 workgroups of four
// waves streamed a 140 MB tensor pair at 1.35 TB/s) so that about 2048 workgroups are in flight either way
static inline int sr_chunk_cap(int N) {
  const int c = 2048 / (N > 0 ? N : 1);
  return c > SR_CHUNKS ? c : SR_CHUNKS;
}
extern "C" int64_t creste_sample_reduce_workspace_bytes(int N, int C) {
  return N > 0 && C > 0 ? (int64_t)N * sr_chunk_cap(N) * C * 4 : -1;
}

extern "C" int creste_sample_reduce_f32(const float* a, int a_cs, const float* b, int b_cs, float* out, int N,
                                        int64_t HW, int C, float scale, void* work, void* stream) {
  CRESTE_REQUIRE(a && out && work && N > 0 && HW > 0 && C > 0, "sample_reduce: bad args");
  const int rows = 256 / C > 0 ? 256 / C : 1;
  const long per = (HW + rows - 1) / rows;
  const int cap = sr_chunk_cap(N);
  const int chunks = (int)(per < cap ? per : cap);
  const size_t smem = (size_t)rows * (C >= 256 ? 256 : C) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  sample_reduce_kernel<<<dim3(chunks, N), 256, smem, s>>>(a, a_cs, b, b_cs, (float*)work, HW, C);
  CRESTE_CHECK_LAUNCH("sample_reduce");
  sample_reduce_finalize_kernel<<<(N * C + 3) / 4, 256, 0, s>>>((const float*)work, out, chunks, N * C, C, scale);
  CRESTE_CHECK_LAUNCH("sample_reduce_finalize");
  return CRESTE_OK;
}

extern "C" int creste_se_fc_forward_f32(const float* s, const float* w1, const float* b1, const float* w2,
                                        const float* b2, float* hpre, float* hact, float* gate, int N, int C, int Cse,
                                        void* stream) {
  CRESTE_REQUIRE(s && w1 && b1 && w2 && b2 && hpre && hact && gate && N > 0 && C > 0 && Cse > 0,
                 "se_fc_forward: bad args");
  se_fc_forward_kernel<<<dim3(N, (C + SEFC_SLICE - 1) / SEFC_SLICE), SEFC_T, (size_t)(C + Cse) * 4, (hipStream_t)stream>>>(
      s, w1, b1, w2, b2, hpre, hact, gate, C, Cse);
  CRESTE_CHECK_LAUNCH("se_fc_forward");
  return CRESTE_OK;
}

extern "C" int creste_se_fc_backward_f32(const float* gg, const float* gate, const float* hpre, const float* w1,
                                         const float* w2, float* gz, float* ghpre, float* gs, int N, int C, int Cse,
                                         void* stream) {
  CRESTE_REQUIRE(gg && gate && hpre && w1 && w2 && gz && ghpre && gs && N > 0 && C > 0 && Cse > 0,
                 "se_fc_backward: bad args");
  se_fc_backward_kernel<<<dim3(N, (C + SEFC_SLICE - 1) / SEFC_SLICE), SEFC_T, (size_t)(C + Cse) * 4, (hipStream_t)stream>>>(
      gg, gate, hpre, w1, w2, gz, ghpre, gs, C, Cse);
  CRESTE_CHECK_LAUNCH("se_fc_backward");
  return CRESTE_OK;
}

extern "C" int creste_fc_wgrad_f32(const float* go, const float* in, float* gw, float* gb, int N, int O, int I,
                                   int accumulate, void* stream) {
  CRESTE_REQUIRE(go && in && gw && gb && N > 0 && O > 0 && I > 0, "fc_wgrad: bad args");
  fc_wgrad_kernel<<<(O * I + 255) / 256, 256, 0, (hipStream_t)stream>>>(go, in, gw, gb, N, O, I, accumulate);
  CRESTE_CHECK_LAUNCH("fc_wgrad");
  return CRESTE_OK;
}

extern "C" int creste_depth_ce_loss_f32(const float* logits, int cs, const float* gt_mm, int64_t P, int num_bins,
                                        float depth_min, float depth_max, float weight, float* g_logits, int g_cs,
                                        float* out3 /* loss, accuracy, n_valid */, void* work, void* stream) {
  CRESTE_REQUIRE(logits && gt_mm && out3 && work && P > 0, "depth_ce_loss: bad args");
  CRESTE_REQUIRE(num_bins == 128 && cs % 4 == 0 && (!g_logits || g_cs % 4 == 0),
                 "depth_ce_loss: built for 128 depth bins (32 lanes x 4 logits per pixel)");
  hipStream_t s = (hipStream_t)stream;
  float* wk = (float*)work;                       // [1024] count partials | [2048] loss partials | [4] sums
  const float bin_size = (depth_max - depth_min) / (float)num_bins;
  const int b1 = grid1d(P, 1024);
  depth_count_valid_kernel<<<b1, 256, 0, s>>>(gt_mm, P, depth_min, bin_size, num_bins, wk);
  CRESTE_CHECK_LAUNCH("depth_count_valid");
  reduce_small_kernel<<<1, 256, 0, s>>>(wk, out3 + 2, b1, 1);
  CRESTE_CHECK_LAUNCH("depth_count_reduce");
  const int b2 = grid1d(P * 32, 1024);
  depth_ce_kernel<<<b2, 256, 0, s>>>(logits, cs, gt_mm, P, depth_min, bin_size, num_bins, out3 + 2, weight, g_logits, g_cs,
                                     wk + 1024);
  CRESTE_CHECK_LAUNCH("depth_ce");
  reduce_small_kernel<<<1, 256, 0, s>>>(wk + 1024, wk + 1024 + 2048, b2, 2);
  CRESTE_CHECK_LAUNCH("depth_ce_reduce");
  depth_ce_finish_kernel<<<1, 1, 0, s>>>(wk + 1024 + 2048, out3);   // loss_sum / n_valid, hits / n_valid
  CRESTE_CHECK_LAUNCH("depth_ce_finish");
  return CRESTE_OK;
}

extern "C" int64_t creste_loss_workspace_bytes(void) { return (1024 + 2048 + 8) * 4; }

extern "C" int creste_mse_loss_f32(const float* pred, int p_cs, const float* gt, int g_cs, int64_t P, int C,
                                   float weight, float* g_pred, int o_cs, float* out2 /* loss, count */, void* work,
                                   void* stream) {
  CRESTE_REQUIRE(pred && gt && out2 && work && P > 0 && C > 0, "mse_loss: bad args");
  hipStream_t s = (hipStream_t)stream;
  float* wk = (float*)work;
  const int b = grid1d(P * C, 1024);
  mse_partial_kernel<<<b, 256, 0, s>>>(pred, p_cs, gt, g_cs, P, C, wk);
  CRESTE_CHECK_LAUNCH("mse_partial");
  reduce_small_kernel<<<1, 256, 0, s>>>(wk, wk + 2048, b, 2);
  CRESTE_CHECK_LAUNCH("mse_reduce");
  if (g_pred) {
    mse_grad_kernel<<<grid1d(P * C), 256, 0, s>>>(pred, p_cs, gt, g_cs, P, C, wk + 2048, weight, g_pred, o_cs);
    CRESTE_CHECK_LAUNCH("mse_grad");
  }
  mse_finish_kernel<<<1, 1, 0, s>>>(wk + 2048, out2);
  CRESTE_CHECK_LAUNCH("mse_finish");
  return CRESTE_OK;
}

// ------------------------------------------------------------------------------------ BEV splat backward
// Forward (csrc/bev_splat.hip, reference splat_projection.py:262-354): for point p with map coords (X, Y),
//   D[i]     += w_tap            S[c][i] += w_tap * f_c(p)          over its (up to) 4 valid taps i
//   bev[c,i]  = S[c][i] / max(D[i], min_w)         dens[i] = D[i]
// Backward, given g_bev [B,GH,GW,F] and (optionally) g_dens [B,GH,GW]:
//   gD[i]   = g_dens[i] - [D[i] >= min_w] * sum_c g_bev[c,i] * bev[c,i] / D[i]                 (cell pass)
//   gf_c(p) = sum_taps w_tap * g_bev[c,i] / max(D[i], min_w)
//   gw_tap  = sum_c g_bev[c,i] * f_c(p) / max(D[i], min_w) + gD[i]
//   gX = sum_taps gw_tap * (2xd-1) * wy_tap ,  gY = sum_taps gw_tap * wx_tap * (2yd-1)     (floor has no gradient)
//   X = (-y + off_x)/vox_x, Y = (-x + off_y)/vox_y  ->  g_y = -gX / vox_x ,  g_x = -gY / vox_y ,  g_z = 0
// One 32-lane half-wave per point, lanes over channels (a gather: no atomics, deterministic).
namespace creste {

__global__ __launch_bounds__(256) void splat_cell_grad_kernel(const float* __restrict__ g_bev,
                                                              const float* __restrict__ bev,
                                                              const float* __restrict__ dens,
                                                              const float* __restrict__ g_dens,
                                                              float* __restrict__ gD, long ncell, int F, float min_w) {
  const int sub = threadIdx.x & 31;
  const long half = (blockIdx.x * 256L + threadIdx.x) >> 5, nhalf = ((long)gridDim.x * 256) >> 5;
  for (long i = half; i < ncell; i += nhalf) {
    const float D = dens[i];
    float s = 0.f;
    if (D >= min_w)                               // torch.clamp(min=) passes the gradient at equality
      for (int c = sub; c < F; c += 32) s += g_bev[i * F + c] * bev[i * F + c];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (sub == 0) gD[i] = (g_dens ? g_dens[i] : 0.f) - (D >= min_w ? s / D : 0.f);
  }
}

// MODE 0 'mean' (bev = sum / max(D, min_w)), 1 'sum', 2 'max' (bev = max(0, max over taps and points of w*f): the
// cotangent of a cell/channel goes to the entries that ATTAIN a positive maximum -- recomputed with the forward's own
// arithmetic, w*f == bev bit for bit; exact ties and a maximum of 0 are measure-zero cases whose sub-gradient the
// reference leaves to torch_scatter's argmax / torch.maximum's tie rule).
template <int MODE>
__global__ __launch_bounds__(256) void splat_point_grad_kernel(
    const float* __restrict__ coords, const float* __restrict__ feats, int feats_cs, const float* __restrict__ g_bev,
    const float* __restrict__ bev, const float* __restrict__ dens, const float* __restrict__ gD,
    float* __restrict__ g_feats, int gf_cs,
    float* __restrict__ g_xyz, long BP, int P, int F, int GH, int GW, float vox_x, float vox_y, float min_w) {
  const int sub = threadIdx.x & 31;
  const long half = (blockIdx.x * 256L + threadIdx.x) >> 5, nhalf = ((long)gridDim.x * 256) >> 5;
  for (long p = half; p < BP; p += nhalf) {
    const long b = p / P;
    const float X = coords[p * 2], Y = coords[p * 2 + 1];
    const float fx = floorf(X), fy = floorf(Y);
    const float rX = X - fx, rY = Y - fy;
    const int X0 = (int)fx, Y0 = (int)fy;
    float gX = 0.f, gY = 0.f;
    float gf[8];                                   // F <= 256: up to 8 channels per lane
#pragma unroll
    for (int j = 0; j < 8; ++j) gf[j] = 0.f;
    const bool near = fx >= -1.f && fx <= (float)(GW - 1) && fy >= -1.f && fy <= (float)(GH - 1);
    if (near) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int xd = t >> 1, yd = t & 1;           // reference tap order (0,0),(0,1),(1,0),(1,1)
        const int xi = X0 + xd, yi = Y0 + yd;
        if ((unsigned)xi >= (unsigned)GW || (unsigned)yi >= (unsigned)GH) continue;
        const float wx = xd ? rX : __fsub_rn(1.f, rX), wy = yd ? rY : __fsub_rn(1.f, rY);
        const float w = __fmul_rn(wx, wy);
        const long cell = (b * GH + yi) * GW + xi;
        const float inv = MODE == 0 ? 1.f / fmaxf(dens[cell], min_w) : 1.f;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = sub + 32 * j;
          if (c < F) {
            const float g = g_bev[cell * F + c];
            const float f = feats[p * feats_cs + c];
            if (MODE == 2) {
              const float v = __fmul_rn(w, f);
              if (v > 0.f && v == bev[cell * F + c]) { gf[j] += w * g; dot += g * f; }
            } else {
              gf[j] += w * g * inv;
              dot += g * f;
            }
          }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        const float gw = dot * inv + gD[cell];
        gX += gw * (xd ? 1.f : -1.f) * wy;
        gY += gw * wx * (yd ? 1.f : -1.f);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = sub + 32 * j;
      if (c < F) g_feats[p * gf_cs + c] = gf[j];
    }
    if (sub == 0 && g_xyz) {
      g_xyz[p * 3 + 0] = -gY / vox_y;
      g_xyz[p * 3 + 1] = -gX / vox_x;
      g_xyz[p * 3 + 2] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void splat_dens_grad_kernel(const float* __restrict__ g_dens, float* __restrict__ gD, long ncell) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < ncell; i += (long)gridDim.x * 256) gD[i] = g_dens ? g_dens[i] : 0.f;
}

// depth = sum_k softmax(logits)_k * bins_k / 1000  ->  g_logits_k (+)= g_depth * p_k * (bins_k/1000 - depth)
__global__ __launch_bounds__(256) void depth_expectation_bwd_kernel(const float* __restrict__ logits, int cs, long P,
                                                                    const float* __restrict__ bin_values,
                                                                    const float* __restrict__ g_depth,
                                                                    float* __restrict__ g_logits, int g_cs,
                                                                    int accumulate) {
  const int sub = threadIdx.x & 31;
  const long half = (blockIdx.x * 256L + threadIdx.x) >> 5, nhalf = ((long)gridDim.x * 256) >> 5;
  const f32x4 bv = ld4(bin_values + sub * 4);
  for (long p = half; p < P; p += nhalf) {
    const f32x4 x = ld4(logits + p * cs + sub * 4);
    float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    f32x4 e;
    float se = 0.f, sw = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { e[j] = expf(x[j] - mx); se += e[j]; sw += e[j] * bv[j]; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { se += __shfl_xor(se, o); sw += __shfl_xor(sw, o); }
    const float depth = (sw / se) / 1000.f, gd = g_depth[p];
    f32x4 g = accumulate ? ld4(g_logits + p * g_cs + sub * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] += gd * (e[j] / se) * (bv[j] / 1000.f - depth);
    st4(g_logits + p * g_cs + sub * 4, g);
  }
}

}  // namespace creste

extern "C" int creste_bev_splat_mode_bwd_f32(const float* coords, const float* feats, int feats_cs, const float* g_bev,
                                             const float* g_dens, const float* bev, const float* dens, int B, int P, int F,
                                             int GH, int GW, float vox_x, float vox_y, float min_weight, int mode,
                                             float* g_feats, int gf_cs, float* g_xyz, float* cell_work, void* stream) {
  CRESTE_REQUIRE(coords && feats && g_bev && bev && dens && g_feats && cell_work, "bev_splat_bwd: null pointer");
  CRESTE_REQUIRE(B > 0 && P > 0 && F > 0 && F <= 256 && GH > 0 && GW > 0 && vox_x > 0.f && vox_y > 0.f,
                 "bev_splat_bwd: bad dims (F <= 256)");
  CRESTE_REQUIRE(mode >= 0 && mode <= 2, "bev_splat_bwd: unknown scatter mode %d", mode);
  hipStream_t s = (hipStream_t)stream;
  const long ncell = (long)B * GH * GW, BP = (long)B * P;
  if (mode == 0) splat_cell_grad_kernel<<<grid1d(ncell * 32), 256, 0, s>>>(g_bev, bev, dens, g_dens, cell_work, ncell, F, min_weight);
  else splat_dens_grad_kernel<<<grid1d(ncell), 256, 0, s>>>(g_dens, cell_work, ncell);
  CRESTE_CHECK_LAUNCH("splat_cell_grad");
#define CRESTE_SPLAT_BWD(M)                                                                                              \
  splat_point_grad_kernel<M><<<grid1d(BP * 32), 256, 0, s>>>(coords, feats, feats_cs, g_bev, bev, dens, cell_work, g_feats, \
                                                            gf_cs, g_xyz, BP, P, F, GH, GW, vox_x, vox_y, min_weight)
  if (mode == 0) CRESTE_SPLAT_BWD(0); else if (mode == 1) CRESTE_SPLAT_BWD(1); else CRESTE_SPLAT_BWD(2);
#undef CRESTE_SPLAT_BWD
  CRESTE_CHECK_LAUNCH("splat_point_grad");
  return CRESTE_OK;
}

extern "C" int creste_bev_splat_bwd_f32(const float* coords, const float* feats, int feats_cs, const float* g_bev,
                                        const float* g_dens, const float* bev, const float* dens, int B, int P, int F,
                                        int GH, int GW, float vox_x, float vox_y, float min_weight, float* g_feats,
                                        int gf_cs, float* g_xyz, float* cell_work, void* stream) {
  return creste_bev_splat_mode_bwd_f32(coords, feats, feats_cs, g_bev, g_dens, bev, dens, B, P, F, GH, GW, vox_x, vox_y,
                                       min_weight, 0, g_feats, gf_cs, g_xyz, cell_work, stream);
}

extern "C" int creste_depth_expectation_bwd_f32(const float* logits, int cs, int64_t P, int C, const float* bin_values,
                                                const float* g_depth, float* g_logits, int g_cs, int accumulate,
                                                void* stream) {
  CRESTE_REQUIRE(logits && bin_values && g_depth && g_logits && P > 0, "depth_expectation_bwd: null pointer");
  CRESTE_REQUIRE(C == 128 && cs % 4 == 0 && g_cs % 4 == 0, "depth_expectation_bwd: built for 128 bins");
  depth_expectation_bwd_kernel<<<grid1d(P * 32), 256, 0, (hipStream_t)stream>>>(logits, cs, P, bin_values, g_depth,
                                                                                g_logits, g_cs, accumulate);
  CRESTE_CHECK_LAUNCH("depth_expectation_bwd");
  return CRESTE_OK;
}

// ------------------------------------------------------------------------------------ strided dgrad helper
// gz[n, s*oy, s*ox, c] = gy[n, oy, ox, c], zeros elsewhere ([N,Hz,Wz,C], Hz = (Ho-1)*s+1): the input gradient of a
// stride-s conv is the stride-1 conv of this zero-inserted cotangent with the flipped, channel-transposed kernel.
namespace creste {
__global__ __launch_bounds__(256) void zero_insert_kernel(const float* __restrict__ gy, int gy_cs, float* __restrict__ gz,
                                                          int N, int Ho, int Wo, int C, int s, int Hz, int Wz) {
  const int cq = C >> 2;
  const long total = (long)N * Hz * Wz * cq;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % cq) * 4;
    long t = i / cq;
    const int x = (int)(t % Wz); t /= Wz;
    const int y = (int)(t % Hz);
    const int n = (int)(t / Hz);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y % s == 0 && x % s == 0) v = ld4(gy + (((long)n * Ho + y / s) * Wo + x / s) * gy_cs + c);
    st4(gz + i * 4, v);
  }
}
}  // namespace creste

extern "C" int creste_zero_insert_nhwc_f32(const float* gy, int gy_cs, float* gz, int N, int Ho, int Wo, int C,
                                           int stride, void* stream) {
  CRESTE_REQUIRE(gy && gz && N > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 && gy_cs % 4 == 0 && stride > 0,
                 "zero_insert: bad args (C % 4)");
  const int Hz = (Ho - 1) * stride + 1, Wz = (Wo - 1) * stride + 1;
  zero_insert_kernel<<<grid1d((long)N * Hz * Wz * (C / 4)), 256, 0, (hipStream_t)stream>>>(gy, gy_cs, gz, N, Ho, Wo, C,
                                                                                          stride, Hz, Wz);
  CRESTE_CHECK_LAUNCH("zero_insert");
  return CRESTE_OK;
}

// ------------------------------------------------------------------------------------ pixel geometry backward
// Forward (csrc/pointwise.hip pixel_geometry_kernel; reference splat_projection.py:19-51,98-104,152-157):
//   c = [u*d, v*d, d, 1] ; xyz_k = P[k] . c ; z = xyz_2 ; h = relu(w1*z + b1) ; zf = relu(W2 h + b2)
// Backward per pixel, given g_xyz [3] and g_zf [zdim] (a slice of the fusion conv's input gradient):
//   gq = g_zf * (zf > 0) ; gh = W2^T gq ; ghp = gh * (h > 0) ; g_z = w1 . ghp + g_xyz_2
//   g_d = sum_k g_xyz'_k * (P[k][0]*u + P[k][1]*v + P[k][2])          (g_xyz' = g_xyz with g_z in component 2)
// gq [P][zdim], ghp [P][zhid], h [P][zhid] and z [P] are written out: the four parameter gradients are then plain
// reductions over pixels (W2: gq^T h, w1: ghp^T z via the 1x1 wgrad kernel; biases: channel sums).
namespace creste {
__global__ __launch_bounds__(256) void pixel_geometry_bwd_kernel(
    const float* __restrict__ depth, const float* __restrict__ p2p, int B, int Hs, int Ws, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, int zhid, int zdim,
    const float* __restrict__ g_xyz, const float* __restrict__ g_zf, int gz_cs, float* __restrict__ g_depth,
    float* __restrict__ gq, float* __restrict__ ghp, float* __restrict__ hbuf, float* __restrict__ zbuf) {
  extern __shared__ float smw[];   // w1[zhid] b1[zhid] w2[zdim*zhid] b2[zdim]
  float* s_w1 = smw; float* s_b1 = s_w1 + zhid; float* s_w2 = s_b1 + zhid; float* s_b2 = s_w2 + zdim * zhid;
  for (int i = threadIdx.x; i < zhid; i += blockDim.x) { s_w1[i] = w1[i]; s_b1[i] = b1[i]; }
  for (int i = threadIdx.x; i < zdim * zhid; i += blockDim.x) s_w2[i] = w2[i];      // [zdim][zhid] row-major
  for (int i = threadIdx.x; i < zdim; i += blockDim.x) s_b2[i] = b2[i];
  __syncthreads();
  const long P = (long)Hs * Ws, total = (long)B * P;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const int b = (int)(g / P); const long p = g % P;
    const int v = (int)(p / Ws), u = (int)(p % Ws);
    const float d = depth[g];
    const float* M = p2p + (long)b * 16;
    const float z = __fmaf_rn(M[11], 1.0f, __fmaf_rn(M[10], d, __fmaf_rn(M[9], (float)v * d, __fmul_rn(M[8], (float)u * d))));
    zbuf[g] = z;
    // hidden layer
    float gz = 0.f;
    for (int h = 0; h < zhid; ++h) hbuf[g * zhid + h] = fmaxf(__fmaf_rn(s_w1[h], z, s_b1[h]), 0.f);
    for (int h = 0; h < zhid; ++h) ghp[g * zhid + h] = 0.f;
    for (int j = 0; j < zdim; ++j) {
      float s = s_b2[j];
      for (int h = 0; h < zhid; ++h) s = __fmaf_rn(s_w2[j * zhid + h], hbuf[g * zhid + h], s);
      const float q = s > 0.f ? g_zf[g * gz_cs + j] : 0.f;
      gq[g * zdim + j] = q;
      if (q != 0.f)
        for (int h = 0; h < zhid; ++h) ghp[g * zhid + h] += s_w2[j * zhid + h] * q;
    }
    for (int h = 0; h < zhid; ++h) {
      const float t = hbuf[g * zhid + h] > 0.f ? ghp[g * zhid + h] : 0.f;
      ghp[g * zhid + h] = t;
      gz += s_w1[h] * t;
    }
    const float gx0 = g_xyz[g * 3], gx1 = g_xyz[g * 3 + 1], gx2 = g_xyz[g * 3 + 2] + gz;
    const float fu = (float)u, fv = (float)v;
    g_depth[g] = gx0 * (M[0] * fu + M[1] * fv + M[2]) + gx1 * (M[4] * fu + M[5] * fv + M[6]) +
                 gx2 * (M[8] * fu + M[9] * fv + M[10]);
  }
}

// The shipped z-MLP (1 -> 64 -> 32): one 32-lane half-wave per pixel.  Lane j owns output j in the forward recompute
// (every lane evaluates the 64 hidden units itself: 128 FMAs, no exchange) and hidden units {j, j+32} in the
// backward (the 32 masked output cotangents arrive by shuffle); W2 sits in LDS in both orientations so either
// access pattern is conflict-free; all stores are coalesced rows.  15 ms -> well under 1 ms per SSC step.
__global__ __launch_bounds__(256) void pixel_geometry_bwd32_kernel(
    const float* __restrict__ depth, const float* __restrict__ p2p, int B, int Hs, int Ws, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    const float* __restrict__ g_xyz, const float* __restrict__ g_zf, int gz_cs, float* __restrict__ g_depth,
    float* __restrict__ gq, float* __restrict__ ghp, float* __restrict__ hbuf, float* __restrict__ zbuf) {
  constexpr int ZH = 64, ZD = 32;
  __shared__ float s_w1[ZH], s_b1[ZH], s_b2[ZD], s_w2[ZD * ZH], s_w2t[ZH * ZD];
  for (int i = threadIdx.x; i < ZH; i += 256) { s_w1[i] = w1[i]; s_b1[i] = b1[i]; }
  for (int i = threadIdx.x; i < ZD; i += 256) s_b2[i] = b2[i];
  for (int i = threadIdx.x; i < ZD * ZH; i += 256) {
    s_w2[i] = w2[i];                                   // [j][h]
    s_w2t[(i % ZH) * ZD + i / ZH] = w2[i];             // [h][j]
  }
  __syncthreads();
  const int sub = threadIdx.x & 31;
  const long P = (long)Hs * Ws, total = (long)B * P;
  const long half = (blockIdx.x * 256L + threadIdx.x) >> 5, nhalf = ((long)gridDim.x * 256) >> 5;
  for (long g = half; g < total; g += nhalf) {
    const int b = (int)(g / P); const long p = g % P;
    const int v = (int)(p / Ws), u = (int)(p % Ws);
    const float d = depth[g];
    const float* M = p2p + (long)b * 16;
    const float z = __fmaf_rn(M[11], 1.0f, __fmaf_rn(M[10], d, __fmaf_rn(M[9], (float)v * d, __fmul_rn(M[8], (float)u * d))));
    float s = s_b2[sub];
#pragma unroll 8
    for (int h = 0; h < ZH; ++h) {
      const float hv = fmaxf(__fmaf_rn(s_w1[h], z, s_b1[h]), 0.f);
      s = __fmaf_rn(s_w2t[h * ZD + sub], hv, s);
    }
    const float q = s > 0.f ? g_zf[g * gz_cs + sub] : 0.f;
    gq[g * ZD + sub] = q;
    const float h0 = fmaxf(__fmaf_rn(s_w1[sub], z, s_b1[sub]), 0.f);
    const float h1 = fmaxf(__fmaf_rn(s_w1[sub + 32], z, s_b1[sub + 32]), 0.f);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
    for (int j = 0; j < ZD; ++j) {
      const float qj = __shfl(q, j, 32);
      a0 += s_w2[j * ZH + sub] * qj;
      a1 += s_w2[j * ZH + sub + 32] * qj;
    }
    a0 = h0 > 0.f ? a0 : 0.f;
    a1 = h1 > 0.f ? a1 : 0.f;
    hbuf[g * ZH + sub] = h0; hbuf[g * ZH + sub + 32] = h1;
    ghp[g * ZH + sub] = a0; ghp[g * ZH + sub + 32] = a1;
    float gz = s_w1[sub] * a0 + s_w1[sub + 32] * a1;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) gz += __shfl_xor(gz, o);
    if (sub == 0) {
      zbuf[g] = z;
      const float gx0 = g_xyz[g * 3], gx1 = g_xyz[g * 3 + 1], gx2 = g_xyz[g * 3 + 2] + gz;
      const float fu = (float)u, fv = (float)v;
      g_depth[g] = gx0 * (M[0] * fu + M[1] * fv + M[2]) + gx1 * (M[4] * fu + M[5] * fv + M[6]) +
                   gx2 * (M[8] * fu + M[9] * fv + M[10]);
    }
  }
}
}  // namespace creste

extern "C" int creste_pixel_geometry_bwd_f32(const float* depth, const float* p2p, int B, int Hs, int Ws, const float* w1,
                                             const float* b1, const float* w2, const float* b2, int zhid, int zdim,
                                             const float* g_xyz, const float* g_zf, int gz_cs, float* g_depth, float* gq,
                                             float* ghp, float* hbuf, float* zbuf, void* stream) {
  CRESTE_REQUIRE(depth && p2p && w1 && b1 && w2 && b2 && g_xyz && g_zf && g_depth && gq && ghp && hbuf && zbuf,
                 "pixel_geometry_bwd: null pointer");
  CRESTE_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && zhid > 0 && zdim > 0, "pixel_geometry_bwd: bad dims");
  if (zhid == 64 && zdim == 32) {
    pixel_geometry_bwd32_kernel<<<grid1d((long)B * Hs * Ws * 32), 256, 0, (hipStream_t)stream>>>(
        depth, p2p, B, Hs, Ws, w1, b1, w2, b2, g_xyz, g_zf, gz_cs, g_depth, gq, ghp, hbuf, zbuf);
    CRESTE_CHECK_LAUNCH("pixel_geometry_bwd32");
    return CRESTE_OK;
  }
  const size_t smem = (size_t)(2 * zhid + zdim * zhid + zdim) * sizeof(float);
  pixel_geometry_bwd_kernel<<<grid1d((long)B * Hs * Ws), 256, smem, (hipStream_t)stream>>>(
      depth, p2p, B, Hs, Ws, w1, b1, w2, b2, zhid, zdim, g_xyz, g_zf, gz_cs, g_depth, gq, ghp, hbuf, zbuf);
  CRESTE_CHECK_LAUNCH("pixel_geometry_bwd");
  return CRESTE_OK;
}
