// Training-mode primitives of the IRL reward network (MultiScaleFCN, reference conv.py:88-161) and the
// pieces of its backward the forward engines cannot provide:
//   * weight gradient of a stride-1 conv on the fp32 MFMA (a GEMM whose reduction runs over pixels);
//   * training-mode BatchNorm as (per-channel moments) + (elementwise apply), in three forms:
//       forward           y  = g*xh + b                        xh = (x - mean)*invstd
//       tangent (JVP)     yd = g*invstd*(xd - m(xd) - xh*m(xh*xd))
//       joint backward    cotangents (gy, gyd) of (y, yd) -> (gx, gxd, g_gamma, g_beta)
//     The tangent/joint pair is what the IRL gradient penalty needs (reference loss_utils.py:1207-1217):
//       d/dtheta <u, grad_x R(x; theta)> = d/dtheta JVP_x R(x; theta)[u]
//     i.e. one tangent forward with xd = u and one backward through the (primal, tangent) graph -- no
//     generic double-backward machinery;
//   * ReLU, 2x2 max-pool with argmax, add, and the transpose of the bilinear x2 upsample.
// Everything is NHWC fp32 with an explicit pixel stride (`cs`) so channel slices of a concat buffer are
// read and written in place.  All reductions run in a fixed order (block partials, then a serial sum):
// run-to-run deterministic.
#include "common.h"

namespace creste {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------ conv wgrad
// partial[chunk][tap][co][ci] = sum over the chunk's pixels of gy[p][co] * x[p + tap][ci]
// One wave = one 32(co) x 32(ci) tile of one tap; v_mfma_f32_32x32x2_f32 consumes two pixels per issue with
// both operands read straight from global memory (lane = channel, so the loads are coalesced rows of the
// NHWC tensors; each element is used exactly once per tile -> nothing to stage in LDS).
constexpr int WG_PIX = 16;  // pixels per unrolled step (8 MFMAs, 16 independent loads in flight per lane)

__global__ __launch_bounds__(256) void wgrad_partial_kernel(const float* __restrict__ x, int x_cs,
                                                            const float* __restrict__ gy, int gy_cs,
                                                            float* __restrict__ partial, int N, int H, int W,
                                                            int Cin, int Cout, int K, int pad, int chunk_px) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int chunk = blockIdx.x, tap = blockIdx.y;
  const int ky = tap / K - pad, kx = tap % K - pad;
  const int tco = (Cout + 31) / 32, tci = (Cin + 31) / 32;
  const int M = N * H * W;
  const int p0 = chunk * chunk_px, p1 = min(M, p0 + chunk_px);
  for (int t = wave; t < tco * tci; t += 4) {
    const int co = (t / tci) * 32 + li, ci = (t % tci) * 32 + li;
    const bool co_ok = co < Cout, ci_ok = ci < Cin;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int p = p0; p < p1; p += WG_PIX) {
      float a[WG_PIX / 2], b[WG_PIX / 2];
#pragma unroll
      for (int j = 0; j < WG_PIX / 2; ++j) {
        const int q = p + 2 * j + lh;
        a[j] = 0.f; b[j] = 0.f;
        if (q < p1) {
          const int rowi = q / W;
          const int xx = q - rowi * W;
          const int yy = rowi % H;
          const int iy = yy + ky, ix = xx + kx;
          if (co_ok) a[j] = gy[(long)q * gy_cs + co];
          if (ci_ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            b[j] = x[(long)(q + ky * W + kx) * x_cs + ci];
        }
      }
#pragma unroll
      for (int j = 0; j < WG_PIX / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
    }
    // D layout: col (ci) = lane & 31, row (co) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* out = partial + ((size_t)chunk * K * K + tap) * Cout * Cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (t / tci) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < Cout && ci_ok) out[(size_t)row * Cin + ci] = acc[r];
    }
  }
}

// ------------------------------------------------------------------------- row-walk wgrad (K = 1, 3, 5)
// The per-tap kernel above reads both operands K*K times (one workgroup per tap): at 5x5 on a 256x256 grid that is
// 11 GB of L2 traffic per call and the matrix cores wait on it.  Here a workgroup walks a band of rows of one
// 64-pixel column segment with the last K rows of x (plus the K-1 halo columns) in an LDS ring and the current gy row
// beside it: every element is fetched once, and all K*K taps are formed from the ring.  A wave owns one
// 16(co) x 16(ci) tile for NTAP taps (v_mfma_f32_16x16x4_f32: 4 pixels per issue; 40 channels pad to 48, not 64).
// LDS pixel strides are = 16 mod 32 floats, so the four 16-lane pixel groups of an operand read hit distinct banks.
// Workgroups are sized small (a whole kernel per wave where the accumulators allow) so that several share a CU and one's
// row hand-over overlaps another's matrix phase; the partial sets (one per workgroup) go to wgrad_reduce4_kernel.
constexpr int WR_TW = 64;   // pixels per column segment
constexpr int WR_XI = 3;    // float4 prefetch registers per thread: x row
constexpr int WR_XI1 = 4;   //   1x1 convs (two-wave workgroups)
constexpr int WR_GI = 2;    //                                      gy row

struct WrArgs {
  const float* x;
  const float* gy;
  float* partial;
  int x_cs, gy_cs, N, H, W, Cin, Cout;
  int cpx, cpy;          // LDS pixel strides (floats)
  int tci, pairs;        // ci tiles, (co, ci) tile pairs
  int nseg, nband, band_rows;
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

// CPX > 0: NTAP == K*K and the x stride is the compile-time CPX, so every tap is an immediate offset from one of K row
// pointers (25 hoisted per-tap addresses would not fit beside 100 accumulator registers).
template <int K, int NTAP, int MAXT, int CPX>
__global__ __launch_bounds__(MAXT) void wgrad_rows_kernel(const WrArgs a) {
  constexpr int PAD = K / 2, XW = WR_TW + K - 1, KK = K * K;
  static_assert(CPX == 0 || (NTAP == KK && K > 1), "immediate tap offsets need the whole kernel in one wave");
  extern __shared__ __attribute__((aligned(16))) float wr_lds[];
  float* xs = wr_lds;                               // [K][XW][cpx]
  float* gs = wr_lds + K * XW * a.cpx;              // [TW][cpy]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int seg = b % a.nseg; b /= a.nseg;
  const int band = b % a.nband;
  const int n = b / a.nband;
  const int x0 = seg * WR_TW;
  const int y0 = band * a.band_rows, y1 = min(a.H, y0 + a.band_rows);
  const int cq_x = a.Cin >> 2, cq_g = a.Cout >> 2;
  const int n_xi = XW * cq_x, n_gi = WR_TW * cq_g;
  const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs;
  const float* gn = a.gy + (size_t)n * a.H * a.W * a.gy_cs;

  constexpr int XI = K == 1 ? WR_XI1 : WR_XI;
  f32x4v px[XI], pg[WR_GI];
  auto fetch_x = [&](int iy) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int it = tid + i * nt;
      f32x4v v = {0.f, 0.f, 0.f, 0.f};
      if (it < n_xi) {
        const int j = it / cq_x, c = (it - j * cq_x) * 4;
        const int ix = x0 - PAD + j;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          v = *reinterpret_cast<const f32x4v*>(xn + ((size_t)iy * a.W + ix) * a.x_cs + c);
      }
      px[i] = v;
    }
  };
  auto store_x = [&](int iy) __attribute__((always_inline)) {
    float* dst = xs + ((iy + K) % K) * XW * a.cpx;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int it = tid + i * nt;
      if (it < n_xi) {
        const int j = it / cq_x, c = (it - j * cq_x) * 4;
        *reinterpret_cast<f32x4v*>(dst + j * a.cpx + c) = px[i];
      }
    }
  };
  auto fetch_g = [&](int y) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WR_GI; ++i) {
      const int it = tid + i * nt;
      f32x4v v = {0.f, 0.f, 0.f, 0.f};
      if (it < n_gi) {
        const int j = it / cq_g, c = (it - j * cq_g) * 4;
        if (y < y1 && x0 + j < a.W) v = *reinterpret_cast<const f32x4v*>(gn + ((size_t)y * a.W + x0 + j) * a.gy_cs + c);
      }
      pg[i] = v;
    }
  };
  auto store_g = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WR_GI; ++i) {
      const int it = tid + i * nt;
      if (it < n_gi) {
        const int j = it / cq_g, c = (it - j * cq_g) * 4;
        *reinterpret_cast<f32x4v*>(gs + j * a.cpy + c) = pg[i];
      }
    }
  };

  // this wave's tile and taps
  const int pair = wave % a.pairs, tap0 = (wave / a.pairs) * NTAP;
  const int co0 = (pair / a.tci) * 16, ci0 = (pair % a.tci) * 16;
  const int lr = lane & 15, lj = lane >> 4;
  const int a_off = lj * a.cpy + min(co0 + lr, a.Cout - 1);      // clamped lanes feed rows / columns nobody stores
  const int b_off = lj * a.cpx + min(ci0 + lr, a.Cin - 1);
  f32x4v acc[NTAP];
#pragma unroll
  for (int i = 0; i < NTAP; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};

  for (int iy = y0 - PAD; iy < y0 + PAD; ++iy) {     // rows y0-PAD .. y0+PAD-1 of the ring
    fetch_x(iy);
    store_x(iy);
  }
  fetch_x(y0 + PAD);
  fetch_g(y0);
  for (int y = y0; y < y1; ++y) {
    store_x(y + PAD);
    store_g();
    __syncthreads();
    fetch_x(y + 1 + PAD);
    fetch_g(y + 1);
    if constexpr (CPX > 0) {
      const float* xr[K];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) xr[ky] = xs + b_off + ((y + ky - PAD + K) % K) * (XW * CPX);
#pragma unroll 2
      for (int q = 0; q < WR_TW; q += 4) {
        const float av = gs[a_off + q * a.cpy];
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
            acc[ky * K + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xr[ky][(q + kx) * CPX], acc[ky * K + kx], 0, 0, 0);
      }
    } else {
      int off[NTAP];
#pragma unroll
      for (int i = 0; i < NTAP; ++i) {
        const int tap = tap0 + i;
        const int ky = tap / K, kx = tap - ky * K;
        off[i] = __builtin_amdgcn_readfirstlane((((y + ky - PAD + K) % K) * XW + kx) * a.cpx);
      }
#pragma unroll 2
      for (int q = 0; q < WR_TW; q += 4) {
        const float av = gs[a_off + q * a.cpy];
        const float* xb = xs + b_off + q * a.cpx;
#pragma unroll
        for (int i = 0; i < NTAP; ++i)
          if (tap0 + i < KK) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xb[off[i]], acc[i], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // D layout of the 16x16x4 MFMA: col (ci) = lane & 15, row (co) = 4 * (lane >> 4) + r
  float* out = a.partial + (size_t)blockIdx.x * KK * a.Cout * a.Cin;
  const int ci = ci0 + lr;
#pragma unroll
  for (int i = 0; i < NTAP; ++i) {
    const int tap = tap0 + i;
    if (tap < KK && ci < a.Cin) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + 4 * lj + r;
        if (co < a.Cout) out[((size_t)tap * a.Cout + co) * a.Cin + ci] = acc[i][r];
      }
    }
  }
}

struct WrPlan {
  bool ok;
  int ntap, nw, tci, pairs, cpx, cpy, nseg, nband, band_rows, nwg;
  size_t smem;
};

static inline int wr_stride(int c) {   // smallest stride >= c that is 16 mod 32 floats
  int s = (c + 15) / 16 * 16;
  return (s % 32 == 16) ? s : s + 16;
}

static WrPlan wr_plan(const float* x, int x_cs, const float* gy, int gy_cs, int N, int H, int W, int Cin, int Cout, int K) {
  WrPlan p{};
  if ((K != 1 && K != 3 && K != 5) || Cin % 4 || Cout % 4 || x_cs % 4 || gy_cs % 4 || Cin < 8 || Cout < 8) return p;
  if (x && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15)) return p;
  const int tco = (Cout + 15) / 16;
  p.tci = (Cin + 15) / 16;
  p.pairs = tco * p.tci;
  if (p.pairs > 16) return p;
  // fewest waves per workgroup first (whole kernel per wave): small workgroups share a CU, so one's row hand-over
  // (barriers, LDS stores) overlaps another's matrix phase
  static const int opts5[] = {25, 13, 7}, opts3[] = {9, 5, 3}, opts1[] = {1, 1, 1};
  const int* opts = K == 5 ? opts5 : (K == 3 ? opts3 : opts1);
  const int xi = K == 1 ? WR_XI1 : WR_XI;
  const int XW = WR_TW + K - 1;
  p.cpx = wr_stride(Cin);
  p.cpy = wr_stride(Cout);
  p.ntap = 0;
  for (int i = 0; i < 3; ++i) {
    const int ts = (K * K + opts[i] - 1) / opts[i];
    const int cap = opts[i] >= 13 ? 12 : 16;          // 52 / 100 accumulator registers: 768 threads at most
    const int nt = p.pairs * ts * 64;
    if (p.pairs * ts > cap || XW * (Cin / 4) > xi * nt || WR_TW * (Cout / 4) > WR_GI * nt) continue;
    if (K > 1 && opts[i] == K * K && p.cpx != 48 && p.cpx != 80) continue;   // whole-kernel waves are built for these strides
    p.ntap = opts[i];
    p.nw = p.pairs * ts;
    break;
  }
  if (!p.ntap) return p;
  p.smem = ((size_t)K * XW * p.cpx + (size_t)WR_TW * p.cpy) * 4;
  if (p.smem > 160 * 1024) return p;
  p.nseg = (W + WR_TW - 1) / WR_TW;
  int per_cu = (int)((160 * 1024) / p.smem);
  per_cu = per_cu < 16 / p.nw ? per_cu : 16 / p.nw;
  per_cu = per_cu < 1 ? 1 : per_cu;
  int nb = 256 * per_cu / (N * p.nseg);
  nb = nb < 1 ? 1 : (nb > H ? H : nb);
  p.band_rows = (H + nb - 1) / nb;
  p.nband = (H + p.band_rows - 1) / p.band_rows;
  p.nwg = N * p.nseg * p.nband;
  p.ok = true;
  return p;
}

// gw[co][ci][ky][kx] (torch OIHW) = (accumulate ? gw : 0) + sum_chunk partial[chunk][tap][co][ci]
// threads follow the partial layout [tap][co][ci] (coalesced reads of every chunk); 64 elements x 4 chunk groups per
// block (a 1x1 conv has ~1000 elements and 512 partial sets), the four group sums added in a fixed order
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ partial, float* __restrict__ gw,
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

// 1x1 convs with fewer than eight output channels (the reward head 48 -> 1, the 2- and 6-class BEV projections): a
// streaming reduction, not a matrix product --
// thread = (pixel lane, input-channel quad), float4 loads of x, the pixel lanes of a block summed through LDS in a fixed
// order; partial[block][co][ci] goes to wgrad_reduce4_kernel like every other partial set
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const float* __restrict__ x, int x_cs, const float* __restrict__ gy,
                                                         int gy_cs, float* __restrict__ partial, long P, int Cin, int Cout) {
  __shared__ f32x4v red[256];
  const int cq = Cin >> 2, pl_n = 256 / cq;
  const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
  const bool active = pl < pl_n;
  f32x4v acc[7];
#pragma unroll
  for (int co = 0; co < 7; ++co) acc[co] = f32x4v{0.f, 0.f, 0.f, 0.f};
  if (active) {
#pragma unroll 2
    for (long p = (long)blockIdx.x * pl_n + pl; p < P; p += (long)gridDim.x * pl_n) {
      const f32x4v xv = *reinterpret_cast<const f32x4v*>(x + p * x_cs + q * 4);
#pragma unroll
      for (int co = 0; co < 7; ++co)
        if (co < Cout) acc[co] += xv * gy[p * gy_cs + co];
    }
  }
#pragma unroll
  for (int co = 0; co < 7; ++co) {
    if (co >= Cout) break;
    __syncthreads();
    red[threadIdx.x] = acc[co];
    __syncthreads();
    if ((int)threadIdx.x < cq) {
      f32x4v t = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < pl_n; ++r) t += red[r * cq + threadIdx.x];
      *reinterpret_cast<f32x4v*>(partial + ((size_t)blockIdx.x * Cout + co) * Cin + threadIdx.x * 4) = t;
    }
  }
}

// OIHW -> dgrad weight: w'[ci][co][K-1-ky][K-1-kx] = w[co][ci][ky][kx] (a stride-1 conv's input gradient is
// the conv of gy with the flipped, channel-transposed kernel, pad K-1-pad); Cout padded with zero input
// channels up to cout_pad (the conv engine wants Cin % 4 == 0).
__global__ void flip_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int K,
                                   int cout_pad) {
  const int total = Cin * cout_pad * K * K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int kx = i % K, ky = (i / K) % K, co = (i / (K * K)) % cout_pad, ci = i / (K * K * cout_pad);
    wt[i] = co < Cout ? w[(((size_t)co * Cin + ci) * K + (K - 1 - ky)) * K + (K - 1 - kx)] : 0.f;
  }
}

// ------------------------------------------------------------------------------------ channel moments
// out[k][c] = (1/P) * sum_p f_k(p, c).  Thread = (pixel row, channel); a block reduces its pixel rows through
// LDS in row order and writes one partial per (k, c); `finalize` sums the block partials serially.
//   MODE 0: f0 = x                                   (mean)
//   MODE 1: f0 = (x - mean)^2                        (biased variance)
//   MODE 2: f0 = xd, f1 = xh * xd                    (tangent moments)
//   MODE 3: f0 = gy, f1 = gy * xh [, f2 = G, f3 = G * xh, f4 = G * t]   t = (xd - mdot) - xh * c
constexpr int MOM_MAXK = 5;
constexpr int MOM_BLOCKS = 2048;          // upper bound; see mom_blocks()
__host__ __device__ inline int mom_blocks(int C) { const int b = 262144 / C; return b < 256 ? 256 : (b > MOM_BLOCKS ? MOM_BLOCKS : b); }

struct MomArgs {
  const float *x, *xd, *gy, *G;
  int x_cs, xd_cs, gy_cs, G_cs;
  const float *mean, *invstd, *mdot, *cc;
  const float *mgamma, *mbeta;   // MODE 3, both non-null: the cotangents pass the derivative of the fused activation at
                                 // z = gamma * xh + beta (mact 1: ReLU mask z > 0; mact 2: swish'(z))
  int mact;
  float* partial;      // [blocks][nk][C]
  long P;
  int C, nk;
};

// derivative of the activation fused behind the BatchNorm, at the forward's own z (same expression: same bits)
__device__ __forceinline__ float bn_act_grad(int act, float z) {
  if (act == 2) {
    const float sg = 1.f / (1.f + expf(-z));
    return sg + z * sg * (1.f - sg);
  }
  return z > 0.f ? 1.f : 0.f;
}

// pivot of the one-pass variance (MODE 4): the mean of 16 pixels spread over the tensor.  sum (x - s) and sum (x - s)^2 in
// ONE read give var = E[(x-s)^2] - E[x-s]^2 without the cancellation of the raw-moment form as long as s lies within a
// few standard deviations of the mean (the two-pass form read the tensor twice: ~6 % of a distillation step)
__device__ __forceinline__ float bn_pivot(const float* __restrict__ x, int x_cs, long P, int c) {
  const long step = P >= 16 ? P / 16 : 1;
  const int n = P >= 16 ? 16 : (int)P;
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += x[(long)k * step * x_cs + c];
  return s / (float)n;
}

template <int MODE>
__global__ __launch_bounds__(256) void moments_kernel(const MomArgs a) {
  extern __shared__ float sm[];   // [nk][rows][Ct]
  // blockIdx.y walks channel tiles of 256; inside a tile thread = (pixel row, channel)
  const int cbase = blockIdx.y * 256;
  const int Ct = min(256, a.C - cbase), rows = 256 / Ct;
  const int c = cbase + threadIdx.x % Ct, row = threadIdx.x / Ct;
  const bool active = row < rows;
  float s[MOM_MAXK] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (active) {
    const float mu = (MODE == 4) ? bn_pivot(a.x, a.x_cs, a.P, c) : (MODE >= 1) ? a.mean[c] : 0.f;
    const float is = (MODE == 2 || MODE == 3) ? a.invstd[c] : 0.f;
    const float md = (MODE == 3 && a.G) ? a.mdot[c] : 0.f;
    const float cc = (MODE == 3 && a.G) ? a.cc[c] : 0.f;
#pragma unroll 4
    for (long p = (long)blockIdx.x * rows + row; p < a.P; p += (long)gridDim.x * rows) {
      const float xv = a.x[p * a.x_cs + c];
      if (MODE == 0) s[0] += xv;
      if (MODE == 1) { const float d = xv - mu; s[0] += d * d; }
      if (MODE == 4) { const float d = xv - mu; s[0] += d; s[1] += d * d; }
      if (MODE == 2) {
        const float xh = (xv - mu) * is, xd = a.xd[p * a.xd_cs + c];
        s[0] += xd; s[1] += xh * xd;
      }
      if (MODE == 3) {
        const float xh = (xv - mu) * is;
        const float on = a.mbeta ? bn_act_grad(a.mact, a.mgamma[c] * xh + a.mbeta[c]) : 1.f;
        if (a.gy) { const float g = a.gy[p * a.gy_cs + c] * on; s[0] += g; s[1] += g * xh; }
        if (a.G) {
          const float g = a.G[p * a.G_cs + c] * on;
          const float t = (a.xd[p * a.xd_cs + c] - md) - xh * cc;
          s[2] += g; s[3] += g * xh; s[4] += g * t;
        }
      }
    }
    for (int k = 0; k < a.nk; ++k) sm[((size_t)k * rows + row) * Ct + (c - cbase)] = s[k];
  }
  __syncthreads();
  if (active && row == 0) {
    for (int k = 0; k < a.nk; ++k) {
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += sm[((size_t)k * rows + r) * Ct + (c - cbase)];
      a.partial[((size_t)blockIdx.x * a.nk + k) * a.C + c] = t;
    }
  }
}

// float4 variant: thread = (pixel row, channel QUAD) -- 16-byte loads and every lane busy for any C % 4 == 0 (the
// scalar mapping idles 44 % of the workgroup at C = 144 and moves 4 bytes per load)
typedef float mq4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void moments4_kernel(const MomArgs a) {
  extern __shared__ float sm[];   // [nk][rows][Ct]
  const int cbase = blockIdx.y * 256;
  const int Ct = min(256, a.C - cbase), Ctq = Ct >> 2, rows = 256 / Ctq;
  const int q = threadIdx.x % Ctq, row = threadIdx.x / Ctq;
  const int c = cbase + q * 4;
  const bool active = row < rows;
  auto ld = [](const float* p) { return *reinterpret_cast<const mq4*>(p); };
  const mq4 z = {0.f, 0.f, 0.f, 0.f};
  mq4 s[MOM_MAXK] = {z, z, z, z, z};
  if (active) {
    mq4 mu = (MODE >= 1 && MODE != 4) ? ld(a.mean + c) : z;
    if (MODE == 4)
      for (int j = 0; j < 4; ++j) mu[j] = bn_pivot(a.x, a.x_cs, a.P, c + j);
    const mq4 is = (MODE == 2 || MODE == 3) ? ld(a.invstd + c) : z;
    const mq4 md = (MODE == 3 && a.G) ? ld(a.mdot + c) : z;
    const mq4 cc = (MODE == 3 && a.G) ? ld(a.cc + c) : z;
#pragma unroll 2
    for (long p = (long)blockIdx.x * rows + row; p < a.P; p += (long)gridDim.x * rows) {
      const mq4 xv = ld(a.x + p * a.x_cs + c);
      if (MODE == 0) s[0] += xv;
      if (MODE == 1) { const mq4 d = xv - mu; s[0] += d * d; }
      if (MODE == 4) { const mq4 d = xv - mu; s[0] += d; s[1] += d * d; }
      if (MODE == 2) {
        const mq4 xh = (xv - mu) * is, xd = ld(a.xd + p * a.xd_cs + c);
        s[0] += xd; s[1] += xh * xd;
      }
      if (MODE == 3) {
        const mq4 xh = (xv - mu) * is;
        mq4 on = {1.f, 1.f, 1.f, 1.f};
        if (a.mbeta) {                                    // the forward's y = gamma * xh + beta, same expression: same sign
          const mq4 yv = ld(a.mgamma + c) * xh + ld(a.mbeta + c);
          for (int j = 0; j < 4; ++j) on[j] = bn_act_grad(a.mact, yv[j]);
        }
        if (a.gy) { const mq4 g = ld(a.gy + p * a.gy_cs + c) * on; s[0] += g; s[1] += g * xh; }
        if (a.G) {
          const mq4 g = ld(a.G + p * a.G_cs + c) * on;
          const mq4 t = (ld(a.xd + p * a.xd_cs + c) - md) - xh * cc;
          s[2] += g; s[3] += g * xh; s[4] += g * t;
        }
      }
    }
    for (int k = 0; k < a.nk; ++k) *reinterpret_cast<mq4*>(sm + ((size_t)k * rows + row) * Ct + q * 4) = s[k];
  }
  __syncthreads();
  if ((int)threadIdx.x < Ct) {
    for (int k = 0; k < a.nk; ++k) {
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += sm[((size_t)k * rows + r) * Ct + threadIdx.x];
      a.partial[((size_t)blockIdx.x * a.nk + k) * a.C + cbase + threadIdx.x] = t;
    }
  }
}

// out[k][c] = sum_blocks partial / P ; with `var_to_invstd` the single output becomes 1/sqrt(var + eps) and
// the BatchNorm running statistics are updated (momentum m, unbiased variance) as nn.BatchNorm2d does.
// one wave per output: lane l sums the partials of blocks l, l+64, ... in order, then a fixed xor tree
__global__ __launch_bounds__(64) void moments_finalize_kernel(const float* __restrict__ partial,
                                                             float* __restrict__ out, int nblocks, int nk, int C,
                                                             float inv_p) {
  const int i = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[(size_t)b * nk * C + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[i] = s * inv_p;
}

__global__ void bn_finish_stats_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                       float* __restrict__ invstd, float* running_mean, float* running_var,
                                       int C, float eps, float momentum, float unbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  invstd[c] = 1.f / sqrtf(var[c] + eps);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (var[c] * unbias);
}

// MODE 4 partials [blocks][2][C] -> mean, biased variance, 1/sqrt(var + eps) and the running statistics: one wave per channel
// (fixed summation order), replacing two finalize launches and bn_finish_stats_kernel
__global__ __launch_bounds__(64) void bn_stats_finish_kernel(const float* __restrict__ partial, const float* __restrict__ x,
                                                             int x_cs, long P, int nblocks, int C, float* __restrict__ mean,
                                                             float* __restrict__ var, float* __restrict__ invstd,
                                                             float* running_mean, float* running_var, float eps,
                                                             float momentum, float unbias) {
  const int c = blockIdx.x;
  float s1 = 0.f, s2 = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 64) {
    s1 += partial[((size_t)b * 2 + 0) * C + c];
    s2 += partial[((size_t)b * 2 + 1) * C + c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (threadIdx.x == 0) {
    const float inv_p = 1.f / (float)P, m1 = s1 * inv_p, m2 = s2 * inv_p;
    const float mu = bn_pivot(x, x_cs, P, c) + m1;
    const float v = fmaxf(m2 - m1 * m1, 0.f);
    mean[c] = mu; var[c] = v; invstd[c] = 1.f / sqrtf(v + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (v * unbias);
  }
}

// ------------------------------------------------------------------------------------ elementwise
struct EwArgs {
  const float *x, *xd, *gy, *G;
  int x_cs, xd_cs, gy_cs, G_cs;
  const float *gamma, *beta, *mean, *invstd;
  const float *mom_t;     // tangent moments [2][C]: m(xd), c = m(xh*xd)
  const float *mom_b;     // backward moments [5][C]
  float *o0, *o1;         // outputs
  float* amax;            // optional running max|o0| (MODE 0 and 2), see block_amax_update
  int o0_cs, o1_cs;
  long P;
  int C, act;             // MODE 0: activation on the output (0 none, 1 ReLU, 2 swish); MODE 2 with beta: its derivative
};

//   MODE 0  bn forward      o0 = act(gamma*xh + beta)
//   MODE 1  bn tangent      o0 = gamma*invstd*((xd - m(xd)) - xh*c)
//   MODE 2  bn backward     o0 = gx, o1 = gxd (joint; G may be null -> plain first-order backward)
template <int MODE>
__global__ __launch_bounds__(256) void bn_elementwise_kernel(const EwArgs a) {
  __shared__ float amax_scratch[4];
  float vmax = 0.f;
  const long total = a.P * a.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / a.C;
    const int c = (int)(i - p * a.C);
    const float g = a.gamma ? a.gamma[c] : 1.f, is = a.invstd[c];
    const float xh = (a.x[p * a.x_cs + c] - a.mean[c]) * is;
    if (MODE == 0) {
      float y = g * xh + (a.beta ? a.beta[c] : 0.f);
      if (a.act == 1) y = fmaxf(y, 0.f);
      else if (a.act == 2) y = y / (1.f + expf(-y));
      vmax = fmaxf(vmax, fabsf(y));
      a.o0[p * a.o0_cs + c] = y;
    } else if (MODE == 1) {
      const float t = (a.xd[p * a.xd_cs + c] - a.mom_t[c]) - xh * a.mom_t[a.C + c];
      a.o0[p * a.o0_cs + c] = g * is * t;
    } else {
      float gx = 0.f;
      const float on = a.beta ? bn_act_grad(a.act, g * xh + a.beta[c]) : 1.f;   // MODE 2 with beta: fused activation
      if (a.gy) gx = g * is * (a.gy[p * a.gy_cs + c] * on - a.mom_b[c] - xh * a.mom_b[a.C + c]);
      if (a.G) {
        const float cc = a.mom_t[a.C + c];
        const float t = (a.xd[p * a.xd_cs + c] - a.mom_t[c]) - xh * cc;
        const float pg = a.G[p * a.G_cs + c] * on - a.mom_b[2 * a.C + c] - xh * a.mom_b[3 * a.C + c];   // P(G)
        gx -= g * is * is * (a.mom_b[4 * a.C + c] * xh + cc * pg + a.mom_b[3 * a.C + c] * t);
        a.o1[p * a.o1_cs + c] = g * is * pg;
      }
      vmax = fmaxf(vmax, fabsf(gx));
      a.o0[p * a.o0_cs + c] = gx;
    }
  }
  if (MODE != 1 && a.amax) block_amax_update(vmax, a.amax, amax_scratch);
}

// float4 variants of the streaming kernels (C and every pixel stride multiples of 4, 16-byte aligned bases, fewer
// than 2^31 quads): one 32-bit division per 16 bytes instead of two 64-bit divisions per 4 bytes
typedef float ew4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ew4 ld4e(const float* p) { return *reinterpret_cast<const ew4*>(p); }
__device__ __forceinline__ void st4e(float* p, ew4 v) { *reinterpret_cast<ew4*>(p) = v; }
__device__ __forceinline__ ew4 splat4(float v) { return ew4{v, v, v, v}; }
__device__ __forceinline__ ew4 relu4(ew4 v) { return ew4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }

template <int MODE>
__global__ __launch_bounds__(256) void bn_elementwise4_kernel(const EwArgs a) {
  __shared__ float amax_scratch[4];
  float vmax = 0.f;
  auto amax4 = [](ew4 v) { return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); };
  const unsigned Cq = (unsigned)a.C >> 2, total = (unsigned)a.P * Cq;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned p = i / Cq;
    const int c = (int)(i - p * Cq) << 2;
    const ew4 g = a.gamma ? ld4e(a.gamma + c) : splat4(1.f), is = ld4e(a.invstd + c);
    const ew4 xh = (ld4e(a.x + (long)p * a.x_cs + c) - ld4e(a.mean + c)) * is;
    if (MODE == 0) {
      ew4 y = g * xh + (a.beta ? ld4e(a.beta + c) : splat4(0.f));
      if (a.act == 1) y = relu4(y);
      else if (a.act == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = y[j] / (1.f + expf(-y[j]));
      }
      vmax = fmaxf(vmax, amax4(y));
      st4e(a.o0 + (long)p * a.o0_cs + c, y);
    } else if (MODE == 1) {
      const ew4 t = (ld4e(a.xd + (long)p * a.xd_cs + c) - ld4e(a.mom_t + c)) - xh * ld4e(a.mom_t + a.C + c);
      st4e(a.o0 + (long)p * a.o0_cs + c, g * is * t);
    } else {
      ew4 gx = splat4(0.f);
      ew4 on = splat4(1.f);
      if (a.beta) {                                        // MODE 2 with beta: the cotangents pass the fused ReLU's mask
        const ew4 yv = g * xh + ld4e(a.beta + c);
        on = ew4{bn_act_grad(a.act, yv[0]), bn_act_grad(a.act, yv[1]), bn_act_grad(a.act, yv[2]), bn_act_grad(a.act, yv[3])};
      }
      if (a.gy) gx = g * is * (ld4e(a.gy + (long)p * a.gy_cs + c) * on - ld4e(a.mom_b + c) - xh * ld4e(a.mom_b + a.C + c));
      if (a.G) {
        const ew4 cc = ld4e(a.mom_t + a.C + c);
        const ew4 t = (ld4e(a.xd + (long)p * a.xd_cs + c) - ld4e(a.mom_t + c)) - xh * cc;
        const ew4 pg = ld4e(a.G + (long)p * a.G_cs + c) * on - ld4e(a.mom_b + 2 * a.C + c) - xh * ld4e(a.mom_b + 3 * a.C + c);
        gx -= g * is * is * (ld4e(a.mom_b + 4 * a.C + c) * xh + cc * pg + ld4e(a.mom_b + 3 * a.C + c) * t);
        st4e(a.o1 + (long)p * a.o1_cs + c, g * is * pg);
      }
      vmax = fmaxf(vmax, amax4(gx));
      st4e(a.o0 + (long)p * a.o0_cs + c, gx);
    }
  }
  if (MODE != 1 && a.amax) block_amax_update(vmax, a.amax, amax_scratch);
}

static inline bool ew_vec_ok(const EwArgs& e) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  auto cs = [](const void* p, int s) { return !p || s % 4 == 0; };
  return e.C % 4 == 0 && e.P * (e.C / 4) < 2147483647L && al(e.x) && al(e.xd) && al(e.gy) && al(e.G) && al(e.o0) &&
         al(e.o1) && al(e.gamma) && al(e.beta) && al(e.mean) && al(e.invstd) && al(e.mom_t) && al(e.mom_b) &&
         cs(e.x, e.x_cs) && cs(e.xd, e.xd_cs) && cs(e.gy, e.gy_cs) && cs(e.G, e.G_cs) && cs(e.o0, e.o0_cs) &&
         cs(e.o1, e.o1_cs);
}

template <int MODE>
__global__ __launch_bounds__(256) void pointwise2_4_kernel(const float* __restrict__ a, int a_cs,
                                                           const float* __restrict__ b, int b_cs,
                                                           float* __restrict__ o, int o_cs, long P, int C) {
  const unsigned Cq = (unsigned)C >> 2, total = (unsigned)P * Cq;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned p = i / Cq;
    const int c = (int)(i - p * Cq) << 2;
    const ew4 av = ld4e(a + (long)p * a_cs + c);
    ew4 r;
    if (MODE == 0) r = relu4(av);
    else if (MODE == 1) {
      const ew4 bv = ld4e(b + (long)p * b_cs + c);
      r = ew4{av[0] > 0.f ? bv[0] : 0.f, av[1] > 0.f ? bv[1] : 0.f, av[2] > 0.f ? bv[2] : 0.f, av[3] > 0.f ? bv[3] : 0.f};
    } else r = av + ld4e(b + (long)p * b_cs + c);
    st4e(o + (long)p * o_cs + c, r);
  }
}

// g_gamma[c] (+)= P*(m(gy*xh) + invstd*m(G*t)),  g_beta[c] (+)= P*m(gy)
__global__ void bn_param_grad_kernel(const float* __restrict__ mom_b, const float* __restrict__ invstd,
                                     float* g_gamma, float* g_beta, int C, float P, int has_gy, int has_G,
                                     int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float gg = 0.f, gb = 0.f;
  if (has_gy) { gg += P * mom_b[C + c]; gb += P * mom_b[c]; }
  if (has_G) gg += P * invstd[c] * mom_b[4 * C + c];
  g_gamma[c] = accumulate ? g_gamma[c] + gg : gg;
  g_beta[c] = accumulate ? g_beta[c] + gb : gb;
}

// MODE 0 relu: o = max(a, 0); MODE 1 relu backward / tangent: o = (y > 0) ? b : 0; MODE 2 add: o = a + b
template <int MODE>
__global__ __launch_bounds__(256) void pointwise2_kernel(const float* __restrict__ a, int a_cs,
                                                         const float* __restrict__ b, int b_cs,
                                                         float* __restrict__ o, int o_cs, long P, int C) {
  const long total = P * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / C;
    const int c = (int)(i - p * C);
    const float av = a[p * a_cs + c];
    float r;
    if (MODE == 0) r = fmaxf(av, 0.f);
    else if (MODE == 1) r = av > 0.f ? b[p * b_cs + c] : 0.f;
    else r = av + b[p * b_cs + c];
    o[p * o_cs + c] = r;
  }
}

// ------------------------------------------------------------------------------------ max-pool 2x2/2
// forward with argmax (first maximum in (dy, dx) scan order, as ATen), tangent (gather by idx), backward
__global__ __launch_bounds__(256) void maxpool2_idx_kernel(const float* __restrict__ in, int in_cs, int H, int W,
                                                           int C, float* __restrict__ out, int out_cs,
                                                           uint8_t* __restrict__ idx, int N, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float best = 0.f; int bi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = in[(((long)n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * in_cs + c];
      if (k == 0 || v > best || v != v) { best = v; bi = k; }
    }
    out[(((long)n * Ho + oy) * Wo + ox) * out_cs + c] = best;
    idx[i] = (uint8_t)bi;
  }
}

// MODE 0: out[pooled] = in[full res at idx]   (tangent)
// MODE 1: out[full res] = (idx == own position) ? in[pooled] : 0   (backward; rows/cols beyond 2*Ho/2*Wo get 0)
template <int MODE>
__global__ __launch_bounds__(256) void maxpool2_route_kernel(const float* __restrict__ in, int in_cs,
                                                             const uint8_t* __restrict__ idx,
                                                             float* __restrict__ out, int out_cs, int N, int H,
                                                             int W, int C, int Ho, int Wo) {
  if (MODE == 0) {
    const long total = (long)N * Ho * Wo * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const int c = (int)(i % C);
      long t = i / C;
      const int ox = (int)(t % Wo); t /= Wo;
      const int oy = (int)(t % Ho);
      const int n = (int)(t / Ho);
      const int k = idx[i];
      out[(((long)n * Ho + oy) * Wo + ox) * out_cs + c] =
          in[(((long)n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * in_cs + c];
    }
  } else {
    const long total = (long)N * H * W * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const int c = (int)(i % C);
      long t = i / C;
      const int x = (int)(t % W); t /= W;
      const int y = (int)(t % H);
      const int n = (int)(t / H);
      float v = 0.f;
      const int oy = y >> 1, ox = x >> 1;
      if (oy < Ho && ox < Wo) {
        const long pi = (((long)n * Ho + oy) * Wo + ox);
        if (idx[pi * C + c] == ((y & 1) << 1 | (x & 1))) v = in[pi * in_cs + c];
      }
      out[(((long)n * H + y) * W + x) * out_cs + c] = v;
    }
  }
}

// ------------------------------------------------------------------------------------ upsample transpose
// gx[n, yi, xi, c] = sum over output pixels (yo, xo) of w(yo->yi) * w(xo->xi) * gy[n, yo, xo, c] with the
// forward's own source-index rule recomputed per candidate output (gather form: deterministic, borders exact).
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ gy, int gy_cs, int Ho, int Wo,
                                                           float* __restrict__ gx, int gx_cs, int N, int H1,
                                                           int W1, int C, float rh, float rw, int span_h,
                                                           int span_w) {
  const long total = (long)N * H1 * W1 * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long t = i / C;
    const int xi = (int)(t % W1); t /= W1;
    const int yi = (int)(t % H1);
    const int n = (int)(t / H1);
    // outputs that can reference row yi lie within +-span of yi / rh
    const int yc = (int)((float)yi / rh), xc = (int)((float)xi / rw);
    float s = 0.f;
    for (int yo = max(0, yc - span_h); yo <= min(Ho - 1, yc + span_h); ++yo) {
      float sy = rh * ((float)yo + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
      const int y0 = (int)sy, y1 = y0 + (y0 < H1 - 1 ? 1 : 0);
      const float ly = sy - (float)y0;
      float wy = 0.f;
      if (y0 == yi) wy += 1.f - ly;
      if (y1 == yi) wy += ly;
      if (wy == 0.f) continue;
      for (int xo = max(0, xc - span_w); xo <= min(Wo - 1, xc + span_w); ++xo) {
        float sx = rw * ((float)xo + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
        const int x0 = (int)sx, x1 = x0 + (x0 < W1 - 1 ? 1 : 0);
        const float lx = sx - (float)x0;
        float wx = 0.f;
        if (x0 == xi) wx += 1.f - lx;
        if (x1 == xi) wx += lx;
        if (wx != 0.f) s += wy * wx * gy[(((long)n * Ho + yo) * Wo + xo) * gy_cs + c];
      }
    }
    gx[(((long)n * H1 + yi) * W1 + xi) * gx_cs + c] = s;
  }
}

// exact 2x case (every Upsample in the network): input row yi receives output rows 2yi-1, 2yi, 2yi+1, 2yi+2 with
// weights 1/4, 3/4, 3/4, 1/4 (align_corners=False); at the borders the clamped source rows fold the outer weight
// into the edge (row 0 gets output row 0 with weight 1, no row -1; likewise at the far edge).  One thread per pixel
// and channel quad, 16 branch-free float4 loads.
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ gy, int gy_cs, float* __restrict__ gx,
                                                             int gx_cs, int N, int H1, int W1, int C) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int Cq = C >> 2, Ho = 2 * H1, Wo = 2 * W1;
  const long total = (long)N * H1 * W1 * Cq;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % Cq);
    long t = i / Cq;
    const int xi = (int)(t % W1); t /= W1;
    const int yi = (int)(t % H1);
    const int n = (int)(t / H1);
    float wy[4], wx[4];
    int yo[4], xo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = 2 * yi - 1 + k, xx = 2 * xi - 1 + k;
      const float base = (k == 0 || k == 3) ? 0.25f : 0.75f;
      // the outermost output row/column of the image maps wholly onto the edge input row/column
      wy[k] = (y < 0 || y >= Ho) ? 0.f : ((y == 0 || y == Ho - 1) ? 1.f : base);
      wx[k] = (xx < 0 || xx >= Wo) ? 0.f : ((xx == 0 || xx == Wo - 1) ? 1.f : base);
      yo[k] = min(max(y, 0), Ho - 1);
      xo[k] = min(max(xx, 0), Wo - 1);
    }
    const float* g = gy + (long)n * Ho * Wo * gy_cs + q * 4;
    f32x4 v[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        v[a * 4 + b] = *reinterpret_cast<const f32x4*>(g + ((long)yo[a] * Wo + xo[b]) * gy_cs);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s += (wy[a] * wx[b]) * v[a * 4 + b];
    *reinterpret_cast<f32x4*>(gx + (((long)n * H1 + yi) * W1 + xi) * gx_cs + q * 4) = s;
  }
}

// pixel chunks of the wgrad reduction: ~2048 workgroups over (chunk, tap), at least 256 pixels per chunk
static inline int wgrad_chunks(long M, int K) {
  long n = 2048 / (K * K);
  n = n < 32 ? 32 : (n > 512 ? 512 : n);
  const long cap = (M + 255) / 256;
  return (int)(n < cap ? n : cap);
}

static inline int grid1d(long work, int cap = 4096) {
  long b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace creste

using namespace creste;

extern "C" int64_t creste_conv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int K) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || K <= 0) return -1;
  const int64_t per_tap = (int64_t)wgrad_chunks((long)N * H * W, K) * K * K * Cout * Cin * 4;
  const WrPlan p = wr_plan(nullptr, 0, nullptr, 0, N, H, W, Cin, Cout, K);     // strides / alignment decide at run time
  const int64_t rows = p.ok ? (int64_t)p.nwg * K * K * Cout * Cin * 4 : 0;
  return per_tap > rows ? per_tap : rows;
}

extern "C" int creste_conv_wgrad_f32(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, int N, int H,
                                     int W, int Cin, int Cout, int K, int pad, int accumulate, void* work,
                                     void* stream) {
  CRESTE_REQUIRE(x && gy && gw && work, "conv_wgrad: null pointer");
  CRESTE_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && K > 0 && pad >= 0 && 2 * pad == K - 1,
                 "conv_wgrad: stride-1 'same' convolutions only (K=%d pad=%d)", K, pad);
  CRESTE_REQUIRE(K * K <= 65535, "conv_wgrad: kernel too large");
  const long M = (long)N * H * W;
  CRESTE_REQUIRE(M < (1L << 31), "conv_wgrad: N*H*W overflows int32");
  hipStream_t s = (hipStream_t)stream;
  const WrPlan wp = wr_plan(x, x_cs, gy, gy_cs, N, H, W, Cin, Cout, K);
  if (wp.ok) {
    static std::atomic<uint64_t> devs[9];
    WrArgs a{x, gy, (float*)work, x_cs, gy_cs, N, H, W, Cin, Cout, wp.cpx, wp.cpy, wp.tci, wp.pairs, wp.nseg, wp.nband,
             wp.band_rows};
#define CRESTE_WR(K_, NTAP_, MAXT_, CPX_, SLOT_)                                                                        \
  {                                                                                                                    \
    CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wgrad_rows_kernel<K_, NTAP_, MAXT_, CPX_>), 160 * 1024,   \
                               devs[SLOT_]));                                                                          \
    wgrad_rows_kernel<K_, NTAP_, MAXT_, CPX_><<<wp.nwg, wp.nw * 64, wp.smem, s>>>(a);                                  \
  }
    if (K == 5 && wp.ntap == 25 && wp.cpx == 48) CRESTE_WR(5, 25, 768, 48, 0)
    else if (K == 5 && wp.ntap == 25) CRESTE_WR(5, 25, 768, 80, 1)
    else if (K == 5 && wp.ntap == 13) CRESTE_WR(5, 13, 768, 0, 2)
    else if (K == 5) CRESTE_WR(5, 7, 1024, 0, 3)
    else if (wp.ntap == 9 && wp.cpx == 48) CRESTE_WR(3, 9, 1024, 48, 4)
    else if (wp.ntap == 9) CRESTE_WR(3, 9, 1024, 80, 5)
    else if (wp.ntap == 5) CRESTE_WR(3, 5, 1024, 0, 6)
    else if (K == 3) CRESTE_WR(3, 3, 1024, 0, 7)
    else CRESTE_WR(1, 1, 1024, 0, 8)
#undef CRESTE_WR
    CRESTE_CHECK_LAUNCH("wgrad_rows");
    wgrad_reduce4_kernel<<<(Cout * Cin * K * K + 63) / 64, 256, 0, s>>>((const float*)work, gw, wp.nwg, Cout, Cin, K * K,
                                                                        accumulate);
    CRESTE_CHECK_LAUNCH("wgrad_reduce");
    return CRESTE_OK;
  }
  if (K == 1 && Cout < 8 && Cin % 4 == 0 && Cin <= 1024 && x_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int nb = wgrad_chunks(M, 1);                       // the per-tap path's workspace bound
    wgrad_thin_kernel<<<nb, 256, 0, s>>>(x, x_cs, gy, gy_cs, (float*)work, M, Cin, Cout);
    CRESTE_CHECK_LAUNCH("wgrad_thin");
    wgrad_reduce4_kernel<<<(Cout * Cin + 63) / 64, 256, 0, s>>>((const float*)work, gw, nb, Cout, Cin, 1, accumulate);
    CRESTE_CHECK_LAUNCH("wgrad_reduce");
    return CRESTE_OK;
  }
  int nchunk = wgrad_chunks(M, K);
  long chunk_px = (M + nchunk - 1) / nchunk;
  chunk_px = (chunk_px + WG_PIX - 1) / WG_PIX * WG_PIX;
  nchunk = (int)((M + chunk_px - 1) / chunk_px);
  wgrad_partial_kernel<<<dim3(nchunk, K * K), 256, 0, s>>>(x, x_cs, gy, gy_cs, (float*)work, N, H, W, Cin, Cout, K,
                                                          pad, (int)chunk_px);
  CRESTE_CHECK_LAUNCH("wgrad_partial");
  wgrad_reduce4_kernel<<<(Cout * Cin * K * K + 63) / 64, 256, 0, s>>>((const float*)work, gw, nchunk, Cout, Cin, K * K,
                                                                      accumulate);
  CRESTE_CHECK_LAUNCH("wgrad_reduce");
  return CRESTE_OK;
}

extern "C" int creste_conv_flip_weight_f32(const float* w, float* wt, int Cout, int Cin, int K, int cout_pad,
                                           void* stream) {
  CRESTE_REQUIRE(w && wt && Cout > 0 && Cin > 0 && K > 0 && cout_pad >= Cout, "conv_flip_weight: bad args");
  flip_weight_kernel<<<grid1d((long)Cin * cout_pad * K * K, 256), 256, 0, (hipStream_t)stream>>>(w, wt, Cout, Cin, K,
                                                                                                 cout_pad);
  CRESTE_CHECK_LAUNCH("flip_weight");
  return CRESTE_OK;
}

// mode 4 (one-pass shifted moments): `out` is unused, *nblocks_out receives the number of partial rows for
// bn_stats_finish_kernel
static int run_moments(int mode, MomArgs a, float* out, float* partial, hipStream_t s, int* nblocks_out = nullptr) {
  CRESTE_REQUIRE(a.C > 0 && a.P > 0, "bn moments: bad dims");
  a.partial = partial;
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  auto cs = [](const void* p, int st) { return !p || st % 4 == 0; };
  if (a.C % 4 == 0 && al(a.x) && al(a.xd) && al(a.gy) && al(a.G) && al(a.mean) && al(a.invstd) && al(a.mdot) && al(a.cc) &&
      al(a.mgamma) && al(a.mbeta) &&
      cs(a.x, a.x_cs) && cs(a.xd, a.xd_cs) && cs(a.gy, a.gy_cs) && cs(a.G, a.G_cs)) {
    const int ctq = a.C >= 256 ? 64 : a.C / 4, rows4 = 256 / ctq;
    const long per4 = (a.P + rows4 - 1) / rows4;
    const int blocks4 = (int)(per4 < mom_blocks(a.C) ? per4 : mom_blocks(a.C));
    const size_t smem4 = (size_t)a.nk * 1024 * sizeof(float);
    const dim3 grid4(blocks4, (a.C + 255) / 256);
    if (mode == 0) moments4_kernel<0><<<grid4, 256, smem4, s>>>(a);
    else if (mode == 1) moments4_kernel<1><<<grid4, 256, smem4, s>>>(a);
    else if (mode == 2) moments4_kernel<2><<<grid4, 256, smem4, s>>>(a);
    else if (mode == 4) moments4_kernel<4><<<grid4, 256, smem4, s>>>(a);
    else moments4_kernel<3><<<grid4, 256, smem4, s>>>(a);
    CRESTE_CHECK_LAUNCH("bn_moments4");
    if (mode == 4) { *nblocks_out = blocks4; return CRESTE_OK; }
    moments_finalize_kernel<<<a.nk * a.C, 64, 0, s>>>(partial, out, blocks4, a.nk, a.C, 1.f / (float)a.P);
    CRESTE_CHECK_LAUNCH("bn_moments_finalize");
    return CRESTE_OK;
  }
  const int rows = a.C >= 256 ? 1 : 256 / a.C;
  const long per = (a.P + rows - 1) / rows;
  const int blocks = (int)(per < mom_blocks(a.C) ? per : mom_blocks(a.C));
  const size_t smem = (size_t)a.nk * 256 * sizeof(float);
  const dim3 grid(blocks, (a.C + 255) / 256);
  if (mode == 0) moments_kernel<0><<<grid, 256, smem, s>>>(a);
  else if (mode == 1) moments_kernel<1><<<grid, 256, smem, s>>>(a);
  else if (mode == 2) moments_kernel<2><<<grid, 256, smem, s>>>(a);
  else if (mode == 4) moments_kernel<4><<<grid, 256, smem, s>>>(a);
  else moments_kernel<3><<<grid, 256, smem, s>>>(a);
  CRESTE_CHECK_LAUNCH("bn_moments");
  if (mode == 4) { *nblocks_out = blocks; return CRESTE_OK; }
  moments_finalize_kernel<<<a.nk * a.C, 64, 0, s>>>(partial, out, blocks, a.nk, a.C, 1.f / (float)a.P);
  CRESTE_CHECK_LAUNCH("bn_moments_finalize");
  return CRESTE_OK;
}

extern "C" int64_t creste_bn_workspace_bytes(int C) { return C > 0 ? (int64_t)mom_blocks(C) * MOM_MAXK * C * 4 : -1; }

extern "C" int creste_bn_train_forward_f32(const float* x, int x_cs, int64_t P, int C, const float* gamma,
                                           const float* beta, float eps, float momentum, float* running_mean,
                                           float* running_var, float* mean, float* invstd, float* var_scratch,
                                           float* y, int y_cs, int relu, float* out_amax, void* work, void* stream) {
  CRESTE_REQUIRE(x && mean && invstd && var_scratch && y && work && P > 1, "bn_train_forward: bad args");
  hipStream_t s = (hipStream_t)stream;
  MomArgs a = {};
  a.x = x; a.x_cs = x_cs; a.P = P; a.C = C; a.nk = 2;
  int nblocks = 0;
  const int rc = run_moments(4, a, nullptr, (float*)work, s, &nblocks);       // one read: sum (x - pivot), sum (x - pivot)^2
  if (rc) return rc;
  bn_stats_finish_kernel<<<C, 64, 0, s>>>((const float*)work, x, x_cs, P, nblocks, C, mean, var_scratch, invstd,
                                         running_mean, running_var, eps, momentum, (float)P / (float)(P - 1));
  CRESTE_CHECK_LAUNCH("bn_stats_finish");
  EwArgs e = {};
  e.x = x; e.x_cs = x_cs; e.gamma = gamma; e.beta = beta; e.mean = mean; e.invstd = invstd;
  e.o0 = y; e.o0_cs = y_cs; e.P = P; e.C = C; e.act = relu; e.amax = out_amax;
  if (ew_vec_ok(e)) bn_elementwise4_kernel<0><<<grid1d(P * C / 4, 1024), 256, 0, s>>>(e);
  else bn_elementwise_kernel<0><<<grid1d(P * C), 256, 0, s>>>(e);
  CRESTE_CHECK_LAUNCH("bn_forward");
  return CRESTE_OK;
}

// statistics from the producing conv's per-workgroup sums ([rows][2][C]: sum x, sum x^2): one wave per channel, lane l adds
// rows l, l + 64, ... in float64, fixed xor tree -> mean, biased variance, 1/sqrt(var + eps), running statistics.
// The epilogues add RAW x and x^2 in fp32 (a few hundred values per row), so E[x^2] - E[x]^2 loses about mean^2 / var x 1e-6 of
// the variance: harmless while |mean| is a few standard deviations, wrong for a channel that sits far from zero
// (large bias, late training).  A channel whose sums say mean^2 > BN_CANCEL_RATIO x var therefore does not trust them: the
// same wave re-reads its channel of x and takes the shifted two-moment sums about the (accurate) mean in float64 -- the
// statistics pass for that channel only, fixed order, so the result stays reproducible.
constexpr double BN_CANCEL_RATIO = 64.0;
__global__ __launch_bounds__(64) void bn_stats_from_partials_kernel(const float* __restrict__ partial, int rows, int C, long P,
                                                                    const float* __restrict__ x, int x_cs,
                                                                    float* __restrict__ mean, float* __restrict__ var,
                                                                    float* __restrict__ invstd, float* running_mean,
                                                                    float* running_var, float eps, float momentum, float unbias) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int r = threadIdx.x; r < rows; r += 64) {
    s1 += (double)partial[((size_t)r * 2 + 0) * C + c];
    s2 += (double)partial[((size_t)r * 2 + 1) * C + c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  double m = s1 / (double)P, v = fmax(s2 / (double)P - m * m, 0.0);          // (every lane holds the same sums)
  if (m * m > BN_CANCEL_RATIO * v) {
    const float pivot = (float)m;
    double d1 = 0.0, d2 = 0.0;
    for (long p0 = threadIdx.x; p0 < P; p0 += 256) {
      float t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const long p = p0 + 64 * k; t[k] = p < P ? x[p * x_cs + c] - pivot : 0.f; }
#pragma unroll
      for (int k = 0; k < 4; ++k) { d1 += (double)t[k]; d2 += (double)t[k] * (double)t[k]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
    const double dm = d1 / (double)P;
    m = (double)pivot + dm;
    v = fmax(d2 / (double)P - dm * dm, 0.0);
  }
  if (threadIdx.x == 0) {
    const float mu = (float)m, vf = (float)v;
    mean[c] = mu; var[c] = vf; invstd[c] = 1.f / sqrtf(vf + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (vf * unbias);
  }
}

extern "C" int creste_bn_train_forward_stats_f32(const float* x, int x_cs, int64_t P, int C, const float* gamma,
                                                 const float* beta, float eps, float momentum, float* running_mean,
                                                 float* running_var, float* mean, float* invstd, float* var_scratch,
                                                 float* y, int y_cs, int relu, float* out_amax, const float* stat_partial,
                                                 int stat_rows, void* stream) {
  CRESTE_REQUIRE(x && mean && invstd && var_scratch && y && stat_partial && stat_rows > 0 && P > 1 && C > 0,
                 "bn_train_forward_stats: bad args");
  hipStream_t s = (hipStream_t)stream;
  bn_stats_from_partials_kernel<<<C, 64, 0, s>>>(stat_partial, stat_rows, C, P, x, x_cs, mean, var_scratch, invstd, running_mean,
                                                 running_var, eps, momentum, (float)P / (float)(P - 1));
  CRESTE_CHECK_LAUNCH("bn_stats_from_partials");
  EwArgs e = {};
  e.x = x; e.x_cs = x_cs; e.gamma = gamma; e.beta = beta; e.mean = mean; e.invstd = invstd;
  e.o0 = y; e.o0_cs = y_cs; e.P = P; e.C = C; e.act = relu; e.amax = out_amax;
  if (ew_vec_ok(e)) bn_elementwise4_kernel<0><<<grid1d(P * C / 4, 1024), 256, 0, s>>>(e);
  else bn_elementwise_kernel<0><<<grid1d(P * C), 256, 0, s>>>(e);
  CRESTE_CHECK_LAUNCH("bn_forward");
  return CRESTE_OK;
}

extern "C" int creste_bn_train_tangent_f32(const float* x, int x_cs, const float* xd, int xd_cs, int64_t P, int C,
                                           const float* gamma, const float* mean, const float* invstd,
                                           float* mom_t, float* yd, int yd_cs, void* work, void* stream) {
  CRESTE_REQUIRE(x && xd && mean && invstd && mom_t && yd && work, "bn_train_tangent: null pointer");
  hipStream_t s = (hipStream_t)stream;
  MomArgs a = {};
  a.x = x; a.x_cs = x_cs; a.xd = xd; a.xd_cs = xd_cs; a.P = P; a.C = C; a.nk = 2; a.mean = mean; a.invstd = invstd;
  const int rc = run_moments(2, a, mom_t, (float*)work, s);
  if (rc) return rc;
  EwArgs e = {};
  e.x = x; e.x_cs = x_cs; e.xd = xd; e.xd_cs = xd_cs; e.gamma = gamma; e.mean = mean; e.invstd = invstd;
  e.mom_t = mom_t; e.o0 = yd; e.o0_cs = yd_cs; e.P = P; e.C = C;
  if (ew_vec_ok(e)) bn_elementwise4_kernel<1><<<grid1d(P * C / 4, 8192), 256, 0, s>>>(e);
  else bn_elementwise_kernel<1><<<grid1d(P * C), 256, 0, s>>>(e);
  CRESTE_CHECK_LAUNCH("bn_tangent");
  return CRESTE_OK;
}

static int bn_train_backward_impl(int act, const float* relu_beta, const float* x, int x_cs, const float* xd, int xd_cs, const float* gy,
                                            int gy_cs, const float* gyd, int gyd_cs, int64_t P, int C,
                                            const float* gamma, const float* mean, const float* invstd,
                                            const float* mom_t, float* mom_b, float* gx, int gx_cs, float* gxd,
                                            int gxd_cs, float* g_gamma, float* g_beta, int accumulate,
                                            float* gx_amax, void* work, void* stream) {
  CRESTE_REQUIRE(x && mean && invstd && mom_b && gx && work && (gy || gyd), "bn_train_backward: null pointer");
  CRESTE_REQUIRE(!gyd || (xd && mom_t && gxd), "bn_train_backward: the tangent cotangent needs xd, mom_t and gxd");
  CRESTE_REQUIRE(!relu_beta || gamma, "bn_act_train_backward: gamma is needed to recompute the activation's input");
  CRESTE_REQUIRE(act != 2 || !gyd, "bn_act_train_backward: the swish form is first order only");
  hipStream_t s = (hipStream_t)stream;
  MomArgs a = {};
  a.x = x; a.x_cs = x_cs; a.xd = xd; a.xd_cs = xd_cs; a.gy = gy; a.gy_cs = gy_cs; a.G = gyd; a.G_cs = gyd_cs;
  a.P = P; a.C = C; a.nk = 5; a.mean = mean; a.invstd = invstd;
  if (gyd) { a.mdot = mom_t; a.cc = mom_t + C; }
  if (relu_beta) { a.mgamma = gamma; a.mbeta = relu_beta; a.mact = act; }
  const int rc = run_moments(3, a, mom_b, (float*)work, s);
  if (rc) return rc;
  EwArgs e = {};
  e.x = x; e.x_cs = x_cs; e.xd = xd; e.xd_cs = xd_cs; e.gy = gy; e.gy_cs = gy_cs; e.G = gyd; e.G_cs = gyd_cs;
  e.gamma = gamma; e.beta = relu_beta; e.act = act; e.mean = mean; e.invstd = invstd; e.mom_t = mom_t; e.mom_b = mom_b;
  e.o0 = gx; e.o0_cs = gx_cs; e.o1 = gxd; e.o1_cs = gxd_cs; e.P = P; e.C = C; e.amax = gx_amax;
  if (ew_vec_ok(e)) bn_elementwise4_kernel<2><<<grid1d(P * C / 4, 1024), 256, 0, s>>>(e);
  else bn_elementwise_kernel<2><<<grid1d(P * C), 256, 0, s>>>(e);
  CRESTE_CHECK_LAUNCH("bn_backward");
  if (g_gamma && g_beta) {
    bn_param_grad_kernel<<<(C + 255) / 256, 256, 0, s>>>(mom_b, invstd, g_gamma, g_beta, C, (float)P, gy != nullptr,
                                                        gyd != nullptr, accumulate);
    CRESTE_CHECK_LAUNCH("bn_param_grad");
  }
  return CRESTE_OK;
}

extern "C" int creste_bn_train_backward_f32(const float* x, int x_cs, const float* xd, int xd_cs, const float* gy,
                                            int gy_cs, const float* gyd, int gyd_cs, int64_t P, int C,
                                            const float* gamma, const float* mean, const float* invstd,
                                            const float* mom_t, float* mom_b, float* gx, int gx_cs, float* gxd,
                                            int gxd_cs, float* g_gamma, float* g_beta, int accumulate,
                                            float* gx_amax, void* work, void* stream) {
  return bn_train_backward_impl(0, nullptr, x, x_cs, xd, xd_cs, gy, gy_cs, gyd, gyd_cs, P, C, gamma, mean, invstd, mom_t, mom_b,
                                gx, gx_cs, gxd, gxd_cs, g_gamma, g_beta, accumulate, gx_amax, work, stream);
}

// BatchNorm + ReLU backward in one: the cotangents are masked with the ReLU's own mask, recomputed from x with the
// forward's expression (gamma * xh + beta > 0: same operations, same sign) -- no separate pass that reads y and gy and
// writes the masked cotangent (3 of the 8 tensor passes of a BatchNorm + ReLU backward)
extern "C" int creste_bn_relu_train_backward_f32(const float* x, int x_cs, const float* xd, int xd_cs, const float* gy,
                                                 int gy_cs, const float* gyd, int gyd_cs, int64_t P, int C,
                                                 const float* gamma, const float* beta, const float* mean,
                                                 const float* invstd, const float* mom_t, float* mom_b, float* gx,
                                                 int gx_cs, float* gxd, int gxd_cs, float* g_gamma, float* g_beta,
                                                 int accumulate, float* gx_amax, void* work, void* stream) {
  CRESTE_REQUIRE(beta, "bn_relu_train_backward: null beta");
  return bn_train_backward_impl(1, beta, x, x_cs, xd, xd_cs, gy, gy_cs, gyd, gyd_cs, P, C, gamma, mean, invstd, mom_t, mom_b,
                                gx, gx_cs, gxd, gxd_cs, g_gamma, g_beta, accumulate, gx_amax, work, stream);
}

// the same for any activation fused behind the BatchNorm (act 1: ReLU, 2: swish = z * sigmoid(z), first order only):
// the cotangents are multiplied by act'(z) with z = gamma * xh + beta recomputed from x -- the unfused swish kept z AND y
// in memory and spent two more passes (forward z -> y, backward (z, gy) -> gz)
extern "C" int creste_bn_act_train_backward_f32(int act, const float* x, int x_cs, const float* gy, int gy_cs, int64_t P,
                                                int C, const float* gamma, const float* beta, const float* mean,
                                                const float* invstd, float* mom_b, float* gx, int gx_cs, float* g_gamma,
                                                float* g_beta, int accumulate, float* gx_amax, void* work, void* stream) {
  CRESTE_REQUIRE(act == 1 || act == 2, "bn_act_train_backward: act must be 1 (ReLU) or 2 (swish)");
  CRESTE_REQUIRE(beta && gy, "bn_act_train_backward: null beta / gy");
  return bn_train_backward_impl(act, beta, x, x_cs, nullptr, 0, gy, gy_cs, nullptr, 0, P, C, gamma, mean, invstd, nullptr,
                                mom_b, gx, gx_cs, nullptr, 0, g_gamma, g_beta, accumulate, gx_amax, work, stream);
}

extern "C" int creste_pointwise2_f32(int op, const float* a, int a_cs, const float* b, int b_cs, float* o, int o_cs,
                                     int64_t P, int C, void* stream) {
  CRESTE_REQUIRE(a && o && (op == 0 || b) && op >= 0 && op <= 2 && P > 0 && C > 0, "pointwise2: bad args");
  hipStream_t s = (hipStream_t)stream;
  const bool vec = C % 4 == 0 && a_cs % 4 == 0 && o_cs % 4 == 0 && (!b || b_cs % 4 == 0) && P * (C / 4) < 2147483647L &&
                   (((uintptr_t)a | (uintptr_t)b | (uintptr_t)o) & 15) == 0;
  if (vec) {
    const int g4 = grid1d(P * C / 4, 8192);
    if (op == 0) pointwise2_4_kernel<0><<<g4, 256, 0, s>>>(a, a_cs, b, b_cs, o, o_cs, P, C);
    else if (op == 1) pointwise2_4_kernel<1><<<g4, 256, 0, s>>>(a, a_cs, b, b_cs, o, o_cs, P, C);
    else pointwise2_4_kernel<2><<<g4, 256, 0, s>>>(a, a_cs, b, b_cs, o, o_cs, P, C);
    CRESTE_CHECK_LAUNCH("pointwise2");
    return CRESTE_OK;
  }
  const int g = grid1d(P * C);
  if (op == 0) pointwise2_kernel<0><<<g, 256, 0, s>>>(a, a_cs, b, b_cs, o, o_cs, P, C);
  else if (op == 1) pointwise2_kernel<1><<<g, 256, 0, s>>>(a, a_cs, b, b_cs, o, o_cs, P, C);
  else pointwise2_kernel<2><<<g, 256, 0, s>>>(a, a_cs, b, b_cs, o, o_cs, P, C);
  CRESTE_CHECK_LAUNCH("pointwise2");
  return CRESTE_OK;
}

extern "C" int creste_maxpool2_idx_f32(const float* in, int in_cs, int N, int H, int W, int C, float* out,
                                       int out_cs, uint8_t* idx, void* stream) {
  CRESTE_REQUIRE(in && out && idx && N > 0 && H > 1 && W > 1 && C > 0, "maxpool2_idx: bad args");
  maxpool2_idx_kernel<<<grid1d((long)N * (H / 2) * (W / 2) * C), 256, 0, (hipStream_t)stream>>>(
      in, in_cs, H, W, C, out, out_cs, idx, N, H / 2, W / 2);
  CRESTE_CHECK_LAUNCH("maxpool2_idx");
  return CRESTE_OK;
}

extern "C" int creste_maxpool2_route_f32(int backward, const float* in, int in_cs, const uint8_t* idx, float* out,
                                         int out_cs, int N, int H, int W, int C, void* stream) {
  CRESTE_REQUIRE(in && out && idx && N > 0 && H > 1 && W > 1 && C > 0, "maxpool2_route: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (!backward)
    maxpool2_route_kernel<0><<<grid1d((long)N * (H / 2) * (W / 2) * C), 256, 0, s>>>(in, in_cs, idx, out, out_cs, N, H,
                                                                                    W, C, H / 2, W / 2);
  else
    maxpool2_route_kernel<1><<<grid1d((long)N * H * W * C), 256, 0, s>>>(in, in_cs, idx, out, out_cs, N, H, W, C,
                                                                        H / 2, W / 2);
  CRESTE_CHECK_LAUNCH("maxpool2_route");
  return CRESTE_OK;
}

extern "C" int creste_upsample_bwd_nhwc_f32(const float* gy, int gy_cs, int Ho, int Wo, float* gx, int gx_cs, int N,
                                            int H1, int W1, int C, float rh, float rw, void* stream) {
  CRESTE_REQUIRE(gy && gx && N > 0 && H1 > 0 && W1 > 0 && Ho > 0 && Wo > 0 && C > 0 && rh > 0.f && rw > 0.f,
                 "upsample_bwd: bad args");
  if (Ho == 2 * H1 && Wo == 2 * W1 && rh == 0.5f && rw == 0.5f && C % 4 == 0 && gy_cs % 4 == 0 && gx_cs % 4 == 0 &&
      H1 > 1 && W1 > 1 && ((uintptr_t)gy & 15) == 0 && ((uintptr_t)gx & 15) == 0) {
    upsample2x_bwd_kernel<<<grid1d((long)N * H1 * W1 * (C / 4), 8192), 256, 0, (hipStream_t)stream>>>(gy, gy_cs, gx, gx_cs,
                                                                                                     N, H1, W1, C);
    CRESTE_CHECK_LAUNCH("upsample2x_bwd");
    return CRESTE_OK;
  }
  const int span_h = (int)(1.f / rh) + 2, span_w = (int)(1.f / rw) + 2;
  upsample_bwd_kernel<<<grid1d((long)N * H1 * W1 * C), 256, 0, (hipStream_t)stream>>>(gy, gy_cs, Ho, Wo, gx, gx_cs, N, H1,
                                                                                      W1, C, rh, rw, span_h, span_w);
  CRESTE_CHECK_LAUNCH("upsample_bwd");
  return CRESTE_OK;
}
