// MBConv front half in ONE pass: expand 1x1 conv (+ folded BN + swish) -> depthwise KxK / stride S conv (+ folded BN +
// swish) -> squeeze-excite partial channel sums, for the early EfficientNet blocks whose expanded tensor is the largest
// HBM object of the encoder (16 -> 96 channels at 304x608x16 = 1.1 GB written by the expand conv and read back by the
// depthwise conv; reference call site creste/models/blocks/effnet.py:83 -> efficientnet_pytorch MBConvBlock.forward).
// The expanded activations never reach HBM.
//
// One workgroup (256 threads) owns a strip of TW output columns x a band of output rows x ALL expanded channels and
// walks down the band one output row at a time:
//   ring  : the last K expanded rows (IW = (TW-1)*S + K pixels x Cexp channels) in LDS -- each expanded value is
//           computed once per strip (horizontal halo (K-S)/(TW*S), vertical halo K-S rows per band)
//   phase 1  S new input rows (K at the top of the band) are staged in LDS 
//   phase 2  expand: thread = (pixel slice, channel quad); the quad's Cin x 4 weights live in registers, the pixel's
//            input vector is a broadcast ds_read_b128 per 4 input channels; EXACT fp32 FMAs in channel order
//            (Cin <= 40: this is VALU work, ~1/20 of the time the tensor's HBM round trip costs), pixels outside the
//            image are stored as zeros (the depthwise conv pads the EXPANDED map)
//   phase 3  depthwise: thread = (output column slice, channel quad), taps read from the ring, same accumulation order
//            as dwconv_se_kernel (bias, then ky-major / kx-minor) -> output row, SE sums, running |max|
#include "common.h"

namespace creste {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MbArgs {
  const float* x; const float* we; const float* be; const float* wd; const float* bd;
  float* out; float* partial; float* amax;
  int H, W, x_cs, Cexp, Ho, Wo, pad_t, pad_l, rows_per_band, nbands, nstrips;
};

// fused multiply-adds, written out: the library is built with -ffp-contract=off (the geometry kernels rely on it), and
// these kernels are VALU-bound -- a * b + c as two instructions doubled their arithmetic
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 fma4(f32x4 a, float b, f32x4 c) { return __builtin_elementwise_fma(a, f32x4{b, b, b, b}, c); }
// CRESTE_MB_BISECT (scripts/micro/pk_pair.sh, a build WITH packed fp32, experiment only): 1 = the expand stage's FMAs as four scalar
// fmaf (the packed build then has v_pk_fma_f32 with a broadcast operand -- op_sel -- nowhere in the fused kernel), 2 = the depthwise
// stage's, 3 = the swish's multiply; which stage's packed instructions are the ones that go wrong beside another kernel's MFMAs
#ifndef CRESTE_MB_BISECT
#define CRESTE_MB_BISECT 0
#endif
// 4: the expand stage's FMAs stay PACKED but without operand-half selection: the broadcast pair {b, b} is materialised in a register pair
// (opaque to the optimiser) and the instruction is a plain v_pk_fma_f32 -- op_sel or not is then the only difference to variant 0
typedef float mbf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 fma4b(f32x4 a, float b, f32x4 c) {
  mbf32x2 bb = {b, b};
  asm volatile("" : "+v"(bb));
  const mbf32x2 lo = __builtin_elementwise_fma(mbf32x2{a[0], a[1]}, bb, mbf32x2{c[0], c[1]});
  const mbf32x2 hi = __builtin_elementwise_fma(mbf32x2{a[2], a[3]}, bb, mbf32x2{c[2], c[3]});
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 fma4s(f32x4 a, f32x4 b, f32x4 c) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { float t; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a[i]), "v"(b[i]), "v"(c[i])); r[i] = t; }
  return r;
}

// swish on the hardware transcendentals: v * rcp(1 + exp2(-v * log2 e)) -- v_exp_f32 and v_rcp_f32 are 1-ulp
// instructions; expf + an IEEE division cost ~25 VALU instructions per value, and this kernel is VALU-bound (the two
// activations were 2/3 of its instruction count).  Relative error <= ~1e-6 for |v| <= 10.
__device__ __forceinline__ f32x4 swish4(f32x4 v) {
  f32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    r[j] = v[j] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v[j] * -1.44269504088896341f));
  return r;
}

template <int K, int S, int CIN, int TW, int MINB>
__global__ __launch_bounds__(256, MINB) void mbconv_expand_dw_kernel(const MbArgs a) {
  constexpr int IW = (TW - 1) * S + K;
  constexpr int CQ = CIN / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Cexp = a.Cexp, NQ = Cexp >> 2, PP = 256 / NQ;
  float* const xrow = smem;                               // [K][IW][CIN]
  float* const ring = xrow + K * IW * CIN;                // [K][IW][Cexp]
  float* const red = ring + K * IW * Cexp;                // [PP][Cexp]
  const int tid = threadIdx.x;
  const int q = tid % NQ, ps = tid / NQ;
  const bool active = ps < PP;
  const int n = blockIdx.y;
  const int chunk = xcd_remap(blockIdx.x, a.nstrips * a.nbands);
  const int strip = chunk % a.nstrips, band = chunk / a.nstrips;
  const int ox0 = strip * TW;
  const int oy0 = band * a.rows_per_band, oy1 = min(a.Ho, oy0 + a.rows_per_band);
  const int ix0 = ox0 * S - a.pad_l;                      // image column of ring / xrow pixel 0
  const int iy_base = oy0 * S - a.pad_t;                  // image row of ring row 0 (may be negative)

  f32x4 wq[CIN], dwq[K * K];
  f32x4 bq = {0.f, 0.f, 0.f, 0.f}, dbq = bq;
  if (active) {
#pragma unroll
    for (int k = 0; k < CIN; ++k) wq[k] = *reinterpret_cast<const f32x4*>(a.we + (size_t)k * Cexp + 4 * q);
#pragma unroll
    for (int t = 0; t < K * K; ++t) dwq[t] = *reinterpret_cast<const f32x4*>(a.wd + (size_t)t * Cexp + 4 * q);
    bq = *reinterpret_cast<const f32x4*>(a.be + 4 * q);
    dbq = *reinterpret_cast<const f32x4*>(a.bd + 4 * q);
  }
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f};
  float vmax = 0.f;
  int next_r = 0;                                         // ring rows [0, next_r) (relative to iy_base) are computed
  for (int oy = oy0; oy < oy1; ++oy) {
    const int need = (oy - oy0) * S + K;                  // rows [0, need) must be present
    const int nr = need - next_r;                         // K on the band's first row, S afterwards
    // phase 1: stage the nr new input rows (zeros outside the image).  (Prefetching them one output row ahead through
    // registers was measured and LOSES: +8..44 VGPRs drop the 16-channel kernel from 3 to 2 waves per SIMD, 442 -> 481 us
    // -- the kernel is VALU-bound, co-resident workgroups already cover the load latency.)
    for (int e = tid; e < nr * IW * CQ; e += 256) {
      const int kq = e % CQ, px = (e / CQ) % IW, r = e / (CQ * IW);
      const int iy = iy_base + next_r + r, ix = ix0 + px;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
        v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.x_cs + 4 * kq);
      *reinterpret_cast<f32x4*>(xrow + (size_t)e * 4) = v;                  // [r][px][kq] == e
    }
    __syncthreads();
    // expand the new rows into the ring
    if (active) {
      for (int p = ps; p < nr * IW; p += PP) {
        const int r = p / IW, px = p - r * IW;
        const int rr = next_r + r, iy = iy_base + rr, ix = ix0 + px;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
          acc = bq;
          const float* xp = xrow + (r * IW + px) * CIN;
#pragma unroll
          for (int kq = 0; kq < CQ; ++kq) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + 4 * kq);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = (CRESTE_MB_BISECT & 1) ? fma4s(wq[4 * kq + j], f32x4{xv[j], xv[j], xv[j], xv[j]}, acc)
                                                                  : (CRESTE_MB_BISECT & 4) ? fma4b(wq[4 * kq + j], xv[j], acc) : fma4(wq[4 * kq + j], xv[j], acc);
          }
          acc = swish4(acc);
        }
        *reinterpret_cast<f32x4*>(ring + ((rr % K) * IW + px) * Cexp + 4 * q) = acc;
      }
    }
    next_r = need;
    __syncthreads();
    // depthwise taps of output row oy
    if (active) {
      const int r0 = (oy - oy0) * S;
      for (int ox = ps; ox < TW; ox += PP) {
        if (ox0 + ox >= a.Wo) break;
        f32x4 acc = dbq;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const float* rp = ring + (((r0 + ky) % K) * IW + ox * S) * Cexp + 4 * q;
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
            acc = (CRESTE_MB_BISECT & 2) ? fma4s(*reinterpret_cast<const f32x4*>(rp + kx * Cexp), dwq[ky * K + kx], acc)
                                         : fma4(*reinterpret_cast<const f32x4*>(rp + kx * Cexp), dwq[ky * K + kx], acc);
        }
        acc = swish4(acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(acc[j]));
        ssum += acc;
        *reinterpret_cast<f32x4*>(a.out + ((size_t)(n * a.Ho + oy) * a.Wo + ox0 + ox) * Cexp + 4 * q) = acc;
      }
    }
    // (no barrier here: the next row's staging only touches xrow, which this row's expand phase finished reading
    //  before the barrier above; its ring writes come after the barrier that follows the staging)
  }
  __syncthreads();
  // squeeze-excite partial sums of this workgroup: fixed-order sum over the pixel slices
  if (active) *reinterpret_cast<f32x4*>(red + ps * Cexp + 4 * q) = ssum;
  __syncthreads();
  if (active && ps == 0) {
    f32x4 tot = ssum;
    for (int k = 1; k < PP; ++k) tot += *reinterpret_cast<const f32x4*>(red + k * Cexp + 4 * q);
    *reinterpret_cast<f32x4*>(a.partial + ((size_t)n * a.nstrips * a.nbands + chunk) * Cexp + 4 * q) = tot;
  }
  if (a.amax) {
    __syncthreads();
    block_amax_update(vmax, a.amax, red);
  }
}

struct MbPlan { int TW, nstrips, nbands, rows_per_band; size_t smem; };

static bool mb_plan(int K, int S, int Cin, int Cexp, int N, int Ho, int Wo, MbPlan* p) {
  if (!((K == 3 || K == 5) && (S == 1 || S == 2))) return false;
  if (!(Cin == 16 || Cin == 24 || Cin == 40) || Cexp % 4 || Cexp < 4 || Cexp > 256) return false;
  // strip width: smallest (horizontal halo factor) x (1.15 if the K-row ring leaves room for only one workgroup per CU)
  const int cand[3] = {32, 16, 8};
  int tw = 0;
  size_t smem = 0;
  float best = 1e30f;
  for (int c = 0; c < 3; ++c) {
    if (cand[c] == 32 && S == 2) continue;                 // instantiated: TW 32 for S = 1, TW 16 / 8 for both
    const int iw = (cand[c] - 1) * S + K;
    const size_t b = ((size_t)K * iw * (Cin + Cexp) + (size_t)(256 / (Cexp / 4)) * Cexp) * sizeof(float);
    if (b > 160u * 1024) continue;
    const float cost = (float)iw / (float)(cand[c] * S) * (b <= 80u * 1024 ? 1.f : 1.15f);
    if (cost < best) { best = cost; tw = cand[c]; smem = b; }
  }
  if (!tw) return false;
  p->TW = tw; p->smem = smem;
  p->nstrips = (Wo + tw - 1) / tw;
  int bands = (2048 + p->nstrips * N - 1) / (p->nstrips * N);
  if (bands < 1) bands = 1;
  int rows = (Ho + bands - 1) / bands;
  const int min_rows = (K - S) * 3 / S > 1 ? (K - S) * 3 / S : 1;      // keep the vertical halo <= ~1/3 of a band
  if (rows < min_rows) rows = min_rows;
  p->rows_per_band = rows;
  p->nbands = (Ho + rows - 1) / rows;
  return true;
}


// ---------------------------------------------------------------------------------------------------------------------
// Encoder stem + first MBConv block's depthwise half in one pass: 3x3 / stride 2 conv of the 4-channel RGB-D image
// (+ folded BN + swish; EfficientNet `_conv_stem`, `_bn0`; reference effnet.py:41-44,83) -> depthwise 3x3 / stride 1
// (+ folded BN + swish; block 0 has no expand conv) -> squeeze-excite sums.  Same strip / band / row-ring structure as
// above with the stem conv as the producer of the ring rows: the 32-channel stem output (378 MB at batch 16) never
// reaches HBM, and the stem leaves the fp32-MFMA implicit-GEMM kernel, where its 4-channel input filled 4 of 16
// K-lanes (0.42 ms for 6.8 GFLOP).  Exact fp32 FMAs, taps in (ky, kx, ci) order.
struct StemArgs {
  const float* x; const float* ws; const float* bs; const float* wd; const float* bd;
  float* out; float* partial; float* amax;
  int H, W, C1, H1, W1, pad_t, pad_l, dpad_t, dpad_l, rows_per_band, nbands, nstrips;
};

template <int TW>
__global__ __launch_bounds__(256, 2) void stem_dw_kernel(const StemArgs a) {
  constexpr int K = 3, IW = TW + K - 1, XW = 2 * IW + 1;   // ring row / staged input row, in pixels
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C1 = a.C1, NQ = C1 >> 2, PP = 256 / NQ;
  float* const xin = smem;                                 // [2 * K + 1][XW][4]
  float* const ring = xin + (2 * K + 1) * XW * 4;          // [K][IW][C1]
  float* const red = ring + K * IW * C1;                   // [PP][C1]
  const int tid = threadIdx.x;
  const int q = tid % NQ, ps = tid / NQ;
  const bool active = ps < PP;
  const int n = blockIdx.y;
  const int chunk = xcd_remap(blockIdx.x, a.nstrips * a.nbands);
  const int strip = chunk % a.nstrips, band = chunk / a.nstrips;
  const int ox0 = strip * TW;
  const int oy0 = band * a.rows_per_band, oy1 = min(a.H1, oy0 + a.rows_per_band);
  const int x1_0 = ox0 - a.dpad_l, y1_base = oy0 - a.dpad_t;       // stem column of ring pixel 0 / stem row of ring row 0
  const int ix_0 = 2 * x1_0 - a.pad_l;                             // image column of staged pixel 0

  f32x4 wsq[36], dwq[9];
  f32x4 bq = {0.f, 0.f, 0.f, 0.f}, dbq = bq;
  if (active) {
#pragma unroll
    for (int t = 0; t < 36; ++t) wsq[t] = *reinterpret_cast<const f32x4*>(a.ws + (size_t)t * C1 + 4 * q);
#pragma unroll
    for (int t = 0; t < 9; ++t) dwq[t] = *reinterpret_cast<const f32x4*>(a.wd + (size_t)t * C1 + 4 * q);
    bq = *reinterpret_cast<const f32x4*>(a.bs + 4 * q);
    dbq = *reinterpret_cast<const f32x4*>(a.bd + 4 * q);
  }
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f};
  float vmax = 0.f;
  constexpr int NPRE = (3 * XW + 255) / 256;
  f32x4 pre[NPRE];
  int next_r = 0;
  for (int oy = oy0; oy < oy1; ++oy) {
    const int need = (oy - oy0) + K;
    const int nr = need - next_r;                          // new stem rows: K at the top of the band, then 1
    const int iy_0 = 2 * (y1_base + next_r) - a.pad_t;     // image row of staged row 0
    if (oy == oy0) {                                       // top of the band: 2K+1 image rows, staged directly
      for (int e = tid; e < (2 * nr + 1) * XW; e += 256) {
        const int r = e / XW, px = e - r * XW;
        const int iy = iy_0 + r, ix = ix_0 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(n * a.H + iy) * a.W + ix) * 4);
        *reinterpret_cast<f32x4*>(xin + (size_t)e * 4) = v;
      }
    } else {                                               // afterwards: the 3 rows prefetched during the previous row
#pragma unroll
      for (int j = 0; j < NPRE; ++j)
        if (tid + 256 * j < 3 * XW) *reinterpret_cast<f32x4*>(xin + (size_t)(tid + 256 * j) * 4) = pre[j];
    }
    __syncthreads();
    if (oy + 1 < oy1) {                                    // next row's image rows: in flight during both phases below
      const int iy_n = 2 * (y1_base + need) - a.pad_t;
#pragma unroll
      for (int j = 0; j < NPRE; ++j) {
        const int e = tid + 256 * j, r = e / XW, px = e - r * XW;
        const int iy = iy_n + r, ix = ix_0 + px;
        const bool ok = e < 3 * XW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? a.x + ((size_t)(n * a.H + iy) * a.W + ix) * 4 : a.x);
        pre[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (active) {
      for (int p = ps; p < nr * IW; p += PP) {
        const int r = p / IW, px = p - r * IW;
        const int rr = next_r + r, y1 = y1_base + rr, x1 = x1_0 + px;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)y1 < (unsigned)a.H1 && (unsigned)x1 < (unsigned)a.W1) {
          acc = bq;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const f32x4 xv = *reinterpret_cast<const f32x4*>(xin + ((2 * r + ky) * XW + 2 * px + kx) * 4);
#pragma unroll
              for (int ci = 0; ci < 4; ++ci) acc = fma4(wsq[(ky * 3 + kx) * 4 + ci], xv[ci], acc);
            }
          acc = swish4(acc);
        }
        *reinterpret_cast<f32x4*>(ring + ((rr % K) * IW + px) * C1 + 4 * q) = acc;
      }
    }
    next_r = need;
    __syncthreads();
    if (active) {
      const int r0 = oy - oy0;
      for (int ox = ps; ox < TW; ox += PP) {
        if (ox0 + ox >= a.W1) break;
        f32x4 acc = dbq;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const float* rp = ring + (((r0 + ky) % K) * IW + ox) * C1 + 4 * q;
#pragma unroll
          for (int kx = 0; kx < K; ++kx) acc = fma4(*reinterpret_cast<const f32x4*>(rp + kx * C1), dwq[ky * K + kx], acc);
        }
        acc = swish4(acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(acc[j]));
        ssum += acc;
        *reinterpret_cast<f32x4*>(a.out + ((size_t)(n * a.H1 + oy) * a.W1 + ox0 + ox) * C1 + 4 * q) = acc;
      }
    }
  }
  __syncthreads();
  if (active) *reinterpret_cast<f32x4*>(red + ps * C1 + 4 * q) = ssum;
  __syncthreads();
  if (active && ps == 0) {
    f32x4 tot = ssum;
    for (int k = 1; k < PP; ++k) tot += *reinterpret_cast<const f32x4*>(red + k * C1 + 4 * q);
    *reinterpret_cast<f32x4*>(a.partial + ((size_t)n * a.nstrips * a.nbands + chunk) * C1 + 4 * q) = tot;
  }
  if (a.amax) {
    __syncthreads();
    block_amax_update(vmax, a.amax, red);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Depthwise KxK / stride S conv + folded BN + swish + squeeze-excite sums of the DEEP MBConv blocks (240 .. 1152 channels
// on 76x152 .. 19x38 maps; reference call site effnet.py:83) from an LDS tile: a workgroup owns TH x TW output pixels of
// a group of 8 channel quads, stages the (TH-1)*S+K x (TW-1)*S+K input pixels of those 32 channels once (7-8
// independent quad loads per thread, issued back to back; ONE barrier), then every thread forms four output quads from
// LDS with its quad's K*K weights in registers.  The register-blocked dwconv_se_kernel re-reads each input quad 2.5 .. 10
// times through L1 / L2 and keeps only 2-3 waves per SIMD in flight (100 weight registers at K = 5): it runs these
// layers at 0.7 .. 1.5 TB/s, latency-bound.  (A row-ring variant of this kernel was built and measured: with <= 1
// output per thread per barrier it is barrier-bound and loses.)
struct DwArgs {
  const float* x; const float* wd; const float* bd;
  float* out; float* partial; float* amax;
  int H, W, C, Ho, Wo, pad_t, pad_l, tiles_x, tiles_y, ngroups;
};

template <int K, int S, bool SWISH>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(const DwArgs a) {
  constexpr int TH = S == 1 ? 8 : 4, TW = S == 1 ? 16 : 8;               // output tile
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K, NPX = IH * IW;
  constexpr int NQ = 8, CG = 4 * NQ, PP = 256 / NQ;                      // 8 channel quads x 32 pixel slices
  __shared__ __attribute__((aligned(16))) float tile[NPX * CG];
  __shared__ __attribute__((aligned(16))) float red[PP * CG];
  const int C = a.C, tid = threadIdx.x, q = tid % NQ, ps = tid / NQ;
  const int n = blockIdx.y;
  int id = blockIdx.x;
  const int grp = id % a.ngroups; id /= a.ngroups;
  const int chunk = id, tx = id % a.tiles_x, ty = id / a.tiles_x;
  const int c0 = grp * CG;
  const bool cok = c0 + 4 * q < C;                                        // the last group may be partial
  const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
  for (int e = tid; e < NPX * NQ; e += 256) {
    const int px = e / NQ, qq = e - px * NQ;
    const int py = px / IW, pxx = px - py * IW;
    const int iy = iy0 + py, ix = ix0 + pxx;
    const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && c0 + 4 * qq < C;
    const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? a.x + ((size_t)(n * a.H + iy) * a.W + ix) * C + c0 + 4 * qq : a.x);
    *reinterpret_cast<f32x4*>(tile + (size_t)e * 4) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 dwq[K * K], dbq = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
#pragma unroll
    for (int t = 0; t < K * K; ++t) dwq[t] = *reinterpret_cast<const f32x4*>(a.wd + (size_t)t * C + c0 + 4 * q);
    if (a.bd) dbq = *reinterpret_cast<const f32x4*>(a.bd + c0 + 4 * q);
  }
  __syncthreads();
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f};
  float vmax = 0.f;
  if (cok) {
#pragma unroll
    for (int j = 0; j < TH * TW / PP; ++j) {
      const int o = ps + PP * j, oyl = o / TW, oxl = o - oyl * TW;
      const int oy = oy0 + oyl, ox = ox0 + oxl;
      if (oy >= a.Ho || ox >= a.Wo) continue;
      f32x4 acc = dbq;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const float* rp = tile + ((oyl * S + ky) * IW + oxl * S) * CG + 4 * q;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc = fma4(*reinterpret_cast<const f32x4*>(rp + kx * CG), dwq[ky * K + kx], acc);
      }
      if (SWISH) acc = swish4(acc);
#pragma unroll
      for (int i = 0; i < 4; ++i) vmax = fmaxf(vmax, fabsf(acc[i]));
      ssum += acc;
      *reinterpret_cast<f32x4*>(a.out + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * C + c0 + 4 * q) = acc;
    }
  }
  if (!a.partial && !a.amax) return;                                       // plain depthwise conv (training path)
  *reinterpret_cast<f32x4*>(red + ps * CG + 4 * q) = ssum;
  __syncthreads();
  if (a.partial && ps == 0 && cok) {
    f32x4 tot = ssum;
    for (int k = 1; k < PP; ++k) tot += *reinterpret_cast<const f32x4*>(red + k * CG + 4 * q);
    *reinterpret_cast<f32x4*>(a.partial + ((size_t)n * a.tiles_x * a.tiles_y + chunk) * C + c0 + 4 * q) = tot;
  }
  if (a.amax) {
    __syncthreads();
    block_amax_update(vmax, a.amax, red);
  }
}

constexpr int STEM_TW = 62;                                // ring rows of 64 pixels: two full rounds of 32 pixel slices

static bool stem_plan(int N, int H1, int W1, int C1, MbPlan* p) {
  if (C1 % 4 || C1 < 4 || C1 > 64 || 256 % (C1 / 4)) return false;
  constexpr int IW = STEM_TW + 2, XW = 2 * IW + 1;
  p->TW = STEM_TW;
  p->smem = ((size_t)7 * XW * 4 + (size_t)3 * IW * C1 + (size_t)(256 / (C1 / 4)) * C1) * sizeof(float);
  p->nstrips = (W1 + STEM_TW - 1) / STEM_TW;
  int bands = (2048 + p->nstrips * N - 1) / (p->nstrips * N);
  int rows = (H1 + bands - 1) / bands;
  if (rows < 6) rows = 6;
  p->rows_per_band = rows;
  p->nbands = (H1 + rows - 1) / rows;
  return true;
}

}  // namespace creste

using namespace creste;

extern "C" int creste_mbconv_partial_count(int N, int Ho, int Wo, int Cin, int Cexp, int K, int stride) {
  MbPlan p;
  if (N <= 0 || Ho <= 0 || Wo <= 0 || !mb_plan(K, stride, Cin, Cexp, N, Ho, Wo, &p)) return -1;
  return p.nstrips * p.nbands;
}

template <int K, int S, int CIN, int TW>
static int launch_mb(const MbArgs& a, int N, const MbPlan& p, hipStream_t s) {
  static std::atomic<uint64_t> devs1{0}, devs2{0};
  const dim3 grid(p.nstrips * p.nbands, N);
  // K = 5 with 40 input channels holds 260 weight registers per thread: one workgroup per CU whatever the LDS use
  // (the two-per-CU register budget would spill 256 B per lane)
  constexpr bool kHeavy = K == 5 && CIN == 40;
  if (!kHeavy && p.smem <= 80 * 1024) {
    constexpr int kMinB = kHeavy ? 1 : 2;                  // (keeps the spilling variant out of the binary)
    if (p.smem > 64 * 1024) CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(mbconv_expand_dw_kernel<K, S, CIN, TW, kMinB>), (int)p.smem, devs2));
    mbconv_expand_dw_kernel<K, S, CIN, TW, kMinB><<<grid, 256, p.smem, s>>>(a);
  } else {
    if (p.smem > 64 * 1024) CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(mbconv_expand_dw_kernel<K, S, CIN, TW, 1>), (int)p.smem, devs1));
    mbconv_expand_dw_kernel<K, S, CIN, TW, 1><<<grid, 256, p.smem, s>>>(a);
  }
  CRESTE_CHECK_LAUNCH("mbconv_expand_dw");
  return CRESTE_OK;
}

template <int K, int S, int CIN>
static int launch_mb_tw(const MbArgs& a, int N, const MbPlan& p, hipStream_t s) {
  if constexpr (S == 1) {
    if (p.TW == 32) return launch_mb<K, S, CIN, 32>(a, N, p, s);
  }
  if (p.TW == 16) return launch_mb<K, S, CIN, 16>(a, N, p, s);
  return launch_mb<K, S, CIN, 8>(a, N, p, s);
}

template <int CIN>
static int launch_mb_ks(const MbArgs& a, int N, int K, int S, const MbPlan& p, hipStream_t s) {
  if (K == 3 && S == 1) return launch_mb_tw<3, 1, CIN>(a, N, p, s);
  if (K == 3) return launch_mb_tw<3, 2, CIN>(a, N, p, s);
  if (S == 1) return launch_mb_tw<5, 1, CIN>(a, N, p, s);
  return launch_mb_tw<5, 2, CIN>(a, N, p, s);
}

extern "C" int creste_mbconv_expand_dw_f32(const float* x, int N, int H, int W, int Cin, int x_cs, const float* w_expand,
                                           const float* b_expand, const float* w_dw, const float* b_dw, float* out,
                                           float* partial, float* out_amax, int Cexp, int Ho, int Wo, int K, int stride,
                                           int pad_t, int pad_l, void* stream) {
  CRESTE_REQUIRE(x && w_expand && b_expand && w_dw && b_dw && out && partial, "mbconv_expand_dw: null pointer");
  CRESTE_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && N < 65536, "mbconv_expand_dw: bad dims");
  CRESTE_REQUIRE(x_cs % 4 == 0 && x_cs >= Cin && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
                 "mbconv_expand_dw: input slice must be 16-byte aligned with a channel stride that is a multiple of 4");
  CRESTE_REQUIRE(pad_t >= 0 && pad_l >= 0 && pad_t < K && pad_l < K, "mbconv_expand_dw: bad padding");
  MbPlan p;
  CRESTE_REQUIRE(mb_plan(K, stride, Cin, Cexp, N, Ho, Wo, &p),
                 "mbconv_expand_dw: not built for K=%d stride=%d Cin=%d Cexp=%d (K 3|5, stride 1|2, Cin 16|24|40, Cexp <= 256)",
                 K, stride, Cin, Cexp);
  CRESTE_REQUIRE((long)(Ho - 1) * stride - pad_t + K - 1 < H + K && (long)(Wo - 1) * stride - pad_l + K - 1 < W + K,
                 "mbconv_expand_dw: output extent exceeds the padded input");
  MbArgs a{x, w_expand, b_expand, w_dw, b_dw, out, partial, out_amax, H, W, x_cs, Cexp, Ho, Wo, pad_t, pad_l,
           p.rows_per_band, p.nbands, p.nstrips};
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 16) return launch_mb_ks<16>(a, N, K, stride, p, s);
  if (Cin == 24) return launch_mb_ks<24>(a, N, K, stride, p, s);
  return launch_mb_ks<40>(a, N, K, stride, p, s);
}

extern "C" int creste_stem_dw_partial_count(int N, int H1, int W1, int C1) {
  MbPlan p;
  if (N <= 0 || H1 <= 0 || W1 <= 0 || !stem_plan(N, H1, W1, C1, &p)) return -1;
  return p.nstrips * p.nbands;
}

extern "C" int creste_stem_dw_f32(const float* x, int N, int H, int W, const float* w_stem, const float* b_stem, int pad_t,
                                  int pad_l, const float* w_dw, const float* b_dw, int dpad_t, int dpad_l, float* out,
                                  float* partial, float* out_amax, int C1, int H1, int W1, void* stream) {
  CRESTE_REQUIRE(x && w_stem && b_stem && w_dw && b_dw && out && partial, "stem_dw: null pointer");
  CRESTE_REQUIRE(N > 0 && N < 65536 && H > 0 && W > 0 && H1 > 0 && W1 > 0, "stem_dw: bad dims");
  CRESTE_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "stem_dw: the 4-channel NHWC image must be 16-byte aligned");
  CRESTE_REQUIRE(pad_t >= 0 && pad_t < 3 && pad_l >= 0 && pad_l < 3 && dpad_t >= 0 && dpad_t < 3 && dpad_l >= 0 && dpad_l < 3,
                 "stem_dw: bad padding");
  CRESTE_REQUIRE(2L * (H1 - 1) - pad_t < H && 2L * (W1 - 1) - pad_l < W, "stem_dw: stem output extent exceeds the image");
  MbPlan p;
  CRESTE_REQUIRE(stem_plan(N, H1, W1, C1, &p), "stem_dw: %d stem channels not built (multiple of 4 dividing 1024, <= 64)", C1);
  StemArgs a{x, w_stem, b_stem, w_dw, b_dw, out, partial, out_amax, H, W, C1, H1, W1, pad_t, pad_l, dpad_t, dpad_l,
             p.rows_per_band, p.nbands, p.nstrips};
  static std::atomic<uint64_t> stem_devs{0};
  if (p.smem > 64 * 1024)      // C1 = 64: 67.7 KB of dynamic LDS needs the per-device opt-in (the shipped stem, C1 = 32, does not)
    CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(stem_dw_kernel<STEM_TW>), (int)p.smem, stem_devs));
  stem_dw_kernel<STEM_TW><<<dim3(p.nstrips * p.nbands, N), 256, p.smem, (hipStream_t)stream>>>(a);
  CRESTE_CHECK_LAUNCH("stem_dw");
  return CRESTE_OK;
}

static bool dw_tile_dims(int K, int S, int C, int Ho, int Wo, int* tx, int* ty) {
  if (!((K == 3 || K == 5) && (S == 1 || S == 2)) || C % 4 || C < 4 || Ho <= 0 || Wo <= 0) return false;
  const int th = S == 1 ? 8 : 4, tw = S == 1 ? 16 : 8;
  *tx = (Wo + tw - 1) / tw; *ty = (Ho + th - 1) / th;
  return true;
}

extern "C" int creste_dwconv_se_tile_partial_count(int Ho, int Wo, int C, int K, int stride) {
  int tx, ty;
  return dw_tile_dims(K, stride, C, Ho, Wo, &tx, &ty) ? tx * ty : -1;
}

template <bool SWISH>
static int launch_dw_tile(const DwArgs& a, int N, int K, int stride, hipStream_t s) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.ngroups), N);
  if (K == 3 && stride == 1) dwconv_tile_kernel<3, 1, SWISH><<<grid, 256, 0, s>>>(a);
  else if (K == 3) dwconv_tile_kernel<3, 2, SWISH><<<grid, 256, 0, s>>>(a);
  else if (stride == 1) dwconv_tile_kernel<5, 1, SWISH><<<grid, 256, 0, s>>>(a);
  else dwconv_tile_kernel<5, 2, SWISH><<<grid, 256, 0, s>>>(a);
  CRESTE_CHECK_LAUNCH("dwconv_tile");
  return CRESTE_OK;
}

static int run_dw_tile(const float* in, const float* w, const float* bias, float* out, float* partial, float* out_amax,
                       int N, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad_t, int pad_l, bool swish,
                       void* stream) {
  CRESTE_REQUIRE(in && w && out, "dwconv_tile: null pointer");
  CRESTE_REQUIRE(N > 0 && N < 65536 && H > 0 && W > 0, "dwconv_tile: bad dims");
  CRESTE_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && pad_t >= 0 && pad_l >= 0 && pad_t < K && pad_l < K,
                 "dwconv_tile: input must be 16-byte aligned, padding below the kernel size");
  int tx, ty;
  CRESTE_REQUIRE(dw_tile_dims(K, stride, C, Ho, Wo, &tx, &ty), "dwconv_tile: not built for C=%d K=%d stride=%d", C, K, stride);
  const int ngroups = (C / 4 + 7) / 8;
  CRESTE_REQUIRE((long)tx * ty * ngroups < (1L << 31), "dwconv_tile: grid too large");
  DwArgs a{in, w, bias, out, partial, out_amax, H, W, C, Ho, Wo, pad_t, pad_l, tx, ty, ngroups};
  return swish ? launch_dw_tile<true>(a, N, K, stride, (hipStream_t)stream)
               : launch_dw_tile<false>(a, N, K, stride, (hipStream_t)stream);
}

extern "C" int creste_dwconv_se_tile_f32(const float* in, const float* w, const float* bias, float* out, float* partial,
                                         float* out_amax, int N, int H, int W, int C, int Ho, int Wo, int K, int stride,
                                         int pad_t, int pad_l, void* stream) {
  CRESTE_REQUIRE(bias && partial, "dwconv_se_tile: null pointer");
  return run_dw_tile(in, w, bias, out, partial, out_amax, N, H, W, C, Ho, Wo, K, stride, pad_t, pad_l, true, stream);
}

extern "C" int creste_dwconv_tile_f32(const float* in, const float* w, const float* bias, float* out, int N, int H, int W,
                                      int C, int Ho, int Wo, int K, int stride, int pad_t, int pad_l, int act, void* stream) {
  CRESTE_REQUIRE(act == CRESTE_ACT_NONE || act == CRESTE_ACT_SWISH, "dwconv_tile: activation %d not built (none, swish)", act);
  return run_dw_tile(in, w, bias, out, nullptr, nullptr, N, H, W, C, Ho, Wo, K, stride, pad_t, pad_l, act == CRESTE_ACT_SWISH,
                     stream);
}
