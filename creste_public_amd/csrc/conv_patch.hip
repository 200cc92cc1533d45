// Stride-1 1x1 / 3x3 convolution on the bf16 matrix cores with an LDS-resident halo patch.
//
// Why a second conv engine: on gfx950 the bf16 MFMA (v_mfma_f32_32x32x16_bf16) runs at 16x the rate of
// the fp32 MFMA.  Two operand precisions are built on it:
//   BF16X3  every fp32 operand is split into bf16 hi + bf16 lo (16 mantissa bits) and a product is
//           formed as hi*hi + hi*lo + lo*hi with fp32 accumulation -- product error ~2^-16 relative,
//           i.e. the same order as fp32 accumulation round-off over K~4000 -- at 16/3 = 5.3x the fp32
//           MFMA rate ("fp32-grade" mode);
//   BF16    one bf16 product (throughput mode).
// Activations stay fp32 NHWC in HBM; the split happens while staging into LDS.  Weights are split and
// laid out once at pack time.
//
// Tile: one workgroup (8 waves) computes an 8 x 32 pixel patch x BN (128 or 64) output channels.
// Per 16-channel chunk the (8+K-1) x (32+K-1) input halo patch is staged ONCE in LDS and reused by all
// K*K taps (the 3x3 conv's 9 shifted reads hit LDS, not L2/HBM: L2-miss traffic drops to the
// algorithmic bytes x 1.33 halo overhead); the per-tap weight tile streams from L2 (8 KB per tap).
// LDS image of both operands: [plane hi|lo][k-octet][row (pixel or channel)][8 bf16 = 16 B].  A lane's
// MFMA fragment (8 consecutive k of one row) is one ds_read_b128; rows are 16 B apart and the k-octet
// planes are a multiple of 256 B apart, so the 16-lane service groups of ds_read_b128 (which mix both
// octets but always cover 16 row indices distinct mod 16) are conflict-free for any tap shift.
#include "common.h"
#include <stdlib.h>

// (nontemporal output stores were measured here and LOSE: f16x3 unchanged, bf16x6 -4 %, bf16 -18 % -- the next layer finds
// part of the map in the 256 MB Infinity Cache, which streaming stores bypass; plain stores it is)
#define CRESTE_OUT_STORE(ptr, v) (*(ptr) = (v))

namespace creste {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PT_TH = 8, PT_TW = 32;   // output pixel patch per workgroup
constexpr int PT_CK = 16;              // channels per chunk = K of one MFMA
#ifndef PATCH_WAVES_PER_SIMD
#define PATCH_WAVES_PER_SIMD 4        // two 8-wave workgroups per CU: one block's barrier hides behind the other
#endif

// sum over piece pairs (pa, pb) with pa + pb < SPLIT, smallest magnitude first:
//   SPLIT 1: a0*b0                       (bf16)
//   SPLIT 2: a1*b0 + a0*b1 + a0*b0       (bf16x3: 16-bit operands)
//   SPLIT 3: a2*b0 + a1*b1 + a0*b2 + a1*b0 + a0*b1 + a0*b0   (bf16x6: 24-bit operands = fp32; the dropped
//            pairs are <= 2^-24 relative, the size of one fp32 product rounding)
//
// F16X3 is SPLIT 2 on fp16 pieces (v_mfma_f32_32x32x16_f16, same rate): hi + lo carry 22 significand bits
// instead of 16, so the three products are fp32-grade (<= 2^-21) at HALF the MFMA count of bf16x6.  fp16 has
// only 5 exponent bits, hence the exact power-of-two rescaling of both operands (activations: per tensor
// from a running |max| the producing kernel maintains; weights: per output channel at pack time), undone in
// the epilogue.
template <bool F16> struct Piece { typedef __bf16 T; typedef bf16x8 V8; typedef bf16x4 V4; };
template <> struct Piece<true> { typedef _Float16 T; typedef f16x8 V8; typedef f16x4 V4; };

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int SPLIT, typename V8>
__device__ __forceinline__ f32x16 split_mfma(const V8 (&a)[SPLIT], const V8 (&b)[SPLIT], f32x16 c) {
#pragma unroll
  for (int order = SPLIT - 1; order >= 0; --order)
#pragma unroll
    for (int pa = order; pa >= 0; --pa) c = mfma16(a[pa], b[order - pa], c);
  return c;
}

struct PatchArgs {
  const float* in;
  const char* wpk;
  const float* bias;
  const float* res;
  const float* a_scale;
  const float* row_mask;
  float* out;
  const float* a_amax;       // F16X3: device upper bound of |in|
  float* out_amax;           // running max |out| (any precision), or nullptr
  const float* w_unscale;    // F16X3: [Cout] inverse weight scale
  int N, H, W, Cin, in_cs;
  int Ho, Wo, Cout, out_cs, out_co, res_cs;
  int pad_t, pad_l;
  int act;
  int nchunk;
  int tiles_x, tiles_y, tiles_n;
  int gate_hw;               // > 0: pixels were re-tiled as one flat image; a_scale row = linear pixel / gate_hw
  long flat_P;               // > 0: conv1x1_deep_kernel's flat tiling -- the N*H*W pixels as one image of width 32, flat_P of them real
  float* stats;              // creste_conv_desc.out_stats ([pixel tiles][2][Cout]; 1x1 bf16-split kernels only), or nullptr
};

// Epilogue.  The MFMAs are issued with the WEIGHT fragment as the first operand, so D = W * X^T: in the
// 32x32 C/D layout the lane indexes the pixel (col = lane&31) and the registers index channels
// (row = (r&3) + 8*(r>>2) + 4*(lane>>5)): registers 4g..4g+3 are 4 CONSECUTIVE output channels of one
// pixel -> one 16-byte store (and one 16-byte residual / bias load) instead of four dword stores; the
// store tail of a conv is issue-bound, not bandwidth-bound (cdna_hip_programming.md T21).
template <int TN, bool F16, bool STATS = false, int MT = 2>
__device__ __forceinline__ void patch_epilogue(const f32x16 (&acc)[MT][TN], const PatchArgs& p, int img,
                                               int oy0, int ox0, int nbase, int wm, int wn, int li, int lh,
                                               float o_mul, float* scratch, int stat_row = 0) {
  static_assert(!STATS || MT == 2, "statistics epilogue: 8 x 32 pixel tiles");
  float vmax = 0.f;
  f32x4 st1[TN][4], st2[TN][4];                    // STATS: this lane's pixel(s), per channel quad of its registers
  if constexpr (STATS) {
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) st1[nt][g] = st2[nt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool vec_ok = (p.Cout & 3) == 0 && (p.out_cs & 3) == 0 && (p.out_co & 3) == 0 &&
                      (!p.res || (p.res_cs & 3) == 0);
  const int ox = ox0 + li;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int oy = oy0 + wm * MT + mt;
    if (oy >= p.Ho || ox >= p.Wo) continue;
    const long m = ((long)img * p.Ho + oy) * p.Wo + ox;
    if (p.flat_P && m >= p.flat_P) continue;             // flat tiling (conv1x1_deep_kernel): the last row of 32 pixels may be partial
    const float rmask = p.row_mask ? p.row_mask[m] : 1.f;
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nbase + (wn * TN + nt) * 32 + 8 * g + 4 * lh;
        if (n >= p.Cout) continue;
        f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
        if (vec_ok) {
          if (F16) v *= *reinterpret_cast<const f32x4*>(p.w_unscale + n) * o_mul;   // exact: powers of two
          if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
          if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + m * p.res_cs + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = act_apply(v[j], p.act) * rmask;
            vmax = fmaxf(vmax, fabsf(v[j]));
          }
          CRESTE_OUT_STORE(reinterpret_cast<f32x4*>(p.out + m * p.out_cs + p.out_co + n), v);
          if constexpr (STATS) { st1[nt][g] += v; st2[nt][g] += v * v; }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (n + j < p.Cout) {
              float x = v[j];
              if (F16) x *= p.w_unscale[n + j] * o_mul;
              x += (p.bias ? p.bias[n + j] : 0.f);
              if (p.res) x += p.res[m * p.res_cs + n + j];
              x = act_apply(x, p.act) * rmask;
              vmax = fmaxf(vmax, fabsf(x));
              p.out[m * p.out_cs + p.out_co + n + j] = x;
            }
          }
        }
      }
    }
  }
  if constexpr (STATS) {
    if (p.stats) {
      // channel sums over the tile's 8 x 32 pixels in a fixed order: the 32 lanes of a half-wave (pixels of a row) by xor
      // shuffles, then the four row-pair waves through LDS (scratch: [4 wm][2 wn][TN][4 g][2 lh][8] floats)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int o = 1; o < 32; o <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              st1[nt][g][j] += __shfl_xor(st1[nt][g][j], o);
              st2[nt][g][j] += __shfl_xor(st2[nt][g][j], o);
            }
      __syncthreads();                               // the operand buffers `scratch` aliases are free
      if (li == 0) {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float* dst = scratch + ((((wm * 2 + wn) * TN + nt) * 4 + g) * 2 + lh) * 8;
            *reinterpret_cast<f32x4*>(dst) = st1[nt][g];
            *reinterpret_cast<f32x4*>(dst + 4) = st2[nt][g];
          }
      }
      __syncthreads();
      // thread = (k, channel of the tile): n = (wn * TN + nt) * 32 + 8 g + 4 lh + j
      for (int i = threadIdx.x; i < 2 * 64 * TN; i += blockDim.x) {
        const int k = i / (64 * TN), cn = i - k * (64 * TN);
        const int wn2 = cn / (32 * TN), r = cn - wn2 * (32 * TN), nt = r >> 5, g = (r >> 3) & 3, lh2 = (r >> 2) & 1, j = r & 3;
        const int n = nbase + cn;
        if (n < p.Cout) {
          float a = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) a += scratch[((((w * 2 + wn2) * TN + nt) * 4 + g) * 2 + lh2) * 8 + k * 4 + j];
          p.stats[((size_t)stat_row * 2 + k) * p.Cout + n] = a;
        }
      }
      __syncthreads();
    }
  }
  if (p.out_amax) block_amax_update(vmax, p.out_amax, scratch);
}

// Epilogue through LDS for the write-bound 1x1 convs (MBConv expand / project, heads): every 32 pixel x 32 channel
// accumulator tile is transposed in a per-wave LDS tile so that EIGHT lanes write 128 contiguous bytes of one pixel
// (and read bias / residual the same way) -- the register epilogue above stores 16 bytes per lane at a pixel stride,
// 64 separate memory transactions per instruction, and the store tail of a K = 16..48 conv IS the kernel.
// `tile`: 32 rows of 36 floats per wave (144-byte rows: conflict-free for both access patterns).
constexpr int EPI_ROW = 36;
template <int TN>
__device__ __forceinline__ void patch_epilogue_lds(const f32x16 (&acc)[2][TN], const PatchArgs& p, int img, int oy0,
                                                   int ox0, int nbase, int wm, int wn, int lane, float o_mul,
                                                   float* smem_f) {
  const int li = lane & 31, lh = lane >> 5;
  const int wave = wm * 2 + wn;
  float* tile = smem_f + wave * (32 * EPI_ROW);
  float* scratch = smem_f + 8 * 32 * EPI_ROW;
  const int rp = lane >> 3, rq = lane & 7;                     // read-back: pixel row (of 8) and channel quad
  float vmax = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int oy = oy0 + wm * 2 + mt;
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(tile + li * EPI_ROW + 8 * g + 4 * lh) =
            f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
      __syncthreads();
      const int n = nbase + (wn * TN + nt) * 32 + rq * 4;
      if (oy < p.Ho && n < p.Cout) {
        const f32x4 us = p.w_unscale ? *reinterpret_cast<const f32x4*>(p.w_unscale + n) * o_mul        // exact: powers of two
                                     : f32x4{o_mul, o_mul, o_mul, o_mul};
        const f32x4 bs = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int px = rp + 8 * k, ox = ox0 + px;
          if (ox >= p.Wo) continue;
          const long m = ((long)img * p.Ho + oy) * p.Wo + ox;
          f32x4 v = *reinterpret_cast<const f32x4*>(tile + px * EPI_ROW + rq * 4) * us + bs;
          if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + m * p.res_cs + n);
          const float rmask = p.row_mask ? p.row_mask[m] : 1.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = act_apply(v[j], p.act) * rmask;
            vmax = fmaxf(vmax, fabsf(v[j]));
          }
          CRESTE_OUT_STORE(reinterpret_cast<f32x4*>(p.out + m * p.out_cs + p.out_co + n), v);
        }
      }
      __syncthreads();
    }
  }
  if (p.out_amax) block_amax_update(vmax, p.out_amax, scratch);
}

// FG: the squeeze-excite gate row is looked up per staged PIXEL (flat re-tiling of a gated 1x1 conv, gate_hw > 0) -- its
// own instantiation: the per-round gate registers pushed the common 128-register kernels into scratch
// ST: the epilogue also keeps per-channel sums of what it writes (PatchArgs.stats, training) -- its own instantiation, the
// sums' registers would push the inference kernels into scratch
template <int K, int SPLIT, int TN, bool F16, bool FG = false, bool ST = false>
__global__ __launch_bounds__(512, (SPLIT * TN >= 6 || (ST && TN == 2)) ? 2 : PATCH_WAVES_PER_SIMD) void conv_patch_kernel(const PatchArgs p) {
  typedef typename Piece<F16>::V8 V8;
  typedef typename Piece<F16>::V4 V4;
  constexpr int T = K * K;
  constexpr int PH = PT_TH + K - 1, PW = PT_TW + K - 1, NPIX = PH * PW;
  constexpr int NPIXP = (NPIX + 15) / 16 * 16;
  constexpr int BN = 64 * TN;
  constexpr int A_OCT = NPIXP * 16;             // bytes of one k-octet plane of A
  constexpr int A_PLANE = 2 * A_OCT;            // hi (or lo) plane
  constexpr int A_BYTES = SPLIT * A_PLANE;
  // weights are packed in UNITS of 64 output channels ([unit][chunk][tap][plane][k-octet][64][8 elements]); a
  // workgroup stages TN consecutive units, so the tile width is a LAUNCH-time choice (patch_tn) on one packed image
  constexpr int U_OCT = 64 * 16, U_PLANE = 2 * U_OCT, U_BYTES = SPLIT * U_PLANE, U_INSTR = U_BYTES / 1024;
  constexpr int B_BYTES = TN * U_BYTES;
  constexpr int NF4 = NPIX * 4;                 // float4 loads per chunk of the A patch
  constexpr int ROUNDS = (NF4 + 511) / 512;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const abase = smem;                    // two A buffers, then two B buffers
  char* const bbase = smem + 2 * A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;      // 4 (pixel rows) x 2 (channel halves)
  const int li = lane & 31, lh = lane >> 5;

  // tile decode: channel tile fastest, then x, y, image; contiguous runs per XCD
  const int nblk = p.tiles_n * p.tiles_x * p.tiles_y * p.N;
  int id = xcd_remap(blockIdx.x, nblk);
  const int tn = id % p.tiles_n; id /= p.tiles_n;
  const int tx = id % p.tiles_x; id /= p.tiles_x;
  const int ty = id % p.tiles_y;
  const int img = id / p.tiles_y;
  const int oy0 = ty * PT_TH, ox0 = tx * PT_TW;

  float a_mul = 1.f, o_mul = 1.f;
  if (F16) a_mul = f16_operand_scale(*p.a_amax, &o_mul);

  // ---- per-thread A staging slots (fixed over chunks).  512 % 4 == 0, so a thread always carries the same
  // channel quad cq of consecutive patch pixels pix = r*128 + tid/4.
  const int cq = tid & 3;
  const int a_lofs0 = (cq >> 1) * A_OCT + (tid >> 2) * 16 + (cq & 1) * 8;   // + r*128*16 per round
  int a_gpix[ROUNDS];      // pixel index (n*H + y)*W + x of the round's patch pixel, -1 if outside / unused
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int pix = r * 128 + (tid >> 2);
    const int py = pix / PW, px = pix % PW;
    const int iy = oy0 + py - p.pad_t, ix = ox0 + px - p.pad_l;
    const bool ok = pix < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    a_gpix[r] = ok ? (img * p.H + iy) * p.W + ix : -1;
  }
  // weight tile of step g: U_BYTES contiguous bytes per unit, copied by LDS-DMA in 1 KiB pieces
  const size_t nsteps_w = (size_t)p.nchunk * T;
  const char* wbase = p.wpk + (size_t)tn * TN * nsteps_w * U_BYTES;
  constexpr int B_INSTR = B_BYTES / 1024;
  auto dma_b = [&](int g) __attribute__((always_inline)) {
    char* dst = bbase + (g & 1) * B_BYTES;
#pragma unroll
    for (int j = 0; j < (B_INSTR + 7) / 8; ++j) {
      const int i = wave + 8 * j;
      if (i < B_INSTR) {
        const int u = i / U_INSTR, r = i % U_INSTR;
        const char* src = wbase + ((size_t)u * nsteps_w + g) * U_BYTES + r * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      }
    }
  };

  // BRANCH-FREE prefetch: a halo / channel-tail element reads the tensor's first quad and is zeroed when the value
  // is converted (a predicated load compiles to a branch with `s_waitcnt vmcnt(0)` behind it: the whole load
  // latency -- and that of the weight DMA issued before it -- sat in front of the MFMAs of every step); the operand
  // scale, the squeeze-excite gate and the zeroing are applied after the MFMAs, in store_a
  auto load_a = [&](int r, int c) __attribute__((always_inline)) -> f32x4 {
    const int ch = c * PT_CK + cq * 4;
    const bool ok = a_gpix[r] >= 0 && ch < p.Cin;
    return *reinterpret_cast<const f32x4*>(ok ? p.in + (size_t)a_gpix[r] * p.in_cs + ch : p.in);
  };
  // squeeze-excite gate of the chunk being staged: one row of a_scale per IMAGE -- the workgroup's image, or (FG) the
  // image of each staged pixel
  constexpr int NG = FG ? ROUNDS : 1;
  f32x4 gate[NG];
#pragma unroll
  for (int r = 0; r < NG; ++r) gate[r] = f32x4{1.f, 1.f, 1.f, 1.f};
  auto load_gate = [&](int c) __attribute__((always_inline)) {
    const int ch = c * PT_CK + cq * 4, chc = ch < p.Cin ? ch : 0;
    if constexpr (FG) {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int im = a_gpix[r] >= 0 ? a_gpix[r] / p.gate_hw : 0;
        gate[r] = *reinterpret_cast<const f32x4*>(p.a_scale + (size_t)im * p.Cin + chc);
      }
    } else {
      gate[0] = *reinterpret_cast<const f32x4*>(p.a_scale + (size_t)img * p.Cin + chc);
    }
  };
  auto store_a = [&](int r, int c, f32x4 v, char* buf) __attribute__((always_inline)) {
    if (r * 128 + (tid >> 2) >= NPIX) return;
    char* dst = buf + a_lofs0 + r * (128 * 16);
    const bool ok = a_gpix[r] >= 0 && c * PT_CK + cq * 4 < p.Cin;
    if (p.a_scale) v *= gate[FG ? r : 0];
    if (F16) v *= a_mul;
    if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 rem = v;
#pragma unroll
    for (int pl = 0; pl < SPLIT; ++pl) {           // hi, then the bf16 of what is left, ...
      const V4 piece = __builtin_convertvector(rem, V4);
      *reinterpret_cast<V4*>(dst + pl * A_PLANE) = piece;
      if (pl + 1 < SPLIT) rem -= __builtin_convertvector(piece, f32x4);
    }
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: chunk 0 of A, weight tile (0,0)
  if (p.a_scale) load_gate(0);
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) store_a(r, 0, load_a(r, 0), abase);
  dma_b(0);
  __syncthreads();

  const int nsteps = p.nchunk * T;
  int step = 0;
  for (int c = 0; c < p.nchunk; ++c) {
    const char* A = abase + (c & 1) * A_BYTES;
    char* Anext = abase + ((c + 1) & 1) * A_BYTES;
    const bool more_a = c + 1 < p.nchunk;
#pragma unroll
    for (int t = 0; t < T; ++t, ++step) {
      const char* B = bbase + (step & 1) * B_BYTES;
      const bool more_b = step + 1 < nsteps;
      f32x4 ra[ROUNDS];
      if (more_b) dma_b(step + 1);
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r)
        if ((r < T ? r : T - 1) == t && more_a) ra[r] = load_a(r, c + 1);
      if (t == 0 && more_a && p.a_scale) load_gate(c + 1);

      // ---- MFMAs of (chunk c, tap t)
      const int ky = t / K, kx = t % K;
      V8 af[2][SPLIT], bfr[TN][SPLIT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int idx = (wm * 2 + mt + ky) * PW + kx + li;
#pragma unroll
        for (int pl = 0; pl < SPLIT; ++pl)
          af[mt][pl] = *reinterpret_cast<const V8*>(A + pl * A_PLANE + lh * A_OCT + idx * 16);
      }
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const int n = (wn * TN + nt) * 32 + li;
#pragma unroll
        for (int pl = 0; pl < SPLIT; ++pl)
          bfr[nt][pl] = *reinterpret_cast<const V8*>(B + (n >> 6) * U_BYTES + pl * U_PLANE + lh * U_OCT + (n & 63) * 16);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          acc[mt][nt] = split_mfma<SPLIT, V8>(bfr[nt], af[mt], acc[mt][nt]);
        }

      // ---- land the prefetched tiles in the other buffers
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r)
        if ((r < T ? r : T - 1) == t && more_a) store_a(r, c + 1, ra[r], Anext);
      __syncthreads();
    }
  }

  if constexpr (K == 1 && F16) {
    // all waves are past the last barrier of the main loop: the operand buffers are free for the transpose tiles
    const bool vec_ok = (p.Cout & 3) == 0 && (p.out_cs & 3) == 0 && (p.out_co & 3) == 0 && (!p.res || (p.res_cs & 3) == 0);
    if (vec_ok) {
      patch_epilogue_lds<TN>(acc, p, img, oy0, ox0, tn * BN, wm, wn, lane, o_mul, reinterpret_cast<float*>(smem));
      return;
    }
  }
  if constexpr (ST) {
    static_assert(K == 1 && !F16 && TN <= 2, "statistics epilogue: 1x1 convs of the bf16 split modes, tiles of <= 128 channels");
    __syncthreads();                                   // (the statistics' LDS scratch aliases the operand buffers)
    patch_epilogue<TN, F16, true>(acc, p, img, oy0, ox0, tn * BN, wm, wn, li, lh, o_mul, reinterpret_cast<float*>(smem),
                                  (img * p.tiles_y + ty) * p.tiles_x + tx);
  } else {
    patch_epilogue<TN, F16>(acc, p, img, oy0, ox0, tn * BN, wm, wn, li, lh, o_mul, reinterpret_cast<float*>(smem));
  }
}

// ---------------------------------------------------------------------------------------------
// 1x1 convs whose launch has FEW workgroups (the MBConv expand / project convs on the 19 x 38 and 38 x 76 maps: 23 .. 180 pixel
// tiles on 256 CUs): at most one workgroup per CU, so nothing hides a step's load latency -- conv_patch_kernel<1, ...> prefetches
// ONE 16-channel chunk, i.e. every step of the K loop waits a full L2 / HBM round trip (72 steps for the 1152-channel project
// convs).  Here chunk c + D is requested while chunk c is multiplied: a ring of D = 4 chunks of the A tile, of the squeeze-excite
// gate and of the weight tile in REGISTERS (plain loads only: the compiler counts vmcnt exactly for in-order returns; beside an
// LDS-DMA it would wait for vmcnt(0)), landed in the same double-buffered LDS images as conv_patch_kernel one step ahead of
// their use.  Tiling is FLAT: the N*H*W pixels are one run of NHWC rows (a stride-1 1x1 conv has no halo), a tile is 128 * MT
// consecutive pixels whatever the map's shape (19 x 38 maps pad an 8 x 32 tiling by 2.1x), and MT = 1 halves the tile where even
// that leaves most CUs idle.  Same pieces, same MFMA order per output: results are conv_patch_kernel's bit for bit.
// GATED: a squeeze-excite gate row per staged pixel (the pixel's image = linear pixel / gate_hw).
template <int SPLIT, int TN, int MT, bool GATED>
__global__ __launch_bounds__(512, 2) void conv1x1_deep_kernel(const PatchArgs p) {
  typedef bf16x8 V8;
  typedef bf16x4 V4;
  constexpr int D = 4;
  constexpr int NPIX = 128 * MT;
  constexpr int A_OCT = NPIX * 16, A_PLANE = 2 * A_OCT, A_BYTES = SPLIT * A_PLANE;
  constexpr int U_OCT = 64 * 16, U_PLANE = 2 * U_OCT, U_BYTES = SPLIT * U_PLANE;
  constexpr int B_BYTES = TN * U_BYTES, B_SLOTS = B_BYTES / 16, RB = (B_SLOTS + 511) / 512;
  constexpr int ROUNDS = MT;                    // 128 MT pixels x 4 channel quads / 512 threads
  constexpr int NG = GATED ? ROUNDS : 0;
  constexpr int BN = 64 * TN;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const abase = smem;
  char* const bbase = smem + 2 * A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;      // 4 (rows of 32 MT pixels) x 2 (channel halves)
  const int li = lane & 31, lh = lane >> 5;

  const int nblk = p.tiles_n * p.tiles_y;
  int id = xcd_remap(blockIdx.x, nblk);
  const int tn = id % p.tiles_n;
  const int ty = id / p.tiles_n;
  const long lin0 = (long)ty * NPIX;            // first pixel of the tile

  const int cq = tid & 3;
  const int a_lofs0 = (cq >> 1) * A_OCT + (tid >> 2) * 16 + (cq & 1) * 8;
  const float* a_src[ROUNDS];                   // the round's pixel (channel 0), or nullptr beyond the last pixel
  const float* g_src[NG > 0 ? NG : 1];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const long lin = lin0 + r * 128 + (tid >> 2);
    const bool ok = lin < p.flat_P;
    a_src[r] = ok ? p.in + lin * p.in_cs : nullptr;
    if constexpr (GATED) g_src[r] = p.a_scale + (ok ? lin / p.gate_hw : 0) * p.Cin;
  }

  const size_t nsteps_w = (size_t)p.nchunk;
  const char* wbase = p.wpk + (size_t)tn * TN * nsteps_w * U_BYTES;
  const char* b_src[RB];                        // this thread's 16 bytes of the step-0 weight tile; a step is U_BYTES further
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int slot = r * 512 + tid, sl = slot < B_SLOTS ? slot : 0;
    b_src[r] = wbase + (size_t)(sl / (U_BYTES / 16)) * nsteps_w * U_BYTES + (sl % (U_BYTES / 16)) * 16;
  }

  f32x4 ra[D][ROUNDS], rg[D][NG > 0 ? NG : 1], rb[D][RB];
  // branch-free requests (a predicated load compiles to a branch with a full vmcnt wait behind it): what lies beyond the last
  // pixel / beyond Cin reads the tensor's first quad and is zeroed at the split
  auto request = [&](int slot, int c) __attribute__((always_inline)) {
    const int ch = c * PT_CK + cq * 4;
    const bool chok = ch < p.Cin;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      ra[slot][r] = *reinterpret_cast<const f32x4*>(a_src[r] && chok ? a_src[r] + ch : p.in);
#pragma unroll
    for (int r = 0; r < NG; ++r) rg[slot][r] = *reinterpret_cast<const f32x4*>(g_src[r] + (chok ? ch : 0));
#pragma unroll
    for (int r = 0; r < RB; ++r) rb[slot][r] = *reinterpret_cast<const f32x4*>(b_src[r] + (size_t)c * U_BYTES);
  };
  auto land = [&](int slot, int c, int buf) __attribute__((always_inline)) {
    const bool chok = c * PT_CK + cq * 4 < p.Cin;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      f32x4 v = ra[slot][r];
      if constexpr (GATED) v *= rg[slot][r];
      if (!(a_src[r] && chok)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      char* dst = abase + buf * A_BYTES + a_lofs0 + r * (128 * 16);
      f32x4 rem = v;
#pragma unroll
      for (int pl = 0; pl < SPLIT; ++pl) {
        const V4 piece = __builtin_convertvector(rem, V4);
        *reinterpret_cast<V4*>(dst + pl * A_PLANE) = piece;
        if (pl + 1 < SPLIT) rem -= __builtin_convertvector(piece, f32x4);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
      if (r * 512 + tid < B_SLOTS) *reinterpret_cast<f32x4*>(bbase + buf * B_BYTES + (r * 512 + tid) * 16) = rb[slot][r];
  };

  f32x16 acc[MT][TN];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int last = p.nchunk - 1;
#pragma unroll
  for (int j = 0; j < D; ++j) request(j, j < last ? j : last);
  land(0, 0, 0);
  __syncthreads();

  for (int c0 = 0; c0 < p.nchunk; c0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int c = c0 + j;
      if (c >= p.nchunk) break;
      request(j, c + D < last ? c + D : last);        // slot j held chunk c, landed one step ago
      const char* A = abase + (j & 1) * A_BYTES;
      const char* B = bbase + (j & 1) * B_BYTES;
      V8 af[MT][SPLIT], bfr[TN][SPLIT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int idx = (wm * MT + mt) * PT_TW + li;
#pragma unroll
        for (int pl = 0; pl < SPLIT; ++pl)
          af[mt][pl] = *reinterpret_cast<const V8*>(A + pl * A_PLANE + lh * A_OCT + idx * 16);
      }
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const int n = (wn * TN + nt) * 32 + li;
#pragma unroll
        for (int pl = 0; pl < SPLIT; ++pl)
          bfr[nt][pl] = *reinterpret_cast<const V8*>(B + (n >> 6) * U_BYTES + pl * U_PLANE + lh * U_OCT + (n & 63) * 16);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = split_mfma<SPLIT, V8>(bfr[nt], af[mt], acc[mt][nt]);
      if (c + 1 < p.nchunk) land((j + 1) % D, c + 1, (j + 1) & 1);
      __syncthreads();
    }
  }
  // (flat geometry set by the launcher: one image of width 32, row = linear pixel / 32)
  patch_epilogue<TN, false, false, MT>(acc, p, 0, (int)(lin0 / PT_TW), 0, tn * BN, wm, wn, li, lh, 1.f, reinterpret_cast<float*>(smem));
}

// ---------------------------------------------------------------------------------------------
// 3x3 variant, second generation: one kernel ROW (3 taps) per barrier interval.
//   * the weight tiles of the 3 taps arrive by LDS-DMA (global_load_lds_dwordx4: the packed weight image
//     in HBM is byte-identical to the LDS image, 1 KiB per wave-instruction, no VGPR round trip, no
//     ds_write), double-buffered;
//   * the A halo patch is single-buffered: chunk c+1 is prefetched into 12 VGPRs during the three row
//     steps of chunk c and written (with the bf16 split) between two barriers at the chunk boundary;
//   => 4 barriers per 16-channel chunk instead of 9, 36 (bf16x3) MFMAs per wave per interval, 70 KB of
//      LDS so two 8-wave workgroups share a CU and cover each other's barriers.
// (A loader variant that formed cat([skip, bilinear_up(x)]) in place was built in round 1, measured slower than the
// separate upsample_concat pass in every mode -- the 4-tap gather lands on the loader's critical path -- and removed.)
// (the f16x3 128-channel tile takes the 256-register budget too: at 128 registers it spilled 52 bytes per lane -- same
// box, batch-16 step 42.80 -> 42.62 ms)
template <int SPLIT, int TN, bool F16>
__global__ __launch_bounds__(512, (SPLIT * TN >= 6 || (F16 && TN == 2)) ? 2 : PATCH_WAVES_PER_SIMD) void conv_patch3_kernel(const PatchArgs p) {
  typedef typename Piece<F16>::V8 V8;
  typedef typename Piece<F16>::V4 V4;
  constexpr int K = 3;
  constexpr int PH = PT_TH + K - 1, PW = PT_TW + K - 1, NPIX = PH * PW;
  constexpr int NPIXP = (NPIX + 15) / 16 * 16;
  constexpr int BN = 64 * TN;
  constexpr int A_OCT = NPIXP * 16, A_PLANE = 2 * A_OCT, A_BYTES = SPLIT * A_PLANE;
  constexpr int U_OCT = 64 * 16, U_PLANE = 2 * U_OCT, U_BYTES = SPLIT * U_PLANE;   // 64-channel weight units
  constexpr int UROW_BYTES = 3 * U_BYTES, UROW_INSTR = UROW_BYTES / 1024;          // one unit's kernel row
  constexpr int B_BYTES = TN * U_BYTES;
  constexpr int ROW_BYTES = 3 * B_BYTES;               // weight tiles of one kernel row: [unit][kx][plane][oct][64]
  constexpr int ROW_INSTR = ROW_BYTES / 1024;          // 1 KiB LDS-DMA pieces
  constexpr int ROUNDS = 3;                            // 340 px * 4 float4 over 512 threads
  static_assert(NPIX * 4 <= ROUNDS * 512, "A patch does not fit the staging rounds");
  static_assert(ROW_BYTES % 1024 == 0, "weight row must be a whole number of 1 KiB pieces");

  // 256-channel f16x3 tiles run one workgroup per CU anyway (118 KB): a SECOND halo-patch buffer (+22 KB) lets every
  // wave convert and store its prefetched patch rows right after the row step that loaded them -- the conversion VALU
  // work overlaps the other wave's MFMAs and the chunk-boundary barrier disappears (3 barriers per chunk, not 4)
  constexpr bool DBA = F16 && TN == 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const abuf0 = smem;
  char* const bbase = smem + (DBA ? 2 : 1) * A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int nblk = p.tiles_n * p.tiles_x * p.tiles_y * p.N;
  int id = xcd_remap(blockIdx.x, nblk);
  const int tn = id % p.tiles_n; id /= p.tiles_n;
  const int tx = id % p.tiles_x; id /= p.tiles_x;
  const int ty = id % p.tiles_y;
  const int img = id / p.tiles_y;
  const int oy0 = ty * PT_TH, ox0 = tx * PT_TW;

  float a_mul = 1.f, o_mul = 1.f;
  if (F16) a_mul = f16_operand_scale(*p.a_amax, &o_mul);

  const int cq = tid & 3;
  const int a_lofs0 = (cq >> 1) * A_OCT + (tid >> 2) * 16 + (cq & 1) * 8;
  int a_yx[ROUNDS];        // (iy << 16) | ix of the round's patch pixel, -1 if outside the image / unused
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int pix = r * 128 + (tid >> 2);
    const int py = pix / PW, px = pix % PW;
    const int iy = oy0 + py - p.pad_t, ix = ox0 + px - p.pad_l;
    const bool ok = pix < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    a_yx[r] = ok ? (iy << 16) | ix : -1;
  }
  // BRANCH-FREE prefetch (see conv_patch_kernel) -- scale, gate and zeroing happen in store_a
  auto load_a = [&](int r, int c) __attribute__((always_inline)) -> f32x4 {
    const int ch = c * PT_CK + cq * 4;
    const bool ok = a_yx[r] >= 0 && ch < p.Cin;
    const int iy = a_yx[r] >> 16, ix = a_yx[r] & 0xffff;
    return *reinterpret_cast<const f32x4*>(ok ? p.in + ((size_t)(img * p.H + iy) * p.W + ix) * p.in_cs + ch : p.in);
  };
  auto load_gate = [&](int c) __attribute__((always_inline)) -> f32x4 {
    const int ch = c * PT_CK + cq * 4;
    return *reinterpret_cast<const f32x4*>(p.a_scale + (size_t)img * p.Cin + (ch < p.Cin ? ch : 0));
  };
  f32x4 gate = {1.f, 1.f, 1.f, 1.f};
  auto store_a = [&](int r, int c, f32x4 v) __attribute__((always_inline)) {
    if (r * 128 + (tid >> 2) >= NPIX) return;
    char* dst = abuf0 + (DBA ? (c & 1) * A_BYTES : 0) + a_lofs0 + r * (128 * 16);
    const bool ok = a_yx[r] >= 0 && c * PT_CK + cq * 4 < p.Cin;
    if (p.a_scale) v *= gate;
    if (F16) v *= a_mul;
    if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 rem = v;
#pragma unroll
    for (int pl = 0; pl < SPLIT; ++pl) {           // hi, then the bf16 of what is left, ...
      const V4 piece = __builtin_convertvector(rem, V4);
      *reinterpret_cast<V4*>(dst + pl * A_PLANE) = piece;
      if (pl + 1 < SPLIT) rem -= __builtin_convertvector(piece, f32x4);
    }
  };
  // weight rows of this channel tile: per unit [chunk][row(ky)] -> UROW_BYTES each, contiguous
  const size_t nrows_w = (size_t)p.nchunk * 3;
  const char* wrow0 = p.wpk + (size_t)tn * TN * nrows_w * UROW_BYTES;
  auto dma_row = [&](int g) __attribute__((always_inline)) {
    char* dst = bbase + (g & 1) * ROW_BYTES;
#pragma unroll
    for (int j = 0; j < (ROW_INSTR + 7) / 8; ++j) {
      const int i = wave + 8 * j;
      if (i < ROW_INSTR) {
        const int u = i / UROW_INSTR, r = i % UROW_INSTR;
        const char* src = wrow0 + ((size_t)u * nrows_w + g) * UROW_BYTES + r * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  dma_row(0);
  if (p.a_scale) gate = load_gate(0);
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) store_a(r, 0, load_a(r, 0));
  __syncthreads();

  const int nrows = p.nchunk * 3;
  int g = 0;
  for (int c = 0; c < p.nchunk; ++c) {
    const bool more_a = c + 1 < p.nchunk;
    const char* abuf = abuf0 + (DBA ? (c & 1) * A_BYTES : 0);
    f32x4 ra[ROUNDS];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky, ++g) {
      if (g + 1 < nrows) dma_row(g + 1);
      if (more_a) ra[ky] = load_a(ky, c + 1);
      if (ky == 0 && more_a && p.a_scale) gate = load_gate(c + 1);
      const char* Brow = bbase + (g & 1) * ROW_BYTES;
      __builtin_amdgcn_s_setprio(1);            // the matrix phase outranks the other wave's staging work (+0.7 %)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const char* B = Brow + kx * U_BYTES;
        V8 af[2][SPLIT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int idx = (wm * 2 + mt + ky) * PW + kx + li;
#pragma unroll
          for (int pl = 0; pl < SPLIT; ++pl)
            af[mt][pl] = *reinterpret_cast<const V8*>(abuf + pl * A_PLANE + lh * A_OCT + idx * 16);
        }
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          V8 bfr[SPLIT];
          const int n = (wn * TN + nt) * 32 + li;
#pragma unroll
          for (int pl = 0; pl < SPLIT; ++pl)
            bfr[pl] = *reinterpret_cast<const V8*>(B + (n >> 6) * UROW_BYTES + pl * U_PLANE + lh * U_OCT + (n & 63) * 16);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = split_mfma<SPLIT, V8>(bfr, af[mt], acc[mt][nt]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if (DBA && more_a) store_a(ky, c + 1, ra[ky]);       // into the other patch buffer
      __syncthreads();     // row g consumed by every wave; DMA of row g+1 landed (vmcnt drained)
    }
    if (!DBA && more_a) {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) store_a(r, c + 1, ra[r]);
      __syncthreads();
    }
  }

  if constexpr (F16) {
    const bool vec_ok = (p.Cout & 3) == 0 && (p.out_cs & 3) == 0 && (p.out_co & 3) == 0 && (!p.res || (p.res_cs & 3) == 0);
    if (vec_ok) {
      patch_epilogue_lds<TN>(acc, p, img, oy0, ox0, tn * BN, wm, wn, lane, o_mul, reinterpret_cast<float*>(smem));
      return;
    }
  }
  patch_epilogue<TN, F16>(acc, p, img, oy0, ox0, tn * BN, wm, wn, li, lh, o_mul, reinterpret_cast<float*>(smem));
}

template <int SPLIT, int TN, bool F16>
static int launch_patch3(const PatchArgs& a, hipStream_t s) {
  constexpr int NPIXP = ((PT_TH + 2) * (PT_TW + 2) + 15) / 16 * 16;
  constexpr int smem = ((F16 && TN == 4) ? 2 : 1) * SPLIT * 2 * NPIXP * 16 + 2 * 3 * (SPLIT * 2 * 64 * TN * 16);
  static std::atomic<uint64_t> attr_devs{0};
  if (smem > 64 * 1024) CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(conv_patch3_kernel<SPLIT, TN, F16>), smem, attr_devs));
  const int nblk = a.tiles_n * a.tiles_x * a.tiles_y * a.N;
  conv_patch3_kernel<SPLIT, TN, F16><<<nblk, 512, smem, s>>>(a);
  CRESTE_CHECK_LAUNCH("conv_patch3");
  return CRESTE_OK;
}

// ---------------------------------------------------------------------------------------------
// Stride-2 convs (K = 1, 3, 7: the BEV stem 7x7/2, the ResNet 3x3/2 and 1x1/2 downsamples) and stride-1 5x5 / 7x7
// convs (reward network, the input gradient of the 7x7/2 stem) in the f16x3 operand mode -- the row-at-a-time
// structure of conv_patch3_kernel for an 8 x 32 OUTPUT tile (described for stride 2; stride 1 needs no parity split):
//   * the halo patch covers (16 + K - 2) x (64 + K - 2) INPUT pixels of one 16-channel chunk (single-buffered, as
//     fp16 hi + lo); even and odd input columns sit in separate halves of a patch row, so the 32 lanes of an MFMA
//     fragment (32 consecutive output columns = every other input column) read consecutive 16-byte slots;
//   * the K weight tiles of one kernel row arrive by LDS-DMA, double-buffered; K * 6 * TN MFMAs per wave and barrier
//     interval; the next chunk's patch is prefetched branch-free during the K row steps and converted once.
// SPLIT / F16: f16x3 (two fp16 pieces, operands rescaled from their |max| bounds) or bf16x6 (three bf16 pieces, no scaling)
// RP = 2 (stride 2 only; the bf16x6 form of the 7x7/2 BEV stem, whose 21 x 69-pixel patch in three pieces + double-buffered
// weight rows would need 227 KiB): a 16-channel chunk is done in TWO passes by input-row parity -- kernel rows 0, 2, 4, 6
// only touch the patch's even rows, rows 1, 3, 5 the odd ones -- so the LDS holds 11 of the 21 patch rows at a time
// (73.5 + 84 KiB).  Same products, same fp32 accumulation per output element up to the order of the kernel rows.
template <int K, int S, int TN, int SPLIT = 2, bool F16 = true, int RP = 1>
__global__ __launch_bounds__(512, 1) void conv_patch_row_kernel(const PatchArgs p) {
  typedef typename Piece<F16>::V8 h8;
  typedef typename Piece<F16>::V4 h4;
  static_assert(S == 1 || S == 2, "stride 1 or 2");
  static_assert(RP == 1 || (RP == 2 && S == 2), "row-parity passes are for stride 2");
  constexpr int PH = S * PT_TH + K - S, PW = S * PT_TW + K - S;    // input patch extent
  constexpr int PHL = RP == 2 ? (PH + 1) / 2 : PH;                 // patch rows resident in LDS
  constexpr int PWH = (PW + 1) / 2, PROW = S == 2 ? 2 * PWH : PW;  // slots per column parity / per patch row
  constexpr int NSLOT = PHL * PROW, NSLOTP = (NSLOT + 15) / 16 * 16;
  constexpr int NPIX = PHL * PW;
  constexpr int BN = 64 * TN;
  constexpr int A_OCT = NSLOTP * 16, A_PLANE = 2 * A_OCT, A_BYTES = SPLIT * A_PLANE;
  constexpr int U_OCT = 64 * 16, U_PLANE = 2 * U_OCT, U_BYTES = SPLIT * U_PLANE;
  constexpr int UROW_BYTES = K * U_BYTES, UROW_INSTR = UROW_BYTES / 1024;
  constexpr int B_BYTES = TN * U_BYTES;
  constexpr int ROW_BYTES = K * B_BYTES, ROW_INSTR = ROW_BYTES / 1024;
  constexpr int ROUNDS = (NPIX * 4 + 511) / 512;
  constexpr int RPS = (ROUNDS + K - 1) / K;                        // prefetch rounds per row step

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const abuf = smem;
  char* const bbase = smem + A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int nblk = p.tiles_n * p.tiles_x * p.tiles_y * p.N;
  int id = xcd_remap(blockIdx.x, nblk);
  const int tn = id % p.tiles_n; id /= p.tiles_n;
  const int tx = id % p.tiles_x; id /= p.tiles_x;
  const int ty = id % p.tiles_y;
  const int img = id / p.tiles_y;
  const int oy0 = ty * PT_TH, ox0 = tx * PT_TW;

  float o_mul = 1.f, a_mul = 1.f;
  if constexpr (F16) a_mul = f16_operand_scale(*p.a_amax, &o_mul);

  // staging slots: thread = (patch pixel, channel quad); rounds of 128 pixels
  const int cq = tid & 3;
  int a_yx[ROUNDS];          // (iy << 16) | ix, or -1;  RP == 2: ((row of the EVEN pass + 8) << 16) | ix, -1 = column outside
  int a_lofs[ROUNDS];        // LDS byte offset of the pixel's slot (+ octet / half-octet of the quad)
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int pix = r * 128 + (tid >> 2);
    const int py = pix / PW, px = pix - py * PW;
    const int iy = oy0 * S + (RP == 2 ? 2 * py : py) - p.pad_t, ix = ox0 * S + px - p.pad_l;
    if constexpr (RP == 2) {
      a_yx[r] = (pix < NPIX && (unsigned)ix < (unsigned)p.W) ? ((iy + 8) << 16) | ix : -1;    // pad_t <= 8: iy + 8 >= 0
    } else {
      const bool ok = pix < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      a_yx[r] = ok ? (iy << 16) | ix : -1;
    }
    const int slot = S == 2 ? py * PROW + (px & 1) * PWH + (px >> 1) : py * PROW + px;
    a_lofs[r] = pix < NPIX ? (cq >> 1) * A_OCT + slot * 16 + (cq & 1) * 8 : -1;
  }
  // c: chunk (RP == 1) or 2 * chunk + row parity (RP == 2)
  auto load_a = [&](int r, int c) __attribute__((always_inline)) -> f32x4 {
    if constexpr (RP == 2) {
      const int ch = (c >> 1) * PT_CK + cq * 4;
      const int iy = (a_yx[r] >> 16) - 8 + (c & 1), ix = a_yx[r] & 0xffff;
      const bool ok = a_yx[r] >= 0 && (unsigned)iy < (unsigned)p.H && ch < p.Cin;
      return *reinterpret_cast<const f32x4*>(ok ? p.in + ((size_t)(img * p.H + iy) * p.W + ix) * p.in_cs + ch : p.in);
    } else {
      const int ch = c * PT_CK + cq * 4;
      const bool ok = a_yx[r] >= 0 && ch < p.Cin;
      const int iy = a_yx[r] >> 16, ix = a_yx[r] & 0xffff;
      return *reinterpret_cast<const f32x4*>(ok ? p.in + ((size_t)(img * p.H + iy) * p.W + ix) * p.in_cs + ch : p.in);
    }
  };
  auto store_a = [&](int r, int c, f32x4 v) __attribute__((always_inline)) {
    if (a_lofs[r] < 0) return;
    bool ok;
    if constexpr (RP == 2) ok = a_yx[r] >= 0 && (unsigned)((a_yx[r] >> 16) - 8 + (c & 1)) < (unsigned)p.H && (c >> 1) * PT_CK + cq * 4 < p.Cin;
    else ok = a_yx[r] >= 0 && c * PT_CK + cq * 4 < p.Cin;
    if constexpr (F16) v *= a_mul;
    if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pl = 0; pl < SPLIT; ++pl) {          // hi, then the rounding of what is left, ...
      const h4 piece = __builtin_convertvector(v, h4);
      *reinterpret_cast<h4*>(abuf + pl * A_PLANE + a_lofs[r]) = piece;
      if (pl + 1 < SPLIT) v -= __builtin_convertvector(piece, f32x4);
    }
  };
  const size_t nrows_w = (size_t)p.nchunk * K;
  const char* wrow0 = p.wpk + (size_t)tn * TN * nrows_w * UROW_BYTES;
  auto dma_row = [&](int g, int buf) __attribute__((always_inline)) {       // weight row g = chunk * K + ky -> buffer `buf`
    char* dst = bbase + buf * ROW_BYTES;
#pragma unroll
    for (int j = 0; j < (ROW_INSTR + 7) / 8; ++j) {
      const int i = wave + 8 * j;
      if (i < ROW_INSTR) {
        const int u = i / UROW_INSTR, r = i % UROW_INSTR;
        const char* src = wrow0 + ((size_t)u * nrows_w + g) * UROW_BYTES + r * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  dma_row(0, 0);
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) store_a(r, 0, load_a(r, 0));
  __syncthreads();

  const int nrows = p.nchunk * K;
  // RP == 2: one kernel row; `srow` = the resident patch row that output row 0 of the tile reads for it
  auto row_step = [&](const char* Brow, int srow) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      h8 af[2][SPLIT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int slot = ((wm * 2 + mt) + srow) * PROW + (kx & 1) * PWH + li + (kx >> 1);
#pragma unroll
        for (int pl = 0; pl < SPLIT; ++pl)
          af[mt][pl] = *reinterpret_cast<const h8*>(abuf + pl * A_PLANE + lh * A_OCT + slot * 16);
      }
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        h8 bfr[SPLIT];
        const int n = (wn * TN + nt) * 32 + li;
#pragma unroll
        for (int pl = 0; pl < SPLIT; ++pl)
          bfr[pl] = *reinterpret_cast<const h8*>(Brow + (n >> 6) * UROW_BYTES + kx * U_BYTES + pl * U_PLANE + lh * U_OCT +
                                                 (n & 63) * 16);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = split_mfma<SPLIT, h8>(bfr, af[mt], acc[mt][nt]);
      }
    }
  };
  if constexpr (RP == 2) {
    // weight rows are consumed in the order ky = 0, 2, 4, 6, 1, 3, 5 of every chunk; buffer = step parity
    constexpr int KE = (K + 1) / 2, KO = K / 2, RPS2 = (ROUNDS + KO - 1) / KO;
    int s = 0;
    for (int c = 0; c < p.nchunk; ++c) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const bool more_a = par == 0 || c + 1 < p.nchunk;
        const int cn = 2 * c + par + 1;                      // the next (chunk, parity) pass
        f32x4 ra[ROUNDS];
#pragma unroll
        for (int i = 0; i < (par ? KO : KE); ++i, ++s) {
          const int ky = par + 2 * i;
          // the next step of the sequence
          const bool last_of_pass = i + 1 == (par ? KO : KE);
          const int nky = last_of_pass ? (par ? 0 : 1) : ky + 2;
          const int nc = (last_of_pass && par) ? c + 1 : c;
          if (nc < p.nchunk) dma_row(nc * K + nky, (s + 1) & 1);
          if (more_a) {
#pragma unroll
            for (int q = 0; q < RPS2; ++q)
              if (i * RPS2 + q < ROUNDS) ra[i * RPS2 + q] = load_a(i * RPS2 + q, cn);
          }
          row_step(bbase + (s & 1) * ROW_BYTES, ky >> 1);
          __syncthreads();
        }
        if (more_a) {
#pragma unroll
          for (int r = 0; r < ROUNDS; ++r) store_a(r, cn, ra[r]);
          __syncthreads();
        }
      }
    }
  } else {
  int g = 0;
  for (int c = 0; c < p.nchunk; ++c) {
    const bool more_a = c + 1 < p.nchunk;
    f32x4 ra[ROUNDS];
#pragma unroll
    for (int ky = 0; ky < K; ++ky, ++g) {
      if (g + 1 < nrows) dma_row(g + 1, (g + 1) & 1);
      if (more_a) {
#pragma unroll
        for (int q = 0; q < RPS; ++q)
          if (ky * RPS + q < ROUNDS) ra[ky * RPS + q] = load_a(ky * RPS + q, c + 1);
      }
      const char* Brow = bbase + (g & 1) * ROW_BYTES;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        h8 af[2][SPLIT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int slot = S == 2 ? (S * (wm * 2 + mt) + ky) * PROW + (kx & 1) * PWH + li + (kx >> 1)
                                  : (wm * 2 + mt + ky) * PROW + li + kx;
#pragma unroll
          for (int pl = 0; pl < SPLIT; ++pl)
            af[mt][pl] = *reinterpret_cast<const h8*>(abuf + pl * A_PLANE + lh * A_OCT + slot * 16);
        }
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          h8 bfr[SPLIT];
          const int n = (wn * TN + nt) * 32 + li;
#pragma unroll
          for (int pl = 0; pl < SPLIT; ++pl)
            bfr[pl] = *reinterpret_cast<const h8*>(Brow + (n >> 6) * UROW_BYTES + kx * U_BYTES + pl * U_PLANE + lh * U_OCT +
                                                   (n & 63) * 16);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = split_mfma<SPLIT, h8>(bfr, af[mt], acc[mt][nt]);
        }
      }
      __syncthreads();
    }
    if (more_a) {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) store_a(r, c + 1, ra[r]);
      __syncthreads();
    }
  }
  }
  const bool vec_ok = (p.Cout & 3) == 0 && (p.out_cs & 3) == 0 && (p.out_co & 3) == 0 && (!p.res || (p.res_cs & 3) == 0);
  if (vec_ok) patch_epilogue_lds<TN>(acc, p, img, oy0, ox0, tn * BN, wm, wn, lane, o_mul, reinterpret_cast<float*>(smem));
  else patch_epilogue<TN, F16>(acc, p, img, oy0, ox0, tn * BN, wm, wn, li, lh, o_mul, reinterpret_cast<float*>(smem));
}

template <int K, int S, int TN, int SPLIT = 2, bool F16 = true, int RP = 1>
static int launch_patch_row(const PatchArgs& a, hipStream_t s) {
  constexpr int PH = S * PT_TH + K - S, PW = S * PT_TW + K - S, PROW = S == 2 ? 2 * ((PW + 1) / 2) : PW;
  constexpr int NSLOTP = ((RP == 2 ? (PH + 1) / 2 : PH) * PROW + 15) / 16 * 16;
  constexpr int smem = SPLIT * 2 * NSLOTP * 16 + 2 * K * (SPLIT * 2 * 64 * TN * 16);
  static_assert(smem <= 160 * 1024, "stride-2 patch does not fit the LDS");
  static std::atomic<uint64_t> attr_devs{0};
  CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(conv_patch_row_kernel<K, S, TN, SPLIT, F16, RP>), smem, attr_devs));
  const int nblk = a.tiles_n * a.tiles_x * a.tiles_y * a.N;
  conv_patch_row_kernel<K, S, TN, SPLIT, F16, RP><<<nblk, 512, smem, s>>>(a);
  CRESTE_CHECK_LAUNCH("conv_patch_row");
  return CRESTE_OK;
}

// F16X3: per output channel, the inverse of the power of two that brings max|w*scale| into [2^7, 2^8)
__global__ void weight_unscale_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                      float* __restrict__ unscale, int per_co) {
  __shared__ float scratch[4];
  const int co = blockIdx.x;
  const float sc = scale ? scale[co] : 1.f;
  float m = 0.f;
  for (int i = threadIdx.x; i < per_co; i += blockDim.x) m = fmaxf(m, fabsf(w[(long)co * per_co + i] * sc));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, scratch[i]);
    const int e = (int)((__float_as_uint(m) >> 23) & 0xff);
    int ue = e - 7;                                     // biased exponent of 2^((e-127)-7)
    ue = ue < 1 ? 1 : (ue > 253 ? 253 : ue);
    unscale[co] = e == 0 ? 1.f : __uint_as_float((unsigned)ue << 23);
  }
}

// weight packing: OIHW fp32 (x scale[co]) -> [unit][chunk][tap][plane][k-octet][BN = 64][8] bf16 / fp16
template <typename ET>
__global__ void pack_weight_patch_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                         const float* __restrict__ unscale, ET* __restrict__ out, int Cout,
                                         int Cin, int K, int BN, int nchunk, int split, long total) {
  const int T = K * K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long t = i;
    const int e = t % 8; t /= 8;
    const int n = t % BN; t /= BN;
    const int oct = t % 2; t /= 2;
    const int pl = t % split; t /= split;
    const int tap = t % T; t /= T;
    const int c = t % nchunk;
    const int tile = (int)(t / nchunk);
    const int co = tile * BN + n, ci = c * PT_CK + oct * 8 + e;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
      v = w[(((long)co * Cin + ci) * K + tap / K) * K + tap % K];
      if (scale) v *= scale[co];
      if (unscale) v *= 1.f / unscale[co];              // exact: power of two
    }
    ET piece = (ET)v;
    for (int q = 0; q < pl; ++q) { v -= (float)piece; piece = (ET)v; }
    out[i] = piece;
  }
}

// 64-channel weight units per workgroup (tile width = 64 * TN), chosen per LAUNCH on one packed weight image.
// f16x3 takes 256-wide tiles where the layer has them AND enough pixel tiles to fill the chip: 72 MFMAs per wave
// and barrier interval, and the halo patch is staged, split and written to LDS once per 256 channels.  Layers with
// few pixel tiles (the deep 19x38 / 38x76 maps) or few input channels (MBConv expand convs: one or two chunks per
// tile, bound by the output write) take narrower tiles: more, lighter workgroups, two to a CU.
static inline int patch_tn(int cout, int prec, int cin, int K, long px_tiles) {
  // 256-wide tiles: f16x3 everywhere; the bf16 split modes only for 1x1 convs (48 MFMAs per wave and barrier instead of 24:
  // 496->256 @152x304x16 in bf16x6 1.47 -> see DESIGN) -- their 3x3 kernel's weight rows would not fit the LDS at 256
  const int max_tn = cout > 128 ? ((prec == CRESTE_PREC_F16X3 || (K == 1 && prec == CRESTE_PREC_BF16X6)) ? 4 : 2) : (cout > 64 ? 2 : 1);
  int tn = max_tn;
  if (K == 1 && cin < 256 && tn == 4) tn = 2;            // write-bound expand convs: 256-wide tiles only add latency
  // 1x1 convs are latency chains of a few chunk steps per workgroup: they want MORE, lighter workgroups than the 3x3
  // kernels (same box, A/B in isolation: 192->1152 @19x38 x16 88 -> 55 us, 112->672 @38x76 140 -> 83; in the network the
  // two changes together are worth ~0.1 ms of the 43.5 ms step)
  const long thr = K == 1 && cin < 256 ? 1600 : (K == 1 ? 800 : 400);
  while (tn > 1 && px_tiles * ((cout + 64 * tn - 1) / (64 * tn)) < thr) tn >>= 1;   // < ~1.5 workgroups per CU
  return tn;
}
constexpr int PATCH_UNIT_PAD = 4;     // packed units are padded to a multiple of the widest tile
static inline int patch_split(int prec) {
  return prec == CRESTE_PREC_BF16X6 ? 3 : (prec == CRESTE_PREC_BF16X3 || prec == CRESTE_PREC_F16X3 ? 2 : 1);
}

template <int K, int SPLIT, int TN, bool F16>
static int launch_patch(const PatchArgs& a, hipStream_t s) {
  constexpr int PH = PT_TH + K - 1, PW = PT_TW + K - 1, NPIXP = (PH * PW + 15) / 16 * 16;
  constexpr int smem = 2 * (SPLIT * 2 * NPIXP * 16) + 2 * (SPLIT * 2 * 64 * TN * 16);
  static std::atomic<uint64_t> attr_devs{0}, attr_devs_fg{0};
  const int nblk = a.tiles_n * a.tiles_x * a.tiles_y * a.N;
  if constexpr (K == 1 && !F16 && TN <= 2) {
    if (a.stats) {
      CRESTE_REQUIRE(a.gate_hw == 0, "conv2d: out_stats with a per-sample gate on a flat re-tiled 1x1 conv is not built");
      static std::atomic<uint64_t> attr_devs_st{0};
      if (smem > 64 * 1024) CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(conv_patch_kernel<K, SPLIT, TN, F16, false, true>), smem, attr_devs_st));
      conv_patch_kernel<K, SPLIT, TN, F16, false, true><<<nblk, 512, smem, s>>>(a);
      CRESTE_CHECK_LAUNCH("conv_patch (statistics)");
      return CRESTE_OK;
    }
  }
  CRESTE_REQUIRE(!a.stats, "conv2d: out_stats on a kernel that keeps none");
  if (K == 1 && a.gate_hw > 0) {
    if (smem > 64 * 1024) CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(conv_patch_kernel<K, SPLIT, TN, F16, true>), smem, attr_devs_fg));
    conv_patch_kernel<K, SPLIT, TN, F16, true><<<nblk, 512, smem, s>>>(a);
  } else {
    if (smem > 64 * 1024) CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(conv_patch_kernel<K, SPLIT, TN, F16>), smem, attr_devs));
    conv_patch_kernel<K, SPLIT, TN, F16><<<nblk, 512, smem, s>>>(a);
  }
  CRESTE_CHECK_LAUNCH("conv_patch");
  return CRESTE_OK;
}

// conv1x1_deep_kernel: flat tiles of 128 * MT pixels.  Chosen (conv_patch_run) for stride-1 1x1 convs of the bf16 split modes whose
// 256-pixel tiling leaves the chip under one workgroup per CU; MT = 1 where even that fills under half of it.
template <int SPLIT, int TN, int MT>
static int launch_deep(PatchArgs a, const creste_conv_desc* d, hipStream_t s) {
  constexpr int smem = 2 * (SPLIT * 2 * 128 * MT * 16) + 2 * (SPLIT * 2 * 64 * TN * 16);
  const long P = (long)d->N * d->H * d->W;
  a.flat_P = P;
  a.gate_hw = d->H * d->W;
  a.N = 1; a.W = a.Wo = PT_TW; a.H = a.Ho = (int)((P + PT_TW - 1) / PT_TW);
  a.tiles_x = 1; a.tiles_y = (int)((P + 128 * MT - 1) / (128 * MT));
  const int nblk = a.tiles_n * a.tiles_y;
  static std::atomic<uint64_t> attr_devs[2] = {{0}, {0}};
  const void* fn = a.a_scale ? reinterpret_cast<const void*>(conv1x1_deep_kernel<SPLIT, TN, MT, true>)
                             : reinterpret_cast<const void*>(conv1x1_deep_kernel<SPLIT, TN, MT, false>);
  if (smem > 64 * 1024) CRESTE_HIP(ensure_dyn_smem(fn, smem, attr_devs[a.a_scale ? 1 : 0]));
  if (a.a_scale) conv1x1_deep_kernel<SPLIT, TN, MT, true><<<nblk, 512, smem, s>>>(a);
  else conv1x1_deep_kernel<SPLIT, TN, MT, false><<<nblk, 512, smem, s>>>(a);
  CRESTE_CHECK_LAUNCH("conv1x1_deep");
  return CRESTE_OK;
}
template <int SPLIT>
static int launch_deep_tn(const PatchArgs& a, const creste_conv_desc* d, int bn, hipStream_t s) {
  // 128-pixel tiles (MT = 1) were at least as fast as 256-pixel ones on every layer of the encoder (scripts/conv1x1_deep_ab.sh)
  return bn == 128 ? launch_deep<SPLIT, 2, 1>(a, d, s) : launch_deep<SPLIT, 1, 1>(a, d, s);
}

bool conv_patch_supported(int prec, int KH, int KW, int stride) {
  if (prec == CRESTE_PREC_F16X3 && KH == KW && (KH == 1 || KH == 3 || KH == 7) && stride == 2) return true;
  if (prec == CRESTE_PREC_F16X3 && KH == KW && (KH == 5 || KH == 7) && stride == 1) return true;
  // bf16x6 on the row kernel: three pieces of the halo patch fit the LDS (the 7x7/2 stem in two row-parity passes per chunk)
  if (prec == CRESTE_PREC_BF16X6 && KH == KW && (((KH == 1 || KH == 3 || KH == 7) && stride == 2) || ((KH == 5 || KH == 7) && stride == 1))) return true;
  return (prec == CRESTE_PREC_BF16 || prec == CRESTE_PREC_BF16X3 || prec == CRESTE_PREC_BF16X6 ||
          prec == CRESTE_PREC_F16X3) && KH == KW && (KH == 1 || KH == 3) && stride == 1;
}

int64_t conv_patch_weight_bytes(int Cout, int Cin, int K, int prec) {
  const long units = ((Cout + 63) / 64 + PATCH_UNIT_PAD - 1) / PATCH_UNIT_PAD * PATCH_UNIT_PAD;
  const long nchunk = (Cin + PT_CK - 1) / PT_CK;
  return units * nchunk * K * K * patch_split(prec) * 2 * 64 * 16;
}

int conv_patch_pack(const float* w, const float* scale, void* wpk, float* w_unscale, int Cout, int Cin, int K,
                    int prec, hipStream_t s) {
  const int bn = 64, split = patch_split(prec), nchunk = (Cin + PT_CK - 1) / PT_CK;   // 64-channel units
  const long total = conv_patch_weight_bytes(Cout, Cin, K, prec) / 2;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (prec == CRESTE_PREC_F16X3) {
    weight_unscale_kernel<<<Cout, 256, 0, s>>>(w, scale, w_unscale, Cin * K * K);
    CRESTE_CHECK_LAUNCH("weight_unscale");
    pack_weight_patch_kernel<_Float16><<<blocks, 256, 0, s>>>(w, scale, w_unscale, (_Float16*)wpk, Cout, Cin, K, bn,
                                                              nchunk, split, total);
  } else {
    pack_weight_patch_kernel<__bf16><<<blocks, 256, 0, s>>>(w, scale, nullptr, (__bf16*)wpk, Cout, Cin, K, bn, nchunk,
                                                            split, total);
  }
  CRESTE_CHECK_LAUNCH("pack_weight_patch");
  return CRESTE_OK;
}

// geometry of a launch: pixel tiles (with the flat re-tiling of a halo-free 1x1 conv, below)
static void patch_geometry(const creste_conv_desc* d, PatchArgs& a) {
  a.N = d->N; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
  a.tiles_x = (d->Wo + PT_TW - 1) / PT_TW;
  a.tiles_y = (d->Ho + PT_TH - 1) / PT_TH;
  a.gate_hw = 0;
  a.flat_P = 0;
  // A stride-1 1x1 conv has no halo: when the 8 x 32 pixel tiles pad the map by more than 12 % (19 x 38: 24 x 64 =
  // 2.1x the pixels, 38 x 76: 1.33x), its N*H*W pixels are re-tiled as ONE image of width 32 -- every tile is 256
  // consecutive pixels of the NHWC buffer, every linear pixel index (input, output, residual, row mask) is unchanged;
  // only the per-image squeeze-excite gate needs the pixel's image (gate_hw).  Same box, A/B in isolation: 192->1152 @19x38 x16
  // 152 -> 88 us, 1152->192 86 -> 65, 112->672 @38x76 155 -> 142
  if (d->KH == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->Ho == d->H && d->Wo == d->W) {
    const long hw = (long)d->H * d->W, P = hw * d->N;
    const long padded = (long)a.tiles_x * a.tiles_y * PT_TH * PT_TW;
    if (P % PT_TW == 0 && padded * 100 > hw * 112 && P / PT_TW < (1L << 30) && hw < (1L << 30)) {
      a.gate_hw = d->a_scale ? (int)hw : 0;
      a.N = 1; a.W = a.Wo = PT_TW; a.H = a.Ho = (int)(P / PT_TW);
      a.tiles_x = 1; a.tiles_y = (a.Ho + PT_TH - 1) / PT_TH;
    }
  }
}

// rows of creste_conv_desc.out_stats (one per pixel tile), or -1: only the stride-1 1x1 kernels of the bf16 split modes keep them
int conv_patch_stat_rows(const creste_conv_desc* d) {
  const bool bf = d->prec == CRESTE_PREC_BF16 || d->prec == CRESTE_PREC_BF16X3 || d->prec == CRESTE_PREC_BF16X6;
  if (!bf || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->res || d->row_mask || (d->Cout & 3) || (d->out_cs & 3) || (d->out_co & 3))
    return -1;
  PatchArgs a;
  patch_geometry(d, a);
  if (a.gate_hw > 0) return -1;                    // gated AND flat re-tiled: the per-pixel-gate instantiation keeps none
  const long rows = (long)a.tiles_x * a.tiles_y * a.N;
  return rows < (1L << 30) ? (int)rows : -1;
}

int conv_patch_run(const creste_conv_desc* d, hipStream_t s) {
  PatchArgs a;
  a.in = d->in; a.wpk = (const char*)d->wpk; a.bias = d->bias; a.res = d->res; a.a_scale = d->a_scale;
  a.row_mask = d->row_mask; a.out = d->out;
  a.a_amax = d->a_amax; a.out_amax = d->out_amax; a.w_unscale = d->w_unscale;
  a.Cin = d->Cin; a.in_cs = d->in_cs;
  a.Cout = d->Cout; a.out_cs = d->out_cs; a.out_co = d->out_co;
  a.res_cs = d->res_cs; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.act = d->act;
  a.nchunk = (d->Cin + PT_CK - 1) / PT_CK;
  patch_geometry(d, a);
  a.stats = d->out_stats;
  CRESTE_REQUIRE(!d->out_stats || conv_patch_stat_rows(d) > 0, "conv2d: out_stats is kept by the stride-1 1x1 kernels of the bf16 split modes "
                                                               "(Cout and the output slice multiples of 4, no residual / row mask)");
  int bn = 64 * patch_tn(d->Cout, d->prec, d->Cin, d->KH, (long)a.tiles_x * a.tiles_y * a.N);
  if (a.stats && bn > 128) bn = 128;               // the statistics epilogue is built for tiles of <= 128 channels
  a.tiles_n = (d->Cout + bn - 1) / bn;
  const int split = patch_split(d->prec);
  const int K = d->KH;
  if ((d->stride == 2 || K > 3) && d->prec == CRESTE_PREC_BF16X6) {      // three bf16 pieces: 64-cout tiles (the LDS budget)
    a.tiles_n = (d->Cout + 63) / 64;
    if (d->stride == 1) return K == 7 ? launch_patch_row<7, 1, 1, 3, false>(a, s) : launch_patch_row<5, 1, 1, 3, false>(a, s);
    if (K == 3) return launch_patch_row<3, 2, 1, 3, false>(a, s);
    if (K == 7) {
      CRESTE_REQUIRE(d->pad_t >= 0 && d->pad_t <= 8 && d->H < 32000, "conv2d: 7x7/2 bf16x6 row-parity kernel: pad_t %d / H %d out of its packed range", d->pad_t, d->H);
      return launch_patch_row<7, 2, 1, 3, false, 2>(a, s);
    }
    if (d->Cout > 64) { a.tiles_n = (d->Cout + 127) / 128; return launch_patch_row<1, 2, 2, 3, false>(a, s); }
    return launch_patch_row<1, 2, 1, 3, false>(a, s);
  }
  if (d->stride == 2 || K > 3) {        // f16x3 (conv_patch_supported): the row-at-a-time kernels
    const int tn2 = d->Cout > 64 && !(K == 7 && d->stride == 2) ? 2 : 1;      // tiles of 64 or 128 channels
    a.tiles_n = (d->Cout + 64 * tn2 - 1) / (64 * tn2);
    if (d->stride == 1) {
      if (K == 7) return tn2 == 2 ? launch_patch_row<7, 1, 2>(a, s) : launch_patch_row<7, 1, 1>(a, s);
      return tn2 == 2 ? launch_patch_row<5, 1, 2>(a, s) : launch_patch_row<5, 1, 1>(a, s);
    }
    if (K == 7) return launch_patch_row<7, 2, 1>(a, s);
    if (K == 3) return tn2 == 2 ? launch_patch_row<3, 2, 2>(a, s) : launch_patch_row<3, 2, 1>(a, s);
    return tn2 == 2 ? launch_patch_row<1, 2, 2>(a, s) : launch_patch_row<1, 2, 1>(a, s);
  }
  if (d->prec == CRESTE_PREC_F16X3) {
    if (K == 3)
      return bn == 256 ? launch_patch3<2, 4, true>(a, s)
                       : bn == 128 ? launch_patch3<2, 2, true>(a, s) : launch_patch3<2, 1, true>(a, s);
    return bn == 256 ? launch_patch<1, 2, 4, true>(a, s)
                     : bn == 128 ? launch_patch<1, 2, 2, true>(a, s) : launch_patch<1, 2, 1, true>(a, s);
  }
  if (K == 3) {
    if (bn == 128)
      return split == 3 ? launch_patch3<3, 2, false>(a, s)
                        : split == 2 ? launch_patch3<2, 2, false>(a, s) : launch_patch3<1, 2, false>(a, s);
    return split == 3 ? launch_patch3<3, 1, false>(a, s)
                      : split == 2 ? launch_patch3<2, 1, false>(a, s) : launch_patch3<1, 1, false>(a, s);
  }
  if (!a.stats && d->pad_t == 0 && d->pad_l == 0 && d->Ho == d->H && d->Wo == d->W && a.nchunk >= 4) {
    // a K loop worth pipelining: the deep-prefetch form on flat 128-pixel tiles (conv1x1_deep_kernel).  Same box, batch 8, old -> deep:
    // 1152->192 @19x38 76 -> 38 us, 192->1152 48 -> 28, 672->112 @38x76 58 -> 43, 256->128 @152x304 202 -> 186; batch 1: 72 -> 34, 44 -> 21.
    // Tile width: 128 channels where that still leaves >= 4 workgroups per CU, else 64 (more, lighter workgroups).  The write-bound
    // thin projections on large maps (144->24 @152x304: thousands of workgroups, two to a CU) stay on conv_patch_kernel: 69 -> 77.
    // (experiment / test knobs, read per call: CRESTE_CONV1X1_DEEP = 0 never, 2 wherever it is built; CRESTE_CONV1X1_DEEP_BN = 64 / 128)
    const char* e_mode = getenv("CRESTE_CONV1X1_DEEP");
    const char* e_bn = getenv("CRESTE_CONV1X1_DEEP_BN");
    const int deep_mode = e_mode ? atoi(e_mode) : 1, deep_bn = e_bn ? atoi(e_bn) : 0;
    const long P = (long)d->N * d->H * d->W, pt = (P + 255) / 256;
    const long n128 = pt * ((d->Cout + 127) / 128), n64 = pt * ((d->Cout + 63) / 64);
    int dbn = d->Cout > 64 && n128 >= 1024 ? 128 : 64;
    if (deep_bn == 64 || (deep_bn == 128 && d->Cout > 64)) dbn = deep_bn;
    const bool take = d->Cout > 64 ? (bn <= 128 || deep_mode == 2) : n64 <= 1024;
    if (P < (1L << 30) && deep_mode != 0 && (deep_mode == 2 || take)) {
      bn = dbn; a.tiles_n = (d->Cout + bn - 1) / bn;
      return split == 3 ? launch_deep_tn<3>(a, d, bn, s) : split == 2 ? launch_deep_tn<2>(a, d, bn, s) : launch_deep_tn<1>(a, d, bn, s);
    }
  }
  if (bn == 256) return launch_patch<1, 3, 4, false>(a, s);          // bf16x6 only (patch_tn)
  if (bn == 128)
    return split == 3 ? launch_patch<1, 3, 2, false>(a, s)
                      : split == 2 ? launch_patch<1, 2, 2, false>(a, s) : launch_patch<1, 1, 2, false>(a, s);
  return split == 3 ? launch_patch<1, 3, 1, false>(a, s)
                    : split == 2 ? launch_patch<1, 2, 1, false>(a, s) : launch_patch<1, 1, 1, false>(a, s);
}

}  // namespace creste
