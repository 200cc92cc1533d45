// Three 1x1 convolutions (+ folded BatchNorm + ReLU each) as ONE kernel: the distillation head
//   MultiLayerConv(kernels [1,1,1], dims [C, 128, 128, 128])        reference creste/models/distillation.py:179, blocks/conv.py:5-32
// on the split-operand bf16 matrix cores (bf16x6 / bf16x3, the products and their order of csrc/conv_patch.hip).
//
// Why: as three launches of the 1x1 engine every layer reads its input from HBM and writes its output back (256 -> 128 -> 128 -> 128 at
// 152 x 304 x 16: 2.7 GB, 0.96 ms) although a pixel's 128 hidden channels never need to leave the wave that computed them.  The
// engine issues its MFMAs as D = W * X^T: a lane owns one PIXEL (column lane & 31) and its 16 accumulator registers per 32-channel tile
// own CHANNELS (row 8 (r >> 2) + 4 (lane >> 5) + (r & 3)).  The B operand of the next layer's MFMA wants, per lane, eight k-values of
// one pixel -- and registers 8j .. 8j+7 of tile t ARE eight channels of the lane's own pixel.  So layer l+1 runs straight out of layer
// l's accumulators (bias + ReLU + the split into bf16 pieces in registers), with the k-order of its WEIGHT image permuted at pack time
// to the order the accumulator layout dictates: MFMA step (t, j), lane half h  <->  channels 32t + 16j + {4h + e, 8 + 4h + e}, e < 4.
// No LDS round trip, no cross-lane traffic.  Only the first layer's input (four 16-byte loads per lane and
// step of 32 channels, four steps ahead) and the last layer's output touch memory; the weight tiles (24 KB per step) stream through two LDS stages, shared
// by the workgroup's eight waves = 256 pixels.
#include "common.h"

namespace creste {

typedef __bf16 chbf16x8 __attribute__((ext_vector_type(8)));
typedef float chf32x16 __attribute__((ext_vector_type(16)));
typedef float chf32x8 __attribute__((ext_vector_type(8)));
typedef float chf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned chu32x4 __attribute__((ext_vector_type(4)));

struct ChainArgs {
  const float* in;          // [P, in_cs] (channel offset applied)
  const char* wimg;         // [nch1 + 16 steps][SPLIT][2 k-octets][128 couts][8] bf16 (ops.pack_conv1x1_chain3)
  const float* bias;        // [3][128] fp32 (conv bias and BatchNorm shift folded)
  float* out;               // [P, out_cs] at channel offset out_co
  long P;
  int in_cs, out_cs, out_co;
  int nch1;                 // 16-channel chunks of the first layer (Cin / 16)
};

template <int SPLIT>
__device__ __forceinline__ void ch_split(const chf32x8& v, chbf16x8 (&pc)[SPLIT]) {
  chf32x8 x = v;
#pragma unroll
  for (int pl = 0; pl < SPLIT; ++pl) {
    pc[pl] = __builtin_convertvector(x, chbf16x8);
    if (pl + 1 < SPLIT) x -= __builtin_convertvector(pc[pl], chf32x8);
  }
}

template <int SPLIT>
__global__ __launch_bounds__(512, 2) void conv1x1_chain3_kernel(const ChainArgs p) {
  // a STEP = 32 input channels = two MFMA k-chunks: one barrier and one 2 x 12 KB weight tile per step, 2 x 24 MFMAs per wave
  constexpr int TILE = SPLIT * 2 * 128 * 16;            // bytes of one 16-channel chunk's weight tile
  constexpr int UNITS = 2 * TILE / 16, UPT = UNITS / 512;
  static_assert(UNITS % 512 == 0, "a step's weight tile is dealt out evenly");
  __shared__ __attribute__((aligned(16))) char wl[2 * 2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const long px = (long)blockIdx.x * 256 + wave * 32 + li;
  const bool pok = px < p.P;
  const float* xrow = p.in + (pok ? px : 0) * p.in_cs + 8 * lh;
  const int ns1 = p.nch1 >> 1;

  chu32x4 wreg[UPT];
  auto load_w = [&](int step) __attribute__((always_inline)) {
    const char* src = p.wimg + (size_t)step * (2 * TILE);
#pragma unroll
    for (int u = 0; u < UPT; ++u) wreg[u] = *reinterpret_cast<const chu32x4*>(src + (size_t)(tid + 512 * u) * 16);
  };
  auto store_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UPT; ++u) *reinterpret_cast<chu32x4*>(wl + buf * (2 * TILE) + (tid + 512 * u) * 16) = wreg[u];
  };
  chf32x16 acc[4];
  auto zero = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  };
  // one chunk's 4 x (SPLIT (SPLIT + 1) / 2) MFMAs: weights (A operand) from stage `buf`, half `sub`; activations `b` from
  // registers; piece products smallest first, as conv_patch.hip's split_mfma
  auto mma = [&](int buf, int sub, const chbf16x8 (&b)[SPLIT]) __attribute__((always_inline)) {
    const char* W = wl + (buf * 2 + sub) * TILE + (lh * 128 + li) * 16;
#pragma unroll
    for (int np = 0; np < 2; ++np) {                    // two cout tiles at a time: consecutive MFMAs go to different accumulators
      chbf16x8 a0[SPLIT], a1[SPLIT];
#pragma unroll
      for (int pl = 0; pl < SPLIT; ++pl) {
        a0[pl] = *reinterpret_cast<const chbf16x8*>(W + (pl * 2 * 128 + (2 * np) * 32) * 16);
        a1[pl] = *reinterpret_cast<const chbf16x8*>(W + (pl * 2 * 128 + (2 * np + 1) * 32) * 16);
      }
#pragma unroll
      for (int order = SPLIT - 1; order >= 0; --order)
#pragma unroll
        for (int pa = order; pa >= 0; --pa) {
          acc[2 * np] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[pa], b[order - pa], acc[2 * np], 0, 0, 0);
          acc[2 * np + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[pa], b[order - pa], acc[2 * np + 1], 0, 0, 0);
        }
    }
  };

  // ---- layer 1: K = 16 nch1 from memory.  The lane's 64 bytes of a step (channels 32 s + 8 h .. + 7 and + 16 ..) are loaded FOUR
  // steps ahead (an HBM round trip is several steps of matrix work; the registers are the ones the hidden layer lives in later)
  zero();
  load_w(0);
  chf32x4 xq[4][4];
  auto load_x = [&](int s, chf32x4 (&q)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      q[k] = *reinterpret_cast<const chf32x4*>(xrow + 32 * s + 16 * (k >> 1) + 4 * (k & 1));
  };
  // (every load below is UNCONDITIONAL -- rows past the last pixel read pixel 0, steps past the last one re-read it: a load behind
  // a branch makes the compiler wait for vmcnt(0) in front of the weight tile's LDS stores, i.e. for the HBM round trip of the
  // input prefetch in every step: 697 -> 4xx us)
#pragma unroll
  for (int u = 0; u < 4; ++u) load_x(u < ns1 ? u : ns1 - 1, xq[u]);
  store_w(0);
  int step = 0;
  for (int base = 0; base < ns1; base += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sidx = base + u;
      if (sidx < ns1) {                                  // (workgroup-uniform)
        __syncthreads();                                 // tile `step` is in stage step & 1; nobody reads the other stage any more
        load_w(step + 1);                                // (layer 2's first tile behind the last step)
        chbf16x8 b0[SPLIT], b1[SPLIT];
        ch_split<SPLIT>(chf32x8{xq[u][0][0], xq[u][0][1], xq[u][0][2], xq[u][0][3], xq[u][1][0], xq[u][1][1], xq[u][1][2], xq[u][1][3]}, b0);
        ch_split<SPLIT>(chf32x8{xq[u][2][0], xq[u][2][1], xq[u][2][2], xq[u][2][3], xq[u][3][0], xq[u][3][1], xq[u][3][2], xq[u][3][3]}, b1);
        load_x(sidx + 4 < ns1 ? sidx + 4 : ns1 - 1, xq[u]);
        mma(step & 1, 0, b0);
        mma(step & 1, 1, b1);
        store_w((step + 1) & 1);
        ++step;
      }
    }
  }
  // ---- layers 2 and 3: K = 128 out of the previous layer's accumulators
  chf32x16 h[4];
#pragma unroll
  for (int layer = 1; layer < 3; ++layer) {
    const float* bs = p.bias + (layer - 1) * 128 + 4 * lh;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const chf32x4 b4 = *reinterpret_cast<const chf32x4*>(bs + 32 * t + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) h[t][4 * g + e] = fmaxf(acc[t][4 * g + e] + b4[e], 0.f);
      }
    zero();
#pragma unroll
    for (int t = 0; t < 4; ++t, ++step) {
      __syncthreads();
      if (layer == 1 || t < 3) load_w(step + 1);
      chbf16x8 b0[SPLIT], b1[SPLIT];
      ch_split<SPLIT>(chf32x8{h[t][0], h[t][1], h[t][2], h[t][3], h[t][4], h[t][5], h[t][6], h[t][7]}, b0);
      ch_split<SPLIT>(chf32x8{h[t][8], h[t][9], h[t][10], h[t][11], h[t][12], h[t][13], h[t][14], h[t][15]}, b1);
      mma(step & 1, 0, b0);
      mma(step & 1, 1, b1);
      if (layer == 1 || t < 3) store_w((step + 1) & 1);
    }
  }
  if (!pok) return;
  const float* bs = p.bias + 2 * 128 + 4 * lh;
  float* dst = p.out + px * p.out_cs + p.out_co + 4 * lh;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const chf32x4 b4 = *reinterpret_cast<const chf32x4*>(bs + 32 * t + 8 * g);
      chf32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[t][4 * g + e] + b4[e], 0.f);
      *reinterpret_cast<chf32x4*>(dst + 32 * t + 8 * g) = v;
    }
}

}  // namespace creste

extern "C" int64_t creste_conv1x1_chain3_weight_bytes(int Cin, int prec) {
  const int split = prec == CRESTE_PREC_BF16X6 ? 3 : (prec == CRESTE_PREC_BF16X3 ? 2 : 0);
  if (split == 0 || Cin <= 0 || Cin % 32) return -1;
  return (int64_t)(Cin / 16 + 16) * split * 2 * 128 * 16;
}

extern "C" int creste_conv1x1_chain3_f32(const float* in, int in_cs, int64_t P, int Cin, const void* wimg, const float* bias, int prec,
                                         float* out, int out_cs, int out_co, void* stream) {
  using namespace creste;
  CRESTE_REQUIRE(in && wimg && bias && out && P > 0, "conv1x1_chain3: null pointer / no pixels");
  CRESTE_REQUIRE(creste_conv1x1_chain3_weight_bytes(Cin, prec) > 0, "conv1x1_chain3: bf16x6 / bf16x3 operands, Cin a multiple of 32");
  CRESTE_REQUIRE(in_cs >= Cin && in_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && out_cs >= out_co + 128 && out_cs % 4 == 0 &&
                     out_co % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(wimg) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(bias) & 15) == 0,
                 "conv1x1_chain3: 16-byte aligned buffers, channel strides / offsets multiples of 4, 128 output channels inside out_cs");
  ChainArgs a;
  a.in = in; a.wimg = (const char*)wimg; a.bias = bias; a.out = out; a.P = P; a.in_cs = in_cs; a.out_cs = out_cs; a.out_co = out_co;
  a.nch1 = Cin / 16;
  const unsigned grid = (unsigned)((P + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (prec == CRESTE_PREC_BF16X6) conv1x1_chain3_kernel<3><<<grid, 512, 0, s>>>(a);
  else conv1x1_chain3_kernel<2><<<grid, 512, 0, s>>>(a);
  CRESTE_CHECK_LAUNCH("conv1x1_chain3");
  return CRESTE_OK;
}
