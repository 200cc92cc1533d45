// Implicit-GEMM convolution on the gfx950 matrix cores (NHWC activations, fused epilogue).
//
//   GEMM view:  M = N*Ho*Wo output pixels, Ncol = Cout, K = KH*KW*Cin (tap-major, channel-minor)
//   A[m][k]  gathered on the fly from the NHWC input (zero outside the image / beyond Cin)
//   B[n][k]  pre-packed weights [cout_pad][KH*KW*cin_pad] (k contiguous, zero padded)
//
// fp32 path (parity mode): v_mfma_f32_32x32x2_f32 -- exact fp32 products and accumulation at the
// fp32 matrix rate (157 TF peak, MI355X_MICROARCH.md).  One MFMA is 64 cycles per SIMD for 4 operand
// bytes per lane, so operand delivery is never the limiter: register-staged global->LDS copies, one
// barrier per 16-deep K step, double-buffered LDS.
//
// Tile: BM = 128 pixels x BN in {128,64,32} channels per 256-thread workgroup (4 waves of 64).
// LDS image of both operands: [row][20 floats] (16 used + 4 pad).  A lane fetches 4 consecutive k of
// its row with one ds_read_b128; row stride 80 B makes the 16-lane read groups hit 16 distinct 16-B
// slots (5 is odd), i.e. conflict-free.  The two half-waves take k = 0..7 and k = 8..15 of the step
// (the hardware k-split of the 32x32x2 MFMA is between lanes 0-31 and 32-63, and it only has to be
// consistent between A and B).
#include "common.h"

namespace creste {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;       // K depth per step
constexpr int LDS_LD = 20;   // floats per LDS row (16 + 4 pad)

struct ConvArgs {
  const float* in;
  const float* wpk;
  const float* bias;
  const float* res;
  const float* a_scale;
  const float* row_mask;
  float* out;
  float* out_amax;
  int N, H, W, Cin, in_cs;
  int Ho, Wo, Cout, out_cs, out_co, res_cs;
  int KH, KW, stride, pad_t, pad_l;
  int act;
  int cin_pad;   // Cin rounded up to BK
  int ktot;      // KH*KW*cin_pad
  int M;         // N*Ho*Wo
  int tiles_m, tiles_n;
};

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs p) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int A_PER_T = BM * 4 / 256;                 // float4 loads of A per thread per step
  constexpr int B_PER_T = (BN * 4 + 255) / 256;         // float4 loads of B per thread per step
  constexpr bool B_PARTIAL = (BN * 4 < 256);

  __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: consecutive logical ids walk the N tiles of one M tile, then the next M
  // tile; each XCD gets a contiguous run so the A rows (and their 3x3 halos) stay in one L2.
  const int logical = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tile_n = logical % p.tiles_n;
  const int tile_m = logical / p.tiles_n;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;

  // ---- per-thread A rows (fixed for the whole K loop)
  const int kq = tid & 3;            // which 4-float quad of the 16-deep step
  const int r_in_pass = tid >> 2;    // 0..63
  int a_n[A_PER_T], a_iy0[A_PER_T], a_ix0[A_PER_T];
  bool a_ok[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int m = bm0 + r_in_pass + 64 * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int ox = mm % p.Wo;
    const int t2 = mm / p.Wo;
    const int oy = t2 % p.Ho;
    a_n[i] = t2 / p.Ho;
    a_iy0[i] = oy * p.stride - p.pad_t;
    a_ix0[i] = ox * p.stride - p.pad_l;
  }
  const float* bptr[B_PER_T];
#pragma unroll
  for (int i = 0; i < B_PER_T; ++i)
    bptr[i] = p.wpk + (size_t)(bn0 + r_in_pass + 64 * i) * p.ktot + kq * 4;

  f32x4 ra[A_PER_T], rb[B_PER_T];

  int ky = 0, kx = 0, c0 = 0;   // position of the NEXT step to be loaded
  auto load_step = [&](int ks) __attribute__((always_inline)) {
    const int c = c0 + kq * 4;
    const bool c_ok = c < p.Cin;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      const bool ok = a_ok[i] && c_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const float* src = p.in + (((size_t)a_n[i] * p.H + iy) * p.W + ix) * p.in_cs + c;
        v = *reinterpret_cast<const f32x4*>(src);
        if (p.a_scale)
          v *= *reinterpret_cast<const f32x4*>(p.a_scale + (size_t)a_n[i] * p.Cin + c);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      if (!B_PARTIAL || tid < BN * 4)
        rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + (size_t)ks * BK);
    }
    c0 += BK;
    if (c0 >= p.cin_pad) { c0 = 0; if (++kx == p.KW) { kx = 0; ++ky; } }
  };
  auto store_step = [&](int buf) __attribute__((always_inline)) {
    float* A = lds[buf];
    float* B = A + BM * LDS_LD;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i)
      *reinterpret_cast<f32x4*>(A + (r_in_pass + 64 * i) * LDS_LD + kq * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i)
      if (!B_PARTIAL || tid < BN * 4)
        *reinterpret_cast<f32x4*>(B + (r_in_pass + 64 * i) * LDS_LD + kq * 4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.ktot / BK;
  const int li = lane & 31, lh = lane >> 5;

  load_step(0);
  store_step(0);
  __syncthreads();

  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) load_step(ks + 1);

    const float* A = lds[cur] + (wm * TM * 32 + li) * LDS_LD + lh * 8;
    const float* B = lds[cur] + BM * LDS_LD + (wn * TN * 32 + li) * LDS_LD + lh * 8;
    f32x4 a4[TM][2], b4[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      a4[i][0] = *reinterpret_cast<const f32x4*>(A + i * 32 * LDS_LD);
      a4[i][1] = *reinterpret_cast<const f32x4*>(A + i * 32 * LDS_LD + 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      b4[j][0] = *reinterpret_cast<const f32x4*>(B + j * 32 * LDS_LD);
      b4[j][1] = *reinterpret_cast<const f32x4*>(B + j * 32 * LDS_LD + 4);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][h][t], b4[j][h][t], acc[i][j], 0, 0, 0);
        }
      }
    }
    if (ks + 1 < nk) store_step(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float vmax = 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = bn0 + wn * TN * 32 + j * 32 + li;
    const bool n_ok = n < p.Cout;
    const float bias = (n_ok && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mbase = bm0 + wm * TM * 32 + i * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (n_ok && m < p.M) {
          float v = acc[i][j][r] + bias;
          if (p.res) v += p.res[(size_t)m * p.res_cs + n];
          v = act_apply(v, p.act);
          if (p.row_mask) v *= p.row_mask[m];
          vmax = fmaxf(vmax, fabsf(v));
          p.out[(size_t)m * p.out_cs + p.out_co + n] = v;
        }
      }
    }
  }
  if (p.out_amax) block_amax_update(vmax, p.out_amax, &lds[0][0]);
}

// ---------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 -> [cout_pad][KH*KW*cin_pad] (k = (ky*KW+kx)*cin_pad + ci), x scale[co]
__global__ void pack_weight_f32_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                       float* __restrict__ out, int Cout, int Cin, int KH, int KW,
                                       int cin_pad, int cout_pad) {
  const long total = (long)cout_pad * KH * KW * cin_pad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ci = i % cin_pad;
    long t = i / cin_pad;
    const int kx = t % KW; t /= KW;
    const int ky = t % KH;
    const int co = t / KH;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
      v = w[(((long)co * Cin + ci) * KH + ky) * KW + kx];
      if (scale) v *= scale[co];
    }
    out[i] = v;
  }
}

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline int pick_bn(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

}  // namespace creste

namespace creste {   // conv_patch.hip
bool conv_patch_supported(int prec, int KH, int KW, int stride);
int64_t conv_patch_weight_bytes(int Cout, int Cin, int K, int prec);
int conv_patch_pack(const float* w, const float* scale, void* wpk, float* w_unscale, int Cout, int Cin, int K,
                    int prec, hipStream_t s);
int conv_patch_run(const creste_conv_desc* d, hipStream_t s);
int conv_patch_stat_rows(const creste_conv_desc* d);
}  // namespace creste
namespace creste {   // conv_wino.hip
bool conv_wino_supported(int prec, int KH, int KW, int stride, int Cin, int Cout);
int64_t conv_wino_weight_bytes(int Cout, int Cin, int prec);
int conv_wino_pack(const float* w, const float* scale, void* wpk, int Cout, int Cin, int prec, hipStream_t s);
int64_t conv_wino_workspace_bytes(int N, int Ho, int Wo, int Cout);
int conv_wino_run(const creste_conv_desc* d, hipStream_t s);
}  // namespace creste
namespace creste {   // conv_wino4.hip
bool conv_wino4_supported(int prec, int KH, int KW, int stride, int Cin, int Cout);
int64_t conv_wino4_weight_bytes(int Cout, int Cin, int prec);
int conv_wino4_pack(const float* w, const float* scale, void* wpk, int Cout, int Cin, int prec, hipStream_t s);
int64_t conv_wino4_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int prec);
int conv_wino4_stat_rows(const creste_conv_desc* d);
int conv_wino4_run(const creste_conv_desc* d, hipStream_t s);
}  // namespace creste

namespace creste {
size_t conv_desc_bytes() { return sizeof(creste_conv_desc); }     // csrc/plan_runtime.cpp checks a plan file against it
}

using namespace creste;

extern "C" int creste_conv_stat_rows(const creste_conv_desc* d) {
  if (!d || d->Cout <= 0 || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0) return -1;
  if (d->algo == CRESTE_ALGO_WINOGRAD4) return conv_wino4_stat_rows(d);
  if (d->algo == CRESTE_ALGO_DIRECT && d->prec != CRESTE_PREC_F32) return conv_patch_stat_rows(d);
  return -1;
}

extern "C" int creste_conv_supported(int prec, int KH, int KW, int stride) {
  if (prec == CRESTE_PREC_F32) return KH > 0 && KW > 0 && stride > 0;
  return conv_patch_supported(prec, KH, KW, stride) ? 1 : 0;
}

extern "C" int creste_conv_wino_supported(int prec, int KH, int KW, int stride, int Cin, int Cout) {
  return conv_wino_supported(prec, KH, KW, stride, Cin, Cout) ? 1 : 0;
}

extern "C" int64_t creste_conv_wino_weight_bytes(int Cout, int Cin, int prec) {
  return conv_wino_supported(prec, 3, 3, 1, Cin, Cout) ? conv_wino_weight_bytes(Cout, Cin, prec) : -1;
}

extern "C" int creste_conv_wino_pack_weight(const float* w, const float* scale, void* wpk, int Cout, int Cin, int prec,
                                            void* stream) {
  CRESTE_REQUIRE(w && wpk && conv_wino_supported(prec, 3, 3, 1, Cin, Cout), "conv_wino_pack_weight: bad args / shape not built");
  return conv_wino_pack(w, scale, wpk, Cout, Cin, prec, (hipStream_t)stream);
}

extern "C" int64_t creste_conv_wino_workspace_bytes(int N, int Ho, int Wo, int Cout) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return -1;
  return conv_wino_workspace_bytes(N, Ho, Wo, Cout);
}

extern "C" int creste_conv_wino4_supported(int prec, int KH, int KW, int stride, int Cin, int Cout) {
  return conv_wino4_supported(prec, KH, KW, stride, Cin, Cout) ? 1 : 0;
}

extern "C" int64_t creste_conv_wino4_weight_bytes(int Cout, int Cin, int prec) {
  return conv_wino4_supported(prec, 3, 3, 1, Cin, Cout) ? conv_wino4_weight_bytes(Cout, Cin, prec) : -1;
}

extern "C" int creste_conv_wino4_pack_weight(const float* w, const float* scale, void* wpk, int Cout, int Cin, int prec,
                                             void* stream) {
  CRESTE_REQUIRE(w && wpk && conv_wino4_supported(prec, 3, 3, 1, Cin, Cout), "conv_wino4_pack_weight: bad args / shape not built");
  return conv_wino4_pack(w, scale, wpk, Cout, Cin, prec, (hipStream_t)stream);
}

extern "C" int64_t creste_conv_wino4_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int prec) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || !conv_wino4_supported(prec, 3, 3, 1, Cin, Cout)) return -1;
  return conv_wino4_workspace_bytes(N, Ho, Wo, Cin, Cout, prec);
}

extern "C" int64_t creste_conv_packed_weight_bytes(int Cout, int Cin, int KH, int KW, int prec) {
  if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return -1;
  if (prec != CRESTE_PREC_F32)
    return conv_patch_supported(prec, KH, KW, 1) || conv_patch_supported(prec, KH, KW, 2)
               ? conv_patch_weight_bytes(Cout, Cin, KH, prec) : -1;
  const int bn = pick_bn(Cout);
  return (int64_t)round_up(Cout, bn) * KH * KW * round_up(Cin, BK) * 4;
}

extern "C" int creste_conv_pack_weight(const float* w, const float* scale, void* wpk, int Cout,
                                       int Cin, int KH, int KW, int prec, void* stream) {
  CRESTE_REQUIRE(w && wpk && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "conv_pack_weight: bad args");
  if (prec != CRESTE_PREC_F32) {
    CRESTE_REQUIRE(conv_patch_supported(prec, KH, KW, 1), "conv_pack_weight: %dx%d not built for precision %d", KH, KW, prec);
    CRESTE_REQUIRE(prec != CRESTE_PREC_F16X3, "conv_pack_weight: F16X3 weights are packed by creste_conv_pack_weight_f16");
    return conv_patch_pack(w, scale, wpk, nullptr, Cout, Cin, KH, prec, (hipStream_t)stream);
  }
  const int cin_pad = round_up(Cin, BK), cout_pad = round_up(Cout, pick_bn(Cout));
  const long total = (long)cout_pad * KH * KW * cin_pad;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  pack_weight_f32_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(w, scale, (float*)wpk, Cout, Cin,
                                                                 KH, KW, cin_pad, cout_pad);
  CRESTE_CHECK_LAUNCH("pack_weight_f32");
  return CRESTE_OK;
}

extern "C" int creste_conv_pack_weight_f16(const float* w, const float* scale, void* wpk, float* w_unscale,
                                           int Cout, int Cin, int KH, int KW, void* stream) {
  CRESTE_REQUIRE(w && wpk && w_unscale && Cout > 0 && Cin > 0, "conv_pack_weight_f16: bad args");
  CRESTE_REQUIRE(conv_patch_supported(CRESTE_PREC_F16X3, KH, KW, 1) || conv_patch_supported(CRESTE_PREC_F16X3, KH, KW, 2),
                 "conv_pack_weight_f16: %dx%d not built", KH, KW);
  return conv_patch_pack(w, scale, wpk, w_unscale, Cout, Cin, KH, CRESTE_PREC_F16X3, (hipStream_t)stream);
}

extern "C" int creste_conv2d_nhwc(const creste_conv_desc* d, void* stream) {
  CRESTE_REQUIRE(d != nullptr, "conv2d: null descriptor");
  CRESTE_REQUIRE(d->wpk && d->out && (d->in || (d->up_src && d->up_C == d->Cin)), "conv2d: null tensor pointer");
  if (d->up_src) {
    CRESTE_REQUIRE(d->algo == CRESTE_ALGO_WINOGRAD4, "conv2d: the fused upsample + concat input is built for CRESTE_ALGO_WINOGRAD4 only");
    CRESTE_REQUIRE(d->up_C > 0 && d->up_C % 4 == 0 && d->up_C <= d->Cin && d->up_cs % 4 == 0 && d->up_cs >= d->up_C &&
                       d->H == 2 * d->up_H && d->W == 2 * d->up_W && (reinterpret_cast<uintptr_t>(d->up_src) & 15) == 0,
                   "conv2d: up_src must be [N, H/2, W/2, up_cs] with up_C (%d) a multiple of 4 and <= Cin (%d)", d->up_C, d->Cin);
  }
  CRESTE_REQUIRE(d->algo == CRESTE_ALGO_DIRECT || d->algo == CRESTE_ALGO_WINOGRAD || d->algo == CRESTE_ALGO_WINOGRAD4, "conv2d: unknown algo %d", d->algo);
  CRESTE_REQUIRE(d->prec == CRESTE_PREC_F32 || conv_patch_supported(d->prec, d->KH, d->KW, d->stride),
                 "conv2d: precision %d not built for %dx%d stride %d", d->prec, d->KH, d->KW, d->stride);
  CRESTE_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 &&
                     d->Wo > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0,
                 "conv2d: non-positive dimension");
  CRESTE_REQUIRE(d->Cin % 4 == 0 && (!d->in || (d->in_cs % 4 == 0 && d->in_cs >= d->Cin - (d->up_src ? d->up_C : 0))),
                 "conv2d: Cin (%d) and in_cs (%d) must be multiples of 4, in_cs >= Cin", d->Cin, d->in_cs);
  CRESTE_REQUIRE((reinterpret_cast<uintptr_t>(d->in) & 15) == 0, "conv2d: input not 16-byte aligned");
  CRESTE_REQUIRE(!(d->flags & (CRESTE_CONV_REPLICATE_PAD | CRESTE_CONV_PHASE2X)) || d->algo == CRESTE_ALGO_WINOGRAD4,
                 "conv2d: REPLICATE_PAD / PHASE2X are flags of the F(4x4,3x3) path");
  CRESTE_REQUIRE(d->out_cs >= d->out_co + ((d->flags & CRESTE_CONV_PHASE2X) ? d->Cout / 4 : d->Cout), "conv2d: output slice exceeds out_cs");
  CRESTE_REQUIRE(!d->res || d->res_cs >= d->Cout, "conv2d: res_cs < Cout");
  CRESTE_REQUIRE((long)d->N * d->Ho * d->Wo < (1L << 31), "conv2d: M overflows int32");
  // sanity: the last output pixel may lie in trailing padding (the input gradient of a strided conv has rows the
  // forward never sampled), but not further than one kernel beyond the input
  CRESTE_REQUIRE((d->Ho - 1) * d->stride - d->pad_t < d->H + d->KH && (d->Wo - 1) * d->stride - d->pad_l < d->W + d->KW,
                 "conv2d: output extent outside the input");
  // split-operand engines: stride-2 and K>3 convs run on the row-at-a-time kernel, whose loader has no per-sample
  // gate and packs (iy << 16) | ix into one register
  const bool row_kernel = d->prec != CRESTE_PREC_F32 && (d->stride != 1 || d->KH > 3);
  CRESTE_REQUIRE(!row_kernel || !d->a_scale,
                 "conv2d: the per-sample input gate is built for stride-1 1x1/3x3 convs only on the split-operand engines");
  CRESTE_REQUIRE(!row_kernel || (d->H < 32768 && d->W < 65536),
                 "conv2d: %dx%d input exceeds the row kernel's packed coordinate range", d->H, d->W);
  CRESTE_REQUIRE(d->prec != CRESTE_PREC_F16X3 || (d->a_amax && d->w_unscale),
                 "conv2d: F16X3 needs a_amax (device bound of |in|) and w_unscale (from creste_conv_pack_weight_f16)");
  CRESTE_REQUIRE(!d->out_stats || creste_conv_stat_rows(d) > 0, "conv2d: this kernel keeps no out_stats (creste_conv_stat_rows)");
  if (d->algo == CRESTE_ALGO_WINOGRAD) return conv_wino_run(d, (hipStream_t)stream);
  if (d->algo == CRESTE_ALGO_WINOGRAD4) return conv_wino4_run(d, (hipStream_t)stream);
  if (d->prec != CRESTE_PREC_F32) return conv_patch_run(d, (hipStream_t)stream);
  ConvArgs a;
  a.in = d->in; a.wpk = (const float*)d->wpk; a.bias = d->bias; a.res = d->res;
  a.a_scale = d->a_scale; a.row_mask = d->row_mask; a.out = d->out; a.out_amax = d->out_amax;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.in_cs = d->in_cs;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.out_cs = d->out_cs; a.out_co = d->out_co;
  a.res_cs = d->res_cs; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t;
  a.pad_l = d->pad_l; a.act = d->act;
  a.cin_pad = round_up(d->Cin, BK);
  a.ktot = d->KH * d->KW * a.cin_pad;
  a.M = d->N * d->Ho * d->Wo;
  const int bn = pick_bn(d->Cout);
  a.tiles_m = (a.M + 127) / 128;
  a.tiles_n = round_up(d->Cout, bn) / bn;
  const dim3 grid(a.tiles_m * a.tiles_n), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (bn == 128) conv_igemm_f32_kernel<2, 2, 2, 2><<<grid, block, 0, s>>>(a);
  else if (bn == 64) conv_igemm_f32_kernel<2, 2, 2, 1><<<grid, block, 0, s>>>(a);
  else conv_igemm_f32_kernel<4, 1, 1, 1><<<grid, block, 0, s>>>(a);
  CRESTE_CHECK_LAUNCH("conv_igemm_f32");
  return CRESTE_OK;
}
