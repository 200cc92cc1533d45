// Memory-bound NHWC kernels of the perception path: depthwise conv, squeeze-excite, bilinear
// upsample+concat, 2x2 max-pool, layout transposes, depth-bin expectation, pixel geometry + z-MLP.
// All are HBM/L2-bandwidth bound: one thread owns 4 consecutive channels (16-B accesses, lanes run
// along the channel axis first so a wave touches whole contiguous pixels), grid-stride loops.
#include "common.h"

namespace creste {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// (nontemporal stores for the upsample / depthwise outputs: measured neutral on the step, -18 % in the conv epilogues of
// the bf16 mode -- the next layer finds part of a map in the 256 MB Infinity Cache, which streaming stores bypass; only
// the splat's 403 MB output, which nothing re-reads soon, uses them)

static inline int grid_for(long work_items, int block = 256, int max_blocks = 256 * 16) {
  long b = (work_items + block - 1) / block;
  if (b < 1) b = 1;
  return (int)(b > max_blocks ? max_blocks : b);
}

// ------------------------------------------------------------------------------ depthwise conv
template <int K>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ in,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ out, int N, int H, int W,
                                                     int C, int Ho, int Wo, int stride, int pad_t,
                                                     int pad_l, int act) {
  const int cq = C >> 2;
  const long total = (long)N * Ho * Wo * cq;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    long t = i / cq;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    f32x4 acc = bias ? ld4(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int iy0 = oy * stride - pad_t, ix0 = ox * stride - pad_l;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = iy0 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ix0 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 x = ld4(in + (((long)n * H + iy) * W + ix) * C + c);
        const f32x4 ww = ld4(w + (ky * K + kx) * C + c);
        acc += x * ww;
      }
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = act_apply(acc[j], act);
    st4(out + i * 4, o);
  }
}

// ------------------------------------------------------------------------------ squeeze-excite
// pixels reduced by one block: >= 16 per pixel-slice (so a thread's serial loop stays short when the
// channel count is large) and large enough that an image needs <= 256 blocks.
__host__ __device__ inline int se_rows_per_block(int HW, int C) {
  const int cq = C >> 2;
  const int slices = 256 / cq > 0 ? 256 / cq : 1;
  const int a = slices * 16, b = (HW + 255) / 256;
  return a > b ? a : b;
}
   // pixels reduced by one block of the partial pass

// partial[n][chunk][c] = sum over the chunk's pixels of x[n][p][c]   (deterministic order)
__global__ __launch_bounds__(256) void se_partial_kernel(const float* __restrict__ x,
                                                         float* __restrict__ partial, int HW, int C,
                                                         int nchunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [slices][C]
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int cq = C >> 2;
  const int slices = 256 / cq > 0 ? 256 / cq : 1;     // pixel slices processed in parallel
  const int rows = se_rows_per_block(HW, C);
  const int p0 = chunk * rows;
  const int p1 = min(HW, p0 + rows);
  // thread -> (slice, channel quad); when C/4 > 256 a thread loops over several quads
  for (int q0 = 0; q0 < cq; q0 += 256) {
    const int tq = (cq >= 256) ? q0 + threadIdx.x : threadIdx.x % cq;
    const int sl = (cq >= 256) ? 0 : threadIdx.x / cq;
    const bool active = tq < cq && sl < slices;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (active)
      for (int p = p0 + sl; p < p1; p += slices) s += ld4(x + ((long)n * HW + p) * C + tq * 4);
    if (active) st4(sm + (long)sl * C + tq * 4, s);
    __syncthreads();
    if (active && sl == 0) {
      f32x4 tot = s;
      for (int k = 1; k < slices; ++k) tot += ld4(sm + (long)k * C + tq * 4);
      st4(partial + ((long)n * nchunk + chunk) * C + tq * 4, tot);
    }
    __syncthreads();
  }
}

// depthwise conv + bias + activation AND the squeeze-excite partial channel sums of its OUTPUT in one pass
// (the SE mean never re-reads the up-to-1.1 GB activated tensor), plus the running |max| of the output for
// the f16x3 conv that consumes it.
// Thread = (pixel slice, channel quad) as in se_partial_kernel; the unit of work is a GROUP of XG = 4
// horizontally adjacent outputs: per kernel row the (XG-1)*S+K input quads are loaded once and feed all
// four outputs (K=5, S=1: 8 loads instead of 20), the K*K weight quads stay in registers for the thread's
// whole run.  L1 traffic per output drops 50 -> 10 loads; accumulation order per output is unchanged
// (bias, then ky-major / kx-minor taps), so results are bit-identical to the one-output-per-thread form.
// Chunks are remapped so that one XCD's L2 sees a contiguous band of rows (vertical halo reuse).
template <int K, int S>
__global__ __launch_bounds__(512) void dwconv_se_kernel(const float* __restrict__ in,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ out, float* __restrict__ partial,
                                                        float* __restrict__ amax, int H, int W, int C, int Ho,
                                                        int Wo, int pad_t, int pad_l, int act, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [slices][C]
  constexpr int XG = 4, NI = (XG - 1) * S + K;
  const int n = blockIdx.y, chunk = xcd_remap(blockIdx.x, nchunk);
  const int cq = C >> 2;
  // blockDim.x = cq * slices for cq <= 512 (every thread owns one channel quad of one pixel slice: no idle lanes at 168 or
  // 288 quads, where 256-thread workgroups left 34 % / 44 % of their threads without work), else 256 threads looping
  const int nt = blockDim.x;
  const int slices = nt / cq > 0 ? nt / cq : 1;
  const int gw = (Wo + XG - 1) / XG;                 // groups per output row
  const int G = Ho * gw;
  const int per = (G + nchunk - 1) / nchunk;
  const int g0 = chunk * per, g1 = min(G, g0 + per);
  const long HWo = (long)Ho * Wo;
  float vmax = 0.f;
  for (int q0 = 0; q0 < cq; q0 += nt) {
    const int tq = (cq >= nt) ? q0 + threadIdx.x : threadIdx.x % cq;
    const int sl = (cq >= nt) ? 0 : threadIdx.x / cq;
    const bool active = tq < cq && sl < slices;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      const int c = tq * 4;
      const f32x4 b4 = bias ? ld4(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 wk[K * K];
#pragma unroll
      for (int t = 0; t < K * K; ++t) wk[t] = ld4(w + t * C + c);
      for (int g = g0 + sl; g < g1; g += slices) {
        const int oy = g / gw, ox0 = (g - oy * gw) * XG;
        const int iy0 = oy * S - pad_t, ix0 = ox0 * S - pad_l;
        f32x4 acc[XG];
#pragma unroll
        for (int o = 0; o < XG; ++o) acc[o] = b4;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const int iy = iy0 + ky;
          if ((unsigned)iy >= (unsigned)H) continue;
          const float* row = in + ((long)n * H + iy) * W * C + c;
          f32x4 xin[NI];
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int ix = ix0 + j;
            xin[j] = (unsigned)ix < (unsigned)W ? ld4(row + (long)ix * C) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int o = 0; o < XG; ++o) {
              acc[o] += xin[o * S + kx] * wk[ky * K + kx];   // a tap outside the image adds 0 * w = +-0
            }
        }
#pragma unroll
        for (int o = 0; o < XG; ++o) {
          if (ox0 + o < Wo) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = act_apply(acc[o][j], act);
              vmax = fmaxf(vmax, fabsf(v[j]));
            }
            st4(out + ((long)n * HWo + (long)oy * Wo + ox0 + o) * C + c, v);
            s += v;
          }
        }
      }
      st4(sm + (long)sl * C + tq * 4, s);
    }
    __syncthreads();
    if (active && sl == 0) {
      f32x4 tot = s;
      for (int k = 1; k < slices; ++k) tot += ld4(sm + (long)k * C + tq * 4);
      st4(partial + ((long)n * nchunk + chunk) * C + tq * 4, tot);
    }
    __syncthreads();
  }
  if (amax) block_amax_update(vmax, amax, sm);
}

// mean -> FC1 + swish -> FC2 -> sigmoid.  Grid (sample, channel slice): every workgroup forms the sample's mean and the Cse hidden
// units itself (a few microseconds of L2 reads) and then only ITS slice of the C gates -- one workgroup per sample left the
// chip idle behind three dependent load phases (36 us for C = 1152 at batch 8, 16 such launches per forward).  Summation orders
// are those of the one-workgroup form, so the gates are the same bits: chunk sums in G interleaved groups of (even, odd) pairs,
// FC1 per lane over c = lane + 64 i then an xor butterfly, FC2 serially over the hidden units.
constexpr int SEG_T = 1024;
constexpr int SEG_SLICE = 128;                           // gates per workgroup
__global__ __launch_bounds__(SEG_T) void se_gate_kernel(const float* __restrict__ partial,
                                                        const float* __restrict__ w1,
                                                        const float* __restrict__ b1,
                                                        const float* __restrict__ w2,
                                                        const float* __restrict__ b2,
                                                        float* __restrict__ gate, int HW, int C, int Cse,
                                                        int nchunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // mean[C] + hid[Cse] + gsum[G][C]
  float* mean = sm;
  float* hid = sm + C;
  const int n = blockIdx.x;
  const float inv = 1.f / (float)HW;
  // G thread groups split the chunk list (fixed assignment -> deterministic), then a serial G-way add
  const int G = C < SEG_T ? SEG_T / C : 1;
  float* gsum = hid + Cse;                               // [G][C] scratch
  for (int c0 = 0; c0 < C; c0 += SEG_T) {
    const int c = c0 + (G > 1 ? threadIdx.x % C : threadIdx.x);
    const int g = G > 1 ? threadIdx.x / C : 0;
    if (c < C && g < G) {
      const float* pp = partial + (long)n * nchunk * C + c;
      float s0 = 0.f, s1 = 0.f;
      int k = g;
      for (; k + 3 * G < nchunk; k += 4 * G) {            // four independent loads in flight, summed as two (even, odd) steps
        const float a0 = pp[(long)k * C], a1 = pp[(long)(k + G) * C], a2 = pp[(long)(k + 2 * G) * C], a3 = pp[(long)(k + 3 * G) * C];
        s0 += a0; s1 += a1; s0 += a2; s1 += a3;
      }
      for (; k + G < nchunk; k += 2 * G) {
        const float a0 = pp[(long)k * C], a1 = pp[(long)(k + G) * C];
        s0 += a0; s1 += a1;
      }
      if (k < nchunk) s0 += pp[(long)k * C];
      gsum[g * C + c] = s0 + s1;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += SEG_T) {
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += gsum[g * C + c];
    mean[c] = s * inv;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = wave; j < Cse; j += SEG_T / 64) {         // one wave per squeezed channel
    float s = 0.f;
    int c = lane;
    for (; c + 192 < C; c += 256) {                      // four loads in flight, added in the order of c
      const float a0 = w1[(long)j * C + c], a1 = w1[(long)j * C + c + 64], a2 = w1[(long)j * C + c + 128], a3 = w1[(long)j * C + c + 192];
      s += a0 * mean[c]; s += a1 * mean[c + 64]; s += a2 * mean[c + 128]; s += a3 * mean[c + 192];
    }
    for (; c < C; c += 64) s += w1[(long)j * C + c] * mean[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) hid[j] = act_apply(s + b1[j], CRESTE_ACT_SWISH);
  }
  __syncthreads();
  const int c_lo = blockIdx.y * SEG_SLICE, c_hi = c_lo + SEG_SLICE < C ? c_lo + SEG_SLICE : C;
  for (int c = c_lo + threadIdx.x; c < c_hi; c += SEG_T) {
    float s = b2[c];
    const float* wr = w2 + (long)c * Cse;
    int j = 0;
    if ((Cse & 3) == 0) {                                // rows of w2 are 16-byte aligned: the row in float4 loads, four in flight
      for (; j + 16 <= Cse; j += 16) {
        const f32x4 q0 = ld4(wr + j), q1 = ld4(wr + j + 4), q2 = ld4(wr + j + 8), q3 = ld4(wr + j + 12);
#pragma unroll
        for (int i = 0; i < 4; ++i) s += q0[i] * hid[j + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) s += q1[i] * hid[j + 4 + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) s += q2[i] * hid[j + 8 + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) s += q3[i] * hid[j + 12 + i];
      }
      for (; j + 4 <= Cse; j += 4) {
        const f32x4 q0 = ld4(wr + j);
#pragma unroll
        for (int i = 0; i < 4; ++i) s += q0[i] * hid[j + i];
      }
    }
    for (; j < Cse; ++j) s += wr[j] * hid[j];
    gate[(long)n * C + c] = 1.f / (1.f + expf(-s));
  }
}

// ------------------------------------------------------------------------------ upsample + concat
// One block row per output row (blockIdx.y = n*Ho + oy): the vertical interpolation taps are block
// constants and the per-element index math is 32-bit (the flat 64-bit div/mod form was ALU-bound at a
// third of the HBM rate).  Also raises *amax to max|out| (both halves) for the f16x3 consumer.
__global__ __launch_bounds__(256) void upsample_concat_kernel(
    const float* __restrict__ x1, int H1, int W1, int C1, int x1_cs, const float* __restrict__ skip,
    int C2, int skip_cs, float* __restrict__ out, int Ho, int Wo, int out_cs, int out_co,
    float rh, float rw, float* __restrict__ amax) {
  __shared__ float scratch[4];
  const int cq = (C1 + C2) >> 2, c2q = C2 >> 2;
  const int n = blockIdx.y / Ho, oy = blockIdx.y - n * Ho;
  // PyTorch area_pixel_compute_source_index (align_corners=False): clamp below at 0
  float sy = rh * ((float)oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
  const int y0 = (int)sy;
  const int y1 = y0 + (y0 < H1 - 1 ? 1 : 0);
  const float ly = sy - (float)y0, hy = 1.f - ly;
  const long orow = ((long)n * Ho + oy) * Wo;
  float vmax = 0.f;
  // a thread keeps ONE channel quad and walks output columns (no per-element division); a skip-connection quad takes
  // the same four-tap path with all taps on its own pixel and weights (1, 0, 0, 0) -- no divergent branch around loads
  const int qpp = cq < 256 ? cq : 256;                       // quads handled per pass of the workgroup
  const int xpp = 256 / qpp;                                 // output columns per pass
  for (int q0 = 0; q0 < cq; q0 += qpp) {
    const int q = q0 + (int)threadIdx.x % qpp, xs = (int)threadIdx.x / qpp;
    if (q >= cq || xs >= xpp) continue;
    const bool is_skip = q < c2q;
    const int c = is_skip ? q * 4 : (q - c2q) * 4;
    const float* r0 = is_skip ? skip + orow * skip_cs + c : x1 + ((long)n * H1 + y0) * W1 * x1_cs + c;
    const float* r1 = is_skip ? r0 : x1 + ((long)n * H1 + y1) * W1 * x1_cs + c;
    const int cs = is_skip ? skip_cs : x1_cs;
    for (int ox = blockIdx.x * xpp + xs; ox < Wo; ox += gridDim.x * xpp) {
      float sx = rw * ((float)ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
      int x0 = (int)sx;
      int x1i = x0 + (x0 < W1 - 1 ? 1 : 0);
      float lx = sx - (float)x0, wy1 = ly;
      if (is_skip) { x0 = x1i = ox; lx = 0.f; wy1 = 0.f; }
      const float hx = 1.f - lx, wy0 = 1.f - wy1;
      const f32x4 v00 = ld4(r0 + (long)x0 * cs), v01 = ld4(r0 + (long)x1i * cs);
      const f32x4 v10 = ld4(r1 + (long)x0 * cs), v11 = ld4(r1 + (long)x1i * cs);
      const f32x4 v = is_skip ? v00 : wy0 * (hx * v00 + lx * v01) + wy1 * (hx * v10 + lx * v11);
      vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      st4(out + (orow + ox) * out_cs + out_co + q * 4, v);
    }
  }
  (void)hy;
  if (amax) block_amax_update(vmax, amax, scratch);
}

// Exact 2x variant (every Up block of the encoder and the heads' second upsample): a thread owns one channel quad of
// one LOW-resolution pixel and writes the 2 x 2 output pixels it maps to -- the 3 x 3 source neighbourhood is loaded
// once (9 quad loads per 4 outputs instead of 16; the generic kernel runs at 2.7 TB/s on vector-memory ISSUE, not on
// bytes).  Each output is formed with the same taps, the same weights and the same expression as the generic kernel
// (PyTorch's source-index rule, clamped at the borders), so the results are bit-identical.
__global__ __launch_bounds__(256) void upsample2x_concat_kernel(
    const float* __restrict__ x1, int H1, int W1, int C1, int x1_cs, const float* __restrict__ skip, int C2,
    int skip_cs, float* __restrict__ out, int out_cs, int out_co, float* __restrict__ amax) {
  __shared__ float scratch[4];
  const int cq = (C1 + C2) >> 2, c2q = C2 >> 2;
  const int n = blockIdx.y / H1, y = blockIdx.y - n * H1;
  const int Ho = 2 * H1, Wo = 2 * W1;
  const int qpp = cq < 256 ? cq : 256, xpp = 256 / qpp;
  float vmax = 0.f;
  // vertical taps of output rows 2y and 2y + 1 (align_corners = False, scale 0.5): src = 0.5 * (dst + 0.5) - 0.5
  float wyb[2];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    float sy = 0.5f * ((float)(2 * y + dy) + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
    wyb[dy] = sy - (float)(int)sy;
  }
  for (int q0 = 0; q0 < cq; q0 += qpp) {
    const int q = q0 + (int)threadIdx.x % qpp, xs = (int)threadIdx.x / qpp;
    if (q >= cq || xs >= xpp) continue;
    float* o0 = out + ((long)(n * Ho + 2 * y) * Wo) * out_cs + out_co + q * 4;
    float* o1 = o0 + (long)Wo * out_cs;
    if (q < c2q) {                                          // skip connection: a 2 x 2 copy
      const float* s0 = skip + ((long)(n * Ho + 2 * y) * Wo) * skip_cs + q * 4;
      const float* s1 = s0 + (long)Wo * skip_cs;
      for (int x = blockIdx.x * xpp + xs; x < W1; x += gridDim.x * xpp) {
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const f32x4 a = ld4(s0 + (long)(2 * x + dx) * skip_cs), b = ld4(s1 + (long)(2 * x + dx) * skip_cs);
#pragma unroll
          for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fmaxf(fabsf(a[j]), fabsf(b[j])));
          st4(o0 + (long)(2 * x + dx) * out_cs, a);
          st4(o1 + (long)(2 * x + dx) * out_cs, b);
        }
      }
      continue;
    }
    const int c = (q - c2q) * 4;
    // rows y-1 (clamped), y, y+1 (clamped)
    const float* rm = x1 + ((long)n * H1 + (y > 0 ? y - 1 : 0)) * W1 * x1_cs + c;
    const float* r0 = x1 + ((long)n * H1 + y) * W1 * x1_cs + c;
    const float* rp = x1 + ((long)n * H1 + (y < H1 - 1 ? y + 1 : y)) * W1 * x1_cs + c;
    for (int x = blockIdx.x * xpp + xs; x < W1; x += gridDim.x * xpp) {
      const int xm = x > 0 ? x - 1 : 0, xp = x < W1 - 1 ? x + 1 : x;
      f32x4 v[3][3];
      v[0][0] = ld4(rm + (long)xm * x1_cs); v[0][1] = ld4(rm + (long)x * x1_cs); v[0][2] = ld4(rm + (long)xp * x1_cs);
      v[1][0] = ld4(r0 + (long)xm * x1_cs); v[1][1] = ld4(r0 + (long)x * x1_cs); v[1][2] = ld4(r0 + (long)xp * x1_cs);
      v[2][0] = ld4(rp + (long)xm * x1_cs); v[2][1] = ld4(rp + (long)x * x1_cs); v[2][2] = ld4(rp + (long)xp * x1_cs);
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        // output row 2y + dy takes neighbourhood rows (dy, dy + 1) and output column 2x + dx columns (dx, dx + 1): in
        // the interior those ARE PyTorch's (y0, y1) / (x0, x1); at a border the clamped duplicate holds the same data
        // as the tap it stands in for, or its weight is exactly 0
        const float wy1 = wyb[dy], wy0 = 1.f - wy1;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float sx = 0.5f * ((float)(2 * x + dx) + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
          const float lx = sx - (float)(int)sx, hx = 1.f - lx;
          const f32x4 r = wy0 * (hx * v[dy][dx] + lx * v[dy][dx + 1]) + wy1 * (hx * v[dy + 1][dx] + lx * v[dy + 1][dx + 1]);
#pragma unroll
          for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(r[j]));
          st4((dy ? o1 : o0) + (long)(2 * x + dx) * out_cs, r);
        }
      }
    }
  }
  if (amax) block_amax_update(vmax, amax, scratch);
}

// Exact 2x, wide maps: the same arithmetic with the low-resolution rows in an LDS ring.  The register kernel above
// loads every source quad nine times (once per 3 x 3 neighbourhood that contains it): 2.25 bytes read per byte
// written -- measured, its stores alone take 186 us of the 321 us on the heads' 256-channel map, the loads the rest.
// Here a workgroup owns a strip of TW source columns and walks down a band of source rows; each source row is
// fetched ONCE per strip (+2 halo columns) into a four-slot ring -- the fetch of row y + 2 is in flight while the two
// output rows of source row y are formed from slots (y-1, y, y+1) -- one barrier per source row.
#ifndef UPS_RING_LDS
#define UPS_RING_LDS (52 * 1024)   // three workgroups per CU: measured best (26 KB: 310 us, 52: 271, 80: 292 on the heads' map)
#endif
template <int TW>
__global__ __launch_bounds__(256) void upsample2x_ring_kernel(
    const float* __restrict__ x1, int H1, int W1, int C1, int x1_cs, const float* __restrict__ skip, int C2,
    int skip_cs, float* __restrict__ out, int out_cs, int out_co, float* __restrict__ amax, int rows_per_band,
    int nstrips) {
  constexpr int IW = TW + 2;
  extern __shared__ __attribute__((aligned(16))) float ring[];            // [4][IW][C1]
  __shared__ float scratch[4];
  const int c1q = C1 >> 2, c2q = C2 >> 2, cq = c1q + c2q;
  const int n = blockIdx.y, strip = blockIdx.x % nstrips, band = blockIdx.x / nstrips;
  const int xs0 = strip * TW, y0 = band * rows_per_band, y1 = min(H1, y0 + rows_per_band);
  const int Ho = 2 * H1, Wo = 2 * W1;
  const int tid = threadIdx.x;
  const int nld = IW * c1q;                                               // quads of one staged row
  constexpr int NPRE = 6;                                                 // staged quads per thread (host: nld <= 6 * 256)
  f32x4 pre[NPRE];
  auto fetch = [&](int row) __attribute__((always_inline)) {              // source row `row` (clamped columns) -> registers
    const float* rp = x1 + ((long)n * H1 + row) * W1 * x1_cs;
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int e = tid + 256 * j;
      const int px = e / c1q, q = e - px * c1q;
      int xc = xs0 - 1 + px; xc = xc < 0 ? 0 : (xc > W1 - 1 ? W1 - 1 : xc);
      pre[j] = ld4(rp + (long)xc * x1_cs + 4 * (e < nld ? q : 0));
    }
  };
  auto commit = [&](int row) __attribute__((always_inline)) {
    float* dst = ring + (size_t)(row & 3) * IW * C1;
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int e = tid + 256 * j;
      if (e < nld) st4(dst + (size_t)e * 4, pre[j]);                       // [px][q] == e
    }
  };
  // prologue: rows y0-1 (clamped), y0, y0+1 (clamped)
  const int rfirst = y0 > 0 ? y0 - 1 : 0, rlast = min(H1 - 1, y0 + 1);
  for (int r = rfirst; r <= rlast; ++r) { fetch(r); commit(r); }
  __syncthreads();
  float vmax = 0.f;
  for (int y = y0; y < y1; ++y) {
    const bool more = y + 2 <= H1 - 1 && y + 1 < y1;                      // the next source row needs row y + 2
    if (more) fetch(y + 2);
    const int ym = y > 0 ? y - 1 : 0, yp = y < H1 - 1 ? y + 1 : y;
    const float* rows[3] = {ring + (size_t)(ym & 3) * IW * C1, ring + (size_t)(y & 3) * IW * C1,
                            ring + (size_t)(yp & 3) * IW * C1};
    float wyb[2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      float sy = 0.5f * ((float)(2 * y + dy) + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
      wyb[dy] = sy - (float)(int)sy;
    }
    float* o0 = out + ((long)(n * Ho + 2 * y) * Wo) * out_cs + out_co;
    float* o1 = o0 + (long)Wo * out_cs;
    for (int e = tid; e < TW * cq; e += 256) {
      const int xl = e / cq, q = e - xl * cq, x = xs0 + xl;
      if (x >= W1) break;
      if (q < c2q) {                                                       // skip connection: a 2 x 2 copy
        const float* s0 = skip + ((long)(n * Ho + 2 * y) * Wo) * skip_cs + q * 4;
        const float* s1 = s0 + (long)Wo * skip_cs;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const f32x4 a = ld4(s0 + (long)(2 * x + dx) * skip_cs), b = ld4(s1 + (long)(2 * x + dx) * skip_cs);
#pragma unroll
          for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fmaxf(fabsf(a[j]), fabsf(b[j])));
          st4(o0 + (long)(2 * x + dx) * out_cs + q * 4, a);
          st4(o1 + (long)(2 * x + dx) * out_cs + q * 4, b);
        }
        continue;
      }
      // staged pixel index of source column x is xl + 1; its clamped neighbours: at the map's borders the duplicate
      const int pm = x > 0 ? xl : xl + 1, pp = x < W1 - 1 ? xl + 2 : xl + 1;
      const int c = (q - c2q) * 4;
      f32x4 v[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        v[i][0] = ld4(rows[i] + pm * C1 + c);
        v[i][1] = ld4(rows[i] + (xl + 1) * C1 + c);
        v[i][2] = ld4(rows[i] + pp * C1 + c);
      }
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const float wy1 = wyb[dy], wy0 = 1.f - wy1;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float sx = 0.5f * ((float)(2 * x + dx) + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
          const float lx = sx - (float)(int)sx, hx = 1.f - lx;
          const f32x4 r = wy0 * (hx * v[dy][dx] + lx * v[dy][dx + 1]) + wy1 * (hx * v[dy + 1][dx] + lx * v[dy + 1][dx + 1]);
#pragma unroll
          for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(r[j]));
          st4((dy ? o1 : o0) + (long)(2 * x + dx) * out_cs + q * 4, r);
        }
      }
    }
    if (more) commit(y + 2);
    __syncthreads();
  }
  if (amax) block_amax_update(vmax, amax, scratch);
}

// ------------------------------------------------------------------------------ max-pool DSxDS/DS
// DS = 2: nn.MaxPool2d(2, 2) / F.max_pool2d(x, ds, ds) of the reward head (vin.py:104-106); DS = 1: the degenerate
// pool of reward_cfg.ds == 1 (a row/column crop into the output slice).
template <int DS>
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, int H, int W,
                                                      int C, int in_cs, float* __restrict__ out,
                                                      int N, int Ho, int Wo, int out_cs,
                                                      float* __restrict__ amax) {
  __shared__ float scratch[4];
  float vmax = 0.f;
  const int cq = C >> 2;
  const long total = (long)N * Ho * Wo * cq;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    long t = i / cq;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const float* b = in + (((long)n * H + DS * oy) * W + DS * ox) * in_cs + c;
    f32x4 m = ld4(b);
#pragma unroll
    for (int dy = 0; dy < DS; ++dy)
#pragma unroll
      for (int dx = 0; dx < DS; ++dx) {
        if (dy == 0 && dx == 0) continue;
        const f32x4 a = ld4(b + ((long)dy * W + dx) * in_cs);
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], a[j]);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(m[j]));
    st4(out + (((long)n * Ho + oy) * Wo + ox) * out_cs + c, m);
  }
  if (amax) block_amax_update(vmax, amax, scratch);
}

// ------------------------------------------------------------------------------ layout transposes
// NCHW -> NHWC through a 32x33 LDS tile: reads coalesced along W, writes coalesced along C.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, int C, long HW,
                                                           int out_cs) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k; const long p = p0 + tx;
    tile[k][tx] = (c < C && p < HW) ? in[((long)n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const long p = p0 + k; const int c = c0 + tx;
    if (c < C && p < HW) out[((long)n * HW + p) * out_cs + c] = tile[tx][k];
  }
}

// C == 4 (the RGB-D input): one thread per pixel -- four coalesced plane reads, one 16-byte store (the 32 x 32 tile
// kernel above would run 4 of its 32 channel rows / write lanes)
__global__ __launch_bounds__(256) void nchw4_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            long HW, int out_cs, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / HW, p = i - n * HW;
    const float* src = in + n * 4 * HW + p;
    const f32x4 v = {src[0], src[HW], src[2 * HW], src[3 * HW]};
    st4(out + i * out_cs, v);
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, int in_cs,
                                                           float* __restrict__ out, int C, long HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const long p = p0 + k; const int c = c0 + tx;
    tile[k][tx] = (c < C && p < HW) ? in[((long)n * HW + p) * in_cs + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k; const long p = p0 + tx;
    if (c < C && p < HW) out[((long)n * C + c) * HW + p] = tile[tx][k];
  }
}

// ------------------------------------------------------------------------------ depth expectation
// One 32-lane half-wave per pixel, 4 logits per lane (nbins = 128).
__global__ __launch_bounds__(256) void depth_expectation_kernel(const float* __restrict__ logits,
                                                                int cs, long P,
                                                                const float* __restrict__ bin_values,
                                                                float* __restrict__ depth_m,
                                                                int64_t* __restrict__ bins) {
  const int sub = threadIdx.x & 31;
  const long half = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 5;
  const long nhalf = ((long)gridDim.x * blockDim.x) >> 5;
  const f32x4 bv = ld4(bin_values + sub * 4);
  for (long p = half; p < P; p += nhalf) {
    const f32x4 x = ld4(logits + p * cs + sub * 4);
    float mx = x[0]; int am = sub * 4;
#pragma unroll
    for (int j = 1; j < 4; ++j) if (x[j] > mx) { mx = x[j]; am = sub * 4 + j; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {      // stays inside the 32-lane half
      const float om = __shfl_xor(mx, o);
      const int oa = __shfl_xor(am, o);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    float se = 0.f, sw = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float e = expf(x[j] - mx); se += e; sw += e * bv[j]; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { se += __shfl_xor(se, o); sw += __shfl_xor(sw, o); }
    if (sub == 0) { depth_m[p] = (sw / se) / 1000.f; bins[p] = am; }
  }
}

// ------------------------------------------------------------------------------ pixel geometry + z MLP
// Per pixel: c = [u*d, v*d, d, 1]; xyz_k = fma(P[k][3],c3, fma(P[k][2],c2, fma(P[k][1],c1, P[k][0]*c0)))
// -- the k-ordered fma chain the reference's CPU bmm produces -- plus the range
// mask and the 1 -> zhid -> zdim ReLU MLP on z.  zdim lanes cooperate on one pixel.
__global__ __launch_bounds__(256) void pixel_geometry_kernel(
    const float* __restrict__ depth, const float* __restrict__ p2p, int B, int Hs, int Ws,
    const float* __restrict__ bounds, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, int zhid, int zdim,
    float* __restrict__ xyz, float* __restrict__ mask, float* __restrict__ zfeat, int z_cs, int z_co) {
  extern __shared__ float smw[];   // w1[zhid] b1[zhid] w2[zdim*zhid] b2[zdim]
  float* s_w1 = smw; float* s_b1 = s_w1 + zhid; float* s_w2 = s_b1 + zhid; float* s_b2 = s_w2 + zdim * zhid;
  for (int i = threadIdx.x; i < zhid; i += blockDim.x) { s_w1[i] = w1[i]; s_b1[i] = b1[i]; }
  for (int i = threadIdx.x; i < zdim * zhid; i += blockDim.x)   // transposed: lanes j read consecutive words
    s_w2[(i % zhid) * zdim + i / zhid] = w2[i];
  for (int i = threadIdx.x; i < zdim; i += blockDim.x) s_b2[i] = b2[i];
  __syncthreads();
  const long P = (long)Hs * Ws, total = (long)B * P;
  const int per = blockDim.x / zdim;                       // pixels per block iteration
  const int j = threadIdx.x % zdim, slot = threadIdx.x / zdim;
  const bool fast = zhid == 64 && zdim == 32;
  float w2r[64];                                            // this lane's column of W2 (fast path)
#pragma unroll
  for (int h = 0; h < 64; ++h) w2r[h] = fast ? s_w2[h * 32 + j] : 0.f;
  // the depth of the NEXT pixel is fetched before the current one is processed (a workgroup walks ~20 pixels per
  // lane group; un-prefetched, every step waited a full memory latency: that chain, not arithmetic, was the kernel)
  const long stride = (long)gridDim.x * per;
  float d_next = 0.f;
  {
    const long g0 = (long)blockIdx.x * per + slot;
    if (g0 < total && slot < per) d_next = depth[g0];
  }
  int cur_b = -1;
  float Mr[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Mr[k] = 0.f;
  for (long base = (long)blockIdx.x * per; base < total; base += stride) {
    const long g = base + slot;
    if (g >= total || slot >= per) continue;
    const float d = d_next;
    if (g + stride < total) d_next = depth[g + stride];
    // 32-bit index arithmetic (the host checks B*Hs*Ws < 2^31): three 64-bit divisions cost more than the z-MLP
    const unsigned g32 = (unsigned)g, P32 = (unsigned)P;
    const int b = (int)(g32 / P32);
    const unsigned p = g32 - (unsigned)b * P32;
    const int v = (int)(p / (unsigned)Ws), u = (int)(p - (unsigned)v * (unsigned)Ws);
    const float c0 = (float)u * d, c1 = (float)v * d;
    if (b != cur_b) {                                         // frame changed: its projection matrix (rows 0..2)
      const float* M = p2p + (long)b * 16;
#pragma unroll
      for (int k = 0; k < 12; ++k) Mr[k] = M[k];
      cur_b = b;
    }
    float q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      q[k] = __fmaf_rn(Mr[k * 4 + 3], 1.0f, __fmaf_rn(Mr[k * 4 + 2], d, __fmaf_rn(Mr[k * 4 + 1], c1, __fmul_rn(Mr[k * 4 + 0], c0))));
    if (j == 0) {
      xyz[g * 3 + 0] = q[0]; xyz[g * 3 + 1] = q[1]; xyz[g * 3 + 2] = q[2];
      const bool ok = q[0] >= bounds[0] && q[1] >= bounds[1] && q[2] >= bounds[2] &&
                      q[0] < bounds[3] && q[1] < bounds[4] && q[2] < bounds[5];
      mask[g] = ok ? 1.f : 0.f;
    }
    float s = s_b2[j];
    if (fast) {
      // the shipped z-MLP (1 -> 64 -> 32): lane j of a pixel's 32 lanes evaluates hidden units j and j + 32 once and
      // the group walks them through shuffles -- one LDS-pipe operation (the shuffle) per hidden unit instead of three, W2's column in registers (the loop was
      // LDS-bound: 0.33 ms per batch of 16); same operations in the same order, bit-identical
      const float hv0 = fmaxf(__fmaf_rn(s_w1[j], q[2], s_b1[j]), 0.f);
      const float hv1 = fmaxf(__fmaf_rn(s_w1[j + 32], q[2], s_b1[j + 32]), 0.f);
#pragma unroll
      for (int h = 0; h < 32; ++h) s = __fmaf_rn(w2r[h], __shfl(hv0, h, 32), s);
#pragma unroll
      for (int h = 0; h < 32; ++h) s = __fmaf_rn(w2r[h + 32], __shfl(hv1, h, 32), s);
    } else {
      for (int h = 0; h < zhid; ++h) {
        const float hv = fmaxf(__fmaf_rn(s_w1[h], q[2], s_b1[h]), 0.f);
        s = __fmaf_rn(s_w2[h * zdim + j], hv, s);
      }
    }
    zfeat[g * z_cs + z_co + j] = fmaxf(s, 0.f);
  }
}

// The shipped z-MLP (1 -> 64 -> 32) with one LANE per pixel: the pixel's 64 hidden units and 32 outputs live in its own
// registers, every weight is a wave-uniform SCALAR operand (the compiler's s_load from the kernel-argument pointers), so the
// 2048 multiply-adds per pixel need no LDS and no shuffles -- the lane-per-output form above is bound by 64 shuffles per
// output (220 us per batch of 16; this form: ~70).  Same operations in the same order (geometry fma chain; s = b2[j], then
// h = 0..63): bit-identical.
// KEY: the point's BEV voxel coordinates and extended base-cell key as well -- the first kernel of the splat's binning plan
// (csrc/bev_splat.hip: splat_key_kernel, the same operations on the same floats), one launch and one read of xyz less.
struct SplatKeyArgs {
  float off_x, off_y, vox_x, vox_y;
  int GH, GW;
  float* coords;
  int* key;
};
template <bool KEY>
__global__ __launch_bounds__(256) void pixel_geometry_px_kernel(
    const float* __restrict__ depth, const float* __restrict__ p2p, int B, int Hs, int Ws,
    const float* __restrict__ bounds, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ xyz, float* __restrict__ mask,
    float* __restrict__ zfeat, int z_cs, int z_co, const SplatKeyArgs sk) {
  const unsigned P32 = (unsigned)Hs * (unsigned)Ws, total = (unsigned)B * P32;
  const unsigned g = blockIdx.x * 256u + threadIdx.x;
  if (g >= total) return;
  const float d = depth[g];
  const int b = (int)(g / P32);
  const unsigned p = g - (unsigned)b * P32;
  const int v = (int)(p / (unsigned)Ws), u = (int)(p - (unsigned)v * (unsigned)Ws);
  const float c0 = (float)u * d, c1 = (float)v * d;
  const float* M = p2p + (long)b * 16;
  float q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    q[k] = __fmaf_rn(M[k * 4 + 3], 1.0f, __fmaf_rn(M[k * 4 + 2], d, __fmaf_rn(M[k * 4 + 1], c1, __fmul_rn(M[k * 4 + 0], c0))));
  xyz[(long)g * 3 + 0] = q[0]; xyz[(long)g * 3 + 1] = q[1]; xyz[(long)g * 3 + 2] = q[2];
  const bool ok = q[0] >= bounds[0] && q[1] >= bounds[1] && q[2] >= bounds[2] &&
                  q[0] < bounds[3] && q[1] < bounds[4] && q[2] < bounds[5];
  mask[g] = ok ? 1.f : 0.f;
  if constexpr (KEY) {
    // map = lidar2map @ [x,y,z,1]: rows (0,-1,0,off_x), (-1,0,0,off_y) -> one rounding each (reference splat_projection.py:185-187)
    const float mx = __fadd_rn(-q[1], sk.off_x), my = __fadd_rn(-q[0], sk.off_y);
    const float X = __fdiv_rn(mx, sk.vox_x), Y = __fdiv_rn(my, sk.vox_y);
    *reinterpret_cast<float2*>(sk.coords + (long)g * 2) = make_float2(X, Y);
    const float fx = floorf(X), fy = floorf(Y);
    int k = -1;
    if (fx >= -1.f && fx <= (float)(sk.GW - 1) && fy >= -1.f && fy <= (float)(sk.GH - 1))
      k = ((int)fy + 1) * (sk.GW + 1) + ((int)fx + 1);
    sk.key[g] = k;
  }
  float hv[64];
#pragma unroll
  for (int h = 0; h < 64; ++h) hv[h] = fmaxf(__fmaf_rn(w1[h], q[2], b1[h]), 0.f);
  float* zo = zfeat + (long)g * z_cs + z_co;
#pragma unroll
  for (int j4 = 0; j4 < 32; j4 += 4) {
    f32x4 o;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float sacc = b2[j4 + jj];
#pragma unroll
      for (int h = 0; h < 64; ++h) sacc = __fmaf_rn(w2[(j4 + jj) * 64 + h], hv[h], sacc);
      o[jj] = fmaxf(sacc, 0.f);
    }
    st4(zo + j4, o);
  }
}

// ------------------------------------------------------------------------------ channel affine + act
// y = act(x * scale[c] + shift[c])  (an eval-mode BatchNorm that follows a ReLU and therefore cannot be
// folded into the preceding conv: MultiScaleFCN trunk, reference conv.py:118-128)
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, int x_cs,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         float* __restrict__ out, int out_cs, long P,
                                                         int C, int act) {
  const int cq = C >> 2;
  const long total = P * cq;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const long p = i / cq;
    const f32x4 v = ld4(x + p * x_cs + c), sc = ld4(scale + c), sh = ld4(shift + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = act_apply(__fadd_rn(__fmul_rn(v[j], sc[j]), sh[j]), act);
    st4(out + p * out_cs + c, o);
  }
}

// ------------------------------------------------------------------------------ planar bilinear resize
// single-channel [N,H,W] -> rows [0,Ho) of a [N,Hd,Wo] plane (F.interpolate(size=...), vin.py:121-125)
__global__ __launch_bounds__(256) void resize_plane_kernel(const float* __restrict__ in, int H, int W,
                                                           float* __restrict__ out, int N, int Ho, int Wo,
                                                           int Hd, float rh, float rw) {
  const long total = (long)N * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const int oy = (int)((i / Wo) % Ho);
    const int n = (int)(i / ((long)Wo * Ho));
    float sy = rh * ((float)oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = rw * ((float)ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* b = in + (long)n * H * W;
    out[((long)n * Hd + oy) * Wo + ox] = hy * (hx * b[(long)y0 * W + x0] + lx * b[(long)y0 * W + x1]) +
                                         ly * (hx * b[(long)y1 * W + x0] + lx * b[(long)y1 * W + x1]);
  }
}

// max |x| over an NHWC slice (C % 4 == 0, 16-byte rows): one streaming read, block reduce, rare atomic
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long pixels, int C, int cs,
                                                     float* __restrict__ amax) {
  __shared__ float scratch[4];
  const int cq = C >> 2;
  const long total = pixels * cq;
  float m = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long px = i / cq;
    const int q = (int)(i - px * cq);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + px * cs + 4 * q);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  block_amax_update(m, amax, scratch);
}

}  // namespace creste

using namespace creste;

extern "C" int creste_dwconv2d_nhwc_f32(const float* in, const float* w, const float* bias, float* out,
                                        int N, int H, int W, int C, int Ho, int Wo, int K, int stride,
                                        int pad_t, int pad_l, int act, void* stream) {
  CRESTE_REQUIRE(in && w && out, "dwconv: null pointer");
  CRESTE_REQUIRE(C % 4 == 0 && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "dwconv: bad dims (C%%4)");
  CRESTE_REQUIRE(K == 3 || K == 5, "dwconv: kernel size %d not built (3 or 5)", K);
  const long total = (long)N * Ho * Wo * (C / 4);
  const int grid = grid_for(total, 256, 256 * 32);
  hipStream_t s = (hipStream_t)stream;
  if (K == 3) dwconv_kernel<3><<<grid, 256, 0, s>>>(in, w, bias, out, N, H, W, C, Ho, Wo, stride, pad_t, pad_l, act);
  else dwconv_kernel<5><<<grid, 256, 0, s>>>(in, w, bias, out, N, H, W, C, Ho, Wo, stride, pad_t, pad_l, act);
  CRESTE_CHECK_LAUNCH("dwconv");
  return CRESTE_OK;
}

extern "C" int creste_se_partial_count(int HW, int C) {
  if (HW <= 0 || C <= 0 || (C & 3)) return -1;
  const int rows = se_rows_per_block(HW, C);
  return (HW + rows - 1) / rows;
}

extern "C" int creste_se_gate_f32(const float* x, float* partial, const float* w1, const float* b1,
                                  const float* w2, const float* b2, float* gate, int N, int HW, int C,
                                  int Cse, void* stream) {
  CRESTE_REQUIRE(partial && w1 && b1 && w2 && b2 && gate, "se_gate: null pointer");
  CRESTE_REQUIRE(C % 4 == 0 && C > 0 && Cse > 0 && N > 0 && HW > 0, "se_gate: bad dims");
  const int nchunk = creste_se_partial_count(HW, C);
  const int cq = C / 4;
  const int slices = 256 / cq > 0 ? 256 / cq : 1;
  hipStream_t s = (hipStream_t)stream;
  if (x) {   // x == NULL: `partial` was already filled by creste_dwconv_se_nhwc_f32
    se_partial_kernel<<<dim3(nchunk, N), 256, (size_t)slices * C * sizeof(float), s>>>(x, partial, HW, C, nchunk);
    CRESTE_CHECK_LAUNCH("se_partial");
  }
  se_gate_kernel<<<dim3(N, (C + SEG_SLICE - 1) / SEG_SLICE), SEG_T, (size_t)(C + Cse + (C < SEG_T ? (SEG_T / C) * C : C)) * sizeof(float), s>>>(partial, w1, b1, w2, b2, gate, HW, C, Cse, nchunk);
  CRESTE_CHECK_LAUNCH("se_gate");
  return CRESTE_OK;
}

extern "C" int creste_se_gate_partial_f32(const float* partial, int nchunk, const float* w1, const float* b1,
                                          const float* w2, const float* b2, float* gate, int N, int HW, int C, int Cse,
                                          void* stream) {
  CRESTE_REQUIRE(partial && w1 && b1 && w2 && b2 && gate, "se_gate_partial: null pointer");
  CRESTE_REQUIRE(C % 4 == 0 && C > 0 && Cse > 0 && N > 0 && HW > 0 && nchunk > 0, "se_gate_partial: bad dims");
  se_gate_kernel<<<dim3(N, (C + SEG_SLICE - 1) / SEG_SLICE), SEG_T, (size_t)(C + Cse + (C < SEG_T ? (SEG_T / C) * C : C)) * sizeof(float), (hipStream_t)stream>>>(
      partial, w1, b1, w2, b2, gate, HW, C, Cse, nchunk);
  CRESTE_CHECK_LAUNCH("se_gate");
  return CRESTE_OK;
}

extern "C" int creste_upsample_concat_nhwc_f32(const float* x1, int N, int H1, int W1, int C1, int x1_cs,
                                               const float* skip, int C2, int skip_cs, float* out,
                                               int Ho, int Wo, int out_cs, int out_co, float rh,
                                               float rw, float* out_amax, void* stream) {
  CRESTE_REQUIRE(x1 && out && (skip || C2 == 0), "upsample_concat: null pointer");
  CRESTE_REQUIRE(C1 % 4 == 0 && C2 % 4 == 0 && x1_cs % 4 == 0 && out_cs % 4 == 0 && out_co % 4 == 0 &&
                     (C2 == 0 || skip_cs % 4 == 0),
                 "upsample_concat: channel counts/strides must be multiples of 4");
  CRESTE_REQUIRE(out_cs >= out_co + C1 + C2, "upsample_concat: output slice exceeds out_cs");
  CRESTE_REQUIRE((long)N * Ho < 65536 * 32768L && (long)Wo * ((C1 + C2) / 4) < (1L << 30), "upsample_concat: extent too large");
  const int cq_all = (C1 + C2) / 4, qpp = cq_all < 256 ? cq_all : 256, xpp = 256 / qpp;
  if (Ho == 2 * H1 && Wo == 2 * W1 && rh == 0.5f && rw == 0.5f && H1 > 1 && W1 > 1) {
    // wide maps: source rows staged once per strip in an LDS ring (four slots of (TW + 2) x C1 floats, three
    // workgroups per CU); the strip is the widest that fits, and a staged row must fit the kernel's 6 register quads per thread
    int tw = 0;
    for (int c = 32; c >= 4 && !tw; c >>= 1)
      if (4L * (c + 2) * C1 * 4 <= UPS_RING_LDS && (long)(c + 2) * (C1 / 4) <= 6 * 256) tw = c;
    if (tw && (long)N * H1 * W1 >= 16384) {
      const int nstrips = (W1 + tw - 1) / tw;
      int bands = (2048 + nstrips * N - 1) / (nstrips * N);
      int rows = (H1 + bands - 1) / bands;
      if (rows < 4) rows = 4;
      bands = (H1 + rows - 1) / rows;
      const size_t smem = 4UL * (tw + 2) * C1 * 4;
      const dim3 grid(nstrips * bands, N);
      hipStream_t st = (hipStream_t)stream;
      static std::atomic<uint64_t> devs32{0}, devs16{0}, devs8{0}, devs4{0};
#define CRESTE_UPS_RING(TW_, DEVS)                                                                                       \
  do {                                                                                                                   \
    if (smem > 64 * 1024) CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(upsample2x_ring_kernel<TW_>), (int)smem, DEVS)); \
    upsample2x_ring_kernel<TW_><<<grid, 256, smem, st>>>(x1, H1, W1, C1, x1_cs, skip, C2, skip_cs, out, out_cs, out_co, \
                                                        out_amax, rows, nstrips);                                        \
  } while (0)
      if (tw == 32) CRESTE_UPS_RING(32, devs32);
      else if (tw == 16) CRESTE_UPS_RING(16, devs16);
      else if (tw == 8) CRESTE_UPS_RING(8, devs8);
      else CRESTE_UPS_RING(4, devs4);
#undef CRESTE_UPS_RING
      CRESTE_CHECK_LAUNCH("upsample2x_ring");
      return CRESTE_OK;
    }
    int bx2 = (W1 + 2 * xpp - 1) / (2 * xpp);                     // ~2 source columns (4 output columns) per thread
    if (bx2 < 1) bx2 = 1;
    upsample2x_concat_kernel<<<dim3(bx2, N * H1), 256, 0, (hipStream_t)stream>>>(x1, H1, W1, C1, x1_cs, skip, C2, skip_cs,
                                                                               out, out_cs, out_co, out_amax);
    CRESTE_CHECK_LAUNCH("upsample2x_concat");
    return CRESTE_OK;
  }
  const int passes = (cq_all + qpp - 1) / qpp;                    // quad passes per workgroup (1 unless C > 1024)
  int bx = (Wo + 2 * xpp - 1) / (2 * xpp);                        // ~2 columns per thread and pass
  if (bx < 1) bx = 1;
  (void)passes;
  upsample_concat_kernel<<<dim3(bx, N * Ho), 256, 0, (hipStream_t)stream>>>(
      x1, H1, W1, C1, x1_cs, skip, C2, skip_cs, out, Ho, Wo, out_cs, out_co, rh, rw, out_amax);
  CRESTE_CHECK_LAUNCH("upsample_concat");
  return CRESTE_OK;
}

// ------------------------------------------------------------------------------ small bookkeeping ops
// (so that an inference forward is C-ABI launches only: what the plan runtime of csrc/plan_runtime.cpp replays)
__global__ void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void max2_f32_kernel(const float* a, const float* b, float* out) { *out = fmaxf(*a, *b); }

__global__ void spin_kernel(long long ticks) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int creste_spin_us(int microseconds, int workgroups, void* stream) {
  CRESTE_REQUIRE(microseconds >= 0 && microseconds <= 100000 && workgroups >= 1 && workgroups <= 1000000,
                 "spin_us: 0 .. 100000 microseconds, 1 .. 1000000 workgroups");
  spin_kernel<<<(unsigned)workgroups, 64, 0, (hipStream_t)stream>>>((long long)microseconds * 100);
  CRESTE_CHECK_LAUNCH("spin");
  return CRESTE_OK;
}

extern "C" int creste_fill_u32(void* dst, uint32_t value, int64_t n, void* stream) {
  CRESTE_REQUIRE(dst && n >= 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0, "fill_u32: bad args");
  if (n == 0) return CRESTE_OK;
  fill_u32_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>((uint32_t*)dst, value, n);
  CRESTE_CHECK_LAUNCH("fill_u32");
  return CRESTE_OK;
}

extern "C" int creste_max2_f32(const float* a, const float* b, float* out, void* stream) {
  CRESTE_REQUIRE(a && b && out, "max2: null pointer");
  max2_f32_kernel<<<1, 1, 0, (hipStream_t)stream>>>(a, b, out);
  CRESTE_CHECK_LAUNCH("max2");
  return CRESTE_OK;
}

extern "C" int creste_maxpool_nhwc_f32(const float* in, int N, int H, int W, int C, int in_cs, float* out,
                                       int Ho, int Wo, int out_cs, int ds, float* out_amax, void* stream) {
  CRESTE_REQUIRE(in && out && C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0, "maxpool: bad args");
  CRESTE_REQUIRE(ds == 1 || ds == 2 || ds == 4, "maxpool: window/stride %d not built (1, 2, 4)", ds);
  CRESTE_REQUIRE(ds * Ho <= H && ds * Wo <= W, "maxpool: pooled extent exceeds input");
  const long total = (long)N * Ho * Wo * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  if (ds == 1) maxpool_kernel<1><<<grid_for(total), 256, 0, s>>>(in, H, W, C, in_cs, out, N, Ho, Wo, out_cs, out_amax);
  else if (ds == 2) maxpool_kernel<2><<<grid_for(total), 256, 0, s>>>(in, H, W, C, in_cs, out, N, Ho, Wo, out_cs, out_amax);
  else maxpool_kernel<4><<<grid_for(total), 256, 0, s>>>(in, H, W, C, in_cs, out, N, Ho, Wo, out_cs, out_amax);
  CRESTE_CHECK_LAUNCH("maxpool");
  return CRESTE_OK;
}

extern "C" int creste_maxpool2_nhwc_f32(const float* in, int N, int H, int W, int C, int in_cs, float* out,
                                        int Ho, int Wo, int out_cs, float* out_amax, void* stream) {
  return creste_maxpool_nhwc_f32(in, N, H, W, C, in_cs, out, Ho, Wo, out_cs, 2, out_amax, stream);
}

extern "C" int creste_nchw_to_nhwc_f32(const float* in, float* out, int out_cs, int N, int C, int H, int W,
                                       void* stream) {
  CRESTE_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0 && out_cs >= C, "nchw_to_nhwc: bad args");
  const long HW = (long)H * W;
  if (C == 4 && out_cs % 4 == 0 && ((uintptr_t)out & 15) == 0) {
    nchw4_to_nhwc_kernel<<<grid_for((long)N * HW, 256, 256 * 64), 256, 0, (hipStream_t)stream>>>(in, out, HW, out_cs, (long)N * HW);
    CRESTE_CHECK_LAUNCH("nchw4_to_nhwc");
    return CRESTE_OK;
  }
  const dim3 grid((unsigned)((HW + 31) / 32), (C + 31) / 32, N);
  nchw_to_nhwc_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, C, HW, out_cs);
  CRESTE_CHECK_LAUNCH("nchw_to_nhwc");
  return CRESTE_OK;
}

extern "C" int creste_nhwc_to_nchw_f32(const float* in, int in_cs, float* out, int N, int C, int H, int W,
                                       void* stream) {
  CRESTE_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0 && in_cs >= C, "nhwc_to_nchw: bad args");
  const long HW = (long)H * W;
  const dim3 grid((unsigned)((HW + 31) / 32), (C + 31) / 32, N);
  nhwc_to_nchw_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, in_cs, out, C, HW);
  CRESTE_CHECK_LAUNCH("nhwc_to_nchw");
  return CRESTE_OK;
}

extern "C" int creste_depth_expectation_f32(const float* logits, int cs, int64_t P, int nbins,
                                            const float* bin_values, float* depth_m, int64_t* bins,
                                            void* stream) {
  CRESTE_REQUIRE(logits && bin_values && depth_m && bins, "depth_expectation: null pointer");
  CRESTE_REQUIRE(nbins == 128 && cs % 4 == 0 && cs >= nbins, "depth_expectation: built for 128 bins");
  depth_expectation_kernel<<<grid_for(P * 32, 256, 256 * 16), 256, 0, (hipStream_t)stream>>>(
      logits, cs, P, bin_values, depth_m, bins);
  CRESTE_CHECK_LAUNCH("depth_expectation");
  return CRESTE_OK;
}

extern "C" int creste_pixel_geometry_f32(const float* depth, const float* p2p, int B, int Hs, int Ws,
                                         const float* bounds6, const float* w1, const float* b1,
                                         const float* w2, const float* b2, int zhid, int zdim,
                                         float* xyz, float* mask, float* zfeat, int z_cs, int z_co,
                                         void* stream) {
  CRESTE_REQUIRE(depth && p2p && bounds6 && w1 && b1 && w2 && b2 && xyz && mask && zfeat,
                 "pixel_geometry: null pointer");
  CRESTE_REQUIRE(zdim > 0 && zdim <= 256 && 256 % zdim == 0 && zhid > 0, "pixel_geometry: zdim must divide 256");
  CRESTE_REQUIRE((long)B * Hs * Ws < (1L << 31), "pixel_geometry: B*Hs*Ws overflows int32");
  const long total = (long)B * Hs * Ws;
  if (zhid == 64 && zdim == 32 && z_cs % 4 == 0 && z_co % 4 == 0 && (reinterpret_cast<uintptr_t>(zfeat) & 15) == 0) {
    pixel_geometry_px_kernel<false><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        depth, p2p, B, Hs, Ws, bounds6, w1, b1, w2, b2, xyz, mask, zfeat, z_cs, z_co, SplatKeyArgs{});
    CRESTE_CHECK_LAUNCH("pixel_geometry_px");
    return CRESTE_OK;
  }
  const int per = 256 / zdim;
  const size_t smem = (size_t)(2 * zhid + zdim * zhid + zdim) * sizeof(float);
  pixel_geometry_kernel<<<grid_for(total, per, 256 * 16), 256, smem, (hipStream_t)stream>>>(
      depth, p2p, B, Hs, Ws, bounds6, w1, b1, w2, b2, zhid, zdim, xyz, mask, zfeat, z_cs, z_co);
  CRESTE_CHECK_LAUNCH("pixel_geometry");
  return CRESTE_OK;
}

// creste_pixel_geometry_f32 + the key kernel of creste_bev_splat_plan_f32 in one launch: also writes bev_coords [B*P][2] and the
// points' extended base-cell keys into `splat_work` (its first B*P ints; creste_bev_splat_workspace_bytes(B, Hs*Ws, GH, GW)
// bytes), for creste_bev_splat_plan_keyed_f32.  Built for the shipped z-MLP (1 -> 64 -> 32) only: CRESTE_ERR_ARG else.
extern "C" int creste_pixel_geometry_keyed_f32(const float* depth, const float* p2p, int B, int Hs, int Ws,
                                               const float* bounds6, const float* w1, const float* b1, const float* w2,
                                               const float* b2, int zhid, int zdim, float* xyz, float* mask, float* zfeat,
                                               int z_cs, int z_co, float off_x, float off_y, float vox_x, float vox_y, int GH,
                                               int GW, float* coords, void* splat_work, void* stream) {
  CRESTE_REQUIRE(depth && p2p && bounds6 && w1 && b1 && w2 && b2 && xyz && mask && zfeat && coords && splat_work,
                 "pixel_geometry_keyed: null pointer");
  CRESTE_REQUIRE((long)B * Hs * Ws < (1L << 31), "pixel_geometry: B*Hs*Ws overflows int32");
  CRESTE_REQUIRE(GH > 0 && GW > 0 && vox_x > 0.f && vox_y > 0.f, "pixel_geometry_keyed: bad grid");
  if (!(zhid == 64 && zdim == 32 && z_cs % 4 == 0 && z_co % 4 == 0 && (reinterpret_cast<uintptr_t>(zfeat) & 15) == 0)) {
    set_error("pixel_geometry_keyed: built for the 1 -> 64 -> 32 z-MLP with 16-byte aligned output slices");
    return CRESTE_ERR_ARG;
  }
  const long total = (long)B * Hs * Ws;
  const SplatKeyArgs sk{off_x, off_y, vox_x, vox_y, GH, GW, coords, (int*)splat_work};
  pixel_geometry_px_kernel<true><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      depth, p2p, B, Hs, Ws, bounds6, w1, b1, w2, b2, xyz, mask, zfeat, z_cs, z_co, sk);
  CRESTE_CHECK_LAUNCH("pixel_geometry_px (keyed)");
  return CRESTE_OK;
}

extern "C" int creste_affine_act_nhwc_f32(const float* x, int x_cs, const float* scale, const float* shift,
                                          float* out, int out_cs, int64_t P, int C, int act, void* stream) {
  CRESTE_REQUIRE(x && scale && shift && out, "affine_act: null pointer");
  CRESTE_REQUIRE(C % 4 == 0 && x_cs % 4 == 0 && out_cs % 4 == 0 && x_cs >= C && out_cs >= C, "affine_act: C%%4");
  affine_act_kernel<<<grid_for(P * (C / 4)), 256, 0, (hipStream_t)stream>>>(x, x_cs, scale, shift, out, out_cs, P, C, act);
  CRESTE_CHECK_LAUNCH("affine_act");
  return CRESTE_OK;
}

extern "C" int creste_resize_plane_f32(const float* in, int N, int H, int W, float* out, int Ho, int Wo,
                                       int Hd, float rh, float rw, void* stream) {
  CRESTE_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Hd >= Ho, "resize_plane: bad args");
  resize_plane_kernel<<<grid_for((long)N * Ho * Wo), 256, 0, (hipStream_t)stream>>>(in, H, W, out, N, Ho, Wo, Hd, rh, rw);
  CRESTE_CHECK_LAUNCH("resize_plane");
  return CRESTE_OK;
}

extern "C" int creste_dwconv_se_nhwc_f32(const float* in, const float* w, const float* bias, float* out,
                                         float* partial, float* out_amax, int N, int H, int W, int C, int Ho,
                                         int Wo, int K, int stride, int pad_t, int pad_l, int act, void* stream) {
  CRESTE_REQUIRE(in && w && out && partial, "dwconv_se: null pointer");
  CRESTE_REQUIRE(C % 4 == 0 && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "dwconv_se: bad dims (C%%4)");
  CRESTE_REQUIRE((K == 3 || K == 5) && (stride == 1 || stride == 2),
                 "dwconv_se: kernel size %d / stride %d not built (3 or 5; 1 or 2)", K, stride);
  const int nchunk = creste_se_partial_count(Ho * Wo, C);
  const int cq = C / 4;
  // a whole number of pixel slices of cq threads, rounded up to whole waves (the spare lanes idle, but take part in the
  // wave reductions)
  const int nt = cq <= 512 ? (cq * (512 / cq) + 63) / 64 * 64 : 256;
  const int slices = nt / cq > 0 ? nt / cq : 1;
  const size_t smem = (size_t)slices * C * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(nchunk, N);
#define CRESTE_DWSE(KK, SS) \
  dwconv_se_kernel<KK, SS><<<grid, nt, smem, s>>>(in, w, bias, out, partial, out_amax, H, W, C, Ho, Wo, pad_t, pad_l, act, nchunk)
  if (K == 3 && stride == 1) CRESTE_DWSE(3, 1);
  else if (K == 3) CRESTE_DWSE(3, 2);
  else if (stride == 1) CRESTE_DWSE(5, 1);
  else CRESTE_DWSE(5, 2);
#undef CRESTE_DWSE
  CRESTE_CHECK_LAUNCH("dwconv_se");
  return CRESTE_OK;
}

extern "C" int creste_absmax_nhwc_f32(const float* x, int64_t pixels, int C, int cs, float* amax, void* stream) {
  CRESTE_REQUIRE(x && amax && pixels > 0 && C > 0, "absmax: bad args");
  CRESTE_REQUIRE(C % 4 == 0 && cs % 4 == 0 && cs >= C && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
                 "absmax: C and cs must be multiples of 4 and the slice 16-byte aligned");
  absmax_kernel<<<grid_for(pixels * (C / 4), 256, 256 * 8), 256, 0, (hipStream_t)stream>>>(x, pixels, C, cs, amax);
  CRESTE_CHECK_LAUNCH("absmax");
  return CRESTE_OK;
}
