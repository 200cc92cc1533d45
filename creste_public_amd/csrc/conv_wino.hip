// Winograd F(2x2,3x3) for the stride-1 3x3 convolutions, on the split-operand bf16 matrix cores.
//
// Why: the fp32-equivalent operand mode (bf16x6: every fp32 operand as three bf16 pieces, six piece products per
// multiply) already runs the direct 3x3 kernel at the chip's power-limited MFMA ceiling (~1.3 PF of raw bf16 MFMA),
// so the only lever left is FEWER matrix-core products per output.  F(2x2,3x3) computes a 2x2 output tile from a 4x4
// input window with 16 multiplies per (cin, cout) pair instead of 36: MFMA work / 2.25, same product grade.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A            (Lavin & Gray; correlation form, as nn.Conv2d computes)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]    G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]    A^T = [1 1 1 0; 0 1 -1 -1]
//
// Three pieces:
//   wino_pack_kernel   U = G g G^T per (cout, cin) in float64 (BatchNorm scale folded), rounded once to fp32, split
//                      into bf16 pieces and laid out as the LDS image of the GEMM's weight tiles (pack time).
//   wino_gemm_kernel   the 16 transform positions are 16 independent GEMMs  M_p[tile, cout] = sum_cin V_p[tile, cin] *
//                      U_p[cout, cin].  A workgroup owns (position p, 256 consecutive tiles, 64*TN couts).  Every row
//                      of B^T has exactly two non-zeros (+-1), so V_p[tile, cin] = +-d[i1][j1] +-d[i1][j2] +-d[i2][j1]
//                      +-d[i2][j2]: the loader forms it from FOUR raw fp32 quads of the NHWC input (no transformed
//                      tensor ever exists in HBM: it would be 4x the input, 6x with the split), splits it into the
//                      bf16 pieces and stages it as the A operand; the weight tiles arrive by LDS-DMA.  The 16
//                      position-workgroups of a tile block sit next to each other in the launch order of one XCD, so
//                      the raw input is fetched from HBM once and re-read from that XCD's L2.
//   wino_out_kernel    Y = A^T M A per (tile, channel quad) + bias + residual + activation + row mask + running |max|
//                      (the direct kernels' epilogue), reading the 16 fp32 products from the workspace.
// Accumulation is fp32 throughout; the transforms are fp32 adds of at most four terms (input) / float64 (weights).
#include "common.h"

namespace creste {

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef float wf32x16 __attribute__((ext_vector_type(16)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));

constexpr int WN_M = 256;     // tiles (GEMM rows) per workgroup
constexpr int WN_CK = 16;     // input channels per step = K of one MFMA

// rows of B^T: V[xi] = s1 * d[i1] + s2 * d[i2]
__device__ __constant__ int kWinoI1[4] = {0, 1, 1, 1};
__device__ __constant__ int kWinoI2[4] = {2, 2, 2, 3};
__device__ __constant__ float kWinoS1[4] = {1.f, 1.f, -1.f, 1.f};
__device__ __constant__ float kWinoS2[4] = {-1.f, 1.f, 1.f, -1.f};

struct WinoArgs {
  const float* in;
  const char* wpk;
  float* M;                  // [16][Cout / 4][T][4]
  int N, H, W, Cin, in_cs;
  int Cout;
  int tiles_y, tiles_x, T;   // 2x2 output tiles per image column / row, total tiles N*tiles_y*tiles_x
  int pad_t, pad_l;
  int nchunk, m_blocks, tiles_n, units;
};

// piece products of one weight fragment against the two row tiles' fragments, smallest products first; consecutive MFMAs
// go to DIFFERENT accumulators (a chain of six dependent MFMAs leaves the pipe waiting on its own result)
template <int SPLIT>
__device__ __forceinline__ void wino_split_mfma2(const wbf16x8 (&a)[SPLIT], const wbf16x8 (&b0)[SPLIT],
                                                 const wbf16x8 (&b1)[SPLIT], wf32x16& c0, wf32x16& c1) {
#pragma unroll
  for (int order = SPLIT - 1; order >= 0; --order)
#pragma unroll
    for (int pa = order; pa >= 0; --pa) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b0[order - pa], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b1[order - pa], c1, 0, 0, 0);
    }
}

// Schedule (measured on the 496 -> 496 layer at 152 x 304 x 16; each line is what the timing stamps / ablations showed):
//   * PING-PONG.  The two waves that share a SIMD (wave w and w + 4) share its matrix pipe and its VALU issue: with both
//     in the same phase the MFMAs of one queue behind the other's and then both stage at once.  The halves run half a
//     step apart -- waves 0-3: | MFMA(c) | stage their rows of A(c+1) |, waves 4-7: | stage | MFMA(c) | -- so every SIMD
//     always has one wave on the matrix pipe and one on the loader's VALU / LDS / DMA work; two barriers per chunk.
//   * ALL loop traffic is LDS-DMA, issued about two phases ahead of the barrier that publishes it (weight tile of chunk
//     c+1 at the head of even phase 2c; the raw quads of a half's next staging phase right after it has read the current
//     ones), raw s_barrier with COUNTED s_waitcnt (__syncthreads() drains the DMA queue every phase; ordinary loads
//     beside an LDS-DMA make hipcc wait vmcnt(0) at their first use).
//   * PERSISTENT workgroups, one per CU: the 256 KB of products a workgroup stores per item took ~30 us of its ~107 us
//     when it had to drain before the CU could start the next workgroup (all CUs finish together and the bursts collide in
//     HBM).  Now the stores are issued behind the NEXT item's first DMAs and drain under its main loop.
//   * B fragments of the next 32-cout tile are read while the current tile's MFMAs run (the ds_read latency was exposed
//     five times per phase: 2000-2260 cycles for 48 MFMAs instead of 1536).
// Items (tile block, position, channel tile) are dealt so that the 32 CUs of an XCD work on the 16 * tiles_n panels of
// the SAME tile block at the same time: its raw input comes from HBM once and is re-read from that XCD's L2.
template <int SPLIT, int TN>
__global__ __launch_bounds__(512, 2) void wino_gemm_kernel(const WinoArgs p) {
  constexpr int A_OCT = WN_M * 16, A_PLANE = 2 * A_OCT, A_BYTES = SPLIT * A_PLANE;       // [piece][k-octet][row][8 bf16]
  constexpr int U_OCT = 64 * 16, U_PLANE = 2 * U_OCT, U_BYTES = SPLIT * U_PLANE;         // one 64-cout weight unit
  constexpr int B_BYTES = TN * U_BYTES, B_INSTR = B_BYTES / 1024;
  constexpr int NT = TN;                       // 32-cout MFMA tiles per wave: a wave owns 64 tiles x 32*TN couts
  constexpr int kB = B_INSTR / 8;              // weight pieces EVERY wave issues per chunk (some issue one more: over-waits)
  constexpr int kStores = 2 * NT * 4;          // product stores per wave and item (always all of them: see below)
  static_assert(B_BYTES % 1024 == 0, "weight tile must be whole 1 KiB DMA pieces");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const abase = smem;                    // two A buffers, two B buffers, 8 x 8 KiB of raw quads in flight
  char* const bbase = smem + 2 * A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;                  // waves w and w + 4 share a SIMD
  const int wm = wave & 3, wn = half;          // 4 row groups of 64 tiles x 2 channel halves
  const int li = lane & 31, lh = lane >> 5;
  const int u = tid & 255, cq = u & 3;
  const int a_lofs0 = (cq >> 1) * A_OCT + (half * 128 + (u >> 2)) * 16 + (cq & 1) * 8;     // + j * 64 * 16
  char* const rawbase = smem + 2 * A_BYTES + 2 * B_BYTES + wave * 8192;
  const int Q = p.Cout >> 2;
  const int per = p.tiles_y * p.tiles_x;

  // ---- item schedule: workgroup b sits on XCD b % 8 (round-robin dispatch); slot s of that XCD is panel s % P of its
  // tile block s / P, and the XCD's CUs take slots j, j + cus, ...   (placement is speed only)
  const int cus = gridDim.x >> 3;              // workgroups per XCD
  const int xcd = blockIdx.x & 7, P = 16 * p.tiles_n;
  int slot = blockIdx.x >> 3;

  // per-item state
  int mb, pos, tn;
  int off[2][4];             // element offsets of the four raw pixels (y1x1, y1x2, y2x1, y2x2), clamped into the tensor
  float mx1[2], mx2[2], my1[2], my2[2];       // +-1, or 0 for an absent pixel / row
  const char* wbase;
  auto setup = [&](int s) __attribute__((always_inline)) -> bool {
    const int mbl = s / P, pnl = s - mbl * P;
    mb = mbl * 8 + xcd;
    if (mb >= p.m_blocks) return false;
    pos = pnl / p.tiles_n; tn = pnl - pos * p.tiles_n;
    const int xi = pos >> 2, nu = pos & 3;
    const int iy1 = kWinoI1[xi], iy2 = kWinoI2[xi], ix1 = kWinoI1[nu], ix2 = kWinoI2[nu];
    // half h stages rows h*128 + u/4 + 64 j (u = thread within the half), channel quad cq.  Absent pixels (image border,
    // rows past the last tile) read a CLAMPED address and enter with multiplier 0: no predicated loads, no selects
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m0 = mb * WN_M + half * 128 + j * 64 + (u >> 2);
      const bool ok = m0 < p.T;
      const int m = ok ? m0 : p.T - 1;
      const int img = m / per, rem = m - img * per;
      const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
      const int y0 = 2 * ty - p.pad_t, x0 = 2 * tx - p.pad_l;
      const int ys[2] = {y0 + iy1, y0 + iy2}, xs[2] = {x0 + ix1, x0 + ix2};
      const bool yok[2] = {(unsigned)ys[0] < (unsigned)p.H, (unsigned)ys[1] < (unsigned)p.H};
      const bool xok[2] = {(unsigned)xs[0] < (unsigned)p.W, (unsigned)xs[1] < (unsigned)p.W};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int yc = min(max(ys[a], 0), p.H - 1), xc = min(max(xs[b], 0), p.W - 1);
          off[j][a * 2 + b] = ((img * p.H + yc) * p.W + xc) * p.in_cs;
        }
      my1[j] = ok && yok[0] ? kWinoS1[xi] : 0.f; my2[j] = ok && yok[1] ? kWinoS2[xi] : 0.f;
      mx1[j] = xok[0] ? kWinoS1[nu] : 0.f; mx2[j] = xok[1] ? kWinoS2[nu] : 0.f;
    }
    wbase = p.wpk + ((size_t)pos * p.units + (size_t)tn * TN) * p.nchunk * U_BYTES;
    return true;
  };

  // Raw quads travel global -> LDS by DMA: a wave owns 8 KiB of the raw area, piece (j, k) = its 64 lanes' 16 bytes of
  // pixel k of row group j, and every lane later reads back exactly the 16 bytes it fetched.
  auto dma_raw = [&](int c) __attribute__((always_inline)) {
    const int ch0 = c * WN_CK + cq * 4, ch = ch0 < p.Cin ? ch0 : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(p.in) + (unsigned)((off[j][k] + ch) << 2)),
                                         (__attribute__((address_space(3))) void*)(rawbase + (j * 4 + k) * 1024), 16, 0, 0);
  };
  auto read_raw = [&](wf32x4 (&d)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) d[j][k] = *reinterpret_cast<const wf32x4*>(rawbase + (j * 4 + k) * 1024 + lane * 16);
  };
  // transform + split + store of a staged chunk, with the NEXT chunk's raw DMA pieces (if `cn` >= 0) issued between the
  // elements: an LDS-DMA costs the wave ~170 cycles to issue next to its partner's MFMAs (timing stamps: the 8 pieces
  // were 1300-1500 cycles of a 2700-cycle staging phase when issued back to back in front of the transform); a piece
  // every ~12 VALU instructions lets the queue drain behind arithmetic instead of in front of it
  auto store_a = [&](const wf32x4 (&d)[2][4], int c, char* buf, int cn) __attribute__((always_inline)) {
    const bool chok = c * WN_CK + cq * 4 < p.Cin;
    const int chn0 = cn * WN_CK + cq * 4, chn = chn0 < p.Cin ? chn0 : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float a1 = chok ? my1[j] : 0.f, a2 = chok ? my2[j] : 0.f;
      wf32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (cn >= 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(p.in) + (unsigned)((off[j][e] + chn) << 2)),
                                           (__attribute__((address_space(3))) void*)(rawbase + (j * 4 + e) * 1024), 16, 0, 0);
        // multiplications by +-1 / 0 are exact: each fma is ONE rounding of a two-term sum
        const float r1 = __fmaf_rn(d[j][1][e], mx2[j], d[j][0][e] * mx1[j]);
        const float r2 = __fmaf_rn(d[j][3][e], mx2[j], d[j][2][e] * mx1[j]);
        v[e] = __fmaf_rn(r2, a2, r1 * a1);
        __builtin_amdgcn_sched_barrier(0);
      }
      char* dst = buf + a_lofs0 + j * (64 * 16);
      wf32x4 rem = v;
#pragma unroll
      for (int pl = 0; pl < SPLIT; ++pl) {           // hi, then the bf16 of what is left, ...
        const wbf16x4 piece = __builtin_convertvector(rem, wbf16x4);
        *reinterpret_cast<wbf16x4*>(dst + pl * A_PLANE) = piece;
        if (pl + 1 < SPLIT) rem -= __builtin_convertvector(piece, wf32x4);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // weight tile of chunk c: per 64-cout unit U_BYTES contiguous bytes, copied by LDS-DMA in 1 KiB pieces (every wave its
  // pieces wave, wave + 8, ...)
  auto dma_b = [&](int c) __attribute__((always_inline)) {
    char* dst = bbase + (c & 1) * B_BYTES;
#pragma unroll
    for (int jj = 0; jj < (B_INSTR + 7) / 8; ++jj) {
      const int i = wave + 8 * jj;
      if (i < B_INSTR) {
        const int uu = i / (U_BYTES / 1024), r = i % (U_BYTES / 1024);
        // wave-uniform base + 32-bit lane offset: the scalar-base form of the instruction issues measurably faster than a
        // 64-bit address per lane (496->496 layer 11.0 -> 10.5 ms with the raw pieces alone)
        const size_t srcv = reinterpret_cast<size_t>(wbase + ((size_t)uu * p.nchunk + c) * U_BYTES + r * 1024);
        const char* src = reinterpret_cast<const char*>(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(srcv >> 32)) << 32) |
                                                        (unsigned)__builtin_amdgcn_readfirstlane((int)srcv));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(lane * 16)),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      }
    }
  };

  wf32x16 acc[2][NT];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  // weights as the first MFMA operand (D = U * V^T: a lane owns one tile, its registers the couts)
  auto mfma_chunk = [&](int c) __attribute__((always_inline)) {
    const char* A = abase + (c & 1) * A_BYTES;
    const char* B = bbase + (c & 1) * B_BYTES;
    wbf16x8 af[2][SPLIT], bfr[2][SPLIT];
    auto read_b = [&](int nt, wbf16x8 (&dst)[SPLIT]) __attribute__((always_inline)) {
      const int n = (wn * NT + nt) * 32 + li;
#pragma unroll
      for (int pl = 0; pl < SPLIT; ++pl)
        dst[pl] = *reinterpret_cast<const wbf16x8*>(B + (n >> 6) * U_BYTES + pl * U_PLANE + lh * U_OCT + (n & 63) * 16);
    };
    read_b(0, bfr[0]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = wm * 64 + mt * 32 + li;
#pragma unroll
      for (int pl = 0; pl < SPLIT; ++pl)
        af[mt][pl] = *reinterpret_cast<const wbf16x8*>(A + pl * A_PLANE + lh * A_OCT + row * 16);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt + 1 < NT) read_b(nt + 1, bfr[(nt + 1) & 1]);     // lands behind this tile's 4 * SPLIT MFMAs
      wino_split_mfma2<SPLIT>(bfr[nt & 1], af[0], af[1], acc[0][nt], acc[1][nt]);
    }
  };

  if (!setup(slot)) return;
  zero_acc();
  dma_b(0);
  dma_raw(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  wf32x4 ra[2][4];
#ifdef WINO_TRACE
  long long t_head = 0, t_loop = 0, t_tail = 0, t_work = 0, t_mf = 0, t_st = 0, w_mf = 0, w_st = 0, w_raw = 0, t0 = __builtin_readcyclecounter(), tw, tx; int n_items = 0;
#endif
  for (;;) {
#ifdef WINO_TRACE
    long long ta = __builtin_readcyclecounter();
#endif
    // ---- item head: chunk 0's weight tile and raw quads have landed (counted waits at the end of the previous item)
    asm volatile("s_barrier" ::: "memory");
    read_raw(ra);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    store_a(ra, 0, abase, p.nchunk > 1 ? 1 : -1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

#ifdef WINO_TRACE
    long long tb = __builtin_readcyclecounter();
#endif
    // phase q: half h computes chunk (q - h) / 2 when q + h is even, else stages the chunk the other half computes next
    for (int q = 0; q < 2 * p.nchunk; ++q) {
#ifdef WINO_TRACE
      tw = __builtin_readcyclecounter();
#endif
      const int cw = (q >> 1) + 1;             // weight tile fetched in even phases
      if (!(q & 1) && cw < p.nchunk) dma_b(cw);
      if (((q + half) & 1) == 0) {
        mfma_chunk((q - half) >> 1);
      } else {
        const int cst = half == 0 ? (q + 1) >> 1 : (q >> 1) + 1;
        if (cst < p.nchunk) {
          // raw quads of chunk cst (issued two phases ago): the only younger operations are kB weight pieces
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kB) : "memory");
#ifdef WINO_TRACE
          w_raw += __builtin_readcyclecounter() - tw;
#endif
          read_raw(ra);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          store_a(ra, cst, abase + (cst & 1) * A_BYTES, cst + 1 < p.nchunk ? cst + 1 : -1);
        }
      }
#ifdef WINO_TRACE
      tx = __builtin_readcyclecounter();
      t_work += tx - tw;
      if (((q + half) & 1) == 0) t_mf += tx - tw; else t_st += tx - tw;
#endif
      // end of an odd phase 2c+1: chunk c+1's weight tile (issued at the head of phase 2c) must have landed; the only
      // younger operations of ANY wave are the 8 raw pieces of chunk c+2 it re-issued while staging (this phase or the last)
      if (q & 1) {
        if ((q >> 1) + 2 < p.nchunk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
#ifdef WINO_TRACE
      if (((q + half) & 1) == 0) w_mf += __builtin_readcyclecounter() - tx; else w_st += __builtin_readcyclecounter() - tx;
#endif
    }

#ifdef WINO_TRACE
    long long tc = __builtin_readcyclecounter();
#endif
    // ---- item tail.  Every LDS buffer is free (last barrier passed).  Products go to M[pos][cout / 4][tile][4]: lane =
    // tile, registers 4g..4g+3 = four consecutive couts, so the 32 lanes of a half-wave write 512 contiguous bytes per
    // instruction.  ALL kStores stores are issued (rows past the last tile / couts past Cout go to a junk line behind
    // the workspace): the count lets the next item's first DMAs be waited for with vmcnt(kStores) while the stores drain
    float* Mp = p.M + (size_t)pos * Q * p.T * 4;
    float* const junk = p.M + (size_t)16 * Q * p.T * 4 + lane * 4;
    const int mb_cur = mb, tn_cur = tn;
    slot += cus;
    const bool more = setup(slot);
    if (more) { dma_b(0); dma_raw(0); }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = mb_cur * WN_M + wm * 64 + mt * 32 + li;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = tn_cur * (64 * TN) + (wn * NT + nt) * 32 + 8 * g + 4 * lh;
          float* dst = (m < p.T && n < p.Cout) ? Mp + ((size_t)(n >> 2) * p.T + m) * 4 : junk;
          *reinterpret_cast<wf32x4*>(dst) =
              wf32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
        }
    }
#ifdef WINO_TRACE
    { long long td = __builtin_readcyclecounter(); t_head += tb - ta; t_loop += tc - tb; t_tail += td - tc; ++n_items; }
#endif
    if (!more) break;
    zero_acc();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStores) : "memory");      // the next item's chunk-0 DMAs have landed
  }
#ifdef WINO_TRACE
  if (blockIdx.x == 100 && (wave & 3) == 0 && lane == 0)
    printf("wave %d items %d total %lld head %lld loop %lld (work %lld: mfma %lld stage %lld; wait after mfma %lld after stage %lld) tail %lld raw-wait %lld per item\n", wave, n_items,
           (long long)(__builtin_readcyclecounter() - t0) / n_items, t_head / n_items, t_loop / n_items, t_work / n_items, t_mf / n_items, t_st / n_items, w_mf / n_items, w_st / n_items, t_tail / n_items, w_raw / n_items);
#endif
}

struct WinoOutArgs {
  const float* M;
  const float* bias;
  const float* res;
  const float* row_mask;
  float* out;
  float* out_amax;
  int N, Ho, Wo, Cout, out_cs, out_co, res_cs, act;
  int tiles_y, tiles_x, T;
};

// One workgroup = 16 consecutive tiles x 64 couts.  Read side: thread = (tile, channel quad), 16 x 16-byte loads (one per
// transform position; the 16 tiles of a quad are 256 contiguous bytes of M), 24 vector adds.  The 2x2 outputs then
// cross an LDS tile so that the write side runs thread = (pixel, channel quad): 16 lanes write 256 contiguous bytes
// of one NHWC pixel and read bias / residual the same way (the direct kernels' epilogue).
constexpr int WO_TILES = 16, WO_QUADS = 16, WO_ROW = WO_QUADS * 4 + 4;       // LDS row: 64 couts + 16 bytes of padding
__global__ __launch_bounds__(256) void wino_out_kernel(const WinoOutArgs p) {
  __shared__ __attribute__((aligned(16))) float tilebuf[WO_TILES * 4 * WO_ROW];
  __shared__ float scratch[4];
  const int Q = p.Cout >> 2;
  const int t = threadIdx.x;
  const int tile0 = blockIdx.x * WO_TILES, quad0 = blockIdx.y * WO_QUADS;
  {
    const int tl = t & (WO_TILES - 1), ql = t >> 4;
    const int tile = tile0 + tl, quad = quad0 + ql;
    if (tile < p.T && quad < Q) {
      wf32x4 m[16];
      const float* src = p.M + ((size_t)quad * p.T + tile) * 4;
      const size_t plane = (size_t)Q * p.T * 4;
#pragma unroll
      for (int k = 0; k < 16; ++k) m[k] = __builtin_nontemporal_load(reinterpret_cast<const wf32x4*>(src + k * plane));
      wf32x4 tt[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tt[r][0] = (m[4 * r] + m[4 * r + 1]) + m[4 * r + 2];
        tt[r][1] = (m[4 * r + 1] - m[4 * r + 2]) - m[4 * r + 3];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        *reinterpret_cast<wf32x4*>(tilebuf + (tl * 4 + j) * WO_ROW + ql * 4) = (tt[0][j] + tt[1][j]) + tt[2][j];
        *reinterpret_cast<wf32x4*>(tilebuf + (tl * 4 + 2 + j) * WO_ROW + ql * 4) = (tt[1][j] - tt[2][j]) - tt[3][j];
      }
    }
  }
  __syncthreads();
  float vmax = 0.f;
  const int cq = t & 15, n = (quad0 + cq) * 4;
  const int per = p.tiles_y * p.tiles_x;
  if (quad0 + cq < Q) {
    const wf32x4 bs = p.bias ? *reinterpret_cast<const wf32x4*>(p.bias + n) : wf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int pl = pass * 16 + (t >> 4);          // pixel slot: tile pl / 4, (i, j) = pl % 4
      const int tile = tile0 + (pl >> 2);
      if (tile >= p.T) continue;
      const int img = tile / per, rem = tile - img * per;
      const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
      const int oy = 2 * ty + ((pl >> 1) & 1), ox = 2 * tx + (pl & 1);
      if (oy >= p.Ho || ox >= p.Wo) continue;
      const long mrow = ((long)img * p.Ho + oy) * p.Wo + ox;
      wf32x4 v = *reinterpret_cast<const wf32x4*>(tilebuf + pl * WO_ROW + cq * 4) + bs;
      if (p.res) v += *reinterpret_cast<const wf32x4*>(p.res + mrow * p.res_cs + n);
      const float rmask = p.row_mask ? p.row_mask[mrow] : 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = act_apply(v[e], p.act) * rmask;
        vmax = fmaxf(vmax, fabsf(v[e]));
      }
      *reinterpret_cast<wf32x4*>(p.out + mrow * p.out_cs + p.out_co + n) = v;
    }
  }
  if (p.out_amax) block_amax_update(vmax, p.out_amax, scratch);
}

// U = G g G^T in float64 from the OIHW fp32 weights (x BatchNorm scale, applied in fp32 as the direct packers do),
// rounded once to fp32, split into bf16 pieces: [pos][unit][chunk][piece][k-octet][64][8]
__global__ void wino_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, __bf16* __restrict__ out,
                                 int Cout, int Cin, int units, int nchunk, int split) {
  const long total = (long)units * 64 * nchunk * WN_CK;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % (nchunk * WN_CK)), co = (int)(i / (nchunk * WN_CK));
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float v = 0.f;
        if (co < Cout && ci < Cin) {
          v = w[(((long)co * Cin + ci) * 3 + a) * 3 + b];
          if (scale) v *= scale[co];
        }
        g[a][b] = (double)v;
      }
    double Gg[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      Gg[0][b] = g[0][b];
      Gg[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
      Gg[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
      Gg[3][b] = g[2][b];
    }
    const int unit = co >> 6, nn = co & 63, c = ci / WN_CK, oct = (ci % WN_CK) >> 3, e = ci & 7;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double u4[4] = {Gg[a][0], 0.5 * (Gg[a][0] + Gg[a][1] + Gg[a][2]), 0.5 * (Gg[a][0] - Gg[a][1] + Gg[a][2]), Gg[a][2]};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int pos = a * 4 + b;
        float v = (float)u4[b];
        __bf16* dst = out + ((((size_t)pos * units + unit) * nchunk + c) * split) * (2 * 64 * 8) + (size_t)oct * 64 * 8 + nn * 8 + e;
        for (int pl = 0; pl < split; ++pl) {
          const __bf16 piece = (__bf16)v;
          dst[(size_t)pl * (2 * 64 * 8)] = piece;
          v -= (float)piece;
        }
      }
    }
  }
}

static inline int wino_split(int prec) {
  return prec == CRESTE_PREC_BF16X6 ? 3 : (prec == CRESTE_PREC_BF16X3 ? 2 : 0);     // the fp32-grade split modes
}
static inline int wino_units(int Cout) { return ((Cout + 63) / 64 + 3) / 4 * 4; }       // padded to the widest tile (TN = 4)

bool conv_wino_supported(int prec, int KH, int KW, int stride, int Cin, int Cout) {
  return wino_split(prec) > 0 && KH == 3 && KW == 3 && stride == 1 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0;
}

int64_t conv_wino_weight_bytes(int Cout, int Cin, int prec) {
  const long nchunk = (Cin + WN_CK - 1) / WN_CK;
  return 16L * wino_units(Cout) * nchunk * wino_split(prec) * 2 * 64 * 16;
}

int conv_wino_pack(const float* w, const float* scale, void* wpk, int Cout, int Cin, int prec, hipStream_t s) {
  const int units = wino_units(Cout), nchunk = (Cin + WN_CK - 1) / WN_CK;
  const long total = (long)units * 64 * nchunk * WN_CK;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  wino_pack_kernel<<<blocks, 256, 0, s>>>(w, scale, (__bf16*)wpk, Cout, Cin, units, nchunk, wino_split(prec));
  CRESTE_CHECK_LAUNCH("wino_pack");
  return CRESTE_OK;
}

int64_t conv_wino_workspace_bytes(int N, int Ho, int Wo, int Cout) {
  return 16L * N * ((Ho + 1) / 2) * ((Wo + 1) / 2) * Cout * 4 + 4096;       // + the junk line of the GEMM's padded stores
}

template <int SPLIT, int TN>
static int launch_wino_gemm(const WinoArgs& a, hipStream_t s) {
  constexpr int smem = 2 * (SPLIT * 2 * WN_M * 16) + 2 * (TN * SPLIT * 2 * 64 * 16) + 8 * 8192;
  static_assert(smem <= 160 * 1024, "Winograd GEMM tile does not fit the LDS");
  static std::atomic<uint64_t> attr_devs{0};
  CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wino_gemm_kernel<SPLIT, TN>), smem, attr_devs));
  // persistent: one workgroup per CU (a multiple of the 8 XCDs), never more than there are items
  int dev = 0, cus = 0;
  CRESTE_HIP(hipGetDevice(&dev));
  CRESTE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const long items = (long)a.m_blocks * 16 * a.tiles_n;
  long per_xcd = cus / 8 > 0 ? cus / 8 : 1;
  const long need = (items + 7) / 8;
  if (per_xcd > need) per_xcd = need;
  wino_gemm_kernel<SPLIT, TN><<<(unsigned)(per_xcd * 8), 512, smem, s>>>(a);
  CRESTE_CHECK_LAUNCH("wino_gemm");
  return CRESTE_OK;
}

int conv_wino_run(const creste_conv_desc* d, hipStream_t s) {
  CRESTE_REQUIRE(conv_wino_supported(d->prec, d->KH, d->KW, d->stride, d->Cin, d->Cout),
                 "conv2d: the Winograd path is built for stride-1 3x3 convs in the bf16 split modes, Cin and Cout multiples of 4");
  CRESTE_REQUIRE(d->work && !d->a_scale, "conv2d: the Winograd path needs its workspace and takes no per-sample input gate");
  CRESTE_REQUIRE((d->out_cs & 3) == 0 && (d->out_co & 3) == 0 && (!d->res || (d->res_cs & 3) == 0) &&
                     (reinterpret_cast<uintptr_t>(d->out) & 15) == 0,
                 "conv2d: the Winograd path needs 16-byte aligned output / residual channel slices");
  CRESTE_REQUIRE((long)d->N * d->H * d->W * d->in_cs < (1L << 30),
                 "conv2d: the Winograd loader addresses the input with 32-bit byte offsets (< 4 GiB per call: split the batch)");
  WinoArgs a;
  a.in = d->in; a.wpk = (const char*)d->wpk; a.M = (float*)d->work;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.in_cs = d->in_cs; a.Cout = d->Cout;
  a.tiles_y = (d->Ho + 1) / 2; a.tiles_x = (d->Wo + 1) / 2;
  const long T = (long)d->N * a.tiles_y * a.tiles_x;
  CRESTE_REQUIRE(T * 16 * d->Cout < (1L << 40) && T < (1L << 27), "conv2d: Winograd workspace too large");
  a.T = (int)T;
  a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.nchunk = (d->Cin + WN_CK - 1) / WN_CK;
  a.m_blocks = (int)((T + WN_M - 1) / WN_M);
  a.units = wino_units(d->Cout);
  // 256-cout tiles where the layer has them, else 128 (a 64-wide tile would amortise the loader's transform over too few products)
  const int tn = d->Cout > 128 ? 4 : 2;
  a.tiles_n = (d->Cout + 64 * tn - 1) / (64 * tn);
  const int split = wino_split(d->prec);
  int rc;
  if (tn == 4) rc = split == 3 ? launch_wino_gemm<3, 4>(a, s) : launch_wino_gemm<2, 4>(a, s);
  else rc = split == 3 ? launch_wino_gemm<3, 2>(a, s) : launch_wino_gemm<2, 2>(a, s);
  if (rc != CRESTE_OK) return rc;
  WinoOutArgs o;
  o.M = a.M; o.bias = d->bias; o.res = d->res; o.row_mask = d->row_mask; o.out = d->out; o.out_amax = d->out_amax;
  o.N = d->N; o.Ho = d->Ho; o.Wo = d->Wo; o.Cout = d->Cout; o.out_cs = d->out_cs; o.out_co = d->out_co; o.res_cs = d->res_cs;
  o.act = d->act; o.tiles_y = a.tiles_y; o.tiles_x = a.tiles_x; o.T = a.T;
  const dim3 ogrid((unsigned)((T + WO_TILES - 1) / WO_TILES), (unsigned)((d->Cout / 4 + WO_QUADS - 1) / WO_QUADS));
  wino_out_kernel<<<ogrid, 256, 0, s>>>(o);
  CRESTE_CHECK_LAUNCH("wino_out");
  return CRESTE_OK;
}

}  // namespace creste
