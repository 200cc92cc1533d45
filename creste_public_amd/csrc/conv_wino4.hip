// Winograd F(4x4,3x3) for the wide stride-1 3x3 convolutions, on the split-operand bf16 matrix cores.
//
// Why a second Winograd: F(2x2,3x3) (conv_wino.hip) forms its transformed input inside the GEMM's loader, and the timing
// stamps in profiles/r03_winograd_notes.md show that loader -- not the matrix pipe -- bounding the kernel (an LDS-DMA
// costs a wave ~170 cycles to issue next to its partner's MFMAs; 41.7 % MFMA-busy).  F(4x4,3x3) needs 36 multiplies per
// 4x4 output tile and (cin, cout) pair instead of 144 direct / 64 for F(2x2): 1.78x less matrix work again, and with
// the transformed input materialised ONCE in HBM, already split into bf16 pieces and laid out as the GEMM's LDS image,
// the GEMM loop is nothing but LDS-DMA + MFMA.  The price is HBM traffic (transformed input = 2.25 x 1.5 of the fp32
// input, products = 2.25 x the fp32 output, each written and read once) in two bandwidth-bound kernels beside the GEMM.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray, interpolation points 0, +-1, +-2, inf)
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//
// Numerics: transforms in fp32 (weights: float64, rounded once), products fp32-grade (bf16x6) with fp32 accumulation.
// Relative rms error against float64 on the 496-channel layers: 1.3e-6 (direct fp32 accumulation: 1.8e-7, F(2x2): 1.1e-6).
//
//   wino4_pack_kernel  U = G g G^T per (cout, cin), 36 positions, the GEMM's weight-tile LDS image (pack time)
//   wino4_in1_kernel   V = B^T d B per (tile, cin): 6x6 window at stride 4, written as fp32
//                      V[pos][tile block][chunk][k-quad][256 tiles][4]  (= the GEMM's A-operand LDS image; round 3 wrote it
//                      as bf16 pieces: wino4_in_kernel, kept as the bit-exact reference and for the weight gradient)
//   wino4_gemm32_kernel 36 independent GEMMs M_p[tile, cout] = sum_cin V_p[tile, cin] U_p[cout, cin]; persistent
//                      workgroups, three LDS stages, every byte by LDS-DMA, one barrier per 16-channel chunk, A split into
//                      its bf16 pieces in registers
//   wino4_out2_kernel  Y = A^T M A per (tile, channel pair) + the direct kernels' epilogue
// Both transforms are sized to fit BESIDE a GEMM workgroup of another stream on the same CU (<= 112 VGPRs, <= 40 KiB LDS)
#include "common.h"
#include <stdlib.h>

namespace creste {

typedef __bf16 w4bf16x8 __attribute__((ext_vector_type(8)));
typedef float w4f32x16 __attribute__((ext_vector_type(16)));
typedef float w4f32x4 __attribute__((ext_vector_type(4)));

constexpr int W4_M = 256;      // tiles (GEMM rows) per tile block
constexpr int W4_CK = 16;      // input channels per chunk = K of one MFMA
constexpr int W4_POS = 36;
constexpr int W4_MBG = 4;      // tile blocks an XCD works on at the same time (weight panels re-read from its L2)

template <int SPLIT>
__device__ __forceinline__ void w4_split_mfma2(const w4bf16x8 (&a)[SPLIT], const w4bf16x8 (&b0)[SPLIT],
                                               const w4bf16x8 (&b1)[SPLIT], w4f32x16& c0, w4f32x16& c1) {
  // smallest products first; consecutive MFMAs go to different accumulators
#pragma unroll
  for (int order = SPLIT - 1; order >= 0; --order)
#pragma unroll
    for (int pa = order; pa >= 0; --pa) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b0[order - pa], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b1[order - pa], c1, 0, 0, 0);
    }
}

// the same for ONE row fragment and two weight tiles (the row-split wave layout: a wave owns 32 tiles x all couts)
template <int SPLIT>
__device__ __forceinline__ void w4_split_mfma2w(const w4bf16x8 (&w0)[SPLIT], const w4bf16x8 (&w1)[SPLIT],
                                                const w4bf16x8 (&v)[SPLIT], w4f32x16& c0, w4f32x16& c1) {
#pragma unroll
  for (int order = SPLIT - 1; order >= 0; --order)
#pragma unroll
    for (int pa = order; pa >= 0; --pa) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[pa], v[order - pa], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[pa], v[order - pa], c1, 0, 0, 0);
    }
}

struct Wino4GemmArgs {
  const char* V;             // [36][m_blocks][nchunk][SPLIT][2][256][8] bf16
  const char* wpk;           // [36][units][nchunk][SPLIT][2][64][8] bf16
  float* M;                  // [36][Cout / 4][T][4]
  int T, Cout;
  int nchunk, m_blocks, tiles_n, units;
  int npos;                  // 36 (forward) or 36 x K-segments (weight gradient)
  long mplane;               // floats between the product planes of two positions (>= Cout * T)
  int mbg;                   // tile blocks per unit: min(W4_MBG, m_blocks) -- see the item comment below
};

// Item = (tile block, position, cout tile).  Items are dealt in UNITS of 8 panels (position, cout tile) x mbg tile blocks
// (mbg = W4_MBG = 4 for every map with at least four tile blocks); unit u goes to XCD u % 8 (workgroup b sits on XCD b % 8) and
// its items to that XCD's workgroups, so a weight panel streams through the XCD's L2 once for four tile blocks and a V tile once
// for its cout tiles.  Placement is speed only.
// Round 6: with FEWER than four tile blocks -- the weight gradient's GEMMs, whose rows are the <= 496 input CHANNELS = two blocks,
// and batch-1 maps -- units of 32 slots left (4 - m_blocks) / 4 of the slots as holes, and because a workgroup's slot advances by a
// multiple of 32 it met the SAME hole every time: with two blocks half of the chip's workgroups never received an item (the
// weight-gradient GEMMs ran at half speed: 1.93 ms per call against 0.95 for the forward's same products).  The unit now has
// 8 x min(4, m_blocks) slots, none of them a hole.
//
// AF32: the A operand (transformed input) arrives as fp32, [k-quad][256 rows][4 floats] per chunk (2/3 of the bytes of the
// three bf16 pieces, in HBM and through the LDS-DMA), and every wave splits its own 64 rows into the bf16 pieces in
// registers right after reading them -- the conversions / subtractions of wino4_in_kernel's piece loop, so the pieces
// and therefore the products are bit-identical to the pre-split form (which the weight gradient keeps using).
typedef float w4f32x8 __attribute__((ext_vector_type(8)));
template <int SPLIT, int TN, bool AF32>
__global__ __launch_bounds__(512, 2) void wino4_gemm_kernel(const Wino4GemmArgs p) {
  constexpr int A_OCT = W4_M * 16, A_PLANE = 2 * A_OCT;                                  // [piece][k-octet][row][8 bf16]
  constexpr int A_BYTES = AF32 ? 4 * W4_M * 16 : SPLIT * A_PLANE;                        // AF32: [k-quad][row][4 f32]
  constexpr int U_OCT = 64 * 16, U_PLANE = 2 * U_OCT, U_BYTES = SPLIT * U_PLANE;         // one 64-cout weight unit
  constexpr int B_BYTES = TN * U_BYTES;
  constexpr int STAGE = A_BYTES + B_BYTES, NS = 3;
  constexpr int A_INSTR = A_BYTES / 1024, B_INSTR = B_BYTES / 1024;
  constexpr int kA = A_INSTR / 8, kBw = (B_INSTR + 7) / 8;     // 1 KiB DMA pieces EVERY wave issues per chunk
  constexpr int kDma = kA + kBw;
  constexpr int NT = TN;                       // a wave owns 64 tiles x 32*TN couts
  constexpr int kStores = 2 * NT * 4;
  static_assert(A_INSTR % 8 == 0 && B_BYTES % 1024 == 0, "tiles must be whole DMA pieces");
  static_assert(kDma + kStores <= 63, "vmcnt is a 6-bit counter");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;     // 4 row groups of 64 tiles x 2 channel halves
  const int li = lane & 31, lh = lane >> 5;

  const int cus = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, P = p.npos * p.tiles_n;
  const int mbg = p.mbg, us = 8 * mbg;
  const int units_pg = (P + 7) >> 3, nunits = units_pg * ((p.m_blocks + mbg - 1) / mbg);
  int slot = blockIdx.x >> 3;

  int mb, pos, tn;
  const char *abase_g, *wbase;
  // -> true when `slot` names an item (advancing over the holes of ragged units), false at the end
  auto setup = [&]() __attribute__((always_inline)) -> bool {
    for (;; slot += cus) {
      const int su = slot / us, unit = su * 8 + xcd, w = slot - su * us;
      if (unit >= nunits) return false;
      const int mg = unit / units_pg, pg = unit - mg * units_pg;   // panel groups fastest: neighbouring XCDs share V tiles in the MALL
      const int pnl = pg * 8 + (w & 7);
      mb = mg * mbg + (w >> 3);
      if (pnl >= P || mb >= p.m_blocks) continue;
      pos = pnl / p.tiles_n; tn = pnl - pos * p.tiles_n;
      abase_g = p.V + ((size_t)pos * p.m_blocks + mb) * p.nchunk * A_BYTES;
      wbase = p.wpk + ((size_t)pos * p.units + (size_t)tn * TN) * p.nchunk * U_BYTES;
      return true;
    }
  };
  auto uniform = [](const char* q) __attribute__((always_inline)) -> const char* {
    const size_t v = reinterpret_cast<size_t>(q);
    return reinterpret_cast<const char*>(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)v));
  };
  // chunk c -> stage c % 3: A tile (contiguous A_BYTES of V) and weight tile (U_BYTES per 64-cout unit), 1 KiB pieces,
  // scalar base + 32-bit lane offset.  Every wave issues exactly kDma pieces (a wave with no weight piece left re-copies
  // its previous one: same bytes to the same place), so the counted waits below hold for all of them
  auto dma = [&](int c) __attribute__((always_inline)) {
    char* st = smem + (c % NS) * STAGE;
#pragma unroll
    for (int jj = 0; jj < kA; ++jj) {
      const int i = wave + 8 * jj;
      const char* src = uniform(abase_g + (size_t)c * A_BYTES + i * 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(lane * 16)),
                                       (__attribute__((address_space(3))) void*)(st + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < kBw; ++jj) {
      int i = wave + 8 * jj;
      if (i >= B_INSTR) i -= 8;
      const int uu = i / (U_BYTES / 1024), r = i % (U_BYTES / 1024);
      const char* src = uniform(wbase + ((size_t)uu * p.nchunk + c) * U_BYTES + r * 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(lane * 16)),
                                       (__attribute__((address_space(3))) void*)(st + A_BYTES + i * 1024), 16, 0, 0);
    }
  };

  w4f32x16 acc[2][NT];
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
    const char* A = smem + (c % NS) * STAGE;
    const char* B = A + A_BYTES;
    w4bf16x8 af[2][SPLIT], bfr[2][SPLIT];
    auto read_b = [&](int nt, w4bf16x8 (&dst)[SPLIT]) __attribute__((always_inline)) {
      const int n = (wn * NT + nt) * 32 + li;
#pragma unroll
      for (int pl = 0; pl < SPLIT; ++pl)
        dst[pl] = *reinterpret_cast<const w4bf16x8*>(B + (n >> 6) * U_BYTES + pl * U_PLANE + lh * U_OCT + (n & 63) * 16);
    };
    read_b(0, bfr[0]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = wm * 64 + mt * 32 + li;
      if constexpr (AF32) {
        const w4f32x4 q0 = *reinterpret_cast<const w4f32x4*>(A + (2 * lh) * A_OCT + row * 16);
        const w4f32x4 q1 = *reinterpret_cast<const w4f32x4*>(A + (2 * lh + 1) * A_OCT + row * 16);
        w4f32x8 x{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
        for (int pl = 0; pl < SPLIT; ++pl) {
          af[mt][pl] = __builtin_convertvector(x, w4bf16x8);
          if (pl + 1 < SPLIT) x -= __builtin_convertvector(af[mt][pl], w4f32x8);
        }
      } else {
#pragma unroll
        for (int pl = 0; pl < SPLIT; ++pl)
          af[mt][pl] = *reinterpret_cast<const w4bf16x8*>(A + pl * A_PLANE + lh * A_OCT + row * 16);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt + 1 < NT) read_b(nt + 1, bfr[(nt + 1) & 1]);     // lands behind this tile's 4 * SPLIT MFMAs
      w4_split_mfma2<SPLIT>(bfr[nt & 1], af[0], af[1], acc[0][nt], acc[1][nt]);
    }
  };

  if (!setup()) return;
  zero_acc();
  dma(0);
  if (p.nchunk > 1) dma(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // first item only: later items count past their predecessor's stores

  for (;;) {
    for (int c = 0; c < p.nchunk; ++c) {
      // this wave's pieces of chunk c have landed: younger operations are chunk c + 1's kDma pieces and, in the first
      // two chunks of an item, the kStores product stores of the previous item issued between them
      if (c + 1 >= p.nchunk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (c < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma + kStores) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
      // everybody's pieces of chunk c are in, and everybody is done with chunk c - 1 = the stage chunk c + 2 goes to
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (c + 2 < p.nchunk) dma(c + 2);
      mfma_chunk(c);
    }
    // ---- item tail: products to M[pos][cout / 4][tile][4] (lane = tile, registers 4g..4g+3 = four consecutive couts:
    // a half-wave writes 512 contiguous bytes per instruction).  ALL kStores stores are issued (rows past the last
    // tile / couts past Cout go to a junk line behind the workspace) so that the next item's waits can count them
    float* Mp = p.M + (size_t)pos * p.mplane;
    float* const junk = p.M + (size_t)p.npos * p.mplane + lane * 4;
    const int mb_cur = mb, tn_cur = tn;
    slot += cus;
    const bool more = setup();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // all waves are done with the last chunk's stage
    if (more) {
      dma(0);
      if (p.nchunk > 1) dma(1);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = mb_cur * W4_M + wm * 64 + mt * 32 + li;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = tn_cur * (64 * TN) + (wn * NT + nt) * 32 + 8 * g + 4 * lh;
          float* dst = (m < p.T && n < p.Cout) ? Mp + ((size_t)(n >> 2) * p.T + m) * 4 : junk;
          *reinterpret_cast<w4f32x4*>(dst) =
              w4f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
        }
    }
    if (!more) break;
    zero_acc();
  }
}

// ---- the forward GEMM on fp32 V (the default): one continuous pipeline over the workgroup's whole item list
// Per chunk c of the stream (stage s = c % 3), after ONE barrier: the LDS-DMA of A(c + 3) and B(c + 2) is issued -- across
// item boundaries, so an item's first chunks are already there when its predecessor ends (the pre-split kernel above pays a
// full memory latency at every item head) --, the MFMAs of chunk c run on the bf16 pieces of A(c) that were split one chunk
// earlier, and between them the wave reads its 64 rows of A(c + 1) as fp32 and splits them (the conversions / subtractions
// of wino4_in_kernel's piece loop: pieces and products bit-identical to the pre-split form).  A is consumed one chunk
// ahead of B, so three stages of each suffice: 3 x (16 + 24) KiB.
// RS (row split): the 8 waves are 8 row groups of 32 tiles, each against ALL 64 * TN couts (MT = 1, NT = 2 * TN weight tiles),
// instead of 4 row groups of 64 tiles x 2 cout halves: every row of A is split into its bf16 pieces by ONE wave, not by two
// (half the split's VALU work per chunk: the split is 9 % of the kernel, profiles/r05_gemm_notes.md), for twice the B-fragment
// LDS reads.  Same products in the same order per accumulator: bit-identical.
// EXP != 0: timing experiments only (WRONG results), see launch_wino4_gemm32.
template <int SPLIT, int TN, int EXP = 0, bool RS = false>
__global__ __launch_bounds__(512, 2) void wino4_gemm32_kernel(const Wino4GemmArgs p) {
  constexpr int A_QUAD = W4_M * 16, A_BYTES = 4 * A_QUAD;                                // [k-quad][row][4 f32]
  constexpr int U_OCT = 64 * 16, U_PLANE = 2 * U_OCT, U_BYTES = SPLIT * U_PLANE;         // one 64-cout weight unit
  constexpr int B_BYTES = TN * U_BYTES;
  constexpr int NS = 3;
  constexpr int A_INSTR = A_BYTES / 1024, B_INSTR = B_BYTES / 1024;
  constexpr int kA = A_INSTR / 8, kBw = (B_INSTR + 7) / 8;     // 1 KiB DMA pieces EVERY wave issues per chunk
  constexpr int kDma = kA + kBw;
  constexpr int MT = RS ? 1 : 2, NT = RS ? 2 * TN : TN;
  constexpr int kStores = MT * NT * 4;
  static_assert(A_INSTR % 8 == 0 && B_BYTES % 1024 == 0, "tiles must be whole DMA pieces");
  static_assert(kDma + kStores <= 63, "vmcnt is a 6-bit counter");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ast = smem;
  char* const Bst = smem + NS * A_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = RS ? wave : wave & 3, wn = RS ? 0 : wave >> 2;     // 4 row groups of 64 tiles x 2 channel halves (RS: 8 x 1)
  const int row0 = RS ? wm * 32 : wm * 64;
  const int li = lane & 31, lh = lane >> 5;

  const int cus = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, P = p.npos * p.tiles_n;
  const int mbg = p.mbg, us = 8 * mbg;
  const int units_pg = (P + 7) >> 3, nunits = units_pg * ((p.m_blocks + mbg - 1) / mbg);
  int slot = blockIdx.x >> 3;

  struct Item { int mb, pos, tn; const char* a; const char* w; };
  // -> true when `slot` names an item (advancing over the holes of ragged units), false at the end
  auto setup = [&](Item& it) __attribute__((always_inline)) -> bool {
    for (;; slot += cus) {
      const int su = slot / us, unit = su * 8 + xcd, w = slot - su * us;
      if (unit >= nunits) return false;
      const int mg = unit / units_pg, pg = unit - mg * units_pg;   // panel groups fastest: neighbouring XCDs share V tiles in the MALL
      const int pnl = pg * 8 + (w & 7);
      it.mb = mg * mbg + (w >> 3);
      if (pnl >= P || it.mb >= p.m_blocks) continue;
      it.pos = pnl / p.tiles_n; it.tn = pnl - it.pos * p.tiles_n;
      it.a = p.V + ((size_t)it.pos * p.m_blocks + it.mb) * p.nchunk * A_BYTES;
      it.w = p.wpk + ((size_t)it.pos * p.units + (size_t)it.tn * TN) * p.nchunk * U_BYTES;
      return true;
    }
  };
  auto uniform = [](const char* q) __attribute__((always_inline)) -> const char* {
    const size_t v = reinterpret_cast<size_t>(q);
    return reinterpret_cast<const char*>(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)v));
  };
  auto dma_a1 = [&](const Item& it, int c, int st, int jj) __attribute__((always_inline)) {
    const int i = wave + 8 * jj;
    const char* src = uniform(it.a + (size_t)c * A_BYTES + i * 1024);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(lane * 16)),
                                     (__attribute__((address_space(3))) void*)(Ast + st * A_BYTES + i * 1024), 16, 0, 0);
  };
  auto dma_a = [&](const Item& it, int c, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < kA; ++jj) dma_a1(it, c, st, jj);
  };
  // a wave with no weight piece left re-copies its previous one (same bytes to the same place): every wave issues kBw
  auto dma_b1 = [&](const Item& it, int c, int st, int jj) __attribute__((always_inline)) {
    int i = wave + 8 * jj;
    if (i >= B_INSTR) i -= 8;
    const int uu = i / (U_BYTES / 1024), r = i % (U_BYTES / 1024);
    const char* src = uniform(it.w + ((size_t)uu * p.nchunk + c) * U_BYTES + r * 1024);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(lane * 16)),
                                     (__attribute__((address_space(3))) void*)(Bst + st * B_BYTES + i * 1024), 16, 0, 0);
  };
  auto dma_b = [&](const Item& it, int c, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < kBw; ++jj) dma_b1(it, c, st, jj);
  };

  w4f32x16 acc[MT][NT];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  w4bf16x8 af[MT][SPLIT];
  // this wave's rows of the A tile in stage `st`: fp32 -> bf16 pieces (lane = row li of the 32-row block, k-octet lh)
  auto read_a = [&](int st, int mt, w4f32x4 (&q)[2]) __attribute__((always_inline)) {
    const char* A = Ast + st * A_BYTES + (row0 + mt * 32 + li) * 16;
    q[0] = *reinterpret_cast<const w4f32x4*>(A + (2 * lh) * A_QUAD);
    q[1] = *reinterpret_cast<const w4f32x4*>(A + (2 * lh + 1) * A_QUAD);
  };
  auto split_a = [&](const w4f32x4 (&q)[2], w4bf16x8 (&dst)[SPLIT]) __attribute__((always_inline)) {
    w4f32x8 x{q[0][0], q[0][1], q[0][2], q[0][3], q[1][0], q[1][1], q[1][2], q[1][3]};
    if (EXP == 1) {                     // no split arithmetic: the raw bits as "pieces"
#pragma unroll
      for (int pl = 0; pl < SPLIT; ++pl) dst[pl] = __builtin_bit_cast(w4bf16x8, q[pl & 1]);
      return;
    }
#pragma unroll
    for (int pl = 0; pl < SPLIT; ++pl) {
      dst[pl] = __builtin_convertvector(x, w4bf16x8);
      if (pl + 1 < SPLIT) x -= __builtin_convertvector(dst[pl], w4f32x8);
    }
  };
  // MFMAs of the chunk in stage `st` (weights as the first operand: D = U * V^T, a lane owns one tile, its registers the
  // couts) with the split of the NEXT chunk's rows (stage `stn`) between them
  // `piece(k)`, k < kDma: issues the k-th LDS-DMA piece of this chunk's prefetch -- between the MFMA groups, behind the
  // chunk's own ds_reads (EXP 5; the guide prices a piece at 100-185 cycles in a phase that carries ds_reads, 25-60 later)
  auto mfma_chunk = [&](int st, int stn, auto&& piece) __attribute__((always_inline)) {
    const char* B = Bst + st * B_BYTES;
    w4bf16x8 afn[MT][SPLIT];
    w4f32x4 raw[MT][2];
    auto read_b = [&](int nt, w4bf16x8 (&dst)[SPLIT]) __attribute__((always_inline)) {
      const int n = (wn * NT + nt) * 32 + li;
#pragma unroll
      for (int pl = 0; pl < SPLIT; ++pl)
        dst[pl] = *reinterpret_cast<const w4bf16x8*>(B + (n >> 6) * U_BYTES + pl * U_PLANE + lh * U_OCT + (n & 63) * 16);
    };
    if constexpr (RS) {
      // weight tiles two at a time against the wave's one row fragment; the next pair's fragments land behind this pair's
      // 4 * SPLIT MFMAs; the split of the next chunk's rows sits behind the first pair
      w4bf16x8 bfr[2][2][SPLIT];
      read_b(0, bfr[0][0]); read_b(1, bfr[0][1]);
      read_a(stn, 0, raw[0]);
      constexpr int NG = NT / 2, PPG = (kDma + NG - 1) / NG;
#pragma unroll
      for (int np = 0; np < NG; ++np) {
        if (np + 1 < NG) { read_b(2 * np + 2, bfr[(np + 1) & 1][0]); read_b(2 * np + 3, bfr[(np + 1) & 1][1]); }
        w4_split_mfma2w<SPLIT>(bfr[np & 1][0], bfr[np & 1][1], af[0], acc[0][2 * np], acc[0][2 * np + 1]);
        if (np == 0) split_a(raw[0], afn[0]);
#pragma unroll
        for (int k = np * PPG; k < (np + 1) * PPG && k < kDma; ++k) piece(k);
      }
    } else {
      w4bf16x8 bfr[2][SPLIT];
      read_b(0, bfr[0]);
      read_a(stn, 0, raw[0]);
      read_a(stn, MT - 1, raw[MT - 1]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt + 1 < NT) read_b(nt + 1, bfr[(nt + 1) & 1]);     // lands behind this tile's 4 * SPLIT MFMAs
        w4_split_mfma2<SPLIT>(bfr[nt & 1], af[0], af[MT - 1], acc[0][nt], acc[MT - 1][nt]);
        if (nt < MT) split_a(raw[nt < MT ? nt : 0], afn[nt < MT ? nt : 0]);
      }
#pragma unroll
      for (int k = 0; k < kDma; ++k) piece(k);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int pl = 0; pl < SPLIT; ++pl) af[mt][pl] = afn[mt][pl];
  };

  Item cur, nxt;
  if (!setup(cur)) return;
  slot += cus;
  bool has_next = setup(nxt);
  zero_acc();
  // prologue: A(0..2), B(0..1) of the first item (nchunk >= 3)
  dma_a(cur, 0, 0); dma_a(cur, 1, 1); dma_a(cur, 2, 2);
  dma_b(cur, 0, 0); dma_b(cur, 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  {
    w4f32x4 raw[2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { read_a(0, mt, raw); split_a(raw, af[mt]); }
  }
  int st = 0;                                   // stage of the current chunk = (chunks so far) % 3
  for (;;) {
    for (int c = 0; c < p.nchunk; ++c) {
      // this wave's pieces of B(c) and A(c + 1) have landed: they were issued two chunks ago; younger operations are the
      // kDma pieces of the previous chunk and, in the first two chunks of an item, the kStores product stores of the
      // previous item issued between them.  At the very end of the list (no next item) batches thin out: wait for all
      if (EXP == 2) {                   // no wait for the DMA pieces
      } else if (!has_next && c + 2 >= p.nchunk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (c < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma + kStores) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
      // everybody's pieces are in, and everybody is done with chunk c - 1: with A(c) (stage st) and B(c - 1) (stage st + 2)
      if (EXP == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // no barrier
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const int st1 = st == 2 ? 0 : st + 1, st2 = st == 0 ? 2 : st - 1;
      if (EXP == 5) {                   // the prefetch's pieces between the MFMA groups (B first: it is needed first)
        const bool a_cur = c + 3 < p.nchunk, b_cur = c + 2 < p.nchunk;
        mfma_chunk(st, st1, [&](int k) __attribute__((always_inline)) {
          if (k < kBw) {
            if (b_cur) dma_b1(cur, c + 2, st2, k);
            else if (has_next) dma_b1(nxt, c + 2 - p.nchunk, st2, k);
          } else {
            if (a_cur) dma_a1(cur, c + 3, st, k - kBw);
            else if (has_next) dma_a1(nxt, c + 3 - p.nchunk, st, k - kBw);
          }
        });
      } else {
        if (EXP == 4) {                 // no LDS-DMA at all (the stages keep the prologue's data)
        } else {
          if (c + 3 < p.nchunk) dma_a(cur, c + 3, st);
          else if (has_next) dma_a(nxt, c + 3 - p.nchunk, st);
          if (c + 2 < p.nchunk) dma_b(cur, c + 2, st2);
          else if (has_next) dma_b(nxt, c + 2 - p.nchunk, st2);
        }
        mfma_chunk(st, st1, [](int) {});
      }
      st = st1;
    }
    // ---- item tail: products to M[pos][cout / 4][tile][4] (lane = tile, registers 4g..4g+3 = four consecutive couts:
    // a half-wave writes 512 contiguous bytes per instruction).  ALL kStores stores are issued (rows past the last
    // tile / couts past Cout go to a junk line behind the workspace) so that the next item's waits can count them
    float* Mp = p.M + (size_t)cur.pos * p.mplane;
    float* const junk = p.M + (size_t)p.npos * p.mplane + lane * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = cur.mb * W4_M + row0 + mt * 32 + li;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = cur.tn * (64 * TN) + (wn * NT + nt) * 32 + 8 * g + 4 * lh;
          float* dst = (m < p.T && n < p.Cout) ? Mp + ((size_t)(n >> 2) * p.T + m) * 4 : junk;
          *reinterpret_cast<w4f32x4*>(dst) =
              w4f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
        }
    }
    if (!has_next) break;
    cur = nxt;
    slot += cus;
    has_next = setup(nxt);
    zero_acc();
  }
}

struct Wino4InArgs {
  const float* in;
  char* V;
  int N, H, W, Cin, in_cs;
  int tiles_y, tiles_x, T;
  int pad_t, pad_l;
  int nchunk, m_blocks;
  const float* up_src;       // UP: channels >= Cin - up_C are the exact 2x bilinear upsample of up_src [N, H/2, W/2, up_cs]
  int up_C, up_cs;
  int order, ncp;            // order 1: 1-D grid, channel-chunk pairs fastest inside an XCD's contiguous range of tile groups
  int repl;                  // CRESTE_CONV_REPLICATE_PAD: window pixels outside the image take the nearest border pixel (wino4_in1_kernel<false>)
};

// One workgroup = 16 tiles x 32 channels (two chunks), tile groups the fast grid dimension; thread = (tile, channel
// PAIR): 36 8-byte loads (a wave reads the 128 contiguous bytes of 4 pixels per instruction), B^T d B on packed pairs in
// registers, split; the packed bf16 pairs cross an LDS tile as dwords, ONE PIECE AT A TIME (36 KiB: four workgroups per
// CU), so that the write side stores 16 bytes per lane = (position, chunk, k-octet, tile) with 16 consecutive tiles =
// 256 contiguous bytes of V.  (Measured forms, profiles/r03_winograd_notes.md: the 496-channel layer sits at 4.0 TB/s
// whatever the structure; this one is 15 % faster than scalar channels + all pieces in LDS on 256 channels at 256 x 256.)
typedef __bf16 w4bf16x2 __attribute__((ext_vector_type(2)));
typedef float w4f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void w4_bt2(const w4f32x2 (&d)[6], w4f32x2 (&t)[6]) {
  const w4f32x2 a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
  t[0] = (4.f * d[0] - 5.f * d[2]) + d[4];
  t[1] = a + b;
  t[2] = a - b;
  t[3] = c + e;
  t[4] = c - e;
  t[5] = (4.f * d[1] - 5.f * d[3]) + d[5];
}
// UP: the conv input is cat([in, up2x(up_src)]) (the reference's Up block / DeconvHead.up2) and is never materialised:
// a pair of upsampled channels forms its 6x6 window from the 4x4 source patch the window maps to -- source rows
// 2ty-1 .. 2ty+2, weights 0.25 / 0.75 -- with the expression of the stand-alone kernel (csrc/pointwise.hip:
// wy0 * (hx * v00 + lx * v01) + wy1 * (hx * v10 + lx * v11), horizontal sums shared by the rows that use them), so the
// result is bit-identical to upsampling first.  Windows that touch the image border (clamped taps, zero padding) take
// the generic per-pixel path.
// (This is the PRE-SPLIT form, V as bf16 pieces: the reference the fp32-V path is checked against bit for bit, CRESTE_W4_F32V=0,
// and the form whose pieces the weight gradient's GEMM consumes.  The forward's default is wino4_in1_kernel below: V as fp32.)
template <int SPLIT, bool UP>
__global__ __launch_bounds__(256, 4) void wino4_in_kernel(const Wino4InArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned tbuf[W4_POS * 4 * 16 * 4];      // [pos][chunk half * 2 + octet][tile][4 pairs]
  const int t = threadIdx.x;
  // gridDim.x is a multiple of 16: workgroup (x, y) sits on XCD x % 8.  Every XCD gets a contiguous range of tile groups,
  // so the 6x6 windows' shared pixels (2 of 6 columns / rows) are re-read from ITS L2 (round-robin: 2.4x the input fetched)
  int tg, chunk0;
  if (p.order) {
    // workgroup L sits on XCD L % 8.  Each XCD owns a contiguous range of tile groups and walks it with the channel-chunk
    // pairs FASTEST: the workgroups in flight on an XCD cover every channel of a few tile groups, so a 128-byte line that
    // two channel groups share (pixel stride 1984 B at 496 channels = 15.5 lines) and the rows / columns neighbouring
    // windows share are fetched from HBM once and then served by that XCD's L2
    const int L = blockIdx.x, xcd = L & 7, idx = L >> 3, q = (p.m_blocks * 16) >> 3;
    const int tgl = idx / p.ncp;
    tg = xcd * q + tgl; chunk0 = (idx - tgl * p.ncp) * 2;
  } else {
    tg = xcd_remap(blockIdx.x, gridDim.x); chunk0 = blockIdx.y * 2;
  }
  const int tl = t >> 4, cp = t & 15;
  const int tile = tg * 16 + tl, ch = chunk0 * W4_CK + cp * 2;
  const int per = p.tiles_y * p.tiles_x;
  w4f32x2 u[6][6];
  {
    w4f32x2 d[6][6];
    const bool ok = tile < p.T && ch < p.Cin;
    const int tcl = ok ? tile : 0;
    const int img = tcl / per, rem = tcl - img * per;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int y0 = 4 * ty - p.pad_t, x0 = 4 * tx - p.pad_l;
    const int C2 = UP ? p.Cin - p.up_C : p.Cin;
    if (UP && ok && ch >= C2) {
      const int H1 = p.H >> 1, W1 = p.W >> 1;
      const float* src = p.up_src + (size_t)img * H1 * W1 * p.up_cs + (ch - C2);
      // ONE branch-free form for interior and border windows (a wave mixes both: the per-pixel border path this replaces
      // ran on 15 % of the waves of a 64 x 64 tile grid at several times the cost).  Window pixel (oy, ox) = (y0 + i, x0 + j)
      // takes PyTorch's taps ya = floor(sy), yb = min(ya + 1, H1 - 1), sy = max(0.5 (oy + 0.5) - 0.5, 0).  With pad 1, y0 is
      // odd and ya = yb0 + (i >> 1) for yb0 = (y0 - 1) / 2 -- except at oy = 0, where the clamp makes sy = 0: there the
      // weights are (1, 0) and the tap standing in for row -1 is the clamped load of row 0, so 1 * r0 + 0 * r0 = r0 as the
      // stand-alone kernel forms it (csrc/pointwise.hip: upsample2x_concat_kernel).  Rows past H1 - 1 clamp likewise (the
      // duplicate holds the tap's own data); window pixels outside the image are the conv's zero padding.
      const int yb0 = 2 * ty - 1, xb0 = 2 * tx - 1;            // pad 1 (conv_wino4_run requires it with up_src): y0 = 4 ty - 1
      w4f32x2 P[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        int ry = yb0 + a; ry = ry < 0 ? 0 : (ry > H1 - 1 ? H1 - 1 : ry);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          int rx = xb0 + b; rx = rx < 0 ? 0 : (rx > W1 - 1 ? W1 - 1 : rx);
          P[a][b] = *reinterpret_cast<const w4f32x2*>(src + ((size_t)ry * W1 + rx) * p.up_cs);
        }
      }
      w4f32x2 hz[4][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int ox = x0 + j;
        const float lx = ox == 0 ? 0.f : ((j & 1) ? 0.75f : 0.25f), hx = 1.f - lx;
#pragma unroll
        for (int a = 0; a < 4; ++a) hz[a][j] = hx * P[a][j >> 1] + lx * P[a][(j >> 1) + 1];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int oy = y0 + i;
        const float wy1 = oy == 0 ? 0.f : ((i & 1) ? 0.75f : 0.25f), wy0 = 1.f - wy1;
        const bool yin = (unsigned)oy < (unsigned)p.H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const w4f32x2 v = wy0 * hz[i >> 1][j] + wy1 * hz[(i >> 1) + 1][j];
          d[i][j] = (yin && (unsigned)(x0 + j) < (unsigned)p.W) ? v : w4f32x2{0.f, 0.f};
        }
      }
    } else if (!ok) {                 // rows past the last tile / channels past Cin: zeros (p.in may be null with UP)
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[i][j] = w4f32x2{0.f, 0.f};
    } else {
      const float* base = p.in + ch;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int yy = y0 + i;
        const bool yok = ok && (unsigned)yy < (unsigned)p.H;
        const size_t rowoff = ((size_t)img * p.H + (yok ? yy : 0)) * p.W;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int xx = x0 + j;
          const bool in = yok && (unsigned)xx < (unsigned)p.W;
          const w4f32x2 v = *reinterpret_cast<const w4f32x2*>(base + (rowoff + (in ? xx : 0)) * p.in_cs);
          d[i][j] = in ? v : w4f32x2{0.f, 0.f};
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const w4f32x2 col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
      w4f32x2 o[6];
      w4_bt2(col, o);
#pragma unroll
      for (int i = 0; i < 6; ++i) u[i][j] = o[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    w4f32x2 o[6];
    w4_bt2(u[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) u[i][j] = o[j];
  }
  const int mb = (tg * 16) >> 8, row0 = (tg * 16) & (W4_M - 1);
  constexpr int UNITS = W4_POS * 4 * 16;       // 16-byte units of one piece: (position, chunk half * 2 + octet, tile)
#pragma unroll
  for (int pl = 0; pl < SPLIT; ++pl) {
    if (pl) __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const w4bf16x2 piece = __builtin_convertvector(u[i][j], w4bf16x2);
        tbuf[(((i * 6 + j) * 4 + (cp >> 2)) * 16 + tl) * 4 + (cp & 3)] = __builtin_bit_cast(unsigned, piece);
        if (pl + 1 < SPLIT) u[i][j] -= __builtin_convertvector(piece, w4f32x2);
      }
    __syncthreads();
    for (int uidx = t; uidx < UNITS; uidx += 256) {
      const int tlw = uidx & 15, co = (uidx >> 4) & 3, pos = uidx >> 6;
      const int chunk = chunk0 + (co >> 1), oct = co & 1;
      if (chunk >= p.nchunk) continue;
      char* dst = p.V + (((((size_t)pos * p.m_blocks + mb) * p.nchunk + chunk) * SPLIT + pl) * 2 + oct) * (size_t)(W4_M * 16) +
                  (size_t)(row0 + tlw) * 16;
      *reinterpret_cast<w4f32x4*>(dst) = *reinterpret_cast<const w4f32x4*>(tbuf + (size_t)uidx * 4);
    }
  }
}

__device__ __forceinline__ void w4_bt1(const float (&d)[6], float (&t)[6]) {
  const float a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
  t[0] = (4.f * d[0] - 5.f * d[2]) + d[4];
  t[1] = a + b;
  t[2] = a - b;
  t[3] = c + e;
  t[4] = c - e;
  t[5] = (4.f * d[1] - 5.f * d[3]) + d[5];
}

// ---- the transforms that FIT BESIDE the GEMM (round 5).  The persistent GEMM workgroup holds 2 x 200 of a SIMD's 512 VGPRs
// and 120 of the CU's 160 KiB of LDS: a kernel of another stream whose waves need <= 112 VGPRs and whose workgroup needs
// <= 40 KiB of LDS is dispatched onto the SAME CUs and streams while the matrix pipe works (scripts/coresidency_probe.py: a
// copy kernel of that size moves 2.4-3.2 TB/s beside the GEMM, which then runs 1.2-1.4 x longer: 20-25 % less time for the
// pair than one after the other).  The pair / quad transforms above need 128-150 VGPRs and only run on CUs the GEMM does
// not hold.  Same arithmetic, expression for expression: V, M and the outputs are bit-identical to theirs.
//
// Input transform: one workgroup = 8 tiles x 32 channels (two chunks); thread = (tile, channel): 36 4-byte loads (a wave
// reads the 128 contiguous bytes of two pixels per instruction), B^T d B in place, the fp32 values cross an LDS tile twelve
// positions at a time ([position][k-quad of the 32 channels][8 tiles][4 floats], rows padded by 16 bytes) and leave as
// 128-byte runs of V (8 consecutive tiles of one k-quad).
constexpr int W4S_ROW = 8 * 16 + 16;
template <bool UP>
__global__ __launch_bounds__(256) void wino4_in1_kernel(const Wino4InArgs p) {
  __shared__ __attribute__((aligned(16))) char tb[12 * 8 * W4S_ROW];
  const int t = threadIdx.x;
  int tg, chunk0;
  if (p.order) {
    // as wino4_in_kernel: an XCD owns a contiguous range of tile groups and walks it with the channel-chunk pairs fastest
    const int L = blockIdx.x, xcd = L & 7, idx = L >> 3, q = (p.m_blocks * 32) >> 3;
    const int tgl = idx / p.ncp;
    tg = xcd * q + tgl; chunk0 = (idx - tgl * p.ncp) * 2;
  } else {
    tg = xcd_remap(blockIdx.x, gridDim.x); chunk0 = blockIdx.y * 2;
  }
  const int tl = t >> 5, c = t & 31;
  const int tile = tg * 8 + tl, ch = chunk0 * W4_CK + c;
  const int per = p.tiles_y * p.tiles_x;
  float d[6][6];
  {
    const bool ok = tile < p.T && ch < p.Cin;
    const int tcl = ok ? tile : 0;
    const int img = tcl / per, rem = tcl - img * per;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int y0 = 4 * ty - p.pad_t, x0 = 4 * tx - p.pad_l;
    const int C2 = UP ? p.Cin - p.up_C : p.Cin;
    if (UP && ok && ch >= C2) {
      // the exact 2x bilinear upsample of up_src formed from the 4x4 source patch of the window (see wino4_in_kernel)
      const int H1 = p.H >> 1, W1 = p.W >> 1;
      const float* src = p.up_src + (size_t)img * H1 * W1 * p.up_cs + (ch - C2);
      const int yb0 = 2 * ty - 1, xb0 = 2 * tx - 1;
      float hz[4][6];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        int ry = yb0 + a; ry = ry < 0 ? 0 : (ry > H1 - 1 ? H1 - 1 : ry);
        float P[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          int rx = xb0 + b; rx = rx < 0 ? 0 : (rx > W1 - 1 ? W1 - 1 : rx);
          P[b] = src[((size_t)ry * W1 + rx) * p.up_cs];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int ox = x0 + j;
          const float lx = ox == 0 ? 0.f : ((j & 1) ? 0.75f : 0.25f), hx = 1.f - lx;
          hz[a][j] = hx * P[j >> 1] + lx * P[(j >> 1) + 1];
        }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int oy = y0 + i;
        const float wy1 = oy == 0 ? 0.f : ((i & 1) ? 0.75f : 0.25f), wy0 = 1.f - wy1;
        const bool yin = (unsigned)oy < (unsigned)p.H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float v = wy0 * hz[i >> 1][j] + wy1 * hz[(i >> 1) + 1][j];
          d[i][j] = (yin && (unsigned)(x0 + j) < (unsigned)p.W) ? v : 0.f;
        }
      }
    } else if (!ok) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[i][j] = 0.f;
    } else {
      const float* base = p.in + ch;
      const bool repl = p.repl != 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int yy = y0 + i;
        const bool yok = (unsigned)yy < (unsigned)p.H;
        const int yc = yy < 0 ? 0 : (yy > p.H - 1 ? p.H - 1 : yy);
        const size_t rowoff = ((size_t)img * p.H + (yok ? yy : (repl ? yc : 0))) * p.W;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int xx = x0 + j;
          const bool xok = (unsigned)xx < (unsigned)p.W;
          const int xc = xx < 0 ? 0 : (xx > p.W - 1 ? p.W - 1 : xx);
          const bool in = yok && xok;
          const float v = base[(rowoff + (xok ? xx : (repl ? xc : 0))) * p.in_cs];
          d[i][j] = (in || repl) ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
      float o[6];
      w4_bt1(col, o);
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i][j] = o[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float o[6];
    w4_bt1(d[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) d[i][j] = o[j];
  }
  const int mb = (tg * 8) >> 8, row0 = (tg * 8) & (W4_M - 1);
  // write side: thread = (position mod 4, k-quad of the 32 channels, tile)
  const int tlw = t & 7, kq8 = (t >> 3) & 7, p4 = t >> 6;
  const int chunk = chunk0 + (kq8 >> 2);
  const size_t pos_stride = (size_t)p.m_blocks * p.nchunk * (size_t)(4 * W4_M * 16);
  char* dst0 = p.V + (((size_t)mb * p.nchunk + chunk) * 4 + (kq8 & 3)) * (size_t)(W4_M * 16) + (size_t)(row0 + tlw) * 16 +
               p4 * pos_stride;
  const char* src0 = tb + (p4 * 8 + kq8) * W4S_ROW + tlw * 16;
  char* wr0 = tb + (c >> 2) * W4S_ROW + tl * 16 + (c & 3) * 4;
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    if (g) __syncthreads();
#pragma unroll
    for (int pp = 0; pp < 12; ++pp) {
      const int pos = g * 12 + pp;
      *reinterpret_cast<float*>(wr0 + pp * 8 * W4S_ROW) = d[pos / 6][pos % 6];
    }
    __syncthreads();
    if (chunk < p.nchunk) {
#pragma unroll
      for (int it = 0; it < 3; ++it)
        *reinterpret_cast<w4f32x4*>(dst0 + (size_t)(g * 12 + 4 * it) * pos_stride) =
            *reinterpret_cast<const w4f32x4*>(src0 + 4 * it * 8 * W4S_ROW);
    }
  }
}

struct Wino4OutArgs {
  const float* M;
  const float* bias;
  const float* res;
  const float* row_mask;
  float* out;
  float* out_amax;
  int N, Ho, Wo, Cout, out_cs, out_co, res_cs, act;
  int tiles_y, tiles_x, T;
  long mplane;
  int order, ncg, ntg8;      // order 1: 1-D grid, cout groups fastest inside an XCD's range of ntg8 tile groups
  float* stats;              // creste_conv_desc.out_stats: [ceil(T / 16)][2][Cout] sums of what this workgroup writes, or nullptr
  int phaseC;                // CRESTE_CONV_PHASE2X: > 0 = channels per phase; channel n of pixel (oy, ox) is channel n % phaseC of pixel
                             // (2 oy + (n / phaseC >> 1), 2 ox + (n / phaseC & 1)) of the [N, 2 Ho, 2 Wo] output; the outermost
                             // ring of that image is written WITHOUT the activation (upconv2x_ring_fix_kernel finishes it)
};

constexpr int W4O_TILES = 16;

// Output transform that fits beside the GEMM: one workgroup = 16 consecutive tiles x 32 couts.  Read side: thread = (tile,
// channel PAIR), 36 x 8-byte loads row by row of the 6x6 product tile (two lanes share a quad's 16 bytes, the 16 tiles of a
// quad are 256 contiguous bytes of M); all 16 output pixels of the tile cross ONE LDS tile; write side: thread = (pixel,
// channel quad): 8 lanes write the 128 contiguous bytes of one NHWC pixel and read bias / residual the same way.
constexpr int W4O2_QUADS = 8, W4O2_ROW = W4O2_QUADS * 4 + 4;
__global__ __launch_bounds__(256) void wino4_out2_kernel(const Wino4OutArgs p) {
  __shared__ __attribute__((aligned(16))) float tilebuf[W4O_TILES * 16 * W4O2_ROW];
  __shared__ float scratch[4];
  const int Q = p.Cout >> 2;
  const int t = threadIdx.x;
  int tile0, quad0;
  if (p.order) {
    const int L = blockIdx.x, xcd = L & 7, idx = L >> 3, tgl = idx / p.ncg;
    tile0 = (xcd * p.ntg8 + tgl) * W4O_TILES; quad0 = (idx - tgl * p.ncg) * W4O2_QUADS;
    if (tile0 >= p.T) return;
  } else {
    tile0 = blockIdx.x * W4O_TILES; quad0 = blockIdx.y * W4O2_QUADS;
  }
  const int tl = t & (W4O_TILES - 1), pr = t >> 4;
  w4f32x2 y[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) y[a][b] = w4f32x2{0.f, 0.f};
  {
    const int tile = tile0 + tl, quad = quad0 + (pr >> 1);
    if (tile < p.T && quad < Q) {
      const float* src = p.M + ((size_t)quad * p.T + tile) * 4 + (pr & 1) * 2;
      const size_t plane = (size_t)p.mplane;
      constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        w4f32x2 m[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) m[j] = __builtin_nontemporal_load(reinterpret_cast<const w4f32x2*>(src + (i * 6 + j) * plane));
        const w4f32x2 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        const w4f32x2 r[4] = {(m[0] + s12) + s34, d12 + 2.f * d34, s12 + 4.f * s34, (d12 + 8.f * d34) + m[5]};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (AT[a][i] == 0.f) continue;
#pragma unroll
          for (int b = 0; b < 4; ++b) y[a][b] = AT[a][i] == 1.f ? y[a][b] + r[b] : (AT[a][i] == -1.f ? y[a][b] - r[b] : y[a][b] + AT[a][i] * r[b]);
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      *reinterpret_cast<w4f32x2*>(tilebuf + (tl * 16 + a * 4 + b) * W4O2_ROW + pr * 2) = y[a][b];
  __syncthreads();
  float vmax = 0.f;
  const int cq = t & 7, n = (quad0 + cq) * 4;
  const int per = p.tiles_y * p.tiles_x;
  const bool nok = quad0 + cq < Q;
  const w4f32x4 bs = (nok && p.bias) ? *reinterpret_cast<const w4f32x4*>(p.bias + n) : w4f32x4{0.f, 0.f, 0.f, 0.f};
  w4f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (nok) {
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int pl = pass * 32 + (t >> 3);            // pixel slot: tile pl / 16, output row (pl / 4) % 4, column pl % 4
      const int tile = tile0 + (pl >> 4);
      if (tile >= p.T) continue;
      const int img = tile / per, rem = tile - img * per;
      const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
      const int oy = 4 * ty + ((pl >> 2) & 3), ox = 4 * tx + (pl & 3);
      if (oy >= p.Ho || ox >= p.Wo) continue;
      w4f32x4 v = *reinterpret_cast<const w4f32x4*>(tilebuf + pl * W4O2_ROW + cq * 4) + bs;
      if (p.phaseC > 0) {
        const int ph = n / p.phaseC, nn = n - ph * p.phaseC;
        const int oy2 = 2 * oy + (ph >> 1), ox2 = 2 * ox + (ph & 1), H2 = 2 * p.Ho, W2 = 2 * p.Wo;
        const bool ring = oy2 == 0 || ox2 == 0 || oy2 == H2 - 1 || ox2 == W2 - 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ring ? v[e] : act_apply(v[e], p.act);
        *reinterpret_cast<w4f32x4*>(p.out + (((long)img * H2 + oy2) * W2 + ox2) * p.out_cs + p.out_co + nn) = v;
        continue;
      }
      const long mrow = ((long)img * p.Ho + oy) * p.Wo + ox;
      if (p.res) v += *reinterpret_cast<const w4f32x4*>(p.res + mrow * p.res_cs + n);
      const float rmask = p.row_mask ? p.row_mask[mrow] : 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = act_apply(v[e], p.act) * rmask;
        vmax = fmaxf(vmax, fabsf(v[e]));
      }
      *reinterpret_cast<w4f32x4*>(p.out + mrow * p.out_cs + p.out_co + n) = v;
      s1 += v; s2 += v * v;
    }
  }
  if (p.stats) {
    // per-channel sums over the workgroup's 16 tiles: the 32 threads that share a channel quad (8 lanes apart) in a fixed
    // order -- within a wave by xor shuffles, across the four waves through the (now free) tile buffer
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], o); s2[e] += __shfl_xor(s2[e], o); }
    __syncthreads();                                   // every thread is done reading tilebuf
    if ((t & 63) < 8) {
      float* dst = tilebuf + ((t >> 6) * 8 + cq) * 8;
      *reinterpret_cast<w4f32x4*>(dst) = s1;
      *reinterpret_cast<w4f32x4*>(dst + 4) = s2;
    }
    __syncthreads();
    if (t < 64) {                                      // thread = (channel quad t >> 3, k = (t >> 2) & 1, element t & 3)
      const int q = t >> 3, k = (t >> 2) & 1, e = t & 3, c = (quad0 + q) * 4 + e;
      if (quad0 + q < Q) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) a += tilebuf[(w * 8 + q) * 8 + k * 4 + e];
        p.stats[((size_t)(tile0 / W4O_TILES) * 2 + k) * p.Cout + c] = a;
      }
    }
  }
  if (p.out_amax) block_amax_update(vmax, p.out_amax, scratch);
}

// ---- Output transform of conv 1 fused with the input transform of conv 2 (round 5): the conv1 -> conv2 pairs of the
// reference's Up blocks (effnet.py:15-28: conv3x3 + BN + ReLU twice; DeconvHead.up1, inpainting.py:52-68).  Unfused, conv 1's
// output Y crosses HBM twice (written by its output transform, read 2.25x -- mostly from L2 -- by conv 2's input transform)
// between the two 2.25x-sized tensors M1 and V2 that have to cross it anyway.  Here one workgroup = a block of 8 x 16 tiles
// (32 x 64 pixels) x ONE channel quad:
//   phase 1  thread = tile of the block PLUS its one-tile ring (10 x 18 tiles: conv 2's 6x6 windows reach one pixel into
//            the neighbouring tiles): 36 16-byte loads of M1, Y = act(A^T M A + bias) with wino4_out2_kernel's expressions,
//            the 34 x 66 pixels conv 2 needs go to four LDS planes (pixels outside the image: conv 2's zero padding);
//   phase 2  thread = (tile, channel): its 6x6 window from the plane, B^T d B with wino4_in1_kernel's expressions, 36 4-byte
//            stores; lane = (tile of a block row, channel of the quad), so a wave's store is 256 contiguous bytes of V2.
// The ring tiles are re-read by the neighbouring blocks: blocks walk an XCD's contiguous block range fastest, so those
// re-reads come from its L2 / the infinity cache.  M1 and V2 are what the unfused kernels read / write, bit for bit; Y is
// never materialised.  Channel quads past Cout (the padded tail of V2's last 16-channel chunk) are written as zeros.
struct Wino4OutInArgs {
  const float* M;
  const float* bias;
  char* V;
  int Ho, Wo, Cout, act;
  int tiles_y, tiles_x, T;
  long mplane;
  int nchunk, m_blocks;      // of V2 (Cin of conv 2 = Cout of conv 1)
  int by, bx, nblk, q8;      // blocks per image column / row, blocks in all, blocks per XCD
};
constexpr int OI_BH = 8, OI_BW = 16;
constexpr int OI_ROWS = 4 * OI_BH + 2, OI_RS = 4 * OI_BW + 2 + 1;            // 34 rows of 66 (+1) floats
constexpr int OI_PS = (OI_ROWS * OI_RS + 63) / 64 * 64 + 1;                   // plane stride = 1 mod 64: conflict-free window reads
__global__ __launch_bounds__(256) void wino4_outin_kernel(const Wino4OutInArgs p) {
  __shared__ float yb[4 * OI_PS];
  const int t = threadIdx.x;
  const int L = blockIdx.x, xcd = L & 7, idx = L >> 3;
  const int quad = idx / p.q8, blk = xcd * p.q8 + (idx - quad * p.q8);
  if (blk >= p.nblk) return;
  const int per_img = p.by * p.bx;
  const int img = blk / per_img, rem = blk - img * per_img;
  const int ty0 = (rem / p.bx) * OI_BH, tx0 = (rem % p.bx) * OI_BW;
  const int Q = p.Cout >> 2;
  const bool live = quad < Q;
  if (live) {
    // thread = (tile of the 10 x 18 ring block, channel PAIR): two lanes share a quad's 16 bytes; 360 items in two rounds
    // (rolled: 80 registers, so that the kernel fits beside another stream's GEMM workgroup like the transforms above)
#pragma unroll 1
    for (int item = t; item < 2 * (OI_BH + 2) * (OI_BW + 2); item += 256) {
      const int ht = item >> 1, pr = item & 1;
      const int hy = ht / (OI_BW + 2), hx = ht - hy * (OI_BW + 2);
      const int ty = ty0 + hy - 1, tx = tx0 + hx - 1;
      const bool tile_ok = (unsigned)ty < (unsigned)p.tiles_y && (unsigned)tx < (unsigned)p.tiles_x;
      w4f32x2 y[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) y[a][b] = w4f32x2{0.f, 0.f};
      if (tile_ok) {
        const int tile = (img * p.tiles_y + ty) * p.tiles_x + tx;
        const float* src = p.M + ((size_t)quad * p.T + tile) * 4 + pr * 2;
        const size_t plane = (size_t)p.mplane;
        constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          w4f32x2 m[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) m[j] = *reinterpret_cast<const w4f32x2*>(src + (i * 6 + j) * plane);
          const w4f32x2 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
          const w4f32x2 r[4] = {(m[0] + s12) + s34, d12 + 2.f * d34, s12 + 4.f * s34, (d12 + 8.f * d34) + m[5]};
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            if (AT[a][i] == 0.f) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) y[a][b] = AT[a][i] == 1.f ? y[a][b] + r[b] : (AT[a][i] == -1.f ? y[a][b] - r[b] : y[a][b] + AT[a][i] * r[b]);
          }
        }
      }
      const w4f32x2 bs = p.bias ? *reinterpret_cast<const w4f32x2*>(p.bias + quad * 4 + pr * 2) : w4f32x2{0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int py = 4 * hy + a - 3;
        if ((unsigned)py >= (unsigned)OI_ROWS) continue;
        const bool yin = tile_ok && 4 * ty + a < p.Ho;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int px = 4 * hx + b - 3;
          if ((unsigned)px >= (unsigned)(4 * OI_BW + 2)) continue;
          const w4f32x2 v = y[a][b] + bs;
          const bool in = yin && 4 * tx + b < p.Wo;
#pragma unroll
          for (int e = 0; e < 2; ++e) yb[(pr * 2 + e) * OI_PS + py * OI_RS + px] = in ? act_apply(v[e], p.act) : 0.f;
        }
      }
    }
    __syncthreads();
  }
  const int lane = t & 63, wv = t >> 6;
  const int bxl = lane >> 2, c = lane & 3;
  const int chunk = quad >> 2, kq = quad & 3;
  const size_t pos_stride = (size_t)p.m_blocks * p.nchunk * (size_t)(4 * W4_M * 16);
#pragma unroll 1
  for (int rr = 0; rr < 2; ++rr) {
    const int byl = wv * 2 + rr;
    const int ty = ty0 + byl, tx = tx0 + bxl;
    if (ty >= p.tiles_y || tx >= p.tiles_x) continue;
    const int tile = (img * p.tiles_y + ty) * p.tiles_x + tx;
    float d[6][6];
    if (live) {
      const float* wsrc = yb + c * OI_PS + (4 * byl) * OI_RS + 4 * bxl;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[i][j] = wsrc[i * OI_RS + j];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float o[6];
        w4_bt1(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i][j] = o[i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float o[6];
        w4_bt1(d[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) d[i][j] = o[j];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[i][j] = 0.f;
    }
    const int mb = tile >> 8, row = tile & (W4_M - 1);
    char* dst = p.V + (((size_t)mb * p.nchunk + chunk) * 4 + kq) * (size_t)(W4_M * 16) + (size_t)row * 16 + c * 4;
#pragma unroll
    for (int pos = 0; pos < 36; ++pos) *reinterpret_cast<float*>(dst + pos * pos_stride) = d[pos / 6][pos % 6];
  }
}

// U = G g G^T in float64 from the OIHW fp32 weights (x BatchNorm scale, applied in fp32 as the direct packers do),
// rounded once to fp32, split into bf16 pieces: [pos][unit][chunk][piece][k-octet][64][8]
__global__ void wino4_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, __bf16* __restrict__ out,
                                  int Cout, int Cin, int units, int nchunk, int split) {
  const long total = (long)units * 64 * nchunk * W4_CK;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(idx % (nchunk * W4_CK)), co = (int)(idx / (nchunk * W4_CK));
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
    const double G[6][3] = {{0.25, 0., 0.}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                            {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    double Gg[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) Gg[a][b] = G[a][0] * g[0][b] + G[a][1] * g[1][b] + G[a][2] * g[2][b];
    const int unit = co >> 6, nn = co & 63, c = ci / W4_CK, oct = (ci % W4_CK) >> 3, e = ci & 7;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const int pos = a * 6 + b;
        float v = (float)(Gg[a][0] * G[b][0] + Gg[a][1] * G[b][1] + Gg[a][2] * G[b][2]);
        __bf16* dst = out + ((((size_t)pos * units + unit) * nchunk + c) * split) * (2 * 64 * 8) + (size_t)oct * 64 * 8 + nn * 8 + e;
        for (int pl = 0; pl < split; ++pl) {
          const __bf16 piece = (__bf16)v;
          dst[(size_t)pl * (2 * 64 * 8)] = piece;
          v -= (float)piece;
        }
      }
  }
}

static inline bool f32v_default() {
  const char* e = getenv("CRESTE_W4_F32V");
  return e ? atoi(e) != 0 : true;
}
static inline int wino4_split(int prec) {
  return prec == CRESTE_PREC_BF16X6 ? 3 : (prec == CRESTE_PREC_BF16X3 ? 2 : 0);
}
static inline int wino4_units(int Cout) { return ((Cout + 63) / 64 + 3) / 4 * 4; }       // padded to the widest tile (TN = 4)
static inline long wino4_tiles(int N, int Ho, int Wo) { return (long)N * ((Ho + 3) / 4) * ((Wo + 3) / 4); }

bool conv_wino4_supported(int prec, int KH, int KW, int stride, int Cin, int Cout) {
  return wino4_split(prec) > 0 && KH == 3 && KW == 3 && stride == 1 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0;
}

int64_t conv_wino4_weight_bytes(int Cout, int Cin, int prec) {
  const long nchunk = (Cin + W4_CK - 1) / W4_CK;
  return (long)W4_POS * wino4_units(Cout) * nchunk * wino4_split(prec) * 2 * 64 * 16;
}

int conv_wino4_pack(const float* w, const float* scale, void* wpk, int Cout, int Cin, int prec, hipStream_t s) {
  const int units = wino4_units(Cout), nchunk = (Cin + W4_CK - 1) / W4_CK;
  const long total = (long)units * 64 * nchunk * W4_CK;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  wino4_pack_kernel<<<blocks, 256, 0, s>>>(w, scale, (__bf16*)wpk, Cout, Cin, units, nchunk, wino4_split(prec));
  CRESTE_CHECK_LAUNCH("wino4_pack");
  return CRESTE_OK;
}

// floats between the product planes of two positions (padding them apart by 4 KiB ... 1 MiB changed the output transform by
// < 3 %: the 36 streams do not alias in HBM)
static inline long wino4_mplane(long T, int Cout) { return T * Cout; }

// the larger of the two V forms (pre-split pieces; fp32 is 2/3 or 1/1 of it) so that either fits the workspace
static inline long wino4_v_bytes(long T, int Cin, int prec) {
  const long m_blocks = (T + W4_M - 1) / W4_M, nchunk = (Cin + W4_CK - 1) / W4_CK;
  const int sp = wino4_split(prec);
  return (long)W4_POS * m_blocks * nchunk * (sp > 2 ? sp : 2) * 2 * W4_M * 16;
}

int64_t conv_wino4_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int prec) {
  const long T = wino4_tiles(N, Ho, Wo);
  return wino4_v_bytes(T, Cin, prec) + (long)W4_POS * wino4_mplane(T, Cout) * 4 + 4096;       // V, M, the junk line of the padded stores
}

// measurement aid (bench.py): HIP events around the GEMM kernel of the next creste_conv2d_nhwc(CRESTE_ALGO_WINOGRAD4) call
static std::atomic<bool> g_w4_probe{false};
static hipEvent_t g_w4_ev[2] = {nullptr, nullptr};

template <int SPLIT, int TN, bool AF32 = false>
static int launch_wino4_gemm(const Wino4GemmArgs& a, hipStream_t s) {
  constexpr int smem = 3 * ((AF32 ? 4 : SPLIT * 2) * W4_M * 16 + TN * SPLIT * 2 * 64 * 16);
  static_assert(smem <= 160 * 1024, "Winograd GEMM stages do not fit the LDS");
  static std::atomic<uint64_t> attr_devs{0};
  CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wino4_gemm_kernel<SPLIT, TN, AF32>), smem, attr_devs));
  int dev = 0, cus = 0;
  CRESTE_HIP(hipGetDevice(&dev));
  CRESTE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const long items = (long)a.m_blocks * a.npos * a.tiles_n;
  long per_xcd = cus / 8 > 0 ? cus / 8 : 1;
  const long need = (items + 7) / 8;
  if (per_xcd > need) per_xcd = need;
  wino4_gemm_kernel<SPLIT, TN, AF32><<<(unsigned)(per_xcd * 8), 512, smem, s>>>(a);
  CRESTE_CHECK_LAUNCH("wino4_gemm");
  return CRESTE_OK;
}

// experiment knobs (scripts/wino_overlap_micro.py, profiles/r04_pipeline_notes.md): CRESTE_W4_GEMM_WGS = persistent GEMM
// workgroups per XCD (default: one per CU), CRESTE_W4_CHAIN = 1 runs the GEMM kernels of ALL streams of a device in host
// issue order (one event per device)
static int w4_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static hipEvent_t g_w4_chain_ev[16] = {};
static int w4_chain_wait(hipStream_t s) {
  if (!w4_env_int("CRESTE_W4_CHAIN", 0)) return CRESTE_OK;
  int dev = 0;
  CRESTE_HIP(hipGetDevice(&dev));
  if (g_w4_chain_ev[dev & 15]) CRESTE_HIP(hipStreamWaitEvent(s, g_w4_chain_ev[dev & 15], 0));
  return CRESTE_OK;
}
static int w4_chain_record(hipStream_t s) {
  if (!w4_env_int("CRESTE_W4_CHAIN", 0)) return CRESTE_OK;
  int dev = 0;
  CRESTE_HIP(hipGetDevice(&dev));
  if (!g_w4_chain_ev[dev & 15]) CRESTE_HIP(hipEventCreateWithFlags(&g_w4_chain_ev[dev & 15], hipEventDisableTiming));
  CRESTE_HIP(hipEventRecord(g_w4_chain_ev[dev & 15], s));
  return CRESTE_OK;
}

template <int SPLIT, int TN>
static int launch_wino4_gemm32(const Wino4GemmArgs& a, hipStream_t s) {
  constexpr int smem = 3 * (4 * W4_M * 16 + TN * SPLIT * 2 * 64 * 16);
  static_assert(smem <= 160 * 1024, "Winograd GEMM stages do not fit the LDS");
  static std::atomic<uint64_t> attr_devs{0};
  CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wino4_gemm32_kernel<SPLIT, TN>), smem, attr_devs));
  int dev = 0, cus = 0;
  CRESTE_HIP(hipGetDevice(&dev));
  CRESTE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const long items = (long)a.m_blocks * a.npos * a.tiles_n;
  long per_xcd = cus / 8 > 0 ? cus / 8 : 1;
  const int lim = w4_env_int("CRESTE_W4_GEMM_WGS", 0);
  if (lim > 0 && lim < per_xcd) per_xcd = lim;
  const long need = (items + 7) / 8;
  if (per_xcd > need) per_xcd = need;
  { const int rc = w4_chain_wait(s); if (rc != CRESTE_OK) return rc; }
#ifdef CRESTE_W4_EXPERIMENTS
  // experiments on the row-split layout: CRESTE_W4_EXP = 1 no split arithmetic, 2 no vmcnt waits, 3 no barrier, 4 no LDS-DMA
  // (1-4: timing only, WRONG results), 5 = the prefetch's LDS-DMA pieces issued between the MFMA groups (correct results)
  if (const int ex = w4_env_int("CRESTE_W4_EXP", 0)) {
    static std::atomic<uint64_t> ad1{0}, ad2{0}, ad3{0}, ad4{0}, ad5{0};
    const unsigned grid = (unsigned)(per_xcd * 8);
    if (ex == 1) { CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wino4_gemm32_kernel<SPLIT, TN, 1, true>), smem, ad1)); wino4_gemm32_kernel<SPLIT, TN, 1, true><<<grid, 512, smem, s>>>(a); }
    if (ex == 2) { CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wino4_gemm32_kernel<SPLIT, TN, 2, true>), smem, ad2)); wino4_gemm32_kernel<SPLIT, TN, 2, true><<<grid, 512, smem, s>>>(a); }
    if (ex == 3) { CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wino4_gemm32_kernel<SPLIT, TN, 3, true>), smem, ad3)); wino4_gemm32_kernel<SPLIT, TN, 3, true><<<grid, 512, smem, s>>>(a); }
    if (ex == 4) { CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wino4_gemm32_kernel<SPLIT, TN, 4, true>), smem, ad4)); wino4_gemm32_kernel<SPLIT, TN, 4, true><<<grid, 512, smem, s>>>(a); }
    if (ex == 5) { CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wino4_gemm32_kernel<SPLIT, TN, 5, true>), smem, ad5)); wino4_gemm32_kernel<SPLIT, TN, 5, true><<<grid, 512, smem, s>>>(a); }
    CRESTE_CHECK_LAUNCH("wino4_gemm32 (experiment)");
    return CRESTE_OK;
  }
#endif
  if (w4_env_int("CRESTE_W4_RS", 1)) {      // the row-split wave layout (default; 0: round 4's 4 x 2 layout, kept as the bit-exact twin)
    static std::atomic<uint64_t> attr_rs{0};
    CRESTE_HIP(ensure_dyn_smem(reinterpret_cast<const void*>(wino4_gemm32_kernel<SPLIT, TN, 0, true>), smem, attr_rs));
    wino4_gemm32_kernel<SPLIT, TN, 0, true><<<(unsigned)(per_xcd * 8), 512, smem, s>>>(a);
  } else {
    wino4_gemm32_kernel<SPLIT, TN><<<(unsigned)(per_xcd * 8), 512, smem, s>>>(a);
  }
  CRESTE_CHECK_LAUNCH("wino4_gemm32");
  return w4_chain_record(s);
}

int conv_wino4_stat_rows(const creste_conv_desc* d) {
  if (!conv_wino4_supported(d->prec, d->KH, d->KW, d->stride, d->Cin, d->Cout) || d->res || d->row_mask ||
      (d->flags & CRESTE_CONV_EMIT_NEXT_V)) return -1;
  return (int)((wino4_tiles(d->N, d->Ho, d->Wo) + W4O_TILES - 1) / W4O_TILES);
}

int conv_wino4_run(const creste_conv_desc* d, hipStream_t s) {
  CRESTE_REQUIRE(conv_wino4_supported(d->prec, d->KH, d->KW, d->stride, d->Cin, d->Cout),
                 "conv2d: the F(4x4,3x3) path is built for stride-1 3x3 convs in the bf16 split modes, Cout a multiple of 4");
  CRESTE_REQUIRE(d->work && !d->a_scale, "conv2d: the Winograd path needs its workspace and takes no per-sample input gate");
  CRESTE_REQUIRE(!d->up_src || (d->pad_t == 1 && d->pad_l == 1 && d->H == 2 * d->up_H && d->W == 2 * d->up_W),
                 "conv2d: the fused upsample of the F(4x4,3x3) input transform is the exact 2x one under pad 1");
  CRESTE_REQUIRE(!(d->flags & CRESTE_CONV_EMIT_NEXT_V) || (!d->res && !d->row_mask && !d->out_amax && !d->out_stats && d->pad_t == 1 && d->pad_l == 1 &&
                                                           d->Ho == d->H && d->Wo == d->W && wino4_split(d->prec) == 3),
                 "conv2d: EMIT_NEXT_V takes a pad-1 bf16x6 conv without residual / row mask / |max| tracking");
  CRESTE_REQUIRE((d->out_cs & 3) == 0 && (d->out_co & 3) == 0 && (!d->res || (d->res_cs & 3) == 0) &&
                     (reinterpret_cast<uintptr_t>(d->out) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->work) & 15) == 0,
                 "conv2d: the Winograd path needs 16-byte aligned output / residual channel slices and workspace");
  const int split = wino4_split(d->prec);
  const int tiles_y = (d->Ho + 3) / 4, tiles_x = (d->Wo + 3) / 4;
  const long T = (long)d->N * tiles_y * tiles_x;
  CRESTE_REQUIRE(T < (1L << 26) && T * d->Cout < (1L << 36), "conv2d: Winograd workspace too large (split the batch)");
  const int nchunk = (d->Cin + W4_CK - 1) / W4_CK, m_blocks = (int)((T + W4_M - 1) / W4_M);
  char* V = (char*)d->work;
  float* M = (float*)(V + wino4_v_bytes(T, d->Cin, d->prec));

  Wino4InArgs ia;
  ia.in = d->in; ia.V = V; ia.N = d->N; ia.H = d->H; ia.W = d->W; ia.Cin = d->Cin; ia.in_cs = d->in_cs;
  ia.tiles_y = tiles_y; ia.tiles_x = tiles_x; ia.T = (int)T; ia.pad_t = d->pad_t; ia.pad_l = d->pad_l;
  ia.nchunk = nchunk; ia.m_blocks = m_blocks;
  ia.up_src = d->up_src; ia.up_C = d->up_C; ia.up_cs = d->up_cs;
  ia.repl = (d->flags & CRESTE_CONV_REPLICATE_PAD) ? 1 : 0;
  const bool phase2x = (d->flags & CRESTE_CONV_PHASE2X) != 0;
  CRESTE_REQUIRE(!(ia.repl || phase2x) || (f32v_default() && !d->up_src && !(d->flags & (CRESTE_CONV_EMIT_NEXT_V | CRESTE_CONV_V_VALID))),
                 "conv2d: REPLICATE_PAD / PHASE2X are built for the plain fp32-V input transform (no fused upsample, no conv pair)");
  CRESTE_REQUIRE(!phase2x || (d->Cout % 16 == 0 && !d->res && !d->row_mask && !d->out_amax && !d->out_stats && d->Ho == d->H && d->Wo == d->W),
                 "conv2d: PHASE2X scatters 4 x (Cout / 4) channels onto the [N, 2H, 2W] output; no residual / row mask / |max| / statistics");
  CRESTE_REQUIRE((d->Cin & 3) == 0 && (!d->in || ((d->in_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(d->in) & 15) == 0)),
                 "conv2d: the F(4x4,3x3) input transform reads channel pairs (Cin / in_cs multiples of 4, as every NHWC conv here)");
  const char* ord_env = getenv("CRESTE_W4_ORDER");
  const int order = ord_env ? atoi(ord_env) : 2;
  ia.order = order & 1; ia.ncp = (nchunk + 1) / 2;
  const dim3 igrid = ia.order ? dim3((unsigned)(m_blocks * 16 * ia.ncp)) : dim3((unsigned)(m_blocks * 16), (unsigned)((nchunk + 1) / 2));
  const dim3 igrid1 = ia.order ? dim3((unsigned)(m_blocks * 32 * ia.ncp)) : dim3((unsigned)(m_blocks * 32), (unsigned)((nchunk + 1) / 2));
  const char* f32_env = getenv("CRESTE_W4_F32V");
  const bool f32v = f32_env ? atoi(f32_env) != 0 : true;
  CRESTE_REQUIRE(f32v || !(d->flags & CRESTE_CONV_EMIT_NEXT_V),
                 "conv2d: EMIT_NEXT_V writes the next conv's transformed input as fp32; CRESTE_W4_F32V=0 (bf16-piece V) cannot consume it");
  // experiment knob (scripts/coresidency_probe.py): CRESTE_W4_ONLY = bit mask of the kernels of the call that run
  // (1 input transform, 2 GEMM, 4 output transform); results are only meaningful with all three
  const int only = w4_env_int("CRESTE_W4_ONLY", 7);
  if ((d->flags & CRESTE_CONV_V_VALID) || !(only & 1)) {
    // the caller vouches that `work` holds V of this very input (creste_hip.h)
  } else if (f32v) {
    if (d->up_src) wino4_in1_kernel<true><<<igrid1, 256, 0, s>>>(ia);
    else wino4_in1_kernel<false><<<igrid1, 256, 0, s>>>(ia);
  } else if (d->up_src) {
    if (split == 3) wino4_in_kernel<3, true><<<igrid, 256, 0, s>>>(ia);
    else wino4_in_kernel<2, true><<<igrid, 256, 0, s>>>(ia);
  } else if (split == 3) wino4_in_kernel<3, false><<<igrid, 256, 0, s>>>(ia);
  else wino4_in_kernel<2, false><<<igrid, 256, 0, s>>>(ia);
  CRESTE_CHECK_LAUNCH("wino4_in");

  Wino4GemmArgs a;
  a.V = V; a.wpk = (const char*)d->wpk; a.M = M; a.T = (int)T; a.Cout = d->Cout;
  a.nchunk = nchunk; a.m_blocks = m_blocks; a.units = wino4_units(d->Cout); a.npos = W4_POS;
  a.mplane = wino4_mplane(T, d->Cout);
  a.mbg = m_blocks < W4_MBG ? m_blocks : W4_MBG;
  int tn = d->Cout > 128 ? 4 : 2;
  // small maps (the BEV trunk's 256 -> 256 convs at 32 x 32: 2 tile blocks x 36 positions = 72 items of a 256-wide tile on 256
  // CUs): 128-wide tiles double the items; an item is a latency chain of Cin / 16 steps, not MFMA-bound, so the halved item
  // costs nearly its full time and the launch shortens (same K order per output: same bits)
  const int small_items = w4_env_int("CRESTE_W4_SMALL_ITEMS", 512);
  if (tn == 4 && (long)m_blocks * W4_POS * ((d->Cout + 255) / 256) <= small_items) tn = 2;
  a.tiles_n = (d->Cout + 64 * tn - 1) / (64 * tn);
  int rc;
  const bool probe = g_w4_probe.load(std::memory_order_relaxed);
  if (probe) {
    if (!g_w4_ev[0]) { CRESTE_HIP(hipEventCreate(&g_w4_ev[0])); CRESTE_HIP(hipEventCreate(&g_w4_ev[1])); }
    CRESTE_HIP(hipEventRecord(g_w4_ev[0], s));
  }
  const bool stream = f32v && nchunk >= 3 && (f32_env ? atoi(f32_env) != 2 : true);
  if (!(only & 2)) rc = CRESTE_OK;
  else if (stream) {
    if (tn == 4) rc = split == 3 ? launch_wino4_gemm32<3, 4>(a, s) : launch_wino4_gemm32<2, 4>(a, s);
    else rc = split == 3 ? launch_wino4_gemm32<3, 2>(a, s) : launch_wino4_gemm32<2, 2>(a, s);
  } else if (f32v) {
    if (tn == 4) rc = split == 3 ? launch_wino4_gemm<3, 4, true>(a, s) : launch_wino4_gemm<2, 4, true>(a, s);
    else rc = split == 3 ? launch_wino4_gemm<3, 2, true>(a, s) : launch_wino4_gemm<2, 2, true>(a, s);
  } else if (tn == 4) rc = split == 3 ? launch_wino4_gemm<3, 4>(a, s) : launch_wino4_gemm<2, 4>(a, s);
  else rc = split == 3 ? launch_wino4_gemm<3, 2>(a, s) : launch_wino4_gemm<2, 2>(a, s);
  if (rc != CRESTE_OK) return rc;
  if (probe) CRESTE_HIP(hipEventRecord(g_w4_ev[1], s));

  if (d->flags & CRESTE_CONV_EMIT_NEXT_V) {
    // `out` is the NEXT conv's workspace: its transformed input V2 is written instead of this conv's output (creste_hip.h)
    Wino4OutInArgs f;
    f.M = M; f.bias = d->bias; f.V = reinterpret_cast<char*>(d->out);
    f.Ho = d->Ho; f.Wo = d->Wo; f.Cout = d->Cout; f.act = d->act;
    f.tiles_y = tiles_y; f.tiles_x = tiles_x; f.T = (int)T; f.mplane = a.mplane;
    f.nchunk = (d->Cout + W4_CK - 1) / W4_CK; f.m_blocks = m_blocks;
    f.by = (tiles_y + OI_BH - 1) / OI_BH; f.bx = (tiles_x + OI_BW - 1) / OI_BW;
    f.nblk = d->N * f.by * f.bx; f.q8 = (f.nblk + 7) / 8;
    const long grid = 8L * f.q8 * (f.nchunk * 4);
    CRESTE_REQUIRE(grid < (1L << 31), "conv2d: fused output -> input transform grid too large");
    if (only & 4) wino4_outin_kernel<<<(unsigned)grid, 256, 0, s>>>(f);
    CRESTE_CHECK_LAUNCH("wino4_outin");
    return CRESTE_OK;
  }
  Wino4OutArgs o;
  o.M = M; o.bias = d->bias; o.res = d->res; o.row_mask = d->row_mask; o.out = d->out; o.out_amax = d->out_amax;
  o.N = d->N; o.Ho = d->Ho; o.Wo = d->Wo; o.Cout = d->Cout; o.out_cs = d->out_cs; o.out_co = d->out_co; o.res_cs = d->res_cs;
  o.act = d->act; o.tiles_y = tiles_y; o.tiles_x = tiles_x; o.T = (int)T; o.mplane = a.mplane;
  o.stats = d->out_stats;
  o.phaseC = phase2x ? d->Cout / 4 : 0;
  CRESTE_REQUIRE(!d->out_stats || (!d->res && !d->row_mask), "conv2d: out_stats takes a conv without residual / row mask");
  o.order = (order >> 1) & 1; o.ncg = (d->Cout / 4 + W4O2_QUADS - 1) / W4O2_QUADS;
  o.ntg8 = (int)(((T + W4O_TILES - 1) / W4O_TILES + 7) / 8);
  const dim3 ogrid = o.order ? dim3((unsigned)(o.ntg8 * 8 * o.ncg))
                             : dim3((unsigned)((T + W4O_TILES - 1) / W4O_TILES), (unsigned)o.ncg);
  if (only & 4) wino4_out2_kernel<<<ogrid, 256, 0, s>>>(o);
  CRESTE_CHECK_LAUNCH("wino4_out");
  return CRESTE_OK;
}


// ------------------------------------------------------------------------------------------------ weight gradient
// dW of the same convs through the same transform: with V = B^T d B (input windows) and Yh = A dY A^T (4x4 tiles of the
// output gradient, A = (A^T)^T: 6x4), dU_p[co, ci] = sum over tiles of Yh_p[tile, co] V_p[tile, ci] per position p, and
// dW = G^T dU G.  36 GEMMs with K = TILES: both operands are written K-major ([row][8 tiles] 16-byte units, the
// forward GEMM's operand images with tiles in the place of channels), the tile range is cut into SEG segments that
// become extra "positions" (36 x SEG independent GEMMs of M = Cin rows, N = Cout columns), and the forward's
// wino4_gemm_kernel runs unchanged.  4x fewer matrix products than the direct weight gradient, same bf16x6 grade.
struct Wino4WgInArgs {
  const float* src;          // x [N,H,W,cs] (window 6x6 at 4t - pad) or gy [N,H,W,cs] (4x4 tile at 4t)
  char* dst;                 // A image (rows = channels of x) or B image (64-channel units of gy)
  int N, H, W, C, cs;
  int tiles_y, tiles_x, T;
  int pad_t, pad_l;
  int nchunk;                // 16-tile chunks per segment
  int blocks;                // A image: 256-row blocks;  B image: 64-row units (padded)
};

// one column of A applied to four values: (A v)[0..5]
__device__ __forceinline__ void w4_a(const float (&v)[4], float (&t)[6]) {
  const float s01 = v[1] + v[2], d01 = v[1] - v[2];
  t[0] = v[0];
  t[1] = (v[0] + s01) + v[3];
  t[2] = (v[0] - v[1]) + (v[2] - v[3]);
  t[3] = (v[0] + 2.f * v[1]) + (4.f * v[2] + 8.f * v[3]);
  t[4] = (v[0] - 2.f * v[1]) + (4.f * v[2] - 8.f * v[3]);
  t[5] = v[3];
  (void)s01; (void)d01;
}

// Workgroup = 16 consecutive tiles (= one K chunk) x 32 channels; thread = (tile PAIR, channel): scalar loads with the
// lanes along channels (128 contiguous bytes per pixel), two tiles transformed in registers, their bf16 pieces packed
// into one dword = K elements 2q, 2q + 1 of the channel's 16-byte unit.  GY: the operand is the output gradient.
template <int SPLIT, bool GY>
__global__ __launch_bounds__(256) void wino4_wg_in_kernel(const Wino4WgInArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned tbuf[W4_POS * 2 * 32 * 4];      // [pos][octet][channel][4 tile pairs]
  const int t = threadIdx.x;
  const int chunk_lin = blockIdx.x;            // global 16-tile chunk: segment = chunk_lin / nchunk
  const int c = t & 31, tp = t >> 5;           // channel within the group, tile pair 0..7
  const int ch = blockIdx.y * 32 + c;
  const int per = p.tiles_y * p.tiles_x;
  float u[2][6][6];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tile = chunk_lin * 16 + tp * 2 + h;
    const bool ok = tile < p.T && ch < p.C;
    const int tcl = ok ? tile : 0;
    const int img = tcl / per, rem = tcl - img * per;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const float* base = p.src + (ok ? ch : 0);
    if (GY) {
      float g[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int yy = 4 * ty + i;
        const bool yok = ok && yy < p.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int xx = 4 * tx + j;
          const bool in = yok && xx < p.W;
          const float v = base[(((size_t)img * p.H + (in ? yy : 0)) * p.W + (in ? xx : 0)) * p.cs];
          g[i][j] = in ? v : 0.f;
        }
      }
      float m[6][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float col[4] = {g[0][j], g[1][j], g[2][j], g[3][j]};
        float o[6];
        w4_a(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i][j] = o[i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float o[6];
        w4_a(m[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) u[h][i][j] = o[j];
      }
    } else {
      float d[6][6];
      const int y0 = 4 * ty - p.pad_t, x0 = 4 * tx - p.pad_l;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int yy = y0 + i;
        const bool yok = ok && (unsigned)yy < (unsigned)p.H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int xx = x0 + j;
          const bool in = yok && (unsigned)xx < (unsigned)p.W;
          const float v = base[(((size_t)img * p.H + (in ? yy : 0)) * p.W + (in ? xx : 0)) * p.cs];
          d[i][j] = in ? v : 0.f;
        }
      }
      float m[6][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float o[6];
        w4_bt1(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i][j] = o[i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float o[6];
        w4_bt1(m[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) u[h][i][j] = o[j];
      }
    }
  }
  const int seg = chunk_lin / p.nchunk, chunk = chunk_lin - seg * p.nchunk;
  constexpr int UNITS = W4_POS * 2 * 32;       // 16-byte units of one piece: (position, octet, channel)
#pragma unroll
  for (int pl = 0; pl < SPLIT; ++pl) {
    if (pl) __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const w4f32x2 pair{u[0][i][j], u[1][i][j]};
        const w4bf16x2 piece = __builtin_convertvector(pair, w4bf16x2);
        tbuf[(((i * 6 + j) * 2 + (tp >> 2)) * 32 + c) * 4 + (tp & 3)] = __builtin_bit_cast(unsigned, piece);
        if (pl + 1 < SPLIT) {
          const w4f32x2 back = __builtin_convertvector(piece, w4f32x2);
          u[0][i][j] -= back[0]; u[1][i][j] -= back[1];
        }
      }
    __syncthreads();
    for (int uidx = t; uidx < UNITS; uidx += 256) {
      const int cw = uidx & 31, oct = (uidx >> 5) & 1, pos = uidx >> 6;
      const int chw = blockIdx.y * 32 + cw;
      const size_t posp = (size_t)seg * W4_POS + pos;
      char* dst;
      if (GY) {      // B image: [pos'][64-channel unit][chunk][piece][octet][64][8]
        if ((chw >> 6) >= p.blocks) continue;
        dst = p.dst + ((((posp * p.blocks + (chw >> 6)) * p.nchunk + chunk) * SPLIT + pl) * 2 + oct) * (size_t)(64 * 16) + (size_t)(chw & 63) * 16;
      } else {       // A image: [pos'][256-channel block][chunk][piece][octet][256][8]
        if ((chw >> 8) >= p.blocks) continue;
        dst = p.dst + ((((posp * p.blocks + (chw >> 8)) * p.nchunk + chunk) * SPLIT + pl) * 2 + oct) * (size_t)(W4_M * 16) + (size_t)(chw & 255) * 16;
      }
      *reinterpret_cast<w4f32x4*>(dst) = *reinterpret_cast<const w4f32x4*>(tbuf + (size_t)uidx * 4);
    }
  }
}

// gw[co][ci][3][3] (+)= G^T (sum over segments of dU) G; thread = (ci, cout quad); M = [seg * 36 + pos][Cout / 4][Cin][4]
__global__ __launch_bounds__(256) void wino4_wg_out_kernel(const float* __restrict__ M, float* __restrict__ gw, int Cin, int Cout,
                                                          int nseg, int accumulate) {
  const int Q = Cout >> 2;
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= (long)Q * Cin) return;
  const int ci = (int)(idx % Cin), q = (int)(idx / Cin);
  const size_t plane = (size_t)Q * Cin * 4;
  const float* src = M + ((size_t)q * Cin + ci) * 4;
  w4f32x4 dU[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      w4f32x4 acc = *reinterpret_cast<const w4f32x4*>(src + (size_t)(i * 6 + j) * plane);
      for (int sgi = 1; sgi < nseg; ++sgi) acc += *reinterpret_cast<const w4f32x4*>(src + ((size_t)sgi * W4_POS + i * 6 + j) * plane);
      dU[i][j] = acc;
    }
  // G^T (3x6): rows [1/4 -1/6 -1/6 1/24 1/24 0; 0 -1/6 1/6 1/12 -1/12 0; 0 -1/6 -1/6 1/6 1/6 1]
  auto gt = [](const w4f32x4 (&v)[6], w4f32x4 (&o)[3]) __attribute__((always_inline)) {
    const w4f32x4 s12 = v[1] + v[2], d21 = v[2] - v[1], s34 = v[3] + v[4], d34 = v[3] - v[4];
    o[0] = (0.25f * v[0] - (1.f / 6.f) * s12) + (1.f / 24.f) * s34;
    o[1] = (1.f / 6.f) * d21 + (1.f / 12.f) * d34;
    o[2] = ((1.f / 6.f) * s34 - (1.f / 6.f) * s12) + v[5];
  };
  w4f32x4 r[3][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const w4f32x4 col[6] = {dU[0][j], dU[1][j], dU[2][j], dU[3][j], dU[4][j], dU[5][j]};
    w4f32x4 o[3];
    gt(col, o);
#pragma unroll
    for (int a = 0; a < 3; ++a) r[a][j] = o[a];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    w4f32x4 o[3];
    gt(r[a], o);
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* dstp = gw + (((size_t)(q * 4 + e) * Cin + ci) * 3 + a) * 3 + b;
        *dstp = accumulate ? *dstp + o[b][e] : o[b][e];
      }
  }
}

static inline int wino4_wg_segments(long T) {
  // enough items for four rounds of the chip on the widest layers (2 x 2 tiles x 36 x SEG), K of >= 64 chunks per segment
  long chunks = (T + 15) / 16;
  int seg = 7;
  while (seg > 1 && chunks / seg < 64) --seg;
  return seg;
}

bool conv_wgrad_wino4_supported(int K, int stride, int H, int W, int Ho, int Wo, int Cin, int Cout) {
  return K == 3 && stride == 1 && Ho == H && Wo == W && Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 128 && Cout >= 128;
}

int64_t conv_wgrad_wino4_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  const long T = wino4_tiles(N, H, W);
  const int seg = wino4_wg_segments(T);
  const long nchunk = ((T + seg - 1) / seg + 15) / 16;
  const long mbl = (Cin + W4_M - 1) / W4_M, units = wino4_units(Cout);
  const long a_bytes = (long)W4_POS * seg * mbl * nchunk * 3 * 2 * W4_M * 16;
  const long b_bytes = (long)W4_POS * seg * units * nchunk * 3 * 2 * 64 * 16;
  return a_bytes + b_bytes + (long)W4_POS * seg * Cout * Cin * 4 + 4096;
}

int conv_wgrad_wino4_run(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, int N, int H, int W, int Cin, int Cout,
                         int pad_t, int pad_l, int accumulate, void* work, hipStream_t s) {
  const int tiles_y = (H + 3) / 4, tiles_x = (W + 3) / 4;
  const long T = (long)N * tiles_y * tiles_x;
  CRESTE_REQUIRE(T < (1L << 26), "conv_wgrad_wino4: too many tiles");
  const int seg = wino4_wg_segments(T);
  const int nchunk = (int)(((T + seg - 1) / seg + 15) / 16);
  const int mbl = (Cin + W4_M - 1) / W4_M, units = wino4_units(Cout);
  char* Aimg = (char*)work;
  char* Bimg = Aimg + (long)W4_POS * seg * mbl * nchunk * 3 * 2 * W4_M * 16;
  float* M = (float*)(Bimg + (long)W4_POS * seg * units * nchunk * 3 * 2 * 64 * 16);
  Wino4WgInArgs ia;
  ia.N = N; ia.H = H; ia.W = W; ia.tiles_y = tiles_y; ia.tiles_x = tiles_x; ia.T = (int)T; ia.pad_t = pad_t; ia.pad_l = pad_l;
  ia.nchunk = nchunk;
  ia.src = x; ia.dst = Aimg; ia.C = Cin; ia.cs = x_cs; ia.blocks = mbl;
  wino4_wg_in_kernel<3, false><<<dim3((unsigned)(seg * nchunk), (unsigned)((Cin + 31) / 32)), 256, 0, s>>>(ia);
  CRESTE_CHECK_LAUNCH("wino4_wg_in(x)");
  ia.src = gy; ia.dst = Bimg; ia.C = Cout; ia.cs = gy_cs; ia.blocks = units;
  wino4_wg_in_kernel<3, true><<<dim3((unsigned)(seg * nchunk), (unsigned)((units * 64 + 31) / 32)), 256, 0, s>>>(ia);
  CRESTE_CHECK_LAUNCH("wino4_wg_in(gy)");
  Wino4GemmArgs a;
  a.V = Aimg; a.wpk = Bimg; a.M = M; a.T = Cin; a.Cout = Cout; a.nchunk = nchunk; a.m_blocks = mbl; a.units = units;
  a.npos = W4_POS * seg; a.mplane = (long)Cout * Cin;
  a.mbg = mbl < W4_MBG ? mbl : W4_MBG;
  const int tn = Cout > 128 ? 4 : 2;
  a.tiles_n = (Cout + 64 * tn - 1) / (64 * tn);
  const int rc = tn == 4 ? launch_wino4_gemm<3, 4>(a, s) : launch_wino4_gemm<3, 2>(a, s);
  if (rc != CRESTE_OK) return rc;
  wino4_wg_out_kernel<<<(unsigned)(((long)(Cout / 4) * Cin + 255) / 256), 256, 0, s>>>(M, gw, Cin, Cout, seg, accumulate);
  CRESTE_CHECK_LAUNCH("wino4_wg_out");
  return CRESTE_OK;
}


// ------------------------------------------------------------------------------------------------ upsample -> conv3x3 ring
// `Upsample(x2, bilinear, align_corners=False) -> conv3x3(pad 1)` (reference DeconvHead.up2, inpainting.py:56-60) runs as four
// phase convolutions on the LOW-resolution map: high-resolution row 2y + a only ever sees low-resolution rows y - 1 .. y + 1, so
// the conv over the upsampled image is a 3x3 conv with 4 x Cout composed kernels on the small map (same products, a transformed
// input a quarter the size, no upsampled tensor) -- exact everywhere except at the image border: the high-resolution conv pads the
// UPSAMPLED image with zeros, the phase form continues it with the replicate-padded interpolation.  This kernel takes those
// taps back out of the outermost ring of the output: for ring pixel (oy, ox) and every tap (ky, kx) whose source
// (oy + ky - 1, ox + kx - 1) lies outside [0, 2H) x [0, 2W):  out -= sum_ci w[co, ci, ky, kx] * u~(source), u~ = the bilinear
// formula on the replicate-padded map (what the phase form used), then the activation the output transform left out there.
// One workgroup = 32 consecutive pixels of one side of one image x 128 couts (wave = 32 couts), K = taps x Cin in chunks of 64
// on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32: the term cancels a term of the same size, so fp32 products).
// Corners belong to the top / bottom sides (five taps there), the left / right sides run over rows 1 .. 2H - 2.
// Measured (scripts/upconv_micro.py, 256 -> 128 at 256 x 256, batch 16): 77-83 us; by ablation 25 us is the empty skeleton, ~17 us the
// loads, ~40 us the 384 dependent fp32 MFMAs per wave (20 us if the 512 workgroups were spread evenly over the SIMDs).
struct RingFixArgs {
  const float* x;            // [N, H, W, x_cs] low-resolution input (channel offset applied)
  const float* w;            // [3][3][Cin][Cout] fp32, BatchNorm scale folded
  float* out;                // [N, 2H, 2W, out_cs] at channel offset out_co
  int N, H, W, Cin, x_cs, Cout, out_cs, out_co, act;
  int cw, ch;                // 32-pixel chunks per horizontal / vertical side
};
constexpr int RF_KC = 64, RF_LD = RF_KC / 2 + 4;      // A tile rows: [k parity][pixel][32 values (+4)]
__global__ __launch_bounds__(256) void upconv2x_ring_fix_kernel(const RingFixArgs p) {
  // The A tile [32 pixels][64 channels (+1)] crosses LDS (two stages: it is formed by (pixel, channel octet) threads and consumed as
  // MFMA rows); the weight operand does not: lane (k half, cout) of an MFMA reads w[k][cout] -- 2 x 128 contiguous bytes per
  // wave and instruction, straight from L2 (all workgroups share the 1.2 MB of weights) into registers, one chunk ahead of the
  // matrix instructions that consume it.  16.6 KB of LDS: several workgroups per CU hide each other's load latency.
  __shared__ __attribute__((aligned(16))) float As[2][2 * 32 * RF_LD];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int H2 = 2 * p.H, W2 = 2 * p.W;
  const int per_img = 2 * p.cw + 2 * p.ch;
  const int img = blockIdx.x / per_img;
  int r = blockIdx.x - img * per_img, side;           // side 0 top, 1 bottom, 2 left, 3 right
  if (r < 2 * p.cw) { side = r / p.cw; r -= side * p.cw; } else { r -= 2 * p.cw; side = 2 + r / p.ch; r -= (side - 2) * p.ch; }
  const int s0 = r * 32, len = side < 2 ? W2 : H2 - 2;
  auto pix = [&](int j, int& oy, int& ox) __attribute__((always_inline)) -> bool {
    const int sidx = s0 + j;
    if (side == 0) { oy = 0; ox = sidx; }
    else if (side == 1) { oy = H2 - 1; ox = sidx; }
    else if (side == 2) { oy = sidx + 1; ox = 0; }
    else { oy = sidx + 1; ox = W2 - 1; }
    return sidx < len;
  };
  // taps any pixel of this workgroup needs (wave-uniform: every wave evaluates the same 32 pixels)
  unsigned need = 0;
  {
    int oy, ox;
    const bool ok = pix(lane & 31, oy, ox);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int Y = oy + tap / 3 - 1, X = ox + tap % 3 - 1;
      const bool outside = ok && ((unsigned)Y >= (unsigned)H2 || (unsigned)X >= (unsigned)W2);
      if (__ballot(outside) != 0ull) need |= 1u << tap;
    }
  }
  if (need == 0) return;
  const int co = blockIdx.y * 128 + wv * 32 + (lane & 31);
  const bool cok = co < p.Cout;
  const int cw = cok ? co : 0;
  // this lane's 16 output addresses: read now, so that the values are there when the matrix work ends
  float* dst[16];
  float old[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int j = 8 * (rr >> 2) + 4 * (lane >> 5) + (rr & 3);
    int oy, ox;
    const bool ok = pix(j, oy, ox) && cok;
    dst[rr] = ok ? p.out + (((size_t)img * H2 + oy) * W2 + ox) * p.out_cs + p.out_co + co : nullptr;
    old[rr] = ok ? *dst[rr] : 0.f;
  }
  w4f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // A-tile thread = (pixel, 8 channels of the 64-channel chunk)
  const int apx = t >> 3, ac8 = (t & 7) * 8;
  int aoy, aox;
  const bool aok = pix(apx, aoy, aox);
  const float* ximg = p.x + (size_t)img * p.H * p.W * p.x_cs;
  w4f32x4 ra[2];
  float wn[RF_KC / 2], wc[RF_KC / 2];
  // chunk (tap, c0) -> registers: this thread's part of the A tile and this lane's weight operands of the chunk's 32 MFMAs
  auto fetch = [&](int tap, int c0) __attribute__((always_inline)) {
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int Y = aoy + ky - 1, X = aox + kx - 1;
    const bool outside = aok && ((unsigned)Y >= (unsigned)H2 || (unsigned)X >= (unsigned)W2);
    ra[0] = ra[1] = w4f32x4{0.f, 0.f, 0.f, 0.f};
    if (outside) {
      // u~(Y, X): source coordinate 0.5 (Y + 0.5) - 0.5 NOT clamped at 0 (so that row -1 = 0.75 x~[-1] + 0.25 x[0]), indices clamped
      const float sy = 0.5f * ((float)Y + 0.5f) - 0.5f, sx = 0.5f * ((float)X + 0.5f) - 0.5f;
      const float fy0 = floorf(sy), fx0 = floorf(sx);
      const float wy1 = sy - fy0, wx1 = sx - fx0, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
      const int yi = (int)fy0, xi = (int)fx0;
      const int ya = yi < 0 ? 0 : (yi > p.H - 1 ? p.H - 1 : yi), yb = yi + 1 < 0 ? 0 : (yi + 1 > p.H - 1 ? p.H - 1 : yi + 1);
      const int xa = xi < 0 ? 0 : (xi > p.W - 1 ? p.W - 1 : xi), xb = xi + 1 < 0 ? 0 : (xi + 1 > p.W - 1 ? p.W - 1 : xi + 1);
      const float* paa = ximg + ((size_t)ya * p.W + xa) * p.x_cs + c0 + ac8;
      const float* pab = ximg + ((size_t)ya * p.W + xb) * p.x_cs + c0 + ac8;
      const float* pba = ximg + ((size_t)yb * p.W + xa) * p.x_cs + c0 + ac8;
      const float* pbb = ximg + ((size_t)yb * p.W + xb) * p.x_cs + c0 + ac8;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        ra[h] = wy0 * (wx0 * *reinterpret_cast<const w4f32x4*>(paa + 4 * h) + wx1 * *reinterpret_cast<const w4f32x4*>(pab + 4 * h)) +
                wy1 * (wx0 * *reinterpret_cast<const w4f32x4*>(pba + 4 * h) + wx1 * *reinterpret_cast<const w4f32x4*>(pbb + 4 * h));
    }
    const float* wt = p.w + ((size_t)tap * p.Cin + c0 + (lane >> 5)) * p.Cout + cw;
#pragma unroll
    for (int s2 = 0; s2 < RF_KC / 2; ++s2) wn[s2] = wt[(size_t)(2 * s2) * p.Cout];
  };
  auto stash = [&](int st) __attribute__((always_inline)) {
    // channel k of the chunk -> [k & 1][pixel][k >> 1]: a lane's 32 MFMA operands (k = 2 s + its k half) are contiguous
    *reinterpret_cast<w4f32x4*>(&As[st][(0 * 32 + apx) * RF_LD + (ac8 >> 1)]) = w4f32x4{ra[0][0], ra[0][2], ra[1][0], ra[1][2]};
    *reinterpret_cast<w4f32x4*>(&As[st][(1 * 32 + apx) * RF_LD + (ac8 >> 1)]) = w4f32x4{ra[0][1], ra[0][3], ra[1][1], ra[1][3]};
  };
  int tap = __builtin_ctz(need), c0 = 0, st = 0;
  fetch(tap, 0);
  stash(0);
  __syncthreads();
  for (;;) {
#pragma unroll
    for (int s2 = 0; s2 < RF_KC / 2; ++s2) wc[s2] = wn[s2];
    int ntap = tap, nc0 = c0 + RF_KC;
    if (nc0 >= p.Cin) {
      nc0 = 0;
      const unsigned rest = need >> (tap + 1);
      ntap = rest ? tap + 1 + __builtin_ctz(rest) : 9;
    }
    if (ntap < 9) fetch(ntap, nc0);
    w4f32x4 av[RF_KC / 8];
#pragma unroll
    for (int q = 0; q < RF_KC / 8; ++q) av[q] = *reinterpret_cast<const w4f32x4*>(&As[st][((lane >> 5) * 32 + (lane & 31)) * RF_LD + 4 * q]);
#pragma unroll
    for (int s2 = 0; s2 < RF_KC / 2; ++s2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2 >> 2][s2 & 3], wc[s2], acc, 0, 0, 0);
    if (ntap >= 9) break;
    stash(st ^ 1);           // (stage st ^ 1 was last read one chunk ago: everybody passed the barrier behind it)
    __syncthreads();
    st ^= 1; tap = ntap; c0 = nc0;
  }
#pragma unroll
  for (int rr = 0; rr < 16; ++rr)
    if (dst[rr]) *dst[rr] = act_apply(old[rr] - acc[rr], p.act);
}

int upconv2x_ring_fix_run(const float* x, int x_cs, int N, int H, int W, int Cin, const float* w_ring, int Cout, int act, float* out,
                          int out_cs, int out_co, hipStream_t s) {
  RingFixArgs a;
  a.x = x; a.w = w_ring; a.out = out; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.x_cs = x_cs; a.Cout = Cout; a.out_cs = out_cs;
  a.out_co = out_co; a.act = act;
  a.cw = (2 * W + 31) / 32; a.ch = (2 * H - 2 + 31) / 32;
  const dim3 grid((unsigned)(N * (2 * a.cw + 2 * a.ch)), (unsigned)((Cout + 127) / 128));
  upconv2x_ring_fix_kernel<<<grid, 256, 0, s>>>(a);
  CRESTE_CHECK_LAUNCH("upconv2x_ring_fix");
  return CRESTE_OK;
}

}  // namespace creste

extern "C" int creste_conv_wgrad_wino4_supported(int K, int stride, int H, int W, int Ho, int Wo, int Cin, int Cout) {
  return creste::conv_wgrad_wino4_supported(K, stride, H, W, Ho, Wo, Cin, Cout) ? 1 : 0;
}

extern "C" int64_t creste_conv_wgrad_wino4_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1;
  return creste::conv_wgrad_wino4_workspace_bytes(N, H, W, Cin, Cout);
}

extern "C" int creste_conv_wgrad_wino4(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, int N, int H, int W,
                                       int Cin, int Cout, int pad_t, int pad_l, int accumulate, void* work, void* stream) {
  using namespace creste;
  CRESTE_REQUIRE(x && gy && gw && work, "conv_wgrad_wino4: null pointer");
  CRESTE_REQUIRE(N > 0 && conv_wgrad_wino4_supported(3, 1, H, W, H, W, Cin, Cout),
                 "conv_wgrad_wino4: built for stride-1 same-size 3x3 convs with >= 128 channels (multiples of 4) on both sides");
  CRESTE_REQUIRE(x_cs >= Cin && gy_cs >= Cout && (reinterpret_cast<uintptr_t>(work) & 15) == 0, "conv_wgrad_wino4: bad strides / workspace alignment");
  return conv_wgrad_wino4_run(x, x_cs, gy, gy_cs, gw, N, H, W, Cin, Cout, pad_t, pad_l, accumulate, work, (hipStream_t)stream);
}

extern "C" int creste_upconv2x_ring_fix_f32(const float* x, int x_cs, int N, int H, int W, int Cin, const float* w_ring, int Cout,
                                            int act, float* out, int out_cs, int out_co, void* stream) {
  using namespace creste;
  CRESTE_REQUIRE(x && w_ring && out, "upconv2x_ring_fix: null pointer");
  CRESTE_REQUIRE(N > 0 && H >= 2 && W >= 2 && Cin > 0 && Cin % 64 == 0 && Cout > 0 && Cout % 4 == 0 && x_cs >= Cin && x_cs % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_ring) & 15) == 0 && out_cs >= out_co + Cout,
                 "upconv2x_ring_fix: maps of at least 2 x 2, Cin a multiple of 64, Cout a multiple of 4, 16-byte aligned x / weights");
  return upconv2x_ring_fix_run(x, x_cs, N, H, W, Cin, w_ring, Cout, act, out, out_cs, out_co, (hipStream_t)stream);
}

extern "C" int creste_conv_wino4_gemm_probe(int enable) {
  creste::g_w4_probe.store(enable != 0);
  return CRESTE_OK;
}

extern "C" int creste_conv_wino4_gemm_last_ms(float* ms) {
  using namespace creste;
  CRESTE_REQUIRE(ms && g_w4_ev[0] && g_w4_ev[1], "conv_wino4_gemm_last_ms: no probed call yet");
  CRESTE_HIP(hipEventSynchronize(g_w4_ev[1]));
  CRESTE_HIP(hipEventElapsedTime(ms, g_w4_ev[0], g_w4_ev[1]));
  return CRESTE_OK;
}
