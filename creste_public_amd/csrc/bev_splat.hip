// Depth-guided camera->BEV bilinear voxel pooling (mean) -- gather formulation, no float atomics.
//
// reference: creste/models/blocks/splat_projection.py:185-187 (map coords) and :293-352 (4-tap
// bilinear scatter_add_ of weights and weighted features, then / clamp(density, min_weight)).
//
// The reference scatters: 8 scatter_add_ launches over a [B,F,P] tensor, a zero-filled [B,F,G]
// accumulator, then a normalisation pass -- >= 3 passes over the 25 MB/frame BEV tensor plus
// float atomics on a GPU.  Here the irregular part is reduced to INTEGER work on 4-byte keys:
//   1. bin    : per point, voxel coords (bev_coords, bit-exact arithmetic), base cell (X0,Y0) on an
//               extended (GH+1)x(GW+1) grid (X0,Y0 in [-1, G-1] still own in-grid taps), and its rank
//               inside that cell (one int atomic on an L2-resident 264 KB/frame histogram)
//   2. scan   : exclusive prefix sum of the histogram (one workgroup per frame)
//   3. fill   : CSR list of point ids per base cell
//   4. gather : each BEV cell visits the <=4 base cells whose taps land on it, in the reference's tap
//               order (xd,yd) = (0,0),(0,1),(1,0),(1,1), accumulates w and w*f in registers, divides by
//               max(density, min_weight) and writes its F channels exactly once (16-B stores).
// HBM traffic = features read once (4x re-reads are L2 hits) + BEV written once = the algorithmic
// 4*(F*P + 2P + F*G + G) bytes of SURVEY.md section 8d; no memset, no read-modify-write.
// Each base cell's list is then sorted by point id, so a cell's sums are accumulated in the
// reference's CPU order (tap-major, point-ascending): run-to-run deterministic, and bit-identical to
// the CPU scatter_add_ given identical inputs.
//
// Round 2 (the frustum distribution was bound by LATENCY CHAINS, not bytes): (a) the per-cell sort works on an LDS
// copy of a workgroup's contiguous CSR chunk (rank = number of smaller ids, LDS broadcast reads) instead of dependent
// global loads per list, and emits one 16-byte {id, frac x, frac y} record per entry in sorted order (the separate
// `frac` pass is gone); (b) the gather loads the CSR ranges of its four cells with ONE load per lane, zero-fills empty
// cells at once, fetches a whole batch of records with one 16-byte load per lane and keeps 8 feature rows in flight
// per lane group (was 4 rows behind three dependent index loads).
#include "common.h"

namespace creste {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SplatWork {
  int* key;      // [B*P]  extended base-cell id or -1
  int* rank;     // [B*P]
  int* count;    // [B*E]
  int* offset;   // [B*(E+1)]
  int* list;     // [B*P]
  int* strip;    // [B*ceil(E/1024)] scan strip sums
  int4* rec;     // [B*P]  {point id, frac x bits, frac y bits, 0} per CSR entry, in sorted list order
};

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static SplatWork carve(void* work, int B, int P, int E) {
  char* p = (char*)work;
  SplatWork w;
  w.key = (int*)p;    p += align256((size_t)B * P * 4);
  w.rank = (int*)p;   p += align256((size_t)B * P * 4);
  w.count = (int*)p;  p += align256((size_t)B * E * 4);
  w.offset = (int*)p; p += align256((size_t)B * (E + 1) * 4);
  w.list = (int*)p;   p += align256((size_t)B * P * 4);
  w.strip = (int*)p;  p += align256((size_t)B * ((E + 1023) / 1024) * 4);
  w.rec = (int4*)p;
  return w;
}

__global__ __launch_bounds__(256) void splat_bin_kernel(const float* __restrict__ xyz, long BP, int P,
                                                        float off_x, float off_y, float vox_x,
                                                        float vox_y, int GH, int GW,
                                                        float* __restrict__ coords,
                                                        int* __restrict__ key, int* __restrict__ rank,
                                                        int* __restrict__ count) {
  const int EW = GW + 1, E = (GH + 1) * (GW + 1);
  for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < BP; g += (long)gridDim.x * blockDim.x) {
    const float x = xyz[g * 3 + 0], y = xyz[g * 3 + 1];
    // map = lidar2map @ [x,y,z,1]: rows (0,-1,0,off_x), (-1,0,0,off_y) -> one rounding each
    const float mx = __fadd_rn(-y, off_x), my = __fadd_rn(-x, off_y);
    const float X = __fdiv_rn(mx, vox_x), Y = __fdiv_rn(my, vox_y);
    coords[g * 2 + 0] = X;
    coords[g * 2 + 1] = Y;
    const float fx = floorf(X), fy = floorf(Y);
    int k = -1;
    if (fx >= -1.f && fx <= (float)(GW - 1) && fy >= -1.f && fy <= (float)(GH - 1)) {
      const int b = (int)(g / P);
      k = ((int)fy + 1) * EW + ((int)fx + 1);
      rank[g] = atomicAdd(&count[(long)b * E + k], 1);
    }
    key[g] = k;
  }
}

// offset[b][0..E] = exclusive scan of count[b][0..E), two launches over 1024-element strips:
//   reduce: strip_sum[b][j] = sum of strip j            (grid = strips x frames, coalesced)
//   apply : every strip adds the (<= 65) preceding strip sums to its local shuffle/LDS scan
constexpr int SCAN_STRIP = 1024;
__global__ __launch_bounds__(256) void splat_scan_reduce_kernel(const int* __restrict__ count,
                                                                int* __restrict__ strip_sum, int E,
                                                                int nstrip) {
  __shared__ int ws[4];
  const int b = blockIdx.y, j = blockIdx.x, t = threadIdx.x;
  const int* c = count + (long)b * E + (long)j * SCAN_STRIP;
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_STRIP / 256; ++k) {
    const int i = k * 256 + t;
    if (j * SCAN_STRIP + i < E) s += c[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((t & 63) == 0) ws[t >> 6] = s;
  __syncthreads();
  if (t == 0) strip_sum[(long)b * nstrip + j] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(1024) void splat_scan_apply_kernel(const int* __restrict__ count,
                                                                const int* __restrict__ strip_sum,
                                                                int* __restrict__ offset, int E, int nstrip) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int b = blockIdx.y, j = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (wave == 0) {                                   // prefix of the preceding strips (nstrip <= 128)
    int v = 0;
    for (int k = lane; k < j; k += 64) v += strip_sum[(long)b * nstrip + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) base_s = v;
  }
  const int i = j * SCAN_STRIP + t;
  const int v = i < E ? count[(long)b * E + i] : 0;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(inc, d);
    if (lane >= d) inc += up;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += wsum[w];
  int* o = offset + (long)b * (E + 1);
  if (i < E) o[i] = base_s + wave_off + inc - v;
  if (i == E - 1) o[E] = base_s + wave_off + inc;
}

__global__ __launch_bounds__(256) void splat_fill_kernel(const int* __restrict__ key,
                                                         const int* __restrict__ rank,
                                                         const int* __restrict__ offset,
                                                         int* __restrict__ list, long BP, int P, int E) {
  for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < BP; g += (long)gridDim.x * blockDim.x) {
    const int k = key[g];
    if (k < 0) continue;
    const int b = (int)(g / P);
    list[(long)b * P + offset[(long)b * (E + 1) + k] + rank[g]] = (int)(g % P);
  }
}

// Sort every base cell's list by point id and emit the gather's entry records.  One workgroup owns SORT_CELLS
// consecutive extended cells of one frame = ONE contiguous chunk of the CSR list: the chunk is copied to LDS (coalesced),
// every entry finds its rank inside its own cell's list as the number of smaller ids (LDS reads; the lanes of a wave
// mostly scan the same list -> broadcasts), and is written straight to its sorted slot together with its record
// {id, frac(X), frac(Y)} (the gather then needs ONE 16-byte load per entry instead of id -> coords chains).
// A 43-entry list costs 43 LDS reads per entry; the old kernel walked heavy lists one at a time per wave through
// global memory (a crowded wave: 64 lists x ~1.5 us -- the kernel's critical path, 84 us at batch 16).
// Chunks beyond SORT_CAP entries (degenerate pile-ups) fall back to a one-thread insertion sort per list in global
// memory up to kMaxSortedList entries; longer lists stay in atomic order (still the exact set).
constexpr int kMaxSortedList = 2048;
constexpr int SORT_CELLS = 128, SORT_CAP = 6144;
__global__ __launch_bounds__(256) void splat_sort_rec_kernel(const int* __restrict__ offset, int* __restrict__ list,
                                                             const float* __restrict__ coords,
                                                             int4* __restrict__ rec, int P, int E) {
  __shared__ int s_off[SORT_CELLS + 1];
  __shared__ int s_ids[SORT_CAP];
  __shared__ unsigned char s_cell[SORT_CAP];
  const int b = blockIdx.y, c0 = blockIdx.x * SORT_CELLS, t = threadIdx.x;
  const int nc = min(SORT_CELLS, E - c0);
  const int* off = offset + (long)b * (E + 1) + c0;
  for (int i = t; i <= nc; i += 256) s_off[i] = off[i];
  __syncthreads();
  const int lo0 = s_off[0], n = s_off[nc] - lo0;
  if (n == 0) return;
  int* l = list + (long)b * P;
  int4* rb = rec + (long)b * P;
  const float* cb = coords + (long)b * P * 2;
  auto emit = [&](int pos, int id) __attribute__((always_inline)) {
    const float Xf = cb[(long)id * 2 + 0], Yf = cb[(long)id * 2 + 1];
    int4 r;
    r.x = id;
    r.y = __float_as_int(__fsub_rn(Xf, floorf(Xf)));
    r.z = __float_as_int(__fsub_rn(Yf, floorf(Yf)));
    r.w = 0;
    rb[pos] = r;
  };
  if (n <= SORT_CAP) {
    for (int e = t; e < n; e += 256) s_ids[e] = l[lo0 + e];
    if (t < nc)
      for (int e = s_off[t] - lo0; e < s_off[t + 1] - lo0; ++e) s_cell[e] = (unsigned char)t;
    __syncthreads();
    for (int e = t; e < n; e += 256) {
      const int c = s_cell[e], a = s_off[c] - lo0, z = s_off[c + 1] - lo0;
      const int v = s_ids[e];
      int rank = 0;
      if (z - a <= kMaxSortedList)
        for (int j = a; j < z; ++j) rank += s_ids[j] < v ? 1 : 0;          // ids of a list are distinct
      else
        rank = e - a;                                                      // degenerate list: atomic order
      const int pos = lo0 + a + rank;
      l[pos] = v;
      emit(pos, v);
    }
    return;
  }
  if (t < nc) {                       // oversized chunk: serial insertion sort per list, in place
    const int a = s_off[t], z = s_off[t + 1];
    if (z - a >= 2 && z - a <= kMaxSortedList)
      for (int i = a + 1; i < z; ++i) {
        const int v = l[i];
        int j = i - 1;
        while (j >= a && l[j] > v) { l[j + 1] = l[j]; --j; }
        l[j + 1] = v;
      }
    for (int i = a; i < z; ++i) emit(i, l[i]);
  }
}

// ------------------------------------------------------------------------------------------------ gather
// The gather is bound by VALU issue and by per-cell latency chains, not by bytes (403 MB written + 284 MB read at
// batch 16).  Fast path for F = 32*NQ <= 128 channels: one workgroup = one BEV row; the two CSR offset rows it needs are
// staged in LDS once; a cell is owned by EIGHT lanes (a lane holds NQ float4 channel quads: the eight lanes of a load
// instruction cover 128 contiguous bytes of a feature row), so a wave works on eight cells at a time and every
// per-cell / per-entry instruction is amortised over eight cells (the 32-lane version amortised over two and spent
// most of its time on index arithmetic and shuffles).  Cells go to the 32 lane groups round-robin along the row, so a
// crowded blob of cells is spread over all groups.  Per entry: one broadcast 16-byte record load (id, frac x, frac y),
// NQ feature loads; SPLAT_ROWS entries are in flight per lane group.  Sums run in the reference's order (tap-major,
// point id ascending) -- bit-identical to the CPU scatter_add_.
constexpr int SPLAT_ROWS = 4;    // entries (feature rows) in flight per lane group
template <int NQ, int MODE>
__global__ __launch_bounds__(256) void splat_gather8_kernel(
    const float* __restrict__ feats, int feats_cs, const int4* __restrict__ rec, const int* __restrict__ offset,
    int B, int P, int GH, int GW, float min_weight, float* __restrict__ bev, float* __restrict__ dens) {
  constexpr int F = NQ * 32;
  extern __shared__ int s_off[];                 // [2][EW + 1]: extended rows Y (taps yd=1) and Y+1 (taps yd=0)
  const int EW = GW + 1, E = (GH + 1) * EW, SW = EW + 1;
  const int row = xcd_remap(blockIdx.x, gridDim.x);          // contiguous row ranges per XCD: rows Y, Y+1 share an L2
  const int b = row / GH, Y = row - b * GH;
  const int* off = offset + (long)b * (E + 1);
  for (int i = threadIdx.x; i < SW; i += 256) {
    s_off[i] = off[Y * EW + i];
    s_off[SW + i] = off[(Y + 1) * EW + i];
  }
  __syncthreads();
  const int g = threadIdx.x >> 3, l = threadIdx.x & 7;
  const int4* rb = rec + (long)b * P;
  const float* fb = feats + (long)b * P * feats_cs + l * 4;
  float* orow = bev + ((long)row * GW) * F + l * 4;
  float* drow = dens + (long)row * GW;
  for (int X = g; X < GW; X += 32) {
    // taps (xd,yd) = (0,0),(0,1),(1,0),(1,1): base cell column X - xd + 1, extended row Y - yd + 1
    const int l0 = s_off[SW + X + 1], c0 = s_off[SW + X + 2] - l0;
    const int l1 = s_off[X + 1], c1 = s_off[X + 2] - l1;
    const int l2 = s_off[SW + X], c2 = l0 - l2;
    const int l3 = s_off[X], c3 = l1 - l3;
    const int c01 = c0 + c1, c012 = c01 + c2, T = c012 + c3;
    const int d1 = l1 - c0, d2 = l2 - c01, d3 = l3 - c012;      // list index = e + d_tap
    f32x4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float d = 0.f;
    for (int e0 = 0; e0 < T; e0 += SPLAT_ROWS) {
      f32x4 f[SPLAT_ROWS][NQ];
      float w[SPLAT_ROWS];
#pragma unroll
      for (int u = 0; u < SPLAT_ROWS; ++u) {
        const int e = min(e0 + u, T - 1);                         // past the end: a valid entry, never accumulated
        const int t1 = e >= c0, t2 = e >= c01, t3 = e >= c012;
        const int idx = e + (t3 ? d3 : t2 ? d2 : t1 ? d1 : l0);
        const int4 r = rb[idx];
        const float rX = __int_as_float(r.y), rY = __int_as_float(r.z);
        const float wX = t2 ? rX : __fsub_rn(1.f, rX);            // xd = tap >> 1
        const float wY = (t1 != t2 || t3) ? rY : __fsub_rn(1.f, rY);   // yd = tap & 1: taps 1 and 3
        w[u] = __fmul_rn(wX, wY);
        const float* fr = fb + (unsigned)(r.x * feats_cs);
#pragma unroll
        for (int q = 0; q < NQ; ++q) f[u][q] = *reinterpret_cast<const f32x4*>(fr + q * 32);
      }
#pragma unroll
      for (int u = 0; u < SPLAT_ROWS; ++u) {
        if (e0 + u < T) {
          d = __fadd_rn(d, w[u]);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[q][j] = MODE == 2 ? fmaxf(acc[q][j], __fmul_rn(w[u], f[u][q][j]))
                                    : __fadd_rn(acc[q][j], __fmul_rn(w[u], f[u][q][j]));
        }
      }
    }
    const float den = fmaxf(d, min_weight);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = MODE == 0 ? __fdiv_rn(acc[q][j], den) : acc[q][j];
      *reinterpret_cast<f32x4*>(orow + (long)X * F + q * 32) = o;
    }
    if (l == 0) drow[X] = d;
  }
}

// Generic path (any F <= 256 that is a multiple of 4): LANES >= F/4 lanes per cell, SPLAT_CPG consecutive cells per lane
// group, records fetched one batch of LANES entries at a time and handed round with shuffles.
constexpr int SPLAT_CPG = 4;
template <int LANES, int MODE>
__global__ __launch_bounds__(256) void splat_gather_kernel(
    const float* __restrict__ feats, int feats_cs, const int4* __restrict__ rec,
    const int* __restrict__ offset, int B, int P, int F, int GH, int GW,
    float min_weight, float* __restrict__ bev, float* __restrict__ dens) {
  const int EW = GW + 1, E = (GH + 1) * (GW + 1);
  const int sub = threadIdx.x % LANES;
  const int fq = F >> 2;
  const bool lane_on = sub < fq;
  const long ncell = (long)B * GH * GW;
  constexpr int CELLS_PER_BLOCK = 256 / LANES;
  const long cell0 = ((long)xcd_remap(blockIdx.x, gridDim.x) * CELLS_PER_BLOCK + threadIdx.x / LANES) * SPLAT_CPG;
  if (cell0 >= ncell) return;
  int X, Y, b;                                     // ONE 64-bit division per lane group, then increments
  {
    const long rowi = cell0 / GW;
    X = (int)(cell0 - rowi * GW);
    b = (int)(rowi / GH);
    Y = (int)(rowi - (long)b * GH);
  }
  for (int ci = 0; ci < SPLAT_CPG; ++ci) {
    const long cell = cell0 + ci;
    if (cell >= ncell) break;
    int lo[4], cnt[4];
    const int* off = offset + (long)b * (E + 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = (Y - (t & 1) + 1) * EW + (X - (t >> 1) + 1);
      lo[t] = off[k];
      cnt[t] = off[k + 1] - lo[t];
    }
    const int4* rb = rec + (long)b * P;
    const float* fb = feats + (long)b * P * feats_cs + (lane_on ? sub : 0) * 4;
    if (++X == GW) { X = 0; if (++Y == GH) { Y = 0; ++b; } }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float d = 0.f;
    const int c0 = cnt[0], c01 = c0 + cnt[1], c012 = c01 + cnt[2], T = c012 + cnt[3];
    for (int base = 0; base < T; base += LANES) {
      const int v = base + sub;
      int p_l = 0;
      float w_l = 0.f;
      if (v < T) {
        const int tap = (v >= c0) + (v >= c01) + (v >= c012);
        const int idx = tap == 0 ? lo[0] + v : tap == 1 ? lo[1] + v - c0 : tap == 2 ? lo[2] + v - c01 : lo[3] + v - c012;
        const int4 r = rb[idx];
        const float rX = __int_as_float(r.y), rY = __int_as_float(r.z);
        const float wX = (tap >> 1) ? rX : __fsub_rn(1.f, rX);
        const float wY = (tap & 1) ? rY : __fsub_rn(1.f, rY);
        p_l = r.x;
        w_l = __fmul_rn(wX, wY);
      }
      const int n = min(LANES, T - base);
      for (int e0 = 0; e0 < n; e0 += 4) {
        f32x4 f[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ee = e0 + u;
          const int pp = __shfl(p_l, ee & (LANES - 1), LANES);
          w[u] = __shfl(w_l, ee & (LANES - 1), LANES);
          f[u] = *reinterpret_cast<const f32x4*>(fb + (unsigned)((ee < n ? pp : 0) * feats_cs));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (e0 + u < n) {                              // group-uniform; keeps the sums bit-identical
            d = __fadd_rn(d, w[u]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[j] = MODE == 2 ? fmaxf(acc[j], __fmul_rn(w[u], f[u][j])) : __fadd_rn(acc[j], __fmul_rn(w[u], f[u][j]));
          }
        }
      }
    }
    const float den = fmaxf(d, min_weight);
    if (lane_on) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = MODE == 0 ? __fdiv_rn(acc[j], den) : acc[j];
      *reinterpret_cast<f32x4*>(bev + cell * F + sub * 4) = o;
    }
    if (sub == 0) dens[cell] = d;
  }
}

}  // namespace creste

using namespace creste;

extern "C" int64_t creste_bev_splat_workspace_bytes(int B, int P, int GH, int GW) {
  if (B <= 0 || P <= 0 || GH <= 0 || GW <= 0) return -1;
  const int E = (GH + 1) * (GW + 1);
  return (int64_t)(3 * align256((size_t)B * P * 4) + align256((size_t)B * E * 4) +
                   align256((size_t)B * (E + 1) * 4) + align256((size_t)B * ((E + 1023) / 1024) * 4) +
                   align256((size_t)B * P * 16));
}

extern "C" int creste_bev_splat_mode_f32(const float* xyz, const float* feats, int feats_cs, int B, int P,
                                         int F, float off_x, float off_y, float vox_x, float vox_y, int GH,
                                         int GW, float min_weight, int mode, float* coords, float* bev,
                                         float* dens, void* work, void* stream) {
  CRESTE_REQUIRE(xyz && feats && coords && bev && dens && work, "bev_splat: null pointer");
  CRESTE_REQUIRE(mode == CRESTE_SPLAT_MEAN || mode == CRESTE_SPLAT_SUM || mode == CRESTE_SPLAT_MAX,
                 "bev_splat: unknown scatter mode %d", mode);
  CRESTE_REQUIRE(B > 0 && P > 0 && F > 0 && F % 4 == 0 && F <= 256 && feats_cs % 4 == 0 && feats_cs >= F,
                 "bev_splat: F must be a multiple of 4 and <= 256");
  CRESTE_REQUIRE(GH > 0 && GW > 0 && vox_x > 0.f && vox_y > 0.f, "bev_splat: bad grid");
  CRESTE_REQUIRE((long)P * feats_cs < (1L << 31), "bev_splat: P * feature stride overflows the 32-bit point offset");
  const int E = (GH + 1) * (GW + 1);
  const long BP = (long)B * P;
  hipStream_t s = (hipStream_t)stream;
  SplatWork w = carve(work, B, P, E);
  CRESTE_HIP(hipMemsetAsync(w.count, 0, (size_t)B * E * 4, s));
  const int g1 = (int)((BP + 255) / 256 > 4096 ? 4096 : (BP + 255) / 256);
  splat_bin_kernel<<<g1, 256, 0, s>>>(xyz, BP, P, off_x, off_y, vox_x, vox_y, GH, GW, coords, w.key,
                                      w.rank, w.count);
  CRESTE_CHECK_LAUNCH("splat_bin");
  const int nstrip = (E + SCAN_STRIP - 1) / SCAN_STRIP;
  splat_scan_reduce_kernel<<<dim3(nstrip, B), 256, 0, s>>>(w.count, w.strip, E, nstrip);
  splat_scan_apply_kernel<<<dim3(nstrip, B), 1024, 0, s>>>(w.count, w.strip, w.offset, E, nstrip);
  CRESTE_CHECK_LAUNCH("splat_scan");
  splat_fill_kernel<<<g1, 256, 0, s>>>(w.key, w.rank, w.offset, w.list, BP, P, E);
  CRESTE_CHECK_LAUNCH("splat_fill");
  splat_sort_rec_kernel<<<dim3((E + SORT_CELLS - 1) / SORT_CELLS, B), 256, 0, s>>>(w.offset, w.list, coords, w.rec, P, E);
  CRESTE_CHECK_LAUNCH("splat_sort_rec");
  const long ncell = (long)B * GH * GW;
  const int fq = F / 4;
  if (F % 32 == 0 && F <= 128 && (long)B * GH < (1L << 30) && (feats_cs % 4) == 0) {      // row-per-workgroup fast path
    const int rows = B * GH;
    const size_t smem = 2 * (size_t)(GW + 2) * sizeof(int);
    CRESTE_REQUIRE(smem <= 64 * 1024, "bev_splat: grid width %d too large for the offset staging", GW);
#define CRESTE_SPLAT_G8(NQ, M) splat_gather8_kernel<NQ, M><<<rows, 256, smem, s>>>(feats, feats_cs, w.rec, w.offset, B, P, GH, GW, min_weight, bev, dens)
#define CRESTE_SPLAT_G8M(NQ)                                                     \
    {                                                                            \
      if (mode == CRESTE_SPLAT_MEAN) CRESTE_SPLAT_G8(NQ, 0);                     \
      else if (mode == CRESTE_SPLAT_SUM) CRESTE_SPLAT_G8(NQ, 1);                 \
      else CRESTE_SPLAT_G8(NQ, 2);                                               \
    }
    if (F == 32) CRESTE_SPLAT_G8M(1) else if (F == 64) CRESTE_SPLAT_G8M(2) else if (F == 96) CRESTE_SPLAT_G8M(3) else CRESTE_SPLAT_G8M(4)
#undef CRESTE_SPLAT_G8M
#undef CRESTE_SPLAT_G8
    CRESTE_CHECK_LAUNCH("splat_gather8");
    return CRESTE_OK;
  }
#define CRESTE_SPLAT_GATHER(L, M)                                                                                   \
  {                                                                                                                 \
    const long per = (256 / L) * SPLAT_CPG;                                                                         \
    /* one pass per workgroup (no grid-stride): crowded cells sit at the same map position in every frame, and a \
       strided walk hands all of them to the same few workgroups */                                              \
    const int g = (int)((ncell + per - 1) / per > 4194304 ? 4194304 : (ncell + per - 1) / per);                     \
    splat_gather_kernel<L, M><<<g, 256, 0, s>>>(feats, feats_cs, w.rec, w.offset, B, P, F, GH, GW,                  \
                                                min_weight, bev, dens);                                             \
  }
#define CRESTE_SPLAT_LANES(M)                                                                                       \
  {                                                                                                                 \
    if (fq <= 8) CRESTE_SPLAT_GATHER(8, M) else if (fq <= 16) CRESTE_SPLAT_GATHER(16, M)                            \
    else if (fq <= 32) CRESTE_SPLAT_GATHER(32, M) else CRESTE_SPLAT_GATHER(64, M)                                   \
  }
  if (mode == CRESTE_SPLAT_MEAN) CRESTE_SPLAT_LANES(0)
  else if (mode == CRESTE_SPLAT_SUM) CRESTE_SPLAT_LANES(1)
  else CRESTE_SPLAT_LANES(2)
#undef CRESTE_SPLAT_LANES
#undef CRESTE_SPLAT_GATHER
  CRESTE_CHECK_LAUNCH("splat_gather");
  return CRESTE_OK;
}

extern "C" int creste_bev_splat_f32(const float* xyz, const float* feats, int feats_cs, int B, int P,
                                    int F, float off_x, float off_y, float vox_x, float vox_y, int GH,
                                    int GW, float min_weight, float* coords, float* bev, float* dens,
                                    void* work, void* stream) {
  return creste_bev_splat_mode_f32(xyz, feats, feats_cs, B, P, F, off_x, off_y, vox_x, vox_y, GH, GW, min_weight,
                                   CRESTE_SPLAT_MEAN, coords, bev, dens, work, stream);
}
