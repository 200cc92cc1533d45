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
  w.strip = (int*)p;
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

// Sort every base cell's list by point id.  One thread per extended cell for the short lists (<= 4 entries: an
// unrolled insertion); longer lists are then taken one at a time by the whole wave: every lane holds one entry, its
// rank is the number of smaller ids (n shuffle steps), and the entry is stored at its rank -- a 43-entry list costs
// 43 shuffle steps instead of ~460 dependent global-memory steps of a one-thread insertion sort (which made one
// crowded cell hold up its whole wave).  Lists beyond 64 entries (degenerate pile-ups) fall back to the serial
// insertion sort up to kMaxSortedList entries and are left in atomic order beyond that (still the exact set).
constexpr int kMaxSortedList = 2048;
__global__ __launch_bounds__(256) void splat_sort_kernel(const int* __restrict__ offset,
                                                         int* __restrict__ list, int B, int P, int E) {
  const long total = (long)B * E;
  const int lane = threadIdx.x & 63;
  for (long i0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) - lane; i0 < total; i0 += (long)gridDim.x * blockDim.x) {
    const long i = i0 + lane;
    int lo = 0, n = 0;
    int* l = list;
    if (i < total) {
      const int b = (int)(i / E), k = (int)(i % E);
      lo = offset[(long)b * (E + 1) + k];
      n = offset[(long)b * (E + 1) + k + 1] - lo;
      l = list + (long)b * P;
    }
    if (n >= 2 && n <= 4) {
      for (int a = lo + 1; a < lo + n; ++a) {
        const int v = l[a];
        int j = a - 1;
        while (j >= lo && l[j] > v) { l[j + 1] = l[j]; --j; }
        l[j + 1] = v;
      }
    }
    unsigned long long heavy = __ballot(n > 4);
    while (heavy) {
      const int src = __ffsll((long long)heavy) - 1;
      heavy &= heavy - 1;
      const int hlo = __shfl(lo, src, 64), hn = __shfl(n, src, 64);
      // the owning lane's list base (64-bit pointer as two shuffles)
      const unsigned long long lp = (unsigned long long)l;
      int* hl = (int*)(((unsigned long long)(unsigned)__shfl((int)(lp >> 32), src, 64) << 32) |
                       (unsigned)__shfl((int)(lp & 0xffffffffu), src, 64));
      if (hn <= 64) {
        const int v = lane < hn ? hl[hlo + lane] : 0x7fffffff;
        int rank = 0;
        for (int e = 0; e < hn; ++e) rank += __shfl(v, e, 64) < v ? 1 : 0;     // ids are distinct
        if (lane < hn) hl[hlo + rank] = v;
      } else if (lane == 0 && hn <= kMaxSortedList) {
        for (int a = hlo + 1; a < hlo + hn; ++a) {
          const int v = hl[a];
          int j = a - 1;
          while (j >= hlo && hl[j] > v) { hl[j + 1] = hl[j]; --j; }
          hl[j + 1] = v;
        }
      }
    }
  }
}

// After the sort: the fractional voxel coordinates of every CSR entry, in list order (the bin phase's key / rank
// arrays are free by now) -- the gather then reads a point's id and its tap-weight factors side by side instead of
// chasing id -> coords.
__global__ __launch_bounds__(256) void splat_frac_kernel(const int* __restrict__ offset, const int* __restrict__ list,
                                                         const float* __restrict__ coords, float* __restrict__ frx,
                                                         float* __restrict__ fry, long BP, int P, int E) {
  for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < BP; g += (long)gridDim.x * blockDim.x) {
    const int b = (int)(g / P), slot = (int)(g - (long)b * P);
    if (slot >= offset[(long)b * (E + 1) + E]) continue;
    const long q = (long)b * P + list[g];
    const float Xf = coords[q * 2 + 0], Yf = coords[q * 2 + 1];
    frx[g] = __fsub_rn(Xf, floorf(Xf));
    fry[g] = __fsub_rn(Yf, floorf(Yf));
  }
}

constexpr int SPLAT_CPG = 4;     // consecutive BEV cells per lane group
template <int LANES, int MODE>   // lanes per BEV cell (>= F/4, power of two)
__global__ __launch_bounds__(256) void splat_gather_kernel(
    const float* __restrict__ feats, int feats_cs, const float* __restrict__ frx, const float* __restrict__ fry,
    const int* __restrict__ offset, const int* __restrict__ list, int B, int P, int F, int GH, int GW,
    float min_weight, float* __restrict__ bev, float* __restrict__ dens) {
  const int EW = GW + 1, E = (GH + 1) * (GW + 1);
  const int sub = threadIdx.x % LANES;
  const int fq = F >> 2;
  const bool lane_on = sub < fq;
  const long ncell = (long)B * GH * GW;
  constexpr int CELLS_PER_BLOCK = 256 / LANES;
  // a lane group walks SPLAT_CPG consecutive cells (wave launches are not free: one cell per group left the chip at
  // 3.5 of 8 waves per SIMD, bound by the dispatcher); the CSR offsets of the next cell are fetched while the
  // current one is accumulated
  // contiguous cell ranges per XCD: a point's four cells (X, X+1 on rows Y, Y+1) are then served by ONE L2
  const long cell0 = ((long)xcd_remap(blockIdx.x, gridDim.x) * CELLS_PER_BLOCK + threadIdx.x / LANES) * SPLAT_CPG;
  if (cell0 >= ncell) return;
  // (frame, row, column) of the group's first cell: ONE division, then increments (64-bit div/mod per cell was
  // most of the kernel's instruction count)
  int X, Y, b;
  {
    const long rowi = cell0 / GW;
    X = (int)(cell0 - rowi * GW);
    b = (int)(rowi / GH);
    Y = (int)(rowi - (long)b * GH);
  }
  auto fetch_ranges = [&](int bb, int yy, int xx, int* lo, int* cnt) __attribute__((always_inline)) {
    const int* off = offset + (long)bb * (E + 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = (yy - (t & 1) + 1) * EW + (xx - (t >> 1) + 1);
      lo[t] = off[k];
      cnt[t] = off[k + 1] - lo[t];
    }
  };
  int nlo[4], ncnt[4];
  fetch_ranges(b, Y, X, nlo, ncnt);
  for (int ci = 0; ci < SPLAT_CPG; ++ci) {
    const long cell = cell0 + ci;
    if (cell >= ncell) break;
    int lo[4], cnt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { lo[t] = nlo[t]; cnt[t] = ncnt[t]; }
    const int* lst = list + (long)b * P;
    const float* fb = feats + (long)b * P * feats_cs + (lane_on ? sub : 0) * 4;
    const float* rxb = frx + (long)b * P;
    const float* ryb = fry + (long)b * P;
    if (++X == GW) { X = 0; if (++Y == GH) { Y = 0; ++b; } }          // next cell (prefetch its CSR ranges)
    if (ci + 1 < SPLAT_CPG && cell + 1 < ncell) fetch_ranges(b, Y, X, nlo, ncnt);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float d = 0.f;
    // the (<= 4) base cells whose taps land here, in the reference's tap order (xd,yd) = (0,0),(0,1),(1,0),(1,1):
    // their CSR ranges are fetched together and walked as ONE concatenated sequence, LANES entries at a time.
    // Every lane fetches one entry's point id and coordinates and forms its tap weight, then the group steps through
    // the batch with shuffles: the offset -> id -> coords -> features chain of dependent loads is paid once per
    // batch (not once per point and tap), and the feature loads of consecutive points are independent.  The
    // accumulation order is the reference's (tap-major, point id ascending).
    const int c0 = cnt[0], c01 = c0 + cnt[1], c012 = c01 + cnt[2], T = c012 + cnt[3];
    for (int base = 0; base < T; base += LANES) {
      const int v = base + sub;
      int p_l = 0;
      float w_l = 0.f;
      if (v < T) {
        const int tap = (v >= c0) + (v >= c01) + (v >= c012);
        const int idx = tap == 0 ? lo[0] + v : tap == 1 ? lo[1] + v - c0 : tap == 2 ? lo[2] + v - c01 : lo[3] + v - c012;
        p_l = lst[idx];
        const float rX = rxb[idx], rY = ryb[idx];
        const float wX = (tap >> 1) ? rX : __fsub_rn(1.f, rX);
        const float wY = (tap & 1) ? rY : __fsub_rn(1.f, rY);
        w_l = __fmul_rn(wX, wY);
      }
      const int n = min(LANES, T - base);
      // four points per step, BRANCH-FREE: the four feature rows are loaded before the first is used (lanes beyond
      // F/4 and steps beyond n read a valid row and contribute an exact +0), so four loads are in flight instead of
      // one (load, s_waitcnt) round trip per point
      for (int e0 = 0; e0 < n; e0 += 4) {
        f32x4 f[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ee = e0 + u;
          const int pp = __shfl(p_l, ee & (LANES - 1), LANES);
          const float ww = __shfl(w_l, ee & (LANES - 1), LANES);
          w[u] = ee < n ? ww : 0.f;
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
                   align256((size_t)B * (E + 1) * 4) + align256((size_t)B * ((E + 1023) / 1024) * 4));
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
  {
    const long tot = (long)B * E;
    const int g = (int)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256);
    splat_sort_kernel<<<g, 256, 0, s>>>(w.offset, w.list, B, P, E);
    CRESTE_CHECK_LAUNCH("splat_sort");
  }
  splat_frac_kernel<<<g1, 256, 0, s>>>(w.offset, w.list, coords, (float*)w.key, (float*)w.rank, BP, P, E);
  CRESTE_CHECK_LAUNCH("splat_frac");
  const long ncell = (long)B * GH * GW;
  const int fq = F / 4;
#define CRESTE_SPLAT_GATHER(L, M)                                                                                   \
  {                                                                                                                 \
    const long per = (256 / L) * SPLAT_CPG;                                                                         \
    /* one pass per workgroup (no grid-stride): crowded cells sit at the same map position in every frame, and a \
       strided walk hands all of them to the same few workgroups */                                              \
    const int g = (int)((ncell + per - 1) / per > 4194304 ? 4194304 : (ncell + per - 1) / per);                     \
    splat_gather_kernel<L, M><<<g, 256, 0, s>>>(feats, feats_cs, (const float*)w.key, (const float*)w.rank, w.offset, \
                                                w.list, B, P, F, GH, GW,                                            \
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
