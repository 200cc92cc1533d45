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

// Sort every base cell's list by point id (serial insertion sort per list; lists are short except
// for degenerate depth maps, which are left in atomic order above kMaxSortedList entries).
// One thread per extended cell.
constexpr int kMaxSortedList = 2048;
__global__ __launch_bounds__(256) void splat_sort_kernel(const int* __restrict__ offset,
                                                         int* __restrict__ list, int B, int P, int E) {
  const long total = (long)B * E;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / E), k = (int)(i % E);
    const int lo = offset[(long)b * (E + 1) + k], hi = offset[(long)b * (E + 1) + k + 1];
    int* l = list + (long)b * P;
    if (hi - lo > kMaxSortedList) continue;   // degenerate pile-up: keep the atomic order (still exact set)
    for (int a = lo + 1; a < hi; ++a) {
      const int v = l[a];
      int j = a - 1;
      while (j >= lo && l[j] > v) { l[j + 1] = l[j]; --j; }
      l[j + 1] = v;
    }
  }
}

// MODE = the reference's scatter_mode (splat_projection.py:334-352): 0 'mean' (sum / clamp(density, min_weight)),
// 1 'sum', 2 'max' (torch_scatter's scatter-max of w*f per tap, empty cells 0, folded with torch.maximum against the
// zero-initialised volume: max(0, max over taps and points of w*f)); the density is the tap-weight sum in all three
template <int LANES, int MODE>   // lanes per BEV cell (>= F/4, power of two)
__global__ __launch_bounds__(256) void splat_gather_kernel(
    const float* __restrict__ feats, int feats_cs, const float* __restrict__ coords,
    const int* __restrict__ offset, const int* __restrict__ list, int B, int P, int F, int GH, int GW,
    float min_weight, float* __restrict__ bev, float* __restrict__ dens) {
  const int EW = GW + 1, E = (GH + 1) * (GW + 1);
  const int sub = threadIdx.x % LANES;
  const int fq = F >> 2;
  const bool lane_on = sub < fq;
  const long ncell = (long)B * GH * GW;
  constexpr int CELLS_PER_BLOCK = 256 / LANES;
  for (long cell = (long)blockIdx.x * CELLS_PER_BLOCK + threadIdx.x / LANES; cell < ncell;
       cell += (long)gridDim.x * CELLS_PER_BLOCK) {
    const int X = (int)(cell % GW);
    const int Y = (int)((cell / GW) % GH);
    const int b = (int)(cell / ((long)GW * GH));
    const int* off = offset + (long)b * (E + 1);
    const int* lst = list + (long)b * P;
    const float* fb = feats + (long)b * P * feats_cs + sub * 4;
    const float* cb = coords + (long)b * P * 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float d = 0.f;
#pragma unroll
    for (int xd = 0; xd < 2; ++xd) {
#pragma unroll
      for (int yd = 0; yd < 2; ++yd) {
        const int k = (Y - yd + 1) * EW + (X - xd + 1);
        const int lo = off[k], hi = off[k + 1];
        for (int e = lo; e < hi; ++e) {
          const int p = lst[e];
          const float Xf = cb[p * 2 + 0], Yf = cb[p * 2 + 1];
          const float rX = __fsub_rn(Xf, floorf(Xf)), rY = __fsub_rn(Yf, floorf(Yf));
          const float wX = xd ? rX : __fsub_rn(1.f, rX);
          const float wY = yd ? rY : __fsub_rn(1.f, rY);
          const float w = __fmul_rn(wX, wY);
          d = __fadd_rn(d, w);
          if (lane_on) {
            const f32x4 f = *reinterpret_cast<const f32x4*>(fb + (long)p * feats_cs);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[j] = MODE == 2 ? fmaxf(acc[j], __fmul_rn(w, f[j])) : __fadd_rn(acc[j], __fmul_rn(w, f[j]));
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
  const long ncell = (long)B * GH * GW;
  const int fq = F / 4;
#define CRESTE_SPLAT_GATHER(L, M)                                                                                   \
  {                                                                                                                 \
    const long per = 256 / L;                                                                                       \
    const int g = (int)((ncell + per - 1) / per > 16384 ? 16384 : (ncell + per - 1) / per);                         \
    splat_gather_kernel<L, M><<<g, 256, 0, s>>>(feats, feats_cs, coords, w.offset, w.list, B, P, F, GH, GW,        \
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
