// Depth-guided camera->BEV bilinear voxel pooling (mean) -- gather formulation, no float atomics.
//
// reference: creste/models/blocks/splat_projection.py:185-187 (map coords) and :293-352 (4-tap
// bilinear scatter_add_ of weights and weighted features, then / clamp(density, min_weight)).
//
// The reference scatters: 8 scatter_add_ launches over a [B,F,P] tensor, a zero-filled [B,F,G]
// accumulator, then a normalisation pass -- >= 3 passes over the 25 MB/frame BEV tensor plus
// float atomics on a GPU.  Here the irregular part is reduced to INTEGER work on 4-byte keys:
//   1. key    : per point, voxel coords (bev_coords, bit-exact arithmetic) and base cell (X0,Y0) on an
//               extended (GH+1)x(GW+1) grid (X0,Y0 in [-1, G-1] still own in-grid taps)
//      build  : a point's rank inside its cell (LDS atomic on a band histogram), the exclusive scan of the
//               counts and the CSR fill -- one launch, no global atomics (splat_build_kernel)
//   2. sort   : every cell's entries by point id, through LDS (splat_sort_rec_kernel)
//   3. gather : each BEV cell visits the <=4 base cells whose taps land on it, in the reference's tap
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
#include <stdlib.h>

#include "common.h"

namespace creste {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SplatWork {
  int* key;      // [B*P]  extended base-cell id or -1
  int* rank;     // [B*P]  a point's arrival rank inside its base cell (LDS atomic order)
  int* offset;   // [B*(E+1)] CSR offsets of the extended base cells (exclusive scan of the per-cell counts)
  int4* recu;    // [B*P]  {point id, frac x bits, frac y bits, 0} per CSR entry, arrival order inside a cell
  int4* rec;     // [B*P]  the same, every cell's entries sorted by point id
};

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static SplatWork carve(void* work, int B, int P, int E) {
  char* p = (char*)work;
  SplatWork w;
  w.key = (int*)p;    p += align256((size_t)B * P * 4);
  w.rank = (int*)p;   p += align256((size_t)B * P * 4);
  w.offset = (int*)p; p += align256((size_t)B * (E + 1) * 4);
  w.recu = (int4*)p;  p += align256((size_t)B * P * 16);
  w.rec = (int4*)p;
  return w;
}

// Per point: voxel coordinates (bev_coords, one rounding per operation as the reference) and the extended base-cell id.
__global__ __launch_bounds__(256) void splat_key_kernel(const float* __restrict__ xyz, long BP, float off_x,
                                                        float off_y, float vox_x, float vox_y, int GH, int GW,
                                                        float* __restrict__ coords, int* __restrict__ key) {
  for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < BP; g += (long)gridDim.x * blockDim.x) {
    const float x = xyz[g * 3 + 0], y = xyz[g * 3 + 1];
    // map = lidar2map @ [x,y,z,1]: rows (0,-1,0,off_x), (-1,0,0,off_y) -> one rounding each
    const float mx = __fadd_rn(-y, off_x), my = __fadd_rn(-x, off_y);
    const float X = __fdiv_rn(mx, vox_x), Y = __fdiv_rn(my, vox_y);
    *reinterpret_cast<float2*>(coords + g * 2) = make_float2(X, Y);
    const float fx = floorf(X), fy = floorf(Y);
    int k = -1;
    if (fx >= -1.f && fx <= (float)(GW - 1) && fy >= -1.f && fy <= (float)(GH - 1))
      k = ((int)fy + 1) * (GW + 1) + ((int)fx + 1);
    key[g] = k;
  }
}

// Histogram + exclusive scan + CSR fill in ONE launch, without global atomics (round 1: memset + one device-scope
// atomic per point + two scan launches + fill, ~50 us at batch 16).
// A workgroup owns one BAND of extended base-cell rows of one frame: its histogram lives in LDS.  Every workgroup of a
// frame walks the keys of ALL points of the frame (185 KB, L2 resident), keeps the ones in its band (rank = LDS atomic)
// and COUNTS the ones in lower bands -- so the band's CSR base is known without any exchange between workgroups.  After
// the in-LDS scan the offsets are final: they are written out and a second walk stores every kept point's record
// {id, frac x, frac y} at offset + rank.
constexpr int BUILD_THREADS = 1024, BUILD_MAX_CELLS = 8192;      // cells per band: <= 8 per thread, <= 14 bits
__global__ __launch_bounds__(BUILD_THREADS) void splat_build_kernel(
    const int* __restrict__ key, const float* __restrict__ coords, int P, int GH, int GW, int RPB,
    int* __restrict__ rank, int* __restrict__ offset) {
  extern __shared__ int s_hist[];                 // [band cells] counts, then absolute CSR offsets
  __shared__ int s_wave[BUILD_THREADS / 64];
  __shared__ int s_base;
  const int q = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int EW = GW + 1, E = (GH + 1) * EW;
  const int r0 = q * RPB, r1 = min(GH + 1, r0 + RPB);
  const int k0 = r0 * EW, nk = (r1 - r0) * EW, k1 = k0 + nk;
  for (int i = t; i < nk; i += BUILD_THREADS) s_hist[i] = 0;
  __syncthreads();
  const int* kb = key + (long)b * P;
  int* rk = rank + (long)b * P;
  int lower = 0;
  for (int g0 = t; g0 < P; g0 += 4 * BUILD_THREADS) {          // four points per trip: the loads are in flight together
    int ks[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ks[u] = kb[min(g0 + u * BUILD_THREADS, P - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = g0 + u * BUILD_THREADS;
      if (g < P && ks[u] >= 0) {
        if (ks[u] < k0) ++lower;
        else if (ks[u] < k1) rk[g] = atomicAdd(&s_hist[ks[u] - k0], 1);
      }
    }
  }
  // base = number of kept points of the frame in lower bands
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lower += __shfl_xor(lower, o);
  if ((t & 63) == 0) s_wave[t >> 6] = lower;
  __syncthreads();                                 // also: histogram complete
  if (t == 0) {
    int v = 0;
    for (int i = 0; i < BUILD_THREADS / 64; ++i) v += s_wave[i];
    s_base = v;
  }
  // exclusive scan of the band's counts: each thread owns IPT consecutive cells
  const int IPT = (nk + BUILD_THREADS - 1) / BUILD_THREADS;
  const int i0 = t * IPT, i1 = min(nk, i0 + IPT);
  int mine = 0;
  for (int i = i0; i < i1; ++i) mine += s_hist[i];
  int inc = mine;
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const int up = __shfl_up(inc, dd);
    if (lane >= dd) inc += up;
  }
  __syncthreads();                                 // s_base written, s_wave free again
  const int base = s_base;
  __syncthreads();
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int run = base + inc - mine;
  for (int w = 0; w < wave; ++w) run += s_wave[w];
  int* ob = offset + (long)b * (E + 1) + k0;
  for (int i = i0; i < i1; ++i) {
    const int c = s_hist[i];
    s_hist[i] = run;
    ob[i] = run;
    run += c;
  }
  if (r1 == GH + 1 && i1 == nk && i0 < i1) ob[nk] = run;       // offset[E]: the thread owning the last cell
}

// The same with every point's key (then its packed {rank, cell}) held in REGISTERS: KPT points per thread, all key loads
// in flight at once, no rank array, no second walk over the keys -- the generic kernel above pays two dependent global
// round trips per four points (41 us at batch 16); this one pays one for the keys and one for the kept points' coords.
// Needs P <= KPT * 1024 and P <= 65536 (rank << 14 | cell must fit an int).
template <int KPT>
__global__ __launch_bounds__(BUILD_THREADS) void splat_build_reg_kernel(
    const int* __restrict__ key, const float* __restrict__ coords, int P, int GH, int GW, int RPB,
    int* __restrict__ rank, int* __restrict__ offset) {
  extern __shared__ int s_hist[];                  // [nk] counts -> absolute CSR offsets
  __shared__ int s_wave[BUILD_THREADS / 64];
  const int q = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int EW = GW + 1, E = (GH + 1) * EW;
  const int r0 = q * RPB, r1 = min(GH + 1, r0 + RPB);
  const int k0 = r0 * EW, nk = (r1 - r0) * EW;
  const int* kb = key + (long)b * P;
  int* rk = rank + (long)b * P;
  int pk[KPT];
#pragma unroll
  for (int u = 0; u < KPT; ++u) {
    const int g = t + u * BUILD_THREADS;
    pk[u] = g < P ? kb[g] : -1;
  }
  for (int i = t; i < nk; i += BUILD_THREADS) s_hist[i] = 0;
  __syncthreads();
  // (an all-lanes, branch-free form -- lanes outside the band adding 0 to a per-lane sink slot -- measured SLOWER:
  // 39.8 vs 29.2 us; LDS atomics with a return value cost per active lane)
  int lower = 0;
#pragma unroll
  for (int u = 0; u < KPT; ++u) {
    const int k = pk[u];
    lower += (unsigned)k < (unsigned)k0 ? 1 : 0;       // k == -1 compares as a huge unsigned
    const unsigned rel = (unsigned)(k - k0);
    if (rel < (unsigned)nk) rk[t + u * BUILD_THREADS] = atomicAdd(&s_hist[rel], 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lower += __shfl_xor(lower, o);
  if ((t & 63) == 0) s_wave[t >> 6] = lower;
  __syncthreads();                                 // also: histogram complete
  int base = 0;
#pragma unroll
  for (int i = 0; i < BUILD_THREADS / 64; ++i) base += s_wave[i];
  // exclusive scan of the band's counts: thread t owns cells [t*IPT, (t+1)*IPT), IPT <= 8
  const int IPT = (nk + BUILD_THREADS - 1) / BUILD_THREADS;
  const int i0 = t * IPT;
  int cnt[8];
  int mine = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cnt[j] = (j < IPT && i0 + j < nk) ? s_hist[i0 + j] : 0;
    mine += cnt[j];
  }
  int inc = mine;
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const int up = __shfl_up(inc, dd);
    if (lane >= dd) inc += up;
  }
  __syncthreads();                                 // every thread has read s_wave and its counts
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int run = base + inc - mine;
  for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < IPT && i0 + j < nk) s_hist[i0 + j] = run;
    run += cnt[j];
  }
  int* ob = offset + (long)b * (E + 1) + k0;
  if (r1 == GH + 1 && t == BUILD_THREADS - 1) ob[nk] = run;    // offset[E] (trailing threads carry the total)
  __syncthreads();
  for (int i = t; i < nk; i += BUILD_THREADS) ob[i] = s_hist[i];          // coalesced
}

// One thread per point: its record {id, frac x, frac y} goes to its CSR slot offset[key] + rank (full-chip parallel;
// inside the band workgroups this step serialised a memory round trip per eight points: 24 of that kernel's 34 us).
__global__ __launch_bounds__(256) void splat_fill_rec_kernel(const int* __restrict__ key, const int* __restrict__ rank,
                                                             const int* __restrict__ offset,
                                                             const float* __restrict__ coords,
                                                             int4* __restrict__ recu, long BP, int P, int E) {
  for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < BP; g += (long)gridDim.x * blockDim.x) {
    const int k = key[g];
    if (k < 0) continue;
    const int b = (int)(g / P);
    const float2 c = *reinterpret_cast<const float2*>(coords + g * 2);
    int4 r;
    r.x = (int)(g - (long)b * P);
    r.y = __float_as_int(__fsub_rn(c.x, floorf(c.x)));
    r.z = __float_as_int(__fsub_rn(c.y, floorf(c.y)));
    r.w = 0;
    recu[(long)b * P + offset[(long)b * (E + 1) + k] + rank[g]] = r;
  }
}

// Sort every base cell's entries by point id.  One workgroup owns SORT_CELLS consecutive extended cells of one frame
// = ONE contiguous chunk of the CSR: every thread loads one 16-byte record of the chunk (coalesced), the ids go to
// LDS, every entry finds its rank inside its own cell's list as the number of smaller ids (LDS reads; the lanes of a
// wave mostly scan the same list -> broadcasts) and its record is written straight to its sorted slot.  A 43-entry list
// costs 43 LDS reads per entry; round 1 walked heavy lists one at a time per wave through global memory (a crowded
// wave: 64 lists x ~1.5 us -- that kernel's critical path, 84 us at batch 16).
// Chunks beyond SORT_CAP entries (degenerate pile-ups) rank through global memory; lists beyond kMaxSortedList entries
// stay in arrival order (still the exact set).
constexpr int kMaxSortedList = 2048;
constexpr int SORT_CELLS = 128, SORT_CAP = 6144;
__global__ __launch_bounds__(256) void splat_sort_rec_kernel(const int* __restrict__ offset,
                                                             const int4* __restrict__ recu,
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
  const int4* ru = recu + (long)b * P + lo0;
  int4* ro = rec + (long)b * P + lo0;
  if (n <= SORT_CAP) {
    for (int e = t; e < n; e += 256) s_ids[e] = ru[e].x;
    if (t < nc)
      for (int e = s_off[t] - lo0; e < s_off[t + 1] - lo0; ++e) s_cell[e] = (unsigned char)t;
    __syncthreads();
    for (int e = t; e < n; e += 256) {
      const int4 r = ru[e];
      const int c = s_cell[e], a = s_off[c] - lo0, z = s_off[c + 1] - lo0;
      int rnk = e - a;                                                     // degenerate list: arrival order
      if (z - a <= kMaxSortedList) {
        rnk = 0;
        for (int j = a; j < z; ++j) rnk += s_ids[j] < r.x ? 1 : 0;         // ids of a list are distinct
      }
      ro[a + rnk] = r;
    }
    return;
  }
  if (t < nc) {                       // oversized chunk: one thread per list, ranks through global memory
    const int a = s_off[t] - lo0, z = s_off[t + 1] - lo0;
    for (int e = a; e < z; ++e) {
      const int4 r = ru[e];
      int rnk = e - a;
      if (z - a <= kMaxSortedList) {
        rnk = 0;
        for (int j = a; j < z; ++j) rnk += ru[j].x < r.x ? 1 : 0;
      }
      ro[a + rnk] = r;
    }
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
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NQ, int MODE, int ROWS, bool NT = false>
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
    f32x2 acc[NQ][2];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q][0] = acc[q][1] = f32x2{0.f, 0.f};
    float d = 0.f;
    // the record (id, frac x, frac y) of entry e of the concatenated tap lists, and its tap weight; entries past the
    // end repeat the last one (a valid load) and are never accumulated.  Records are fetched ONE STEP AHEAD of the
    // feature rows they address, so a step's chain is one load latency, not two.
    int rid[ROWS];
    float rw[ROWS];
#define CRESTE_SPLAT_FETCH(E0)                                                                    \
  _Pragma("unroll") for (int u = 0; u < ROWS; ++u) {                                              \
    const int e = min((E0) + u, T - 1);                                                           \
    const bool t1 = e >= c0, t2 = e >= c01, t3 = e >= c012;                                       \
    const int4 r = rb[e + (t3 ? d3 : t2 ? d2 : t1 ? d1 : l0)];                                    \
    const float rX = __int_as_float(r.y), rY = __int_as_float(r.z);                               \
    const float wX = t2 ? rX : __fsub_rn(1.f, rX);                      /* xd = tap >> 1 */       \
    const float wY = ((t1 && !t2) || t3) ? rY : __fsub_rn(1.f, rY);     /* yd = tap & 1 */        \
    rid[u] = r.x * feats_cs;                                                                      \
    rw[u] = __fmul_rn(wX, wY);                                                                    \
  }
    if (T > 0) { CRESTE_SPLAT_FETCH(0) }
    for (int e0 = 0; e0 < T; e0 += ROWS) {
      f32x4 f[ROWS][NQ];
      float w[ROWS];
#pragma unroll
      for (int u = 0; u < ROWS; ++u) {
        w[u] = rw[u];
        const float* fr = fb + (unsigned)rid[u];
#pragma unroll
        for (int q = 0; q < NQ; ++q) f[u][q] = *reinterpret_cast<const f32x4*>(fr + q * 32);
      }
      if (e0 + ROWS < T) { CRESTE_SPLAT_FETCH(e0 + ROWS) }
#pragma unroll
      for (int u = 0; u < ROWS; ++u) {
        if (e0 + u < T) {
          d = __fadd_rn(d, w[u]);
          const f32x2 w2 = {w[u], w[u]};
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x2 fv = {f[u][q][2 * h], f[u][q][2 * h + 1]};
              if (MODE == 2) {
                acc[q][h][0] = fmaxf(acc[q][h][0], __fmul_rn(w[u], fv[0]));
                acc[q][h][1] = fmaxf(acc[q][h][1], __fmul_rn(w[u], fv[1]));
              } else {
                const f32x2 pr = w2 * fv;              // v_pk_mul_f32 then v_pk_add_f32 (-ffp-contract=off): the same
                acc[q][h] = acc[q][h] + pr;            // two IEEE roundings per element as the scalar form
              }
            }
        }
      }
    }
    const float den = fmaxf(d, min_weight);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = MODE == 0 ? __fdiv_rn(acc[q][j >> 1][j & 1], den) : acc[q][j >> 1][j & 1];
      if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(orow + (long)X * F + q * 32));
      else *reinterpret_cast<f32x4*>(orow + (long)X * F + q * 32) = o;
    }
    if (l == 0) drow[X] = d;
  }
#undef CRESTE_SPLAT_FETCH
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
  return (int64_t)(2 * align256((size_t)B * P * 4) + align256((size_t)B * (E + 1) * 4) + 2 * align256((size_t)B * P * 16));
}

// Binning plan of one batch of points: needs ONLY xyz (reference splat_projection.py:185-187 + the index half of :293-333),
// so the model enqueues it right behind the pixel geometry, ahead of the 288 -> 96 fusion conv that produces the features.
static int splat_plan(const float* xyz, int B, int P, float off_x, float off_y, float vox_x, float vox_y, int GH, int GW,
                      float* coords, void* work, void* stream) {      // xyz == nullptr: coords and keys are already there
  const int E = (GH + 1) * (GW + 1);
  const long BP = (long)B * P;
  hipStream_t s = (hipStream_t)stream;
  SplatWork w = carve(work, B, P, E);
  {
    // bands of extended rows: as many workgroups as the chip has CUs, at most BUILD_MAX_CELLS cells of LDS each
    const int EW = GW + 1;
    int want = 256 / B;
    if (want < 1) want = 1;
    int rpb = (GH + 1 + want - 1) / want;                       // rows per band
    const int cap_rows = BUILD_MAX_CELLS / EW;
    CRESTE_REQUIRE(cap_rows >= 1, "bev_splat: grid width %d too large for the LDS histogram", GW);
    if (rpb > cap_rows) rpb = cap_rows;
    if (rpb < 1) rpb = 1;
    const int Q = (GH + 1 + rpb - 1) / rpb;
    const size_t smem = (size_t)rpb * EW * sizeof(int);
    const int g1 = (int)((BP + 255) / 256 > 8192 ? 8192 : (BP + 255) / 256);
    if (xyz) {
      splat_key_kernel<<<g1, 256, 0, s>>>(xyz, BP, off_x, off_y, vox_x, vox_y, GH, GW, coords, w.key);
      CRESTE_CHECK_LAUNCH("splat_key");
    }
    const int kpt = (P + BUILD_THREADS - 1) / BUILD_THREADS;
    static_assert(BUILD_MAX_CELLS <= 16384, "the packed {rank, cell} word keeps 14 bits for the cell");
    if (P <= 65536 && kpt <= 16)
      splat_build_reg_kernel<16><<<dim3(Q, B), BUILD_THREADS, smem, s>>>(w.key, coords, P, GH, GW, rpb, w.rank, w.offset);
    else if (P <= 65536 && kpt <= 32)
      splat_build_reg_kernel<32><<<dim3(Q, B), BUILD_THREADS, smem, s>>>(w.key, coords, P, GH, GW, rpb, w.rank, w.offset);
    else if (P <= 65536 && kpt <= 48)
      splat_build_reg_kernel<48><<<dim3(Q, B), BUILD_THREADS, smem, s>>>(w.key, coords, P, GH, GW, rpb, w.rank, w.offset);
    else
      splat_build_kernel<<<dim3(Q, B), BUILD_THREADS, smem, s>>>(w.key, coords, P, GH, GW, rpb, w.rank, w.offset);
    CRESTE_CHECK_LAUNCH("splat_build");
    splat_fill_rec_kernel<<<g1, 256, 0, s>>>(w.key, w.rank, w.offset, coords, w.recu, BP, P, E);
    CRESTE_CHECK_LAUNCH("splat_fill_rec");
  }
  splat_sort_rec_kernel<<<dim3((E + SORT_CELLS - 1) / SORT_CELLS, B), 256, 0, s>>>(w.offset, w.recu, w.rec, P, E);
  CRESTE_CHECK_LAUNCH("splat_sort_rec");
  return CRESTE_OK;
}

extern "C" int creste_bev_splat_plan_f32(const float* xyz, int B, int P, float off_x, float off_y, float vox_x, float vox_y,
                                         int GH, int GW, float* coords, void* work, void* stream) {
  CRESTE_REQUIRE(xyz && coords && work, "bev_splat_plan: null pointer");
  CRESTE_REQUIRE(B > 0 && P > 0 && GH > 0 && GW > 0 && vox_x > 0.f && vox_y > 0.f, "bev_splat_plan: bad dims / grid");
  return splat_plan(xyz, B, P, off_x, off_y, vox_x, vox_y, GH, GW, coords, work, stream);
}

// The rest of the plan when creste_pixel_geometry_keyed_f32 has already written bev_coords and the keys (the first B*P ints of
// `work`): CSR build, record fill, per-cell sort.
extern "C" int creste_bev_splat_plan_keyed_f32(int B, int P, int GH, int GW, const float* coords, void* work, void* stream) {
  CRESTE_REQUIRE(coords && work, "bev_splat_plan_keyed: null pointer");
  CRESTE_REQUIRE(B > 0 && P > 0 && GH > 0 && GW > 0, "bev_splat_plan_keyed: bad dims / grid");
  return splat_plan(nullptr, B, P, 0.f, 0.f, 1.f, 1.f, GH, GW, const_cast<float*>(coords), work, stream);
}

// The gather over a plan (creste_bev_splat_plan_f32 of the same B, P, GH, GW into the same `work`): every BEV cell's F
// channels written once (reference splat_projection.py:320-352).
extern "C" int creste_bev_splat_gather_f32(const float* feats, int feats_cs, int B, int P, int F, int GH, int GW,
                                           float min_weight, int mode, float* bev, float* dens, void* work, void* stream) {
  CRESTE_REQUIRE(feats && bev && dens && work, "bev_splat_gather: null pointer");
  CRESTE_REQUIRE(mode == CRESTE_SPLAT_MEAN || mode == CRESTE_SPLAT_SUM || mode == CRESTE_SPLAT_MAX,
                 "bev_splat: unknown scatter mode %d", mode);
  CRESTE_REQUIRE(B > 0 && P > 0 && F > 0 && F % 4 == 0 && F <= 256 && feats_cs % 4 == 0 && feats_cs >= F,
                 "bev_splat: F must be a multiple of 4 and <= 256");
  CRESTE_REQUIRE(GH > 0 && GW > 0, "bev_splat: bad grid");
  CRESTE_REQUIRE((long)P * feats_cs < (1L << 31), "bev_splat: P * feature stride overflows the 32-bit point offset");
  const int E = (GH + 1) * (GW + 1);
  hipStream_t s = (hipStream_t)stream;
  SplatWork w = carve(work, B, P, E);
  const long ncell = (long)B * GH * GW;
  const int fq = F / 4;
  if (F % 32 == 0 && F <= 128 && (long)B * GH < (1L << 30) && (feats_cs % 4) == 0) {      // row-per-workgroup fast path
    const int rows = B * GH;
    const size_t smem = 2 * (size_t)(GW + 2) * sizeof(int);
    CRESTE_REQUIRE(smem <= 64 * 1024, "bev_splat: grid width %d too large for the offset staging", GW);
    // 4 entries in flight per lane group (2 / 3 / 6 / 8 measured within 10 % of each other, 4 best on the frustum);
    // nontemporal stores for the 403 MB output: the map is not re-read by this kernel and keeping it out of the L2
    // leaves the cache to the feature rows (4 re-reads each) -- 201 -> 176 us on the whole call
#define CRESTE_SPLAT_G8M(NQ)                                                                                              \
    do {                                                                                                                  \
      if (mode == CRESTE_SPLAT_MEAN) splat_gather8_kernel<NQ, 0, 4, true><<<rows, 256, smem, s>>>(feats, feats_cs, w.rec, w.offset, B, P, GH, GW, min_weight, bev, dens); \
      else if (mode == CRESTE_SPLAT_SUM) splat_gather8_kernel<NQ, 1, 4, true><<<rows, 256, smem, s>>>(feats, feats_cs, w.rec, w.offset, B, P, GH, GW, min_weight, bev, dens); \
      else splat_gather8_kernel<NQ, 2, 4, true><<<rows, 256, smem, s>>>(feats, feats_cs, w.rec, w.offset, B, P, GH, GW, min_weight, bev, dens); \
    } while (0)
    if (F == 32) CRESTE_SPLAT_G8M(1); else if (F == 64) CRESTE_SPLAT_G8M(2); else if (F == 96) CRESTE_SPLAT_G8M(3); else CRESTE_SPLAT_G8M(4);
#undef CRESTE_SPLAT_G8M
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


extern "C" int creste_bev_splat_mode_f32(const float* xyz, const float* feats, int feats_cs, int B, int P,
                                         int F, float off_x, float off_y, float vox_x, float vox_y, int GH,
                                         int GW, float min_weight, int mode, float* coords, float* bev,
                                         float* dens, void* work, void* stream) {
  CRESTE_REQUIRE(xyz && feats && coords && bev && dens && work, "bev_splat: null pointer");
  CRESTE_REQUIRE(mode == CRESTE_SPLAT_MEAN || mode == CRESTE_SPLAT_SUM || mode == CRESTE_SPLAT_MAX,
                 "bev_splat: unknown scatter mode %d", mode);
  CRESTE_REQUIRE(B > 0 && P > 0 && F > 0 && F % 4 == 0 && F <= 256 && feats_cs % 4 == 0 && feats_cs >= F,
                 "bev_splat: F must be a multiple of 4 and <= 256");
  const int rc = creste_bev_splat_plan_f32(xyz, B, P, off_x, off_y, vox_x, vox_y, GH, GW, coords, work, stream);
  if (rc != CRESTE_OK) return rc;
  return creste_bev_splat_gather_f32(feats, feats_cs, B, P, F, GH, GW, min_weight, mode, bev, dens, work, stream);
}

extern "C" int creste_bev_splat_f32(const float* xyz, const float* feats, int feats_cs, int B, int P,
                                    int F, float off_x, float off_y, float vox_x, float vox_y, int GH,
                                    int GW, float min_weight, float* coords, float* bev, float* dens,
                                    void* work, void* stream) {
  return creste_bev_splat_mode_f32(xyz, feats, feats_cs, B, P, F, off_x, off_y, vox_x, vox_y, GH, GW, min_weight,
                                   CRESTE_SPLAT_MEAN, coords, bev, dens, work, stream);
}
