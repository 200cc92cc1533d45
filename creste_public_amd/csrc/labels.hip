// Label bookkeeping of the supervised pixel-contrastive loss on the device.
//
// reference: creste/utils/utils.py:59-77 (`remap_labels_in_batch`: per sample, labels -> their index in the sorted
// unique list of that sample + a running offset; the ignore label kept), creste/utils/train_utils.py:324-352
// (`extract_max_per_class`: for every class in ascending order its element indices in ascending order, at most
// max_per_class of them chosen by a host-side torch.randperm) and creste/utils/loss_utils.py:203-286 (boolean-mask
// gathers of the valid cells).  The reference runs these as Python loops over samples, labels and classes: hundreds of
// launches and host round trips per step.  Here: presence tables + scans for the remap; a STABLE grouping of the valid
// cells by class (chunk histograms -> scan -> in-order ranks, one wave per chunk) so that "the r-th cell of class c in
// row-major order" is an array lookup -- the host only draws the permutations (same generator, same order as the
// reference) and uploads (class, rank) pairs.
#include "common.h"

namespace creste {

__global__ __launch_bounds__(256) void label_minmax_kernel(const int64_t* __restrict__ x, long n, long long* __restrict__ out2) {
  long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long long v = x[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const long long a = __shfl_xor(lo, o, 64), b = __shfl_xor(hi, o, 64);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&out2[0], lo);
    atomicMax(&out2[1], hi);
  }
}

__global__ __launch_bounds__(256) void label_mark_kernel(const int64_t* __restrict__ gt, long HW, int L, int* __restrict__ table) {
  const int b = blockIdx.y;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < HW; i += (long)gridDim.x * 256) table[(long)b * L + gt[(long)b * HW + i]] = 1;
}

// one workgroup per sample: rank of every present label among the sample's present labels (ascending) -> table (absent
// labels and the ignore label: -1), and the sample's count of present non-ignore labels / its largest rank
__global__ __launch_bounds__(1024) void label_rank_kernel(int* __restrict__ table, int L, int ignore, int* __restrict__ nonign,
                                                          int* __restrict__ toprank) {
  __shared__ int s_w[16];
  __shared__ int s_carry, s_nonign, s_top;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) { s_carry = 0; s_nonign = 0; s_top = -1; }
  __syncthreads();
  int* tb = table + (long)b * L;
  int mytop = -1, mine = 0;
  for (int base = 0; base < L; base += 1024) {
    const int l = base + t;
    const int v = l < L ? tb[l] : 0;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(inc, d); if (lane >= d) inc += up; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int pre = s_carry;
    for (int k = 0; k < w; ++k) pre += s_w[k];
    if (l < L) {
      int r = -1;
      if (v && l != ignore) { r = pre + inc - v; mytop = max(mytop, r); ++mine; }   // index in the sorted unique list
      tb[l] = r;
    }
    __syncthreads();
    if (t == 1023) s_carry = pre + inc;
    __syncthreads();
  }
  atomicAdd(&s_nonign, mine);
  atomicMax(&s_top, mytop);
  __syncthreads();
  if (t == 0) { nonign[b] = s_nonign; toprank[b] = s_top; }
}

// new label = rank + running offset (offset of sample b = present non-ignore labels of the samples before it:
// utils.py:66-76); the ignore label stays.  Thread 0 of block (0,0) also publishes the class count.
__global__ __launch_bounds__(256) void label_map_kernel(const int64_t* __restrict__ gt, long HW, int L, int B, int ignore,
                                                        const int* __restrict__ table, const int* __restrict__ nonign,
                                                        const int* __restrict__ toprank, int64_t* __restrict__ out,
                                                        int* __restrict__ nclass) {
  const int b = blockIdx.y;
  int off = 0;
  for (int k = 0; k < b; ++k) off += nonign[k];
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
    int top = ignore, o = 0;
    for (int k = 0; k < B; ++k) { if (toprank[k] >= 0) top = max(top, toprank[k] + o); o += nonign[k]; }
    *nclass = top + 1;
  }
  for (long i = blockIdx.x * 256L + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
    const int r = table[(long)b * L + gt[(long)b * HW + i]];
    out[(long)b * HW + i] = r < 0 ? ignore : r + off;
  }
}

// ---- stable grouping of the valid cells by class
constexpr int GRP_CHUNK = 2048;

__global__ __launch_bounds__(256) void group_hist_kernel(const int64_t* __restrict__ lab, const uint8_t* __restrict__ fov,
                                                         long n, int K, int ignore, int* __restrict__ chunk_hist) {
  extern __shared__ int s_h[];
  for (int k = threadIdx.x; k < K; k += 256) s_h[k] = 0;
  __syncthreads();
  const long i0 = (long)blockIdx.x * GRP_CHUNK;
  for (int j = threadIdx.x; j < GRP_CHUNK; j += 256) {
    const long i = i0 + j;
    if (i < n) {
      const long long c = lab[i];
      if (c != ignore && c >= 0 && c < K && (!fov || fov[i])) atomicAdd(&s_h[(int)c], 1);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) chunk_hist[(long)blockIdx.x * K + k] = s_h[k];
}

// per class: exclusive scan over the chunks (in place) and the class total
__global__ __launch_bounds__(256) void group_scan_chunks_kernel(int* __restrict__ chunk_hist, int nchunk, int K, int* __restrict__ counts) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  int run = 0;
  for (int c = 0; c < nchunk; ++c) {
    const int v = chunk_hist[(long)c * K + k];
    chunk_hist[(long)c * K + k] = run;
    run += v;
  }
  counts[k] = run;
}

__global__ __launch_bounds__(1024) void group_offsets_kernel(const int* __restrict__ counts, int K, int* __restrict__ offsets) {
  __shared__ int s_w[16];
  __shared__ int s_carry;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < K; base += 1024) {
    const int k = base + t;
    const int v = k < K ? counts[k] : 0;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(inc, d); if (lane >= d) inc += up; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int pre = s_carry;
    for (int j = 0; j < w; ++j) pre += s_w[j];
    if (k < K) offsets[k] = pre + inc - v;
    __syncthreads();
    if (t == 1023) s_carry = pre + inc;
    __syncthreads();
  }
  if (t == 0) offsets[K] = s_carry;
}

// one wave per chunk, cells in row-major order: a cell's slot = class offset + cells of its class in earlier chunks
// + cells of its class earlier in this chunk (running LDS counters + earlier lanes of the wave with the same class)
__global__ __launch_bounds__(64) void group_fill_kernel(const int64_t* __restrict__ lab, const uint8_t* __restrict__ fov,
                                                        long n, int K, int ignore, const int* __restrict__ chunk_base,
                                                        const int* __restrict__ offsets, int* __restrict__ class_list) {
  extern __shared__ int s_cnt[];
  const int lane = threadIdx.x;
  for (int k = lane; k < K; k += 64) s_cnt[k] = chunk_base[(long)blockIdx.x * K + k] + offsets[k];
  __syncthreads();
  const long i0 = (long)blockIdx.x * GRP_CHUNK;
  for (int j0 = 0; j0 < GRP_CHUNK; j0 += 64) {
    const long i = i0 + j0 + lane;
    int c = -1;
    if (i < n) {
      const long long v = lab[i];
      if (v != ignore && v >= 0 && v < K && (!fov || fov[i])) c = (int)v;
    }
    int before = 0, total = 0;
    for (int k = 0; k < 64; ++k) {                      // same-class lanes: earlier ones order me, all of them count
      const int ck = __shfl(c, k, 64);
      const bool same = ck == c;
      before += (same && k < lane) ? 1 : 0;
      total += same ? 1 : 0;
    }
    int base = 0;
    if (c >= 0) base = s_cnt[c];
    __syncthreads();
    if (c >= 0) {
      class_list[base + before] = (int)i;
      if (before == total - 1) s_cnt[c] = base + total;       // the last lane of the class advances the counter
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void pick_cells_kernel(const int* __restrict__ class_list, const int* __restrict__ offsets,
                                                         const int* __restrict__ sel_cls, const int* __restrict__ sel_rank, int S,
                                                         int* __restrict__ cell) {
  for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256)
    cell[s] = class_list[offsets[sel_cls[s]] + sel_rank[s]];
}

// rows: out[s][0..Z) = src[cell[s]][0..Z) (gather) ; dst[cell[s]][0..Z) = g[s][0..Z) (scatter; cells are distinct)
template <bool SCATTER>
__global__ __launch_bounds__(256) void rows_kernel(float* __restrict__ grid, int cs, int Z, const int* __restrict__ cell, long S,
                                                   float* __restrict__ rows) {
  const long total = S * Z;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long s = e / Z;
    const int z = (int)(e - s * Z);
    if (SCATTER) grid[(long)cell[s] * cs + z] = rows[e];
    else rows[e] = grid[(long)cell[s] * cs + z];
  }
}

}  // namespace creste

using namespace creste;

static inline int lgrid(long n, int cap = 2048) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

extern "C" int creste_label_minmax_i64(const int64_t* labels, int64_t n, int64_t* out2, void* stream) {
  CRESTE_REQUIRE(labels && out2 && n > 0, "label_minmax: bad args");
  hipStream_t s = (hipStream_t)stream;
  const long long init[2] = {0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1};
  CRESTE_HIP(hipMemcpyAsync(out2, init, sizeof(init), hipMemcpyHostToDevice, s));
  label_minmax_kernel<<<lgrid(n, 512), 256, 0, s>>>(labels, n, (long long*)out2);
  CRESTE_CHECK_LAUNCH("label_minmax");
  return CRESTE_OK;
}

extern "C" int creste_remap_labels_i64(const int64_t* gt, int B, int64_t HW, int ignore_idx, int L, int* table,
                                       int64_t* out, int* nclass, void* stream) {
  CRESTE_REQUIRE(gt && table && out && nclass && B > 0 && HW > 0 && L > 0, "remap_labels: bad args");
  CRESTE_REQUIRE(ignore_idx >= 0 && ignore_idx < L, "remap_labels: ignore index %d outside [0, %d)", ignore_idx, L);
  hipStream_t s = (hipStream_t)stream;
  CRESTE_HIP(hipMemsetAsync(table, 0, ((size_t)B * L + 2 * (size_t)B) * sizeof(int), s));
  int* nonign = table + (size_t)B * L;
  int* toprank = nonign + B;
  label_mark_kernel<<<dim3(lgrid(HW, 256), B), 256, 0, s>>>(gt, HW, L, table);
  label_rank_kernel<<<B, 1024, 0, s>>>(table, L, ignore_idx, nonign, toprank);
  label_map_kernel<<<dim3(lgrid(HW, 256), B), 256, 0, s>>>(gt, HW, L, B, ignore_idx, table, nonign, toprank, out, nclass);
  CRESTE_CHECK_LAUNCH("remap_labels");
  return CRESTE_OK;
}

extern "C" int64_t creste_group_by_class_workspace_bytes(int64_t n, int K) {
  if (n <= 0 || K <= 0) return -1;
  return (int64_t)((n + GRP_CHUNK - 1) / GRP_CHUNK) * K * 4;
}

extern "C" int creste_group_by_class_i64(const int64_t* labels, const uint8_t* fov, int64_t n, int K, int ignore_idx,
                                         int* counts, int* offsets, int* class_list, void* work, void* stream) {
  CRESTE_REQUIRE(labels && counts && offsets && class_list && work && n > 0, "group_by_class: bad args");
  CRESTE_REQUIRE(K > 0 && K <= 16384 && n < (1L << 31), "group_by_class: %d classes (1..16384), n < 2^31", K);
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = (int)((n + GRP_CHUNK - 1) / GRP_CHUNK);
  int* ch = (int*)work;
  const size_t smem = (size_t)K * sizeof(int);
  if (smem > 64 * 1024) {
    CRESTE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(group_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CRESTE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(group_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  }
  group_hist_kernel<<<nchunk, 256, smem, s>>>(labels, fov, n, K, ignore_idx, ch);
  group_scan_chunks_kernel<<<(K + 255) / 256, 256, 0, s>>>(ch, nchunk, K, counts);
  group_offsets_kernel<<<1, 1024, 0, s>>>(counts, K, offsets);
  group_fill_kernel<<<nchunk, 64, smem, s>>>(labels, fov, n, K, ignore_idx, ch, offsets, class_list);
  CRESTE_CHECK_LAUNCH("group_by_class");
  return CRESTE_OK;
}

extern "C" int creste_pick_cells_i32(const int* class_list, const int* offsets, const int* sel_cls, const int* sel_rank,
                                     int S, int* cell, void* stream) {
  CRESTE_REQUIRE(class_list && offsets && sel_cls && sel_rank && cell && S > 0, "pick_cells: bad args");
  pick_cells_kernel<<<lgrid(S, 256), 256, 0, (hipStream_t)stream>>>(class_list, offsets, sel_cls, sel_rank, S, cell);
  CRESTE_CHECK_LAUNCH("pick_cells");
  return CRESTE_OK;
}

extern "C" int creste_gather_rows_f32(const float* grid, int cs, int Z, const int* cell, int64_t S, float* rows, void* stream) {
  CRESTE_REQUIRE(grid && cell && rows && S > 0 && Z > 0 && cs >= Z, "gather_rows: bad args");
  rows_kernel<false><<<lgrid(S * Z), 256, 0, (hipStream_t)stream>>>(const_cast<float*>(grid), cs, Z, cell, S, rows);
  CRESTE_CHECK_LAUNCH("gather_rows");
  return CRESTE_OK;
}

extern "C" int creste_scatter_rows_f32(const float* rows, const int* cell, int64_t S, int Z, float* grid, int cs, void* stream) {
  CRESTE_REQUIRE(grid && cell && rows && S > 0 && Z > 0 && cs >= Z, "scatter_rows: bad args");
  rows_kernel<true><<<lgrid(S * Z), 256, 0, (hipStream_t)stream>>>(grid, cs, Z, cell, S, const_cast<float*>(rows));
  CRESTE_CHECK_LAUNCH("scatter_rows");
  return CRESTE_OK;
}
