// Value iteration on the 8-connected stochastic grid MDP (reference creste/models/blocks/vin.py:36-80).
//
//   v <- 0 ; repeat { x = r + gamma*v ; q_a = sum_taps w_a[tap]*x[neighbour] (zero outside the grid) ;
//   v' = max_a q_a ; delta = max_{batch,grid} |v' - v| ; v <- v' } until !(delta > threshold) ;
//   q = eval_q(r, v) ; policy = softmax_a(q)          (Jacobi sweeps, hard max, batch-global test)
//
// HBM-bound in principle (12 B/cell/sweep) but the whole state (r, v: 0.8 MB for 8x64x128, 6 MB for
// 8x256x256) is cache resident, so the cost of a naive port is launch/sync latency (the reference does
// one `.item()` host sync per sweep, ~690 of them).  Here the sweeps are temporally blocked in LDS
// (below) and the batch-global convergence test is a device-side max (atomicMax on the float bits,
// deltas are >= 0) checked across kernel boundaries; the host only peeks at a `done` flag every few
// launches.
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace creste {

// Action a moves by DYN[a] = (drow, dcol); its 0.8 tap sits at window position 1+DYN[a]; the two
// 0.1 taps are the ring neighbours of that position (vin.py:36-46).  Taps are listed in row-major
// window order -- the accumulation order of a direct 3x3 cross-correlation.
struct Tap { int dy, dx; float w; };
constexpr Tap kTaps[8][3] = {
    {{-1, -1, 0.8f}, {-1, 0, 0.1f}, {0, -1, 0.1f}},   // a0 NW
    {{-1, -1, 0.1f}, {-1, 0, 0.8f}, {-1, 1, 0.1f}},   // a1 N
    {{-1, 0, 0.1f}, {-1, 1, 0.8f}, {0, 1, 0.1f}},     // a2 NE
    {{-1, -1, 0.1f}, {0, -1, 0.8f}, {1, -1, 0.1f}},   // a3 W
    {{-1, 1, 0.1f}, {0, 1, 0.8f}, {1, 1, 0.1f}},      // a4 E
    {{0, -1, 0.1f}, {1, -1, 0.8f}, {1, 0, 0.1f}},     // a5 SW
    {{1, -1, 0.1f}, {1, 0, 0.8f}, {1, 1, 0.1f}},      // a6 S
    {{0, 1, 0.1f}, {1, 0, 0.1f}, {1, 1, 0.8f}},       // a7 SE
};

constexpr int TW = 64, TH = 16;   // tile of cells per 256-thread workgroup (4 rows per thread)

__device__ __forceinline__ void q_values(const float (*xs)[TW + 2], int ly, int lx, float q[8]) {
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
      s = __fmaf_rn(kTaps[a][t].w, xs[ly + 1 + kTaps[a][t].dy][lx + 1 + kTaps[a][t].dx], s);
    q[a] = s;
  }
}

__device__ __forceinline__ void load_x_tile(float (*xs)[TW + 2], const float* __restrict__ r,
                                            const float* __restrict__ v, int H, int W, int y0, int x0,
                                            float gamma) {
  for (int i = threadIdx.x; i < (TH + 2) * (TW + 2); i += 256) {
    const int ly = i / (TW + 2), lx = i % (TW + 2);
    const int y = y0 + ly - 1, x = x0 + lx - 1;
    float val = 0.f;                                    // zero padding of the conv
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
      val = __fadd_rn(r[(long)y * W + x], __fmul_rn(v[(long)y * W + x], gamma));
    xs[ly][lx] = val;
  }
}

// ---------------------------------------------------------------------------------------------
// Temporally blocked sweeps: one launch advances every tile by S Jacobi sweeps out of LDS.
//   * a workgroup owns a TH x TW tile and loads it with a halo of S cells (cells beyond the GRID are
//     exact zeros = the conv's zero padding, so a tile that spans the whole grid needs no halo and S
//     can be large); every loaded cell is updated each sweep, cells within j of a non-border region
//     edge are stale after j sweeps and never reach the tile interior within S sweeps;
//   * a thread keeps r and v of its <= 8 cells in registers, LDS holds x = r + gamma*v with a zero frame;
//     two barriers per sweep, no global traffic between sweeps;
//   * sweep k's batch-global delta is an atomicMax over the tile interiors, as before.  Launch l checks
//     the deltas of chunk l-1 after the kernel boundary; if the test failed at sweep j of that chunk,
//     launch l REDOES chunk l-1 from its (untouched) input buffer with exactly j+1 sweeps, publishes
//     `done` = l+1, and every later launch is a no-op.  The result is bit-identical to single-sweep launches.
constexpr int VM_THREADS = 1024, VM_CPT = 8;     // region <= 8192 cells

struct VmState { int done, converged_at, final_buf, pad; };

__global__ __launch_bounds__(VM_THREADS) void vi_multi_kernel(const float* __restrict__ r, float* buf0,
                                                             float* buf1, VmState* st, unsigned* delta,
                                                             int l, int S, int H, int W, int TH, int TW,
                                                             int halo, float gamma, float threshold) {
  extern __shared__ float lds[];
  __shared__ int s_mode[2];
  // `done` holds (index of the launch that published it) + 1: only LATER launches may skip -- a block of the redo
  // launch itself that is dispatched after block 0 retired must still redo its tile
  { const int d = st->done; if (d != 0 && d <= l) return; }
  const int tid = threadIdx.x;
  if (tid == 0) {
    int first = -1;
    if (l > 0)
      for (int j = 0; j < S; ++j)
        if (!(__uint_as_float(delta[(l - 1) * S + j]) > threshold)) { first = j; break; }
    s_mode[0] = first;
  }
  __syncthreads();
  const int first = s_mode[0];
  const bool redo = first >= 0;
  const int nsweeps = redo ? first + 1 : S;
  const float* src = redo ? (((l - 1) & 1) ? buf1 : buf0) : ((l & 1) ? buf1 : buf0);
  float* dst = redo ? ((l & 1) ? buf1 : buf0) : (((l + 1) & 1) ? buf1 : buf0);

  const int b = blockIdx.z, ty0 = blockIdx.y * TH, tx0 = blockIdx.x * TW;
  const int ry0 = max(0, ty0 - halo), ry1 = min(H, ty0 + TH + halo);     // loaded region, clipped to the grid
  const int rx0 = max(0, tx0 - halo), rx1 = min(W, tx0 + TW + halo);
  const int RH = ry1 - ry0, RW = rx1 - rx0, RN = RH * RW, LW = RW + 2;
  float* xs = lds;                                  // (RH+2) x (RW+2), zero frame
  float* red = lds + (RH + 2) * LW;                 // [S][16] per-wave maxima
  const long plane = (long)b * H * W;

  for (int i = tid; i < (RH + 2) * LW; i += VM_THREADS) xs[i] = 0.f;
  float rr[VM_CPT], vv[VM_CPT];
  int li[VM_CPT];                                   // LDS index of the cell, -1 if none
  bool interior[VM_CPT];
#pragma unroll
  for (int c = 0; c < VM_CPT; ++c) {
    const int i = tid + c * VM_THREADS;
    li[c] = -1; interior[c] = false; rr[c] = 0.f; vv[c] = 0.f;
    if (i < RN) {
      const int y = ry0 + i / RW, x = rx0 + i % RW;
      li[c] = (y - ry0 + 1) * LW + (x - rx0 + 1);
      rr[c] = r[plane + (long)y * W + x];
      vv[c] = src[plane + (long)y * W + x];
      interior[c] = y >= ty0 && y < ty0 + TH && x >= tx0 && x < tx0 + TW;
    }
  }
  __syncthreads();                                  // frame zeroed
#pragma unroll
  for (int c = 0; c < VM_CPT; ++c)
    if (li[c] >= 0) xs[li[c]] = __fadd_rn(rr[c], __fmul_rn(vv[c], gamma));
  __syncthreads();

  for (int j = 0; j < nsweeps; ++j) {
    float dmax = 0.f;
#pragma unroll
    for (int c = 0; c < VM_CPT; ++c) {
      if (li[c] < 0) continue;
      const float* ctr = xs + li[c];
      float m = -3.0e38f;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) s = __fmaf_rn(kTaps[a][t].w, ctr[kTaps[a][t].dy * LW + kTaps[a][t].dx], s);
        m = fmaxf(m, s);
      }
      if (interior[c]) dmax = fmaxf(dmax, fabsf(__fsub_rn(m, vv[c])));
      vv[c] = m;
    }
    if (!redo) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
      if ((tid & 63) == 0) red[j * 16 + (tid >> 6)] = dmax;
    }
    __syncthreads();                                // every thread has read the old x
#pragma unroll
    for (int c = 0; c < VM_CPT; ++c)
      if (li[c] >= 0) xs[li[c]] = __fadd_rn(rr[c], __fmul_rn(vv[c], gamma));
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < VM_CPT; ++c)
    if (interior[c]) {
      const int i = tid + c * VM_THREADS;
      dst[plane + (long)(ry0 + i / RW) * W + (rx0 + i % RW)] = vv[c];
    }
  if (!redo && tid < nsweeps) {
    float m = 0.f;
    for (int w = 0; w < 16; ++w) m = fmaxf(m, red[tid * 16 + w]);
    atomicMax(&delta[l * S + tid], __float_as_uint(m));
  }
  if (redo && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    st->converged_at = (l - 1) * S + first + 1;
    st->final_buf = l & 1;
    __threadfence();
    st->done = l + 1;
  }
}

// ---------------------------------------------------------------------------------------------
// Persistent solver: ONE launch runs the whole iteration.  (The chunk-per-launch form above needs the host to peek at
// `done` every few launches -- a stream synchronisation inside a call the header promises to be asynchronous, and a
// forward that cannot be captured into a hipGraph -- and its 64 x 64 tiles put 128 workgroups on 256 CUs.)
//   * a workgroup owns a TH x TW tile for the whole solve; r, v and the chunk-start copy of v of its (TH+2S) x (TW+2S)
//     region live in registers (a thread owns 1 x 4 strips of cells), x = r + gamma*v in two LDS planes with a zero frame
//     (one barrier per sweep: sweep j reads plane j&1 and writes the other);
//   * per chunk of S sweeps the workgroups exchange their tile interiors through a ping-pong pair of global planes and
//     meet at a device-scope barrier (all workgroups are co-resident: checked at launch); the halo ring is re-read from
//     the neighbours' interiors, the interior itself never leaves the registers;
//   * sweep k's batch-global delta is an atomicMax over tile interiors as before; after the barrier every workgroup
//     reads the chunk's S deltas and, if the test failed first at sweep j, restores its chunk-start values and redoes
//     exactly j+1 sweeps -- the same arithmetic and the same sweep count as sweep-per-launch execution.
// Tiles: 32 x 32 (+8) for the large grids (512 workgroups at 8 x 256 x 256), 16 x 16 (+8) when that leaves CUs idle
// (256 workgroups at the reference's 8 x 64 x 128, which the tile-per-sample form ran on 8).
constexpr int VI_MAXW = 16;        // waves per workgroup of the persistent solver (<= 1024 threads)
constexpr unsigned VI_ABORT = 0xffffffffu;          // verdict word: a workgroup gave up waiting (launch not co-resident)
constexpr int VI_SPIN_LIMIT = 1 << 21;              // polls (~0.1 ... 1 us each) before a waiter declares the launch stuck
struct ViPArgs {
  const float* r;
  float* vbuf0;
  float* vbuf1;
  VmState* st;
  unsigned* arrive;      // [nwg] chunk count each workgroup has finished
  unsigned* go;          // ((chunk + 1) << 8) | (first failing sweep + 1, or 0): the master's verdict for that chunk
  float* dwg;            // [2][nwg][S] per-workgroup interior deltas of the chunk (parity-indexed)
  int B, H, W, TH, TW, S, tiles_x, tiles_y, nwg, max_chunks;
  float gamma, thr;
};

// Device-scope rendezvous of the co-resident workgroups, once per chunk.  Everything exchanged (tile interiors, deltas,
// flags) is written and read with agent-scope (sc1) accesses that go to the device's coherence point, so no cache
// maintenance is needed: an agent-scope release / acquire fence writes back / invalidates the XCD's whole 4 MB L2, and
// with ~400 of them per chunk and XCD the first version of this kernel spent 130 us per chunk there.  No read-modify-
// write either: 512 workgroups incrementing one counter and atomicMax-ing eight delta words serialise at the memory
// side (~15 us and ~25 us per chunk).  Instead every workgroup STORES its S deltas and an arrival flag; workgroup 0
// polls the flags (one coalesced load per pass), reduces the deltas, applies the convergence test and publishes the
// verdict in the word everyone else polls.
// Every wait is bounded: a waiter that has polled VI_SPIN_LIMIT times publishes VI_ABORT in the verdict word and everybody
// leaves (-> -2; the launch was not co-resident, e.g. two solves on two streams or a graph replay next to a full-chip
// kernel that never drains; *sweeps_out becomes INT32_MIN).  Floor of the protocol, measured with empty chunks
// (profiles/r04_value_iteration.md): ~6 us at 256 workgroups, ~10 us at 512 -- four dependent trips to the coherence point.
template <int S>
__device__ __forceinline__ int vi_rendezvous(const ViPArgs& p, int chunk, float* red, int* s_first) {
  const int tid = threadIdx.x, nt = blockDim.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's sc1 stores have reached the coherence point
  __syncthreads();
  if (tid == 0) __hip_atomic_store(p.arrive + blockIdx.x, (unsigned)(chunk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool stuck = false;
  if (blockIdx.x == 0) {
    for (int spin = 0;; ++spin) {
      int ok = 1;
      for (int i = tid; i < p.nwg; i += nt)
        ok &= __hip_atomic_load(p.arrive + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(chunk + 1);
      // (spin is uniform; the other waiters' abort word is looked at once per 1024 passes, by every thread alike)
      if ((spin & 1023) == 1023 && __hip_atomic_load(p.go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == VI_ABORT) ok = -1;
      const int all = __syncthreads_and(ok == 1), any_abort = __syncthreads_or(ok < 0);
      if (all) break;
      if (any_abort || spin > VI_SPIN_LIMIT) { stuck = true; break; }
    }
    float m[S];
#pragma unroll
    for (int j = 0; j < S; ++j) m[j] = 0.f;
    const float* d = p.dwg + (size_t)(chunk & 1) * p.nwg * S;
    for (int i = tid; i < p.nwg; i += nt)
#pragma unroll
      for (int j = 0; j < S; ++j) m[j] = fmaxf(m[j], __hip_atomic_load(d + (size_t)i * S + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int j = 0; j < S; ++j) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m[j] = fmaxf(m[j], __shfl_xor(m[j], o));
      if ((tid & 63) == 0) red[j * VI_MAXW + (tid >> 6)] = m[j];
    }
    __syncthreads();
    if (tid == 0) {
      int first = -1;
      for (int j = 0; j < S && first < 0; ++j) {
        float mm = 0.f;
        for (int w = 0; w < (nt + 63) / 64; ++w) mm = fmaxf(mm, red[j * VI_MAXW + w]);
        if (!(mm > p.thr)) first = j;
      }
      __hip_atomic_store(p.go, stuck ? VI_ABORT : (((unsigned)(chunk + 1) << 8) | (unsigned)(first + 1)), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) {
    unsigned g;
    int spin = 0;
    while (((g = __hip_atomic_load(p.go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 8) != (unsigned)(chunk + 1) && g != VI_ABORT) {
      if (++spin > VI_SPIN_LIMIT) {
        __hip_atomic_store(p.go, VI_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g = VI_ABORT;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    *s_first = g == VI_ABORT ? -2 : (int)(g & 255u) - 1;
  }
  __syncthreads();
  return *s_first;
}

#ifdef VI_TRACE
__device__ long long g_vi_trace[8];      // {chunks, sweeps + store, rendezvous, halo} cycles per chunk of workgroups 0 and 100
#endif
// (launch bounds: 6 waves per SIMD = at most 80 registers, so that two 9-wave workgroups fit a CU even when both put three
// waves on the same SIMD -- see the residency note at the launch.)
template <int NSTRIP>
__global__ __launch_bounds__(NSTRIP == 1 ? 576 : 256, NSTRIP == 1 ? 6 : 4) void vi_persist_kernel(const ViPArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_first;
  const int S = p.S, RH = p.TH + 2 * S, RW = p.TW + 2 * S, LW = RW + 8, SW = RW >> 2, nstr = RH * SW;
  const int plane = (RH + 2) * LW;
  float* const xb[2] = {lds, lds + plane};
  float* const red = lds + 2 * plane;                 // [S][VI_MAXW] per-wave maxima
  const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6;
  int id = blockIdx.x;
  const int tx = id % p.tiles_x; id /= p.tiles_x;
  const int ty = id % p.tiles_y;
  const int b = id / p.tiles_y;
  const int gy0 = ty * p.TH - S, gx0 = tx * p.TW - S;
  const long pl = (long)b * p.H * p.W;

  float rr[NSTRIP][4], vv[NSTRIP][4], vs[NSTRIP][4];
  int li[NSTRIP], gofs[NSTRIP];          // LDS index of the strip's first cell (-1: none), global offset of that cell
  unsigned inb[NSTRIP], inner[NSTRIP];   // 4-bit masks: cell inside the grid / inside this workgroup's tile
#pragma unroll
  for (int k = 0; k < NSTRIP; ++k) {
    const int s = tid + k * nt;
    li[k] = -1; inb[k] = inner[k] = 0; gofs[k] = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { rr[k][e] = 0.f; vv[k][e] = 0.f; vs[k][e] = 0.f; }
    if (s < nstr) {
      const int row = s / SW, c0 = (s - row * SW) * 4;
      const int gy = gy0 + row, gx = gx0 + c0;
      li[k] = (row + 1) * LW + 4 + c0;
      gofs[k] = gy * p.W + gx;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = (unsigned)gy < (unsigned)p.H && (unsigned)(gx + e) < (unsigned)p.W;
        if (in) { inb[k] |= 1u << e; rr[k][e] = p.r[pl + gofs[k] + e]; }
        if (in && row >= S && row < S + p.TH && c0 + e >= S && c0 + e < S + p.TW) inner[k] |= 1u << e;
      }
    }
  }
  for (int i = tid; i < 2 * plane; i += nt) lds[i] = 0.f;     // zero frames (and out-of-grid cells) of both planes
  __syncthreads();

  auto put_x = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NSTRIP; ++k)
      if (li[k] >= 0) {
        float4 x;
        x.x = (inb[k] & 1) ? __fadd_rn(rr[k][0], __fmul_rn(vv[k][0], p.gamma)) : 0.f;
        x.y = (inb[k] & 2) ? __fadd_rn(rr[k][1], __fmul_rn(vv[k][1], p.gamma)) : 0.f;
        x.z = (inb[k] & 4) ? __fadd_rn(rr[k][2], __fmul_rn(vv[k][2], p.gamma)) : 0.f;
        x.w = (inb[k] & 8) ? __fadd_rn(rr[k][3], __fmul_rn(vv[k][3], p.gamma)) : 0.f;
        *reinterpret_cast<float4*>(dst + li[k]) = x;
      }
  };
  // one Jacobi sweep of the whole region: plane `src` -> registers + plane `dst`; returns this thread's interior delta
  auto sweep = [&](const float* src, float* dst) __attribute__((always_inline)) -> float {
    float dmax = 0.f;
#pragma unroll
    for (int k = 0; k < NSTRIP; ++k) {
      if (li[k] < 0) continue;
      float w[3][6];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const float* row = src + li[k] + (dy - 1) * LW;
        const float4 c = *reinterpret_cast<const float4*>(row);
        w[dy][0] = row[-1]; w[dy][1] = c.x; w[dy][2] = c.y; w[dy][3] = c.z; w[dy][4] = c.w; w[dy][5] = row[4];
      }
      // action-major: the four cells' fma chains of one action are independent and issue back to back (a cell-major
      // order leaves each 3-deep chain waiting on its own result when there is one wave per SIMD)
      float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        float sacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            sacc[e] = __fmaf_rn(kTaps[a][t].w, w[1 + kTaps[a][t].dy][e + 1 + kTaps[a][t].dx], sacc[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], sacc[e]);
      }
      float xn[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m = mx[e];
        if (inner[k] & (1u << e)) dmax = fmaxf(dmax, fabsf(__fsub_rn(m, vv[k][e])));
        vv[k][e] = m;
        xn[e] = (inb[k] & (1u << e)) ? __fadd_rn(rr[k][e], __fmul_rn(m, p.gamma)) : 0.f;
      }
      *reinterpret_cast<float4*>(dst + li[k]) = make_float4(xn[0], xn[1], xn[2], xn[3]);
    }
    return dmax;
  };
  auto store_interior = [&](float* vdst) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NSTRIP; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (inner[k] & (1u << e)) __hip_atomic_store(vdst + pl + gofs[k] + e, vv[k][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

#ifdef VI_TRACE
  long long t_sw = 0, t_rv = 0, t_halo = 0, tA, tB, tC;
#endif
  for (int chunk = 0; chunk < p.max_chunks; ++chunk) {
    float* const vnext = ((chunk + 1) & 1) ? p.vbuf1 : p.vbuf0;
#ifdef VI_TRACE
    tA = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int k = 0; k < NSTRIP; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) vs[k][e] = vv[k][e];
    put_x(xb[0]);
    __syncthreads();
    // per-thread deltas of the chunk's sweeps stay in registers; ONE wave reduction per chunk (a reduction per sweep put six
    // dependent cross-lane steps, ~700 cycles, on every sweep's critical path)
    float dm[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dm[j] = sweep(xb[j & 1], xb[(j + 1) & 1]);
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dm[j] = fmaxf(dm[j], __shfl_xor(dm[j], o));
      if ((tid & 63) == 0) red[j * VI_MAXW + wave] = dm[j];
    }
    __syncthreads();
    if (tid < S) {
      float m = 0.f;
      for (int w = 0; w < (nt + 63) / 64; ++w) m = fmaxf(m, red[tid * VI_MAXW + w]);
      __hip_atomic_store(p.dwg + ((size_t)(chunk & 1) * p.nwg + blockIdx.x) * S + tid, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    store_interior(vnext);
#ifdef VI_TRACE
    tB = __builtin_readcyclecounter();
#endif
    const int first = vi_rendezvous<8>(p, chunk, red, &s_first);
#ifdef VI_TRACE
    tC = __builtin_readcyclecounter();
    t_sw += tB - tA; t_rv += tC - tB;
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {      // (no printf: its host service would wait on this
      long long* o = g_vi_trace + (blockIdx.x ? 4 : 0);              //  spinning kernel)
      o[0] = chunk + 1; o[1] = t_sw / (chunk + 1); o[2] = t_rv / (chunk + 1); o[3] = t_halo / (chunk + 1);
    }
#endif
    if (first == -2) {                   // somebody gave up waiting: the launch is not co-resident
      if (blockIdx.x == 0 && tid == 0) p.st->done = -1;
      return;
    }
    if (first >= 0) {
      // the test failed first at sweep `first` of this chunk: the answer is the chunk-start state advanced first+1 sweeps
#pragma unroll
      for (int k = 0; k < NSTRIP; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[k][e] = vs[k][e];
      put_x(xb[0]);
      __syncthreads();
      for (int j = 0; j <= first; ++j) {
        (void)sweep(xb[j & 1], xb[(j + 1) & 1]);
        __syncthreads();
      }
      store_interior(vnext);
      if (blockIdx.x == 0 && tid == 0) {
        p.st->converged_at = chunk * S + first + 1;
        p.st->final_buf = (chunk + 1) & 1;
        __threadfence();
        p.st->done = chunk + 1;
      }
      return;
    }
    // halo ring <- the neighbours' fresh interiors (this tile's own interior is already in registers)
#pragma unroll
    for (int k = 0; k < NSTRIP; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((inb[k] & ~inner[k]) & (1u << e))
          vv[k][e] = __hip_atomic_load(vnext + pl + gofs[k] + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef VI_TRACE
    t_halo += __builtin_readcyclecounter() - tC;
#endif
  }
}

__global__ __launch_bounds__(256) void vi_final2_kernel(const float* __restrict__ r,
                                                        const float* __restrict__ buf0,
                                                        const float* __restrict__ buf1, const VmState* st,
                                                        int last_buf, int sweeps_run, int H, int W, float gamma,
                                                        float* __restrict__ v_out, float* __restrict__ q_out,
                                                        float* __restrict__ pi_out, int32_t* sweeps_out) {
  __shared__ float xs[TH + 2][TW + 2];
  const bool aborted = st->done < 0;          // the persistent solver gave up waiting for workgroups that never became resident
  const bool done = st->done > 0;
  const float* v = (done ? st->final_buf : last_buf) ? buf1 : buf0;
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const long plane = (long)b * H * W;
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    *sweeps_out = aborted ? INT32_MIN : (done ? st->converged_at : -sweeps_run);
  load_x_tile(xs, r + plane, v + plane, H, W, y0, x0, gamma);
  __syncthreads();
  const int lx = threadIdx.x & 63, lyb = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ly = lyb * 4 + j, y = y0 + ly, x = x0 + lx;
    if (y < H && x < W) {
      float q[8];
      q_values(xs, ly, lx, q);
      float m = q[0];
#pragma unroll
      for (int a = 1; a < 8; ++a) m = fmaxf(m, q[a]);
      float e[8], s = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) { e[a] = expf(__fsub_rn(q[a], m)); s = __fadd_rn(s, e[a]); }
      const long c = (long)y * W + x;
      v_out[plane + c] = v[plane + c];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        q_out[((long)b * 8 + a) * H * W + c] = q[a];
        pi_out[((long)b * 8 + a) * H * W + c] = __fdiv_rn(e[a], s);
      }
    }
  }
}

}  // namespace creste

using namespace creste;

static inline size_t vi_align(size_t x) { return (x + 255) / 256 * 256; }

extern "C" int64_t creste_value_iteration_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return -1;
  // two v buffers + state header + one delta word per sweep (bounded by 1<<20 sweeps, + one chunk)
  return (int64_t)(2 * vi_align((size_t)B * H * W * 4) + vi_align(sizeof(VmState) + 4u * ((1u << 20) + 256)));
}

extern "C" int creste_value_iteration_f32(const float* r, int B, int H, int W, float discount,
                                          float threshold, int max_sweeps, float* v, float* q,
                                          float* policy, int32_t* sweeps_out, void* work, void* stream) {
  CRESTE_REQUIRE(r && v && q && policy && sweeps_out && work, "value_iteration: null pointer");
  CRESTE_REQUIRE(B > 0 && H > 0 && W > 0, "value_iteration: bad dims");
  CRESTE_REQUIRE(max_sweeps > 0 && max_sweeps <= (1 << 20), "value_iteration: max_sweeps out of range");
  hipStream_t s = (hipStream_t)stream;
  char* wp = (char*)work;
  float* buf0 = (float*)wp;
  float* buf1 = (float*)(wp + vi_align((size_t)B * H * W * 4));
  VmState* st = (VmState*)(wp + 2 * vi_align((size_t)B * H * W * 4));
  unsigned* delta = (unsigned*)(st + 1);

  const dim3 fgrid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
  // ---- persistent solver (one launch, no host synchronisation): tiles of 32 x 32, or 16 x 16 while that still leaves
  // CUs idle; every workgroup must be resident at once (device-scope barrier)
  {
    constexpr int PS = 8;                                           // sweeps per chunk = halo width
    int dev = 0, cus = 0;
    CRESTE_HIP(hipGetDevice(&dev));
    CRESTE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int T = 32;
    if ((long)B * ((H + 31) / 32) * ((W + 31) / 32) < cus) T = 16;
    const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
    const long nwg = (long)B * tiles_x * tiles_y;
    const int RH = T + 2 * PS, RW = T + 2 * PS, nstr = RH * (RW / 4);
    // one 1 x 4 strip per thread (576 threads for the 48 x 48 region, 256 for 32 x 32): many light waves per SIMD -- with 3
    // strips per thread and 1.5 waves per SIMD every LDS / barrier / dependent-issue latency was exposed (4.2 us per sweep)
    const int threads = nstr, nstrip = 1;
    const size_t psmem = (size_t)(2 * (RH + 2) * (RW + 8) + PS * VI_MAXW) * sizeof(float);
    int per_cu = 0;
    const void* fn = reinterpret_cast<const void*>(vi_persist_kernel<1>);
    CRESTE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, psmem));
    {
      // The occupancy query assumes the best placement of a workgroup's waves; the hardware deals them to the four SIMDs
      // round-robin from a varying start, so ceil(waves / 4) of them can land on one SIMD for every workgroup of the CU.
      // Size the launch with that worst case (measured: a 93-register build of this kernel, 5 waves per SIMD, query
      // answer 2 per CU, left the second 9-wave workgroup of some CUs waiting for ever).
      hipFuncAttributes fa;
      CRESTE_HIP(hipFuncGetAttributes(&fa, fn));
      const int alloc = (fa.numRegs + 7) / 8 * 8, wps = alloc > 0 ? (512 / alloc > 8 ? 8 : 512 / alloc) : 8;
      const int wg_waves = (threads + 63) / 64, worst = wps / ((wg_waves + 3) / 4);
      if (per_cu > worst) per_cu = worst;
    }
    const char* force_multi = getenv("CRESTE_VI_MULTI");
    if (nwg <= (long)per_cu * cus && !(force_multi && force_multi[0] == '1')) {
      const int max_chunks = (max_sweeps + PS - 1) / PS;
      // Two solves at once (two streams of this process) could each end up partially resident and wait for each other
      // until the bounded spins give up: launches of the persistent solver are chained through one event per device, so
      // the second one starts behind the first whatever stream it is on.  (Not inside a stream capture: a captured launch
      // cannot wait on an event of another stream; graph replays rely on the bounded spin alone.)
      static std::mutex mu;
      static hipEvent_t chain[64] = {};
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      CRESTE_HIP(hipStreamIsCapturing(s, &cap));
      const bool chained = cap == hipStreamCaptureStatusNone && dev < 64;
      std::unique_lock<std::mutex> lock(mu, std::defer_lock);
      if (chained) {
        lock.lock();
        if (!chain[dev]) CRESTE_HIP(hipEventCreateWithFlags(&chain[dev], hipEventDisableTiming));
        else CRESTE_HIP(hipStreamWaitEvent(s, chain[dev], 0));
      }
      CRESTE_HIP(hipMemsetAsync(buf0, 0, (size_t)B * H * W * 4, s));
      CRESTE_HIP(hipMemsetAsync(st, 0, sizeof(VmState) + 4u * (size_t)((nwg + 63) / 64 * 64 + 64 + 2 * nwg * PS), s));
      // rendezvous area (inside the memset above): arrival flags, the verdict word on its own line, per-workgroup deltas
      unsigned* arrive = delta;
      unsigned* go = delta + (nwg + 63) / 64 * 64;
      float* dwg = (float*)(go + 64);
      CRESTE_REQUIRE((nwg + 63) / 64 * 64 + 64 + 2 * nwg * PS <= (1L << 20), "value_iteration: workspace too small for the rendezvous area");
      ViPArgs a{r, buf0, buf1, st, arrive, go, dwg, B, H, W, T, T, PS, tiles_x, tiles_y, (int)nwg, max_chunks, discount, threshold};
      vi_persist_kernel<1><<<(unsigned)nwg, threads, psmem, s>>>(a);
      (void)nstrip;
      CRESTE_CHECK_LAUNCH("vi_persist");
      if (chained) CRESTE_HIP(hipEventRecord(chain[dev], s));
      // without convergence the last chunk wrote plane max_chunks & 1; *sweeps_out < 0 reports it (the call itself stays
      // asynchronous: a non-converged solve is a negative sweep count, INT32_MIN a launch that was not co-resident)
      vi_final2_kernel<<<fgrid, 256, 0, s>>>(r, buf0, buf1, st, max_chunks & 1, max_chunks * PS, H, W, discount, v, q, policy,
                                             sweeps_out);
      CRESTE_CHECK_LAUNCH("vi_final");
      return CRESTE_OK;
    }
  }
  // ---- fallback (grids too large to be co-resident): one launch per chunk, host peeks at `done` every few launches.
  // tiling: the whole grid in one tile when it fits 8192 cells (no halo, long chunks), otherwise
  // 64x64 tiles with an 8-cell halo (80x80 region) and 8 sweeps per launch.
  int TH_, TW_, halo, S;
  if ((long)H * W <= VM_THREADS * VM_CPT) { TH_ = H; TW_ = W; halo = 0; S = 64; }
  else { TH_ = 64; TW_ = 64; halo = 8; S = 8; }
  const int nchunks = (max_sweeps + S - 1) / S;
  CRESTE_HIP(hipMemsetAsync(buf0, 0, (size_t)B * H * W * 4, s));
  CRESTE_HIP(hipMemsetAsync(st, 0, sizeof(VmState) + 4u * (size_t)(nchunks + 1) * S, s));
  const dim3 grid((W + TW_ - 1) / TW_, (H + TH_ - 1) / TH_, B);
  const int RHmax = (TH_ + 2 * halo < H ? TH_ + 2 * halo : H), RWmax = (TW_ + 2 * halo < W ? TW_ + 2 * halo : W);
  const size_t smem = ((size_t)(RHmax + 2) * (RWmax + 2) + (size_t)S * 16) * sizeof(float);
  CRESTE_REQUIRE((long)RHmax * RWmax <= VM_THREADS * VM_CPT, "value_iteration: tile region too large");
  const int peek = S >= 32 ? 4 : 16;               // launches between host peeks at `done`
  int l = 0, done = 0;
  while (!done && l <= nchunks) {                  // launch `nchunks` is the check/redo of the last chunk
    const int end = (l + peek <= nchunks + 1) ? l + peek : nchunks + 1;
    for (; l < end; ++l)
      vi_multi_kernel<<<grid, VM_THREADS, smem, s>>>(r, buf0, buf1, st, delta, l, S, H, W, TH_, TW_, halo,
                                                     discount, threshold);
    CRESTE_CHECK_LAUNCH("vi_multi");
    CRESTE_HIP(hipMemcpyAsync(&done, &st->done, 4, hipMemcpyDeviceToHost, s));
    CRESTE_HIP(hipStreamSynchronize(s));
  }
  // without convergence the last normal launch (index nchunks-1) wrote buf[nchunks & 1]
  vi_final2_kernel<<<fgrid, 256, 0, s>>>(r, buf0, buf1, st, nchunks & 1, nchunks * S, H, W, discount, v, q,
                                         policy, sweeps_out);
  CRESTE_CHECK_LAUNCH("vi_final");
  if (!done) {
    set_error("value_iteration: no convergence in %d sweeps", nchunks * S);
    return CRESTE_ERR_NOCONV;
  }
  return CRESTE_OK;
}

#ifdef VI_TRACE
extern "C" int creste_vi_trace_read(long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(creste::g_vi_trace), 64) == hipSuccess ? 0 : -1;
}
#endif
