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

// ---------------------------------------------------------------------------------------------
// The persistent solver WITHOUT a barrier (the default).  vi_persist_kernel above spends 43-55 % of a solve in its
// rendezvous (profiles/r04_value_iteration.md: four dependent trips to the coherence point per chunk, ~10 us at 512
// workgroups even with nothing to wait for).  Here nothing waits for a global decision:
//   * after its 8 sweeps of chunk c a workgroup publishes its tile's interior as 8-byte {value, c + 1} granules
//     (write-through 16-byte stores, two granules each) and its 8 per-sweep interior deltas as {delta, c + 1} granules;
//   * it then polls ONLY the granules of its halo ring (its <= 8 neighbours' interiors) until they carry this chunk's
//     tag -- the tag travels with the data, so there is no flag to order behind the payload and no fence -- and goes on
//     with chunk c + 1 SPECULATIVELY;
//   * the batch-global convergence test of chunk c - 1 is evaluated one chunk late, by every workgroup for itself: the
//     loads of all workgroups' delta granules of chunk c - 1 (published at least a chunk ago) are issued before the halo
//     polling and consumed after it, i.e. their latency hides behind the neighbour exchange.  If the test failed first at
//     sweep j of chunk c - 1, the workgroup drops what it did since, restores the state it had at the start of chunk
//     c - 1 (a two-deep register snapshot), redoes exactly j + 1 sweeps and writes the final values.
// Same arithmetic, same sweep count as sweep-per-launch execution (tests: ..._persistent_equals_chunk_per_launch);
// at most two chunks (16 of ~690 sweeps) of discarded work at the end.  The exchange planes are double-buffered by chunk
// parity: a workgroup can only be one chunk ahead of a NEIGHBOUR (it needs the neighbour's previous chunk to proceed).
// The delta records are read by everybody, one chunk late: a workgroup enters chunk c once all have finished the sweeps of
// c - 2, while a slow one may still have to read the records of c - 3 -- four slots (chunk & 3).  (With two, a fast
// workgroup overwrote a record a slow one was still polling for: found as an INT32_MIN abort at 512 workgroups.)
// Every poll is bounded (abort word, INT32_MIN sweeps) as above.
struct ViSArgs {
  const float* r;
  unsigned long long* vex;       // [2][B*H*W] {v bits, chunk + 1}: tile interiors after that chunk
  float* vfinal;                 // [B*H*W] the answer (plain stores; read by vi_final2_kernel after the kernel boundary)
  VmState* st;
  unsigned* abort_word;
  unsigned* rec;                 // [4][nwg] (chunk + 1) << 8 | bit j: the tile's interior moved by more than thr in sweep j
  int B, H, W, TH, TW, S, tiles_x, tiles_y, nwg, max_chunks, vec;
  float gamma, thr;
};

typedef unsigned long long vi_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void vi_store16_sc1(void* p, vi_u64x2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// two 16-byte loads in flight together (32 bytes = the four granules of a strip), one wait
__device__ __forceinline__ void vi_load32_sc1(const void* p, vi_u64x2& a, vi_u64x2& b) {
  asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}

// Thread mapping: a wave holds WHOLE rows of 1 x 4 strips (64 / SW rows; a few tail lanes idle), so a strip's left / right
// neighbour cells are its neighbour LANES' values: they come by DPP (v_mov_b32_dpp wave_shr / wave_shl), the strip's own
// row stays in registers from the sweep before, and a sweep touches the LDS with two 16-byte reads (rows above / below)
// and one 16-byte write -- round 3's form read three rows as nine ds_read2_b32 (the sweeps wait on LDS latency; the VALU
// work is free: profiles/r04_value_iteration.md).
// MAXT: 640 threads (32 x 32 tiles: 48 x 48 region = 10 waves of 5 rows; 16 x 16: 32 x 32 = 4 waves of 8 rows) or 1024
// (32 x 64 tiles, 48 x 80 = 16 waves of 3 rows).  ONE workgroup per CU in every case (the launch is refused otherwise: two
// workgroups per CU only get in each other's way, see the tile choice at the launch).
__device__ __forceinline__ float vi_lane_below(float v) {      // value of lane - 1 (0 for lane 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float vi_lane_above(float v) {      // value of lane + 1 (0 for lane 63)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
template <int MAXT>
__global__ __launch_bounds__(MAXT, MAXT == 640 ? 3 : 4) void vi_spec_kernel(const ViSArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_flag[2];
  constexpr int SS = 8;
  const int S = p.S, RH = p.TH + 2 * S, RW = p.TW + 2 * S, LW = RW + 8, SW = RW >> 2, RPW = 64 / SW;
  const int plane = (RH + 2) * LW;
  float* const xb[2] = {lds, lds + plane};
  unsigned* const smask = reinterpret_cast<unsigned*>(lds + 2 * plane);      // [2][VI_MAXW] per-wave ORs: own chunk / everybody's
  const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63;
  int id = blockIdx.x;
  const int tx = id % p.tiles_x; id /= p.tiles_x;
  const int ty = id % p.tiles_y;
  const int b = id / p.tiles_y;
  const int gy0 = ty * p.TH - S, gx0 = tx * p.TW - S;
  const long pl = (long)b * p.H * p.W, BHW = (long)p.B * p.H * p.W;

  float rr[4], vv[4], s0[4], s1[4];      // s1: state at the start of the current chunk, s0: of the one before
  float own[4] = {0.f, 0.f, 0.f, 0.f};   // x = r + gamma v of this strip (what the LDS plane holds for it)
  int li = -1, gofs = 0;
  unsigned inb = 0, inner = 0;           // 4-bit masks: cell inside the grid / inside this workgroup's tile
#pragma unroll
  for (int e = 0; e < 4; ++e) { rr[e] = 0.f; vv[e] = 0.f; s0[e] = 0.f; s1[e] = 0.f; }
  const int lrow = lane / SW, cs = lane - lrow * SW, srow = wave * RPW + lrow;
  const bool first_cs = cs == 0, last_cs = cs == SW - 1;
  if (lrow < RPW && srow < RH) {
    const int row = srow, c0 = cs * 4;
    const int gy = gy0 + row, gx = gx0 + c0;
    li = (row + 1) * LW + 4 + c0;
    gofs = gy * p.W + gx;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = (unsigned)gy < (unsigned)p.H && (unsigned)(gx + e) < (unsigned)p.W;
      if (in) { inb |= 1u << e; rr[e] = p.r[pl + gofs + e]; }
      if (in && row >= S && row < S + p.TH && c0 + e >= S && c0 + e < S + p.TW) inner |= 1u << e;
    }
  }
  const unsigned ring = inb & ~inner;
  const bool vec = (p.vec & 1) && inb == 15u;          // a whole strip, 32-byte aligned granules: 16-byte loads
  // ... 16-byte stores (CRESTE_VI_VEC=3, NOT the default): a 16-byte `sc0 sc1` store is not atomic per 8-byte granule --
  // readers on other CUs saw a fresh tag next to a stale value (160+ wrong cells after two chunks once two workgroups
  // share a CU; 8 x 160 x 256, scripts/vi_dbg.py).  Granules are therefore published with one 8-byte atomic store each.
  const bool vecs = (p.vec & 2) && inb == 15u;
  for (int i = tid; i < 2 * plane; i += nt) lds[i] = 0.f;     // zero frames (and out-of-grid cells) of both planes
  __syncthreads();

  const int lis = li >= 0 ? li : LW + 4;               // idle lanes read (never write) a valid slot: every lane runs the DPP
  auto put_x = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) own[e] = (inb & (1u << e)) ? __fadd_rn(rr[e], __fmul_rn(vv[e], p.gamma)) : 0.f;
    if (li >= 0) *reinterpret_cast<float4*>(__builtin_assume_aligned(dst + li, 16)) = make_float4(own[0], own[1], own[2], own[3]);
  };
  // one Jacobi sweep of the whole region: plane `src` -> registers + plane `dst`; returns this thread's interior delta.
  // No lane leaves early: the neighbour lanes' values are read by DPP.
  auto sweep = [&](const float* src, float* dst) __attribute__((always_inline)) -> float {
    float dmax = 0.f;
    float w[3][6];
    const float4 up = *reinterpret_cast<const float4*>(__builtin_assume_aligned(src + lis - LW, 16));
    const float4 dn = *reinterpret_cast<const float4*>(__builtin_assume_aligned(src + lis + LW, 16));
    w[0][1] = up.x; w[0][2] = up.y; w[0][3] = up.z; w[0][4] = up.w;
    w[1][1] = own[0]; w[1][2] = own[1]; w[1][3] = own[2]; w[1][4] = own[3];
    w[2][1] = dn.x; w[2][2] = dn.y; w[2][3] = dn.z; w[2][4] = dn.w;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const float l = vi_lane_below(w[dy][4]), r = vi_lane_above(w[dy][1]);
      w[dy][0] = first_cs ? 0.f : l;          // beyond the region's edge: the zero frame
      w[dy][5] = last_cs ? 0.f : r;
    }
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
      if (inner & (1u << e)) dmax = fmaxf(dmax, fabsf(__fsub_rn(m, vv[e])));
      vv[e] = m;
      xn[e] = (inb & (1u << e)) ? __fadd_rn(rr[e], __fmul_rn(m, p.gamma)) : 0.f;
      own[e] = xn[e];
    }
    if (li >= 0) *reinterpret_cast<float4*>(__builtin_assume_aligned(dst + li, 16)) = make_float4(xn[0], xn[1], xn[2], xn[3]);
    return dmax;
  };
  auto gran = [](float v, unsigned tag) __attribute__((always_inline)) -> unsigned long long {
    return ((unsigned long long)tag << 32) | __float_as_uint(v);
  };
  auto finish = [&]() __attribute__((always_inline)) {      // the answer, plain stores (visible after the kernel boundary)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (inner & (1u << e)) p.vfinal[pl + gofs + e] = vv[e];
  };

#ifdef VI_TRACE
  long long t_sw = 0, t_pub = 0, t_halo = 0, t_ver = 0, tA, tB, tC, tD, tE;
  int npoll = 0;
#endif
  for (int chunk = 0; chunk <= p.max_chunks; ++chunk) {
    const unsigned tag = (unsigned)(chunk + 1);
    const bool work = chunk < p.max_chunks;
#ifdef VI_TRACE
    tA = __builtin_readcyclecounter();
#endif
    unsigned long long* const ex = p.vex + (size_t)(chunk & 1) * BHW + pl + gofs;
    if (work) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s0[e] = s1[e]; s1[e] = vv[e]; }
      put_x(xb[0]);
      __syncthreads();
      float dm[SS];
#pragma unroll
      for (int j = 0; j < SS; ++j) {
        dm[j] = sweep(xb[j & 1], xb[(j + 1) & 1]);
        __syncthreads();
      }
#ifdef VI_TRACE
      tB = __builtin_readcyclecounter();
#endif
      // publish the interior (the neighbours' halo for chunk + 1) first, then the deltas
      if (inner) {
        if (vecs && inner == 15u) {
          vi_store16_sc1(ex, vi_u64x2{gran(vv[0], tag), gran(vv[1], tag)});
          vi_store16_sc1(ex + 2, vi_u64x2{gran(vv[2], tag), gran(vv[3], tag)});
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (inner & (1u << e)) __hip_atomic_store(ex + e, gran(vv[e], tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // the batch-global test `max |v' - v| > thr` of sweep j is an OR over cells: one bit per sweep and workgroup
      unsigned mask = 0;
#pragma unroll
      for (int j = 0; j < SS; ++j) mask |= (dm[j] > p.thr ? 1u : 0u) << j;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mask |= (unsigned)__shfl_xor((int)mask, o);
      if ((tid & 63) == 0) smask[wave] = mask;
      __syncthreads();
      if (tid == 0) {
        unsigned m = 0;
        for (int w = 0; w < (nt + 63) / 64; ++w) m |= smask[w];
        __hip_atomic_store(p.rec + (size_t)(chunk & 3) * p.nwg + blockIdx.x, (tag << 8) | m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#ifdef VI_TRACE
    tC = __builtin_readcyclecounter();
#endif
    // ---- the verdict words of chunk - 1, of every workgroup: loads first (they are old: one trip, hidden behind the halo polls)
    int fail = 0;
    const unsigned* rbase = p.rec + (size_t)((chunk - 1) & 3) * p.nwg;
    unsigned g0 = 0;
    const bool have_rec = chunk >= 1 && tid < p.nwg;
    if (have_rec) g0 = __hip_atomic_load(rbase + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- halo ring <- the neighbours' interiors of this chunk
    if (work && ring) {
      for (int spin = 0;; ++spin) {
        bool ok = true;
        if (vec && ring == 15u) {
          vi_u64x2 a, c;
          vi_load32_sc1(ex, a, c);
          ok = (unsigned)(a[0] >> 32) == tag && (unsigned)(a[1] >> 32) == tag && (unsigned)(c[0] >> 32) == tag &&
               (unsigned)(c[1] >> 32) == tag;
          if (ok) {
            vv[0] = __uint_as_float((unsigned)a[0]); vv[1] = __uint_as_float((unsigned)a[1]);
            vv[2] = __uint_as_float((unsigned)c[0]); vv[3] = __uint_as_float((unsigned)c[1]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ring & (1u << e)) {
              const unsigned long long g = __hip_atomic_load(ex + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((unsigned)(g >> 32) == tag) vv[e] = __uint_as_float((unsigned)g); else ok = false;
            }
        }
        if (ok) break;
        if ((spin & 255) == 255 && __hip_atomic_load(p.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { fail = 1; break; }
        if (spin > VI_SPIN_LIMIT) {
          __hip_atomic_store(p.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fail = 1;
          break;
        }
      }
    }
#ifdef VI_TRACE
    __syncthreads();
    tD = __builtin_readcyclecounter();
#endif
    // ---- verdict of chunk - 1 (every workgroup computes the same one)
    int first = -1;
    if (chunk >= 1) {
      unsigned any = 0;
      for (int i = tid; i < p.nwg; i += nt) {
        unsigned g = g0;
        for (int spin = 0;; ++spin) {
          if (!(i == tid && spin == 0)) g = __hip_atomic_load(rbase + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((g >> 8) == (unsigned)chunk) break;
          if ((spin & 255) == 255 && __hip_atomic_load(p.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { fail = 1; break; }
          if (spin > VI_SPIN_LIMIT) {
            __hip_atomic_store(p.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fail = 1;
            break;
          }
        }
        any |= g & 255u;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) any |= (unsigned)__shfl_xor((int)any, o);
      if ((tid & 63) == 0) smask[VI_MAXW + wave] = any;
    }
    if (tid == 0) s_flag[0] = 0;
    __syncthreads();
    if (fail) s_flag[0] = 1;
    if (chunk >= 1 && tid == 0) {
      unsigned m = 0;
      for (int w = 0; w < (nt + 63) / 64; ++w) m |= smask[VI_MAXW + w];
      const unsigned quiet = ~m & 255u;                  // sweeps in which nobody moved by more than thr
      s_flag[1] = quiet ? __builtin_ctz(quiet) : -1;     // the first of them ends the iteration (vin.py:68-74)
    }
    __syncthreads();
#ifdef VI_TRACE
    tE = __builtin_readcyclecounter();
    if (work) { t_sw += tB - tA; t_pub += tC - tB; t_halo += tD - tC; t_ver += tE - tD; }
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && chunk >= 1) {
      long long* o = g_vi_trace + (blockIdx.x ? 4 : 0);
      o[0] = t_sw / chunk; o[1] = t_pub / chunk; o[2] = t_halo / chunk; o[3] = t_ver / chunk;
    }
#endif
    if (s_flag[0]) {                     // somebody gave up waiting: the launch is not co-resident
      if (blockIdx.x == 0 && tid == 0) p.st->done = -1;
      return;
    }
    if (chunk >= 1) first = s_flag[1];
    if (first >= 0) {
      // the test failed first at sweep `first` of chunk - 1: the answer is that chunk's start state advanced first + 1 sweeps
#pragma unroll
      for (int e = 0; e < 4; ++e) vv[e] = work ? s0[e] : s1[e];
      put_x(xb[0]);
      __syncthreads();
      for (int j = 0; j <= first; ++j) {
        (void)sweep(xb[j & 1], xb[(j + 1) & 1]);
        __syncthreads();
      }
      finish();
      if (blockIdx.x == 0 && tid == 0) {
        p.st->converged_at = (chunk - 1) * S + first + 1;
        p.st->final_buf = 0;
        __threadfence();
        p.st->done = chunk;
      }
      return;
    }
    __syncthreads();                     // s_flag / red are rewritten by the next chunk
  }
  finish();                              // no convergence within max_chunks chunks: the state after the last one
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
  // two exchange buffers of 8-byte {value, tag} granules (the launch-per-chunk fallback uses the same memory as two fp32
  // planes) + the answer + state header + one delta word per sweep (bounded by 1<<20 sweeps, + one chunk)
  return (int64_t)(2 * vi_align((size_t)B * H * W * 8) + vi_align((size_t)B * H * W * 4) +
                   vi_align(sizeof(VmState) + 4u * ((1u << 20) + 256)));
}

static int value_iteration_run(const float* r, int B, int H, int W, float discount, float threshold, int max_sweeps, float* v,
                               float* q, float* policy, int32_t* sweeps_out, void* work, void* stream, bool chunked);

extern "C" int creste_value_iteration_f32(const float* r, int B, int H, int W, float discount,
                                          float threshold, int max_sweeps, float* v, float* q,
                                          float* policy, int32_t* sweeps_out, void* work, void* stream) {
  return value_iteration_run(r, B, H, W, discount, threshold, max_sweeps, v, q, policy, sweeps_out, work, stream, false);
}

// the launch-per-chunk form on request (the retry of a persistent solve that reported INT32_MIN; creste_hip.h)
extern "C" int creste_value_iteration_chunked_f32(const float* r, int B, int H, int W, float discount,
                                                  float threshold, int max_sweeps, float* v, float* q,
                                                  float* policy, int32_t* sweeps_out, void* work, void* stream) {
  return value_iteration_run(r, B, H, W, discount, threshold, max_sweeps, v, q, policy, sweeps_out, work, stream, true);
}

static int value_iteration_run(const float* r, int B, int H, int W, float discount, float threshold, int max_sweeps, float* v,
                               float* q, float* policy, int32_t* sweeps_out, void* work, void* stream, bool chunked) {
  CRESTE_REQUIRE(r && v && q && policy && sweeps_out && work, "value_iteration: null pointer");
  CRESTE_REQUIRE(B > 0 && H > 0 && W > 0, "value_iteration: bad dims");
  CRESTE_REQUIRE(max_sweeps > 0 && max_sweeps <= (1 << 20), "value_iteration: max_sweeps out of range");
  hipStream_t s = (hipStream_t)stream;
  char* wp = (char*)work;
  float* buf0 = (float*)wp;
  float* buf1 = (float*)(wp + vi_align((size_t)B * H * W * 4));
  unsigned long long* vex = (unsigned long long*)wp;                          // [2][B*H*W] granules (barrier-free solver)
  float* vfinal = (float*)(wp + 2 * vi_align((size_t)B * H * W * 8));
  VmState* st = (VmState*)(wp + 2 * vi_align((size_t)B * H * W * 8) + vi_align((size_t)B * H * W * 4));
  unsigned* delta = (unsigned*)(st + 1);

  const dim3 fgrid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
  // ---- persistent solver (one launch, no host synchronisation): tiles of 32 x 32, or 16 x 16 while that still leaves
  // CUs idle; every workgroup must be resident at once (device-scope barrier)
  {
    constexpr int PS = 8;                                           // sweeps per chunk = halo width
    int dev = 0, cus = 0;
    CRESTE_HIP(hipGetDevice(&dev));
    CRESTE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const char* sync_env = getenv("CRESTE_VI_SYNC");
    const bool spec = !(sync_env && sync_env[0] == '1');           // the barrier-free solver (default) / the rendezvous form
    int T = 32, TWd = 32;
    if ((long)B * ((H + 31) / 32) * ((W + 31) / 32) < cus) T = TWd = 16;
    // more 32 x 32 tiles than CUs: two 9-wave workgroups per CU get in each other's way (a polling workgroup takes issue
    // slots from its computing CU-mate: 8 x 256 x 256 spent 25 k of 56 k cycles per chunk waiting).  32 x 64 tiles, one
    // 15-wave workgroup per CU, when that covers the grid: 17 % fewer halo cells as well
    else if (spec && (long)B * ((H + 31) / 32) * ((W + 31) / 32) > cus && (long)B * ((H + 31) / 32) * ((W + 63) / 64) <= cus) TWd = 64;
    const int tiles_x = (W + TWd - 1) / TWd, tiles_y = (H + T - 1) / T;
    const long nwg = (long)B * tiles_x * tiles_y;
    const int RH = T + 2 * PS, RW = TWd + 2 * PS;
    const int nstr = spec ? (RH + 64 / (RW / 4) - 1) / (64 / (RW / 4)) * 64        // whole rows of strips per wave
                          : RH * (RW / 4);
    // one 1 x 4 strip per thread (576 threads for the 48 x 48 region, 256 for 32 x 32): many light waves per SIMD -- with 3
    // strips per thread and 1.5 waves per SIMD every LDS / barrier / dependent-issue latency was exposed (4.2 us per sweep)
    const int threads = nstr, nstrip = 1;
    const size_t psmem = (size_t)(2 * (RH + 2) * (RW + 8) + 2 * PS * VI_MAXW) * sizeof(float);
    int per_cu = 0;
    const void* fn = !spec ? reinterpret_cast<const void*>(vi_persist_kernel<1>)
                           : (threads > 640 ? reinterpret_cast<const void*>(vi_spec_kernel<1024>) : reinterpret_cast<const void*>(vi_spec_kernel<640>));
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
      if (spec && per_cu > 1) per_cu = 1;          // the barrier-free solver is built and tuned for one workgroup per CU
    }
    const char* force_multi = getenv("CRESTE_VI_MULTI");
    if (nwg <= (long)per_cu * cus && !chunked && !(force_multi && force_multi[0] == '1')) {
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
      if (spec) {
        const long rec_words = 64 + 4 * nwg;                // abort word on its own line + [4][nwg] verdict words
        CRESTE_REQUIRE(rec_words <= (1L << 20) && (threads >= nwg || true), "value_iteration: workspace too small for the delta records");
        CRESTE_HIP(hipMemsetAsync(vex, 0, 2 * vi_align((size_t)B * H * W * 8), s));          // tags 0 = nothing published
        CRESTE_HIP(hipMemsetAsync(st, 0, sizeof(VmState) + 4u * (size_t)rec_words, s));
        const char* vec_env = getenv("CRESTE_VI_VEC");
        const int vec = (W % 4 == 0 && T % 4 == 0 && PS % 4 == 0) ? (vec_env ? atoi(vec_env) : 1) : 0;
        ViSArgs a{r, vex, vfinal, st, delta, delta + 64, B, H, W, T, TWd, PS, tiles_x,
                  tiles_y, (int)nwg, max_chunks, vec, discount, threshold};
        // second exchange plane starts B*H*W granules after the first (the kernel indexes [parity][B*H*W])
        if (threads > 640) vi_spec_kernel<1024><<<(unsigned)nwg, threads, psmem, s>>>(a);
        else vi_spec_kernel<640><<<(unsigned)nwg, threads, psmem, s>>>(a);
        CRESTE_CHECK_LAUNCH("vi_spec");
        if (chained) CRESTE_HIP(hipEventRecord(chain[dev], s));
        vi_final2_kernel<<<fgrid, 256, 0, s>>>(r, vfinal, vfinal, st, 0, max_chunks * PS, H, W, discount, v, q, policy, sweeps_out);
        CRESTE_CHECK_LAUNCH("vi_final");
        return CRESTE_OK;
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
