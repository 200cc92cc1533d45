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
#include <string.h>

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

__global__ __launch_bounds__(256) void vi_final2_kernel(const float* __restrict__ r,
                                                        const float* __restrict__ buf0,
                                                        const float* __restrict__ buf1, const VmState* st,
                                                        int last_buf, int sweeps_run, int H, int W, float gamma,
                                                        float* __restrict__ v_out, float* __restrict__ q_out,
                                                        float* __restrict__ pi_out, int32_t* sweeps_out) {
  __shared__ float xs[TH + 2][TW + 2];
  const bool done = st->done != 0;
  const float* v = (done ? st->final_buf : last_buf) ? buf1 : buf0;
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const long plane = (long)b * H * W;
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    *sweeps_out = done ? st->converged_at : -sweeps_run;
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
  const dim3 fgrid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
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
