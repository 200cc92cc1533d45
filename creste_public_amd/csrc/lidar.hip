// LiDAR scan -> sparse depth image (the "128x1024 LiDAR" half of the input contract).
// reference: creste/utils/projection.py:64-155 `pixels_to_depth` (float64 numpy + torch_scatter):
//   p_cam = lidar2camrect[:3,:] @ [x,y,z,1] ; (u,v) = trunc(clip(p_cam[:2]/p_cam[2], int32 range)) ;
//   keep z_cam > 0 and 0 <= u < W, 0 <= v < H ; depth[v,u] = reduce_{points}(z_cam), reduce = max
//   (farthest return wins, projection.py:122-128), 0 where no point lands.
// HBM-bound integer/scatter work: 131,072 points x 12 B in, <= one 4-byte atomic each on an image that
// stays in L2 (2.9 MB/frame).  z_cam > 0, so the IEEE bit pattern of the float32 depth is monotone and
// max/min are plain unsigned atomics (no CAS loop); the image is zero-filled first (= "no return").
#include "common.h"

namespace creste {

__global__ __launch_bounds__(256) void lidar_depth_kernel(const float* __restrict__ pts, int ps,
                                                          const double* __restrict__ M, int mat_stride,
                                                          long NP, int H, int W, int reduce_min,
                                                          double scale, unsigned* __restrict__ depth,
                                                          long depth_bstride) {
  const int b = blockIdx.y;
  const double* m = M + (long)b * mat_stride;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < NP; i += (long)gridDim.x * blockDim.x) {
    const float* p = pts + ((long)b * NP + i) * ps;
    const double x = p[0], y = p[1], z = p[2];
    double c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      c[k] = fma(m[k * 4 + 3], 1.0, fma(m[k * 4 + 2], z, fma(m[k * 4 + 1], y, m[k * 4 + 0] * x)));
    if (!(c[2] > 0.0)) continue;
    double u = c[0] / c[2], v = c[1] / c[2];
    u = fmin(fmax(u, -2147483648.0), 2147483647.0);        // np.clip to the int32 range, then astype(int32)
    v = fmin(fmax(v, -2147483648.0), 2147483647.0);
    if (u != u || v != v) continue;
    const int ui = (int)u, vi = (int)v;                     // truncation toward zero
    if (ui < 0 || ui >= W || vi < 0 || vi >= H) continue;
    const float d = (float)(c[2] * scale);
    unsigned* cell = depth + (long)b * depth_bstride + (long)vi * W + ui;
    if (reduce_min) {
      // 0 means "empty": min over positive values only
      unsigned bits = __float_as_uint(d), old = *cell;
      while (old == 0u || bits < old) {
        const unsigned seen = atomicCAS(cell, old, bits);
        if (seen == old) break;
        old = seen;
      }
    } else {
      atomicMax(cell, __float_as_uint(d));
    }
  }
}

// The whole of `pixels_to_depth` for ONE scan, in the reference's float64 (every `return_keys` entry of
// projection.py:139-153).  Per point: uv = trunc(clip(xy / z)) (nan -> INT32_MIN as numpy casts it), mask = z > 0 and
// in-image.  Per pixel, two 64-bit atomicMax images (bit patterns of positive doubles order like the doubles):
//   red[pix]  = bits(z)  (max)  or ~bits(z) (min: the smallest z has the largest complement; 0 = empty either way)
//   last[pix] = (point index + 1) << 32 | bits((float)z): the highest index wins = numpy's fancy assignment with
//               repeated indices (projection.py:116-118), the low word is its depth
template <typename PT>
__global__ __launch_bounds__(256) void lidar_p2d_kernel(const PT* __restrict__ pts, int ps, const double* __restrict__ m,
                                                        long NP, int H, int W, int reduce_min, int* __restrict__ uv,
                                                        unsigned char* __restrict__ mask,
                                                        unsigned long long* __restrict__ red,
                                                        unsigned long long* __restrict__ last) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < NP; i += (long)gridDim.x * blockDim.x) {
    const PT* p = pts + i * ps;
    const double x = (double)p[0], y = (double)p[1], z = (double)p[2];
    double c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      c[k] = fma(m[k * 4 + 3], 1.0, fma(m[k * 4 + 2], z, fma(m[k * 4 + 1], y, m[k * 4 + 0] * x)));
    double u = c[0] / c[2], v = c[1] / c[2];
    u = fmin(fmax(u, -2147483648.0), 2147483647.0);
    v = fmin(fmax(v, -2147483648.0), 2147483647.0);
    // fmin / fmax drop a nan operand where np.clip keeps it: restore, then cast as numpy does (nan -> INT32_MIN)
    const double ur = c[0] / c[2], vr = c[1] / c[2];
    const int ui = (ur != ur) ? (int)0x80000000 : (int)u, vi = (vr != vr) ? (int)0x80000000 : (int)v;
    const bool keep = c[2] > 0.0 && ui >= 0 && ui < W && vi >= 0 && vi < H;
    if (uv) { uv[2 * i] = ui; uv[2 * i + 1] = vi; }
    if (mask) mask[i] = keep ? 1 : 0;
    if (!keep) continue;
    const long pix = (long)vi * W + ui;
    const unsigned long long zb = (unsigned long long)__double_as_longlong(c[2]);
    atomicMax(red + pix, reduce_min ? ~zb : zb);
    if (last) atomicMax(last + pix, ((unsigned long long)(i + 1) << 32) | __float_as_uint((float)c[2]));
  }
}

__global__ __launch_bounds__(256) void lidar_p2d_decode_kernel(const unsigned long long* __restrict__ red,
                                                               const unsigned long long* __restrict__ last, long HW,
                                                               int reduce_min, double* __restrict__ reduced,
                                                               float* __restrict__ lastw) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= HW) return;
  const unsigned long long r = red[i];
  reduced[i] = r == 0ull ? 0.0 : __longlong_as_double((long long)(reduce_min ? ~r : r));
  if (lastw) lastw[i] = __uint_as_float((unsigned)(last[i] & 0xffffffffull));
}

}  // namespace creste

using namespace creste;

extern "C" int creste_lidar_depth_image_f32(const float* points, int point_stride, const double* lidar2cam,
                                            int mat_stride, int B, int64_t NP, int H, int W, int reduce_min,
                                            double scale, float* depth, int64_t depth_batch_stride,
                                            void* stream) {
  CRESTE_REQUIRE(points && lidar2cam && depth, "lidar_depth_image: null pointer");
  CRESTE_REQUIRE(B > 0 && NP > 0 && H > 0 && W > 0 && point_stride >= 3 && mat_stride >= 12 &&
                     depth_batch_stride >= (int64_t)H * W && scale > 0.0,
                 "lidar_depth_image: bad dims");
  hipStream_t s = (hipStream_t)stream;
  for (int b = 0; b < B; ++b)
    CRESTE_HIP(hipMemsetAsync(depth + (size_t)b * depth_batch_stride, 0, (size_t)H * W * 4, s));
  const int gx = (int)((NP + 255) / 256 > 1024 ? 1024 : (NP + 255) / 256);
  lidar_depth_kernel<<<dim3(gx, B), 256, 0, s>>>(points, point_stride, lidar2cam, mat_stride, NP, H, W,
                                                 reduce_min, scale, reinterpret_cast<unsigned*>(depth),
                                                 depth_batch_stride);
  CRESTE_CHECK_LAUNCH("lidar_depth");
  return CRESTE_OK;
}

extern "C" int creste_lidar_pixels_to_depth_f64(const void* points, int points_f64, int point_stride,
                                                const double* lidar2cam, int64_t NP, int H, int W, int reduce_min,
                                                int* uv, unsigned char* mask, double* reduced, float* last_write,
                                                void* work, void* stream) {
  CRESTE_REQUIRE(points && lidar2cam && reduced && work, "lidar_pixels_to_depth: null pointer");
  CRESTE_REQUIRE(NP > 0 && NP < (1ll << 31) && H > 0 && W > 0 && point_stride >= 3, "lidar_pixels_to_depth: bad dims");
  CRESTE_REQUIRE((reinterpret_cast<uintptr_t>(work) & 7) == 0, "lidar_pixels_to_depth: work must be 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const long HW = (long)H * W;
  unsigned long long* red = (unsigned long long*)work;
  unsigned long long* last = last_write ? red + HW : nullptr;
  CRESTE_HIP(hipMemsetAsync(work, 0, (size_t)HW * 8 * (last_write ? 2 : 1), s));
  const int gx = (int)((NP + 255) / 256 > 2048 ? 2048 : (NP + 255) / 256);
  if (points_f64)
    lidar_p2d_kernel<double><<<gx, 256, 0, s>>>((const double*)points, point_stride, lidar2cam, NP, H, W, reduce_min, uv, mask, red, last);
  else
    lidar_p2d_kernel<float><<<gx, 256, 0, s>>>((const float*)points, point_stride, lidar2cam, NP, H, W, reduce_min, uv, mask, red, last);
  CRESTE_CHECK_LAUNCH("lidar_p2d");
  lidar_p2d_decode_kernel<<<(unsigned)((HW + 255) / 256), 256, 0, s>>>(red, last, HW, reduce_min, reduced, last_write);
  CRESTE_CHECK_LAUNCH("lidar_p2d_decode");
  return CRESTE_OK;
}
