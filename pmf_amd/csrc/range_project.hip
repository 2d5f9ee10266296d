// Spherical (range-image) projection of a LiDAR sweep and the SalsaNext input tensor (gfx950).
// Follows pc_processor/dataset/preprocess/projection.py:31-86 (RangeProjection.doProjection),
// pc_processor/dataset/salsanext_loader.py:48-84 (feature / label / mask assembly) and
// pc_processor/dataset/preprocess/augmentor.py:97-180 (rigid augmentation of the sweep).
//
//   depth = sqrt((x*x + y*y) + z*z)                         float32, left to right (numpy add.reduce over 3 terms)
//   yaw = -arctan2(y, x) ; pitch = arcsin(z / depth)       float32
//   col = clamp(floor((yaw + |fov_left|) / fov_h * W)), row = clamp(floor((1 - (pitch + |fov_down|) / fov_v) * H))
//   the reference sorts by decreasing depth and lets the LAST writer win -> the pixel keeps its NEAREST point.
//
// The sort + scatter becomes one 64-bit atomicMin per point on the key (depth bits << 32 | point index): depth >= 0, so
// the float bit pattern orders like the value; equal depths resolve to the smaller index (what a stable sort gives --
// numpy's default argsort is not stable, so the reference leaves that case open).  arctan2 / arcsin are evaluated in
// float64 and rounded once: numpy's float32 routines are platform dependent (libm / SIMD variants, <= 1 ulp apart), so
// a column or row index can differ from a given numpy build only for a point whose projected coordinate sits within
// float32 rounding of a pixel edge (the parity tests check exactly that).  A point at the origin (depth 0) is a NaN
// index in the reference (it raises); here it is clamped like any other value.
#include "common.h"

#pragma clang fp contract(off)

#define RB 256

__global__ __launch_bounds__(RB) void range_index_k(const float* __restrict__ pts, int64_t P, int C, float fov_left_abs,
                                                    float fov_h, float fov_down_abs, float fov_v, int H, int W,
                                                    unsigned long long* __restrict__ keys, int32_t* __restrict__ ux,
                                                    int32_t* __restrict__ uy, float* __restrict__ ud) {
  const int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x;
  if (i >= P) return;
  const float x = pts[i * C], y = pts[i * C + 1], z = pts[i * C + 2];
  const float depth = sqrtf((x * x + y * y) + z * z);
  const float yaw = -(float)atan2((double)y, (double)x);
  const float pitch = (float)asin((double)(z / depth));
  float px = (yaw + fov_left_abs) / fov_h;
  float py = 1.0f - (pitch + fov_down_abs) / fov_v;
  px *= (float)W;
  py *= (float)H;
  const int cx = (int)fmaxf(fminf((float)(W - 1), floorf(px)), 0.f);
  const int cy = (int)fmaxf(fminf((float)(H - 1), floorf(py)), 0.f);
  if (ux) { ux[i] = cx; uy[i] = cy; ud[i] = depth; }
  const unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned long long)(uint32_t)i;
  atomicMin(keys + (size_t)cy * W + cx, key);
}

__global__ __launch_bounds__(RB) void range_gather_k(const float* __restrict__ pts, int C,
                                                     const unsigned long long* __restrict__ keys, int H, int W,
                                                     const int32_t* __restrict__ mapped_label,
                                                     const float* __restrict__ mean, const float* __restrict__ stds,
                                                     float* __restrict__ feature, float* __restrict__ label,
                                                     int32_t* __restrict__ mask, float* __restrict__ range,
                                                     int32_t* __restrict__ idx_out, float* __restrict__ proj_points) {
  const int p = blockIdx.x * RB + threadIdx.x;
  const int HW = H * W;
  if (p >= HW) return;
  const unsigned long long key = keys[p];
  const bool hit = key != ~0ull;
  const int idx = hit ? (int)(uint32_t)key : -1;
  const float r = hit ? __uint_as_float((uint32_t)(key >> 32)) : -1.f;
  float v[4] = {-1.f, -1.f, -1.f, -1.f};
  if (hit) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = k < C ? pts[(size_t)idx * C + k] : -1.f;
  }
  const int m = idx > 0 ? 1 : 0;            // the reference's mask drops point 0 as well as empty pixels
  if (range) range[p] = r;
  if (idx_out) idx_out[p] = idx;
  if (mask) mask[p] = m;
  if (proj_points) {
    for (int k = 0; k < C; ++k) proj_points[(size_t)p * C + k] = hit ? pts[(size_t)idx * C + k] : -1.f;
  }
  if (label) {
    const float l = (m && mapped_label) ? (float)mapped_label[idx] : 0.f;
    label[p] = l * (float)m;
  }
  if (feature) {
    const float mf = (float)m;
    const float inten = (v[3] != -1.f ? 1.f : 0.f) * v[3];
    const float raw[5] = {r, v[0], v[1], v[2], inten};
#pragma unroll
    for (int k = 0; k < 5; ++k) feature[(size_t)k * HW + p] = ((raw[k] - mean[k]) / stds[k]) * mf;
  }
}

// augmentor.py:97-121: flips, float32 translation, rotation p <- float32(float64(p) . R^T)
__global__ __launch_bounds__(RB) void points_transform_k(float* __restrict__ pts, int64_t P, int C, int flipx, int flipy,
                                                         float tx, float ty, float tz, const double* __restrict__ rot) {
  const int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x;
  if (i >= P) return;
  float x = pts[i * C], y = pts[i * C + 1], z = pts[i * C + 2];
  if (flipx) x = -x;
  if (flipy) y = -y;
  x += tx; y += ty; z += tz;
  if (rot) {
    const double a = (double)x, b = (double)y, c = (double)z;
    // k-ordered FMA chain (the order an FMA dgemm micro-kernel accumulates a length-3 dot product)
    x = (float)fma(c, rot[2], fma(b, rot[1], a * rot[0]));
    y = (float)fma(c, rot[5], fma(b, rot[4], a * rot[3]));
    z = (float)fma(c, rot[8], fma(b, rot[7], a * rot[6]));
  }
  pts[i * C] = x; pts[i * C + 1] = y; pts[i * C + 2] = z;
}

extern "C" int pmf_range_project_index(const float* points, int64_t P, int32_t C, float fov_left_abs, float fov_h,
                                       float fov_down_abs, float fov_v, int32_t H, int32_t W, uint64_t* keys,
                                       int32_t* uproj_x, int32_t* uproj_y, float* uproj_depth, pmf_stream_t s) {
  if ((P > 0 && !points) || !keys || P < 0 || C < 3 || H <= 0 || W <= 0 || P > 0x7fffffffll) return PMF_E_ARG;
  if (!(fov_h > 0.f) || !(fov_v > 0.f)) return PMF_E_ARG;
  if ((uproj_x != nullptr) != (uproj_y != nullptr) || (uproj_x != nullptr) != (uproj_depth != nullptr)) return PMF_E_ARG;
  hipError_t e = hipMemsetAsync(keys, 0xff, (size_t)H * W * sizeof(uint64_t), (hipStream_t)s);
  if (e != hipSuccess) return (int)e;
  if (P == 0) return 0;
  hipLaunchKernelGGL(range_index_k, dim3((unsigned)cdiv64(P, RB)), dim3(RB), 0, (hipStream_t)s, points, P, C,
                     fov_left_abs, fov_h, fov_down_abs, fov_v, H, W, (unsigned long long*)keys, uproj_x, uproj_y,
                     uproj_depth);
  PMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pmf_range_project_gather(const float* points, int64_t P, int32_t C, const uint64_t* keys, int32_t H,
                                        int32_t W, const int32_t* mapped_label, const float* mean5, const float* std5,
                                        float* feature, float* label, int32_t* mask, float* range, int32_t* idx,
                                        float* proj_points, pmf_stream_t s) {
  if (!keys || P < 0 || C < 3 || H <= 0 || W <= 0 || (P > 0 && !points)) return PMF_E_ARG;
  if (feature && (!mean5 || !std5 || C < 4)) return PMF_E_ARG;
  hipLaunchKernelGGL(range_gather_k, dim3(cdiv(H * W, RB)), dim3(RB), 0, (hipStream_t)s, points, C,
                     (const unsigned long long*)keys, H, W, mapped_label, mean5, std5, feature, label, mask, range, idx,
                     proj_points);
  PMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pmf_points_transform(float* points, int64_t P, int32_t C, int32_t flipx, int32_t flipy, float tx, float ty,
                                    float tz, const double* rot9, pmf_stream_t s) {
  if (P < 0 || C < 3 || (P > 0 && !points)) return PMF_E_ARG;
  if (P == 0) return 0;
  hipLaunchKernelGGL(points_transform_k, dim3((unsigned)cdiv64(P, RB)), dim3(RB), 0, (hipStream_t)s, points, P, C, flipx,
                     flipy, tx, ty, tz, rot9);
  PMF_LAUNCH_CHECK();
  return 0;
}
