// Perspective projection + last-writer-wins scatter of a LiDAR sweep into the camera plane (gfx950).
// Follows pc_processor/dataset/semantic_kitti/parser.py:209-227 and perspective_view_loader.py:89-131:
//   keep = x > 0.5 ; (u,v,s) = P(3x4, float64) . [x y z 1] ; u/=s, v/=s ; keep &= 0<u<w & 0<v<h ;
//   row = int32(v), col = int32(u) (truncation) ; duplicates: the LAST point in file order wins.
// Order-preserving compaction (x_data / y_data) = block counts -> single-block scan -> ballot prefix.
// The winner per pixel is an atomicMax over point indices, then one gather pass builds the [10,h,w] tensor.
#include "common.h"

// bit-exact float32 / float64 arithmetic: no fused contraction except the explicit fma() calls below
#pragma clang fp contract(off)

#define PB 1024

struct Proj { double m[12]; };

__device__ __forceinline__ bool project_point(const float* __restrict__ pt, const double* __restrict__ m, int h, int w,
                                              int& row, int& col) {
  const float xf = pt[0];
  if (!(xf > 0.5f)) return false;
  const double x = (double)xf, y = (double)pt[1], z = (double)pt[2];
  // k-ordered FMA chain, the order an FMA dgemm micro-kernel accumulates a length-4 dot product
  const double a = fma(m[3], 1.0, fma(m[2], z, fma(m[1], y, m[0] * x)));
  const double b = fma(m[7], 1.0, fma(m[6], z, fma(m[5], y, m[4] * x)));
  const double c = fma(m[11], 1.0, fma(m[10], z, fma(m[9], y, m[8] * x)));
  const double u = a / c, v = b / c;
  if (!(u > 0.0 && u < (double)w && v > 0.0 && v < (double)h)) return false;
  row = (int)v;
  col = (int)u;
  return true;
}

__global__ __launch_bounds__(PB) void proj_count_k(const float* __restrict__ pts, int64_t P,
                                                   const double* __restrict__ m, int h, int w,
                                                   uint8_t* __restrict__ keep, float* __restrict__ depth,
                                                   int32_t* __restrict__ blk_cnt) {
  const int64_t i = blockIdx.x * (int64_t)PB + threadIdx.x;
  int k = 0;
  if (i < P) {
    int r, c;
    k = project_point(pts + i * 4, m, h, w, r, c) ? 1 : 0;
    keep[i] = (uint8_t)k;
    const float x = pts[i * 4], y = pts[i * 4 + 1], z = pts[i * 4 + 2];
    // numpy: sqrt(add.reduce(x*x)) in float32, left to right; sqrtf / '/' are correctly rounded (hipcc default)
    depth[i] = sqrtf((x * x + y * y) + z * z);
  }
  const int cnt = __syncthreads_count(k);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = cnt;
}

__global__ __launch_bounds__(1024) void proj_scan_k(int32_t* __restrict__ blk_cnt, int nblk, int32_t* __restrict__ n_kept) {
  // exclusive scan in place, nblk small (P / 1024)
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblk ? blk_cnt[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblk) blk_cnt[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_kept = carry;
}

__global__ __launch_bounds__(PB) void proj_scatter_k(const float* __restrict__ pts, int64_t P,
                                                     const double* __restrict__ m, int h, int w,
                                                     const int32_t* __restrict__ blk_off, int32_t* __restrict__ x_data,
                                                     int32_t* __restrict__ y_data, int32_t* __restrict__ pix_idx) {
  __shared__ int wave_cnt[PB / 64];
  const int64_t i = blockIdx.x * (int64_t)PB + threadIdx.x;
  int r = 0, c = 0;
  const bool k = i < P && project_point(pts + i * 4, m, h, w, r, c);
  const unsigned long long bal = __ballot(k);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_cnt[wv] = __popcll(bal);
  __syncthreads();
  int woff = 0;
  for (int j = 0; j < wv; ++j) woff += wave_cnt[j];
  if (k) {
    const int dst = blk_off[blockIdx.x] + woff + before;
    x_data[dst] = r;
    y_data[dst] = c;
    atomicMax(pix_idx + (size_t)r * w + c, (int)i);
  }
}

__global__ void proj_gather_k(const float* __restrict__ pts, const int32_t* __restrict__ sem,
                              const float* __restrict__ depth, const uint8_t* __restrict__ img,
                              const int32_t* __restrict__ lut, int nlut, const int32_t* __restrict__ pix_idx, int h, int w,
                              float* __restrict__ out) {
  const int64_t hw = (int64_t)h * w;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
    const int i = pix_idx[p];
    float d = 0.f, x = 0.f, y = 0.f, z = 0.f, it = 0.f, mk = 0.f, lb = 0.f;
    if (i >= 0) {
      const f32x4 q = *(const f32x4*)(pts + (size_t)i * 4);
      x = q.x; y = q.y; z = q.z; it = q.w;
      d = depth[i];
      mk = 1.f;
      const int sl = sem[i];
      lb = (float)((sl >= 0 && sl < nlut) ? lut[sl] : 0);
    }
    out[0 * hw + p] = d; out[1 * hw + p] = x; out[2 * hw + p] = y; out[3 * hw + p] = z; out[4 * hw + p] = it;
    out[5 * hw + p] = (float)img[p * 3 + 0] / 255.0f;
    out[6 * hw + p] = (float)img[p * 3 + 1] / 255.0f;
    out[7 * hw + p] = (float)img[p * 3 + 2] / 255.0f;
    out[8 * hw + p] = mk;
    out[9 * hw + p] = lb;
  }
}

extern "C" int pmf_project_scatter(const float* points, const int32_t* sem, int64_t P, const uint8_t* image,
                                   int32_t h, int32_t w, const double* proj, const int32_t* lut, int32_t nlut,
                                   float* proj_out, uint8_t* keep, int32_t* x_data, int32_t* y_data, float* depth,
                                   int32_t* n_kept, int32_t* pix_idx, int32_t* blk_cnt, pmf_stream_t s) {
  hipStream_t st = (hipStream_t)s;
  if (P < 0 || h < 1 || w < 1) return PMF_E_ARG;
  hipError_t e = hipMemsetAsync(pix_idx, 0xFF, (size_t)h * w * 4, st);
  if (e != hipSuccess) return (int)e;
  const int nblk = (int)cdiv64(P > 0 ? P : 1, PB);
  hipLaunchKernelGGL(proj_count_k, dim3(nblk), dim3(PB), 0, st, points, P, proj, h, w, keep, depth, blk_cnt);
  hipLaunchKernelGGL(proj_scan_k, dim3(1), dim3(1024), 0, st, blk_cnt, nblk, n_kept);
  hipLaunchKernelGGL(proj_scatter_k, dim3(nblk), dim3(PB), 0, st, points, P, proj, h, w, blk_cnt, x_data, y_data, pix_idx);
  int64_t hw = (int64_t)h * w;
  int g = (int)cdiv64(hw, 256);
  hipLaunchKernelGGL(proj_gather_k, dim3(g > 2048 ? 2048 : g), dim3(256), 0, st, points, sem, depth, image, lut, nlut,
                     pix_idx, h, w, proj_out);
  PMF_LAUNCH_CHECK();
  return 0;
}

__global__ void crop_pad_k(const float* __restrict__ src, int C, int h, int w, int top, int left, float* __restrict__ dst,
                           int oh, int ow, int pad_top, int pad_left, int ch, int cw) {
  const int64_t total = (int64_t)C * oh * ow;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % ow), y = (int)((i / ow) % oh), c = (int)(i / ((int64_t)ow * oh));
    const int cy = y - pad_top, cx = x - pad_left;  // position inside the crop window
    float v = 0.f;
    if (cy >= 0 && cy < ch && cx >= 0 && cx < cw) {
      const int sy = cy + top, sx = cx + left;
      if (sy >= 0 && sy < h && sx >= 0 && sx < w) v = src[((size_t)c * h + sy) * w + sx];
    }
    dst[i] = v;
  }
}
extern "C" int pmf_crop_pad(const float* src, int32_t C, int32_t h, int32_t w, int32_t top, int32_t left, float* dst,
                            int32_t oh, int32_t ow, int32_t pad_top, int32_t pad_left, int32_t ch, int32_t cw,
                            pmf_stream_t s) {
  int64_t total = (int64_t)C * oh * ow;
  int g = (int)cdiv64(total, 256);
  hipLaunchKernelGGL(crop_pad_k, dim3(g > 4096 ? 4096 : g), dim3(256), 0, (hipStream_t)s, src, C, h, w, top, left, dst, oh,
                     ow, pad_top, pad_left, ch, cw);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ---- EPMF loader (perspective_view_loader_v2.py:42-157 + parser.py:229-257 mapLidar2CameraCropYaw) ----------------
// keep = |xyz| > 0.5 && fov_left <= -atan2(y, x) <= fov_right (NO image-bounds filter); same float64 projection;
// (row, col) = trunc(v), trunc(u) may be negative -- the frame is the points' bounding box.
// Pass 1 (pmf_project_v2_index): keep mask, order-preserving compaction of (source index, row, col, (v,u) float64,
// depth) and the bounding box.  The caller reads n_kept / bbox (the output size is data dependent, as in the
// reference), then pass 2 (pmf_project_v2_scatter) resolves duplicates (last point wins) and writes [10][h][w] =
// depth, x, y, z, intensity, r, g, b (image window, zero outside), mask, label.
__device__ __forceinline__ bool v2_point(const float* __restrict__ pt, const double* __restrict__ m, float fl, float fr,
                                         double sc, int& row, int& col, double& v_out, double& u_out, float& dep) {
  const float xf = pt[0], yf = pt[1], zf = pt[2];
  dep = sqrtf((xf * xf + yf * yf) + zf * zf);
  const float yaw = -atan2f(yf, xf);
  if (!(dep > 0.5f) || !((double)yaw >= (double)fl && (double)yaw <= (double)fr)) return false;
  const double x = (double)xf, y = (double)yf, z = (double)zf;
  const double a = fma(m[3], 1.0, fma(m[2], z, fma(m[1], y, m[0] * x)));
  const double b = fma(m[7], 1.0, fma(m[6], z, fma(m[5], y, m[4] * x)));
  const double c = fma(m[11], 1.0, fma(m[10], z, fma(m[9], y, m[8] * x)));
  u_out = (a / c) * sc; v_out = (b / c) * sc;      // xy_index * img_scale (training: the image is rescaled, :53-57,74)
  row = (int)v_out; col = (int)u_out;
  return true;
}

__global__ __launch_bounds__(PB) void v2_count_k(const float* __restrict__ pts, int64_t P, const double* __restrict__ m,
                                                 float fl, float fr, uint8_t* __restrict__ keep,
                                                 int32_t* __restrict__ blk_cnt) {
  const int64_t i = blockIdx.x * (int64_t)PB + threadIdx.x;
  int k = 0;
  if (i < P) {
    int r, c; double v, u; float d;
    k = v2_point(pts + i * 4, m, fl, fr, 1.0, r, c, v, u, d) ? 1 : 0;
    keep[i] = (uint8_t)k;
  }
  const int cnt = __syncthreads_count(k);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = cnt;
}

__global__ __launch_bounds__(PB) void v2_compact_k(const float* __restrict__ pts, int64_t P, const double* __restrict__ m,
                                                   float fl, float fr, double sc, const int32_t* __restrict__ blk_off,
                                                   int32_t* __restrict__ src_idx, int32_t* __restrict__ x_data,
                                                   int32_t* __restrict__ y_data, double* __restrict__ xy,
                                                   float* __restrict__ depth, int32_t* __restrict__ bbox) {
  __shared__ int wave_cnt[PB / 64];
  const int64_t i = blockIdx.x * (int64_t)PB + threadIdx.x;
  int r = 0, c = 0; double v = 0, u = 0; float d = 0.f;
  const bool k = i < P && v2_point(pts + i * 4, m, fl, fr, sc, r, c, v, u, d);
  const unsigned long long bal = __ballot(k);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_cnt[wv] = __popcll(bal);
  __syncthreads();
  int woff = 0;
  for (int j = 0; j < wv; ++j) woff += wave_cnt[j];
  if (k) {
    const int dst = blk_off[blockIdx.x] + woff + before;
    src_idx[dst] = (int)i; x_data[dst] = r; y_data[dst] = c;
    xy[2 * (size_t)dst] = v; xy[2 * (size_t)dst + 1] = u;
    depth[dst] = d;
    atomicMin(bbox + 0, r); atomicMax(bbox + 1, r); atomicMin(bbox + 2, c); atomicMax(bbox + 3, c);
  }
}

extern "C" int pmf_project_v2_index_scaled(const float* points, int64_t P, const double* proj, float fov_left,
                                           float fov_right, double img_scale, uint8_t* keep, int32_t* src_idx,
                                           int32_t* x_data, int32_t* y_data, double* xy_index, float* depth,
                                           int32_t* n_kept, int32_t* bbox, int32_t* blk_cnt, pmf_stream_t s);
extern "C" int pmf_project_v2_index(const float* points, int64_t P, const double* proj, float fov_left, float fov_right,
                                    uint8_t* keep, int32_t* src_idx, int32_t* x_data, int32_t* y_data, double* xy_index,
                                    float* depth, int32_t* n_kept, int32_t* bbox, int32_t* blk_cnt, pmf_stream_t s) {
  return pmf_project_v2_index_scaled(points, P, proj, fov_left, fov_right, 1.0, keep, src_idx, x_data, y_data, xy_index,
                                     depth, n_kept, bbox, blk_cnt, s);
}
extern "C" int pmf_project_v2_index_scaled(const float* points, int64_t P, const double* proj, float fov_left,
                                           float fov_right, double img_scale, uint8_t* keep, int32_t* src_idx,
                                           int32_t* x_data, int32_t* y_data, double* xy_index, float* depth,
                                           int32_t* n_kept, int32_t* bbox, int32_t* blk_cnt, pmf_stream_t s) {
  hipStream_t st = (hipStream_t)s;
  if (P < 0 || !(img_scale > 0.0)) return PMF_E_ARG;
  const int32_t init[4] = {2147483647, -2147483647 - 1, 2147483647, -2147483647 - 1};
  hipError_t e = hipMemcpyAsync(bbox, init, sizeof(init), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return (int)e;
  const int nblk = (int)cdiv64(P > 0 ? P : 1, PB);
  hipLaunchKernelGGL(v2_count_k, dim3(nblk), dim3(PB), 0, st, points, P, proj, fov_left, fov_right, keep, blk_cnt);
  hipLaunchKernelGGL(proj_scan_k, dim3(1), dim3(1024), 0, st, blk_cnt, nblk, n_kept);
  hipLaunchKernelGGL(v2_compact_k, dim3(nblk), dim3(PB), 0, st, points, P, proj, fov_left, fov_right, img_scale, blk_cnt, src_idx,
                     x_data, y_data, xy_index, depth, bbox);
  PMF_LAUNCH_CHECK();
  return 0;
}

__global__ void v2_winner_k(const int32_t* __restrict__ x_data, const int32_t* __restrict__ y_data, int K, int x_min,
                            int y_min, int w, int32_t* __restrict__ pix_idx) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) atomicMax(pix_idx + (size_t)(x_data[k] - x_min) * w + (y_data[k] - y_min), k);   // last kept point wins
}

__global__ void v2_gather_k(const float* __restrict__ pts, const int32_t* __restrict__ sem, const int32_t* __restrict__ src_idx,
                            const float* __restrict__ depth, const uint8_t* __restrict__ img, int ih, int iw,
                            const int32_t* __restrict__ lut, int nlut, const int32_t* __restrict__ pix_idx, int h, int w,
                            int x_min, int y_min, float* __restrict__ out) {
  const int64_t hw = (int64_t)h * w;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
    const int k = pix_idx[p];
    float d = 0.f, x = 0.f, y = 0.f, z = 0.f, it = 0.f, mk = 0.f, lb = 0.f;
    if (k >= 0) {
      const int i = src_idx[k];
      const f32x4 q = *(const f32x4*)(pts + (size_t)i * 4);
      x = q.x; y = q.y; z = q.z; it = q.w;
      d = depth[k];
      mk = 1.f;
      const int sl = sem[i];
      lb = (float)((sl >= 0 && sl < nlut) ? lut[sl] : 0);
    }
    const int r = (int)(p / w), c = (int)(p - (int64_t)r * w);
    const int ir = r + x_min, ic = c + y_min;          // image window (perspective_view_loader_v2.py:105-125)
    float cr = 0.f, cg = 0.f, cb = 0.f;
    if (ir >= 0 && ir < ih && ic >= 0 && ic < iw) {
      const uint8_t* px = img + ((size_t)ir * iw + ic) * 3;
      cr = (float)px[0] / 255.0f; cg = (float)px[1] / 255.0f; cb = (float)px[2] / 255.0f;
    }
    out[0 * hw + p] = d; out[1 * hw + p] = x; out[2 * hw + p] = y; out[3 * hw + p] = z; out[4 * hw + p] = it;
    out[5 * hw + p] = cr; out[6 * hw + p] = cg; out[7 * hw + p] = cb;
    out[8 * hw + p] = mk;
    out[9 * hw + p] = lb;
  }
}

extern "C" int pmf_project_v2_scatter(const float* points, const int32_t* sem, const int32_t* src_idx,
                                      const int32_t* x_data, const int32_t* y_data, const float* depth, int32_t K,
                                      const uint8_t* image, int32_t ih, int32_t iw, const int32_t* lut, int32_t nlut,
                                      int32_t x_min, int32_t y_min, int32_t h, int32_t w, float* proj_out,
                                      int32_t* pix_idx, pmf_stream_t s) {
  hipStream_t st = (hipStream_t)s;
  if (K < 0 || h < 1 || w < 1) return PMF_E_ARG;
  hipError_t e = hipMemsetAsync(pix_idx, 0xFF, (size_t)h * w * 4, st);
  if (e != hipSuccess) return (int)e;
  if (K > 0) hipLaunchKernelGGL(v2_winner_k, dim3(cdiv(K, 256)), dim3(256), 0, st, x_data, y_data, K, x_min, y_min, w, pix_idx);
  const int64_t hw = (int64_t)h * w;
  const int g = (int)cdiv64(hw, 256);
  hipLaunchKernelGGL(v2_gather_k, dim3(g > 2048 ? 2048 : g), dim3(256), 0, st, points, sem, src_idx, depth, image, ih, iw, lut,
                     nlut, pix_idx, h, w, x_min, y_min, proj_out);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ---- training-time tensor augmentation (perspective_view_loader.py:63-69,138-141) -------------------------------------
// RandomHorizontalFlip -> RandomRotation(nearest, zero fill, about the image centre) -> RandomCrop -> Pad, as ONE gather:
// every output pixel maps back through the crop offset, the inverse rotation and the flip to one source pixel.
// Coordinates follow torchvision's tensor path (affine grid over pixel centres, grid_sample(nearest, zeros,
// align_corners=False)): xg = x - w/2 + 0.5, gx = (m0*xg + m1*yg) / (w/2), ix = ((gx + 1)*w - 1)/2, rint(ix) in float32.
__global__ void aug_gather_k(const float* __restrict__ src, int C, int h, int w, int flip, float m0, float m1, float m3,
                             float m4, int top, int left, int ch, int cw, int pad_top, int pad_left, float* __restrict__ dst,
                             int oh, int ow) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  if (ox >= ow) return;
  const int cy = oy - pad_top, cx = ox - pad_left;
  int sy = -1, sx = -1;
  if (cy >= 0 && cy < ch && cx >= 0 && cx < cw) {
    const float xg = (float)(left + cx) - 0.5f * (float)w + 0.5f, yg = (float)(top + cy) - 0.5f * (float)h + 0.5f;
    const float gx = fmaf(yg, m1 / (0.5f * (float)w), xg * (m0 / (0.5f * (float)w)));
    const float gy = fmaf(yg, m4 / (0.5f * (float)h), xg * (m3 / (0.5f * (float)h)));
    const float ix = ((gx + 1.f) * (float)w - 1.f) / 2.f, iy = ((gy + 1.f) * (float)h - 1.f) / 2.f;
    const float rx = rintf(ix), ry = rintf(iy);
    if (rx >= 0.f && rx <= (float)(w - 1) && ry >= 0.f && ry <= (float)(h - 1)) {
      sx = (int)rx; sy = (int)ry;
      if (flip) sx = w - 1 - sx;
    }
  }
  for (int c = 0; c < C; ++c)
    dst[((size_t)c * oh + oy) * ow + ox] = sy >= 0 ? src[((size_t)c * h + sy) * w + sx] : 0.f;
}
extern "C" int pmf_flip_rotate_crop(const float* src, int32_t C, int32_t h, int32_t w, int32_t flip, const float* matrix6,
                                    int32_t top, int32_t left, int32_t crop_h, int32_t crop_w, int32_t pad_top,
                                    int32_t pad_left, float* dst, int32_t oh, int32_t ow, pmf_stream_t s) {
  if (!src || !dst || !matrix6 || C <= 0 || h <= 0 || w <= 0 || crop_h <= 0 || crop_w <= 0 || oh <= 0 || ow <= 0)
    return PMF_E_ARG;
  if (top < 0 || left < 0 || top + crop_h > h || left + crop_w > w || pad_top < 0 || pad_left < 0 ||
      pad_top + crop_h > oh || pad_left + crop_w > ow) return PMF_E_ARG;
  hipLaunchKernelGGL(aug_gather_k, dim3(cdiv(ow, 128), oh), dim3(128), 0, (hipStream_t)s, src, C, h, w, flip, matrix6[0],
                     matrix6[1], matrix6[3], matrix6[4], top, left, crop_h, crop_w, pad_top, pad_left, dst, oh, ow);
  PMF_LAUNCH_CHECK();
  return 0;
}
