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

// ---- the same in TWO launches and without the memset (the loader's per-frame path) ----------------------------------------
// pmf_project_scatter is five enqueues (memset, count, scan, scatter, gather) for ~2 MB of data: launch latency, not work.
// Here (1) ONE kernel projects, compacts and scatters: a block publishes its kept-point count in a slot tagged with the
// call's generation (relaxed agent-scope atomics: the value travels in the same word as the tag, no fence needed) and reads
// the slots of the blocks in front of it.  "In front" is by TICKET, not by blockIdx: a block's logical index is what it
// draws from a device counter when it starts (slots[0]; the gather pass behind this kernel puts the counter back to 0
// for the next call on this workspace, which is stream-ordered behind this one), so every block it waits for has already
// started -- no assumption about the dispatch order of HIP, which promises none, and no deadlock; the winner of a pixel is an atomicMax over (generation << 20 | point index), so entries of
// earlier frames lose against the current one and the per-pixel table is never cleared (the caller zeroes it once every
// 4095 frames); (2) the gather pass.  Same outputs, bit for bit.
#define PROJ_SPIN_MAX (1 << 22)      // ~0.5 s of polling: far beyond any healthy wait (the whole kernel runs < 40 us)
__global__ __launch_bounds__(PB) void proj_fused_k(const float* __restrict__ pts, int64_t P, const double* __restrict__ m,
                                                   int h, int w, uint8_t* __restrict__ keep, float* __restrict__ depth,
                                                   int32_t* __restrict__ x_data, int32_t* __restrict__ y_data,
                                                   unsigned* __restrict__ pix_tag, unsigned long long* __restrict__ slots,
                                                   unsigned gen, int32_t* __restrict__ n_kept) {
  __shared__ int wave_cnt[PB / 64];
  __shared__ int red[PB / 64];
  __shared__ unsigned s_bid;
  if (threadIdx.x == 0) s_bid = (unsigned)atomicAdd(slots, 1ull);       // ticket = logical block index
  __syncthreads();
  const unsigned bid = s_bid;
  // A ticket beyond the grid means the counter was not 0 on entry (an unzeroed or foreign workspace, a call that aborted
  // mid-way): write nothing out of bounds and wait for nobody; bit 63 of the counter tells the gather pass, which is
  // stream-ordered behind this kernel, to report *n_kept = -1 to the host; it puts the counter back to 0 whatever happened
  // here.
  if (bid >= gridDim.x) {
    if (threadIdx.x == 0) atomicOr(slots, 1ull << 63);
    return;
  }
  slots += 1;
  const int64_t i = bid * (int64_t)PB + threadIdx.x;
  int r = 0, c = 0;
  const bool k = i < P && project_point(pts + i * 4, m, h, w, r, c);
  if (i < P) {
    keep[i] = (uint8_t)k;
    const float x = pts[i * 4], y = pts[i * 4 + 1], z = pts[i * 4 + 2];
    depth[i] = sqrtf((x * x + y * y) + z * z);
  }
  const unsigned long long bal = __ballot(k);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_cnt[wv] = __popcll(bal);
  __syncthreads();
  int woff = 0, cnt = 0;
  for (int j = 0; j < PB / 64; ++j) { if (j < wv) woff += wave_cnt[j]; cnt += wave_cnt[j]; }
  if (threadIdx.x == 0)
    __hip_atomic_store(slots + bid, ((unsigned long long)gen << 32) | (unsigned)cnt, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  // counts of the blocks in front: thread t waits for slot t, t + 1024, ...
  int part = 0;
  for (int b = threadIdx.x; b < (int)bid; b += PB) {
    unsigned long long v;
    int spins = 0;
    do {    // (bounded: a slot that is never published -- see the ticket check above -- must not hang the device)
      v = __hip_atomic_load(slots + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } while ((unsigned)(v >> 32) != gen && ++spins < PROJ_SPIN_MAX);
    if ((unsigned)(v >> 32) != gen) { atomicOr(slots - 1, 1ull << 63); v = 0ull; }
    part += (int)(unsigned)v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) red[wv] = part;
  __syncthreads();
  int boff = 0;
  for (int j = 0; j < PB / 64; ++j) boff += red[j];
  if (k) {
    const int dst = boff + woff + before;
    x_data[dst] = r;
    y_data[dst] = c;
    atomicMax(pix_tag + (size_t)r * w + c, (gen << 20) | (unsigned)i);
  }
  if (bid == gridDim.x - 1 && threadIdx.x == 0) *n_kept = boff + cnt;
}

__global__ void proj_gather_tag_k(const float* __restrict__ pts, const int32_t* __restrict__ sem,
                                  const float* __restrict__ depth, const uint8_t* __restrict__ img,
                                  const int32_t* __restrict__ lut, int nlut, const unsigned* __restrict__ pix_tag, unsigned gen,
                                  int h, int w, float* __restrict__ out, unsigned long long* __restrict__ ticket,
                                  int32_t* __restrict__ n_kept) {
  // every ticket of this call's projection kernel is drawn (stream order): the counter goes back to 0 for the next call;
  // bit 63 = that kernel met a ticket beyond its grid or a slot that was never published
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 63) *n_kept = -1;
    __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int64_t hw = (int64_t)h * w;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
    const unsigned tag = pix_tag[p];
    const int i = (tag >> 20) == gen ? (int)(tag & 0xFFFFFu) : -1;
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

extern "C" int pmf_project_scatter2(const float* points, const int32_t* sem, int64_t P, const uint8_t* image, int32_t h,
                                    int32_t w, const double* proj, const int32_t* lut, int32_t nlut, float* proj_out,
                                    uint8_t* keep, int32_t* x_data, int32_t* y_data, float* depth, int32_t* n_kept,
                                    uint32_t* pix_tag, uint64_t* slots, int32_t generation, pmf_stream_t s) {
  hipStream_t st = (hipStream_t)s;
  if (P < 0 || h < 1 || w < 1 || generation < 1 || generation > 4095 || !pix_tag || !slots) return PMF_E_ARG;
  if (P > (1 << 20)) return PMF_E_UNSUPPORTED;            // 20 bits of point index in a pixel's tag
  const int nblk = (int)cdiv64(P > 0 ? P : 1, PB);
  hipLaunchKernelGGL(proj_fused_k, dim3(nblk), dim3(PB), 0, st, points, P, proj, h, w, keep, depth, x_data, y_data,
                     (unsigned*)pix_tag, (unsigned long long*)slots, (unsigned)generation, n_kept);
  const int64_t hw = (int64_t)h * w;
  const int g = (int)cdiv64(hw, 256);
  hipLaunchKernelGGL(proj_gather_tag_k, dim3(g > 2048 ? 2048 : g), dim3(256), 0, st, points, sem, depth, image, lut, nlut,
                     (const unsigned*)pix_tag, (unsigned)generation, h, w, proj_out, (unsigned long long*)slots, n_kept);
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

// ---------------------------------------------------------------------------------------------------------------------
// Image jitter (perspective_view_loader.py:46-49,84-85; perspective_view_loader_v2.py:19-23,46-47: torchvision
// ColorJitter on the PIL image) on the uint8 [h*w][3] frame, in place, bit for bit what Pillow computes:
//   brightness / contrast / saturation = Image.blend(degenerate, image, f): float32  d + f * (x - d), truncated (clipped to
//   [0, 255] when f is outside [0, 1]); degenerate = 0 / the rounded mean luma of the whole image / the pixel's luma
//   (luma = (19595 R + 38470 G + 7471 B + 0x8000) >> 16);
//   hue = RGB -> HSV (Convert.c: float32 ratios, float64 hue arithmetic, truncation), H += shift (uint8 wrap), HSV -> RGB.
// Each operation reads the uint8 result of the previous one, as the PIL pipeline does.
__device__ __forceinline__ int cj_luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ uint8_t cj_blend1(int d, int x, float a, bool interp) {
  // plain operators under `fp contract(off)`: the __fmul_rn / __fadd_rn header inlines carry the contract flag and
  // fuse into one FMA (128 + 0.6f * -125 then truncates to 52 where Pillow's two roundings give 53)
  const float m = a * (float)(x - d);
  const float t = (float)d + m;
  if (interp) return (uint8_t)(int)t;
  return t <= 0.f ? (uint8_t)0 : (t >= 255.f ? (uint8_t)255 : (uint8_t)(int)t);
}

// the whole jitter of one pixel in registers: the enabled operations in their drawn order, each on the uint8 result of the one
// before it (what the PIL pipeline computes).  Round 5: ONE read + ONE write of the frame (plus one read-only pass when contrast
// is among them: its degenerate image is the mean luma of the frame AS THE OPERATIONS IN FRONT OF IT LEFT IT) instead of one
// read-modify-write pass per operation.
struct CjArgs {
  int nops;          // enabled operations, in order
  int op[4];         // 0 brightness, 1 contrast, 2 saturation, 3 hue
  float a[4];        // blend factor of op[k]
  int shift;         // hue shift (uint8)
};

__device__ __forceinline__ void cj_hue1(int& r, int& g, int& b, int shift) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;        // float32 divisions are correctly rounded (hipcc default)
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    const double hh = (double)h / 6.0 + 1.0;
    h = (float)(hh - floor(hh));                       // fmod(x, 1.0) for x >= 0
    uh = min(max((int)((double)h * 255.0), 0), 255);
    us = min(max((int)((double)s * 255.0), 0), 255);
  }
  uh = (uh + shift) & 0xff;
  int ro = uv, go = uv, bo = uv;
  if (us != 0) {
    const double h6 = (double)(float)uh * 6.0 / 255.0;
    const int i = (int)floor(h6);
    const double f = (double)(float)(h6 - (double)(float)i);
    const double fs = (double)(float)((double)(float)us / 255.0);
    const double vf = (double)(float)uv;
    const int pp = min(max((int)floor(vf * (1.0 - fs) + 0.5), 0), 255);
    const int q = min(max((int)floor(vf * (1.0 - fs * f) + 0.5), 0), 255);
    const int t = min(max((int)floor(vf * (1.0 - fs * (1.0 - f)) + 0.5), 0), 255);
    switch (i % 6) {
      case 0: ro = uv; go = t; bo = pp; break;
      case 1: ro = q; go = uv; bo = pp; break;
      case 2: ro = pp; go = uv; bo = t; break;
      case 3: ro = pp; go = q; bo = uv; break;
      case 4: ro = t; go = pp; bo = uv; break;
      default: ro = uv; go = pp; bo = q; break;
    }
  }
  r = ro; g = go; b = bo;
}

// operations [k0, k1) of A on one pixel; `mean` = the contrast operation's degenerate value
__device__ __forceinline__ void cj_pixel(int& r, int& g, int& b, const CjArgs& A, int k0, int k1, int mean) {
  for (int k = k0; k < k1; ++k) {
    const int op = A.op[k];
    if (op == 3) { cj_hue1(r, g, b, A.shift); continue; }
    const float a = A.a[k];
    const bool interp = a >= 0.f && a <= 1.f;
    const int d = op == 0 ? 0 : (op == 1 ? mean : cj_luma(r, g, b));
    const int r2 = cj_blend1(d, r, a, interp), g2 = cj_blend1(d, g, a, interp), b2 = cj_blend1(d, b, a, interp);
    r = r2; g = g2; b = b2;
  }
}

// a thread owns four pixels = three aligned dwords (the frame is 4-byte aligned: a torch allocation); tail pixels byte by byte
template <bool WRITE>
__global__ __launch_bounds__(256) void cj_fused_k(uint8_t* __restrict__ img, int64_t npix, const CjArgs A, int k0, int k1,
                                                  unsigned long long* __restrict__ sum) {
  const int64_t quads = npix >> 2;
  int mean = 0;
  if (WRITE && sum) mean = (int)((double)*sum / (double)npix + 0.5);
  unsigned long long acc = 0;
  for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < quads + (npix & 3); t += (int64_t)gridDim.x * 256) {
    if (t < quads) {
      uint32_t* w = (uint32_t*)img + t * 3;
      const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
      int c[12];
#pragma unroll
      for (int i = 0; i < 4; ++i) { c[i] = (w0 >> (8 * i)) & 255; c[4 + i] = (w1 >> (8 * i)) & 255; c[8 + i] = (w2 >> (8 * i)) & 255; }
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        cj_pixel(c[3 * px], c[3 * px + 1], c[3 * px + 2], A, k0, k1, mean);
        if (!WRITE) acc += (unsigned)cj_luma(c[3 * px], c[3 * px + 1], c[3 * px + 2]);
      }
      if (WRITE) {
        w[0] = (uint32_t)c[0] | (uint32_t)c[1] << 8 | (uint32_t)c[2] << 16 | (uint32_t)c[3] << 24;
        w[1] = (uint32_t)c[4] | (uint32_t)c[5] << 8 | (uint32_t)c[6] << 16 | (uint32_t)c[7] << 24;
        w[2] = (uint32_t)c[8] | (uint32_t)c[9] << 8 | (uint32_t)c[10] << 16 | (uint32_t)c[11] << 24;
      }
    } else {
      const int64_t p = quads * 4 + (t - quads);
      int r = img[p * 3], g = img[p * 3 + 1], b = img[p * 3 + 2];
      cj_pixel(r, g, b, A, k0, k1, mean);
      if (WRITE) { img[p * 3] = (uint8_t)r; img[p * 3 + 1] = (uint8_t)g; img[p * 3 + 2] = (uint8_t)b; }
      else acc += (unsigned)cj_luma(r, g, b);
    }
  }
  if (!WRITE) {
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ unsigned long long sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, sh[0] + sh[1] + sh[2] + sh[3]);     // integer sum: order-independent
  }
}

extern "C" int pmf_color_jitter(uint8_t* image, int32_t h, int32_t w, const int32_t* order4, const double* factor4,
                                const int32_t* enabled4, uint64_t* scratch, pmf_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  if (!image || !order4 || !factor4 || !enabled4 || !scratch || h < 1 || w < 1) return PMF_E_ARG;
  if (((uintptr_t)image & 3) != 0) return PMF_E_ARG;            // dword accesses
  const int64_t npix = (int64_t)h * w;
  CjArgs A;
  A.nops = 0; A.shift = 0;
  int kc = -1;                                                  // position of the contrast operation
  for (int k = 0; k < 4; ++k) {
    const int op = order4[k];
    if (op < 0 || op > 3) return PMF_E_ARG;
    if (!enabled4[op]) continue;
    if (op == 3) {
      if (!(factor4[3] >= -0.5 && factor4[3] <= 0.5)) return PMF_E_ARG;
      A.shift = (int)(factor4[3] * 255.0) & 0xff;
    }
    if (op == 1) kc = A.nops;
    A.op[A.nops] = op; A.a[A.nops] = (float)factor4[op];
    ++A.nops;
  }
  for (int k = A.nops; k < 4; ++k) { A.op[k] = 0; A.a[k] = 1.f; }
  if (A.nops == 0) return 0;
  const int64_t work = (npix >> 2) + (npix & 3);
  unsigned grid = (unsigned)cdiv64(work, 256);
  if (kc >= 0) {     // mean luma of the frame as the operations in front of the contrast leave it: a read-only pass
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(uint64_t), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cj_fused_k<false>, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, s, image, npix, A, 0, kc,
                       (unsigned long long*)scratch);
    PMF_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(cj_fused_k<true>, dim3(grid), dim3(256), 0, s, image, npix, A, 0, A.nops,
                     kc >= 0 ? (unsigned long long*)scratch : (unsigned long long*)nullptr);
  PMF_LAUNCH_CHECK();
  return 0;
}
