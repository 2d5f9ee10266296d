// HBM-bound companion kernels of the conv stack (gfx950): residual adds, pooling, bilinear x2, PixelShuffle,
// fusion gate, channel softmax, layout change, column reductions.  All NHWC fp32, one float4 (4 channels) per
// lane so a wavefront moves 1 KiB per instruction; "views" fold BatchNorm-apply / ReLU / Dropout2d into the load.
#include "common.h"

#define EW_BLOCK 256
static inline int ew_grid(int64_t items) {
  int64_t b = cdiv64(items, EW_BLOCK);
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

__device__ __forceinline__ f32x4 ldv(const pmf_view_t& v, int64_t pix, int n, int c) {
  return pmf_view_load4(v.x, v.scale, v.shift, v.cmul ? v.cmul + (size_t)n * v.cmul_ld : nullptr, v.flags,
                        (size_t)pix * v.ldc + c, c);
}
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// ------------------------------------------------------------------ add + activation
__global__ void add_act_k(pmf_view_t a, pmf_view_t b, int has_b, int act, float* __restrict__ out, int out_ldc,
                          int64_t npix, int HW, int Q) {
  const int64_t total = npix * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4, n = (int)(p / HW);
    f32x4 v = ldv(a, p, n, c);
    if (has_b) v += ldv(b, p, n, c);
    v.x = pmf_act(v.x, act); v.y = pmf_act(v.y, act); v.z = pmf_act(v.z, act); v.w = pmf_act(v.w, act);
    *(f32x4*)(out + p * out_ldc + c) = v;
  }
}
extern "C" int pmf_add_act(const pmf_view_t* a, const pmf_view_t* b, int32_t act, float* out, int32_t out_ldc,
                           int64_t npix, int32_t HW, int32_t C, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  pmf_view_t bb = b ? *b : *a;
  hipLaunchKernelGGL(add_act_k, dim3(ew_grid(npix * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, *a, bb, b != nullptr,
                     act, out, out_ldc, npix, HW, C / 4);
  PMF_LAUNCH_CHECK();
  return 0;
}

__global__ void add_act_bwd_k(const float* __restrict__ gout, int g_ldc, const float* __restrict__ out, int out_ldc,
                              int act, float* ga, int ga_ldc, int ga_acc, float* gb, int gb_ldc, int gb_acc,
                              int64_t npix, int Q) {
  const int64_t total = npix * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4;
    f32x4 g = *(const f32x4*)(gout + p * g_ldc + c);
    if (act == PMF_ACT_RELU) {
      f32x4 o = *(const f32x4*)(out + p * out_ldc + c);
      g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
    }
    if (ga) {
      f32x4* q = (f32x4*)(ga + p * ga_ldc + c);
      *q = ga_acc ? *q + g : g;
    }
    if (gb) {
      f32x4* q = (f32x4*)(gb + p * gb_ldc + c);
      *q = gb_acc ? *q + g : g;
    }
  }
}
extern "C" int pmf_add_act_bwd(const float* gout, int32_t g_ldc, const float* out, int32_t out_ldc, int32_t act,
                               float* ga, int32_t ga_ldc, int32_t ga_acc, float* gb, int32_t gb_ldc, int32_t gb_acc,
                               int64_t npix, int32_t C, pmf_stream_t s) {
  if (C % 4 || (act != PMF_ACT_NONE && act != PMF_ACT_RELU)) return PMF_E_ARG;
  hipLaunchKernelGGL(add_act_bwd_k, dim3(ew_grid(npix * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, gout, g_ldc, out,
                     out_ldc, act, ga, ga_ldc, ga_acc, gb, gb_ldc, gb_acc, npix, C / 4);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ column-reduction skeleton
// blockDim = rows*Qg (<=256); thread -> (row, channel quad); per-thread partials, LDS fold, one atomic per channel.
struct ColLaunch { dim3 grid, block; };
static ColLaunch col_launch(int64_t npix, int Q, int nz) {
  int Qg = Q < 256 ? Q : 256;
  int rows = 256 / Qg;
  int64_t gx = cdiv64(npix, (int64_t)rows * 4);   // same rule as bn.hip col_l / pmf_col_rows
  if (gx > g_pmf_col_cap) gx = g_pmf_col_cap;
  if (gx < 1) gx = 1;
  ColLaunch L;
  L.grid = dim3((unsigned)gx, (unsigned)cdiv(Q, 256), (unsigned)nz);
  L.block = dim3(rows * Qg);
  return L;
}
#define COL_SETUP(Q)                                                   \
  const int Qg_ = min((Q) - (int)blockIdx.y * 256, 256);               \
  const int rows_ = 256 / min((Q), 256);                               \
  const int tid_ = threadIdx.x;                                        \
  const int row_ = tid_ / Qg_max_(Q), cql_ = tid_ - row_ * Qg_max_(Q); \
  const bool active_ = cql_ < Qg_;                                     \
  const int c = ((int)blockIdx.y * 256 + cql_) * 4;
__device__ __forceinline__ int Qg_max_(int Q) { return Q < 256 ? Q : 256; }

__device__ __forceinline__ void col_fold_atomic(f32x4 part, float* dst, int c, int row, int cql, int rows, int Qgm,
                                                bool active, f32x4* sh) {
  sh[row * Qgm + cql] = part;
  __syncthreads();
  if (row == 0 && active) {
    for (int r = 1; r < rows; ++r) part += sh[r * Qgm + cql];
    atomicAdd(dst + c + 0, part.x); atomicAdd(dst + c + 1, part.y);
    atomicAdd(dst + c + 2, part.z); atomicAdd(dst + c + 3, part.w);
  }
  __syncthreads();
}

// g *= act'(a) in place (act NONE: g untouched); per-workgroup partial column sums of the result go to
// dbias_rows[blockIdx.x][*] (no atomics) for the weight-gradient kernel to fold into the conv-bias gradient
__global__ void act_bwd_k(float* __restrict__ g, int g_ldc, const float* __restrict__ a, int a_ldc, int act,
                          float* dbias_rows, int dbias_ld, int64_t npix, int Q) {
  __shared__ f32x4 sh[256];
  COL_SETUP(Q)
  f32x4 part = zero4();
  if (active_) {
    // four pixels per trip: eight loads in flight per thread (one pixel per trip left the 512 workgroups of this launch
    // with 4 MB in flight chip-wide, a quarter of what HBM needs); same order of the column sum
    const int64_t step = (int64_t)gridDim.x * rows_;
    const float sl = act == PMF_ACT_LRELU ? 0.01f : 0.f;
    for (int64_t p = (int64_t)blockIdx.x * rows_ + row_; p < npix; p += 4 * step) {
      f32x4 v[4], x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t pp = p + u * step;
        if (pp < npix) {
          v[u] = *(const f32x4*)(g + pp * g_ldc + c);
          if (act != PMF_ACT_NONE) x[u] = *(const f32x4*)(a + pp * a_ldc + c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t pp = p + u * step;
        if (pp < npix) {
          if (act != PMF_ACT_NONE) {
            v[u].x *= x[u].x > 0.f ? 1.f : sl; v[u].y *= x[u].y > 0.f ? 1.f : sl;
            v[u].z *= x[u].z > 0.f ? 1.f : sl; v[u].w *= x[u].w > 0.f ? 1.f : sl;
            *(f32x4*)(g + pp * g_ldc + c) = v[u];
          }
          part += v[u];
        }
      }
    }
  }
  if (dbias_rows) {
    sh[row_ * Qg_max_(Q) + cql_] = part;
    __syncthreads();
    if (row_ == 0 && active_) {
      for (int r = 1; r < rows_; ++r) part += sh[r * Qg_max_(Q) + cql_];
      *(f32x4*)(dbias_rows + (size_t)blockIdx.x * dbias_ld + c) = part;
    }
  }
}
extern "C" int pmf_act_bwd(float* g, int32_t g_ldc, const float* a, int32_t a_ldc, int32_t act, float* dbias_rows,
                           int32_t dbias_ld, int64_t npix, int32_t C, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  ColLaunch L = col_launch(npix, C / 4, 1);
  hipLaunchKernelGGL(act_bwd_k, L.grid, L.block, 0, (hipStream_t)s, g, g_ldc, a, a_ldc, act, dbias_rows, dbias_ld, npix,
                     C / 4);
  PMF_LAUNCH_CHECK();
  return 0;
}

// out[z][c] += sum_p x[z][p][c]
__global__ void colsum_k(const float* __restrict__ x, int ldc, int64_t npix, int Q, float* out, int64_t x_sn,
                         int64_t out_sn, float mul) {
  __shared__ f32x4 sh[256];
  COL_SETUP(Q)
  const float* xz = x + blockIdx.z * x_sn;
  f32x4 part = zero4();
  if (active_)
    for (int64_t p = (int64_t)blockIdx.x * rows_ + row_; p < npix; p += (int64_t)gridDim.x * rows_)
      part += *(const f32x4*)(xz + p * ldc + c);
  part *= mul;
  col_fold_atomic(part, out + blockIdx.z * out_sn, c, row_, cql_, rows_, Qg_max_(Q), active_, sh);
}
extern "C" int pmf_colsum(const float* x, int32_t ldc, int64_t npix, int32_t C, float* out, int32_t nz,
                          pmf_stream_t s) {
  // nz > 1: per-sample sums, x advances npix*ldc and out advances C per sample
  if (C % 4) return PMF_E_ARG;
  if (nz < 1) nz = 1;
  ColLaunch L = col_launch(npix, C / 4, nz);
  hipLaunchKernelGGL(colsum_k, L.grid, L.block, 0, (hipStream_t)s, x, ldc, npix, C / 4, out, npix * (int64_t)ldc,
                     (int64_t)C, 1.f);
  PMF_LAUNCH_CHECK();
  return 0;
}

// The same sums without atomics: stage 1 writes one partial row per workgroup, stage 2 folds the rows of a channel in a
// fixed order and adds the result to out -- deterministic (the atomic form's result depends on the arrival order of up
// to 512 workgroups per address, which also serialises them: 29 us for a 32-channel bias gradient at 64x2048, 11 us here).
// rows: [nz][gridDim.x of stage 1][C] floats of scratch.
__global__ void colsum_rows_k(const float* __restrict__ x, int ldc, int64_t npix, int Q, float* __restrict__ rows, int64_t x_sn,
                              int C) {
  __shared__ f32x4 sh[256];
  COL_SETUP(Q)
  const float* xz = x + blockIdx.z * x_sn;
  f32x4 part = zero4();
  if (active_)
    for (int64_t p = (int64_t)blockIdx.x * rows_ + row_; p < npix; p += (int64_t)gridDim.x * rows_)
      part += *(const f32x4*)(xz + p * ldc + c);
  sh[row_ * Qg_max_(Q) + cql_] = part;
  __syncthreads();
  if (row_ == 0 && active_) {
    for (int r = 1; r < rows_; ++r) part += sh[r * Qg_max_(Q) + cql_];
    *(f32x4*)(rows + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * C + c) = part;
  }
}
__global__ void colsum_fold_k(const float* __restrict__ rows, int nrows, int C, float* out, int64_t out_sn, float mul) {
  __shared__ float sh[256];
  const int c = blockIdx.x, z = blockIdx.y;      // one workgroup per (channel, sample): rows dealt to the threads in order
  float s = 0.f;
  for (int r = threadIdx.x; r < nrows; r += 256) s += rows[((size_t)z * nrows + r) * C + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[z * out_sn + c] += sh[0] * mul;
}
extern "C" int pmf_colsum_rows(const float* x, int32_t ldc, int64_t npix, int32_t C, float* out, int32_t nz, float* scratch,
                               pmf_stream_t s) {
  if (C % 4 || !scratch) return PMF_E_ARG;
  if (nz < 1) nz = 1;
  ColLaunch L = col_launch(npix, C / 4, nz);
  hipLaunchKernelGGL(colsum_rows_k, L.grid, L.block, 0, (hipStream_t)s, x, ldc, npix, C / 4, scratch, npix * (int64_t)ldc, C);
  hipLaunchKernelGGL(colsum_fold_k, dim3(C, nz), dim3(256), 0, (hipStream_t)s, (const float*)scratch, (int)L.grid.x, C, out,
                     (int64_t)C, 1.f);
  PMF_LAUNCH_CHECK();
  return 0;
}

// per-(n,c) mean of a view (the ASPP image-level feature: the map is H/16 x W/16).  R = 256 / min(C/4, 256) row threads
// per channel quad walk the pixels in a fixed order and their partial sums fold in a fixed order: deterministic, no
// atomics -- the summation order of rounds 1-2, kept so that results stay bit-identical.  What changed: a workgroup now
// holds 16 channel quads (R x 16 threads) instead of all of them, and a row thread keeps eight loads in flight with the
// view transform fetched once (round 2: ONE workgroup per sample, one load at a time: 40 -> 17 us for 256 channels at
// 4x128, 128 -> 36 us for EPMF's 512, on the main lane).
#define GM_Q 16
__global__ __launch_bounds__(256) void gmean_k(pmf_view_t v, int HW, int Q, int R, int QW, float* out, int C) {
  __shared__ f32x4 sh[256];
  const int tid = threadIdx.x, ql = tid % QW, row = tid / QW;           // R rows x QW quads
  const int q = (int)blockIdx.y * QW + ql, n = blockIdx.z;
  const bool active = q < Q;
  const int c = q * 4;
  f32x4 part = zero4();
  if (active) {
    // the view's per-channel transform is the same for every pixel of this thread: fetched once (pmf_view_load4's
    // expressions, so that the values are the ones the other kernels see)
    const bool aff = v.scale != nullptr, relu = (v.flags & PMF_SRC_RELU) != 0, has_cm = v.cmul != nullptr;
    f32x4 sc = zero4(), sf = zero4(), cm = zero4();
    if (aff) { sc = *(const f32x4*)(v.scale + c); sf = *(const f32x4*)(v.shift + c); }
    if (has_cm) cm = *(const f32x4*)(v.cmul + (size_t)n * v.cmul_ld + c);
    const float* __restrict__ xb = v.x + (size_t)n * HW * v.ldc + c;
    auto xf = [&](f32x4 t) {
      if (aff) t = t * sc + sf;
      if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      if (has_cm) t = t * cm;
      return t;
    };
    int64_t p = row;
    for (; p + 7 * (int64_t)R < HW; p += 8 * (int64_t)R) {
      f32x4 a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = *(const f32x4*)(xb + (size_t)(p + j * R) * v.ldc);
#pragma unroll
      for (int j = 0; j < 8; ++j) part += xf(a[j]);
    }
    for (; p < HW; p += R) part += xf(*(const f32x4*)(xb + (size_t)p * v.ldc));
  }
  sh[row * QW + ql] = part;
  __syncthreads();
  if (row == 0 && active) {
    for (int r = 1; r < R; ++r) part += sh[r * QW + ql];
    part *= 1.f / (float)HW;
    *(f32x4*)(out + (size_t)n * C + c) = part;
  }
}
extern "C" int pmf_global_mean(const pmf_view_t* in, int32_t N, int32_t HW, int32_t C, float* out, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  const int Q = C / 4, R = 256 / (Q < 256 ? Q : 256), QW = Q < GM_Q ? Q : GM_Q;
  hipLaunchKernelGGL(gmean_k, dim3(1, cdiv(Q, QW), N), dim3(R * QW), 0, (hipStream_t)s, *in, HW, Q, R, QW, out, C);
  PMF_LAUNCH_CHECK();
  return 0;
}
__global__ void gmean_bwd_k(const float* __restrict__ gout, int HW, int Q, int C, const float* __restrict__ cmul,
                            int cmul_ld, float* gin, int gin_ldc, int acc, int64_t npix) {
  const int64_t total = npix * Q;
  const float inv = 1.f / (float)HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4, n = (int)(p / HW);
    f32x4 g = *(const f32x4*)(gout + (size_t)n * C + c) * inv;
    if (cmul) g *= *(const f32x4*)(cmul + (size_t)n * cmul_ld + c);
    f32x4* q = (f32x4*)(gin + p * gin_ldc + c);
    *q = acc ? *q + g : g;
  }
}
extern "C" int pmf_global_mean_bwd(const float* gout, int32_t N, int32_t HW, int32_t C, const float* cmul,
                                   int32_t cmul_ld, float* gin, int32_t gin_ldc, int32_t acc, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  const int64_t npix = (int64_t)N * HW;
  hipLaunchKernelGGL(gmean_bwd_k, dim3(ew_grid(npix * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, gout, HW, C / 4, C,
                     cmul, cmul_ld, gin, gin_ldc, acc, npix);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ AvgPool2d(3, 2, 1), count_include_pad
__global__ void avgpool_k(pmf_view_t v, int N, int H, int W, int OH, int OW, int Q, float* __restrict__ out,
                          int out_ldc) {
  const int64_t total = (int64_t)N * OH * OW * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t op = i / Q;
    const int c = (int)(i - op * Q) * 4;
    const int ox = (int)(op % OW);
    const int oy = (int)((op / OW) % OH), n = (int)(op / ((int64_t)OW * OH));
    f32x4 acc = zero4();
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int y = 2 * oy + dy, x = 2 * ox + dx;
        if (y >= 0 && y < H && x >= 0 && x < W) acc += ldv(v, ((int64_t)n * H + y) * W + x, n, c);
      }
    *(f32x4*)(out + op * out_ldc + c) = acc / 9.0f;
  }
}
extern "C" int pmf_avgpool3s2(const pmf_view_t* in, int32_t N, int32_t H, int32_t W, int32_t C, float* out,
                              int32_t out_ldc, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(avgpool_k, dim3(ew_grid((int64_t)N * OH * OW * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, *in,
                     N, H, W, OH, OW, C / 4, out, out_ldc);
  PMF_LAUNCH_CHECK();
  return 0;
}
__global__ void avgpool_bwd_k(const float* __restrict__ gout, int g_ldc, int N, int H, int W, int OH, int OW, int Q,
                              const float* __restrict__ cmul, int cmul_ld, float* gin, int gin_ldc, int acc) {
  const int64_t total = (int64_t)N * H * W * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4;
    const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    f32x4 g = zero4();
    const int oy0 = y >> 1, oy1 = (y + 1) >> 1, ox0 = x >> 1, ox1 = (x + 1) >> 1;
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox)
        if (oy < OH && ox < OW) g += *(const f32x4*)(gout + (((int64_t)n * OH + oy) * OW + ox) * g_ldc + c);
    g = g / 9.0f;
    if (cmul) g *= *(const f32x4*)(cmul + (size_t)n * cmul_ld + c);
    f32x4* q = (f32x4*)(gin + p * gin_ldc + c);
    *q = acc ? *q + g : g;
  }
}
extern "C" int pmf_avgpool3s2_bwd(const float* gout, int32_t g_ldc, int32_t N, int32_t H, int32_t W, int32_t C,
                                  const float* cmul, int32_t cmul_ld, float* gin, int32_t gin_ldc, int32_t acc,
                                  pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(avgpool_bwd_k, dim3(ew_grid((int64_t)N * H * W * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, gout,
                     g_ldc, N, H, W, OH, OW, C / 4, cmul, cmul_ld, gin, gin_ldc, acc);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ MaxPool2d(3, 2, 1) on a view, argmax kept
__global__ void maxpool_k(pmf_view_t v, int N, int H, int W, int OH, int OW, int Q, float* __restrict__ out,
                          int out_ldc, uint8_t* __restrict__ idx, int C) {
  const int64_t total = (int64_t)N * OH * OW * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t op = i / Q;
    const int c = (int)(i - op * Q) * 4;
    const int ox = (int)(op % OW);
    const int oy = (int)((op / OW) % OH), n = (int)(op / ((int64_t)OW * OH));
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int y = 2 * oy + dy, x = 2 * ox + dx;
        if (y >= 0 && y < H && x >= 0 && x < W) {
          f32x4 t = ldv(v, ((int64_t)n * H + y) * W + x, n, c);
          const int pos = (dy + 1) * 3 + dx + 1;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (first || t[k] > m[k] || t[k] != t[k]) { m[k] = t[k]; bi[k] = pos; }
          first = false;
        }
      }
    *(f32x4*)(out + op * out_ldc + c) = m;
    if (idx) *(uchar4*)(idx + op * C + c) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
  }
}
extern "C" int pmf_maxpool3s2(const pmf_view_t* in, int32_t N, int32_t H, int32_t W, int32_t C, float* out,
                              int32_t out_ldc, uint8_t* idx, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_k, dim3(ew_grid((int64_t)N * OH * OW * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, *in,
                     N, H, W, OH, OW, C / 4, out, out_ldc, idx, C);
  PMF_LAUNCH_CHECK();
  return 0;
}
__global__ void maxpool_bwd_k(const float* __restrict__ gout, int g_ldc, const uint8_t* __restrict__ idx, int N, int H,
                              int W, int OH, int OW, int Q, int C, pmf_view_t v, float* gin, int gin_ldc, int acc) {
  const int64_t total = (int64_t)N * H * W * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4;
    const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    f32x4 g = zero4();
    const int oy0 = y >> 1, oy1 = (y + 1) >> 1, ox0 = x >> 1, ox1 = (x + 1) >> 1;
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox)
        if (oy < OH && ox < OW) {
          const int pos = (y - 2 * oy + 1) * 3 + (x - 2 * ox + 1);
          const int64_t op = ((int64_t)n * OH + oy) * OW + ox;
          const uchar4 b = *(const uchar4*)(idx + op * C + c);
          const f32x4 t = *(const f32x4*)(gout + op * g_ldc + c);
          if (b.x == pos) g.x += t.x;
          if (b.y == pos) g.y += t.y;
          if (b.z == pos) g.z += t.z;
          if (b.w == pos) g.w += t.w;
        }
    if (v.flags & PMF_SRC_RELU) {  // gradient w.r.t. the BN output: mask by relu'
      pmf_view_t vv = v;
      vv.flags &= ~PMF_SRC_RELU;
      f32x4 yv = ldv(vv, p, n, c);
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    }
    f32x4* q = (f32x4*)(gin + p * gin_ldc + c);
    *q = acc ? *q + g : g;
  }
}
extern "C" int pmf_maxpool3s2_bwd(const float* gout, int32_t g_ldc, const uint8_t* idx, int32_t N, int32_t H,
                                  int32_t W, int32_t C, const pmf_view_t* in, float* gin, int32_t gin_ldc, int32_t acc,
                                  pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_k, dim3(ew_grid((int64_t)N * H * W * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, gout,
                     g_ldc, idx, N, H, W, OH, OW, C / 4, C, *in, gin, gin_ldc, acc);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ bilinear x2, align_corners = False
__device__ __forceinline__ void bil_src(int o, int L, int& i0, int& i1, float& l1) {
  float s = 0.5f * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + (i0 < L - 1 ? 1 : 0);
  l1 = s - (float)i0;
}
__global__ void bilinear_k(pmf_view_t v, int N, int H, int W, int Q, float* __restrict__ out, int out_ldc) {
  const int OH = 2 * H, OW = 2 * W;
  const int64_t total = (int64_t)N * OH * OW * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t op = i / Q;
    const int c = (int)(i - op * Q) * 4;
    const int ox = (int)(op % OW), oy = (int)((op / OW) % OH), n = (int)(op / ((int64_t)OW * OH));
    int y0, y1, x0, x1;
    float ly, lx;
    bil_src(oy, H, y0, y1, ly);
    bil_src(ox, W, x0, x1, lx);
    const int64_t b = (int64_t)n * H;
    f32x4 v00 = ldv(v, (b + y0) * W + x0, n, c), v01 = ldv(v, (b + y0) * W + x1, n, c);
    f32x4 v10 = ldv(v, (b + y1) * W + x0, n, c), v11 = ldv(v, (b + y1) * W + x1, n, c);
    f32x4 r = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    *(f32x4*)(out + op * out_ldc + c) = r;
  }
}
extern "C" int pmf_bilinear2x(const pmf_view_t* in, int32_t N, int32_t H, int32_t W, int32_t C, float* out,
                              int32_t out_ldc, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  hipLaunchKernelGGL(bilinear_k, dim3(ew_grid((int64_t)N * 4 * H * W * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, *in,
                     N, H, W, C / 4, out, out_ldc);
  PMF_LAUNCH_CHECK();
  return 0;
}
__device__ __forceinline__ void bil_wts(int y, int L, float w[5]) {  // weight of input y in outputs 2y-2 .. 2y+2
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int o = 2 * y - 2 + k;
    float ww = 0.f;
    if (o >= 0 && o < 2 * L) {
      int i0, i1;
      float l1;
      bil_src(o, L, i0, i1, l1);
      if (i0 == y) ww += 1.f - l1;
      if (i1 == y) ww += l1;
    }
    w[k] = ww;
  }
}
__global__ void bilinear_bwd_k(const float* __restrict__ gout, int g_ldc, int N, int H, int W, int Q, float* gin,
                               int gin_ldc, int acc) {
  const int OH = 2 * H, OW = 2 * W;
  const int64_t total = (int64_t)N * H * W * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4;
    const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    float wy[5], wx[5];
    bil_wts(y, H, wy);
    bil_wts(x, W, wx);
    f32x4 g = zero4();
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      if (wy[a] == 0.f) continue;
      const int oy = 2 * y - 2 + a;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        if (wx[b] == 0.f) continue;
        const int ox = 2 * x - 2 + b;
        g += (wy[a] * wx[b]) * *(const f32x4*)(gout + (((int64_t)n * OH + oy) * OW + ox) * g_ldc + c);
      }
    }
    f32x4* q = (f32x4*)(gin + p * gin_ldc + c);
    *q = acc ? *q + g : g;
  }
}
extern "C" int pmf_bilinear2x_bwd(const float* gout, int32_t g_ldc, int32_t N, int32_t H, int32_t W, int32_t C,
                                  float* gin, int32_t gin_ldc, int32_t acc, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  hipLaunchKernelGGL(bilinear_bwd_k, dim3(ew_grid((int64_t)N * H * W * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, gout,
                     g_ldc, N, H, W, C / 4, gin, gin_ldc, acc);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ PixelShuffle(2): out[n,2h+i,2w+j,c] = in[n,h,w,4c+2i+j]
__global__ void pshuffle_k(pmf_view_t v, int N, int H, int W, int Co, const float* __restrict__ ocm, int ocm_ld,
                           float* __restrict__ out, int out_ldc) {
  const int64_t total = (int64_t)N * H * W * Co;  // one float4 of the input (= 4 output pixels of channel c) per item
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Co;
    const int c = (int)(i - p * Co);
    const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    f32x4 t = ldv(v, p, n, 4 * c);
    const float m = ocm ? ocm[(size_t)n * ocm_ld + c] : 1.f;
    const int64_t ob = ((int64_t)n * 2 * H + 2 * y) * (2 * W) + 2 * x;
    out[(ob) * out_ldc + c] = t.x * m;
    out[(ob + 1) * out_ldc + c] = t.y * m;
    out[(ob + 2 * W) * out_ldc + c] = t.z * m;
    out[(ob + 2 * W + 1) * out_ldc + c] = t.w * m;
  }
}
extern "C" int pmf_pixel_shuffle2(const pmf_view_t* in, int32_t N, int32_t H, int32_t W, int32_t Cout,
                                  const float* out_cmul, int32_t out_cmul_ld, float* out, int32_t out_ldc,
                                  pmf_stream_t s) {
  hipLaunchKernelGGL(pshuffle_k, dim3(ew_grid((int64_t)N * H * W * Cout)), dim3(EW_BLOCK), 0, (hipStream_t)s, *in, N, H, W,
                     Cout, out_cmul, out_cmul_ld, out, out_ldc);
  PMF_LAUNCH_CHECK();
  return 0;
}
__global__ void pshuffle_bwd_k(const float* __restrict__ gout, int g_ldc, int N, int H, int W, int Co,
                               const float* __restrict__ ocm, int ocm_ld, const float* __restrict__ icm, int icm_ld,
                               float* gin, int gin_ldc, int acc) {
  const int64_t total = (int64_t)N * H * W * Co;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Co;
    const int c = (int)(i - p * Co);
    const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    const int64_t ob = ((int64_t)n * 2 * H + 2 * y) * (2 * W) + 2 * x;
    f32x4 g;
    g.x = gout[(ob) * g_ldc + c];
    g.y = gout[(ob + 1) * g_ldc + c];
    g.z = gout[(ob + 2 * W) * g_ldc + c];
    g.w = gout[(ob + 2 * W + 1) * g_ldc + c];
    if (ocm) g *= ocm[(size_t)n * ocm_ld + c];
    if (icm) g *= *(const f32x4*)(icm + (size_t)n * icm_ld + 4 * c);
    f32x4* q = (f32x4*)(gin + p * gin_ldc + 4 * c);
    *q = acc ? *q + g : g;
  }
}
extern "C" int pmf_pixel_shuffle2_bwd(const float* gout, int32_t g_ldc, int32_t N, int32_t H, int32_t W, int32_t Cout,
                                      const float* out_cmul, int32_t out_cmul_ld, const float* in_cmul,
                                      int32_t in_cmul_ld, float* gin, int32_t gin_ldc, int32_t acc, pmf_stream_t s) {
  hipLaunchKernelGGL(pshuffle_bwd_k, dim3(ew_grid((int64_t)N * H * W * Cout)), dim3(EW_BLOCK), 0, (hipStream_t)s, gout,
                     g_ldc, N, H, W, Cout, out_cmul, out_cmul_ld, in_cmul, in_cmul_ld, gin, gin_ldc, acc);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ fusion gate: out = f * sigmoid(att) + pcd
__device__ __forceinline__ f32x4 sigmoid4(f32x4 a) {
  f32x4 r;
  r.x = 1.f / (1.f + expf(-a.x)); r.y = 1.f / (1.f + expf(-a.y));
  r.z = 1.f / (1.f + expf(-a.z)); r.w = 1.f / (1.f + expf(-a.w));
  return r;
}
__global__ void gate_k(pmf_view_t f, pmf_view_t att, const float* __restrict__ pcd, int pcd_ldc, float* __restrict__ out,
                       int out_ldc, int64_t npix, int Q) {
  const int64_t total = npix * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4;
    f32x4 fv = ldv(f, p, 0, c), sg = sigmoid4(ldv(att, p, 0, c));
    *(f32x4*)(out + p * out_ldc + c) = fv * sg + *(const f32x4*)(pcd + p * pcd_ldc + c);
  }
}
extern "C" int pmf_fusion_gate(const pmf_view_t* f, const pmf_view_t* att, const float* pcd, int32_t pcd_ldc,
                               float* out, int32_t out_ldc, int64_t npix, int32_t C, pmf_stream_t s) {
  if (C % 4 || f->cmul || att->cmul) return PMF_E_ARG;
  hipLaunchKernelGGL(gate_k, dim3(ew_grid(npix * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, *f, *att, pcd, pcd_ldc, out,
                     out_ldc, npix, C / 4);
  PMF_LAUNCH_CHECK();
  return 0;
}
__global__ void gate_bwd_k(const float* __restrict__ gout, int g_ldc, pmf_view_t f, pmf_view_t att, float* gf,
                           int gf_ldc, int gf_acc, float* gatt, int gatt_ldc, float* gpcd, int gpcd_ldc, int gpcd_acc,
                           int64_t npix, int Q) {
  const int64_t total = npix * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4;
    const f32x4 g = *(const f32x4*)(gout + p * g_ldc + c);
    const f32x4 fv = ldv(f, p, 0, c), sg = sigmoid4(ldv(att, p, 0, c));
    f32x4* q = (f32x4*)(gf + p * gf_ldc + c);
    *q = gf_acc ? *q + g * sg : g * sg;
    *(f32x4*)(gatt + p * gatt_ldc + c) = g * fv * sg * (1.f - sg);
    if (gpcd) {
      f32x4* r = (f32x4*)(gpcd + p * gpcd_ldc + c);
      *r = gpcd_acc ? *r + g : g;
    }
  }
}
extern "C" int pmf_fusion_gate_bwd(const float* gout, int32_t g_ldc, const pmf_view_t* f, const pmf_view_t* att,
                                   float* gf, int32_t gf_ldc, int32_t gf_acc, float* gatt, int32_t gatt_ldc,
                                   float* gpcd, int32_t gpcd_ldc, int32_t gpcd_acc, int64_t npix, int32_t C,
                                   pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  hipLaunchKernelGGL(gate_bwd_k, dim3(ew_grid(npix * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, gout, g_ldc, *f, *att,
                     gf, gf_ldc, gf_acc, gatt, gatt_ldc, gpcd, gpcd_ldc, gpcd_acc, npix, C / 4);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ channel softmax, NHWC logits <-> NCHW probabilities
#define SM_MAXC 32
// VEC: the NHWC side moves as 16-byte vectors (ldc a multiple of 4, base 16-byte aligned): a thread owns one pixel, so a
// scalar access per channel touches as many cache lines per instruction as a vector access that moves four channels
// IDENT: no softmax, the logits themselves go out (SalsaNext(softmax=False), salsanext.py:167,206-207)
template <bool VEC, bool IDENT = false>
__global__ void softmax_k(const float* __restrict__ lg, int ldc, int N, int HW, int C, float* __restrict__ prob) {
  const int64_t total = (int64_t)N * HW;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(p / HW);
    const int64_t hw = p - (int64_t)n * HW;
    float v[SM_MAXC];
    if (VEC) {
#pragma unroll
      for (int q = 0; q < SM_MAXC / 4; ++q)
        if (q * 4 < C) {
          const f32x4 t = *(const f32x4*)(lg + p * ldc + q * 4);
          v[q * 4] = t.x; v[q * 4 + 1] = t.y; v[q * 4 + 2] = t.z; v[q * 4 + 3] = t.w;
        }
    } else {
#pragma unroll
      for (int c = 0; c < SM_MAXC; ++c)
        if (c < C) v[c] = lg[p * ldc + c];
    }
    float inv = 1.f;
    if (!IDENT) {
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < SM_MAXC; ++c)
        if (c < C) m = fmaxf(m, v[c]);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < SM_MAXC; ++c)
        if (c < C) { v[c] = expf(v[c] - m); sum += v[c]; }
      inv = 1.f / sum;
    }
#pragma unroll
    for (int c = 0; c < SM_MAXC; ++c)
      if (c < C) prob[((int64_t)n * C + c) * HW + hw] = v[c] * inv;
  }
}
extern "C" int pmf_softmax_nhwc_to_nchw(const float* logits, int32_t ldc, int32_t N, int32_t HW, int32_t C,
                                        float* prob_nchw, pmf_stream_t s) {
  if (C > SM_MAXC || C < 1) return PMF_E_UNSUPPORTED;
  const bool vec = (ldc & 3) == 0 && ((uintptr_t)logits & 15) == 0 && ((C + 3) & ~3) <= ldc;
  if (vec) hipLaunchKernelGGL(softmax_k<true>, dim3(ew_grid((int64_t)N * HW)), dim3(EW_BLOCK), 0, (hipStream_t)s, logits, ldc,
                              N, HW, C, prob_nchw);
  else hipLaunchKernelGGL(softmax_k<false>, dim3(ew_grid((int64_t)N * HW)), dim3(EW_BLOCK), 0, (hipStream_t)s, logits, ldc, N,
                          HW, C, prob_nchw);
  PMF_LAUNCH_CHECK();
  return 0;
}
extern "C" int pmf_logits_nhwc_to_nchw(const float* logits, int32_t ldc, int32_t N, int32_t HW, int32_t C, float* out_nchw,
                                       pmf_stream_t s) {
  if (C > SM_MAXC || C < 1) return PMF_E_UNSUPPORTED;
  const bool vec = (ldc & 3) == 0 && ((uintptr_t)logits & 15) == 0 && ((C + 3) & ~3) <= ldc;
  if (vec) hipLaunchKernelGGL((softmax_k<true, true>), dim3(ew_grid((int64_t)N * HW)), dim3(EW_BLOCK), 0, (hipStream_t)s,
                              logits, ldc, N, HW, C, out_nchw);
  else hipLaunchKernelGGL((softmax_k<false, true>), dim3(ew_grid((int64_t)N * HW)), dim3(EW_BLOCK), 0, (hipStream_t)s, logits,
                          ldc, N, HW, C, out_nchw);
  PMF_LAUNCH_CHECK();
  return 0;
}
template <bool VEC, bool IDENT = false>
__global__ void softmax_bwd_k(const float* __restrict__ prob, const float* __restrict__ g, int N, int HW, int C,
                              float* __restrict__ dl, int ldc) {
  const int64_t total = (int64_t)N * HW;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(p / HW);
    const int64_t hw = p - (int64_t)n * HW;
    float pv[SM_MAXC], gv[SM_MAXC];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < SM_MAXC; ++c)
      if (c < C) {
        pv[c] = IDENT ? 1.f : prob[((int64_t)n * C + c) * HW + hw];
        gv[c] = g[((int64_t)n * C + c) * HW + hw];
        dot += IDENT ? 0.f : pv[c] * gv[c];
      }
    if (VEC) {
#pragma unroll
      for (int q = 0; q < SM_MAXC / 4; ++q)
        if (q * 4 < ldc) {
          f32x4 t;
          t.x = q * 4 + 0 < C ? pv[q * 4 + 0] * (gv[q * 4 + 0] - dot) : 0.f;
          t.y = q * 4 + 1 < C ? pv[q * 4 + 1] * (gv[q * 4 + 1] - dot) : 0.f;
          t.z = q * 4 + 2 < C ? pv[q * 4 + 2] * (gv[q * 4 + 2] - dot) : 0.f;
          t.w = q * 4 + 3 < C ? pv[q * 4 + 3] * (gv[q * 4 + 3] - dot) : 0.f;
          *(f32x4*)(dl + p * ldc + q * 4) = t;
        }
    } else {
#pragma unroll
      for (int c = 0; c < SM_MAXC; ++c)
        if (c < ldc) dl[p * ldc + c] = c < C ? pv[c] * (gv[c] - dot) : 0.f;
    }
  }
}
extern "C" int pmf_softmax_bwd_nchw_to_nhwc(const float* prob_nchw, const float* g_nchw, int32_t N, int32_t HW,
                                            int32_t C, float* dlogits, int32_t ldc, pmf_stream_t s) {
  if (C > SM_MAXC || ldc > SM_MAXC) return PMF_E_UNSUPPORTED;
  const bool vec = (ldc & 3) == 0 && ((uintptr_t)dlogits & 15) == 0;
  if (vec) hipLaunchKernelGGL(softmax_bwd_k<true>, dim3(ew_grid((int64_t)N * HW)), dim3(EW_BLOCK), 0, (hipStream_t)s, prob_nchw,
                              g_nchw, N, HW, C, dlogits, ldc);
  else hipLaunchKernelGGL(softmax_bwd_k<false>, dim3(ew_grid((int64_t)N * HW)), dim3(EW_BLOCK), 0, (hipStream_t)s, prob_nchw,
                          g_nchw, N, HW, C, dlogits, ldc);
  PMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pmf_logits_bwd_nchw_to_nhwc(const float* g_nchw, int32_t N, int32_t HW, int32_t C, float* dlogits, int32_t ldc,
                                           pmf_stream_t s) {
  if (C > SM_MAXC || ldc > SM_MAXC) return PMF_E_UNSUPPORTED;
  const bool vec = (ldc & 3) == 0 && ((uintptr_t)dlogits & 15) == 0;
  if (vec) hipLaunchKernelGGL((softmax_bwd_k<true, true>), dim3(ew_grid((int64_t)N * HW)), dim3(EW_BLOCK), 0, (hipStream_t)s,
                              nullptr, g_nchw, N, HW, C, dlogits, ldc);
  else hipLaunchKernelGGL((softmax_bwd_k<false, true>), dim3(ew_grid((int64_t)N * HW)), dim3(EW_BLOCK), 0, (hipStream_t)s,
                          nullptr, g_nchw, N, HW, C, dlogits, ldc);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ per-sample vector broadcast over the image
// out[n][p][c] = src[n][c]: the image-level branch of ASPP (`F.interpolate(1x1 -> size)` is a pure broadcast, pmf_net.py:124-125)
// as a materialised operand, so that the 1x1 projection over the five concatenated branches runs the fast 1x1 paths (direct
// split-bf16 forward, ONE merged input-gradient launch, LDS-free weight gradient) instead of the generic loop the
// broadcast-operand flag forces (round 6: aspp.out forward 37 -> 15 us, five input-gradient launches -> one)
__global__ void bcast_rows_k(const float* __restrict__ src, int sldc, int64_t HW, int Q, int64_t total, float* __restrict__ out,
                             int out_ldc) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4;
    const int64_t n = p / HW;
    *(f32x4*)(out + p * out_ldc + c) = *(const f32x4*)(src + n * sldc + c);
  }
}
extern "C" int pmf_broadcast_rows(const float* src, int32_t src_ldc, int32_t N, int64_t HW, int32_t C, float* out,
                                  int32_t out_ldc, pmf_stream_t s) {
  if (C % 4 || src_ldc % 4 || out_ldc % 4 || N < 1 || HW < 1) return PMF_E_ARG;
  const int64_t total = (int64_t)N * HW * (C / 4);
  hipLaunchKernelGGL(bcast_rows_k, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)s, src, src_ldc, HW, C / 4, total, out,
                     out_ldc);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ boundary layout change / fill
__global__ void nchw2nhwc_k(const float* __restrict__ x, int64_t sn, int64_t sc, int N, int C, int HW,
                            float* __restrict__ out, int ldc) {
  const int64_t total = (int64_t)N * HW;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(p / HW);
    const int64_t hw = p - (int64_t)n * HW;
    if ((ldc & 3) == 0 && ((uintptr_t)out & 15) == 0) {     // (one 16-byte store per channel quad instead of four strided ones)
      for (int q = 0; q < ldc; q += 4) {
        f32x4 t;
        t.x = q + 0 < C ? x[n * sn + (q + 0) * sc + hw] : 0.f;
        t.y = q + 1 < C ? x[n * sn + (q + 1) * sc + hw] : 0.f;
        t.z = q + 2 < C ? x[n * sn + (q + 2) * sc + hw] : 0.f;
        t.w = q + 3 < C ? x[n * sn + (q + 3) * sc + hw] : 0.f;
        *(f32x4*)(out + p * ldc + q) = t;
      }
    } else {
      for (int c = 0; c < ldc; ++c) out[p * ldc + c] = c < C ? x[n * sn + c * sc + hw] : 0.f;
    }
  }
}
extern "C" int pmf_nchw_to_nhwc(const float* x, int64_t stride_n, int64_t stride_c, int32_t N, int32_t C, int32_t HW,
                                float* out, int32_t out_ldc, pmf_stream_t s) {
  hipLaunchKernelGGL(nchw2nhwc_k, dim3(ew_grid((int64_t)N * HW)), dim3(EW_BLOCK), 0, (hipStream_t)s, x, stride_n, stride_c,
                     N, C, HW, out, out_ldc);
  PMF_LAUNCH_CHECK();
  return 0;
}
__global__ void fill_k(float* __restrict__ p, float v, int64_t n) {
  const int64_t n4 = n >> 2;
  f32x4 vv = {v, v, v, v};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    ((f32x4*)p)[i] = vv;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = v;
}
extern "C" int pmf_fill(float* p, float v, int64_t n, pmf_stream_t s) {
  if (((uintptr_t)p) & 15) return PMF_E_ARG;
  hipLaunchKernelGGL(fill_k, dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, (hipStream_t)s, p, v, n);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ---- per-pixel validity masks: EPMF SparseVariantConv / ResContextBlock (pc_processor/models/epmf_net.py:30-50, 66-80)
// mask = (sum_c |x| != 0);  dilated mask = max-pool of the zero-padded mask with the conv's kernel / stride / dilation;
// x * mask on the way in, (conv + bias) * dilated mask on the way out (the latter lives in the conv epilogue).
__global__ void pmask_from_k(pmf_view_t v, int64_t npix, int HW, int Q, float* __restrict__ m) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(p / HW);
    float s = 0.f;
    for (int q = 0; q < Q; ++q) {
      const f32x4 x = ldv(v, p, n, q * 4);
      s += fabsf(x.x) + fabsf(x.y) + fabsf(x.z) + fabsf(x.w);
    }
    m[p] = s != 0.f ? 1.f : 0.f;
  }
}
extern "C" int pmf_pmask_from(const pmf_view_t* in, int64_t npix, int32_t HW, int32_t C, float* mask, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  hipLaunchKernelGGL(pmask_from_k, dim3(ew_grid(npix)), dim3(EW_BLOCK), 0, (hipStream_t)s, *in, npix, HW, C / 4, mask);
  PMF_LAUNCH_CHECK();
  return 0;
}

__global__ void pmask_pool_k(const float* __restrict__ m, int N, int H, int W, int OH, int OW, int kh, int kw, int dil,
                             int pad, int stride, float* __restrict__ out) {
  const int64_t total = (int64_t)N * OH * OW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), n = (int)(i / ((int64_t)OW * OH));
    float best = 0.f;    // F.pad pads with zeros and the mask is non-negative
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) {
        const int iy = oy * stride + ky * dil - pad, ix = ox * stride + kx * dil - pad;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) best = fmaxf(best, m[((int64_t)n * H + iy) * W + ix]);
      }
    out[i] = best;
  }
}
extern "C" int pmf_pmask_pool(const float* mask, int32_t N, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t dil,
                              int32_t pad, int32_t stride, float* out, int32_t OH, int32_t OW, pmf_stream_t s) {
  hipLaunchKernelGGL(pmask_pool_k, dim3(ew_grid((int64_t)N * OH * OW)), dim3(EW_BLOCK), 0, (hipStream_t)s, mask, N, H, W, OH,
                     OW, kh, kw, dil, pad, stride, out);
  PMF_LAUNCH_CHECK();
  return 0;
}

__global__ void pmask_mul_k(pmf_view_t v, const float* __restrict__ m, int64_t npix, int HW, int Q, float* __restrict__ out,
                            int out_ldc) {
  const int64_t total = npix * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4, n = (int)(p / HW);
    *(f32x4*)(out + p * out_ldc + c) = ldv(v, p, n, c) * m[p];
  }
}
extern "C" int pmf_pmask_mul(const pmf_view_t* in, const float* mask, int64_t npix, int32_t HW, int32_t C, float* out,
                             int32_t out_ldc, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  hipLaunchKernelGGL(pmask_mul_k, dim3(ew_grid(npix * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, *in, mask, npix, HW,
                     C / 4, out, out_ldc);
  PMF_LAUNCH_CHECK();
  return 0;
}

// gx (+)= gy * mask   (gx == gy, acc == 0: in place)
__global__ void pmask_mul_bwd_k(const float* gy, int gy_ldc, const float* __restrict__ m, int64_t npix, int Q, float* gx,
                                int gx_ldc, int acc) {
  const int64_t total = npix * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Q;
    const int c = (int)(i - p * Q) * 4;
    f32x4 g = *(const f32x4*)(gy + p * gy_ldc + c) * m[p];
    float* o = gx + p * gx_ldc + c;
    if (acc) g += *(const f32x4*)o;
    *(f32x4*)o = g;
  }
}
extern "C" int pmf_pmask_mul_bwd(const float* gy, int32_t gy_ldc, const float* mask, int64_t npix, int32_t C, float* gx,
                                 int32_t gx_ldc, int32_t acc, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  hipLaunchKernelGGL(pmask_mul_bwd_k, dim3(ew_grid(npix * (C / 4))), dim3(EW_BLOCK), 0, (hipStream_t)s, gy, gy_ldc, mask,
                     npix, C / 4, gx, gx_ldc, acc);
  PMF_LAUNCH_CHECK();
  return 0;
}

// out[i] = a[i] + (b ? b[i] : 0)    (the two biases of SparseVariantConv; copy of a bias gradient)
__global__ void vec_add_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + (b ? b[i] : 0.f);
}
extern "C" int pmf_vec_add(const float* a, const float* b, float* out, int32_t n, pmf_stream_t s) {
  hipLaunchKernelGGL(vec_add_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, a, b, out, n);
  PMF_LAUNCH_CHECK();
  return 0;
}
