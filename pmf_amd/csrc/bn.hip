// BatchNorm2d support kernels (gfx950): statistics finalisation, eval affine, and the two-pass backward.
// The normalisation itself never runs as a kernel: producers emit per-channel sum / sum-of-squares from their
// epilogue and consumers apply scale/shift on load (see conv_fwd.hip).
#include "common.h"
#include <stdlib.h>

// grid = C workgroups: workgroup c folds the nrows partial (sum, sumsq) rows of channel c in a fixed order
__global__ void bn_finalize_k(const double* __restrict__ stats, int nrows, float count, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float* running_mean, float* running_var, float momentum,
                              float eps, float* scale, float* shift, float* save_mean, float* save_invstd, int C) {
  __shared__ double sh[2][256];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  // four rows per trip: eight independent loads in flight per thread (a load -> add loop is one memory round trip per
  // row, and this kernel is nothing but latency)
  for (int r = threadIdx.x; r < nrows; r += 1024) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + 256 * u;
      a[u] = rr < nrows ? stats[(size_t)rr * 2 * C + c] : 0.0;
      b[u] = rr < nrows ? stats[(size_t)rr * 2 * C + C + c] : 0.0;
    }
    s1 += (a[0] + a[1]) + (a[2] + a[3]);
    s2 += (b[0] + b[1]) + (b[2] + b[3]);
  }
  // wave shuffles, then the four wave sums through LDS: one barrier instead of eight (this kernel is nothing but latency)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x) return;
  const double t1 = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]), t2 = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  const double mean = t1 / (double)count;
  double var = t2 / (double)count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (save_mean) { save_mean[c] = (float)mean; save_invstd[c] = invstd; }
  if (running_mean) {
    const double unb = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}
extern "C" int pmf_bn_finalize(const double* stats, int32_t nrows, float count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* scale,
                               float* shift, float* save_mean, float* save_invstd, int32_t C, pmf_stream_t s) {
  hipLaunchKernelGGL(bn_finalize_k, dim3(C), dim3(256), 0, (hipStream_t)s, stats, nrows, count, gamma, beta,
                     running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd, C);
  PMF_LAUNCH_CHECK();
  return 0;
}

__global__ void bn_eval_k(const float* __restrict__ gamma, const float* __restrict__ beta,
                          const float* __restrict__ rm, const float* __restrict__ rv, float eps, float* scale,
                          float* shift, float* save_mean, float* save_invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
  if (save_mean) { save_mean[c] = rm[c]; save_invstd[c] = invstd; }
}
extern "C" int pmf_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, float* save_mean,
                                  float* save_invstd, int32_t C, pmf_stream_t s) {
  hipLaunchKernelGGL(bn_eval_k, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)s, gamma, beta, running_mean, running_var,
                     eps, scale, shift, save_mean, save_invstd, C);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ---- backward -------------------------------------------------------------------------------------------
// pass 1 (bn_bwd_reduce): per-workgroup partial rows of  sum gy  and  sum gy*(a - mean)   (float64, no atomics)
// fold  (bn_bwd_fold):    rows -> dgamma, dbeta (+= into the gradient buffer) and the per-channel coefficients
//                         coef[0][c] = gamma*invstd, coef[1][c] = invstd^2 * mean(gy*(a-mean)), coef[2][c] = mean(gy)
// pass 2 (bn_bwd_apply):  dz = coef0 * ((gy - coef2) - (a - mean) * coef1) * act'(a); per-workgroup partial rows of
//                         sum dz (the conv-bias gradient) for the weight-gradient kernel to fold
int g_pmf_col_cap = PMF_COL_ROWS, g_pmf_col_unroll = 4;
// tools/bench_elem.py: sweep the launch shape of the column kernels (cap > PMF_COL_ROWS only with caller-sized row buffers)
extern "C" int pmf_debug_col(int32_t cap, int32_t unroll) {
  if (cap > 0) g_pmf_col_cap = cap;
  if (unroll == 4 || unroll == 8) g_pmf_col_unroll = unroll;
  return g_pmf_col_cap;
}
__device__ __forceinline__ int qgmax(int Q) { return Q < 256 ? Q : 256; }
struct ColL { dim3 grid, block; };
static ColL col_l(int64_t npix, int Q) {
  int Qg = Q < 256 ? Q : 256, rows = 256 / Qg;
  int64_t gx = cdiv64(npix, (int64_t)rows * 4);
  gx = gx > g_pmf_col_cap ? g_pmf_col_cap : (gx < 1 ? 1 : gx);
  ColL L;
  L.grid = dim3((unsigned)gx, (unsigned)cdiv(Q, 256), 1);
  L.block = dim3(rows * Qg);
  return L;
}
extern "C" int pmf_col_rows(int64_t npix, int32_t C) { return (int)col_l(npix, C / 4).grid.x; }

template <int U>
__global__ void bn_bwd_reduce_k(const float* __restrict__ gy, int gy_ldc, const float* __restrict__ a, int a_ldc,
                                int64_t npix, int Q, int C, const float* __restrict__ save_mean, double* part) {
  __shared__ double sh[2][256][4];
  const int Qm = qgmax(Q), Qg = min(Q - (int)blockIdx.y * 256, 256), rows = 256 / Qm;
  const int row = threadIdx.x / Qm, cql = threadIdx.x - row * Qm;
  const bool active = cql < Qg;
  const int c = ((int)blockIdx.y * 256 + cql) * 4;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};   // float64 accumulation (HBM-bound kernel: free)
  if (active) {
    const f32x4 mu = *(const f32x4*)(save_mean + c);   // centre first: sum g*(a - mean) has no cancellation
    // four pixels per trip: 8 independent 16-byte loads in flight per thread (one pixel per trip left the kernel at
    // 1.8 TB/s: 512 workgroups x 2 loads do not cover the HBM latency)
    const int64_t step = (int64_t)gridDim.x * rows;
    for (int64_t p = (int64_t)blockIdx.x * rows + row; p < npix; p += U * step) {
      f32x4 g[U], x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t pp = p + u * step;
        if (pp < npix) { g[u] = *(const f32x4*)(gy + pp * gy_ldc + c); x[u] = *(const f32x4*)(a + pp * a_ldc + c); }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (p + u * step < npix) {
          const f32x4 xc = x[u] - mu;
#pragma unroll
          for (int k = 0; k < 4; ++k) { s1[k] += (double)g[u][k]; s2[k] += (double)g[u][k] * (double)xc[k]; }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { sh[0][row * Qm + cql][k] = s1[k]; sh[1][row * Qm + cql][k] = s2[k]; }
  __syncthreads();
  if (row == 0 && active) {
    for (int r = 1; r < rows; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) { s1[k] += sh[0][r * Qm + cql][k]; s2[k] += sh[1][r * Qm + cql][k]; }
    double* prow = part + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) { prow[c + k] = s1[k]; prow[C + c + k] = s2[k]; }
  }
}

__global__ void bn_bwd_fold_k(const double* __restrict__ part, int nrows, int C, float invM, int train,
                              const float* __restrict__ gamma, const float* __restrict__ save_invstd, float* coef,
                              float* dgamma, float* dbeta) {
  __shared__ double sh[2][256];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int r = threadIdx.x; r < nrows; r += 1024) {       // 256 threads, four rows per trip (see bn_finalize_k)
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + 256 * u;
      a[u] = rr < nrows ? part[(size_t)rr * 2 * C + c] : 0.0;
      b[u] = rr < nrows ? part[(size_t)rr * 2 * C + C + c] : 0.0;
    }
    s1 += (a[0] + a[1]) + (a[2] + a[3]);
    s2 += (b[0] + b[1]) + (b[2] + b[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x) return;
  const float sg = (float)((sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3])), sgc = (float)((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]));
  const float r = save_invstd[c], g = gamma[c];
  const float dgam = r * sgc;
  dgamma[c] += dgam;
  dbeta[c] += sg;
  coef[c] = g * r;
  coef[C + c] = train ? r * dgam * invM : 0.f;
  coef[2 * C + c] = train ? sg * invM : 0.f;
}

extern "C" int pmf_bn_bwd_reduce(const float* gy, int32_t gy_ldc, const float* a, int32_t a_ldc, int64_t npix,
                                 int32_t C, const float* save_mean, const float* gamma, const float* save_invstd,
                                 int32_t train, double* part, float* coef, float* dgamma, float* dbeta,
                                 pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  ColL L = col_l(npix, C / 4);
  if (g_pmf_col_unroll == 8)
    hipLaunchKernelGGL(bn_bwd_reduce_k<8>, L.grid, L.block, 0, (hipStream_t)s, gy, gy_ldc, a, a_ldc, npix, C / 4, C,
                       save_mean, part);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_k<4>, L.grid, L.block, 0, (hipStream_t)s, gy, gy_ldc, a, a_ldc, npix, C / 4, C,
                       save_mean, part);
  hipLaunchKernelGGL(bn_bwd_fold_k, dim3(C), dim3(256), 0, (hipStream_t)s, (const double*)part, (int)L.grid.x, C,
                     1.f / (float)npix, train, gamma, save_invstd, coef, dgamma, dbeta);
  PMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pmf_bn_bwd_fold(const double* part, int32_t nrows, int32_t C, int64_t npix, int32_t train,
                               const float* gamma, const float* save_invstd, float* coef, float* dgamma, float* dbeta,
                               pmf_stream_t s) {
  if (nrows < 1) return PMF_E_ARG;
  hipLaunchKernelGGL(bn_bwd_fold_k, dim3(C), dim3(256), 0, (hipStream_t)s, part, (int)nrows, C, 1.f / (float)npix, train,
                     gamma, save_invstd, coef, dgamma, dbeta);
  PMF_LAUNCH_CHECK();
  return 0;
}

template <int U>
__global__ void bn_bwd_apply_k(const float* __restrict__ gy, int gy_ldc, const float* __restrict__ a, int a_ldc,
                               int64_t npix, int Q, int C, const float* __restrict__ coef,
                               const float* __restrict__ save_mean, int act, float* __restrict__ dz, int dz_ldc,
                               float* dbias_rows, int dbias_ld) {
  __shared__ f32x4 sh[256];
  const int Qm = qgmax(Q), Qg = min(Q - (int)blockIdx.y * 256, 256), rows = 256 / Qm;
  const int row = threadIdx.x / Qm, cql = threadIdx.x - row * Qm;
  const bool active = cql < Qg;
  const int c = ((int)blockIdx.y * 256 + cql) * 4;
  f32x4 part = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    // differences of nearby float32 values first (exact), scaling last -- PyTorch's conditioning
    const f32x4 A = *(const f32x4*)(coef + c), K = *(const f32x4*)(coef + C + c), MG = *(const f32x4*)(coef + 2 * C + c);
    const f32x4 MU = *(const f32x4*)(save_mean + c);
    const float sl = act == PMF_ACT_LRELU ? 0.01f : (act == PMF_ACT_RELU ? 0.f : 1.f);
    const int64_t step = (int64_t)gridDim.x * rows;
    for (int64_t p = (int64_t)blockIdx.x * rows + row; p < npix; p += U * step) {     // four pixels per trip (see above)
      f32x4 g[U], x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t pp = p + u * step;
        if (pp < npix) { g[u] = *(const f32x4*)(gy + pp * gy_ldc + c); x[u] = *(const f32x4*)(a + pp * a_ldc + c); }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t pp = p + u * step;
        if (pp < npix) {
          f32x4 d = A * ((g[u] - MG) - (x[u] - MU) * K);
          if (act != PMF_ACT_NONE) {
            d.x *= x[u].x > 0.f ? 1.f : sl; d.y *= x[u].y > 0.f ? 1.f : sl;
            d.z *= x[u].z > 0.f ? 1.f : sl; d.w *= x[u].w > 0.f ? 1.f : sl;
          }
          *(f32x4*)(dz + pp * dz_ldc + c) = d;
          part += d;
        }
      }
    }
  }
  if (dbias_rows) {
    sh[row * Qm + cql] = part;
    __syncthreads();
    if (row == 0 && active) {
      for (int r = 1; r < rows; ++r) part += sh[r * Qm + cql];
      *(f32x4*)(dbias_rows + (size_t)blockIdx.x * dbias_ld + c) = part;
    }
  }
}
extern "C" int pmf_bn_bwd_apply(const float* gy, int32_t gy_ldc, const float* a, int32_t a_ldc, int64_t npix,
                                int32_t C, const float* coef, const float* save_mean, int32_t act, float* dz,
                                int32_t dz_ldc, float* dbias_rows, int32_t dbias_ld, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  ColL L = col_l(npix, C / 4);
  if (g_pmf_col_unroll == 8)
    hipLaunchKernelGGL(bn_bwd_apply_k<8>, L.grid, L.block, 0, (hipStream_t)s, gy, gy_ldc, a, a_ldc, npix, C / 4, C, coef,
                       save_mean, act, dz, dz_ldc, dbias_rows, dbias_ld);
  else
    hipLaunchKernelGGL(bn_bwd_apply_k<4>, L.grid, L.block, 0, (hipStream_t)s, gy, gy_ldc, a, a_ldc, npix, C / 4, C, coef,
                       save_mean, act, dz, dz_ldc, dbias_rows, dbias_ld);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ---- small maps: the whole BatchNorm backward of a layer in ONE launch ------------------------------------------------
// At <= 2048 pixels (the 4x128 stage and below: a fifth of the network's BatchNorm layers) the three-launch form (reduce ->
// fold -> apply, or fold -> apply behind an input-gradient epilogue that carried the sums) is nothing but launch latency:
// 13-14 us for 1-2 MB that sit in L2.  Here a workgroup of 1024 threads owns EIGHT channels (two threads per pixel, one
// float4 each) of the whole map: gy and a are read ONCE into registers (<= 4 pixels per thread), the two float64 column
// sums are folded inside the workgroup (wave shuffles over the lanes of equal parity, then 16 rows in LDS, fixed order:
// deterministic), the coefficients are computed in place (same formulas as bn_bwd_fold_k), dz is written from the
// registers and its column sum -- the conv-bias gradient -- leaves as ONE exact row.  grid = C / 8.
// (Measured, tools/bench_elem.py: with FOUR channels per workgroup every workgroup pulls a 128-byte line for 16 bytes of
// it -- 25.7 us at 4096 x 256 against 14.8 us for the three launches; 4096-pixel maps would need the pixels split over
// workgroups, i.e. a second launch: they stay on the three-launch form.)
template <int PPT>
__global__ __launch_bounds__(1024) void bn_bwd_small_k(const float* __restrict__ gy, int gy_ldc, const float* __restrict__ a,
                                                       int a_ldc, int npix, int C, const float* __restrict__ save_mean,
                                                       const float* __restrict__ gamma, const float* __restrict__ save_invstd,
                                                       int train, int act, float invM, float* __restrict__ dz, int dz_ldc,
                                                       float* __restrict__ dbias_row, float* dgamma, float* dbeta) {
  __shared__ double shd[16][2][8];
  __shared__ f32x4 shf[16][2];
  __shared__ float co[3][8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = tid & 1;
  const int c = blockIdx.x * 8 + half * 4;
  const bool cok = c < C;                       // C % 8 == 4: the last workgroup's upper half has no channels
  const f32x4 MU = cok ? *(const f32x4*)(save_mean + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 g[PPT], x[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int p = (tid >> 1) + 512 * u;
    g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    x[u] = MU;
    if (cok && p < npix) { g[u] = *(const f32x4*)(gy + (size_t)p * gy_ldc + c); x[u] = *(const f32x4*)(a + (size_t)p * a_ldc + c); }
  }
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const f32x4 xc = x[u] - MU;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] += (double)g[u][k]; s[4 + k] += (double)g[u][k] * (double)xc[k]; }
  }
#pragma unroll
  for (int o = 32; o > 1; o >>= 1)              // (offset 1 would mix the two channel halves)
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] += __shfl_xor(s[k], o);
  if (lane < 2)
#pragma unroll
    for (int k = 0; k < 8; ++k) shd[wave][lane][k] = s[k];
  __syncthreads();
  if (tid < 8 && blockIdx.x * 8 + tid < C) {
    const int h = tid >> 2, k = tid & 3, cc = blockIdx.x * 8 + tid;
    double sg = 0.0, sgc = 0.0;
    for (int w = 0; w < 16; ++w) { sg += shd[w][h][k]; sgc += shd[w][h][4 + k]; }
    const float fg = (float)sg, fgc = (float)sgc, r = save_invstd[cc], gm = gamma[cc];
    const float dgam = r * fgc;
    dgamma[cc] += dgam;
    dbeta[cc] += fg;
    co[0][tid] = gm * r;
    co[1][tid] = train ? r * dgam * invM : 0.f;
    co[2][tid] = train ? fg * invM : 0.f;
  }
  __syncthreads();
  const f32x4 A = *(const f32x4*)(co[0] + half * 4), K = *(const f32x4*)(co[1] + half * 4), MG = *(const f32x4*)(co[2] + half * 4);
  const float sl = act == PMF_ACT_LRELU ? 0.01f : (act == PMF_ACT_RELU ? 0.f : 1.f);
  f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int p = (tid >> 1) + 512 * u;
    if (cok && p < npix) {
      f32x4 d = A * ((g[u] - MG) - (x[u] - MU) * K);      // differences of nearby values first (bn_bwd_apply_k)
      if (act != PMF_ACT_NONE) {
        d.x *= x[u].x > 0.f ? 1.f : sl; d.y *= x[u].y > 0.f ? 1.f : sl;
        d.z *= x[u].z > 0.f ? 1.f : sl; d.w *= x[u].w > 0.f ? 1.f : sl;
      }
      *(f32x4*)(dz + (size_t)p * dz_ldc + c) = d;
      part += d;
    }
  }
  if (dbias_row) {
#pragma unroll
    for (int o = 32; o > 1; o >>= 1) {
      part.x += __shfl_xor(part.x, o); part.y += __shfl_xor(part.y, o);
      part.z += __shfl_xor(part.z, o); part.w += __shfl_xor(part.w, o);
    }
    if (lane < 2) shf[wave][lane] = part;
    __syncthreads();
    if (tid < 2 && blockIdx.x * 8 + tid * 4 < C) {
      f32x4 t = shf[0][tid];
      for (int w = 1; w < 16; ++w) t += shf[w][tid];
      *(f32x4*)(dbias_row + blockIdx.x * 8 + tid * 4) = t;
    }
  }
}

extern "C" int pmf_bn_bwd_small_ok(int64_t npix, int32_t C) { return npix > 0 && npix <= 2048 && C % 4 == 0; }

extern "C" int pmf_bn_bwd_small(const float* gy, int32_t gy_ldc, const float* a, int32_t a_ldc, int64_t npix, int32_t C,
                                const float* save_mean, const float* gamma, const float* save_invstd, int32_t train,
                                int32_t act, float* dz, int32_t dz_ldc, float* dbias_row, float* dgamma, float* dbeta,
                                pmf_stream_t s) {
  if (!pmf_bn_bwd_small_ok(npix, C)) return PMF_E_ARG;
  const dim3 grid(cdiv(C, 8)), block(1024);
  const float invM = 1.f / (float)npix;
  hipStream_t st = (hipStream_t)s;
#define PMF_BN_SMALL(P)                                                                                                 \
  hipLaunchKernelGGL(bn_bwd_small_k<P>, grid, block, 0, st, gy, gy_ldc, a, a_ldc, (int)npix, C, save_mean, gamma,       \
                     save_invstd, train, act, invM, dz, dz_ldc, dbias_row, dgamma, dbeta)
  if (npix <= 512) PMF_BN_SMALL(1);
  else if (npix <= 1024) PMF_BN_SMALL(2);
  else PMF_BN_SMALL(4);
#undef PMF_BN_SMALL
  PMF_LAUNCH_CHECK();
  return 0;
}
