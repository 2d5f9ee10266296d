// BatchNorm2d support kernels (gfx950): statistics finalisation, eval affine, and the two-pass backward.
// The normalisation itself never runs as a kernel: producers emit per-channel sum / sum-of-squares from their
// epilogue and consumers apply scale/shift on load (see conv_fwd.hip).
#include "common.h"

__global__ void bn_finalize_k(const double* __restrict__ stats, float count, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float* running_mean, float* running_var, float momentum,
                              float eps, float* scale, float* shift, float* save_mean, float* save_invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = stats[c] / (double)count;
  double var = stats[C + c] / (double)count - mean * mean;
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
extern "C" int pmf_bn_finalize(const double* stats, float count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* scale,
                               float* shift, float* save_mean, float* save_invstd, int32_t C, pmf_stream_t s) {
  hipLaunchKernelGGL(bn_finalize_k, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)s, stats, count, gamma, beta,
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
__device__ __forceinline__ int qgmax(int Q) { return Q < 256 ? Q : 256; }
struct ColL { dim3 grid, block; };
static ColL col_l(int64_t npix, int Q) {
  int Qg = Q < 256 ? Q : 256, rows = 256 / Qg;
  int64_t gx = cdiv64(npix, (int64_t)rows * 8);
  gx = gx > 1024 ? 1024 : (gx < 1 ? 1 : gx);
  ColL L;
  L.grid = dim3((unsigned)gx, (unsigned)cdiv(Q, 256), 1);
  L.block = dim3(rows * Qg);
  return L;
}

__global__ void bn_bwd_reduce_k(const float* __restrict__ gy, int gy_ldc, const float* __restrict__ a, int a_ldc,
                                int64_t npix, int Q, int C, const float* __restrict__ save_mean, double* red) {
  __shared__ double sh[2][256][4];
  const int Qm = qgmax(Q), Qg = min(Q - (int)blockIdx.y * 256, 256), rows = 256 / Qm;
  const int row = threadIdx.x / Qm, cql = threadIdx.x - row * Qm;
  const bool active = cql < Qg;
  const int c = ((int)blockIdx.y * 256 + cql) * 4;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};   // float64 accumulation (HBM-bound kernel: free)
  if (active) {
    const f32x4 mu = *(const f32x4*)(save_mean + c);   // centre first: sum g*(a - mean) has no cancellation
    for (int64_t p = (int64_t)blockIdx.x * rows + row; p < npix; p += (int64_t)gridDim.x * rows) {
      const f32x4 g = *(const f32x4*)(gy + p * gy_ldc + c);
      const f32x4 x = *(const f32x4*)(a + p * a_ldc + c) - mu;
#pragma unroll
      for (int k = 0; k < 4; ++k) { s1[k] += (double)g[k]; s2[k] += (double)g[k] * (double)x[k]; }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { sh[0][row * Qm + cql][k] = s1[k]; sh[1][row * Qm + cql][k] = s2[k]; }
  __syncthreads();
  if (row == 0 && active) {
    for (int r = 1; r < rows; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) { s1[k] += sh[0][r * Qm + cql][k]; s2[k] += sh[1][r * Qm + cql][k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { atomicAdd(red + c + k, s1[k]); atomicAdd(red + C + c + k, s2[k]); }
  }
}
extern "C" int pmf_bn_bwd_reduce(const float* gy, int32_t gy_ldc, const float* a, int32_t a_ldc, int64_t npix,
                                 int32_t C, const float* save_mean, double* red, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  ColL L = col_l(npix, C / 4);
  hipLaunchKernelGGL(bn_bwd_reduce_k, L.grid, L.block, 0, (hipStream_t)s, gy, gy_ldc, a, a_ldc, npix, C / 4, C, save_mean, red);
  PMF_LAUNCH_CHECK();
  return 0;
}

__global__ void bn_bwd_apply_k(const float* __restrict__ gy, int gy_ldc, const float* __restrict__ a, int a_ldc,
                               int64_t npix, int Q, int C, const double* __restrict__ red,
                               const float* __restrict__ gamma, const float* __restrict__ save_mean,
                               const float* __restrict__ save_invstd, int act, int train, float* __restrict__ dz,
                               int dz_ldc, float* dgamma, float* dbeta, float* dbias) {
  __shared__ f32x4 sh[256];
  const int Qm = qgmax(Q), Qg = min(Q - (int)blockIdx.y * 256, 256), rows = 256 / Qm;
  const int row = threadIdx.x / Qm, cql = threadIdx.x - row * Qm;
  const bool active = cql < Qg;
  const int c = ((int)blockIdx.y * 256 + cql) * 4;
  f32x4 part = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    // dz = A * ((gy - mean(gy)) - (a - mean) * K) * act'(a), K = invstd^2 * mean(gy*(a-mean)): differences of
    // nearby float32 values first (exact), scaling last -- the same conditioning as PyTorch's CPU kernel
    f32x4 A, B, Cc, MU;   // B = K, Cc = mean(gy)
    const float invM = 1.f / (float)npix;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = gamma[c + k], r = save_invstd[c + k], mu = save_mean[c + k];
      const float sg = (float)red[c + k], sgc = (float)red[C + c + k];
      const float dgam = r * sgc;
      MU[k] = mu;
      if (train) {
        A[k] = g * r;
        B[k] = r * dgam * invM;
        Cc[k] = sg * invM;
      } else {
        A[k] = g * r; B[k] = 0.f; Cc[k] = 0.f;
      }
      if (blockIdx.x == 0 && row == 0) { dgamma[c + k] += dgam; dbeta[c + k] += sg; }
    }
    const float sl = act == PMF_ACT_LRELU ? 0.01f : (act == PMF_ACT_RELU ? 0.f : 1.f);
    for (int64_t p = (int64_t)blockIdx.x * rows + row; p < npix; p += (int64_t)gridDim.x * rows) {
      const f32x4 g = *(const f32x4*)(gy + p * gy_ldc + c);
      const f32x4 x = *(const f32x4*)(a + p * a_ldc + c);
      f32x4 d = A * ((g - Cc) - (x - MU) * B);
      if (act != PMF_ACT_NONE) {
        d.x *= x.x > 0.f ? 1.f : sl; d.y *= x.y > 0.f ? 1.f : sl; d.z *= x.z > 0.f ? 1.f : sl; d.w *= x.w > 0.f ? 1.f : sl;
      }
      *(f32x4*)(dz + p * dz_ldc + c) = d;
      part += d;
    }
  }
  if (dbias) {
    sh[row * Qm + cql] = part;
    __syncthreads();
    if (row == 0 && active) {
      for (int r = 1; r < rows; ++r) part += sh[r * Qm + cql];
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(dbias + c + k, part[k]);
    }
  }
}
extern "C" int pmf_bn_bwd_apply(const float* gy, int32_t gy_ldc, const float* a, int32_t a_ldc, int64_t npix,
                                int32_t C, const double* red, const float* gamma, const float* save_mean,
                                const float* save_invstd, int32_t act, int32_t train, float* dz, int32_t dz_ldc,
                                float* dgamma, float* dbeta, float* dbias, pmf_stream_t s) {
  if (C % 4) return PMF_E_ARG;
  ColL L = col_l(npix, C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_k, L.grid, L.block, 0, (hipStream_t)s, gy, gy_ldc, a, a_ldc, npix, C / 4, C, red, gamma,
                     save_mean, save_invstd, act, train, dz, dz_ldc, dgamma, dbeta, dbias);
  PMF_LAUNCH_CHECK();
  return 0;
}
