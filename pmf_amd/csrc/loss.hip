// Loss-side kernels (gfx950).  Lovasz-softmax (pc_processor/loss/lovasz_softmax.py:56-68, lovasz_grad): for every
// class row of the DESCENDING-error-sorted foreground indicator fg[c][i] produce
//     jaccard_i = 1 - (G - cumsum(fg)_i) / (G + cumsum(1 - fg)_i),   grad_i = jaccard_i - jaccard_{i-1}
// in one pass (two launches: per-block sums, then block-local scan with the carried prefix).  torch's cumsum over
// the innermost dimension cost 0.57 ms per [20, 262144] call (4 calls per iteration); this is HBM-bound at
// 8 B per element.  Only the first n_valid positions of a row are real (ignored pixels were sorted to the tail).
#include "common.h"

#define LV_BLOCK 256
#define LV_PER_THREAD 16
#define LV_CHUNK (LV_BLOCK * LV_PER_THREAD)

__global__ __launch_bounds__(LV_BLOCK) void lovasz_sums_k(const float* __restrict__ fg, int64_t P, int nb,
                                                          float* __restrict__ bsum) {
  __shared__ float sh[LV_BLOCK];
  const int c = blockIdx.y, b = blockIdx.x;
  const int64_t base = (int64_t)b * LV_CHUNK + (int64_t)threadIdx.x * LV_PER_THREAD;
  const float* row = fg + (int64_t)c * P;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LV_PER_THREAD; ++k) if (base + k < P) s += row[base + k];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = LV_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[c * nb + b] = sh[0];
}

__global__ __launch_bounds__(LV_BLOCK) void lovasz_grad_k(const float* __restrict__ fg, int64_t P, int nb,
                                                          const float* __restrict__ bsum,
                                                          const int64_t* __restrict__ n_valid_p,
                                                          float* __restrict__ grad) {
  __shared__ float sh[LV_BLOCK];
  __shared__ float carry_s, total_s;
  const int c = blockIdx.y, b = blockIdx.x;
  const int64_t nvalid = *n_valid_p;
  if (threadIdx.x == 0) {    // counts are exact in float32 (< 2^24 pixels)
    float pre = 0.f, tot = 0.f;
    for (int j = 0; j < nb; ++j) { const float v = bsum[c * nb + j]; if (j < b) pre += v; tot += v; }
    carry_s = pre; total_s = tot;
  }
  const int64_t base = (int64_t)b * LV_CHUNK + (int64_t)threadIdx.x * LV_PER_THREAD;
  const float* row = fg + (int64_t)c * P;
  float v[LV_PER_THREAD];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LV_PER_THREAD; ++k) { v[k] = base + k < P ? row[base + k] : 0.f; s += v[k]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  // exclusive scan of the 256 thread sums (Hillis-Steele on LDS)
  for (int o = 1; o < LV_BLOCK; o <<= 1) {
    const float t = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0.f;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  float run = carry_s + sh[threadIdx.x] - s;   // cumsum(fg) just before this thread's first element
  const float G = total_s;
  // jaccard at position i-1 (needed for the first difference)
  float jprev;
  {
    const int64_t i = base - 1;
    if (i < 0) jprev = 0.f;
    else {
      const float uni = G + ((float)(i + 1) - run);
      jprev = 1.f - (G - run) / fmaxf(uni, 1e-12f);
    }
  }
  float* orow = grad + (int64_t)c * P;
#pragma unroll
  for (int k = 0; k < LV_PER_THREAD; ++k) {
    const int64_t i = base + k;
    if (i >= P) break;
    run += v[k];
    const float uni = G + ((float)(i + 1) - run);
    const float j = 1.f - (G - run) / fmaxf(uni, 1e-12f);
    orow[i] = i < nvalid ? (i == 0 ? j : j - jprev) : 0.f;
    jprev = j;
  }
}

// fg_sorted float[C][P] (0/1, rows sorted by descending error, ignored pixels last), n_valid = #valid pixels (device
// int64), bsum = float[C][ceil(P/4096)] scratch  ->  grad float[C][P]
extern "C" int pmf_lovasz_grad(const float* fg_sorted, int32_t C, int64_t P, const int64_t* n_valid, float* bsum,
                               float* grad, pmf_stream_t s) {
  if (C < 1 || P < 1) return PMF_E_ARG;
  const int nb = (int)cdiv64(P, LV_CHUNK);
  hipLaunchKernelGGL(lovasz_sums_k, dim3(nb, C), dim3(LV_BLOCK), 0, (hipStream_t)s, fg_sorted, P, nb, bsum);
  hipLaunchKernelGGL(lovasz_grad_k, dim3(nb, C), dim3(LV_BLOCK), 0, (hipStream_t)s, fg_sorted, P, nb,
                     (const float*)bsum, n_valid, grad);
  PMF_LAUNCH_CHECK();
  return 0;
}
