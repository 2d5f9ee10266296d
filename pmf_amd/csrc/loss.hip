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

// (chunk b, row r) of a row-wise kernel.  2-D launch: (blockIdx.x, blockIdx.y).  1-D launch of 8 * ceil(R / 8) * nb
// workgroups (rows_grid()): consecutive workgroup ids go round-robin over the 8 XCDs, so workgroup L runs on XCD L % 8 and
// that XCD works through rows L % 8, L % 8 + 8, ... chunk by chunk -- every chunk of a row on ONE XCD at about the same
// time.  The row-wise scatters (radix passes into the row's [0, n) range, the Lovasz gradient through the permutation
// into the row's two gradient planes) then read-modify-write lines that sit in that XCD's L2, instead of every L2 holding
// -- and writing back -- a part of every line.  Returns false for the padding rows.
__device__ __forceinline__ bool rows_map(int nb, int R, int* b, int* r) {
  if (gridDim.y > 1 || R <= 0) { *b = blockIdx.x; *r = blockIdx.y; return true; }
  const int L = blockIdx.x, j = L >> 3, slot = j / nb;
  *b = j - slot * nb;
  *r = slot * 8 + (L & 7);
  return *r < R;
}
static inline dim3 rows_grid(int nb, int R) { return dim3((unsigned)(8 * ((R + 7) / 8) * nb), 1, 1); }

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

// ------------------------------------------------------------------------------------------------------------
// PMF training objective, both heads, value AND gradient in one pass (tasks/pmf/trainer.py:231-252, 303-332;
// pc_processor/loss/focal_softmax.py:37-63; lovasz_softmax.py:132-160).  Replaces ~250 element-wise torch launches
// per iteration by:  label histogram -> per-pixel kernel -> [caller: one batched descending sort of the 2C Lovasz key
// rows] -> Jaccard kernels (value + gradient scattered back through the permutation) -> fold.
//   total = foc_l + foc_c + lambda * (lov_l + lov_c) + gamma * per
// Analytic gradients (p = LiDAR probabilities, q = camera probabilities of one pixel, logC = ln C):
//   plog_c = ln max(p_c, 1e-8),  e_p = -sum p_c plog_c / logC,  de_p/dp_c = -(plog_c + [p_c >= 1e-8]) / logC
//   d = e_q - e_p (= confidence difference),  w_pcd = [d>0] |d| [1-e_p >= tau],  w_img = [d<0] |d| [1-e_q >= tau]
//   Lp = sum_c xlogy(q_c,q_c) - q_c plog_c  (KL(q || p), weighted by w_img),  Lq the mirror image (weighted by w_pcd)
//   per = mean over N*C*H*W of  w_img Lp + w_pcd Lq;  the weights are NOT detached in the reference, so
//   d(per M)/dp_c = -w_img q_c [p_c>=1e-8]/p_c + w_pcd (ln p_c + 1 - qlog_c) + (Lq [d>0][pc>=tau] - Lp [d<0][ic>=tau]) dd_c,
//   dd_c = (plog_c + [p_c >= 1e-8]) / logC, and symmetrically for q.
//   focal: f = -(1-pt)^g ln max(pt,1e-6) alpha_t,  df/dpt = alpha_t (g (1-pt)^(g-1) ln max(pt,1e-6) - (1-pt)^g [pt>=1e-6]/pt)
#define LP_BLOCK 256
#define LP_MAXC 32

__global__ __launch_bounds__(256) void loss_count_k(const int64_t* __restrict__ label, int64_t P, int C,
                                                    unsigned long long* __restrict__ cnt) {
  __shared__ unsigned int h[LP_MAXC];
  if (threadIdx.x < LP_MAXC) h[threadIdx.x] = 0u;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)label[i];
    if (t >= 0 && t < C) atomicAdd(&h[t], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < C && h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

struct LossPixelArgs {
  const float* pl; const float* pc;          // probabilities [N][C][HW]
  const int64_t* label;                      // [N][HW]
  const float* alpha;                        // [C]
  const unsigned long long* cnt;             // label histogram [C]
  float* gl; float* gc;                      // gradients (focal + perception part), same layout as pl / pc
  float* key;                                // [2C][P] Lovasz sort keys
  double* rows;                              // [nblocks][4] partial sums: focal lidar, focal camera, perception
  unsigned long long* conf_l; unsigned long long* conf_c;   // [C][C] confusion matrices (pred, label), may be null
  int64_t HW, P;
  int C;
  float fgamma, tau, gamma_per;
  const float* w6;                           // device weights {foc_l, lov_l, foc_c, lov_c, per_p, per_q} or null
};

__global__ __launch_bounds__(LP_BLOCK) void loss_pixel_k(const LossPixelArgs a) {
  __shared__ double red[4][LP_BLOCK];
  __shared__ unsigned int hist[2][LP_MAXC * LP_MAXC];
  const int C = a.C;
  const bool want_conf = a.conf_l != nullptr;
  if (want_conf) {
    for (int k = threadIdx.x; k < C * C; k += LP_BLOCK) { hist[0][k] = 0u; hist[1][k] = 0u; }
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * LP_BLOCK + threadIdx.x;
  double s_fl = 0.0, s_fc = 0.0, s_per = 0.0, s_per2 = 0.0;
  if (i < a.P) {
    const int64_t n = i / a.HW, hw = i - n * a.HW;
    const float* pl = a.pl + n * C * a.HW + hw;
    const float* pc = a.pc + n * C * a.HW + hw;
    const int t = (int)a.label[i];
    const float logC = logf((float)C), ilogC = 1.f / logC;
    float p[LP_MAXC], q[LP_MAXC];
    float ep = 0.f, eq = 0.f, Lp = 0.f, Lq = 0.f;
    int am_l = 0, am_c = 0;
    float mx_l = -1.f, mx_c = -1.f;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
      const float pv = pl[c * a.HW], qv = pc[c * a.HW];
      p[c] = pv; q[c] = qv;
      const float plog = logf(fmaxf(pv, 1e-8f)), qlog = logf(fmaxf(qv, 1e-8f));
      ep -= pv * plog; eq -= qv * qlog;
      Lp += (qv > 0.f ? qv * logf(qv) : 0.f) - qv * plog;     // F.kl_div(plog, q): xlogy(q,q) - q*plog
      Lq += (pv > 0.f ? pv * logf(pv) : 0.f) - pv * qlog;
      if (pv > mx_l) { mx_l = pv; am_l = c; }
      if (qv > mx_c) { mx_c = qv; am_c = c; }
    }
    ep *= ilogC; eq *= ilogC;
    const float conf_p = 1.f - ep, conf_q = 1.f - eq;
    const float d = conf_p - conf_q;
    const float gate_p = (d > 0.f && conf_p >= a.tau) ? 1.f : 0.f;   // w_pcd = gate_p * |d|
    const float gate_q = (d < 0.f && conf_q >= a.tau) ? 1.f : 0.f;   // w_img = gate_q * |d|
    const float w_pcd = gate_p * fabsf(d), w_img = gate_q * fabsf(d);
    const double M = (double)a.P * C;
    s_per = (double)w_img * Lp / M;            // per_p: KL(q || p) weighted by w_img   (trainer.py:247-248)
    s_per2 = (double)w_pcd * Lq / M;           // per_q: KL(p || q) weighted by w_pcd   (trainer.py:249-250)
    // weights of the six terms in the total: the PMF objective (1, lambda, 1, lambda, gamma, gamma) or, for the EPMF
    // multi-task objective (tasks/epmf/trainer.py:409-430), 1 / (2 sigma_i^2) read from device memory
    const float w_fl = a.w6 ? a.w6[0] : 1.f, w_fc = a.w6 ? a.w6[2] : 1.f;
    const float spa = (a.w6 ? a.w6[4] : a.gamma_per) / (float)M, spb = (a.w6 ? a.w6[5] : a.gamma_per) / (float)M;
    // focal
    const bool valid = t > 0 && t < C;
    const double msum = (double)a.P - (double)a.cnt[0];
    float dfl = 0.f, dfc = 0.f;
    if (valid) {
      const float al = a.alpha[t];
      const float ptl = p[t], ptc = q[t];
      const float ll = logf(fmaxf(ptl, 1e-6f)), lc = logf(fmaxf(ptc, 1e-6f));
      const float ol = 1.f - ptl, oc = 1.f - ptc;
      const float pwl = powf(ol, a.fgamma), pwc = powf(oc, a.fgamma);
      s_fl = (double)(-pwl * ll * al) / msum;
      s_fc = (double)(-pwc * lc * al) / msum;
      const float pwl1 = a.fgamma == 1.f ? 1.f : powf(ol, a.fgamma - 1.f), pwc1 = a.fgamma == 1.f ? 1.f : powf(oc, a.fgamma - 1.f);
      dfl = w_fl * al * (a.fgamma * pwl1 * ll - (ptl >= 1e-6f ? pwl / ptl : 0.f)) / (float)msum;
      dfc = w_fc * al * (a.fgamma * pwc1 * lc - (ptc >= 1e-6f ? pwc / ptc : 0.f)) / (float)msum;
    }
    float* gl = a.gl + n * C * a.HW + hw;
    float* gc = a.gc + n * C * a.HW + hw;
    // d(w_pcd Lq)/dd = Lq gate_p and d(w_img Lp)/dd = -Lp gate_q through the (not detached) weights
    const bool lov_valid = t != 0;                  // Lovasz ignore label 0
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
      const float pv = p[c], qv = q[c];
      const float plog = logf(fmaxf(pv, 1e-8f)), qlog = logf(fmaxf(qv, 1e-8f));
      const float ip = pv >= 1e-8f ? 1.f : 0.f, iq = qv >= 1e-8f ? 1.f : 0.f;
      const float ddp = (plog + ip) * ilogC;       // dd/dp_c
      const float ddq = -(qlog + iq) * ilogC;      // dd/dq_c
      float g1 = spa * (-w_img * qv * (ip > 0.f ? 1.f / pv : 0.f) - Lp * gate_q * ddp) +
                 spb * (w_pcd * (logf(fmaxf(pv, 1e-38f)) + 1.f - qlog) + Lq * gate_p * ddp);
      float g2 = spa * (w_img * (logf(fmaxf(qv, 1e-38f)) + 1.f - plog) - Lp * gate_q * ddq) +
                 spb * (-w_pcd * pv * (iq > 0.f ? 1.f / qv : 0.f) + Lq * gate_p * ddq);
      if (c == t) { g1 += dfl; g2 += dfc; }
      gl[c * a.HW] = g1;
      gc[c * a.HW] = g2;
      const float fg = (lov_valid && c == t) ? 1.f : 0.f;
      a.key[(int64_t)c * a.P + i] = lov_valid ? fabsf(fg - pv) : -1.f;
      a.key[(int64_t)(C + c) * a.P + i] = lov_valid ? fabsf(fg - qv) : -1.f;
    }
    if (want_conf && t >= 0 && t < C) {
      atomicAdd(&hist[0][am_l * C + t], 1u);
      atomicAdd(&hist[1][am_c * C + t], 1u);
    }
  }
  red[0][threadIdx.x] = s_fl; red[1][threadIdx.x] = s_fc; red[2][threadIdx.x] = s_per; red[3][threadIdx.x] = s_per2;
  __syncthreads();
  for (int o = LP_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
      red[3][threadIdx.x] += red[3][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* r = a.rows + (size_t)blockIdx.x * 4;
    r[0] = red[0][0]; r[1] = red[1][0]; r[2] = red[2][0]; r[3] = red[3][0];
  }
  if (want_conf) {
    for (int k = threadIdx.x; k < C * C; k += LP_BLOCK) {
      if (hist[0][k]) atomicAdd(&a.conf_l[k], (unsigned long long)hist[0][k]);
      if (hist[1][k]) atomicAdd(&a.conf_c[k], (unsigned long long)hist[1][k]);
    }
  }
}

// Lovasz stage (rows r = head * C + class, sorted descending by error; perm[r][k] = pixel of rank k):
// per-chunk sums of the foreground indicator along the permutation ...
// IT: index type of the permutation (int64: torch.sort's; unsigned: the in-library sort's).  cnt != nullptr (in-library
// sort): only the first P - cnt[0] ranks of a row exist and rows of absent classes / class 0 were never sorted.
template <typename IT>
__global__ __launch_bounds__(LV_BLOCK) void lovasz2_sums_k(const IT* __restrict__ perm, const int64_t* __restrict__ label,
                                                           int64_t P, int C, int nb, float* __restrict__ bsum,
                                                           const unsigned long long* __restrict__ cnt) {
  __shared__ float sh[LV_BLOCK];
  int r, b;
  if (!rows_map(nb, 2 * C, &b, &r)) return;
  const int cls = r % C;
  const int64_t lim = cnt ? P - (int64_t)cnt[0] : P;
  if (cnt && (cls == 0 || cnt[cls] == 0)) return;
  if ((int64_t)b * LV_CHUNK >= lim) {              // nothing ranked here
    if (threadIdx.x == 0) bsum[r * nb + b] = 0.f;
    return;
  }
  // consecutive threads read consecutive ranks (a thread walking its own 16 ranks touched 64 cache lines per load
  // instruction: 171 us per call); the sum of 0/1 values is exact in any order
  const int64_t base = (int64_t)b * LV_CHUNK + threadIdx.x;
  const IT* row = perm + (int64_t)r * P;
  int64_t pix[LV_PER_THREAD];
#pragma unroll
  for (int k = 0; k < LV_PER_THREAD; ++k) pix[k] = base + k * LV_BLOCK < lim ? (int64_t)row[base + k * LV_BLOCK] : -1;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LV_PER_THREAD; ++k)
    s += (cls != 0 && pix[k] >= 0 && pix[k] < P && label[pix[k]] == cls) ? 1.f : 0.f;
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = LV_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[r * nb + b] = sh[0];
}

// ... then the Jaccard first differences, the dot product with the sorted errors (value) and the gradient
// lambda * [class present] / n_present * grad * d|fg - p|/dp, scattered back through the permutation (every (class, pixel)
// pair occurs exactly once per row: plain read-modify-write, no atomics).
template <typename IT, bool LIMIT>
__global__ __launch_bounds__(LV_BLOCK) void lovasz2_grad_k(const IT* __restrict__ perm, const float* __restrict__ key_sorted,
                                                           const int64_t* __restrict__ label, int64_t P, int64_t HW, int C,
                                                           int nb, const float* __restrict__ bsum,
                                                           const unsigned long long* __restrict__ cnt, float lambda,
                                                           const float* __restrict__ w6, float* __restrict__ gl,
                                                           float* __restrict__ gc, double* __restrict__ dots) {
  __shared__ float sh[LV_BLOCK];
  __shared__ double shd[LV_BLOCK];
  __shared__ float carry_s, total_s;
  int r, b;
  if (!rows_map(nb, 2 * C, &b, &r)) return;
  const int cls = r % C, head = r / C;
  const int64_t nvalid = P - (int64_t)cnt[0];
  const int64_t lim = LIMIT ? nvalid : P;
  if (LIMIT && (cls == 0 || cnt[cls] == 0)) return;          // (loss_fold_k skips these rows' dots as well)
  if (LIMIT && (int64_t)b * LV_CHUNK >= lim) {
    if (threadIdx.x == 0) dots[r * nb + b] = 0.0;
    return;
  }
  if (threadIdx.x == 0) {
    float pre = 0.f, tot = 0.f;
    for (int j = 0; j < nb; ++j) { const float v = bsum[r * nb + j]; if (j < b) pre += v; tot += v; }
    carry_s = pre; total_s = tot;
  }
  int npresent = 0;
  for (int c = 1; c < C; ++c) npresent += cnt[c] > 0 ? 1 : 0;
  const float lam = w6 ? w6[head == 0 ? 1 : 3] : lambda;
  const float wcls = (cls != 0 && cnt[cls] > 0) ? lam / (float)(npresent > 0 ? npresent : 1) : 0.f;
  const int64_t base = (int64_t)b * LV_CHUNK + (int64_t)threadIdx.x * LV_PER_THREAD;
  const IT* row = perm + (int64_t)r * P;
  const float* krow = key_sorted + (int64_t)r * P;
  // the chunk is read with consecutive threads on consecutive ranks (coalesced) and handed to its owner -- thread t scans
  // ranks 16 t .. 16 t + 15 -- through LDS: rank j sits at word j + j / 16 (17-word pitch per owner: conflict-free);
  // bit 31 of the pixel word carries the foreground flag (pixels < 2^31)
  __shared__ unsigned s_pix[LV_CHUNK + LV_BLOCK];
  __shared__ float s_err[LV_CHUNK + LV_BLOCK];
  {
    const int64_t cb = (int64_t)b * LV_CHUNK;
    int64_t pl[LV_PER_THREAD];
    float el[LV_PER_THREAD];
#pragma unroll
    for (int k = 0; k < LV_PER_THREAD; ++k) {
      const int64_t i = cb + k * LV_BLOCK + threadIdx.x;
      pl[k] = i < lim ? (int64_t)row[i] : -1;
      if (pl[k] >= P) pl[k] = -1;                  // (never with a well-formed permutation: no out-of-range scatter)
      el[k] = i < lim ? krow[i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < LV_PER_THREAD; ++k) {
      const int j = k * LV_BLOCK + threadIdx.x;
      const bool fg = cls != 0 && pl[k] >= 0 && label[pl[k]] == cls;
      s_pix[j + (j >> 4)] = (pl[k] >= 0 ? (unsigned)pl[k] : 0u) | (fg ? 0x80000000u : 0u);
      s_err[j + (j >> 4)] = el[k];
    }
  }
  __syncthreads();
  float v[LV_PER_THREAD], err[LV_PER_THREAD];
  int64_t pix[LV_PER_THREAD];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LV_PER_THREAD; ++k) {
    const unsigned w = s_pix[threadIdx.x * (LV_PER_THREAD + 1) + k];
    pix[k] = (int64_t)(w & 0x7fffffffu);
    v[k] = (w >> 31) ? 1.f : 0.f;
    err[k] = s_err[threadIdx.x * (LV_PER_THREAD + 1) + k];
    s += v[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < LV_BLOCK; o <<= 1) {
    const float t = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0.f;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  float run = carry_s + sh[threadIdx.x] - s;
  const float G = total_s;
  float jprev;
  {
    const int64_t i = base - 1;
    if (i < 0) jprev = 0.f;
    else {
      const float uni = G + ((float)(i + 1) - run);
      jprev = 1.f - (G - run) / fmaxf(uni, 1e-12f);
    }
  }
  float* g = head == 0 ? gl : gc;
  double dot = 0.0;
#pragma unroll
  for (int k = 0; k < LV_PER_THREAD; ++k) {
    const int64_t i = base + k;
    if (i >= P) break;
    run += v[k];
    const float uni = G + ((float)(i + 1) - run);
    const float j = 1.f - (G - run) / fmaxf(uni, 1e-12f);
    const float gr = i < nvalid ? (i == 0 ? j : j - jprev) : 0.f;
    jprev = j;
    if (i < nvalid) {
      const float e = err[k];
      dot += (double)e * (double)gr;
      if (wcls != 0.f && e != 0.f) {
        const int64_t n = pix[k] / HW, hw = pix[k] - n * HW;
        float* gp = g + (n * C + cls) * HW + hw;
        *gp += wcls * gr * (v[k] > 0.f ? -1.f : 1.f);
      }
    }
  }
  shd[threadIdx.x] = dot;
  __syncthreads();
  for (int o = LV_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) dots[r * nb + b] = shd[0];
}

// out[8] = {total, foc, lov, foc_cam, lov_cam, per (= per_p + per_q), per_p, per_q}; single workgroup, fixed summation order
__global__ __launch_bounds__(256) void loss_fold_k(const double* __restrict__ rows, int nrows, const double* __restrict__ dots,
                                                   int C, int nb, const unsigned long long* __restrict__ cnt, float lambda,
                                                   float gamma_per, const float* __restrict__ w6, float* __restrict__ out) {
  __shared__ double sh[6][256];
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int i = threadIdx.x; i < nrows; i += 256) {
    a0 += rows[(size_t)i * 4]; a1 += rows[(size_t)i * 4 + 1]; a2 += rows[(size_t)i * 4 + 2]; a3 += rows[(size_t)i * 4 + 3];
  }
  // Lovasz: per-class dot products, classes present only
  double l0 = 0, l1 = 0;
  for (int idx = threadIdx.x; idx < 2 * C * nb; idx += 256) {
    const int r = idx / nb, cls = r % C;
    if (cls != 0 && cnt[cls] > 0) { if (r < C) l0 += dots[idx]; else l1 += dots[idx]; }
  }
  sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1; sh[2][threadIdx.x] = a2; sh[3][threadIdx.x] = l0; sh[4][threadIdx.x] = l1;
  sh[5][threadIdx.x] = a3;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 6; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int npresent = 0;
    for (int c = 1; c < C; ++c) npresent += cnt[c] > 0 ? 1 : 0;
    const double np = npresent > 0 ? npresent : 1;
    const double foc = sh[0][0], focc = sh[1][0], perp = sh[2][0], perq = sh[5][0], lov = sh[3][0] / np, lovc = sh[4][0] / np;
    out[1] = (float)foc; out[2] = (float)lov; out[3] = (float)focc; out[4] = (float)lovc; out[5] = (float)(perp + perq);
    out[6] = (float)perp; out[7] = (float)perq;
    if (w6) out[0] = (float)(w6[0] * foc + w6[1] * lov + w6[2] * focc + w6[3] * lovc + w6[4] * perp + w6[5] * perq);
    else out[0] = (float)(foc + focc + lambda * (lov + lovc) + gamma_per * (perp + perq));
  }
}

extern "C" int pmf_loss_rows(int64_t P) { return (int)cdiv64(P, LP_BLOCK); }
extern "C" int pmf_loss_chunks(int64_t P) { return (int)cdiv64(P, LV_CHUNK); }

static int loss_pixel_impl(const float* lidar_prob, const float* camera_prob, const int64_t* label, const float* alpha,
                           int32_t N, int32_t C, int64_t HW, float focal_gamma, float tau, float gamma_per,
                           const float* w6, unsigned long long* cnt, float* grad_lidar, float* grad_camera, float* key,
                           double* rows, unsigned long long* conf_lidar, unsigned long long* conf_camera, pmf_stream_t s) {
  if (C < 2 || C > LP_MAXC || N < 1 || HW < 1) return PMF_E_ARG;
  const int64_t P = (int64_t)N * HW;
  hipStream_t st = (hipStream_t)s;
  hipError_t e = hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * C, st);
  if (e != hipSuccess) return (int)e;
  int gb = (int)cdiv64(P, 256 * 8);
  hipLaunchKernelGGL(loss_count_k, dim3(gb > 1024 ? 1024 : gb), dim3(256), 0, st, label, P, C, cnt);
  LossPixelArgs a;
  a.pl = lidar_prob; a.pc = camera_prob; a.label = label; a.alpha = alpha; a.cnt = cnt;
  a.gl = grad_lidar; a.gc = grad_camera; a.key = key; a.rows = rows; a.conf_l = conf_lidar; a.conf_c = conf_camera;
  a.HW = HW; a.P = P; a.C = C; a.fgamma = focal_gamma; a.tau = tau; a.gamma_per = gamma_per; a.w6 = w6;
  hipLaunchKernelGGL(loss_pixel_k, dim3((unsigned)cdiv64(P, LP_BLOCK)), dim3(LP_BLOCK), 0, st, a);
  PMF_LAUNCH_CHECK();
  return 0;
}
extern "C" int pmf_loss_pixel(const float* lidar_prob, const float* camera_prob, const int64_t* label, const float* alpha,
                              int32_t N, int32_t C, int64_t HW, float focal_gamma, float tau, float gamma_per,
                              unsigned long long* cnt, float* grad_lidar, float* grad_camera, float* key, double* rows,
                              unsigned long long* conf_lidar, unsigned long long* conf_camera, pmf_stream_t s) {
  return loss_pixel_impl(lidar_prob, camera_prob, label, alpha, N, C, HW, focal_gamma, tau, gamma_per, nullptr, cnt,
                         grad_lidar, grad_camera, key, rows, conf_lidar, conf_camera, s);
}
extern "C" int pmf_loss_pixel_w(const float* lidar_prob, const float* camera_prob, const int64_t* label, const float* alpha,
                                int32_t N, int32_t C, int64_t HW, float focal_gamma, float tau, const float* w6,
                                unsigned long long* cnt, float* grad_lidar, float* grad_camera, float* key, double* rows,
                                unsigned long long* conf_lidar, unsigned long long* conf_camera, pmf_stream_t s) {
  if (!w6) return PMF_E_ARG;
  return loss_pixel_impl(lidar_prob, camera_prob, label, alpha, N, C, HW, focal_gamma, tau, 0.f, w6, cnt, grad_lidar,
                         grad_camera, key, rows, conf_lidar, conf_camera, s);
}

static int loss_lovasz_impl(const int64_t* perm, const float* key_sorted, const int64_t* label, int32_t N, int32_t C,
                            int64_t HW, const unsigned long long* cnt, float lambda, float gamma_per, const float* w6,
                            float* bsum, double* dots, const double* rows, float* grad_lidar, float* grad_camera,
                            float* out6, pmf_stream_t s) {
  if (C < 2 || C > LP_MAXC || N < 1 || HW < 1) return PMF_E_ARG;
  const int64_t P = (int64_t)N * HW;
  const int nb = (int)cdiv64(P, LV_CHUNK);
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(lovasz2_sums_k<int64_t>, dim3(nb, 2 * C), dim3(LV_BLOCK), 0, st, perm, label, P, C, nb, bsum,
                     (const unsigned long long*)nullptr);
  hipLaunchKernelGGL((lovasz2_grad_k<int64_t, false>), dim3(nb, 2 * C), dim3(LV_BLOCK), 0, st, perm, key_sorted, label, P, HW, C,
                     nb, (const float*)bsum, cnt, lambda, w6, grad_lidar, grad_camera, dots);
  hipLaunchKernelGGL(loss_fold_k, dim3(1), dim3(256), 0, st, rows, (int)cdiv64(P, LP_BLOCK), (const double*)dots, C, nb, cnt,
                     lambda, gamma_per, w6, out6);
  PMF_LAUNCH_CHECK();
  return 0;
}
extern "C" int pmf_loss_lovasz(const int64_t* perm, const float* key_sorted, const int64_t* label, int32_t N, int32_t C,
                               int64_t HW, const unsigned long long* cnt, float lambda, float gamma_per, float* bsum,
                               double* dots, const double* rows, float* grad_lidar, float* grad_camera, float* out8,
                               pmf_stream_t s) {
  return loss_lovasz_impl(perm, key_sorted, label, N, C, HW, cnt, lambda, gamma_per, nullptr, bsum, dots, rows, grad_lidar,
                          grad_camera, out8, s);
}
extern "C" int pmf_loss_lovasz_w(const int64_t* perm, const float* key_sorted, const int64_t* label, int32_t N, int32_t C,
                                 int64_t HW, const unsigned long long* cnt, const float* w6, float* bsum, double* dots,
                                 const double* rows, float* grad_lidar, float* grad_camera, float* out8, pmf_stream_t s) {
  if (!w6) return PMF_E_ARG;
  return loss_lovasz_impl(perm, key_sorted, label, N, C, HW, cnt, 0.f, 0.f, w6, bsum, dots, rows, grad_lidar, grad_camera,
                          out8, s);
}


// ---- in-library Lovasz sort -----------------------------------------------------------------------------------------
// torch.sort on the [2C, P] key matrix runs one device radix sort per row (40 rows of 262 144 keys: ~13 us each, 0.53 ms per
// step) and sorts every pixel, although only pixels with a label (label != 0: the image-plane fill of a LiDAR sweep is
// 5-15 %) take part in the Lovasz extension and rows of class 0 / absent classes are never read.  Here all rows are sorted
// together by four stable 8-bit counting passes over the 30 significant bits of the error (errors lie in [0, 1], so their
// float bit patterns order like integers; key' = 0x3FFFFFFF - bits gives descending errors from an ascending sort); the
// first pass reads the raw key matrix and drops ignored pixels (key -1) -- it is the compaction -- and every later pass
// touches only the P - cnt[0] labelled pixels of the rows that are used; the element count is read on the device, no host
// round trip.  Equal errors keep ascending pixel order (deterministic).  Per pass: per-chunk digit histograms, one scan
// workgroup per row, stable scatter (wave-level digit matching with ballots).
#define RS_TILE 256
#define RS_SUB 16
#define RS_CHUNK (RS_TILE * RS_SUB)

__device__ __forceinline__ unsigned rs_key_of(float e) { return 0x3FFFFFFFu - __float_as_uint(e); }

template <bool FIRST>
__global__ __launch_bounds__(RS_TILE) void rs_hist_k(const float* __restrict__ key, const unsigned* __restrict__ kin, int64_t P,
                                                     int C, int nb, int shift, const unsigned long long* __restrict__ cnt,
                                                     unsigned* __restrict__ hist) {
  __shared__ unsigned h[256];
  const int r = blockIdx.y, b = blockIdx.x, cls = r % C;
  if (cls == 0 || cnt[cls] == 0) return;
  const int64_t n = FIRST ? P : P - (int64_t)cnt[0];
  const int64_t base = (int64_t)b * RS_CHUNK;
  h[threadIdx.x] = 0u;
  __syncthreads();
  if (base < n) {
#pragma unroll 4
    for (int s = 0; s < RS_SUB; ++s) {
      const int64_t i = base + s * RS_TILE + threadIdx.x;
      if (i < n) {
        unsigned k;
        bool ok = true;
        if (FIRST) { const float e = key[(int64_t)r * P + i]; ok = !(e < 0.f); k = rs_key_of(e); }   // (NaN is kept: see rs_scatter_k)
        else k = kin[(int64_t)r * P + i];
        if (ok) atomicAdd(&h[(k >> shift) & 255u], 1u);
      }
    }
  }
  __syncthreads();
  hist[((int64_t)r * nb + b) * 256 + threadIdx.x] = h[threadIdx.x];      // [row][chunk][digit]: coalesced here and in the scan
}

// one workgroup per row: hist[r][chunk][digit] -> exclusive offsets in (digit, chunk) order, in place
// (nb_all chunks per row in the table; after the compacting first pass only the chunks below P - cnt[0] keys hold anything:
// P > 0 limits both loops to those -- 18 instead of 128 at the bench's fill)
__global__ __launch_bounds__(256) void rs_scan_k(unsigned* __restrict__ hist, int C, int nb_all, const unsigned long long* __restrict__ cnt,
                                                 int64_t P) {
  __shared__ unsigned tot[256];
  const int r = blockIdx.x, cls = r % C, d = threadIdx.x;
  if (cls == 0 || cnt[cls] == 0) return;
  int nb = nb_all;
  if (P > 0) {
    const int64_t na = (P - (int64_t)cnt[0] + RS_CHUNK - 1) / RS_CHUNK;
    nb = na < nb_all ? (int)na : nb_all;
  }
  unsigned* hp = hist + (int64_t)r * nb_all * 256 + d;
  unsigned s = 0;
  for (int b0 = 0; b0 < nb; b0 += 8) {          // eight independent loads per trip
    unsigned c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = b0 + u < nb ? hp[(b0 + u) * 256] : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += c[u];
  }
  tot[d] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned t = d >= o ? tot[d - o] : 0u;
    __syncthreads();
    tot[d] += t;
    __syncthreads();
  }
  unsigned run = tot[d] - s;
  for (int b0 = 0; b0 < nb; b0 += 8) {          // eight independent loads per trip
    unsigned c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = b0 + u < nb ? hp[(b0 + u) * 256] : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (b0 + u < nb) { hp[(b0 + u) * 256] = run; run += c[u]; }
  }
}

// Stable scatter of one 4096-key chunk.  Wave w owns the contiguous quarter [w * 1024, (w + 1) * 1024) of the chunk, 16 rounds of
// 64 consecutive keys: a key's rank among the equal digits of its OWN wave needs no workgroup barrier (ballot matching +
// a wave-private running counter per digit in LDS; LDS operations of one wave execute in order), so the sixteen rounds run
// back to back; ONE barrier later the per-wave digit totals are turned into the waves' start offsets and the keys go out.
// (Round 3 ranked workgroup-wide round by round: three barriers in each of the 16 rounds -- 45 us per pass for 2.8 M keys.)
template <bool FIRST>
__global__ __launch_bounds__(RS_TILE) void rs_scatter_k(const float* __restrict__ key, const unsigned* __restrict__ kin,
                                                        const unsigned* __restrict__ vin, int64_t P, int C, int nb, int shift,
                                                        const unsigned long long* __restrict__ cnt,
                                                        const unsigned* __restrict__ offs, unsigned* __restrict__ kout,
                                                        unsigned* __restrict__ vout) {
  constexpr int NW = RS_TILE / 64;
  __shared__ unsigned wc[NW][256];
  int r, b;
  if (!rows_map(nb, 2 * C, &b, &r)) return;
  const int cls = r % C;
  if (cls == 0 || cnt[cls] == 0) return;
  const int64_t n = FIRST ? P : P - (int64_t)cnt[0];
  const int64_t base = (int64_t)b * RS_CHUNK;
  if (base >= n) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned run0 = offs[((int64_t)r * nb + b) * 256 + t];
#pragma unroll
  for (int w = 0; w < NW; ++w) wc[w][t] = 0u;
  __syncthreads();
  const int64_t ro = (int64_t)r * P;
  // the wave's 1024 keys first (16 independent loads per thread in flight), then the sixteen ranking rounds
  unsigned ks[RS_SUB], vs[RS_SUB], pos[RS_SUB];
  unsigned okm = 0u;
#pragma unroll
  for (int s = 0; s < RS_SUB; ++s) {
    const int64_t i = base + wave * (RS_CHUNK / NW) + s * 64 + lane;
    bool ok = i < n;
    ks[s] = 0u; vs[s] = 0u;
    if (ok) {
      if (FIRST) {
        // dropped: exactly the -1 sentinel of ignored pixels.  NOT "e >= 0": a NaN error (diverged training) must stay, or
        // fewer than P - cnt[0] keys survive and the row tails every later pass / the Lovasz kernels read are never written
        const float e = key[ro + i]; ok = !(e < 0.f); ks[s] = rs_key_of(e); vs[s] = (unsigned)i;
      }
      else { ks[s] = kin[ro + i]; vs[s] = vin[ro + i]; }
    }
    okm |= ok ? (1u << s) : 0u;
  }
  unsigned* __restrict__ mine = wc[wave];
#pragma unroll
  for (int s = 0; s < RS_SUB; ++s) {
    const bool ok = (okm >> s) & 1u;
    const unsigned d = (ks[s] >> shift) & 255u;
    unsigned long long m = __ballot(ok);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (d >> bit) & 1u;
      const unsigned long long bal = __ballot(one);
      m &= one ? bal : ~bal;
    }
    const unsigned long long peers = ok ? m : 0ull;
    const unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
    const unsigned before = ok ? mine[d] : 0u;                    // equal digits of this wave in earlier rounds
    pos[s] = before + rank;
    __builtin_amdgcn_wave_barrier();                              // every lane has read before the group's first lane adds
    if (ok && rank == 0u) mine[d] = before + (unsigned)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {   // digit t: chunk offset + the totals of the waves in front
    unsigned a = run0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const unsigned c = wc[w][t]; wc[w][t] = a; a += c; }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < RS_SUB; ++s) {
    if ((okm >> s) & 1u) {
      const int64_t dst = ro + mine[(ks[s] >> shift) & 255u] + pos[s];
      kout[dst] = ks[s];
      vout[dst] = vs[s];
    }
  }
}

// sorted key' -> sorted errors (float), for the rows that were sorted
__global__ __launch_bounds__(256) void rs_unkey_k(const unsigned* __restrict__ kin, float* __restrict__ e, int64_t P, int C,
                                                  const unsigned long long* __restrict__ cnt) {
  const int r = blockIdx.y, cls = r % C;
  if (cls == 0 || cnt[cls] == 0) return;
  const int64_t n = P - (int64_t)cnt[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    e[(int64_t)r * P + i] = __uint_as_float(0x3FFFFFFFu - kin[(int64_t)r * P + i]);
}

extern "C" int64_t pmf_loss_sort_workspace(int32_t C, int64_t P) {
  const int64_t nb = cdiv64(P, RS_CHUNK);
  return 4 * (int64_t)2 * C * P * 4 + (int64_t)2 * C * 256 * nb * 4;       // 2 key + 2 value buffers, histograms
}

static int loss_lovasz_sort_impl(const float* key, const int64_t* label, int32_t N, int32_t C, int64_t HW,
                                 const unsigned long long* cnt, float lambda, float gamma_per, const float* w6, void* ws,
                                 float* bsum, double* dots, const double* rows, float* grad_lidar, float* grad_camera,
                                 float* out8, pmf_stream_t s) {
  if (C < 2 || C > LP_MAXC || N < 1 || HW < 1 || !ws) return PMF_E_ARG;
  const int64_t P = (int64_t)N * HW;
  if (P >= (1ll << 31)) return PMF_E_UNSUPPORTED;
  const int nb = (int)cdiv64(P, RS_CHUNK), R = 2 * C;
  hipStream_t st = (hipStream_t)s;
  unsigned* kA = (unsigned*)ws;
  unsigned* kB = kA + (int64_t)R * P;
  unsigned* vA = kB + (int64_t)R * P;
  unsigned* vB = vA + (int64_t)R * P;
  unsigned* hist = vB + (int64_t)R * P;
  const dim3 grid(nb, R), blk(RS_TILE);
  static const bool xcd_rows = getenv("PMF_LOSS_XCD_ROWS") == nullptr || atoi(getenv("PMF_LOSS_XCD_ROWS")) != 0;   // A/B knob
  const dim3 xgrid = xcd_rows ? rows_grid(nb, R) : grid;
  // pass 0 (bits 0-7): raw keys -> A (compaction); 1: A -> B; 2: B -> A; 3: A -> B
  hipLaunchKernelGGL(rs_hist_k<true>, grid, blk, 0, st, key, (const unsigned*)nullptr, P, C, nb, 0, cnt, hist);
  hipLaunchKernelGGL(rs_scan_k, dim3(R), dim3(256), 0, st, hist, C, nb, cnt, (int64_t)0);
  hipLaunchKernelGGL(rs_scatter_k<true>, xgrid, blk, 0, st, key, (const unsigned*)nullptr, (const unsigned*)nullptr, P, C, nb, 0,
                     cnt, (const unsigned*)hist, kA, vA);
  unsigned* ki = kA; unsigned* vi = vA; unsigned* ko = kB; unsigned* vo = vB;
  for (int pass = 1; pass < 4; ++pass) {
    hipLaunchKernelGGL(rs_hist_k<false>, grid, blk, 0, st, (const float*)nullptr, (const unsigned*)ki, P, C, nb, 8 * pass, cnt, hist);
    hipLaunchKernelGGL(rs_scan_k, dim3(R), dim3(256), 0, st, hist, C, nb, cnt, P);
    hipLaunchKernelGGL(rs_scatter_k<false>, xgrid, blk, 0, st, (const float*)nullptr, (const unsigned*)ki, (const unsigned*)vi, P, C,
                       nb, 8 * pass, cnt, (const unsigned*)hist, ko, vo);
    unsigned* t = ki; ki = ko; ko = t;
    t = vi; vi = vo; vo = t;
  }
  // sorted: keys in ki, pixel indices in vi; the errors as floats go to the spare key buffer
  float* es = (float*)ko;
  hipLaunchKernelGGL(rs_unkey_k, dim3(nb, R), dim3(256), 0, st, (const unsigned*)ki, es, P, C, cnt);
  const int nbl = (int)cdiv64(P, LV_CHUNK);
  const dim3 lgrid = xcd_rows ? rows_grid(nbl, R) : dim3(nbl, R);
  hipLaunchKernelGGL(lovasz2_sums_k<unsigned>, lgrid, dim3(LV_BLOCK), 0, st, (const unsigned*)vi, label, P, C, nbl, bsum, cnt);
  hipLaunchKernelGGL((lovasz2_grad_k<unsigned, true>), lgrid, dim3(LV_BLOCK), 0, st, (const unsigned*)vi, (const float*)es,
                     label, P, HW, C, nbl, (const float*)bsum, cnt, lambda, w6, grad_lidar, grad_camera, dots);
  hipLaunchKernelGGL(loss_fold_k, dim3(1), dim3(256), 0, st, rows, (int)cdiv64(P, LP_BLOCK), (const double*)dots, C, nbl, cnt,
                     lambda, gamma_per, w6, out8);
  PMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pmf_loss_lovasz_sort(const float* key, const int64_t* label, int32_t N, int32_t C, int64_t HW,
                                    const unsigned long long* cnt, float lambda, float gamma_per, void* workspace, float* bsum,
                                    double* dots, const double* rows, float* grad_lidar, float* grad_camera, float* out8,
                                    pmf_stream_t s) {
  return loss_lovasz_sort_impl(key, label, N, C, HW, cnt, lambda, gamma_per, nullptr, workspace, bsum, dots, rows, grad_lidar,
                               grad_camera, out8, s);
}
extern "C" int pmf_loss_lovasz_sort_w(const float* key, const int64_t* label, int32_t N, int32_t C, int64_t HW,
                                      const unsigned long long* cnt, const float* w6, void* workspace, float* bsum, double* dots,
                                      const double* rows, float* grad_lidar, float* grad_camera, float* out8, pmf_stream_t s) {
  if (!w6) return PMF_E_ARG;
  return loss_lovasz_sort_impl(key, label, N, C, HW, cnt, 0.f, 0.f, w6, workspace, bsum, dots, rows, grad_lidar, grad_camera,
                               out8, s);
}
