// KNN post-processing vote (pc_processor/postproc/knn.py:55-143) as one lane per point on gfx950.
// The reference materialises two [1, S*S, H*W] unfolds and four [1, S*S, P] gathers; here each lane keeps its
// S*S window (weighted |range difference| and neighbour labels) in registers, selects the k smallest by
// repeated first-minimum (ties -> smaller window index, the rule the CPU oracle pins) and votes.
// HBM-bound: algorithmic bytes 12*H*W + 28*P per call.
#include "common.h"
#pragma clang fp contract(off)

template <int S>
__global__ __launch_bounds__(256) void knn_k(const float* __restrict__ pr, const float* __restrict__ ur,
                                             const int64_t* __restrict__ am, const int64_t* __restrict__ px,
                                             const int64_t* __restrict__ py, int H, int W, int64_t P, int knn,
                                             const float* __restrict__ invg, float cutoff, int nclasses,
                                             int64_t* __restrict__ labels, const int32_t* __restrict__ am32 = nullptr) {
  constexpr int S2 = S * S, PAD = (S - 1) / 2, CENTER = (S2 - 1) / 2;
  __shared__ float wsh[S2];
  if (threadIdx.x < S2) wsh[threadIdx.x] = invg[threadIdx.x];
  __syncthreads();
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int cx = (int)px[i], cy = (int)py[i];
  const float r = ur[i];
  float dist[S2];
  int lab[S2];
#pragma unroll
  for (int t = 0; t < S2; ++t) {
    const int y = cy + t / S - PAD, x = cx + t % S - PAD;
    float v = 0.f;   // F.unfold zero padding: range 0, label 0
    int l = 0;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      v = pr[(size_t)y * W + x];
      l = am32 ? am32[(size_t)y * W + x] : (int)am[(size_t)y * W + x];
      if (v < 0.f) v = INFINITY;
    }
    if (t == CENTER) v = r;
    dist[t] = fabsf(v - r) * wsh[t];   // |neigh - range| * (1 - gauss), float32, knn.py:97-108
    lab[t] = l;
  }
  // k x first-minimum selection
  unsigned long long used = 0ull;
  int sel[S2 < 8 ? 8 : 8];
  int nsel = knn < 8 ? knn : 8;
  for (int k = 0; k < nsel; ++k) {
    float best = 0.f;
    int bi = -1;
#pragma unroll
    for (int t = 0; t < S2; ++t) {
      const bool free_ = !((used >> t) & 1ull);
      if (free_ && (bi < 0 || dist[t] < best)) { best = dist[t]; bi = t; }
    }
    used |= 1ull << bi;
    int l = 0;
#pragma unroll
    for (int t = 0; t < S2; ++t) if (t == bi) l = lab[t];
    if (cutoff > 0.f && best > cutoff) l = nclasses;
    sel[k] = l;
  }
  int best_cnt = 0, best_cls = 1;
  for (int a = 0; a < nsel; ++a) {
    const int cls = sel[a];
    if (cls < 1 || cls >= nclasses) continue;
    int cnt = 0;
    for (int b = 0; b < nsel; ++b) cnt += sel[b] == cls;
    if (cnt > best_cnt || (cnt == best_cnt && cls < best_cls)) { best_cnt = cnt; best_cls = cls; }
  }
  labels[i] = best_cls;
}

// ---- any odd window (search = 1, 9, 11, ...; the reference accepts every odd size, knn.py:73-74; the nuScenes config ships
// search 11, tasks/pmf_eval_nuscenes/config_server_nus.yaml) -------------------------------------------------------------
// One lane per point, window size at run time.  k x first-minimum with ties -> smaller window index is what a STABLE
// insertion into an ascending list of length k produces (an equal distance seen later never moves in front of an earlier
// one); the vote only counts labels, so the order inside the list does not matter.  Same float32 arithmetic as knn_k:
// bit-identical labels where both apply (tests).  Frames of a batch through the offsets table (nullptr: one frame).
__global__ __launch_bounds__(64) void knn_any_k(const float* __restrict__ pr, const float* __restrict__ ur,
                                                const int64_t* __restrict__ am, const int64_t* __restrict__ px,
                                                const int64_t* __restrict__ py, const int64_t* __restrict__ offsets, int B,
                                                int H, int W, int64_t P, int knn, int S, const float* __restrict__ invg,
                                                float cutoff, int nclasses, int64_t* __restrict__ labels,
                                                const int32_t* __restrict__ am32 = nullptr) {
  const int64_t i = blockIdx.x * (int64_t)64 + threadIdx.x;
  if (i >= P) return;
  int b = 0;
  if (offsets) for (int k = 1; k < B; ++k) b += offsets[k] <= i;
  const float* __restrict__ prb = pr + (size_t)b * H * W;
  const int64_t* __restrict__ amb = am + (size_t)b * H * W;
  const int32_t* __restrict__ amb32 = am32 ? am32 + (size_t)b * H * W : nullptr;
  const int cx = (int)px[i], cy = (int)py[i], PAD = (S - 1) / 2, CENTER = (S * S - 1) / 2;
  const float r = ur[i];
  float bd[8];
  int bl[8];
  const int nsel = knn < 8 ? knn : 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) { bd[k] = INFINITY; bl[k] = 0; }
  int filled = 0;
  for (int ty = 0, t = 0; ty < S; ++ty)
    for (int tx = 0; tx < S; ++tx, ++t) {
      const int y = cy + ty - PAD, x = cx + tx - PAD;
      float v = 0.f;
      int l = 0;
      if (y >= 0 && y < H && x >= 0 && x < W) {
        v = prb[(size_t)y * W + x];
        l = amb32 ? amb32[(size_t)y * W + x] : (int)amb[(size_t)y * W + x];
        if (v < 0.f) v = INFINITY;
      }
      if (t == CENTER) v = r;
      float d = fabsf(v - r) * invg[t];
      // position = number of kept entries with distance <= d (stable); NaN-free: distances are >= 0 or +inf
      int pos = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) pos += (k < filled && bd[k] <= d) ? 1 : 0;
      if (pos < nsel) {
#pragma unroll
        for (int k = 7; k > 0; --k)
          if (k > pos && k < nsel) { bd[k] = bd[k - 1]; bl[k] = bl[k - 1]; }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k == pos) { bd[k] = d; bl[k] = l; }
        filled = filled < nsel ? filled + 1 : filled;
      }
    }
  int best_cnt = 0, best_cls = 1;
  for (int a = 0; a < nsel; ++a) {
    int cls = bl[a];
    if (cutoff > 0.f && bd[a] > cutoff) cls = nclasses;
    if (cls < 1 || cls >= nclasses) continue;
    int cnt = 0;
    for (int c = 0; c < nsel; ++c) {
      int cc = bl[c];
      if (cutoff > 0.f && bd[c] > cutoff) cc = nclasses;
      cnt += cc == cls;
    }
    if (cnt > best_cnt || (cnt == best_cnt && cls < best_cls)) { best_cnt = cnt; best_cls = cls; }
  }
  labels[i] = best_cls;
}

template <int S>
__global__ void knn_batch_lds_k(const float* __restrict__ pr, const float* __restrict__ ur, const int64_t* __restrict__ am,
                                const int64_t* __restrict__ px, const int64_t* __restrict__ py,
                                const int64_t* __restrict__ offsets, int B, int H, int W, int64_t P1, int knn,
                                const float* __restrict__ invg, float cutoff, int nclasses, int64_t* __restrict__ labels,
                                const int32_t* __restrict__ am32 = nullptr);

extern "C" int pmf_knn_vote(const float* proj_range, const float* unproj_range, const int64_t* proj_argmax,
                            const int64_t* px, const int64_t* py, int32_t H, int32_t W, int64_t P, int32_t knn,
                            int32_t search, const float* inv_gauss, float cutoff, int32_t nclasses, int64_t* labels,
                            pmf_stream_t s) {
  if (search % 2 == 0) return PMF_E_ARG;  // knn.py:73-74 raises ValueError
  if (knn < 1 || knn > 8 || knn > search * search) return PMF_E_UNSUPPORTED;
  if (P <= 0) return 0;
  dim3 grid((unsigned)cdiv64(P, 256)), block(256);
  hipStream_t st = (hipStream_t)s;
  constexpr bool no_lds = false;
  if (search < 1 || search > 255) return PMF_E_ARG;
  if (!no_lds && (search == 3 || search == 5)) {        // the LDS-staged form (below), one frame, no offsets table
    if (search == 3) hipLaunchKernelGGL(knn_batch_lds_k<3>, grid, block, 0, st, proj_range, unproj_range, proj_argmax, px, py, (const int64_t*)nullptr, 1, H, W, P, knn, inv_gauss, cutoff, nclasses, labels);
    else hipLaunchKernelGGL(knn_batch_lds_k<5>, grid, block, 0, st, proj_range, unproj_range, proj_argmax, px, py, (const int64_t*)nullptr, 1, H, W, P, knn, inv_gauss, cutoff, nclasses, labels);
    PMF_LAUNCH_CHECK();
    return 0;
  }
  switch (search) {
    case 3: hipLaunchKernelGGL(knn_k<3>, grid, block, 0, st, proj_range, unproj_range, proj_argmax, px, py, H, W, P, knn, inv_gauss, cutoff, nclasses, labels); break;
    case 5: hipLaunchKernelGGL(knn_k<5>, grid, block, 0, st, proj_range, unproj_range, proj_argmax, px, py, H, W, P, knn, inv_gauss, cutoff, nclasses, labels); break;
    case 7: hipLaunchKernelGGL(knn_k<7>, grid, block, 0, st, proj_range, unproj_range, proj_argmax, px, py, H, W, P, knn, inv_gauss, cutoff, nclasses, labels); break;
    default:     // every other odd window (1, 9, 11, ...): run-time window size
      hipLaunchKernelGGL(knn_any_k, dim3((unsigned)cdiv64(P, 64)), dim3(64), 0, st, proj_range, unproj_range, proj_argmax, px, py, (const int64_t*)nullptr, 1, H, W, P, knn, search, inv_gauss, cutoff, nclasses, labels);
  }
  PMF_LAUNCH_CHECK();
  return 0;
}

// ---- all frames of a batch in ONE launch (BASELINE configs[1]: bs = 4) ------------------------------------------------
// Range images / argmax maps are [B, H, W]; the points of all frames are concatenated, frame b owning
// [offsets[b], offsets[b+1]).  64-thread workgroups (one wave): a 25 k-point frame is 391 workgroups instead of 98, so four
// frames fill the 256 CUs ~6x over and the ~50 dependent-address loads per lane of one wave overlap with its
// neighbours' on the CU.  Same arithmetic as knn_k (bit-identical labels).
template <int S>
__global__ __launch_bounds__(64) void knn_batch_k(const float* __restrict__ pr, const float* __restrict__ ur,
                                                  const int64_t* __restrict__ am, const int64_t* __restrict__ px,
                                                  const int64_t* __restrict__ py, const int64_t* __restrict__ offsets,
                                                  int B, int H, int W, int64_t P, int knn, const float* __restrict__ invg,
                                                  float cutoff, int nclasses, int64_t* __restrict__ labels,
                                                  const int32_t* __restrict__ am32 = nullptr) {
  constexpr int S2 = S * S, PAD = (S - 1) / 2, CENTER = (S2 - 1) / 2;
  __shared__ float wsh[S2];
  if (threadIdx.x < S2) wsh[threadIdx.x] = invg[threadIdx.x];
  __syncthreads();
  const int64_t i = blockIdx.x * (int64_t)64 + threadIdx.x;
  if (i >= P) return;
  int b = 0;
  for (int k = 1; k < B; ++k) b += offsets[k] <= i;     // B is small (a batch): wave-uniform scalar loads
  const float* __restrict__ prb = pr + (size_t)b * H * W;
  const int64_t* __restrict__ amb = am + (size_t)b * H * W;
  const int32_t* __restrict__ amb32 = am32 ? am32 + (size_t)b * H * W : nullptr;
  const int cx = (int)px[i], cy = (int)py[i];
  const float r = ur[i];
  float dist[S2];
  int lab[S2];
#pragma unroll
  for (int t = 0; t < S2; ++t) {
    const int y = cy + t / S - PAD, x = cx + t % S - PAD;
    float v = 0.f;
    int l = 0;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      v = prb[(size_t)y * W + x];
      l = amb32 ? amb32[(size_t)y * W + x] : (int)amb[(size_t)y * W + x];
      if (v < 0.f) v = INFINITY;
    }
    if (t == CENTER) v = r;
    dist[t] = fabsf(v - r) * wsh[t];
    lab[t] = l;
  }
  unsigned long long used = 0ull;
  int sel[8];
  int nsel = knn < 8 ? knn : 8;
  for (int k = 0; k < nsel; ++k) {
    float best = 0.f;
    int bi = -1;
#pragma unroll
    for (int t = 0; t < S2; ++t) {
      const bool free_ = !((used >> t) & 1ull);
      if (free_ && (bi < 0 || dist[t] < best)) { best = dist[t]; bi = t; }
    }
    used |= 1ull << bi;
    int l = 0;
#pragma unroll
    for (int t = 0; t < S2; ++t) if (t == bi) l = lab[t];
    if (cutoff > 0.f && best > cutoff) l = nclasses;
    sel[k] = l;
  }
  int best_cnt = 0, best_cls = 1;
  for (int a = 0; a < nsel; ++a) {
    const int cls = sel[a];
    if (cls < 1 || cls >= nclasses) continue;
    int cnt = 0;
    for (int c = 0; c < nsel; ++c) cnt += sel[c] == cls;
    if (cnt > best_cnt || (cnt == best_cnt && cls < best_cls)) { best_cnt = cnt; best_cls = cls; }
  }
  labels[i] = best_cls;
}

// ---- the same vote with the window staged through LDS ------------------------------------------------------------------
// A gather of one window tap touches one cache line PER LANE when the lanes of a wave sit on different image rows -- and in
// sweep-file order (azimuth by azimuth) consecutive points are the lasers of one column: 50 fully divergent gathers per
// point, the texture-address unit processes them line by line (21 us for 102 k points; random order 33 us).  Here a workgroup
// of 256 consecutive points of ONE frame takes the bounding box of its points (+ the window margin): in sweep order that is
// ~5 columns x all rows, a few hundred pixels; (range, label) of the box are staged into LDS once (rows of the box are
// contiguous: ~10 wave loads per map) and all window taps are read from there.  Zero padding / negative-range handling
// happen at staging time with the same rules, the selection and the vote are the code above: bit-identical labels.  A box
// above KNN_LDS_PIX pixels (points in random order) uses the global gathers as before, decided per workgroup.
#define KNN_LDS_PIX 4096
template <int S>
__global__ __launch_bounds__(256) void knn_batch_lds_k(const float* __restrict__ pr, const float* __restrict__ ur,
                                                       const int64_t* __restrict__ am, const int64_t* __restrict__ px,
                                                       const int64_t* __restrict__ py, const int64_t* __restrict__ offsets,
                                                       int B, int H, int W, int64_t P1, int knn, const float* __restrict__ invg,
                                                       float cutoff, int nclasses, int64_t* __restrict__ labels,
                                                       const int32_t* __restrict__ am32) {
  constexpr int S2 = S * S, PAD = (S - 1) / 2, CENTER = (S2 - 1) / 2;
  __shared__ float s_v[KNN_LDS_PIX];
  __shared__ int s_l[KNN_LDS_PIX];
  __shared__ int s_wbox[4][4];          // per wave: min x, max x, min y, max y (no initialisation, no atomics: one barrier less)
  // frame of this workgroup: frame b owns ceil(n_b / 256) consecutive workgroups.  B is small; the walk has no early exit so
  // that its scalar loads are independent of each other (one memory latency, not one per frame).  (Measured: indexing the
  // concatenated list directly, so that the point loads do not wait for this walk, is SLOWER -- 13.2 vs 11.7 us -- the
  // per-thread frame search and the mixed-frame handling cost more than the overlap gives.)
  int b = -1, wg = 0;
  int64_t lo = 0, hi = 0;
  if (!offsets) { b = 0; wg = (int)blockIdx.x; hi = P1; }        // one frame of P1 points (pmf_knn_vote): no table
  else {
    int first = 0;
    int64_t o0 = offsets[0];
    for (int k = 0; k < B; ++k) {
      const int64_t o1 = offsets[k + 1];
      const int nb = (int)((o1 - o0 + 255) >> 8);
      if (b < 0 && (int)blockIdx.x < first + nb) { b = k; wg = (int)blockIdx.x - first; lo = o0; hi = o1; }
      first += nb;
      o0 = o1;
    }
  }
  if (b < 0) return;                                    // (the grid is an upper bound)
  const int64_t i = lo + (int64_t)wg * 256 + threadIdx.x;
  const bool valid = i < hi;
  const float* __restrict__ prb = pr + (size_t)b * H * W;
  const int64_t* __restrict__ amb = am + (size_t)b * H * W;
  const int32_t* __restrict__ amb32 = am32 ? am32 + (size_t)b * H * W : nullptr;
  int cx = 0, cy = 0;
  float r = 0.f;
  if (valid) { cx = (int)px[i]; cy = (int)py[i]; r = ur[i]; }
  {
    // bounding box: butterfly over the wave, one LDS atomic per wave and bound.  (Points far outside the image only enlarge
    // the box: it then exceeds the LDS budget and the global path, which clips tap by tap, takes over.)
    int mnx = valid ? cx : 0x7fffffff, mxx = valid ? cx : -0x7fffffff, mny = valid ? cy : 0x7fffffff, mxy = valid ? cy : -0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mnx = min(mnx, __shfl_xor(mnx, o)); mxx = max(mxx, __shfl_xor(mxx, o));
      mny = min(mny, __shfl_xor(mny, o)); mxy = max(mxy, __shfl_xor(mxy, o));
    }
    if ((threadIdx.x & 63) == 0) {
      int* wb = s_wbox[threadIdx.x >> 6];
      wb[0] = mnx; wb[1] = mxx; wb[2] = mny; wb[3] = mxy;
    }
  }
  __syncthreads();
  const int bx0 = min(min(s_wbox[0][0], s_wbox[1][0]), min(s_wbox[2][0], s_wbox[3][0]));
  const int bx1 = max(max(s_wbox[0][1], s_wbox[1][1]), max(s_wbox[2][1], s_wbox[3][1]));
  const int by0 = min(min(s_wbox[0][2], s_wbox[1][2]), min(s_wbox[2][2], s_wbox[3][2]));
  const int by1 = max(max(s_wbox[0][3], s_wbox[1][3]), max(s_wbox[2][3], s_wbox[3][3]));
  const int x0 = bx0 - PAD, y0 = by0 - PAD;
  const long bw = (long)bx1 - bx0 + 1 + 2 * PAD, bh = (long)by1 - by0 + 1 + 2 * PAD;
  const bool staged = bw > 0 && bh > 0 && bw * bh <= KNN_LDS_PIX;
  if (staged) {
    // all loads of the box first (up to 16 pixels per thread, independent), then the LDS stores: one memory latency
    constexpr int PER = KNN_LDS_PIX / 256;
    const int n = (int)(bw * bh), w_ = (int)bw;
    const float inv_w = 1.f / (float)w_;
    float vv[PER];
    int ll[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int j = threadIdx.x + u * 256;
      vv[u] = 0.f; ll[u] = 0;   // F.unfold zero padding: range 0, label 0
      if (j < n) {
        int yy = (int)(((float)j + 0.5f) * inv_w);       // j < 4096, w_ >= S: exact up to one step, corrected below
        int xx = j - yy * w_;
        if (xx < 0) { --yy; xx += w_; } else if (xx >= w_) { ++yy; xx -= w_; }
        const int y = y0 + yy, x = x0 + xx;
        if (y >= 0 && y < H && x >= 0 && x < W) {
          vv[u] = prb[(size_t)y * W + x];
          ll[u] = amb32 ? amb32[(size_t)y * W + x] : (int)amb[(size_t)y * W + x];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int j = threadIdx.x + u * 256;
      if (j < n) { s_v[j] = vv[u] < 0.f ? INFINITY : vv[u]; s_l[j] = ll[u]; }
    }
  }
  __syncthreads();
  if (!valid) return;
  float dist[S2];
  int lab[S2];
  if (staged) {
    const int w_ = (int)bw, base = (cy - PAD - y0) * w_ + (cx - PAD - x0);
#pragma unroll
    for (int t = 0; t < S2; ++t) {
      const int j = base + (t / S) * w_ + (t % S);
      float v = s_v[j];
      if (t == CENTER) v = r;
      dist[t] = fabsf(v - r) * invg[t];
      lab[t] = s_l[j];
    }
  } else {
#pragma unroll
    for (int t = 0; t < S2; ++t) {
      const int y = cy + t / S - PAD, x = cx + t % S - PAD;
      float v = 0.f;
      int l = 0;
      if (y >= 0 && y < H && x >= 0 && x < W) {
        v = prb[(size_t)y * W + x];
        l = amb32 ? amb32[(size_t)y * W + x] : (int)amb[(size_t)y * W + x];
        if (v < 0.f) v = INFINITY;
      }
      if (t == CENTER) v = r;
      dist[t] = fabsf(v - r) * invg[t];
      lab[t] = l;
    }
  }
  // k x first-minimum selection (ties -> smaller window index), the same rule as knn_k with a 32-bit taken mask (S2 <= 25 on
  // this path) and the label carried along the scan
  static_assert(S2 <= 32, "taken mask");
  unsigned used = 0u;
  int sel[8];
  int nsel = knn < 8 ? knn : 8;
  for (int k = 0; k < nsel; ++k) {
    float best = 0.f;
    int bi = -1, l = 0;
#pragma unroll
    for (int t = 0; t < S2; ++t) {
      const bool take = !(used & (1u << t)) && (bi < 0 || dist[t] < best);
      best = take ? dist[t] : best; l = take ? lab[t] : l; bi = take ? t : bi;
    }
    used |= 1u << bi;
    if (cutoff > 0.f && best > cutoff) l = nclasses;
    sel[k] = l;
  }
  int best_cnt = 0, best_cls = 1;
  for (int a = 0; a < nsel; ++a) {
    const int cls = sel[a];
    if (cls < 1 || cls >= nclasses) continue;
    int cnt = 0;
    for (int c = 0; c < nsel; ++c) cnt += sel[c] == cls;
    if (cnt > best_cnt || (cnt == best_cnt && cls < best_cls)) { best_cnt = cnt; best_cls = cls; }
  }
  labels[i] = best_cls;
}

static int knn_vote_batch_impl(const float* proj_range, const float* unproj_range, const int64_t* proj_argmax,
                               const int32_t* am32, const int64_t* px, const int64_t* py, const int64_t* offsets, int32_t B,
                               int32_t H, int32_t W, int64_t P_total, int32_t knn, int32_t search, const float* inv_gauss,
                               float cutoff, int32_t nclasses, int64_t* labels, pmf_stream_t s) {
  if (search % 2 == 0) return PMF_E_ARG;
  if (B < 1 || B > 1024 || !offsets) return PMF_E_ARG;
  if (knn < 1 || knn > 8 || knn > search * search) return PMF_E_UNSUPPORTED;
  if (P_total <= 0) return 0;
  hipStream_t st = (hipStream_t)s;
  constexpr bool no_lds = false;
  if (search < 1 || search > 255) return PMF_E_ARG;
  if (!no_lds && (search == 3 || search == 5)) {       // (7x7: 49 + 49 window registers next to 32 KB of LDS -- stays on the gather form)
    const dim3 g2((unsigned)(cdiv64(P_total, 256) + B)), b2(256);      // sum_b ceil(n_b / 256) <= ceil(P / 256) + B
    if (search == 3) hipLaunchKernelGGL(knn_batch_lds_k<3>, g2, b2, 0, st, proj_range, unproj_range, proj_argmax, px, py, offsets, B, H, W, (int64_t)0, knn, inv_gauss, cutoff, nclasses, labels, am32);
    else hipLaunchKernelGGL(knn_batch_lds_k<5>, g2, b2, 0, st, proj_range, unproj_range, proj_argmax, px, py, offsets, B, H, W, (int64_t)0, knn, inv_gauss, cutoff, nclasses, labels, am32);
    PMF_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((unsigned)cdiv64(P_total, 64)), block(64);
  switch (search) {
    case 3: hipLaunchKernelGGL(knn_batch_k<3>, grid, block, 0, st, proj_range, unproj_range, proj_argmax, px, py, offsets, B, H, W, P_total, knn, inv_gauss, cutoff, nclasses, labels, am32); break;
    case 5: hipLaunchKernelGGL(knn_batch_k<5>, grid, block, 0, st, proj_range, unproj_range, proj_argmax, px, py, offsets, B, H, W, P_total, knn, inv_gauss, cutoff, nclasses, labels, am32); break;
    case 7: hipLaunchKernelGGL(knn_batch_k<7>, grid, block, 0, st, proj_range, unproj_range, proj_argmax, px, py, offsets, B, H, W, P_total, knn, inv_gauss, cutoff, nclasses, labels, am32); break;
    default:
      hipLaunchKernelGGL(knn_any_k, grid, block, 0, st, proj_range, unproj_range, proj_argmax, px, py, offsets, B, H, W, P_total, knn, search, inv_gauss, cutoff, nclasses, labels, am32);
  }
  PMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pmf_knn_vote_batch(const float* proj_range, const float* unproj_range, const int64_t* proj_argmax,
                                  const int64_t* px, const int64_t* py, const int64_t* offsets, int32_t B, int32_t H,
                                  int32_t W, int64_t P_total, int32_t knn, int32_t search, const float* inv_gauss,
                                  float cutoff, int32_t nclasses, int64_t* labels, pmf_stream_t s) {
  if (!proj_argmax) return PMF_E_ARG;
  return knn_vote_batch_impl(proj_range, unproj_range, proj_argmax, nullptr, px, py, offsets, B, H, W, P_total, knn, search,
                             inv_gauss, cutoff, nclasses, labels, s);
}

// ---- the vote straight from the network's probability maps (tasks/pmf_eval_semantickitti/infer.py:96-112: the reference takes
// torch's argmax over the class axis -- an int64 [B, H, W] map -- and hands it to KNN) -------------------------------------------
// Launch 1: channel argmax of the NCHW probabilities into an int32 label map (four pixels per lane, class planes read as
// 16-byte vectors: 4 C H W bytes in, 4 H W out); ties go to the lowest class and a NaN wins, as torch.argmax decides.
// Launch 2: the vote above on that map.  Fusing the argmax INTO the vote would recompute it per window: in sweep order a
// workgroup's bounding box is ~9 x 68 pixels for 256 points, i.e. 2.2 class-axis scans per image pixel on a 120 k-point sweep.
__global__ __launch_bounds__(256) void argmax_nchw_k(const float* __restrict__ prob, int C, int64_t HW, int64_t total4,
                                                     int32_t* __restrict__ out) {
  const int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (q >= total4) return;
  const int64_t per = (HW + 3) >> 2, b = q / per, p0 = (q - b * per) * 4;
  const float* base = prob + (size_t)b * C * HW + p0;
  const bool vec = p0 + 3 < HW && (HW & 3) == 0;
  float best[4];
  int bi[4] = {0, 0, 0, 0};
  for (int c = 0; c < C; ++c) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      const f32x4 t = *(const f32x4*)(base + (size_t)c * HW);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      for (int k = 0; k < 4; ++k) if (p0 + k < HW) v[k] = base[(size_t)c * HW + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool take = c == 0 || v[k] > best[k] || (v[k] != v[k] && best[k] == best[k]);
      best[k] = take ? v[k] : best[k];
      bi[k] = take ? c : bi[k];
    }
  }
  for (int k = 0; k < 4; ++k) if (p0 + k < HW) out[(size_t)b * HW + p0 + k] = bi[k];
}
extern "C" int pmf_knn_vote_batch_prob(const float* proj_range, const float* unproj_range, const float* prob_nchw,
                                       const int64_t* px, const int64_t* py, const int64_t* offsets, int32_t B, int32_t H,
                                       int32_t W, int64_t P_total, int32_t knn, int32_t search, const float* inv_gauss,
                                       float cutoff, int32_t nclasses, int32_t* argmax_ws, int64_t* labels, pmf_stream_t s) {
  if (!prob_nchw || !argmax_ws || B < 1 || H < 1 || W < 1 || nclasses < 1) return PMF_E_ARG;
  if (search % 2 == 0) return PMF_E_ARG;
  const int64_t HW = (int64_t)H * W, total4 = (int64_t)B * ((HW + 3) >> 2);
  hipLaunchKernelGGL(argmax_nchw_k, dim3((unsigned)cdiv64(total4, 256)), dim3(256), 0, (hipStream_t)s, prob_nchw, nclasses, HW,
                     total4, argmax_ws);
  PMF_LAUNCH_CHECK();
  return knn_vote_batch_impl(proj_range, unproj_range, nullptr, argmax_ws, px, py, offsets, B, H, W, P_total, knn, search,
                             inv_gauss, cutoff, nclasses, labels, s);
}

// ---- multi-camera merge (tasks/pmf_eval_nuscenes/infer.py:18-38 getMergePred) --------------------------------------
// Per LiDAR point the prediction of the camera with the highest confidence; a camera that does not see the point
// counts as confidence 0 / label -1, torch.argmax breaks ties towards the FIRST camera.  One 64-bit atomicMax per
// (camera, visible point) on key = confidence bits << 32 | (n_cams-1-camera) << 16 | (label+1): confidences are
// probabilities (>= 0), so the float bit pattern orders like the value; every key starts as "camera 0, absent".
// The reference fills a [6, P] table and then walks the P points in a Python loop.
__global__ __launch_bounds__(256) void merge_scatter_k(const int64_t* __restrict__ point_idx, const float* __restrict__ conf,
                                                       const int64_t* __restrict__ label, int64_t n, int cam, int n_cams,
                                                       int64_t pc_size, unsigned long long* __restrict__ keys) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= n) return;
  const int64_t p = point_idx[i];
  if (p < 0 || p >= pc_size) return;
  const unsigned long long key = ((unsigned long long)__float_as_uint(conf[i]) << 32) |
                                 ((unsigned long long)(n_cams - 1 - cam) << 16) | (unsigned long long)((label[i] + 1) & 0xffff);
  atomicMax(keys + p, key);
}
__global__ __launch_bounds__(256) void merge_init_k(unsigned long long* __restrict__ keys, int64_t pc_size, int n_cams) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i < pc_size) keys[i] = (unsigned long long)(n_cams - 1) << 16;
}
// fallback (optional): label of a LiDAR-only model for every point; it replaces the -1 of points no camera sees
// (more_experiment_config.md:10: "For LiDAR points that are outside the camera view, we use predictions of SalsaNext")
__global__ __launch_bounds__(256) void merge_final_k(const unsigned long long* __restrict__ keys, int64_t pc_size,
                                                     const int64_t* __restrict__ fallback, int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= pc_size) return;
  const int64_t lab = (int64_t)(keys[i] & 0xffffull) - 1;
  out[i] = (lab < 0 && fallback) ? fallback[i] : lab;
}

static int merge_impl(int32_t n_cams, const int64_t* const* point_idx, const float* const* conf,
                      const int64_t* const* label, const int64_t* counts, int64_t pc_size, const int64_t* fallback,
                      uint64_t* keys, int64_t* merged, pmf_stream_t s) {
  if (n_cams < 1 || n_cams > 255 || pc_size < 0 || !counts || (pc_size > 0 && (!keys || !merged))) return PMF_E_ARG;
  for (int j = 0; j < n_cams; ++j)
    if (counts[j] < 0 || (counts[j] > 0 && (!point_idx || !conf || !label || !point_idx[j] || !conf[j] || !label[j])))
      return PMF_E_ARG;
  if (pc_size == 0) return 0;
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(merge_init_k, dim3((unsigned)cdiv64(pc_size, 256)), dim3(256), 0, st, (unsigned long long*)keys,
                     pc_size, n_cams);
  for (int j = 0; j < n_cams; ++j)
    if (counts[j] > 0)
      hipLaunchKernelGGL(merge_scatter_k, dim3((unsigned)cdiv64(counts[j], 256)), dim3(256), 0, st, point_idx[j], conf[j],
                         label[j], counts[j], j, n_cams, pc_size, (unsigned long long*)keys);
  hipLaunchKernelGGL(merge_final_k, dim3((unsigned)cdiv64(pc_size, 256)), dim3(256), 0, st,
                     (const unsigned long long*)keys, pc_size, fallback, merged);
  PMF_LAUNCH_CHECK();
  return 0;
}
extern "C" int pmf_merge_pred(int32_t n_cams, const int64_t* const* point_idx, const float* const* conf,
                              const int64_t* const* label, const int64_t* counts, int64_t pc_size, uint64_t* keys,
                              int64_t* merged, pmf_stream_t s) {
  return merge_impl(n_cams, point_idx, conf, label, counts, pc_size, nullptr, keys, merged, s);
}
extern "C" int pmf_merge_pred_fallback(int32_t n_cams, const int64_t* const* point_idx, const float* const* conf,
                                       const int64_t* const* label, const int64_t* counts, int64_t pc_size,
                                       const int64_t* fallback, uint64_t* keys, int64_t* merged, pmf_stream_t s) {
  if (pc_size > 0 && !fallback) return PMF_E_ARG;
  return merge_impl(n_cams, point_idx, conf, label, counts, pc_size, fallback, keys, merged, s);
}
