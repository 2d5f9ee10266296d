// Pieces shared by the convolution kernels (conv_fwd.hip, conv_ps.hip): phase tracing, the kernel-argument batch
// fetch, and the epilogue of conv_fwd_k-shaped kernels (4 waves, wave w owns 32-pixel segments [w MT, w MT + MT) x BN
// output channels as MT x BN/32 accumulators of v_mfma 32x32).
#pragma once
#include "common.h"

// Phase tracing (tools/trace_conv.py builds a private copy of this file with -DPMF_CONV_TRACE): thread 0 of every
// workgroup stamps s_memtime at phase boundaries.  Compiled out of libpmf_amd.so.
#ifdef PMF_CONV_TRACE
#ifndef PMF_TRACE_SLOTS
#define PMF_TRACE_SLOTS 64
#endif
static __device__ unsigned long long* pmf_trace_buf = nullptr;
#define TR()                                                                                            \
  do {                                                                                                  \
    if (threadIdx.x == 0 && pmf_trace_buf && tri_ < PMF_TRACE_SLOTS - 5)                                                 \
      pmf_trace_buf[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * PMF_TRACE_SLOTS + tri_++] = \
          __builtin_amdgcn_s_memtime();                                                                 \
  } while (0)
#define TR_END()                                                                                        \
  do {                                                                                                  \
    if (threadIdx.x == 0 && pmf_trace_buf) {                                                            \
      unsigned long long* t_ = pmf_trace_buf + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * PMF_TRACE_SLOTS; \
      t_[PMF_TRACE_SLOTS - 5] = tr_w0_;                                                                                  \
      t_[PMF_TRACE_SLOTS - 4] = wall_clock64();                                                                          \
      t_[PMF_TRACE_SLOTS - 3] = __builtin_amdgcn_s_getreg(63508);                                                        \
      t_[PMF_TRACE_SLOTS - 2] = __builtin_amdgcn_s_getreg(63492);                                                        \
      t_[PMF_TRACE_SLOTS - 1] = tri_;                                                                                    \
    }                                                                                                   \
  } while (0)
#define TR_START() const unsigned long long tr_w0_ = wall_clock64()
#else
#define TR_START() do { } while (0)
#define TR() do { } while (0)
#define TR_END() do { } while (0)
#endif

// Kernel-argument fields travel through the scalar cache.  Left alone, hipcc sinks every s_load to its first use and
// waits for it there: 9 tap offsets + 6 operand fields were 15 dependent scalar-cache round trips (~5.5k cycles, 2.6 us)
// at the start of EVERY workgroup, before its first global load was issued.  Naming the values in one empty asm
// statement makes the compiler fetch them as one batch with a single wait.
#define PMF_SGPR_BATCH(...) asm volatile("" ::__VA_ARGS__)

// NCO > 1 (conv_ws.hip, the N-split kernel): the workgroup's four waves own NCO DIFFERENT 32-channel output tiles (wave w:
// tile w % NCO, n0 = that tile's first channel, BN = 32) -- only the statistics reduction differs: the partial sums of a
// tile come from the 4 / NCO waves that share it.
template <int BN, int MT, int NCO = 1>
__device__ __forceinline__ void conv_epilogue(const pmf_conv_desc_t& d, const ConvGeom& g, f32x16 (&acc)[MT][BN / 32],
                                              const int (&segrow)[MT], const int (&segcol)[MT], int n, int n0, int ks,
                                              int oy0, int ox0, int tile, float* smem, int& tri_
#ifdef PMF_CONV_TRACE
                                              , unsigned long long tr_w0_
#endif
                                              ) {
  constexpr int NT = BN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // ---- epilogue: branch-free.  Every element is one buffer store (buffer loads for the ReLU mask / accumulate) at
  // a 32-bit byte offset; masked-off elements get offset 0xffffffff, which the hardware range check drops (host
  // guarantees all tensors < 2 GiB).  The loads of a 16-row group are issued together, not load -> wait -> store.
  TR();
  if (g.ksplit > 1) {   // raw partial sums -> slab ks; bias / activation / statistics happen in conv_finish_k
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)g.ws, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int co = n0 + u * 32 + li;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oy = oy0 + segrow[m], oxb = ox0 + segcol[m] * 32 + 4 * lh;
        const bool rok = co < g.ws_ld && oy < d.OH;
        const int base = ((((ks * d.N + n) * d.OH + oy) * d.OW + oxb) * g.ws_ld + co) * 4, estep = g.ws_ld * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dx = (r & 3) + 8 * (r >> 2);
          const unsigned off = (rok && oxb + dx < d.OW) ? (unsigned)(base + dx * estep) : 0xffffffffu;
          float v = acc[m][u][r];
          asm volatile("" : "+v"(v));   // hipcc (ROCm 7.2) otherwise stores element 0 of each accumulator quad 4x
          // (in-kernel combine: write-through `sc1` stores -- MI355X_MICROARCH.md "publish-large": a release fence instead
          // writes the whole dirty L2 back, 8.2 vs 3.0 us per 64 KB slab)
          if (g.tickets) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), wr, off, 0, 16);
          else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), wr, off, 0, 0);
        }
      }
    }
    if (!g.tickets) {
      TR();
      TR_END();
      return;
    }
    // ---- in-kernel combine (round 6; MI355X_MICROARCH.md "splitk-seam"): every thread releases its slab stores at agent scope,
    // thread 0 draws the output tile's ticket (atomicInc wraps at ksplit - 1: the counter is back at 0 when the last
    // workgroup has drawn), and the LAST workgroup to arrive re-reads all ksplit slabs of the tile in slab order -- its own
    // included, so that the sum does not depend on who arrived last -- and falls through into the ordinary epilogue.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through stores have landed
    __syncthreads();
    unsigned* tk = (unsigned*)smem;
    if (tid == 0) {
      // (`tile` is the kernel's XCD-aware tile index, not blockIdx.x)
      const int co_tiles = (int)gridDim.y / g.ksplit;
      tk[0] = atomicInc(g.tickets + ((size_t)n * (g.tiles_x * g.tiles_y) + tile) * co_tiles + n0 / BN, (unsigned)(g.ksplit - 1));
    }
    __syncthreads();
    const bool last = tk[0] == (unsigned)(g.ksplit - 1);
    __syncthreads();                                   // (smem is reused by the statistics reduction below)
    if (!last) {
      TR();
      TR_END();
      return;
    }
    // (the slabs are read with `sc1` loads: they come from memory, not from a stale line of this XCD's L2)
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][u][r] = 0.f;
    for (int s = 0; s < g.ksplit; ++s) {
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int co = n0 + u * 32 + li;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int oy = oy0 + segrow[m], oxb = ox0 + segcol[m] * 32 + 4 * lh;
          const bool rok = co < g.ws_ld && oy < d.OH;
          const int base = ((((s * d.N + n) * d.OH + oy) * d.OW + oxb) * g.ws_ld + co) * 4, estep = g.ws_ld * 4;
          float t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dx = (r & 3) + 8 * (r >> 2);
            const unsigned off = (rok && oxb + dx < d.OW) ? (unsigned)(base + dx * estep) : 0xffffffffu;
            t[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, off, 0, 16));
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][u][r] += t[r];
        }
      }
    }
  }
  // BatchNorm statistics in float64: float*float is exact in double, so var = E[x^2] - mean^2 keeps full
  // float32 accuracy even for nearly-constant channels (the classic cancellation), at ~2 DP ops per output
  double ssum[NT], ssq[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) ssum[u] = ssq[u] = 0.0;
  PMF_SGPR_BATCH("s"(d.out), "s"(d.bias), "s"(d.act), "s"(d.Cout), "s"(d.out_ldc), "s"(d.out_H), "s"(d.out_W), "s"(d.out_sy),
                 "s"(d.out_sx), "s"(d.out_oy), "s"(d.out_ox), "s"(d.accumulate), "s"(d.ep_cmul), "s"(d.ep_cmul_ld),
                 "s"(d.ep_relu_x), "s"(d.ep_relu_scale), "s"(d.ep_relu_shift), "s"(d.ep_relu_ldc), "s"(d.stats),
                 "s"(d.ep_pmask), "s"(d.ep_flags), "s"(d.ep_stat_mean), "s"(d.ndst));
  // the destination of this workgroup's output channels: the descriptor's own fields, or (ndst > 0) the entry of dst[]
  // whose channel range holds n0; channel indices below are relative to the destination (n0e), the bias index is not
  float* e_out = d.out;
  const float* e_cmul = d.ep_cmul;
  const float* e_rx = d.ep_relu_x;
  const float* e_rsc = d.ep_relu_scale;
  const float* e_rsh = d.ep_relu_shift;
  double* e_stats = d.stats;
  const float* e_smean = d.ep_stat_mean;
  int e_Cout = d.Cout, e_ldc = d.out_ldc, e_acc = d.accumulate, e_cmul_ld = d.ep_cmul_ld, e_rldc = d.ep_relu_ldc,
      e_flags = d.ep_flags, n0e = n0;
  if (d.ndst > 0) {
    int k = 0, c0 = 0;
    while (k + 1 < d.ndst && n0 >= c0 + d.dst[k].C) { c0 += d.dst[k].C; ++k; }
    const pmf_conv_dst_t& t = d.dst[k];
    e_out = t.out; e_cmul = t.ep_cmul; e_rx = t.ep_relu_x; e_rsc = t.ep_relu_scale; e_rsh = t.ep_relu_shift;
    e_stats = t.stats; e_smean = t.ep_stat_mean;
    e_Cout = t.C; e_ldc = t.out_ldc; e_acc = t.accumulate; e_cmul_ld = t.ep_cmul_ld; e_rldc = t.ep_relu_ldc;
    e_flags = t.ep_flags; n0e = n0 - c0;
  }
  {
    const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)e_out, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrr =
        __builtin_amdgcn_make_buffer_rsrc((void*)e_rx, 0, e_rx ? 0x7fffffff : 0, 0x00020000);
    const float slope = d.act == PMF_ACT_LRELU ? 0.01f : (d.act == PMF_ACT_RELU ? 0.f : 1.f);
    const bool sig = d.act == PMF_ACT_SIGMOID, has_rx = e_rx != nullptr, accum = e_acc != 0;
    const bool want_stats = e_stats != nullptr;
    // BatchNorm-backward reduction riding on the last input-gradient launch into a gradient map: second column
    // sum v*(x - mean) instead of sum v^2 (x = the BN input, the same tensor the ReLU mask reads when there is one)
    const bool stat_bwd = e_smean != nullptr, x_only = (e_flags & PMF_EP_STAT_X_ONLY) != 0;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int co = n0e + u * 32 + li;
      const bool cok = co < e_Cout;
      const float bias = (cok && d.bias) ? d.bias[n0 + u * 32 + li] : 0.f;
      const float ecm = (cok && e_cmul) ? e_cmul[(size_t)n * e_cmul_ld + co] : 1.f;
      const float smu = (cok && stat_bwd) ? e_smean[co] : 0.f;
      float rs = 1.f, rt = 0.f;
      if (cok && has_rx && e_rsc) { rs = e_rsc[co]; rt = e_rsh[co]; }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oy = oy0 + segrow[m], oxb = ox0 + segcol[m] * 32 + 4 * lh;
        const bool rok = cok && oy < d.OH;
        const int pix0 = (n * d.out_H + oy * d.out_sy + d.out_oy) * d.out_W + oxb * d.out_sx + d.out_ox;
        const int obase = (pix0 * e_ldc + co) * 4, ostep = d.out_sx * e_ldc * 4;
        const int xbase = (pix0 * e_rldc + co) * 4, xstep = d.out_sx * e_rldc * 4;
        unsigned off[16];
        float xr[16], old[16], pm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dx = (r & 3) + 8 * (r >> 2);
          const bool ok = rok && oxb + dx < d.OW;
          off[r] = ok ? (unsigned)(obase + dx * ostep) : 0xffffffffu;
          xr[r] = 1.f; old[r] = 0.f; pm[r] = 1.f;
        }
        if (d.ep_pmask) {   // per-pixel multiplier (EPMF dilated validity mask): one value per output pixel
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dx = (r & 3) + 8 * (r >> 2);
            if (off[r] != 0xffffffffu) pm[r] = d.ep_pmask[pix0 + dx * d.out_sx];
          }
        }
        if (has_rx) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dx = (r & 3) + 8 * (r >> 2);
            const unsigned xo = off[r] == 0xffffffffu ? 0xffffffffu : (unsigned)(xbase + dx * xstep);
            xr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrr, xo, 0, 0));
          }
        }
        if (accum) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(orr, off[r], 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[m][u][r] + bias;
          if (sig) v = 1.f / (1.f + __expf(-v));
          else v = v > 0.f ? v : v * slope;
          v *= ecm * pm[r];
          if (!(xr[r] * rs + rt > 0.f) && !x_only) v = 0.f;
          v += old[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orr, off[r], 0, 0);
          if (want_stats && off[r] != 0xffffffffu) {
            ssum[u] += (double)v;
            ssq[u] += (double)v * (double)(stat_bwd ? xr[r] - smu : v);
          }
        }
      }
    }
  }
  TR();
  if constexpr (NCO > 1) {
    static_assert(BN == 32, "N-split epilogue: one 32-channel tile per wave");
    // every wave leaves its (sum, second column) per channel; thread t < 32 NCO folds tile t / 32 over the waves that own it
    // (fixed order) and writes the row of THAT tile's destination (its own e_* describe another tile)
    const bool any_stats = d.ndst > 0 ? true : d.stats != nullptr;      // (uniform over the workgroup)
    if (any_stats) {
      __syncthreads();
      double* red = (double*)smem;  // [4 waves][32][2]
      double a = ssum[0] + __shfl_xor(ssum[0], 32);
      double b = ssq[0] + __shfl_xor(ssq[0], 32);
      if (lh == 0) { red[(wave * 32 + li) * 2 + 0] = a; red[(wave * 32 + li) * 2 + 1] = b; }
      __syncthreads();
      if (tid < 32 * NCO) {
        const int cwt = tid >> 5, l = tid & 31;
        const int n0w = n0 - (wave % NCO) * 32;                 // first channel of the workgroup
        const int n0t = n0w + cwt * 32;                         // first channel of tile cwt
        double* t_stats = d.stats;
        int t_Cout = d.Cout, t_n0e = n0t;
        if (d.ndst > 0) {
          int k = 0, c0 = 0;
          while (k + 1 < d.ndst && n0t >= c0 + d.dst[k].C) { c0 += d.dst[k].C; ++k; }
          t_stats = d.dst[k].stats; t_Cout = d.dst[k].C; t_n0e = n0t - c0;
        }
        const int co = t_n0e + l;
        if (t_stats && co < t_Cout) {
          double sa = 0.0, sb = 0.0;
#pragma unroll
          for (int w = 0; w < 4 / NCO; ++w) {
            sa += red[((cwt + w * NCO) * 32 + l) * 2 + 0];
            sb += red[((cwt + w * NCO) * 32 + l) * 2 + 1];
          }
          double* row = t_stats + ((size_t)tile + (size_t)gridDim.x * n) * 2 * t_Cout;
          row[co] = sa;
          row[t_Cout + co] = sb;
        }
      }
    }
    TR();
    TR_END();
    return;
  }
  if (e_stats) {
    __syncthreads();
    double* red = (double*)smem;  // [4 waves][NT][32][2]
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      double a = ssum[u] + __shfl_xor(ssum[u], 32);
      double b = ssq[u] + __shfl_xor(ssq[u], 32);
      if (lh == 0) {
        red[((wave * NT + u) * 32 + li) * 2 + 0] = a;
        red[((wave * NT + u) * 32 + li) * 2 + 1] = b;
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int u = tid >> 5, l = tid & 31, co = n0e + tid;
      if (co < e_Cout) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a += red[((w * NT + u) * 32 + l) * 2 + 0];
          b += red[((w * NT + u) * 32 + l) * 2 + 1];
        }
        // one partial row per (tile, sample): no atomics (contended f64 atomics cost ~80 us per launch); the
        // BatchNorm finalize kernel folds the rows in a fixed order (deterministic)
        double* row = e_stats + ((size_t)tile + (size_t)gridDim.x * n) * 2 * e_Cout;
        row[co] = a;
        row[e_Cout + co] = b;
      }
    }
  }
  TR();
  TR_END();
}
