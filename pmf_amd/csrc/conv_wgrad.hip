// Convolution weight gradient on fp32 MFMA (gfx950), NHWC, LDS halo staging, deterministic two-stage reduce.
//
// GEMM view per tap t:  dW_t[ci][co] = sum_pixels X_t[pixel][ci] * dz[pixel][co]   (M = 32 input channels,
// N = BN output channels, K = pixels).  A workgroup owns one 32-channel chunk of the (virtually concatenated)
// input, BN = 32/64/128 output channels and a batch of TB taps, and walks a strided list of 4x32-pixel tiles: per
// tile the input chunk INCLUDING ITS HALO is staged once in LDS (BatchNorm-apply / ReLU / Dropout2d multiplier
// folded into the load) next to the dz tile.  Two pixels per v_mfma_f32_32x32x2_f32:
//   A[i=ci=lane&31][k=lane>>5] = ds_read_b32 X[pixel+tap offset][ci]   (32 consecutive banks)
//   B[k][j=co=lane&31]         = ds_read_b32 dz[pixel][co]
// The TB x BN/32 output tiles are dealt to the four waves (see conv_wgrad_k); each wave sees every pixel, so its
// accumulators are complete and go straight to this workgroup's partial slab.  Stage 2 sums the slabs in a fixed
// order (deterministic) and writes the gradient directly in PyTorch's OIHW layout.
#include "common.h"
#include <stdlib.h>
#include <utility>

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) -- unrolled by the front end (a
// "#pragma unroll" the optimiser declines leaves the accumulator arrays dynamically indexed, i.e. in scratch memory)
template <class F, int... Is>
__device__ __forceinline__ void wg_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
  wg_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

#define WG_ROWS 4
#define WG_CI 32

struct WgGeom {
  int tiles_x, tiles_y, total_tiles;
  int in_rows, in_cols, dy_min, dx_min;
  int x_floats;
  int Ktot, Cout32;
  int nchunks;       // 32-channel chunks over all operands
  int co_tiles, tap_batches;
};

// one staged 4x32-pixel tile: NA accumulators (compile-time) share the B value of each pixel pair.
// Operands of pixel pair kp+1 are fetched from LDS BEFORE the MFMAs of pair kp are issued (explicit double
// buffer + sched_barrier), so the ~100-cycle ds_read latency hides under NA x 64 cycles of MFMA even with one
// wave per SIMD; left alone, hipcc emits read -> wait -> mfma triplets through a single temporary register.
template <int UPW, int NA, int BN, int IS = 1>
__device__ __forceinline__ void wg_tile(f32x16 (&acc)[UPW], const float* __restrict__ Xs, const float* __restrict__ zbase,
                                        const int (&toff)[UPW], int li, int lh, int in_cols, int row0 = 0, int row1 = WG_ROWS) {
  constexpr int is = IS;                                 // compile-time stride: the offsets fold into the ds_read
                                                         // (IS = 2, round 5: the stride-2 3x3 layers ran the scalar-guard loop)
  constexpr int xstep = 2 * is * WG_CI, zstep = 2 * BN;
#pragma unroll 1
  for (int row = row0; row < row1; ++row) {
    const float* xp = Xs + (row * is * in_cols) * WG_CI + li + lh * is * WG_CI;
    const float* zp = zbase + (row * 32) * BN + lh * BN;
    float ac[NA], an[NA], bc, bn;
    bc = zp[0];
#pragma unroll
    for (int j = 0; j < NA; ++j) ac[j] = xp[toff[j]];
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      if (kp < 15) {
        bn = zp[(kp + 1) * zstep];
#pragma unroll
        for (int j = 0; j < NA; ++j) an[j] = xp[(kp + 1) * xstep + toff[j]];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[j], bc, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bc = bn;
#pragma unroll
      for (int j = 0; j < NA; ++j) ac[j] = an[j];
    }
  }
}
// ragged tap batches (e.g. the last batch of the 49-tap stem): scalar guards
template <int UPW>
__device__ __forceinline__ void wg_tile_any(f32x16 (&acc)[UPW], const float* __restrict__ Xs,
                                            const float* __restrict__ zbase, const int (&toff)[UPW], int nact, int li,
                                            int lh, int is, int in_cols, int BN, int row0 = 0, int row1 = WG_ROWS) {
#pragma unroll 1
  for (int row = row0; row < row1; ++row) {
    const float* xrow = Xs + (row * is * in_cols) * WG_CI + li;
    const float* zrow = zbase + (row * 32) * BN;
#pragma unroll 1
    for (int kp = 0; kp < 16; ++kp) {
      const int px = 2 * kp + lh;
      const float b = zrow[px * BN];
      const float* xp = xrow + px * is * WG_CI;
#pragma unroll
      for (int j = 0; j < UPW; ++j)
        if (j < nact) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[toff[j]], b, acc[j], 0, 0, 0);
    }
  }
}

// Work decomposition inside a workgroup: the output slab [TB taps][32 ci][NT*32 co] is TB*NT MFMA tiles ("units",
// u = tap*NT + co_tile).  Wave w owns units w, w+4, w+8, ... and walks EVERY pixel of the staged tile, so its
// accumulators are final for the pixels it saw: no cross-wave reduction, and one staged tile feeds
// 64 pixel-pairs x UPW MFMAs per wave.  Because 4 % NT == 0 a wave's units share one co tile (one B read per
// pixel pair) and differ only in the tap (one A read each).
// Fewer units than waves (U = 1: the one-tap layers with <= 32 output channels -- logits, downCntx.s, dec.up1; U = 2): the
// waves split the ROWS of the staged tile instead (PG = 4 / U pixel groups, wave w = unit w % U of group w / U) and the
// groups are folded through LDS in a fixed order at the end -- round 4 left three of four waves without an MFMA there
// (logits 32 -> 20 at 64 x 2048: 73 us for 59 MB).
template <int TB, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_k(const pmf_wgrad_desc_t d, const WgGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BN = NT * 32;
  constexpr int U = TB * NT, UPW = (U + 3) / 4;
  constexpr int PG = U == 1 ? 4 : (U == 2 ? 2 : 1);             // pixel groups (rows of a tile dealt to the waves)
  float* __restrict__ Xs = smem;
  float* __restrict__ Zs = smem + g.x_floats;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = PG > 1 ? wave_id % U : wave_id;             // wave-uniform (SGPR): unit bookkeeping stays scalar
  const int pgrp = PG > 1 ? wave_id / U : 0;
  const int row0 = pgrp * (WG_ROWS / PG), row1 = row0 + WG_ROWS / PG;
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, chunk = blockIdx.y;
  const int cot = blockIdx.z % g.co_tiles, tb = blockIdx.z / g.co_tiles;
  const int co0 = cot * BN;
  const int t0 = tb * TB;
  const int nt = min(TB, d.ntaps - t0);
  const int is = d.in_stride;

  // locate the operand / channel offset of this chunk
  int si = 0, c0 = 0, k0 = 0;
  {
    int rem = chunk;
    for (;;) {
      const int nch = (d.src[si].C + WG_CI - 1) / WG_CI;
      if (rem < nch) { c0 = rem * WG_CI; k0 += c0; break; }
      rem -= nch; k0 += d.src[si].C; ++si;
    }
  }
  const float* __restrict__ sx = d.src[si].x;
  const float* __restrict__ sscale = d.src[si].scale;
  const float* __restrict__ sshift = d.src[si].shift;
  const int sC = d.src[si].C, sld = d.src[si].ldc, sflags = d.src[si].flags;
  const bool bc = (sflags & PMF_SRC_BCAST) != 0;
  const int sH = bc ? d.OH * is : d.src[si].H, sW = bc ? d.OW * is : d.src[si].W;
  const int kc = min(WG_CI, sC - c0);   // multiple of 8
  const int nq = kc >> 2;

  f32x16 acc[UPW];
#pragma unroll
  for (int j = 0; j < UPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int gy0 = d.gather ? (int)d.tdy[t0] : g.dy_min;
  const int gx0 = d.gather ? (int)d.tdx[t0] : g.dx_min;
  // this wave's units
  int toff[UPW];
  int nact = 0;                // active units form a prefix j < nact (taps grow with j)
  const int myc = wave % NT;   // shared co tile of all units of this wave
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    const int u = wave + 4 * j, t = u / NT;
    const bool ok = u < U && t < nt;
    nact += ok ? 1 : 0;
    const int tt = t0 + (ok ? t : 0);
    toff[j] = (((int)d.tdy[tt] - gy0) * g.in_cols + ((int)d.tdx[tt] - gx0)) * WG_CI;
  }

  for (int tile = split; tile < g.total_tiles; tile += d.nsplit) {
    const int tx = tile % g.tiles_x, ty = (tile / g.tiles_x) % g.tiles_y, n = tile / (g.tiles_x * g.tiles_y);
    const int oy0 = ty * WG_ROWS, ox0 = tx * 32;
    const float* __restrict__ scm = d.src[si].cmul ? d.src[si].cmul + (size_t)n * d.src[si].cmul_ld : nullptr;
    __syncthreads();
    {  // stage X: a batch of independent global loads first, transform + LDS write after (latency paid once per
       // batch); missing channels of a short chunk are zero-filled so stale LDS never reaches the MFMA
      constexpr int XB = 5;
      const int total = g.in_rows * g.in_cols * (WG_CI / 4);
      const int q = tid & 7, cch = c0 + q * 4;
      const bool qok = q < nq;
      f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f}, cm4 = {1.f, 1.f, 1.f, 1.f};
      if (qok && sscale) { sc4 = *(const f32x4*)(sscale + cch); sh4 = *(const f32x4*)(sshift + cch); }
      if (qok && scm) cm4 = *(const f32x4*)(scm + cch);
      for (int base = 0; base < total; base += 256 * XB) {
        f32x4 v[XB];
        bool ok[XB];
#pragma unroll
        for (int j = 0; j < XB; ++j) {
          const int f = base + tid + 256 * j;
          const int pix = f >> 3;
          const int r = pix / g.in_cols, c = pix - r * g.in_cols;
          const int iy = oy0 * is + gy0 + r, ix = ox0 * is + gx0 + c;
          ok[j] = f < total && qok && iy >= 0 && iy < sH && ix >= 0 && ix < sW;
          if (ok[j]) {
            const size_t off = bc ? (size_t)n * sld + cch : ((size_t)(n * sH + iy) * sW + ix) * sld + cch;
            v[j] = *(const f32x4*)(sx + off);
          }
        }
#pragma unroll
        for (int j = 0; j < XB; ++j) {
          const int f = base + tid + 256 * j;
          if (f < total) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (ok[j]) {
              t = v[j];
              if (sscale) t = t * sc4 + sh4;
              if (sflags & PMF_SRC_RELU) {
                t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
              }
              t = t * cm4;
            }
            *(f32x4*)(Xs + (f >> 3) * WG_CI + q * 4) = t;
          }
        }
      }
    }
    {  // stage dz tile [4*32 pixels][BN], batches of <= 8 independent loads
      constexpr int rq = BN / 4;
      constexpr int ZT = WG_ROWS * 32 * rq / 256, ZB = ZT < 4 ? ZT : 4;
#pragma unroll
      for (int zb = 0; zb < ZT; zb += ZB) {
        f32x4 v[ZB];
#pragma unroll
        for (int j = 0; j < ZB; ++j) {
          const int f = tid + 256 * (zb + j);
          const int pix = f / rq, qq = f - pix * rq;
          const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
          const int co = co0 + qq * 4;
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
          if (oy < d.OH && ox < d.OW && co < d.Cout) {
            const float* p = d.dz + ((size_t)(n * d.OH + oy) * d.OW + ox) * d.dz_ldc + co;
            if (co + 3 < d.Cout) t = *(const f32x4*)p;
            else { t.x = p[0]; if (co + 1 < d.Cout) t.y = p[1]; if (co + 2 < d.Cout) t.z = p[2]; }
          }
          v[j] = t;
        }
#pragma unroll
        for (int j = 0; j < ZB; ++j) {
          const int f = tid + 256 * (zb + j);
          const int pix = f / rq, qq = f - pix * rq;
          *(f32x4*)(Zs + pix * BN + qq * 4) = v[j];
        }
      }
    }
    __syncthreads();
    const float* zbase = Zs + myc * 32 + li;
    if (is == 1 && nact == UPW) wg_tile<UPW, UPW, BN>(acc, Xs, zbase, toff, li, lh, g.in_cols, row0, row1);
    else if (is == 1 && nact == UPW - 1 && UPW > 1)
      wg_tile<UPW, (UPW > 1 ? UPW - 1 : 1), BN>(acc, Xs, zbase, toff, li, lh, g.in_cols, row0, row1);
    else if (is == 2 && nact == UPW) wg_tile<UPW, UPW, BN, 2>(acc, Xs, zbase, toff, li, lh, g.in_cols, row0, row1);
    else if (is == 2 && nact == UPW - 1 && UPW > 1)
      wg_tile<UPW, (UPW > 1 ? UPW - 1 : 1), BN, 2>(acc, Xs, zbase, toff, li, lh, g.in_cols, row0, row1);
    else wg_tile_any<UPW>(acc, Xs, zbase, toff, nact, li, lh, is, g.in_cols, BN, row0, row1);
  }

  if constexpr (PG > 1) {       // fold the pixel groups in a fixed order: group 0 of every unit ends up with the sum
    static_assert(UPW == 1, "pixel groups: one unit per wave");
    __syncthreads();            // (everyone is through with the staged tile: the fold reuses Xs)
    float* red = smem;          // [PG - 1][U][16][64]
    if (pgrp > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(((pgrp - 1) * U + wave) * 16 + r) * 64 + lane] = acc[0][r];
    }
    __syncthreads();
    if (pgrp > 0) return;
#pragma unroll
    for (int p = 1; p < PG; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] += red[(((p - 1) * U + wave) * 16 + r) * 64 + lane];
  }

  // ---- every wave owns complete sums for its units: write them straight into the partial slab
  float* part = d.partial + (size_t)split * d.ntaps * g.Ktot * g.Cout32;
  const int co = co0 + myc * 32 + li;
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    if (j >= nact) continue;
    const int t = (wave + 4 * j) / NT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ci < kc && co < g.Cout32) part[((size_t)(t0 + t) * g.Ktot + k0 + ci) * g.Cout32 + co] = acc[j][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Software-pipelined variant for the common case (host-checked, wg_simple()): stride 1, halo staging, one tap batch
// (ntaps == TB), every operand a multiple of 32 channels with the gradient's H x W, OH % 4 == 0, OW % 32 == 0,
// Cout % BN == 0.
//   * dz needs no transform: it goes global -> LDS by LDS-DMA (1 KiB per wave instruction), in two 2-row halves
//     that ping-pong: while the MFMAs read half h the DMA fills the other half (of this tile or the next one);
//   * the input tile of the NEXT 4x32-pixel tile is fetched into registers (buffer loads, out-of-image pixels get a
//     negative offset = hardware zero fill) while this tile is multiplied; BatchNorm-apply / ReLU / mask + ds_write
//     happen between tiles;
//   * work split: wave w owns output-channel tile w % NT for ALL TB taps and every (4/NT)-th pixel pair, so the four
//     waves always carry TB accumulators each (the unit-dealing kernel above leaves 9 units on 4 waves as 3/2/2/2);
//     pixel groups are summed through LDS once, after the tile loop, in a fixed order.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int TB, int NT, int XSL>   // XSL: float4 slots per thread of the input tile (7: 6 x 34 pixels x 8, 9: 8 x 36)
__device__ __forceinline__ void conv_wgrad_pipe_body(const pmf_wgrad_desc_t& d, const WgGeom& g, float* __restrict__ smem) {
  constexpr int BN = NT * 32, PG = 4 / NT;
  constexpr int HPX = 64;                   // pixels of a half tile
  constexpr int ZPW = HPX * BN / 256 / 4;   // DMA instructions per wave per half (2, 4, 8)
  constexpr int PPI = 256 / BN;             // pixels one DMA instruction covers
  float* __restrict__ Xs = smem;
  float* __restrict__ Z0 = smem + g.x_floats;
  float* __restrict__ Z1 = Z0 + HPX * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, chunk = blockIdx.y;
  const int co0 = (int)blockIdx.z * BN;
  const int in_cols = g.in_cols;

  int si = 0, c0 = 0, k0 = 0;
  {
    int rem = chunk;
    for (;;) {
      const int nch = d.src[si].C / WG_CI;
      if (rem < nch) { c0 = rem * WG_CI; k0 += c0; break; }
      rem -= nch; k0 += d.src[si].C; ++si;
    }
  }
  const int sld = d.src[si].ldc, sflags = d.src[si].flags;
  const int sH = d.OH, sW = d.OW;
  const bool aff = d.src[si].scale != nullptr;
  const int q = tid & 7, cch = c0 + q * 4;
  const int totalX = g.in_rows * in_cols * 8;
  int rc[XSL];   // tile-relative (row << 8 | col) of every slot, -1 = unused
#pragma unroll
  for (int j = 0; j < XSL; ++j) {
    const int f = tid + 256 * j, pix = f >> 3;
    const int r = pix / in_cols, c = pix - r * in_cols;
    rc[j] = f < totalX ? (r << 8 | c) : -1;
  }
  int offZ[ZPW];   // per-lane source offset (floats) of DMA instruction wave + 4*jj relative to the half's first pixel
#pragma unroll
  for (int jj = 0; jj < ZPW; ++jj) {
    const int p = (wave + 4 * jj) * PPI + lane / (BN / 4);
    offZ[jj] = ((p >> 5) * d.OW + (p & 31)) * d.dz_ldc + (lane % (BN / 4)) * 4;
  }
  const int cot = wave % NT, pg = wave / NT;
  int toff[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j)
    toff[j] = (((int)d.tdy[j] - g.dy_min) * in_cols + ((int)d.tdx[j] - g.dx_min)) * WG_CI;

  f32x16 acc[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const __amdgpu_buffer_rsrc_t xrs_on =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, d.N * sH * sW * sld * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs_off = __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, 0, 0x00020000);
  f32x4 rX[XSL];
  unsigned okX = 0u;
  int ncur = 0;   // sample index of the tile held in rX
  const int tiles_per_n = g.tiles_x * g.tiles_y;
  auto fetch = [&](int tile, bool on) {   // input tile of `tile` -> registers; validity mask + (n,c) multiplier
    const int n = tile / tiles_per_n, rem = tile - n * tiles_per_n;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    const int by = ty * WG_ROWS + g.dy_min, bx = tx * 32 + g.dx_min;
    const __amdgpu_buffer_rsrc_t rs = on ? xrs_on : xrs_off;
    okX = 0u;
#pragma unroll
    for (int j = 0; j < XSL; ++j) {
      const int iy = by + (rc[j] >> 8), ix = bx + (rc[j] & 255);
      const bool ok = rc[j] >= 0 && iy >= 0 && iy < sH && ix >= 0 && ix < sW;
      const int gp = ok ? (n * sH + iy) * sW + ix : -1;   // negative byte offset -> out of range -> reads 0
      okX |= ok ? (1u << j) : 0u;
      rX[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (gp * sld + cch) * 4, 0, 0));
    }
    ncur = n;
  };
  auto zsrc = [&](int tile, int half) -> const float* {   // first dz element of a half tile (wave-uniform)
    const int n = tile / tiles_per_n, rem = tile - n * tiles_per_n;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    return d.dz + ((size_t)(n * d.OH + ty * WG_ROWS + 2 * half) * d.OW + tx * 32) * d.dz_ldc + co0;
  };
  auto dma = [&](const float* __restrict__ src, float* __restrict__ dst) {
#pragma unroll
    for (int jj = 0; jj < ZPW; ++jj)
      __builtin_amdgcn_global_load_lds(src + offZ[jj], (lds_ptr_t)(dst + (wave + 4 * jj) * 256), 16, 0, 0);
  };
  // one half tile: this wave's pixel pairs of two rows, TB MFMAs per pair, operands of the next pair prefetched
  auto half = [&](const float* __restrict__ Zh, int h) {
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const float* xp = Xs + ((2 * h + rr) * in_cols + lh) * WG_CI + li;
      const float* zp = Zh + (rr * 32 + lh) * BN + cot * 32 + li;
      constexpr int NP = 16 / PG;
      float ac[TB], an[TB], bc, bn;
      bc = zp[2 * pg * BN];
#pragma unroll
      for (int j = 0; j < TB; ++j) ac[j] = xp[2 * pg * WG_CI + toff[j]];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (i + 1 < NP) {
          const int kp = pg + (i + 1) * PG;
          bn = zp[2 * kp * BN];
#pragma unroll
          for (int j = 0; j < TB; ++j) an[j] = xp[2 * kp * WG_CI + toff[j]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[j], bc, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bc = bn;
#pragma unroll
        for (int j = 0; j < TB; ++j) ac[j] = an[j];
      }
    }
  };

  int tile = split;
  if (tile < g.total_tiles) {
    fetch(tile, true);
    dma(zsrc(tile, 0), Z0);
  }
  while (tile < g.total_tiles) {
    __syncthreads();                       // X: everyone finished the previous tile
    // channel transform of this operand: re-read here (L1/L2 hits) instead of pinning 12 VGPRs across the MFMAs
    f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f}, cm4 = {1.f, 1.f, 1.f, 1.f};
    if (aff) { sc4 = *(const f32x4*)(d.src[si].scale + cch); sh4 = *(const f32x4*)(d.src[si].shift + cch); }
    if (d.src[si].cmul) cm4 = *(const f32x4*)(d.src[si].cmul + (size_t)ncur * d.src[si].cmul_ld + cch);
#pragma unroll
    for (int j = 0; j < XSL; ++j) {
      if (rc[j] >= 0) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if ((okX >> j) & 1u) {
          t = rX[j];
          if (aff) t = t * sc4 + sh4;
          if (sflags & PMF_SRC_RELU) {
            t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
          }
          t = t * cm4;
        }
        *(f32x4*)(Xs + ((tid + 256 * j) >> 3) * WG_CI + q * 4) = t;
      }
    }
    const int next = tile + d.nsplit;
    const bool have = next < g.total_tiles;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of dz half 0 is in LDS
    __syncthreads();                       // Y: input tile + half 0 visible
    dma(zsrc(tile, 1), Z1);
    fetch(have ? next : tile, have);       // next input tile -> registers, lands under the MFMAs
    __builtin_amdgcn_sched_barrier(0);
    half(Z0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // Z: everyone finished half 0
    if (have) dma(zsrc(next, 0), Z0);
    __builtin_amdgcn_sched_barrier(0);
    half(Z1, 1);
    tile = next;
  }

  // ---- sum the pixel groups (fixed order) and write this workgroup's partial slab
  if (PG > 1) {
    float* red = smem;   // [4 waves][16][64]
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      __syncthreads();
      if (pg > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[j][r];
      }
      __syncthreads();
      if (pg == 0) {
#pragma unroll
        for (int p = 1; p < PG; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] += red[((wave + NT * p) * 16 + r) * 64 + lane];
      }
    }
  }
  if (pg == 0) {
    float* part = d.partial + (size_t)split * d.ntaps * g.Ktot * g.Cout32;
    const int co = co0 + cot * 32 + li;
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
        part[((size_t)j * g.Ktot + k0 + ci) * g.Cout32 + co] = acc[j][r];
      }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same pipelined kernel with the products on the bf16 matrix pipe (operands split three ways, six products per
// fp32 product: see conv_fwd.hip PIPE 5 for the arithmetic).  K of v_mfma_f32_32x32x16_bf16 = 16 consecutive pixels of
// one row; both operands are wanted pixel-major per channel while memory is channel-major per pixel:
//   * input tile: split while it is written to LDS as [pixel][plane][32 ci] bf16 (192 B per pixel: four consecutive
//     pixels land in four different 64-B bank quarters) and read with ds_read_b64_tr_b16, the hardware transpose read:
//     a 16-lane group fetches a [4 pixels][16 channels] block and every lane receives the 4 pixels of ITS channel --
//     two reads per plane make the 8-pixel A fragment of a tap (any tap offset: it only shifts the pixel index);
//   * dz keeps its LDS-DMA path (raw fp32 [pixel][32 co]); the wave that owns a 16-pixel slab reads its B fragment as
//     8 ds_read_b32 and splits it in registers -- once per slab, shared by all TB taps (54 MFMAs).
// Wave w owns slab w of every half tile (4 slabs of 16 pixels), TB accumulators as before.
// Phase tracing of conv_wgrad_s3_k (tools/trace_wgrad.py builds a private copy with -DPMF_WG_TRACE): thread 0 of every
// workgroup stamps s_memtime at phase boundaries.  Compiled out of libpmf_amd.so.
#ifdef PMF_WG_TRACE
__device__ unsigned long long* pmf_wg_trace_buf = nullptr;
extern "C" int pmf_wg_trace_set(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(pmf_wg_trace_buf), &p, sizeof(p)); }
#define WTR()                                                                                              \
  do {                                                                                                     \
    if (threadIdx.x == 0 && pmf_wg_trace_buf && wtri_ < 62)                                                \
      pmf_wg_trace_buf[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 64 + wtri_++] = \
          __builtin_amdgcn_s_memtime();                                                                    \
  } while (0)
#define WTR_END()                                                                                          \
  do {                                                                                                     \
    if (threadIdx.x == 0 && pmf_wg_trace_buf)                                                              \
      pmf_wg_trace_buf[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 64 + 63] = wtri_; \
  } while (0)
#else
#define WTR() do { } while (0)
#define WTR_END() do { } while (0)
#endif
typedef __attribute__((ext_vector_type(8))) __bf16 wbf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 wbf16x2;
typedef __attribute__((ext_vector_type(4))) short ws16x4;
typedef __attribute__((ext_vector_type(8))) short ws16x8;
typedef __attribute__((ext_vector_type(2))) float wf32x2;
typedef __attribute__((ext_vector_type(2))) unsigned wu32x2;
typedef __attribute__((ext_vector_type(4))) unsigned wu32x4;
#define WS3_XPB 192

__device__ __forceinline__ unsigned ws3_pk(wf32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wbf16x2)); }
__device__ __forceinline__ wf32x2 ws3_unpk(unsigned u) {
  return wf32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
__device__ __forceinline__ float ws3_vmax(float a, float b) {   // plain v_max_f32 (fmaxf adds a canonicalising v_max)
  float r;
  asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void ws3_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  wf32x2 r = {a, b};
  p0 = ws3_pk(r);
  r = r - ws3_unpk(p0);
  p1 = ws3_pk(r);
  r = r - ws3_unpk(p1);
  p2 = ws3_pk(r);
}
// the same with plain v_sub_f32: a packed-f32 instruction next to an MFMA stalls the wave (MI355X_MICROARCH: +26 cycles
// per pair), and this form runs between the MFMAs of conv_wgrad_s3_swp_body
__device__ __forceinline__ float ws3_sub(float a, float b) {
  float r;
  asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void ws3_split2_np(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = ws3_pk(wf32x2{a, b});
  a = ws3_sub(a, __builtin_bit_cast(float, p0 << 16)); b = ws3_sub(b, __builtin_bit_cast(float, p0 & 0xffff0000u));
  p1 = ws3_pk(wf32x2{a, b});
  a = ws3_sub(a, __builtin_bit_cast(float, p1 << 16)); b = ws3_sub(b, __builtin_bit_cast(float, p1 & 0xffff0000u));
  p2 = ws3_pk(wf32x2{a, b});
}
__device__ __forceinline__ float ws3_fma(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float ws3_mul(float a, float b) {
  float r;
  asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int TB, int XSL>
__device__ __forceinline__ void conv_wgrad_s3_body(const pmf_wgrad_desc_t& d, const WgGeom& g, float* __restrict__ smem) {
  constexpr int BN = 32;
  constexpr int HPX = 64;
  constexpr int ZPW = HPX * BN / 256 / 4;   // 2 DMA instructions per wave per half
  constexpr int PPI = 256 / BN;
  char* __restrict__ Xs = (char*)smem;
  float* __restrict__ Z0 = smem + g.x_floats;
  float* __restrict__ Z1 = Z0 + HPX * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, chunk = blockIdx.y;
  const int co0 = (int)blockIdx.z * BN;
  const int in_cols = g.in_cols;

  int wtri_ = 0;
  (void)wtri_;
  WTR();
  int si = 0, c0 = 0, k0 = 0;
  {
    int rem = chunk;
    for (;;) {
      const int nch = (d.src[si].C + WG_CI - 1) / WG_CI;     // (a 16-channel operand is one half-empty chunk)
      if (rem < nch) { c0 = rem * WG_CI; k0 += c0; break; }
      rem -= nch; k0 += d.src[si].C; ++si;
    }
  }
  const int sld = d.src[si].ldc, sflags = d.src[si].flags;
  const int sH = d.OH, sW = d.OW;
  const bool aff = d.src[si].scale != nullptr;
  const int q = tid & 7, cch = c0 + q * 4;
  const int kc = min(WG_CI, d.src[si].C - c0);      // channels of this chunk that exist
  const bool qok = q * 4 < kc;                      // this thread's four channels exist (else: zeros)
  const int totalX = g.in_rows * in_cols * 8;
  int rc[XSL];
#pragma unroll
  for (int j = 0; j < XSL; ++j) {
    const int f = tid + 256 * j, pix = f >> 3;
    const int r = pix / in_cols, c = pix - r * in_cols;
    rc[j] = f < totalX ? (r << 8 | c) : -1;
  }
  int offZ[ZPW];
#pragma unroll
  for (int jj = 0; jj < ZPW; ++jj) {
    const int p = (ZPW * wave + jj) * PPI + lane / (BN / 4);       // wave w DMAs the 16 pixels of ITS slab (it alone reads them)
    offZ[jj] = ((p >> 5) * d.OW + (p & 31)) * d.dz_ldc + (lane % (BN / 4)) * 4;
  }
  int toff[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j)
    toff[j] = (((int)d.tdy[j] - g.dy_min) * in_cols + ((int)d.tdx[j] - g.dx_min)) * WS3_XPB;
  // transpose-read addressing: lane p of a 16-lane group supplies [pixel p/4][channels 4 (p%4) ...]; groups 1 / 3 read
  // channels 16-31, groups 2 / 3 pixels 8-11 (the k = 8..15 half of the MFMA operand)
  const int trofs = (((lane >> 5) * 8 + ((lane & 15) >> 2)) * WS3_XPB) + ((((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);

  f32x16 acc[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const __amdgpu_buffer_rsrc_t xrs_on =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, d.N * sH * sW * sld * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs_off = __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, 0, 0x00020000);
  f32x4 rX[XSL];
  unsigned okX = 0u;
  int ncur = 0;
  const int tiles_per_n = g.tiles_x * g.tiles_y;
  auto fetch = [&](int tile, bool on) {
    const int n = tile / tiles_per_n, rem = tile - n * tiles_per_n;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    const int by = ty * WG_ROWS + g.dy_min, bx = tx * 32 + g.dx_min;
    const __amdgpu_buffer_rsrc_t rs = on ? xrs_on : xrs_off;
    okX = 0u;
#pragma unroll
    for (int j = 0; j < XSL; ++j) {
      const int iy = by + (rc[j] >> 8), ix = bx + (rc[j] & 255);
      const bool ok = qok && rc[j] >= 0 && iy >= 0 && iy < sH && ix >= 0 && ix < sW;
      const int gp = ok ? (n * sH + iy) * sW + ix : -1;
      okX |= ok ? (1u << j) : 0u;
      rX[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (gp * sld + cch) * 4, 0, 0));
    }
    ncur = n;
  };
  auto zsrc = [&](int tile, int half) -> const float* {
    const int n = tile / tiles_per_n, rem = tile - n * tiles_per_n;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    return d.dz + ((size_t)(n * d.OH + ty * WG_ROWS + 2 * half) * d.OW + tx * 32) * d.dz_ldc + co0;
  };
  auto dma = [&](const float* __restrict__ src, float* __restrict__ dst) {
#pragma unroll
    for (int jj = 0; jj < ZPW; ++jj)
      __builtin_amdgcn_global_load_lds(src + offZ[jj], (lds_ptr_t)(dst + (ZPW * wave + jj) * 256), 16, 0, 0);
  };
  typedef __attribute__((address_space(3))) ws16x4* lds_tr_t;
  auto afrag = [&](const char* __restrict__ base, wbf16x8 (&a)[3]) {   // 8-pixel A fragment of one tap, three planes
#ifdef PMF_WG_NOTR       /* ablation build: no transposing LDS reads */
#pragma unroll
    for (int p = 0; p < 3; ++p) { wu32x4 t = {(unsigned)(size_t)base, (unsigned)p, 1u, 2u}; asm volatile("" : "+v"(t)); a[p] = __builtin_bit_cast(wbf16x8, t); }
    return;
#endif
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const ws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + p * 64));
      const ws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + p * 64 + 4 * WS3_XPB));
      a[p] = __builtin_bit_cast(wbf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };
  // one half tile = 2 rows x 32 pixels = 4 slabs of 16 pixels; wave w owns slab w of both halves.
  // prep: B fragment dz[pixel 8 lh + e][co li], e = 0..7, of the wave's slab, split in registers
  const int rr = wave >> 1, xs = (wave & 1) * 16;
  auto prep = [&](const float* __restrict__ Zh, wbf16x8 (&bf)[3]) {
    const float* zp = Zh + (rr * 32 + xs + lh * 8) * BN + li;
    float z[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = zp[e * BN];
    wu32x4 b0, b1, b2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned p0, p1, p2;
      ws3_split2(z[2 * e], z[2 * e + 1], p0, p1, p2);
      b0[e] = p0; b1[e] = p1; b2[e] = p2;
    }
    bf[0] = __builtin_bit_cast(wbf16x8, b0); bf[1] = __builtin_bit_cast(wbf16x8, b1); bf[2] = __builtin_bit_cast(wbf16x8, b2);
  };
  // mma: TB taps x 6 MFMAs of half h.  Taps go in groups of G with their MFMAs interleaved product-major (six MFMAs in a
  // row into ONE accumulator are a dependent chain); the next group's A fragments are read while this group multiplies.
  auto mma = [&](int h, const wbf16x8 (&bf)[3]) {
    const char* xb = Xs + ((2 * h + rr) * in_cols + xs) * WS3_XPB + trofs;
    constexpr int G = TB % 3 == 0 ? 3 : (TB % 2 == 0 ? 2 : 1), NG = TB / G;
    wbf16x8 a[2][G][3];
#pragma unroll
    for (int t = 0; t < G; ++t) afrag(xb + toff[t], a[0][t]);
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};   // smallest terms first
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      const int cur = gq & 1, nxt = cur ^ 1;
      if (gq + 1 < NG) {
#pragma unroll
        for (int t = 0; t < G; ++t) afrag(xb + toff[(gq + 1) * G + t], a[nxt][t]);
      }
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int t = 0; t < G; ++t)
#ifdef PMF_WG_NOMFMA     /* ablation build of tools/trace_wgrad.py: keep the LDS reads alive, no matrix work */
          acc[gq * G + t][pr] += __builtin_bit_cast(float, __builtin_bit_cast(wu32x4, a[cur][t][PA[pr]])[pr & 3] ^ __builtin_bit_cast(wu32x4, bf[PB[pr]])[pr & 3]);
#else
          acc[gq * G + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][t][PA[pr]], bf[PB[pr]], acc[gq * G + t], 0, 0, 0);
#endif
    }
  };

  // Vector-memory queue of a wave (operations complete in order) while it computes tile t:  [dz half 0 (t)] [dz half 1 (t)]
  // [input tile (t+1)].  Every wave DMAs and reads only its own dz slab, so the dz buffers need no barrier: a wave waits for
  // ITS DMA (vmcnt), copies the slab into registers (prep) and re-issues the DMA of the next tile into the same slab at once.
  // Both B fragments are prepared up front and the 2 x 54 MFMAs run as one stream.  Two barriers per tile (input tile
  // write-after-read / read-after-write).
  int tile = split;
  if (tile < g.total_tiles) {
    fetch(tile, true);
    dma(zsrc(tile, 0), Z0);
    dma(zsrc(tile, 1), Z1);
  }
  WTR();
  while (tile < g.total_tiles) {
    __syncthreads();                       // X: everyone finished reading the previous input tile
    WTR();
    f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f}, cm4 = {1.f, 1.f, 1.f, 1.f};
    if (aff && qok) { sc4 = *(const f32x4*)(d.src[si].scale + cch); sh4 = *(const f32x4*)(d.src[si].shift + cch); }
    if (d.src[si].cmul && qok) cm4 = *(const f32x4*)(d.src[si].cmul + (size_t)ncur * d.src[si].cmul_ld + cch);
#pragma unroll
    for (int j = 0; j < XSL; ++j) {
      if (rc[j] >= 0) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if ((okX >> j) & 1u) {
          t = rX[j];
          if (aff) t = t * sc4 + sh4;
          if (sflags & PMF_SRC_RELU) {
            t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
          }
          t = t * cm4;
        }
        unsigned l0, l1, l2, h0, h1, h2;
        ws3_split2(t.x, t.y, l0, l1, l2);
        ws3_split2(t.z, t.w, h0, h1, h2);
        char* o = Xs + ((tid + 256 * j) >> 3) * WS3_XPB + q * 8;
        *(wu32x2*)(o) = wu32x2{l0, h0};
        *(wu32x2*)(o + 64) = wu32x2{l1, h1};
        *(wu32x2*)(o + 128) = wu32x2{l2, h2};
      }
    }
    const int next = tile + d.nsplit;
    const bool have = next < g.total_tiles;
    const int nt = have ? next : tile;     // past the end: re-request the current tile (keeps the queue shape; never used)
    WTR();
    __syncthreads();                       // Y: input tile visible
    WTR();
    fetch(nt, have);
    wbf16x8 bf0[3], bf1[3];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XSL + ZPW) : "memory");    // my dz slab of half 0 landed
    prep(Z0, bf0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // slab read: the DMA below may overwrite it
    dma(zsrc(nt, 0), Z0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XSL + ZPW) : "memory");    // ... of half 1
    prep(Z1, bf1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dma(zsrc(nt, 1), Z1);
    WTR();
    mma(0, bf0);
    mma(1, bf1);
    WTR();
    tile = next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WTR();

  // ---- sum the four pixel groups (fixed order) and write this workgroup's partial slab
  {
    float* red = smem;   // [4 waves][16][64]
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      __syncthreads();
      if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[j][r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int p = 1; p < 4; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] += red[((p) * 16 + r) * 64 + lane];
      }
    }
  }
  if (wave == 0) {
    float* part = d.partial + (size_t)split * d.ntaps * g.Ktot * g.Cout32;
    const int co = co0 + li;
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < kc) part[((size_t)j * g.Ktot + k0 + ci) * g.Cout32 + co] = acc[j][r];
      }
  }
  WTR();
  WTR_END();
}

// ---- conv_wgrad_s3_body, software-pipelined inside the wave (SWP) ---------------------------------------------------
// 144 accumulator registers leave room for ONE wave per SIMD, so nothing in conv_wgrad_s3_body overlaps: per tile the
// split + store of the input tile (3.2k cycles), the B-fragment preparation (3.0k) and two barrier rendezvous sit in
// front of 108 MFMAs (4.7k; the pipe needs 3.5k) -- 28 % MFMA-busy (profiles/r02_trace_wgrad.txt).  Here the input tile
// is double-buffered in LDS and tile t + 1 is split and stored WHILE tile t is multiplied: the vector-ALU work and the
// ds_writes are placed slot by slot between the tap groups of mma(), where they issue in the shadow of the 8-pass
// MFMAs (one wave per SIMD: a filler next to an MFMA costs its issue slot only).  One barrier per tile instead of two.
//   vector-memory queue of a wave at the top of iteration t:  [dz half 0 (t)] [dz half 1 (t)] [input tile (t+1)]
template <int TB, int XSL>
__device__ __forceinline__ void conv_wgrad_s3_swp_body(const pmf_wgrad_desc_t& d, const WgGeom& g, float* __restrict__ smem) {
  constexpr int BN = 32;
  constexpr int HPX = 64;
  constexpr int ZPW = HPX * BN / 256 / 4;   // 2 DMA instructions per wave per half
  constexpr int PPI = 256 / BN;
  char* __restrict__ Xs0 = (char*)smem;
  float* __restrict__ Z0 = smem + g.x_floats;
  float* __restrict__ Z1 = Z0 + HPX * BN;
  char* __restrict__ Xs1 = (char*)(Z1 + HPX * BN);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, chunk = blockIdx.y;
  const int co0 = (int)blockIdx.z * BN;
  const int in_cols = g.in_cols;

  int wtri_ = 0;
  (void)wtri_;
  WTR();
  int si = 0, c0 = 0, k0 = 0;
  {
    int rem = chunk;
    for (;;) {
      const int nch = (d.src[si].C + WG_CI - 1) / WG_CI;     // (a 16-channel operand is one half-empty chunk)
      if (rem < nch) { c0 = rem * WG_CI; k0 += c0; break; }
      rem -= nch; k0 += d.src[si].C; ++si;
    }
  }
  const int sld = d.src[si].ldc, sflags = d.src[si].flags;
  const int sH = d.OH, sW = d.OW;
  const bool aff = d.src[si].scale != nullptr, has_cm = d.src[si].cmul != nullptr;
  const int q = tid & 7, cch = c0 + q * 4;
  const int kc = min(WG_CI, d.src[si].C - c0);      // channels of this chunk that exist
  const bool qok = q * 4 < kc;                      // this thread's four channels exist (else: zeros)
  const int totalX = g.in_rows * in_cols * 8;
  // per slot, fixed for the life of the workgroup: (row, column) inside the input tile and the byte offset of the
  // slot's 16 bytes relative to the tile's first pixel; a slot beyond the tile (or channels that do not exist) gets a
  // row no image has
  int sr[XSL], sc[XSL], so[XSL];
  const float rcols = 1.f / (float)in_cols;
#pragma unroll
  for (int j = 0; j < XSL; ++j) {
    const int f = tid + 256 * j, pix = f >> 3;
    const int r = pmf_fdiv(pix, in_cols, rcols), c = pix - r * in_cols;
    sr[j] = (f < totalX && qok) ? r : 0x7fff;
    sc[j] = c;
    so[j] = ((r * sW + c) * sld + cch) * 4;
  }
  int offZ[ZPW];
#pragma unroll
  for (int jj = 0; jj < ZPW; ++jj) {
    const int p = (ZPW * wave + jj) * PPI + lane / (BN / 4);       // wave w DMAs the 16 pixels of ITS slab (it alone reads them)
    offZ[jj] = ((p >> 5) * d.OW + (p & 31)) * d.dz_ldc + (lane % (BN / 4)) * 4;
  }
  int toff[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j)
    toff[j] = (((int)d.tdy[j] - g.dy_min) * in_cols + ((int)d.tdx[j] - g.dx_min)) * WS3_XPB;
  const int trofs = (((lane >> 5) * 8 + ((lane & 15) >> 2)) * WS3_XPB) + ((((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);

  f32x16 acc[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const __amdgpu_buffer_rsrc_t xrs_on =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, d.N * sH * sW * sld * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs_off = __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, 0, 0x00020000);
  // per-sample channel multiplier (Dropout2d): fetched with the tile it belongs to (one more entry of the queue; a
  // zero-sized resource -- the load returns 0 without touching memory -- when the operand has none)
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(has_cm ? d.src[si].cmul : d.src[si].x), 0, (has_cm && qok) ? d.N * d.src[si].cmul_ld * 4 : 0, 0x00020000);
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  if (aff && qok) { sc4 = *(const f32x4*)(d.src[si].scale + cch); sh4 = *(const f32x4*)(d.src[si].shift + cch); }
  f32x4 rX[XSL], rC;
  float mX[XSL];                           // 1 where the slot's element exists, 0 where it is padding
  const int tiles_per_n = g.tiles_x * g.tiles_y;
  // dz of a tile from its (sample, tile row, tile column): the coordinates are computed once per tile (fetch_coords)
  // and handed down -- every scalar instruction is an issue slot of the only wave on this SIMD
  auto zsrc_c = [&](int n, int ty, int tx, int half) -> const float* {
    return d.dz + ((size_t)(n * d.OH + ty * WG_ROWS + 2 * half) * d.OW + tx * 32) * d.dz_ldc + co0;
  };
  auto dma = [&](const float* __restrict__ src, float* __restrict__ dst) {
#pragma unroll
    for (int jj = 0; jj < ZPW; ++jj)
      __builtin_amdgcn_global_load_lds(src + offZ[jj], (lds_ptr_t)(dst + (ZPW * wave + jj) * 256), 16, 0, 0);
  };
  typedef __attribute__((address_space(3))) ws16x4* lds_tr_t;
  auto afrag = [&](const char* __restrict__ base, wbf16x8 (&a)[3]) {   // 8-pixel A fragment of one tap, three planes
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const ws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + p * 64));
      const ws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + p * 64 + 4 * WS3_XPB));
      a[p] = __builtin_bit_cast(wbf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };
  const int rr = wave >> 1, xs = (wave & 1) * 16;
  // B fragment dz[pixel 8 lh + e][co li], e = 0..7, of the wave's slab: read (prep_load), then split in registers
  auto prep_load = [&](const float* __restrict__ Zh, float (&z)[8]) {
    const float* zp = Zh + (rr * 32 + xs + lh * 8) * BN + li;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = zp[e * BN];
  };
  auto prep_split = [&](const float (&z)[8], wbf16x8 (&bf)[3]) {
    wu32x4 b0, b1, b2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned p0, p1, p2;
      ws3_split2(z[2 * e], z[2 * e + 1], p0, p1, p2);
      b0[e] = p0; b1[e] = p1; b2[e] = p2;
    }
    bf[0] = __builtin_bit_cast(wbf16x8, b0); bf[1] = __builtin_bit_cast(wbf16x8, b1); bf[2] = __builtin_bit_cast(wbf16x8, b2);
  };
  // transform + split + store of slot j of the tile held in rX (rC: its sample's channel multiplier) into Xd.
  // Branch-free (one basic block with the MFMAs around it, or nothing interleaves): a padding element is selected to
  // zero AFTER the transform, a slot beyond the tile writes to the spare pixel behind the buffer.
  const int trash = g.in_rows * in_cols;
  f32x4 cmS = {1.f, 1.f, 1.f, 1.f};        // channel multiplier of the tile being split (set once per tile)
  const float lo = (sflags & PMF_SRC_RELU) ? 0.f : -__builtin_inff();
  auto store_slot = [&](int j, char* __restrict__ Xd) {
    f32x4 t;                                // (scale 1, shift 0 without a BatchNorm view; no packed-f32 forms here)
    t.x = ws3_fma(rX[j].x, sc4.x, sh4.x); t.y = ws3_fma(rX[j].y, sc4.y, sh4.y);
    t.z = ws3_fma(rX[j].z, sc4.z, sh4.z); t.w = ws3_fma(rX[j].w, sc4.w, sh4.w);
    t.x = ws3_vmax(t.x, lo); t.y = ws3_vmax(t.y, lo); t.z = ws3_vmax(t.z, lo); t.w = ws3_vmax(t.w, lo);
    const float m = mX[j];
    t.x = ws3_mul(t.x, ws3_mul(cmS.x, m)); t.y = ws3_mul(t.y, ws3_mul(cmS.y, m));
    t.z = ws3_mul(t.z, ws3_mul(cmS.z, m)); t.w = ws3_mul(t.w, ws3_mul(cmS.w, m));
    unsigned l0, l1, l2, h0, h1, h2;
    ws3_split2_np(t.x, t.y, l0, l1, l2);
    ws3_split2_np(t.z, t.w, h0, h1, h2);
    const int pix = (tid + 256 * j) < totalX ? ((tid + 256 * j) >> 3) : trash;
    char* o = Xd + pix * WS3_XPB + q * 8;
    *(wu32x2*)(o) = wu32x2{l0, h0};
    *(wu32x2*)(o + 64) = wu32x2{l1, h1};
    *(wu32x2*)(o + 128) = wu32x2{l2, h2};
  };
  // ---- one tile: the MFMAs of tile t (2 x NG tap groups) with the split + store of tile t + 1 between the groups (slot
  // by slot, into the other input buffer).  Measured and not kept: also moving the loads of tile t + 2 and the B-fragment
  // preparation of tile t + 1 between the groups (128 vs 120 us on 64 -> 64 3x3 d2 at 64x2048) -- with ONE wave per
  // SIMD the loop is bound by the number of instructions the wave has to issue (~770 per tile at one issue slot every
  // four cycles plus dependent-issue latency: the same 5.4k / 6.8k cycles with the MFMAs compiled out), not by the pipe.
  // Vector-memory queue of a wave, oldest first, at the top of iteration t:  [dz0 t][dz1 t][input t+1]
  constexpr int G = TB % 3 == 0 ? 3 : (TB % 2 == 0 ? 2 : 1), NG = TB / G;
  constexpr int NGT = 2 * NG;                                 // groups per tile
  // (sample, tile row, tile column) of the tile being fetched advance by the workgroup's stride without a division: every
  // scalar instruction is an issue slot of the only wave on this SIMD
  int f_by = 0, f_bx = 0, f_base = 0, f_n = 0, f_ty = 0, f_tx = 0;
  bool f_on = false;
  const int st_n = d.nsplit / tiles_per_n, st_r = d.nsplit - st_n * tiles_per_n;
  const int st_ty = st_r / g.tiles_x, st_tx = st_r - st_ty * g.tiles_x;
  auto coords_set = [&]() {
    f_by = f_ty * WG_ROWS + g.dy_min; f_bx = f_tx * 32 + g.dx_min;
    f_base = ((f_n * sH + f_by) * sW + f_bx) * sld * 4;
  };
  auto fetch_first = [&](int tile) {
    f_n = tile / tiles_per_n;
    const int rem = tile - f_n * tiles_per_n;
    f_ty = rem / g.tiles_x; f_tx = rem - f_ty * g.tiles_x;
    f_on = true;
    coords_set();
  };
  auto fetch_next = [&](bool on) {            // on: the next tile exists (else the current coordinates are requested again)
    if (on) {
      f_tx += st_tx;
      if (f_tx >= g.tiles_x) { f_tx -= g.tiles_x; ++f_ty; }
      f_ty += st_ty;
      if (f_ty >= g.tiles_y) { f_ty -= g.tiles_y; ++f_n; }
      f_n += st_n;
      coords_set();
    }
    f_on = on;
  };
  // branch-free: the tile's first pixel is one scalar, the per-slot part is precomputed; a padding / non-existent
  // element gets an offset beyond every resource (the load returns 0 without touching memory)
  auto fetch_slot = [&](int j) {
    const bool ok = (unsigned)(f_by + sr[j]) < (unsigned)sH && (unsigned)(f_bx + sc[j]) < (unsigned)sW;
    mX[j] = ok ? 1.f : 0.f;
    const unsigned off = ok ? (unsigned)(f_base + so[j]) : 0x80000000u;
    rX[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f_on ? xrs_on : xrs_off, off, 0, 0));
    if (j == XSL - 1)
      rC = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(crs, (f_n * d.src[si].cmul_ld + cch) * 4, 0, 0));
  };
  auto tile_mma = [&](const char* __restrict__ Xc, char* __restrict__ Xn, const wbf16x8 (&bf0)[3],
                      const wbf16x8 (&bf1)[3]) {
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};   // smallest terms first
    wbf16x8 a[2][G][3];
    const char* xb0 = Xc + ((0 + rr) * in_cols + xs) * WS3_XPB + trofs;
    const char* xb1 = Xc + ((2 + rr) * in_cols + xs) * WS3_XPB + trofs;
#pragma unroll
    for (int t = 0; t < G; ++t) afrag(xb0 + toff[t], a[0][t]);
#pragma unroll
    for (int gt = 0; gt < NGT; ++gt) {
      const int cur = gt & 1, nxt = cur ^ 1;
      const int h = gt / NG, gq = gt % NG;
      if (gt + 1 < NGT) {
        const int h2 = (gt + 1) / NG, g2 = (gt + 1) % NG;
#pragma unroll
        for (int t = 0; t < G; ++t) afrag((h2 ? xb1 : xb0) + toff[g2 * G + t], a[nxt][t]);
      }
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int t = 0; t < G; ++t)
          acc[gq * G + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][t][PA[pr]], (h ? bf1 : bf0)[PB[pr]], acc[gq * G + t], 0, 0, 0);
#ifndef PMF_WG_NOSPLIT   /* ablation build of tools/trace_wgrad.py */
#pragma unroll
      for (int j = gt * XSL / NGT; j < (gt + 1) * XSL / NGT; ++j) store_slot(j, Xn);
#endif
    }
  };

  int tile = split;
  int cur = 0;
  if (tile < g.total_tiles) {
    fetch_first(tile);
#pragma unroll
    for (int j = 0; j < XSL; ++j) fetch_slot(j);
    dma(zsrc_c(f_n, f_ty, f_tx, 0), Z0);
    dma(zsrc_c(f_n, f_ty, f_tx, 1), Z1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ZPW) : "memory");      // the first input tile landed
    if (has_cm) cmS = rC;
#pragma unroll
    for (int j = 0; j < XSL; ++j) store_slot(j, Xs0);
    fetch_next(tile + d.nsplit < g.total_tiles);
#pragma unroll
    for (int j = 0; j < XSL; ++j) fetch_slot(j);
  }
  int z_n = f_n, z_ty = f_ty, z_tx = f_tx;      // coordinates of tile t + 1 (or of t again past the end): whose dz is DMA'd in iteration t
  WTR();
  while (tile < g.total_tiles) {
    const int next = tile + d.nsplit;
    const char* Xc = cur ? Xs1 : Xs0;
    char* Xn = cur ? Xs0 : Xs1;
    __syncthreads();                       // tile t complete in Xc; everyone finished reading Xn (tile t - 1)
    WTR();
    wbf16x8 bf0[3], bf1[3];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XSL + 1) : "memory");      // my dz slabs (both halves) landed
    float z0[8], z1[8];
    prep_load(Z0, z0);
    prep_load(Z1, z1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // slabs read: the DMAs below may overwrite them
    dma(zsrc_c(z_n, z_ty, z_tx, 0), Z0);
    dma(zsrc_c(z_n, z_ty, z_tx, 1), Z1);
    prep_split(z0, bf0);                    // (both halves' splits interleave, under the DMA issue)
    prep_split(z1, bf1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ZPW) : "memory");      // input tile t + 1 landed in registers
    cmS.x = has_cm ? rC.x : 1.f; cmS.y = has_cm ? rC.y : 1.f; cmS.z = has_cm ? rC.z : 1.f; cmS.w = has_cm ? rC.w : 1.f;
    WTR();
    // (past the last tile the registers hold the zeros of a zero-sized resource: the store into the idle buffer is
    // harmless, and one code path keeps the accumulators in place)
    tile_mma(Xc, Xn, bf0, bf1);
    WTR();
    fetch_next(next + d.nsplit < g.total_tiles);
    z_n = f_n; z_ty = f_ty; z_tx = f_tx;
#pragma unroll
    for (int j = 0; j < XSL; ++j) fetch_slot(j);
    WTR();
    tile = next;
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WTR();

  // ---- sum the four pixel groups (fixed order) and write this workgroup's partial slab
  {
    float* red = smem;   // [4 waves][16][64]
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      __syncthreads();
      if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[j][r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int p = 1; p < 4; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] += red[((p) * 16 + r) * 64 + lane];
      }
    }
  }
  if (wave == 0) {
    float* part = d.partial + (size_t)split * d.ntaps * g.Ktot * g.Cout32;
    const int co = co0 + li;
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < kc) part[((size_t)j * g.Ktot + k0 + ci) * g.Cout32 + co] = acc[j][r];
      }
  }
  WTR();
  WTR_END();
}

template <int TB, int XSL>
__global__ __launch_bounds__(256) void conv_wgrad_s3_swp_k(const pmf_wgrad_desc_t d, const WgGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_wgrad_s3_swp_body<TB, XSL>(d, g, smem);
}

// ---- N-split form (S3N): the four waves own DIFFERENT output-channel tiles of one staged input tile ----------------------
// In conv_wgrad_s3_swp_body all four waves carry the same (32 ci x 32 co) tile for a quarter of the pixels each: the input
// tile -- whose transform + three-way split + LDS store is ~300 of the ~770 instructions a wave issues per tile -- feeds
// only 108 MFMAs per wave, and the loop is bound by instruction issue, not by the matrix pipe (round 3 phase stamps: the
// same time with the MFMAs compiled out).  Here a workgroup owns 32 ci x (32 NCO) co: wave w carries output-channel tile
// w % NCO for the pixel group w / NCO (NCO = 4: every wave sees all eight 16-pixel slabs of a tile; NCO = 2: four), so one
// staged tile feeds NCO x as many MFMAs per wave (432 / 216 per tile at nine taps) for the same staging work.
//   * dz never passes through LDS: the B fragment of a slab (pixel 8 lh + e, channel co0 + li) is eight 4-byte buffer
//     loads per lane -- 128 contiguous bytes per pixel and half wave -- issued one slab ahead and split in registers at
//     the end of the slab before; the tile/slab/pixel part of the address is a scalar (soffset);
//   * the input tile is double-buffered in LDS and tile t + 1 is split and stored between the tap groups of tile t, as in
//     the SWP body; one barrier per tile; LDS = the two input buffers only;
//   * pixel groups (NCO = 2) are folded in a fixed order through LDS at the end; every wave with pixel group 0 writes its
//     co tile of the partial slab.  Partial-slab layout, stage 2 and arithmetic are those of the SWP body.
//   * RAG (round 5): output maps whose height is not a multiple of 4 or whose width is not a multiple of 32 (S_B: 60 x 80 and
//     30 x 40 maps -- those layers ran the fp32-MFMA unit-dealing kernel at 27-37 TFLOP/s): a dz load of a pixel beyond the map
//     takes an out-of-range offset (hardware zero) instead of wrapping into the next row -- two vector instructions per load,
//     only in this instantiation; the input tile needs nothing (a product with dz = 0 is 0 whatever the input pixel is).
template <int TB, int XSL, int NCO, bool RAG>
__device__ __forceinline__ void conv_wgrad_s3n_body(const pmf_wgrad_desc_t& d, const WgGeom& g, float* __restrict__ smem) {
  constexpr int NPX = 4 / NCO;              // pixel groups
  constexpr int NSL = 8 / NPX;              // 16-pixel slabs of a 4 x 32 tile per wave
  char* __restrict__ Xs0 = (char*)smem;
  char* __restrict__ Xs1 = (char*)(smem + g.x_floats);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave % NCO, pg = wave / NCO;
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, chunk = blockIdx.y;
  const int co0 = ((int)blockIdx.z * NCO + cw) * 32;
  const int in_cols = g.in_cols;

  int wtri_ = 0;
  (void)wtri_;
  WTR();
  int si = 0, c0 = 0, k0 = 0;
  {
    int rem = chunk;
    for (;;) {
      const int nch = (d.src[si].C + WG_CI - 1) / WG_CI;
      if (rem < nch) { c0 = rem * WG_CI; k0 += c0; break; }
      rem -= nch; k0 += d.src[si].C; ++si;
    }
  }
  const int sld = d.src[si].ldc, sflags = d.src[si].flags;
  const int sH = d.OH, sW = d.OW;
  const bool aff = d.src[si].scale != nullptr, has_cm = d.src[si].cmul != nullptr;
  const int q = tid & 7, cch = c0 + q * 4;
  const int kc = min(WG_CI, d.src[si].C - c0);
  const bool qok = q * 4 < kc;
  const int totalX = g.in_rows * in_cols * 8;
  int sr[XSL], sc[XSL], so[XSL];
  const float rcols = 1.f / (float)in_cols;
#pragma unroll
  for (int j = 0; j < XSL; ++j) {
    const int f = tid + 256 * j, pix = f >> 3;
    const int r = pmf_fdiv(pix, in_cols, rcols), c = pix - r * in_cols;
    sr[j] = (f < totalX && qok) ? r : 0x7fff;
    sc[j] = c;
    so[j] = ((r * sW + c) * sld + cch) * 4;
  }
  int toff[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j)
    toff[j] = (((int)d.tdy[j] - g.dy_min) * in_cols + ((int)d.tdx[j] - g.dx_min)) * WS3_XPB;
  const int trofs = (((lane >> 5) * 8 + ((lane & 15) >> 2)) * WS3_XPB) + ((((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);

  f32x16 acc[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const __amdgpu_buffer_rsrc_t xrs_on =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, d.N * sH * sW * sld * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs_off = __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(has_cm ? d.src[si].cmul : d.src[si].x), 0, (has_cm && qok) ? d.N * d.src[si].cmul_ld * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t zrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.dz, 0, d.N * d.OH * d.OW * d.dz_ldc * 4, 0x00020000);
  const int zvoff = (lh * 8 * d.dz_ldc + co0 + li) * 4;          // the lane's part of a B-fragment address
  const int zpix = d.dz_ldc * 4;                                  // bytes per pixel of dz
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  if (aff && qok) { sc4 = *(const f32x4*)(d.src[si].scale + cch); sh4 = *(const f32x4*)(d.src[si].shift + cch); }
  f32x4 rX[XSL], rC;
  float mX[XSL];
  const int tiles_per_n = g.tiles_x * g.tiles_y;
  typedef __attribute__((address_space(3))) ws16x4* lds_tr_t;
  auto afrag = [&](const char* __restrict__ base, wbf16x8 (&a)[3]) {
#ifdef PMF_WG_NOTR       /* ablation build: no transposing LDS reads */
#pragma unroll
    for (int p = 0; p < 3; ++p) { wu32x4 t = {(unsigned)(size_t)base, (unsigned)p, 1u, 2u}; asm volatile("" : "+v"(t)); a[p] = __builtin_bit_cast(wbf16x8, t); }
    return;
#endif
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const ws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + p * 64));
      const ws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + p * 64 + 4 * WS3_XPB));
      a[p] = __builtin_bit_cast(wbf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };
  // slab i of this wave: tile row / first column, and the scalar byte offset of its first pixel relative to the tile's
  auto slab_rr = [&](int i) { return (i * NPX + pg) >> 1; };
  auto slab_xs = [&](int i) { return ((i * NPX + pg) & 1) * 16; };
  const int trash = g.in_rows * in_cols;
  f32x4 cmS = {1.f, 1.f, 1.f, 1.f};
  const float lo = (sflags & PMF_SRC_RELU) ? 0.f : -__builtin_inff();
  auto store_slot = [&](int j, char* __restrict__ Xd) {
    f32x4 t;
    t.x = ws3_fma(rX[j].x, sc4.x, sh4.x); t.y = ws3_fma(rX[j].y, sc4.y, sh4.y);
    t.z = ws3_fma(rX[j].z, sc4.z, sh4.z); t.w = ws3_fma(rX[j].w, sc4.w, sh4.w);
    t.x = ws3_vmax(t.x, lo); t.y = ws3_vmax(t.y, lo); t.z = ws3_vmax(t.z, lo); t.w = ws3_vmax(t.w, lo);
    const float m = mX[j];
    t.x = ws3_mul(t.x, ws3_mul(cmS.x, m)); t.y = ws3_mul(t.y, ws3_mul(cmS.y, m));
    t.z = ws3_mul(t.z, ws3_mul(cmS.z, m)); t.w = ws3_mul(t.w, ws3_mul(cmS.w, m));
    unsigned l0, l1, l2, h0, h1, h2;
    ws3_split2_np(t.x, t.y, l0, l1, l2);
    ws3_split2_np(t.z, t.w, h0, h1, h2);
    const int pix = (tid + 256 * j) < totalX ? ((tid + 256 * j) >> 3) : trash;
    char* o = Xd + pix * WS3_XPB + q * 8;
    *(wu32x2*)(o) = wu32x2{l0, h0};
    *(wu32x2*)(o + 64) = wu32x2{l1, h1};
    *(wu32x2*)(o + 128) = wu32x2{l2, h2};
  };
  constexpr int G = TB % 3 == 0 ? 3 : (TB % 2 == 0 ? 2 : 1), NG = TB / G;
  constexpr int NGT = NSL * NG;                                // tap groups per tile and wave
  int f_by = 0, f_bx = 0, f_base = 0, f_n = 0, f_ty = 0, f_tx = 0;
  bool f_on = false;
  const int st_n = d.nsplit / tiles_per_n, st_r = d.nsplit - st_n * tiles_per_n;
  const int st_ty = st_r / g.tiles_x, st_tx = st_r - st_ty * g.tiles_x;
  auto coords_set = [&]() {
    f_by = f_ty * WG_ROWS + g.dy_min; f_bx = f_tx * 32 + g.dx_min;
    f_base = ((f_n * sH + f_by) * sW + f_bx) * sld * 4;
  };
  auto fetch_first = [&](int tile) {
    f_n = tile / tiles_per_n;
    const int rem = tile - f_n * tiles_per_n;
    f_ty = rem / g.tiles_x; f_tx = rem - f_ty * g.tiles_x;
    f_on = true;
    coords_set();
  };
  auto fetch_next = [&](bool on) {
    if (on) {
      f_tx += st_tx;
      if (f_tx >= g.tiles_x) { f_tx -= g.tiles_x; ++f_ty; }
      f_ty += st_ty;
      if (f_ty >= g.tiles_y) { f_ty -= g.tiles_y; ++f_n; }
      f_n += st_n;
      coords_set();
    }
    f_on = on;
  };
  auto zbase_f = [&]() { return ((f_n * d.OH + f_ty * WG_ROWS) * d.OW + f_tx * 32) * zpix; };   // dz of the tile at the fetch coordinates
  // rows << 16 | columns of that tile inside the map (RAG)
  auto zvalid_f = [&]() { return (min(WG_ROWS, d.OH - f_ty * WG_ROWS) << 16) | min(32, d.OW - f_tx * 32); };
  const int lh8 = lh * 8;
  auto fetch_slot = [&](int j) {
    const bool ok = (unsigned)(f_by + sr[j]) < (unsigned)sH && (unsigned)(f_bx + sc[j]) < (unsigned)sW;
    mX[j] = ok ? 1.f : 0.f;
    const unsigned off = ok ? (unsigned)(f_base + so[j]) : 0x80000000u;
    rX[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f_on ? xrs_on : xrs_off, off, 0, 0));
    if (j == XSL - 1)
      rC = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(crs, (f_n * d.src[si].cmul_ld + cch) * 4, 0, 0));
  };
  // ---- one tile as a STATIC SCHEDULE of MFMA slots ------------------------------------------------------------------------
  // Left to itself hipcc emits the MFMAs of a tap group back to back and the split / store / address work in bursts between
  // the groups (phase stamps: a tile took 22.6k cycles where its 432 MFMAs need 13.8k; without the A reads -4.4k, without the
  // split of tile t + 1 -3.1k, without the dz path -5.3k: none of it ran under the MFMAs).  One wave per SIMD hides about five
  // single-issue instructions in the shadow of an 8-pass MFMA (MI355X_MICROARCH), so the tile is written as NSL * TB * 6
  // slots = { one MFMA; one ds_read_b64_tr_b16 of the NEXT tap group's A fragments; at most a few "micro-ops" of <= 6
  // instructions; sched_barrier(0) }, nothing crosses a slot boundary.  Micro-op streams:
  //   Z (per slab i): slots 0, 1: the eight dz loads of slab i + 2 (two register sets alternate); the last 13 slots: wait for
  //     the loads of slab i + 1 and split it pair by pair, step by step, into the other B-fragment buffer;
  //   S: transform + split + store of input tile t + 1, ten micro-ops per 16-byte slot, spread over the free slots;
  //   F: the loads of input tile t + 2 in the last XSL free slots of the tile (slab NSL - 1; behind the last S micro-op: rX /
  //     mX are free).  Vector-memory queue at the wait behind F: [dz i+1][dz i+2: 8][F: XSL + 1] -> vmcnt(8 + XSL + 1).
  constexpr int SL = TB * 6;                                   // MFMA slots per slab
  constexpr int NMF = NSL * SL;                                // ... per tile
  constexpr int GM = G * 6;                                    // ... per tap group (= A reads of a group)
  constexpr int ZS0 = SL - 13;                                 // first slot of a slab's wait + split run
  constexpr int FREE = ZS0 - 2;                                // free slots per slab (behind the two load slots)
  constexpr int TFREE = NSL * FREE;                            // ... per tile
  static_assert(FREE >= XSL && NSL >= 2 && NSL % 2 == 0, "schedule");
  constexpr int NS_OPS = XSL * 10;                             // S micro-ops
  constexpr int NF_OPS = XSL;                                  // F micro-ops (one load each; the last also fetches the multiplier):
  constexpr int F_SLOT0 = TFREE - NF_OPS;                      // ... the LAST free slots of the tile, all in slab NSL - 1
  constexpr int S_PER = (NS_OPS + F_SLOT0 - 1) / F_SLOT0;      // S micro-ops per free slot (the free slots before F)
  wu32x4 bq[2][3];                          // B fragments (three planes) of the current / the next slab
  float zr[2][8];                           // dz of slab i + 1 (being split) / slab i + 2 (in flight)
  ws16x4 ah[2][G][3][2];                    // A fragments of the current / the next tap group: [tap][plane][pixel half]
  f32x4 st_t, st_m;                         // S stream state (one 16-byte slot at a time)
  unsigned st_l0 = 0, st_l1 = 0, st_l2 = 0, st_h0 = 0, st_h1 = 0, st_h2 = 0;
  float st_a = 0.f, st_b = 0.f, st_c = 0.f, st_d = 0.f;
  char* st_o = nullptr;
  auto lo16 = [](unsigned p) { return __builtin_bit_cast(float, p << 16); };
  auto hi16 = [](unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); };
  auto zload4 = [&](int tile_base, int tile_zv, int i, int par, int e0) {
    const int sb = tile_base + (slab_rr(i) * d.OW + slab_xs(i)) * zpix;
    if constexpr (RAG) {
      const bool rowok = slab_rr(i) < (tile_zv >> 16);
      const int cols = (tile_zv & 0xffff) - slab_xs(i);                  // columns of this slab's 16 inside the map
#pragma unroll
      for (int e = e0; e < e0 + 4; ++e) {
        const unsigned vo = (rowok && lh8 < cols - e) ? (unsigned)zvoff : 0x80000000u;
        zr[par][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zrs, vo, sb + e * zpix, 0));
      }
    } else {
#pragma unroll
      for (int e = e0; e < e0 + 4; ++e)
        zr[par][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zrs, zvoff, sb + e * zpix, 0));
    }
  };
  auto zsplit_op = [&](int par, int bsel, int e, int step) {     // pair e of zr[par] -> planes of bq[bsel], one step
    float& a = zr[par][2 * e];
    float& b = zr[par][2 * e + 1];
    const unsigned pk = ws3_pk(wf32x2{a, b});
    bq[bsel][step][e] = pk;
    if (step < 2) { a = ws3_sub(a, lo16(pk)); b = ws3_sub(b, hi16(pk)); }
  };
  auto store_op = [&](int j, int k, char* __restrict__ Xd) {     // micro-op k of the transform + split + store of slot j
    switch (k) {
      case 0: st_t.x = ws3_fma(rX[j].x, sc4.x, sh4.x); st_t.y = ws3_fma(rX[j].y, sc4.y, sh4.y);
              st_t.z = ws3_fma(rX[j].z, sc4.z, sh4.z); st_t.w = ws3_fma(rX[j].w, sc4.w, sh4.w); break;
      case 1: st_t.x = ws3_vmax(st_t.x, lo); st_t.y = ws3_vmax(st_t.y, lo); st_t.z = ws3_vmax(st_t.z, lo); st_t.w = ws3_vmax(st_t.w, lo); break;
      case 2: st_m.x = ws3_mul(cmS.x, mX[j]); st_m.y = ws3_mul(cmS.y, mX[j]); st_m.z = ws3_mul(cmS.z, mX[j]); st_m.w = ws3_mul(cmS.w, mX[j]); break;
      case 3: st_t.x = ws3_mul(st_t.x, st_m.x); st_t.y = ws3_mul(st_t.y, st_m.y); st_t.z = ws3_mul(st_t.z, st_m.z); st_t.w = ws3_mul(st_t.w, st_m.w); break;
      case 4: st_l0 = ws3_pk(wf32x2{st_t.x, st_t.y}); st_a = ws3_sub(st_t.x, lo16(st_l0)); st_b = ws3_sub(st_t.y, hi16(st_l0)); break;
      case 5: st_l1 = ws3_pk(wf32x2{st_a, st_b}); st_a = ws3_sub(st_a, lo16(st_l1)); st_b = ws3_sub(st_b, hi16(st_l1)); break;
      case 6: st_l2 = ws3_pk(wf32x2{st_a, st_b});
              st_h0 = ws3_pk(wf32x2{st_t.z, st_t.w}); st_c = ws3_sub(st_t.z, lo16(st_h0)); st_d = ws3_sub(st_t.w, hi16(st_h0)); break;
      case 7: st_h1 = ws3_pk(wf32x2{st_c, st_d}); st_c = ws3_sub(st_c, lo16(st_h1)); st_d = ws3_sub(st_d, hi16(st_h1)); break;
      case 8: { st_h2 = ws3_pk(wf32x2{st_c, st_d});
                const int pix = (tid + 256 * j) < totalX ? ((tid + 256 * j) >> 3) : trash;
                st_o = Xd + pix * WS3_XPB + q * 8; } break;
      default: *(wu32x2*)(st_o) = wu32x2{st_l0, st_h0};
               *(wu32x2*)(st_o + 64) = wu32x2{st_l1, st_h1};
               *(wu32x2*)(st_o + 128) = wu32x2{st_l2, st_h2}; break;
    }
  };
  auto aread = [&](const char* __restrict__ base, int buf, int r) {     // read r of a tap group's 6 G: plane-major, tap, pixel half
    const int p = r / (2 * G), t = (r % (2 * G)) / 2, h = r % 2;
    (void)t;
    ah[buf][(r % (2 * G)) / 2][p][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + p * 64 + h * 4 * WS3_XPB));
  };
  auto tile_mma = [&](const char* __restrict__ Xc, char* __restrict__ Xn, int zb_cur, int zb_nxt, int zv_cur, int zv_nxt) {
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};   // smallest terms first
    auto xb = [&](int i) { return Xc + (slab_rr(i) * in_cols + slab_xs(i)) * WS3_XPB + trofs; };
    wg_static_for<GM>([&](auto R) {           // A fragments of the first tap group (exposed once per tile)
      constexpr int r = decltype(R)::value;
      aread(xb(0) + toff[(r % (2 * G)) / 2], 0, r);
    });
    wg_static_for<NMF>([&](auto SI) {
      constexpr int s = decltype(SI)::value;
      constexpr int i = s / SL, ss = s % SL;                       // slab, slot inside the slab
      constexpr int gt = s / GM, gs = s % GM;                      // tap group of the tile, slot inside the group
      constexpr int gq = gt % NG, cur = gt & 1, nxt = cur ^ 1;
      constexpr int pr = gs / G, t = gs % G;
      {
        const wbf16x8 av = __builtin_bit_cast(wbf16x8, __builtin_shufflevector(ah[cur][t][PA[pr]][0], ah[cur][t][PA[pr]][1], 0, 1, 2, 3, 4, 5, 6, 7));
#ifdef PMF_WG_NOMFMA
        acc[gq * G + t][pr] += __builtin_bit_cast(float, __builtin_bit_cast(wu32x4, av)[pr & 3] ^ bq[i & 1][PB[pr]][pr & 3]);
#else
        acc[gq * G + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(wbf16x8, bq[i & 1][PB[pr]]), acc[gq * G + t], 0, 0, 0);
#endif
      }
#ifndef PMF_WG_NOTR
      if constexpr (s + GM - gs < NMF) {                           // one A read of the next tap group
        constexpr int s2 = s - gs + GM, i2 = s2 / SL, g2 = (s2 / GM) % NG;
        aread(xb(i2) + toff[g2 * G + (gs % (2 * G)) / 2], nxt, gs);
      }
#endif
#ifndef PMF_WG_NOZ
      if constexpr (ss < 2) {                                      // dz loads of slab i + 2
        if constexpr (i + 2 < NSL) zload4(zb_cur, zv_cur, i + 2, i & 1, 4 * ss); else zload4(zb_nxt, zv_nxt, i + 2 - NSL, i & 1, 4 * ss);
      } else if constexpr (ss >= ZS0) {
        constexpr int k = ss - ZS0;                                // 0: wait (+ nothing), 1 .. 12: split steps
        if constexpr (k == 0) {
          if constexpr (i == NSL - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + XSL + 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          zsplit_op((i + 1) & 1, (i + 1) & 1, (k - 1) / 3, (k - 1) % 3);
        }
      }
#endif
      if constexpr (ss >= 2 && ss < ZS0) {
        constexpr int fs = i * FREE + ss - 2;                      // free slot of the tile
        if constexpr (fs < F_SLOT0) {
#ifndef PMF_WG_NOSPLIT
          wg_static_for<S_PER>([&](auto KK) {
            constexpr int k = fs * S_PER + decltype(KK)::value;
            if constexpr (k < NS_OPS) store_op(k / 10, k % 10, Xn);
          });
#endif
        } else {
          fetch_slot(fs - F_SLOT0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  int tile = split;
  int cur = 0;
  int zb_cur = 0, zb_nxt = 0, zv_cur = 0, zv_nxt = 0;
  // prologue: input tile 0 split + stored, B fragment of slab 0 ready, dz of slab 1 and input tile 1 in flight
  if (tile < g.total_tiles) {
    fetch_first(tile);
    zb_cur = zbase_f();
    zv_cur = zvalid_f();
#pragma unroll
    for (int j = 0; j < XSL; ++j) fetch_slot(j);
    zload4(zb_cur, zv_cur, 0, 0, 0);
    zload4(zb_cur, zv_cur, 0, 0, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (has_cm) cmS = rC;
#pragma unroll
    for (int j = 0; j < XSL; ++j) store_slot(j, Xs0);
#pragma unroll
    for (int k = 0; k < 12; ++k) zsplit_op(0, 0, k / 3, k % 3);
    zload4(zb_cur, zv_cur, 1, 1, 0);
    zload4(zb_cur, zv_cur, 1, 1, 4);
    fetch_next(tile + d.nsplit < g.total_tiles);
    zb_nxt = zbase_f();
    zv_nxt = zvalid_f();
#pragma unroll
    for (int j = 0; j < XSL; ++j) fetch_slot(j);
  }
  WTR();
  while (tile < g.total_tiles) {
    const int next = tile + d.nsplit;
    const char* Xc = cur ? Xs1 : Xs0;
    char* Xn = cur ? Xs0 : Xs1;
    __syncthreads();                       // tile t complete in Xc; everyone finished reading Xn (tile t - 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // dz of slab 1 and the input tile t + 1 landed
    WTR();
    cmS.x = has_cm ? rC.x : 1.f; cmS.y = has_cm ? rC.y : 1.f; cmS.z = has_cm ? rC.z : 1.f; cmS.w = has_cm ? rC.w : 1.f;
    // coordinates of tile t + 2 before the schedule runs (its F micro-ops use them); dz bases: t (cur), t + 1 (nxt)
    const int zb_c = zb_cur, zb_n = zb_nxt, zv_c = zv_cur, zv_n = zv_nxt;
    fetch_next(next + d.nsplit < g.total_tiles);
    zb_cur = zb_nxt; zv_cur = zv_nxt;
    zb_nxt = zbase_f(); zv_nxt = zvalid_f();
    tile_mma(Xc, Xn, zb_c, zb_n, zv_c, zv_n);
    WTR();
    WTR();
    tile = next;
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WTR();

  // ---- fold the pixel groups (fixed order) and write the partial slab: wave (cw, 0) owns co tile cw
  if constexpr (NPX > 1) {
    float* red = smem;   // [NCO][16][64]
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      __syncthreads();
      if (pg > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((cw * (NPX - 1) + pg - 1) * 16 + r) * 64 + lane] = acc[j][r];
      }
      __syncthreads();
      if (pg == 0) {
#pragma unroll
        for (int p = 1; p < NPX; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] += red[((cw * (NPX - 1) + p - 1) * 16 + r) * 64 + lane];
      }
    }
  }
  if (pg == 0) {
    float* part = d.partial + (size_t)split * d.ntaps * g.Ktot * g.Cout32;
    const int co = co0 + li;
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < kc) part[((size_t)j * g.Ktot + k0 + ci) * g.Cout32 + co] = acc[j][r];
      }
  }
  WTR();
  WTR_END();
}

template <int TB, int XSL, int NCO, bool RAG = false>
__global__ __launch_bounds__(256) void conv_wgrad_s3n_k(const pmf_wgrad_desc_t d, const WgGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_wgrad_s3n_body<TB, XSL, NCO, RAG>(d, g, smem);
}

// ---- eight waves per workgroup (W8): two waves per SIMD ----------------------------------------------------------------
// What bounds conv_wgrad_s3_swp_body is the instruction stream of the ONE wave a SIMD holds (144 accumulator registers
// per wave): ~770 instructions per tile at one issue slot every four cycles plus dependent-issue latency -- the loop
// takes the same time with the MFMAs compiled out.  Here the taps of a 16-pixel slab are dealt to TWO waves (taps
// [0, TBW) and [TBW, TB): 80 accumulator registers each), 512 threads per workgroup, so that every SIMD holds two
// waves whose instruction streams interleave and the per-thread share of the split halves.
//   * input tile: double-buffered in LDS as in the SWP body, 512 threads x XSL slots (4 or 5);
//   * dz: both halves of a tile are 16 DMA instructions of 1 KiB, two per wave, into a double-buffered slab pair
//     (a slab is read by the two waves that own its pixels: the barrier at the top of an iteration publishes it);
//   * one barrier per tile.
template <int TB, int XSL>
__device__ __forceinline__ void conv_wgrad_s3_w8_body(const pmf_wgrad_desc_t& d, const WgGeom& g, float* __restrict__ smem) {
  constexpr int BN = 32;
  constexpr int HPX = 64;
  constexpr int TBW = (TB + 1) / 2;             // taps of the first wave of a pair; the second takes TB - TBW
  constexpr int NT = 512;
  char* __restrict__ Xs0 = (char*)smem;
  char* __restrict__ Xs1 = Xs0 + g.x_floats * 4;
  float* __restrict__ Zb0 = (float*)(Xs1 + g.x_floats * 4);      // [2 halves][64 pixels][32 co], three tiles in flight
  float* __restrict__ Zb1 = Zb0 + 2 * HPX * BN;
  float* __restrict__ Zb2 = Zb1 + 2 * HPX * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = wave & 3, tset = wave >> 2;
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, chunk = blockIdx.y;
  const int co0 = (int)blockIdx.z * BN;
  const int in_cols = g.in_cols;
  int wtri_ = 0;
  (void)wtri_;
  WTR();

  int si = 0, c0 = 0, k0 = 0;
  {
    int rem = chunk;
    for (;;) {
      const int nch = (d.src[si].C + WG_CI - 1) / WG_CI;     // (a 16-channel operand is one half-empty chunk)
      if (rem < nch) { c0 = rem * WG_CI; k0 += c0; break; }
      rem -= nch; k0 += d.src[si].C; ++si;
    }
  }
  const int sld = d.src[si].ldc, sflags = d.src[si].flags;
  const int sH = d.OH, sW = d.OW;
  const bool aff = d.src[si].scale != nullptr, has_cm = d.src[si].cmul != nullptr;
  const int q = tid & 7, cch = c0 + q * 4;
  const int kc = min(WG_CI, d.src[si].C - c0);      // channels of this chunk that exist
  const bool qok = q * 4 < kc;                      // this thread's four channels exist (else: zeros)
  const int totalX = g.in_rows * in_cols * 8;
  int rc[XSL], so[XSL];
#pragma unroll
  for (int j = 0; j < XSL; ++j) {
    const int f = tid + NT * j, pix = f >> 3;
    const int r = pix / in_cols, c = pix - r * in_cols;
    rc[j] = (f < totalX && qok) ? (r << 8 | c) : (0x7fff << 8);
    so[j] = ((r * sW + c) * sld + cch) * 4;
  }
  // dz: chunk k (0..15) of a tile = 8 pixels x 32 channels (1 KiB); wave w DMAs chunks 2 w and 2 w + 1
  int offZ[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int p = (2 * wave + jj) * 8 + lane / 8;            // pixel 0..127 of the tile: row p / 32, column p % 32
    offZ[jj] = ((p >> 5) * d.OW + (p & 31)) * d.dz_ldc + (lane % 8) * 4;
  }
  // own taps
  const int t0 = tset ? TBW : 0;
  int toff[TBW];
#pragma unroll
  for (int j = 0; j < TBW; ++j) {
    const int t = min(t0 + j, TB - 1);
    toff[j] = (((int)d.tdy[t] - g.dy_min) * in_cols + ((int)d.tdx[t] - g.dx_min)) * WS3_XPB;
  }
  const int trofs = (((lane >> 5) * 8 + ((lane & 15) >> 2)) * WS3_XPB) + ((((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);

  f32x16 acc[TBW];
#pragma unroll
  for (int j = 0; j < TBW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const __amdgpu_buffer_rsrc_t xrs_on =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, d.N * sH * sW * sld * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs_off = __builtin_amdgcn_make_buffer_rsrc((void*)d.src[si].x, 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(has_cm ? d.src[si].cmul : d.src[si].x), 0, (has_cm && qok) ? d.N * d.src[si].cmul_ld * 4 : 0, 0x00020000);
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  if (aff && qok) { sc4 = *(const f32x4*)(d.src[si].scale + cch); sh4 = *(const f32x4*)(d.src[si].shift + cch); }
  f32x4 rX[XSL], rC;
  unsigned okX = 0u;
  const int tiles_per_n = g.tiles_x * g.tiles_y;
  int f_by = 0, f_bx = 0, f_base = 0, f_n = 0, f_ty = 0, f_tx = 0;
  bool f_on = false;
  auto fetch_coords = [&](int tile, bool on) {
    const int n = tile / tiles_per_n, rem = tile - n * tiles_per_n;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    f_by = ty * WG_ROWS + g.dy_min; f_bx = tx * 32 + g.dx_min;
    f_base = ((n * sH + f_by) * sW + f_bx) * sld * 4;
    f_n = n; f_on = on; f_ty = ty; f_tx = tx;
  };
  auto fetch_slot = [&](int j) {
    const unsigned iy = (unsigned)(f_by + (rc[j] >> 8)), ix = (unsigned)(f_bx + (rc[j] & 255));
    const bool ok = iy < (unsigned)sH && ix < (unsigned)sW;
    okX = (okX & ~(1u << j)) | (ok ? (1u << j) : 0u);
    const unsigned off = ok ? (unsigned)(f_base + so[j]) : 0x80000000u;
    rX[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f_on ? xrs_on : xrs_off, off, 0, 0));
    if (j == XSL - 1)
      rC = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(crs, (f_n * d.src[si].cmul_ld + cch) * 4, 0, 0));
  };
  auto dma_z = [&](int n, int ty, int tx, float* __restrict__ Zb) {
    const float* src = d.dz + ((size_t)(n * d.OH + ty * WG_ROWS) * d.OW + tx * 32) * d.dz_ldc + co0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      __builtin_amdgcn_global_load_lds(src + offZ[jj], (lds_ptr_t)(Zb + (2 * wave + jj) * 256), 16, 0, 0);
  };
  typedef __attribute__((address_space(3))) ws16x4* lds_tr_t;
  auto afrag = [&](const char* __restrict__ base, wbf16x8 (&a)[3]) {   // 8-pixel A fragment of one tap, three planes
#ifdef PMF_WG_NOTR       /* ablation build: no transposing LDS reads */
#pragma unroll
    for (int p = 0; p < 3; ++p) { wu32x4 t = {(unsigned)(size_t)base, (unsigned)p, 1u, 2u}; asm volatile("" : "+v"(t)); a[p] = __builtin_bit_cast(wbf16x8, t); }
    return;
#endif
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const ws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + p * 64));
      const ws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + p * 64 + 4 * WS3_XPB));
      a[p] = __builtin_bit_cast(wbf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };
  // the slab's pixels: half h = tile rows 2 h, 2 h + 1; slab s = row s / 2 of the half, pixels 16 (s % 2) ..
  const int rr = slab >> 1, xs = (slab & 1) * 16;
  auto prep = [&](const float* __restrict__ Zh, wbf16x8 (&bf)[3]) {
    const float* zp = Zh + (rr * 32 + xs + lh * 8) * BN + li;
    float z[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = zp[e * BN];
    wu32x4 b0, b1, b2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned p0, p1, p2;
      ws3_split2(z[2 * e], z[2 * e + 1], p0, p1, p2);
      b0[e] = p0; b1[e] = p1; b2[e] = p2;
    }
    bf[0] = __builtin_bit_cast(wbf16x8, b0); bf[1] = __builtin_bit_cast(wbf16x8, b1); bf[2] = __builtin_bit_cast(wbf16x8, b2);
  };
  const int trash = g.in_rows * in_cols;
  f32x4 cmS = {1.f, 1.f, 1.f, 1.f};
  const float lo = (sflags & PMF_SRC_RELU) ? 0.f : -__builtin_inff();
  auto store_slot = [&](int j, char* __restrict__ Xd) {
    const bool ok = (okX >> j) & 1u;
    f32x4 t;
    t.x = ws3_fma(rX[j].x, sc4.x, sh4.x); t.y = ws3_fma(rX[j].y, sc4.y, sh4.y);
    t.z = ws3_fma(rX[j].z, sc4.z, sh4.z); t.w = ws3_fma(rX[j].w, sc4.w, sh4.w);
    t.x = ws3_vmax(t.x, lo); t.y = ws3_vmax(t.y, lo); t.z = ws3_vmax(t.z, lo); t.w = ws3_vmax(t.w, lo);
    const float m = ok ? 1.f : 0.f;
    t.x = ws3_mul(t.x, ws3_mul(cmS.x, m)); t.y = ws3_mul(t.y, ws3_mul(cmS.y, m));
    t.z = ws3_mul(t.z, ws3_mul(cmS.z, m)); t.w = ws3_mul(t.w, ws3_mul(cmS.w, m));
    unsigned l0, l1, l2, h0, h1, h2;
    ws3_split2_np(t.x, t.y, l0, l1, l2);
    ws3_split2_np(t.z, t.w, h0, h1, h2);
    const int pix = (tid + NT * j) < totalX ? ((tid + NT * j) >> 3) : trash;
    char* o = Xd + pix * WS3_XPB + q * 8;
    *(wu32x2*)(o) = wu32x2{l0, h0};
    *(wu32x2*)(o + 64) = wu32x2{l1, h1};
    *(wu32x2*)(o + 128) = wu32x2{l2, h2};
  };
  // one tile: 2 halves x TBW taps; the A fragments of the next (half, tap) are read while this one multiplies; the slots
  // of the next tile's split follow the taps one by one.  The second wave of a pair has one tap less when TB is odd: its
  // last tap position is skipped (wave-uniform branch).
  constexpr int NGT = 2 * TBW;
  const bool last_tap = (t0 + TBW - 1) < TB;
  wbf16x8 bf0[3], bf1[3];          // B fragments of the tile being multiplied
  // Between the taps of tile t: split + store of tile t + 1, the loads of tile t + 2 into the registers a slot has
  // freed, and the B fragments of tile t + 1 from the dz slabs the barrier at the top of the iteration has published.
  auto tile_mma = [&](const char* __restrict__ Xc, char* __restrict__ Xn, const float* __restrict__ Zp) {
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};   // smallest terms first
    wbf16x8 a[2][3], bn0[3], bn1[3];
    const char* xb0 = Xc + ((0 + rr) * in_cols + xs) * WS3_XPB + trofs;
    const char* xb1 = Xc + ((2 + rr) * in_cols + xs) * WS3_XPB + trofs;
    afrag(xb0 + toff[0], a[0]);
#pragma unroll
    for (int gt = 0; gt < NGT; ++gt) {
      const int cur = gt & 1, nxt = cur ^ 1;
      const int h = gt / TBW, k = gt % TBW;
      if (gt + 1 < NGT) afrag(((gt + 1) / TBW ? xb1 : xb0) + toff[(gt + 1) % TBW], a[nxt]);
      if (k < TBW - 1 || last_tap) {
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][PA[pr]], (h ? bf1 : bf0)[PB[pr]], acc[k], 0, 0, 0);
      }
#ifndef PMF_WG_NOSPLIT
#pragma unroll
      for (int j = gt * XSL / NGT; j < (gt + 1) * XSL / NGT; ++j) { store_slot(j, Xn); fetch_slot(j); }
#endif
      if (gt == NGT / 2 - 1) prep(Zp, bn0);
      if (gt == NGT - 2) prep(Zp + HPX * BN, bn1);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) { bf0[p] = bn0[p]; bf1[p] = bn1[p]; }
  };

  // Vector-memory queue of a wave, oldest first, at the top of iteration t: [input t+1 (XSL + 1)]; my share of dz(t+1)
  // has landed (the barrier publishes it)
  int tile = split;
  int cur = 0;
  float* Zq0 = Zb0;   // dz(t)   -- consumed (B fragments prepared during iteration t - 1)
  float* Zq1 = Zb1;   // dz(t+1) -- read during iteration t
  float* Zq2 = Zb2;   // dz(t+2) -- DMA'd during iteration t
  if (tile < g.total_tiles) {
    fetch_coords(tile, true);
#pragma unroll
    for (int j = 0; j < XSL; ++j) fetch_slot(j);
    dma_z(f_n, f_ty, f_tx, Zq0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2) : "memory");            // the first input tile landed
    if (has_cm) cmS = rC;
#pragma unroll
    for (int j = 0; j < XSL; ++j) store_slot(j, Xs0);
    const int n1 = tile + d.nsplit;
    fetch_coords(n1 < g.total_tiles ? n1 : tile, n1 < g.total_tiles);
#pragma unroll
    for (int j = 0; j < XSL; ++j) fetch_slot(j);
    dma_z(f_n, f_ty, f_tx, Zq1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XSL + 1 + 2) : "memory");  // my share of dz(t) landed
    __syncthreads();
    prep(Zq0, bf0);
    prep(Zq0 + HPX * BN, bf1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XSL + 1) : "memory");      // ... of dz(t + 1); queue: [input t+1]
  }
  WTR();
  while (tile < g.total_tiles) {
    const int next = tile + d.nsplit;
    const char* Xc = cur ? Xs1 : Xs0;
    char* Xn = cur ? Xs0 : Xs1;
    const int n2 = next + d.nsplit;
    const bool have2 = n2 < g.total_tiles;
    fetch_coords(have2 ? n2 : tile, have2);
    __syncthreads();                       // input tile t and dz(t + 1) complete; everyone finished with tile t - 1
    WTR();
    dma_z(f_n, f_ty, f_tx, Zq2);           // dz(t + 2); queue: [input t+1][dz t+2 (2)]
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2) : "memory");            // input tile t + 1 landed in registers
    cmS.x = has_cm ? rC.x : 1.f; cmS.y = has_cm ? rC.y : 1.f; cmS.z = has_cm ? rC.z : 1.f; cmS.w = has_cm ? rC.w : 1.f;
    WTR();
    tile_mma(Xc, Xn, Zq1);                 // queue afterwards: [dz t+2 (2)][input t+2]
    WTR();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XSL + 1) : "memory");      // my share of dz(t + 2) landed (published by the next barrier)
    WTR();
    tile = next;
    cur ^= 1;
    float* zt = Zq0; Zq0 = Zq1; Zq1 = Zq2; Zq2 = zt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WTR();

  // ---- sum the four pixel groups (fixed order) and write this workgroup's partial slab: per tap set, slabs 1..3 hand
  // their accumulators to slab 0 through LDS
  {
    float* red = smem + tset * (3 * 16 * 64);   // [3 slabs][16][64] per tap set
#pragma unroll
    for (int j = 0; j < TBW; ++j) {
      __syncthreads();
      if (slab > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((slab - 1) * 16 + r) * 64 + lane] = acc[j][r];
      }
      __syncthreads();
      if (slab == 0) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] += red[(p * 16 + r) * 64 + lane];
      }
    }
  }
  if (slab == 0) {
    float* part = d.partial + (size_t)split * d.ntaps * g.Ktot * g.Cout32;
    const int co = co0 + li;
#pragma unroll
    for (int j = 0; j < TBW; ++j) {
      if (t0 + j < TB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (ci < kc) part[((size_t)(t0 + j) * g.Ktot + k0 + ci) * g.Cout32 + co] = acc[j][r];
        }
      }
    }
  }
  WTR();
  WTR_END();
}

template <int TB, int XSL>
__global__ __launch_bounds__(512) void conv_wgrad_s3_w8_k(const pmf_wgrad_desc_t d, const WgGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_wgrad_s3_w8_body<TB, XSL>(d, g, smem);
}

template <int TB, int XSL>
__global__ __launch_bounds__(256) void conv_wgrad_s3_k(const pmf_wgrad_desc_t d, const WgGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_wgrad_s3_body<TB, XSL>(d, g, smem);
}

template <int TB, int NT, int XSL>
__global__ __launch_bounds__(256) void conv_wgrad_pipe_k(const pmf_wgrad_desc_t d, const WgGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_wgrad_pipe_body<TB, NT, XSL>(d, g, smem);
}

// ------------------------------------------------------------------------------------------------------------
// Few-input-channel convolutions (the 7x7x3 ResNet stem): GEMM rows = (tap, channel) pairs, k = tap * Cin + c, so the
// 147 products of the stem fill 5 MFMA row tiles instead of 49 taps x one 32-row tile that is 29/32 padding.
// The input tile is kept PACKED in LDS ([row][col][Cin]); lane i of row tile g reads its own (tap, channel) through a
// per-lane offset  koff = (dy * in_cols + dx) * Cin + c  -- an im2col view without materialising the im2col matrix.
// Rows k >= ntaps*Cin read a zeroed word.  dz halves arrive by LDS-DMA exactly as in conv_wgrad_pipe_k; wave w owns
// output tile w % 2 and every other pixel pair; the partial slab keeps the [tap][Ktot][Cout32] layout of stage 2.
template <int KG>   // row tiles of 32 (tap, channel) pairs
__device__ __forceinline__ void wgrad_fewc_body(const pmf_wgrad_desc_t& d, const WgGeom& g, float* __restrict__ smem) {
  constexpr int BN = 64, NT = 2, PG = 2, HPX = 64, ZPW = HPX * BN / 256 / 4, PPI = 256 / BN;
  const int cin = d.Cin_real;                           // packed channels per pixel in LDS
  const int in_cols = g.in_cols, in_rows = g.in_rows;
  const int xfl = (in_rows * in_cols * cin + 4 + 3) & ~3;   // + one zero word for the padding rows
  float* __restrict__ Xs = smem;
  float* __restrict__ Z0 = smem + xfl;
  float* __restrict__ Z1 = Z0 + HPX * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x;
  const int co0 = (int)blockIdx.z * BN;
  const int sld = d.src[0].ldc, sH = d.OH, sW = d.OW;
  const int kreal = d.ntaps * cin;
  const int zero_word = in_rows * in_cols * cin;       // index of the zeroed word
  int koff[KG];
#pragma unroll
  for (int j = 0; j < KG; ++j) {
    const int k = j * 32 + li;
    if (k < kreal) {
      const int t = k / cin, c = k - t * cin;
      koff[j] = (((int)d.tdy[t] - g.dy_min) * in_cols + ((int)d.tdx[t] - g.dx_min)) * cin + c;
    } else koff[j] = -1;
  }
  int offZ[ZPW];
#pragma unroll
  for (int jj = 0; jj < ZPW; ++jj) {
    const int p = (wave + 4 * jj) * PPI + lane / (BN / 4);
    offZ[jj] = ((p >> 5) * d.OW + (p & 31)) * d.dz_ldc + (lane % (BN / 4)) * 4;
  }
  const int cot = wave % NT, pg = wave / NT;
  f32x16 acc[KG];
#pragma unroll
  for (int j = 0; j < KG; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int tiles_per_n = g.tiles_x * g.tiles_y;
  const int npixX = in_rows * in_cols;
  auto zsrc = [&](int tile, int half) -> const float* {
    const int n = tile / tiles_per_n, rem = tile - n * tiles_per_n;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    return d.dz + ((size_t)(n * d.OH + ty * WG_ROWS + 2 * half) * d.OW + tx * 32) * d.dz_ldc + co0;
  };
  auto dma = [&](const float* __restrict__ src, float* __restrict__ dst) {
#pragma unroll
    for (int jj = 0; jj < ZPW; ++jj)
      __builtin_amdgcn_global_load_lds(src + offZ[jj], (lds_ptr_t)(dst + (wave + 4 * jj) * 256), 16, 0, 0);
  };
  auto half = [&](const float* __restrict__ Zh, int h) {
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const int xrow = ((2 * h + rr) * in_cols + lh) * cin;
      const float* zp = Zh + (rr * 32 + lh) * BN + cot * 32 + li;
      constexpr int NP = 16 / PG;
      float ac[KG], an[KG], bc, bn;
      bc = zp[2 * pg * BN];
#pragma unroll
      for (int j = 0; j < KG; ++j) ac[j] = Xs[koff[j] < 0 ? zero_word : xrow + 2 * pg * cin + koff[j]];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (i + 1 < NP) {
          const int kp = pg + (i + 1) * PG;
          bn = zp[2 * kp * BN];
#pragma unroll
          for (int j = 0; j < KG; ++j) an[j] = Xs[koff[j] < 0 ? zero_word : xrow + 2 * kp * cin + koff[j]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[j], bc, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bc = bn;
#pragma unroll
        for (int j = 0; j < KG; ++j) ac[j] = an[j];
      }
    }
  };
  if (tid == 0) Xs[zero_word] = 0.f;
  int tile = split;
  if (tile < g.total_tiles) dma(zsrc(tile, 0), Z0);
  while (tile < g.total_tiles) {
    const int n = tile / tiles_per_n, rem = tile - n * tiles_per_n;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    const int by = ty * WG_ROWS + g.dy_min, bx = tx * 32 + g.dx_min;
    __syncthreads();                       // X: everyone finished the previous tile
    for (int p = tid; p < npixX; p += 256) {     // packed input tile (zero padding; the raw image has no view)
      const int r = p / in_cols, c = p - r * in_cols;
      const int iy = by + r, ix = bx + c;
      const bool ok = iy >= 0 && iy < sH && ix >= 0 && ix < sW;
      const float* px = d.src[0].x + ((size_t)(n * sH + iy) * sW + ix) * sld;
      for (int ch = 0; ch < cin; ++ch) Xs[p * cin + ch] = ok ? px[ch] : 0.f;
    }
    const int next = tile + d.nsplit;
    const bool have = next < g.total_tiles;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // Y
    dma(zsrc(tile, 1), Z1);
    __builtin_amdgcn_sched_barrier(0);
    half(Z0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // Z
    if (have) dma(zsrc(next, 0), Z0);
    __builtin_amdgcn_sched_barrier(0);
    half(Z1, 1);
    tile = next;
  }
  // sum the two pixel groups, then write [tap][c][co] rows of this workgroup's partial slab
  float* red = smem;
#pragma unroll
  for (int j = 0; j < KG; ++j) {
    __syncthreads();
    if (pg > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[j][r];
    }
    __syncthreads();
    if (pg == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] += red[((wave + NT) * 16 + r) * 64 + lane];
    }
  }
  if (pg == 0) {
    float* part = d.partial + (size_t)split * d.ntaps * g.Ktot * g.Cout32;
    const int co = co0 + cot * 32 + li;
#pragma unroll
    for (int j = 0; j < KG; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (k < kreal) {
          const int t = k / cin, c = k - t * cin;
          part[((size_t)t * g.Ktot + c) * g.Cout32 + co] = acc[j][r];
        }
      }
  }
}

template <int KG>
__global__ __launch_bounds__(256) void wgrad_fewc_k(const pmf_wgrad_desc_t d, const WgGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  wgrad_fewc_body<KG>(d, g, smem);
}

// ------------------------------------------------------------------------------------------------------------
// 1x1 convolutions: dW[ci][co] = sum_px X[px][ci] * dz[px][co] is a GEMM whose MFMA operands ARE the memory layout --
// the A register of v_mfma_f32_32x32x2 wants, per lane, (row m = lane % 32, k = lane / 32): 32 consecutive channels of
// pixel p in the lower half wave and of pixel p+1 in the upper half; B likewise along Cout.  So the operands go
// global -> registers -> MFMA with no LDS staging and every byte of X and dz read once per (128-channel, CW-channel)
// output block: a lane loads a float4 of X (channels 4q..4q+3) and NCO floats of dz (channels NCO*q..), component j of
// X times component j' of dz feeds accumulator tile (j, j') whose row m stands for channel 4m + j (a permuted tile --
// only the final store needs to know).  The four waves take every fourth pixel pair and are folded through LDS in a
// fixed order; BatchNorm-apply / ReLU / dropout multipliers of the operand are applied in registers.
// ---- streaming kernel for the full-resolution one-tap layers with few channels (round 5) ------------------------------------
// logits 32 -> 20, downCntx.s 5 -> 32, downCntx2/3.s 32 -> 32, resBlock1.s 32 -> 64, dec.up1 80 -> 16, upBlock4.e 96 -> 32 at
// 2 x 64 x 2048: 40-130 MB of operands for 0.1-1.6 GFLOP -- HBM-bound by 10x -- and the tiled kernels spend their time staging
// 4 x 32-pixel tiles through LDS between barriers for 16 MFMAs per wave (35-73 us where the bytes need 7-22).  Here nothing is
// staged: the fp32 MFMA operand layout of v_mfma_f32_32x32x2_f32 (A[i = lane & 31][k = lane >> 5]) IS "channel lane & 31 of pixel
// 2 p + (lane >> 5)", i.e. one dword load per lane and operand, 256 contiguous bytes per wave instruction at a 32-float pixel
// pitch; the view of an operand (BatchNorm scale / shift, ReLU, Dropout2d multiplier) is a per-lane constant because a lane
// keeps its channel.  Every wave streams its own pixel pairs with U pairs of loads in flight and KT x NTL accumulator tiles,
// no barrier until the four waves fold through LDS in a fixed order.  A workgroup stays inside ONE sample (the multiplier is
// per (sample, channel)): grid.x = N * S.  No LDS while streaming, so the launch is 512 workgroups wide without taking the main
// lane's LDS.  Partial-slab layout and stage 2 are those of every other kernel here.
template <int KT, int NTL>
__global__ __launch_bounds__(256) void wgrad_stream_k(const pmf_wgrad_desc_t d, int Ktot, int Cout32, int S) {
  constexpr int U = 8;                       // pixel pairs in flight per wave
  __shared__ float red[3 * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, n = split / S, sp = split - n * S;
  const int P = d.OH * d.OW, PP = (P + 1) >> 1;
  const int per = (PP + S - 1) / S;
  const int p0 = sp * per, p1 = min(PP, p0 + per);

  // per-lane view of the A operand: channel k = kt * 32 + li of the virtual concat
  const float* xp[KT];
  int xld[KT];
  float sc[KT], sh[KT], lo[KT], cm[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    int k = kt * 32 + li, si = 0;
    const bool valid = k < Ktot;
    if (!valid) k = 0;
    while (si + 1 < d.nsrc && k >= d.src[si].C) { k -= d.src[si].C; ++si; }
    const pmf_src_t& sv = d.src[si];
    xp[kt] = sv.x + (size_t)n * P * sv.ldc + k;
    xld[kt] = sv.ldc;
    sc[kt] = valid ? (sv.scale ? sv.scale[k] : 1.f) : 0.f;
    sh[kt] = (valid && sv.scale) ? sv.shift[k] : 0.f;
    lo[kt] = (sv.flags & PMF_SRC_RELU) ? 0.f : -__builtin_inff();
    cm[kt] = valid ? (sv.cmul ? sv.cmul[(size_t)n * sv.cmul_ld + k] : 1.f) : 0.f;
  }
  const __amdgpu_buffer_rsrc_t zrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(d.dz + (size_t)n * P * d.dz_ldc), 0, P * d.dz_ldc * 4, 0x00020000);

  f32x16 acc[KT][NTL];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][nt][r] = 0.f;

  for (int pb = p0 + wave * U; pb < p1; pb += 4 * U) {
    float a[KT][U], b[NTL][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pix = 2 * (pb + u) + lh;
      const bool ok = pb + u < p1 && pix < P;
      const int pc = ok ? pix : 0;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) a[kt][u] = xp[kt][(size_t)pc * xld[kt]];
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt)       // (a channel beyond the pitch reads the next pixel: a column stage 2 never looks at)
        b[nt][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            zrs, ok ? (unsigned)((pix * d.dz_ldc + nt * 32 + li) * 4) : 0xffffffffu, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = pb + u < p1 && 2 * (pb + u) + lh < P;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const float v = ok ? fmaxf(a[kt][u] * sc[kt] + sh[kt], lo[kt]) * cm[kt] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, b[nt][u], acc[kt][nt], 0, 0, 0);
      }
    }
  }

  // fold the four waves in a fixed order, one accumulator tile at a time (12 KiB of LDS); wave 0 writes the slab
  float* part = d.partial + (size_t)split * Ktot * Cout32;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
      if (kt + nt > 0) __syncthreads();
      if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[kt][nt][r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[kt][nt][r];
#pragma unroll
          for (int w = 1; w < 4; ++w) v += red[((w - 1) * 16 + r) * 64 + lane];
          const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (k < Ktot) part[(size_t)k * Cout32 + nt * 32 + li] = v;
        }
      }
    }
}

// conditions of the streaming kernel: one tap at (0, 0), stride 1, operands of the gradient's size, <= 96 input and <= 64
// output channels, >= 32768 pixels
static bool wg_stream(const pmf_wgrad_desc_t* d) {
  if (d->ntaps != 1 || d->gather || d->in_stride != 1 || d->tdy[0] || d->tdx[0]) return false;
  int Ktot = 0;
  for (int i = 0; i < d->nsrc; ++i) {
    const pmf_src_t& s = d->src[i];
    if ((s.flags & PMF_SRC_BCAST) || s.H != d->OH || s.W != d->OW || (s.flags & ~PMF_SRC_RELU)) return false;
    if ((int64_t)d->OH * d->OW * s.ldc * 4 >= (1ll << 31)) return false;
    Ktot += s.C;
  }
  if (Ktot > 96 || d->Cout > 64) return false;
  if ((int64_t)d->OH * d->OW * d->dz_ldc * 4 >= (1ll << 31)) return false;
  return (int64_t)d->N * d->OH * d->OW >= 32768;
}
static int wg_stream_splits(const pmf_wgrad_desc_t* d) {     // S: workgroups per sample
  constexpr int target = 512;
  int S = target / (d->N > 0 ? d->N : 1);
  const int pairs = (d->OH * d->OW + 1) / 2;
  if (S > pairs / 64) S = pairs / 64;           // >= 64 pixel pairs per workgroup
  return S < 1 ? 1 : S;
}

template <int NCO>
__global__ __launch_bounds__(256, 2) void wgrad_1x1_k(const pmf_wgrad_desc_t d, int Ktot, int Cout32) {
  extern __shared__ __attribute__((aligned(16))) float fold[];       // [4 * NCO tiles][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, kb = blockIdx.y, ob = blockIdx.z;
  constexpr int CW = 32 * NCO;
  // this lane's four input channels: operand, offset inside it, transform
  const int kch = kb * 128 + q * 4;
  int si = 0, cin = kch;
  bool kok = kch < Ktot;
  if (kok) while (cin >= d.src[si].C) { cin -= d.src[si].C; ++si; }
  const float* __restrict__ sx = kok ? d.src[si].x + cin : nullptr;
  const int sld = kok ? d.src[si].ldc : 0;
  const bool has_sc = kok && d.src[si].scale != nullptr, relu = kok && (d.src[si].flags & PMF_SRC_RELU);
  const float* __restrict__ scm = kok ? d.src[si].cmul : nullptr;
  const int cm_ld = kok ? d.src[si].cmul_ld : 0;
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  if (has_sc) { sc4 = *(const f32x4*)(d.src[si].scale + cin); sh4 = *(const f32x4*)(d.src[si].shift + cin); }
  const int co = ob * CW + q * NCO;
  const bool cok = co < d.Cout;                                          // Cout % NCO == 0 (host-checked)
  const float* __restrict__ zp = d.dz + co;

  f32x16 acc[4][NCO];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][jj][r] = 0.f;

  const int64_t hw = (int64_t)d.OH * d.OW, npix = hw * d.N, npair = (npix + 1) >> 1;
  const int64_t per = (npair + d.nsplit - 1) / d.nsplit, p_begin = (int64_t)split * per,
                p_end = p_begin + per < npair ? p_begin + per : npair;
  constexpr int U = 4;                                                   // pixel pairs per wave and step
  // Operands of step s+1 are requested before the MFMAs of step s are issued and transformed after them (register
  // double buffer).  The loads are branch-free -- an out-of-range lane reads its operand's first pixel and is zeroed
  // by a select afterwards -- so all of a step's requests are in flight together instead of one round trip per `if`.
  const float* __restrict__ xbase = kok ? sx : d.src[0].x;
  const float* __restrict__ zbase = cok ? zp : d.dz;
  auto issue = [&](int64_t pr0, f32x4 (&xv)[U], float (&zv)[U][NCO], f32x4 (&cv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pr = pr0 + 4 * u, px = 2 * pr + lh;
      const int64_t pxs = (pr < p_end && px < npix) ? px : 0;
      xv[u] = *(const f32x4*)(xbase + pxs * sld);
      if (scm) cv[u] = *(const f32x4*)(scm + (pxs / hw) * cm_ld + cin);
      const float* z = zbase + pxs * d.dz_ldc;
      if constexpr (NCO == 2) { const float2 t = *(const float2*)z; zv[u][0] = t.x; zv[u][1] = t.y; }
      else zv[u][0] = z[0];
    }
  };
  auto finish = [&](int64_t pr0, f32x4 (&xv)[U], float (&zv)[U][NCO], const f32x4 (&cv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pr = pr0 + 4 * u, px = 2 * pr + lh;
      const bool ok = pr < p_end && px < npix;
      f32x4 t = xv[u];
      if (has_sc) t = t * sc4 + sh4;
      if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      if (scm) t = t * cv[u];
      const bool xo = ok && kok, zo = ok && cok;
      xv[u] = f32x4{xo ? t.x : 0.f, xo ? t.y : 0.f, xo ? t.z : 0.f, xo ? t.w : 0.f};
#pragma unroll
      for (int jj = 0; jj < NCO; ++jj) zv[u][jj] = zo ? zv[u][jj] : 0.f;
    }
  };
  f32x4 xa[U], xb[U], ca[U], cb[U];
  float za[U][NCO], zb[U][NCO];
  int64_t pr0 = p_begin + wave;
  issue(pr0, xa, za, ca);
  finish(pr0, xa, za, ca);
  for (; pr0 < p_end; pr0 += 4 * U) {
    issue(pr0 + 4 * U, xb, zb, cb);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int jj = 0; jj < NCO; ++jj)
          acc[j][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u][j], za[u][jj], acc[j][jj], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    finish(pr0 + 4 * U, xb, zb, cb);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xa[u] = xb[u];
#pragma unroll
      for (int jj = 0; jj < NCO; ++jj) za[u][jj] = zb[u][jj];
    }
  }
  // fold waves 1..3 into wave 0 (fixed order), one wave's tiles at a time through LDS
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) fold[((j * NCO + jj) * 16 + r) * 64 + lane] = acc[j][jj][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][jj][r] += fold[((j * NCO + jj) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave) return;
  float* part = d.partial + (size_t)split * Ktot * Cout32;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int k = kb * 128 + 4 * m + j;
#pragma unroll
      for (int jj = 0; jj < NCO; ++jj) {
        const int c = ob * CW + q * NCO + jj;
        if (k < Ktot && c < Cout32) part[(size_t)k * Cout32 + c] = acc[j][jj][r];
      }
    }
}

// The same on the bf16 matrix pipe with split operands (PMF_WGRAD_S3): v_mfma_f32_32x32x16_bf16 wants 8 consecutive k
// (= pixels here) per lane, so a step is a group of 16 pixels: lane (m, g) loads two channels (2m, 2m+1 of the block's
// 64) of X and two of dz for the 8 pixels 8g .. 8g+7 of the group -- sixteen 8-byte loads, 256 contiguous bytes per
// pixel and half wave -- and component j of X against component j' of dz feeds accumulator tile (j, j') whose row m
// stands for channel 2m + j, column n for output channel 2n + j'.  Every pair of pixels is split into its three bf16
// planes in registers (no LDS staging); 24 MFMAs per 16 pixels and wave on a 64 x 64 block.
__global__ __launch_bounds__(256, 2) void wgrad_1x1_s3_k(const pmf_wgrad_desc_t d, int Ktot, int Cout32) {
  extern __shared__ __attribute__((aligned(16))) float fold[];       // [4 tiles][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, kb = blockIdx.y, ob = blockIdx.z;
  const int kch = kb * 64 + q * 2;
  int si = 0, cin = kch;
  const bool kok = kch < Ktot;
  if (kok) while (cin >= d.src[si].C) { cin -= d.src[si].C; ++si; }
  const float* __restrict__ sx = kok ? d.src[si].x + cin : d.src[0].x;
  const int sld = kok ? d.src[si].ldc : 0;
  const bool has_sc = kok && d.src[si].scale != nullptr, relu = kok && (d.src[si].flags & PMF_SRC_RELU);
  const float* __restrict__ scm = kok ? d.src[si].cmul : nullptr;
  const int cm_ld = kok ? d.src[si].cmul_ld : 0;
  float2 sc2 = {1.f, 1.f}, sh2 = {0.f, 0.f};
  if (has_sc) { sc2 = *(const float2*)(d.src[si].scale + cin); sh2 = *(const float2*)(d.src[si].shift + cin); }
  const float relu_lo = relu ? 0.f : -__builtin_inff();
  const int co = ob * 64 + q * 2;
  const bool cok = co < d.Cout;                                          // Cout even (host-checked)
  const float* __restrict__ zp = cok ? d.dz + co : d.dz;

  f32x16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][jj][r] = 0.f;

  const int64_t hw = (int64_t)d.OH * d.OW, npix = hw * d.N, ngrp = (npix + 15) >> 4;
  const int64_t per = (ngrp + d.nsplit - 1) / d.nsplit, g_begin = (int64_t)split * per,
                g_end = g_begin + per < ngrp ? g_begin + per : ngrp;
  // loads of group s+1 are requested before group s is split and multiplied; branch-free (an out-of-range pixel reads
  // pixel 0 and its dz is zeroed afterwards: 0 * finite = 0)
  // stride-2 1x1 layers (the ResNet downsample projections, torchvision Bottleneck / BasicBlock `downsample.0`): output pixel
  // (n, oy, ox) reads input pixel (n, 2 oy, 2 ox) of the operand's own H x W map
  const int istr = d.in_stride, sH = kok ? d.src[si].H : 1, sW = kok ? d.src[si].W : 1;
  auto issue = [&](int64_t grp, float2 (&xv)[8], float2 (&zv)[8], float2 (&cv)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t px = grp * 16 + lh * 8 + e;
      const int64_t pxs = (grp < g_end && px < npix) ? px : 0;
      int64_t pxi = pxs;
      if (istr != 1) {
        const int n = (int)(pxs / hw), rem = (int)(pxs - (int64_t)n * hw), oy = rem / d.OW, ox = rem - oy * d.OW;
        pxi = ((int64_t)n * sH + oy * istr) * sW + ox * istr;
      }
      xv[e] = *(const float2*)(sx + pxi * sld);
      if (scm) cv[e] = *(const float2*)(scm + (pxs / hw) * cm_ld + cin);
      zv[e] = *(const float2*)(zp + pxs * d.dz_ldc);
    }
  };
  auto planes = [&](int64_t grp, const float2 (&xv)[8], const float2 (&zv)[8], const float2 (&cv)[8],
                    wbf16x8 (&a)[2][3], wbf16x8 (&b)[2][3]) {
    wu32x4 ap[2][3], bp[2][3];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      float x[2][2], z[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t px = grp * 16 + lh * 8 + e + h;
        const bool ok = grp < g_end && px < npix;
        float2 t = xv[e + h];
        if (has_sc) { t.x = t.x * sc2.x + sh2.x; t.y = t.y * sc2.y + sh2.y; }
        t.x = ws3_vmax(t.x, relu_lo); t.y = ws3_vmax(t.y, relu_lo);
        if (scm) { t.x *= cv[e + h].x; t.y *= cv[e + h].y; }
        x[0][h] = kok ? t.x : 0.f; x[1][h] = kok ? t.y : 0.f;
        z[0][h] = (ok && cok) ? zv[e + h].x : 0.f; z[1][h] = (ok && cok) ? zv[e + h].y : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unsigned p0, p1, p2;
        ws3_split2(x[j][0], x[j][1], p0, p1, p2);
        ap[j][0][e >> 1] = p0; ap[j][1][e >> 1] = p1; ap[j][2][e >> 1] = p2;
        ws3_split2(z[j][0], z[j][1], p0, p1, p2);
        bp[j][0][e >> 1] = p0; bp[j][1][e >> 1] = p1; bp[j][2][e >> 1] = p2;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) { a[j][p] = __builtin_bit_cast(wbf16x8, ap[j][p]); b[j][p] = __builtin_bit_cast(wbf16x8, bp[j][p]); }
  };
  float2 xa[8], za[8], ca[8], xb[8], zb[8], cb[8];
  int64_t grp = g_begin + wave;
  issue(grp, xa, za, ca);
  for (; grp < g_end; grp += 4) {
    issue(grp + 4, xb, zb, cb);
    wbf16x8 a[2][3], b[2][3];
    planes(grp, xa, za, ca, a, b);
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          acc[j][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j][PA[pr]], b[jj][PB[pr]], acc[j][jj], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) { xa[e] = xb[e]; za[e] = zb[e]; ca[e] = cb[e]; }
  }
  // fold waves 1..3 into wave 0 (fixed order)
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) fold[((j * 2 + jj) * 16 + r) * 64 + lane] = acc[j][jj][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][jj][r] += fold[((j * 2 + jj) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave) return;
  float* part = d.partial + (size_t)split * Ktot * Cout32;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int k = kb * 64 + 2 * m + j;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int c = ob * 64 + q * 2 + jj;
        if (k < Ktot && c < Cout32) part[(size_t)k * Cout32 + c] = acc[j][jj][r];
      }
    }
}

// conditions of the split-bf16 direct 1x1 kernel (64 x 64 output blocks)
static bool wg_direct_s3(const pmf_wgrad_desc_t* d) {
  // smallest map: rounds 2-4 16384 pixels (measured under 256-wide launches); round 5, at the 128-workgroup width of the step:
  // resBlock4.r5 768 -> 256 at 2 x 8 x 256 100 -> 26 us, resBlock4.s 256 -> 256 27 -> 15, upBlock1.e 384 -> 128 27 -> 14,
  // resBlock5.r5 768 -> 256 at 2 x 4 x 128 27 -> 13 -- but 256 -> 256 at 1024 pixels 10 -> 13: from 4096 pixels, or from 1024
  // with >= 512 input channels
  constexpr int min_pix = 0;
  if (!(d->flags & PMF_WGRAD_S3) || ((d->cfg >> 8) & 0xff) == 2) return false;
  if (d->ntaps != 1 || d->gather || (d->in_stride != 1 && d->in_stride != 2) || d->tdy[0] || d->tdx[0] || (d->Cout & 1) || (d->dz_ldc & 1)) return false;
  const int64_t npix = (int64_t)d->N * d->OH * d->OW;
  if (d->Cout < 64) return false;   // full 64 x 64 blocks only (measured: narrower layers are faster on the other kernels)
  for (int i = 0; i < d->nsrc; ++i) {
    const pmf_src_t& s = d->src[i];
    // (stride 2, round 6: the downsample projections -- 78 us at 13.6 TFLOP/s on the fp32 unit-dealing kernel in PMF-ResNet50)
    const bool geo = d->in_stride == 1 ? (s.H == d->OH && s.W == d->OW)
                                       : ((s.H - 1) / 2 + 1 == d->OH && (s.W - 1) / 2 + 1 == d->OW && d->nsrc == 1);
    if ((s.flags & PMF_SRC_BCAST) || !geo || (s.C & 63) || (s.ldc & 1)) return false;
    if (s.flags & ~(PMF_SRC_RELU)) return false;
    if ((int64_t)d->N * s.H * s.W * s.ldc * 4 >= (1ll << 40)) return false;
  }
  if (min_pix > 0) return npix >= min_pix;
  int Ktot = 0;
  for (int i = 0; i < d->nsrc; ++i) Ktot += d->src[i].C;
  return npix >= 4096 || (npix >= 1024 && Ktot >= 512);
}

// conditions of the direct 1x1 kernel + its grid
static bool wg_direct_1x1(const pmf_wgrad_desc_t* d) {
  if (wg_direct_s3(d)) return true;
  if (((d->cfg >> 8) & 0xff) == 2) return false;
  if (d->ntaps != 1 || d->gather || d->in_stride != 1 || d->tdy[0] || d->tdx[0] || (d->Cout & 1) || (d->dz_ldc & 1)) return false;
  for (int i = 0; i < d->nsrc; ++i) {
    const pmf_src_t& s = d->src[i];
    if ((s.flags & PMF_SRC_BCAST) || s.H != d->OH || s.W != d->OW || (s.C & 3) || (s.ldc & 3)) return false;
    if (s.flags & ~(PMF_SRC_RELU)) return false;
  }
  int Ktot = 0;
  for (int i = 0; i < d->nsrc; ++i) Ktot += d->src[i].C;
  if (Ktot < 96) return false;                       // a lane owns 4 of 128 channels: narrow operands idle most lanes
  if (d->Cout > 128 && Ktot < 256) return false;     // X is re-read once per 64 output channels
  // measured (64x2048, bs 2): 192->64 158 -> 119 us, 384->128 115 -> 92, 768->256 111 -> 95; below ~16 k pixels the
  // tiled kernels win (fewer, larger tiles; the per-workgroup fold and slab dominate here)
  constexpr int min_pix1 = 16384;
  return (int64_t)d->N * d->OH * d->OW >= min_pix1;
}
static void wg_direct_grid(const pmf_wgrad_desc_t* d, int* kblocks, int* oblocks, int* nco) {
  int Ktot = 0;
  for (int i = 0; i < d->nsrc; ++i) Ktot += d->src[i].C;
  if (wg_direct_s3(d)) {               // 64 x 64 blocks
    *nco = 2; *kblocks = cdiv(Ktot, 64); *oblocks = cdiv(round_up(d->Cout, 32), 64);
    return;
  }
  *nco = d->Cout > 32 ? 2 : 1;
  *kblocks = cdiv(Ktot, 128);
  *oblocks = cdiv(round_up(d->Cout, 32), 32 * *nco);
}

// stage 2: dw_oihw[(co*Cin_real + k)*KHW + widx[t]] (+)= sum_s partial[s][t][k][co]
// 256 threads = 32 consecutive outputs x 8 split slices (independent, unrolled loads), folded through LDS in a
// fixed order -> deterministic, and no thread walks hundreds of slabs serially.
__device__ __forceinline__ void red_flat_body(const pmf_wgrad_desc_t& d, int Ktot, int Cout32, int bid, int nblk) {
  __shared__ double shr[8][32];      // (float64 fold of the slabs, see red_tile_body)
  const int64_t total = (int64_t)d.ntaps * Ktot * Cout32;
  const int64_t slab = total;
  const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
  for (int64_t base = (int64_t)bid * 32; base < total; base += (int64_t)nblk * 32) {
    const int64_t i = base + ol;
    double sd = 0.0;
    if (i < total) {
      for (int sp0 = sl; sp0 < d.nsplit; sp0 += 64) {   // 8 independent loads in flight per thread, fixed summation order
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = sp0 + 8 * j < d.nsplit ? d.partial[(int64_t)(sp0 + 8 * j) * slab + i] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sd += (double)t[j];
      }
    }
    shr[sl][ol] = sd;
    __syncthreads();
    if (sl == 0 && i < total) {
      const float s = (float)(((shr[0][ol] + shr[1][ol]) + (shr[2][ol] + shr[3][ol])) + ((shr[4][ol] + shr[5][ol]) + (shr[6][ol] + shr[7][ol])));
      const int co = (int)(i % Cout32);
      const int64_t r = i / Cout32;
      const int k = (int)(r % Ktot), t = (int)(r / Ktot);
      if (co < d.Cout && k < d.Cin_real) {
        float* o = d.dw_oihw + ((size_t)co * d.Cin_real + k) * d.KHW + d.tap_widx[t];
        *o = d.accumulate ? *o + s : s;
      }
    }
    __syncthreads();
  }
  // conv-bias gradient: fold the partial column sums of dz left by the kernel that produced dz
  // (one output channel per workgroup round, rows split over the threads, LDS tree)
  if (d.dbias_rows) {
    __shared__ double shb[256];
    for (int co = bid; co < d.Cout; co += nblk) {
      double s = 0.0;
      for (int r = threadIdx.x; r < d.dbias_nrows; r += blockDim.x) s += (double)d.dbias_rows[(size_t)r * d.dbias_ld + co];
      shb[threadIdx.x] = s;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shb[threadIdx.x] += shb[threadIdx.x + o];
        __syncthreads();
      }
      if (threadIdx.x == 0) d.dbias_out[co] += (float)shb[0];
      __syncthreads();
    }
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_k(const pmf_wgrad_desc_t d, int Ktot, int Cout32) {
  red_flat_body(d, Ktot, Cout32, blockIdx.x, gridDim.x);
}

// Stage 2 for layers with many weights: the slabs are [split][tap][k][co32] (co fastest) while the gradient is OIHW
// (tap fastest), so a thread-per-output reduction scatters 4-byte writes one weight row apart.  Here a workgroup owns
// 32 output channels x KB input channels x all taps: float4 loads along co (every split summed by the same thread, in
// slab order), a transpose through LDS, then runs of KB*KHW contiguous floats per output channel.
#define WGR_ROWS 128
__device__ __forceinline__ void red_tile_body(const pmf_wgrad_desc_t& d, int Ktot, int Cout32, int KB, int bx, int by,
                                              int nbx, int nby) {
  __shared__ float tile[WGR_ROWS][33];
  const int T = d.ntaps, RB = T * KB;
  const int co0 = bx * 32, k0 = by * KB;
  const int q = threadIdx.x & 7, rs = threadIdx.x >> 3;
  const int64_t slab = (int64_t)T * Ktot * Cout32;
  for (int j = rs; j < RB; j += 32) {
    const int kk = j / T, t = j - kk * T, k = k0 + kk;
    // the slabs are folded in float64 (round 5; the kernel is bound by reading them): the only fp32 accumulation chain left
    // in a weight gradient is the one inside a partial slab
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (k < Ktot) {
      const float* src = d.partial + ((int64_t)t * Ktot + k) * Cout32 + co0 + q * 4;
      for (int s0 = 0; s0 < d.nsplit; s0 += 8) {
        f32x4 u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) if (s0 + i < d.nsplit) u[i] = *(const f32x4*)(src + (int64_t)(s0 + i) * slab);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (s0 + i < d.nsplit) { v[0] += (double)u[i].x; v[1] += (double)u[i].y; v[2] += (double)u[i].z; v[3] += (double)u[i].w; }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[j][q * 4 + i] = (float)v[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int cl = w; cl < 32; cl += 4) {
    const int co = co0 + cl;
    if (co >= d.Cout) break;
    for (int j = lane; j < RB; j += 64) {
      const int kk = j / T, t = j - kk * T, k = k0 + kk;
      if (k >= d.Cin_real) continue;
      float* o = d.dw_oihw + ((size_t)co * d.Cin_real + k) * d.KHW + d.tap_widx[t];
      const float x = tile[j][cl];
      *o = d.accumulate ? *o + x : x;
    }
  }
  if (d.dbias_rows) {
    __shared__ double shb[256];
    const int nb = nbx * nby;
    for (int co = by * nbx + bx; co < d.Cout; co += nb) {
      double s = 0.0;
      for (int r = threadIdx.x; r < d.dbias_nrows; r += blockDim.x) s += (double)d.dbias_rows[(size_t)r * d.dbias_ld + co];
      shb[threadIdx.x] = s;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shb[threadIdx.x] += shb[threadIdx.x + o];
        __syncthreads();
      }
      if (threadIdx.x == 0) d.dbias_out[co] += (float)shb[0];
      __syncthreads();
    }
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_tile_k(const pmf_wgrad_desc_t d, int Ktot, int Cout32, int KB) {
  red_tile_body(d, Ktot, Cout32, KB, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// Stage 2 of MANY layers in one launch (a training plan queues the reductions of a run of layers: every kernel of a
// replayed graph costs >= 3.6 us, 110 reductions per iteration are mostly that).  jobs: device copies of the layers'
// descriptors; meta[j] = {first workgroup, kind (0 flat / 1 tiled), Ktot, Cout32, KB, grid x, grid y}.  Each workgroup
// finds its job by bisection and runs the same body as the single-layer kernels (bit-identical results).
struct RedMeta { int32_t block_start, kind, Ktot, Cout32, KB, gx, gy, pad_; };
__global__ __launch_bounds__(256) void wgrad_reduce_multi_k(const pmf_wgrad_desc_t* __restrict__ jobs,
                                                            const RedMeta* __restrict__ meta, int njobs) {
  int lo = 0, hi = njobs - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (meta[mid].block_start <= b) lo = mid; else hi = mid - 1;
  }
  const RedMeta m = meta[lo];
  const int lb = b - m.block_start;
  if (m.kind == 1) red_tile_body(jobs[lo], m.Ktot, m.Cout32, m.KB, lb % m.gx, lb / m.gx, m.gx, m.gy);
  else red_flat_body(jobs[lo], m.Ktot, m.Cout32, lb, m.gx);
}

static int wg_geometry(const pmf_wgrad_desc_t* d, int TB, int BN, WgGeom* g, int* lds) {
  int Ktot = 0, nchunks = 0;
  for (int i = 0; i < d->nsrc; ++i) { Ktot += d->src[i].C; nchunks += cdiv(d->src[i].C, WG_CI); }
  g->Ktot = Ktot; g->nchunks = nchunks;
  g->Cout32 = round_up(d->Cout, 32);
  g->tiles_x = cdiv(d->OW, 32); g->tiles_y = cdiv(d->OH, WG_ROWS);
  g->total_tiles = g->tiles_x * g->tiles_y * d->N;
  int dy_min = 127, dy_max = -127, dx_min = 127, dx_max = -127;
  for (int t = 0; t < d->ntaps; ++t) {
    dy_min = d->tdy[t] < dy_min ? d->tdy[t] : dy_min; dy_max = d->tdy[t] > dy_max ? d->tdy[t] : dy_max;
    dx_min = d->tdx[t] < dx_min ? d->tdx[t] : dx_min; dx_max = d->tdx[t] > dx_max ? d->tdx[t] : dx_max;
  }
  g->dy_min = dy_min; g->dx_min = dx_min;
  int rows = (WG_ROWS - 1) * d->in_stride + 1, cols = 31 * d->in_stride + 1;
  if (!d->gather) { rows += dy_max - dy_min; cols += dx_max - dx_min; }
  g->in_rows = rows; g->in_cols = cols;
  g->x_floats = rows * cols * WG_CI;
  g->co_tiles = cdiv(d->Cout, BN);
  g->tap_batches = cdiv(d->ntaps, TB);
  *lds = (g->x_floats + WG_ROWS * 32 * BN) * 4;
  return 0;
}

// conditions of the software-pipelined kernel
// cmod: operand channel counts must be multiples of it (32; the split-bf16 kernel also takes half-empty chunks: 16)
// ragged: Cout need not fill its last 32-channel tile (only the N-split kernel, which reads dz straight from global memory:
// a lane beyond Cout reads the next pixel's channels -- or the buffer's hardware zero -- into a column of the partial slab
// that stage 2 never looks at)
static bool wg_simple(const pmf_wgrad_desc_t* d, const WgGeom& g, int TB, int BN, int cmod = WG_CI, bool ragged = false) {
  if (d->gather || d->in_stride != 1 || d->ntaps != TB) return false;
  if (!ragged && (d->OH % WG_ROWS || d->OW % 32 || d->Cout % BN)) return false;    // (ragged: also partial 4 x 32-pixel tiles)
  if (g.in_rows * g.in_cols * 8 > 256 * 9 || g.in_cols > 255) return false;
  for (int i = 0; i < d->nsrc; ++i) {
    if (d->src[i].C % cmod || (d->src[i].flags & PMF_SRC_BCAST)) return false;
    if (d->src[i].H != d->OH || d->src[i].W != d->OW) return false;
    if ((int64_t)d->N * d->OH * d->OW * d->src[i].ldc * 4 >= (1ll << 31)) return false;
  }
  return true;
}

// taps per workgroup (TB) and output-channel tiles (NT): 3x3 -> 9 taps, 2x2 -> 4, per-tap staging -> 1.
// Pipelined kernel (when its conditions hold): NT = 1 -- every wave carries all TB taps of one 32-channel tile, the
// narrow tile keeps the split count (hence the partial-slab traffic) low and measured fastest at every resolution.
// Unit-dealing kernel: 64 output channels per workgroup (2 waves/SIMD fit; 128 measured slower).
static int wg_s3n_nco(const pmf_wgrad_desc_t* d, const WgGeom& g, int TB);
static void wg_config(const pmf_wgrad_desc_t* d, int* TB, int* NT) {
  if (d->gather || d->ntaps == 1) *TB = 1;
  else if (d->ntaps <= 4) *TB = 4;
  else *TB = 9;
  WgGeom g;
  int lds;
  wg_geometry(d, *TB, 32, &g, &lds);
  // 1x1 convolutions with wide outputs are plain GEMMs: one MFMA per operand pair either way, so the wider tile of
  // the unit-dealing kernel (fewer re-reads of the input tile) wins there (measured 123 vs 164 us on 384 -> 128)
  const bool wide_1x1 = d->ntaps == 1 && d->Cout > 64;
  const int cmod = (d->flags & PMF_WGRAD_S3) ? 16 : WG_CI;
  if (d->cfg) {                        // caller-tuned
    int nt = d->cfg & 0xff;
    const int kern = (d->cfg >> 8) & 0xff;
    if (nt != 1 && nt != 2 && nt != 4) nt = 1;
    if (nt == 4 && *TB != 1) nt = 2;                        // 128-wide tiles are only built for per-tap staging
    while (nt > 1 && (nt - 1) * 32 >= d->Cout) nt >>= 1;
    if (kern == 1 && wg_simple(d, g, *TB, 32, cmod)) nt = 1;
    *NT = nt;
    return;
  }
  if (!wide_1x1 && wg_simple(d, g, *TB, 32, cmod)) { *NT = 1; return; }
  if (!wide_1x1 && wg_s3n_nco(d, g, *TB) > 0) { *NT = 1; return; }   // ragged tiles / last channel tile
  *NT = wide_1x1 ? 4 : (d->Cout > 32 ? 2 : 1);
}

// output-channel tiles per workgroup of the N-split split-bf16 kernel (conv_wgrad_s3n_k): 4 / 2, or 0 = not this kernel.
// PMF_WG_S3N=0 switches it off (A/B), =2 caps it at two tiles.
static int wg_s3n_nco(const pmf_wgrad_desc_t* d, const WgGeom& g, int TB) {
  const char* e_n = getenv("PMF_WG_S3N");     // (read per call, not cached: the tests switch variants)
  const int mode = e_n ? atoi(e_n) : 4;
  // (operands of 8 channels too -- EPMF's 3x3 5 -> 32 first layer, padded to 8: a quarter-full chunk; the rows of the 32-row
  // tile beyond the operand's channels are staged as zeros and never written to the slab)
  if (mode <= 1 || TB <= 1 || !(d->flags & PMF_WGRAD_S3) || !wg_simple(d, g, TB, 32, 8, true)) return 0;
  if ((int64_t)d->N * d->OH * d->OW * d->dz_ldc * 4 >= (1ll << 31)) return 0;
  const char* e_w8 = getenv("PMF_WG_W8");
  const char* e_swp = getenv("PMF_WG_SWP");
  if ((e_w8 && e_w8[0] == '1') || (e_swp && e_swp[0] == '0')) return 0;        // the test switches for the older variants
  if (d->Cout % 32) return mode == 3 ? 0 : 1;          // ragged last tile: one tile per workgroup
  if (mode >= 4 && d->Cout % 128 == 0) return 4;
  if (d->Cout % 64 == 0) return 2;
  return mode == 3 ? 0 : 1;       // (3: 32-channel tiles on the round-3 software-pipelined kernel)
}

static bool wg_fewc(const pmf_wgrad_desc_t* d);
extern "C" int pmf_conv_wgrad_nsplit(const pmf_wgrad_desc_t* d) {
  if (wg_fewc(d)) {   // one output-channel tile pair per workgroup: split the pixel tiles 256 ways
    const int tiles = cdiv(d->OW, 32) * cdiv(d->OH, WG_ROWS) * d->N, ns = 256 / cdiv(d->Cout, 64);
    return tiles < ns ? tiles : (ns < 1 ? 1 : ns);
  }
  if (wg_stream(d)) return d->N * wg_stream_splits(d);
  if (wg_direct_1x1(d)) {  // two resident workgroups per CU in total; every workgroup gets >= 64 pixel pairs
    int kb, ob, nco;
    wg_direct_grid(d, &kb, &ob, &nco);
    constexpr int dtarget = 512;
    int ns = dtarget / (kb * ob);
    const int64_t pairs = ((int64_t)d->N * d->OH * d->OW + 1) / 2;
    if (ns > pairs / 64) ns = (int)(pairs / 64);
    return ns < 1 ? 1 : ns;
  }
  int TB, NT, lds;
  WgGeom g;
  wg_config(d, &TB, &NT);
  wg_geometry(d, TB, NT * 32, &g, &lds);
  int other = g.nchunks * g.co_tiles * g.tap_batches;
  if (NT == 1) { const int nco = wg_s3n_nco(d, g, TB); if (nco) other = g.nchunks * cdiv(d->Cout, 32 * nco); }
  // workgroups per launch: one per TWO CUs.  More make the launch itself faster in isolation (512: 4.8 ms over the 110 layers,
  // 256: 5.4 ms), but the weight gradients run on side lanes next to the input-gradient launches of the main lane -- which IS
  // the step -- and a weight-gradient workgroup holds 80-110 KiB of its CU's LDS: with one on every CU a main-lane conv
  // launch gets one workgroup per CU instead of the two it is tuned for.  Step time, alternating runs on one box (round 4):
  // 64 workgroups 17.97 ms, 96: 16.25, 128: 15.13-15.17, 160: 15.79, 256: 15.41-15.48, 320: 16.40 (512, round 2: +0.1 over 256);
  // PMF-ResNet50 32x1024 10.07 -> 9.79 ms, EPMF 14.66 -> 14.61, SalsaNext 11.33 -> 11.22.  Half the partial slabs is also half
  // the stage-2 traffic.  PMF_WGRAD_WGS overrides.
  static const int target = getenv("PMF_WGRAD_WGS") ? atoi(getenv("PMF_WGRAD_WGS")) : 128;
  int ns = target / (other > 0 ? other : 1);
  if (ns < 1) ns = 1;
  if (ns > g.total_tiles) ns = g.total_tiles;
  return ns;
}

extern "C" int64_t pmf_conv_wgrad_workspace(const pmf_wgrad_desc_t* d) {
  int Ktot = 0;
  for (int i = 0; i < d->nsrc; ++i) Ktot += d->src[i].C;
  return (int64_t)d->nsplit * d->ntaps * Ktot * round_up(d->Cout, 32) * 4;
}

// which stage-2 kernel and grid a layer gets
static void red_plan(const pmf_wgrad_desc_t* d, int Ktot, int Cout32, int* kind, int* KB, int* gx, int* gy) {
  const int64_t total = (int64_t)d->ntaps * Ktot * Cout32;
  if (total >= 131072 && d->nsplit <= 32 && d->ntaps <= WGR_ROWS) {
    int kb = WGR_ROWS / d->ntaps;
    if (kb > Ktot) kb = Ktot;
    const int cog = Cout32 / 32;
    while (kb > 4 && cog * cdiv(Ktot, kb) < 512) kb = (kb + 1) / 2;
    *kind = 1; *KB = kb; *gx = cog; *gy = cdiv(Ktot, kb);
    return;
  }
  const int gb = (int)cdiv64(total, 32);
  *kind = 0; *KB = 0; *gx = gb > 4096 ? 4096 : gb; *gy = 1;
}
static int wg_reduce(const pmf_wgrad_desc_t* d, const WgGeom& g, hipStream_t s) {
  int kind, KB, gx, gy;
  red_plan(d, g.Ktot, g.Cout32, &kind, &KB, &gx, &gy);
  if (kind == 1) hipLaunchKernelGGL(wgrad_reduce_tile_k, dim3(gx, gy), dim3(256), 0, s, *d, g.Ktot, g.Cout32, KB);
  else hipLaunchKernelGGL(wgrad_reduce_k, dim3(gx), dim3(256), 0, s, *d, g.Ktot, g.Cout32);
  PMF_LAUNCH_CHECK();
  return 0;
}
extern "C" int pmf_conv_wgrad_reduce_plan(const pmf_wgrad_desc_t* d, int32_t* meta8) {
  if (!d || !meta8 || d->nsrc < 1 || d->nsrc > PMF_MAX_SRC || d->ntaps < 1 || d->ntaps > PMF_MAX_TAPS || d->nsplit < 1)
    return PMF_E_ARG;
  int Ktot = 0;
  for (int i = 0; i < d->nsrc; ++i) Ktot += d->src[i].C;
  const int Cout32 = round_up(d->Cout, 32);
  int kind, KB, gx, gy;
  red_plan(d, Ktot, Cout32, &kind, &KB, &gx, &gy);
  meta8[0] = 0; meta8[1] = kind; meta8[2] = Ktot; meta8[3] = Cout32; meta8[4] = KB; meta8[5] = gx; meta8[6] = gy; meta8[7] = 0;
  return gx * gy;
}
extern "C" int pmf_conv_wgrad_reduce_multi(const pmf_wgrad_desc_t* jobs_dev, const int32_t* meta_dev, int32_t njobs,
                                           int32_t total_blocks, pmf_stream_t s) {
  if (!jobs_dev || !meta_dev || njobs < 1 || total_blocks < 1) return PMF_E_ARG;
  hipLaunchKernelGGL(wgrad_reduce_multi_k, dim3(total_blocks), dim3(256), 0, (hipStream_t)s, jobs_dev,
                     (const RedMeta*)meta_dev, njobs);
  PMF_LAUNCH_CHECK();
  return 0;
}

template <int TB, int NT>
static int wg_launch(const pmf_wgrad_desc_t* d, hipStream_t s, int phase) {
  WgGeom g;
  int lds;
  wg_geometry(d, TB, NT * 32, &g, &lds);
  if (lds > 160 * 1024) return PMF_E_UNSUPPORTED;
  if (!(phase & 1)) return wg_reduce(d, g, s);
  static unsigned long long attr_set = 0ull;
  if (pmf_first_on_device(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_k<TB, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (NT == 1) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad_pipe_k<TB, 1, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_wgrad_pipe_k<TB, 1, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
  }
  dim3 grid(d->nsplit, g.nchunks, g.co_tiles * g.tap_batches);
  bool piped = false;
  if constexpr (NT == 1) {
    if ((d->flags & PMF_WGRAD_S3) && (wg_simple(d, g, TB, 32, 16) || wg_s3n_nco(d, g, TB))) {
      static unsigned long long attr3 = 0ull;
      if (pmf_first_on_device(&attr3)) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_s3_k<TB, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_s3_k<TB, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      }
      g.x_floats = g.in_rows * g.in_cols * (WS3_XPB / 4);
      int lds3 = (g.x_floats + WG_ROWS * 32 * 32) * 4;
      if (lds3 < 16 * 1024) lds3 = 16 * 1024;
      // (read per launch, not cached: the tests switch variants; a replayed graph never comes here)
      const char* e_swp = getenv("PMF_WG_SWP");
      const char* e_w8 = getenv("PMF_WG_W8");
      const bool swp = !(e_swp && e_swp[0] == '0');
      const bool small7 = g.in_rows * g.in_cols * 8 <= 256 * 7;
      const bool w8 = e_w8 && e_w8[0] == '1';
      const int nco = wg_s3n_nco(d, g, TB);
      if (nco) {
        if constexpr (TB > 1) {
          static unsigned long long attr6 = 0ull;
          if (pmf_first_on_device(&attr6)) {
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 7, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 9, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 7, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 9, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 7, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 9, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 7, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 9, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 7, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 9, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 7, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3n_k<TB, 9, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          }
          g.x_floats += WS3_XPB / 4;           // the spare pixel that slots beyond the tile write to
          int lds6 = 2 * g.x_floats * 4;
          if (lds6 < 16 * 1024) lds6 = 16 * 1024;
          const dim3 grid6(d->nsplit, g.nchunks, cdiv(d->Cout, 32 * nco));
          const bool rag = d->OH % WG_ROWS || d->OW % 32;
#define S3N_GO(xsl, nc, rg) hipLaunchKernelGGL((conv_wgrad_s3n_k<TB, xsl, nc, rg>), grid6, dim3(256), lds6, s, *d, g)
#define S3N_PICK(nc) do { if (rag) { if (small7) S3N_GO(7, nc, true); else S3N_GO(9, nc, true); } \
                          else { if (small7) S3N_GO(7, nc, false); else S3N_GO(9, nc, false); } } while (0)
          if (nco == 4) S3N_PICK(4);
          else if (nco == 2) S3N_PICK(2);
          else S3N_PICK(1);
#undef S3N_PICK
#undef S3N_GO
        }
      } else
      // eight waves (the taps of a slab on two waves, two waves per SIMD): 3-10 % faster launch by launch, but 110 KiB
      // of LDS and 512 threads leave no room for the input-gradient launches the weight gradients run next to:
      // 16.08 vs 15.97 ms per step -- off unless PMF_WG_W8=1
      if (swp && w8 && TB > 1) {
        if constexpr (TB > 1) {
          static unsigned long long attr5 = 0ull;
          if (pmf_first_on_device(&attr5)) {
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3_w8_k<TB, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_s3_w8_k<TB, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          }
          g.x_floats += WS3_XPB / 4;           // the spare pixel that slots beyond the tile write to
          const int lds5 = (2 * g.x_floats + 3 * WG_ROWS * 32 * 32) * 4;
          if (g.in_rows * g.in_cols * 8 <= 512 * 4) hipLaunchKernelGGL((conv_wgrad_s3_w8_k<TB, 4>), grid, dim3(512), lds5, s, *d, g);
          else hipLaunchKernelGGL((conv_wgrad_s3_w8_k<TB, 5>), grid, dim3(512), lds5, s, *d, g);
        }
      } else if (swp) {      // input tile double-buffered: tile t + 1 is split while tile t is multiplied
        static unsigned long long attr4 = 0ull;
        if (pmf_first_on_device(&attr4)) {
          (void)hipFuncSetAttribute((const void*)conv_wgrad_s3_swp_k<TB, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void)hipFuncSetAttribute((const void*)conv_wgrad_s3_swp_k<TB, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        g.x_floats += WS3_XPB / 4;           // the spare pixel that slots beyond the tile write to
        const int lds4 = (2 * g.x_floats + WG_ROWS * 32 * 32) * 4;
        if (small7) hipLaunchKernelGGL((conv_wgrad_s3_swp_k<TB, 7>), grid, dim3(256), lds4, s, *d, g);
        else hipLaunchKernelGGL((conv_wgrad_s3_swp_k<TB, 9>), grid, dim3(256), lds4, s, *d, g);
      } else if (small7) hipLaunchKernelGGL((conv_wgrad_s3_k<TB, 7>), grid, dim3(256), lds3, s, *d, g);
      else hipLaunchKernelGGL((conv_wgrad_s3_k<TB, 9>), grid, dim3(256), lds3, s, *d, g);
      piped = true;
    } else if (wg_simple(d, g, TB, 32) && ((d->cfg >> 8) & 0xff) != 2) {
      const int lds2 = lds < 16 * 1024 ? 16 * 1024 : lds;   // room for the pixel-group reduction
      if (g.in_rows * g.in_cols * 8 <= 256 * 7) hipLaunchKernelGGL((conv_wgrad_pipe_k<TB, 1, 7>), grid, dim3(256), lds2, s, *d, g);
      else hipLaunchKernelGGL((conv_wgrad_pipe_k<TB, 1, 9>), grid, dim3(256), lds2, s, *d, g);
      piped = true;
    }
  }
  if (!piped) hipLaunchKernelGGL((conv_wgrad_k<TB, NT>), grid, dim3(256), lds, s, *d, g);
  PMF_LAUNCH_CHECK();
  return (phase & 2) ? wg_reduce(d, g, s) : 0;
}

// few-input-channel path (ResNet stem): conditions
static bool wg_fewc(const pmf_wgrad_desc_t* d) {
  if (d->nsrc != 1 || d->in_stride != 1 || d->Cin_real > 4 || d->ntaps < 9) return false;
  if (d->ntaps * d->Cin_real > 160 || d->Cout % 64 || d->OH % WG_ROWS || d->OW % 32) return false;
  if (d->src[0].flags || d->src[0].scale || d->src[0].cmul) return false;        // raw input only
  if (d->src[0].H != d->OH || d->src[0].W != d->OW) return false;
  return true;
}

static void wg_geometry_fewc(const pmf_wgrad_desc_t* d, WgGeom* g, int* lds) {
  int l0;
  wg_geometry(d, 1, 64, g, &l0);
  int dy_min = 127, dy_max = -127, dx_min = 127, dx_max = -127;
  for (int t = 0; t < d->ntaps; ++t) {
    dy_min = d->tdy[t] < dy_min ? d->tdy[t] : dy_min; dy_max = d->tdy[t] > dy_max ? d->tdy[t] : dy_max;
    dx_min = d->tdx[t] < dx_min ? d->tdx[t] : dx_min; dx_max = d->tdx[t] > dx_max ? d->tdx[t] : dx_max;
  }
  g->in_rows = WG_ROWS + dy_max - dy_min; g->in_cols = 32 + dx_max - dx_min;   // always a halo tile here
  g->dy_min = dy_min; g->dx_min = dx_min;
  g->co_tiles = d->Cout / 64; g->tap_batches = 1;
  const int xfl = (g->in_rows * g->in_cols * d->Cin_real + 4 + 3) & ~3;
  *lds = (xfl + 2 * 64 * 64) * 4;
  if (*lds < 16 * 1024) *lds = 16 * 1024;
}

static int wg_launch_fewc(const pmf_wgrad_desc_t* d, hipStream_t s, int phase) {
  WgGeom g;
  int lds;
  wg_geometry_fewc(d, &g, &lds);
  if (!(phase & 1)) return wg_reduce(d, g, s);
  const int KGn = cdiv(d->ntaps * d->Cin_real, 32);
  static unsigned long long attr_set = 0ull;
  if (pmf_first_on_device(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)wgrad_fewc_k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)wgrad_fewc_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)wgrad_fewc_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)wgrad_fewc_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  dim3 grid(d->nsplit, 1, g.co_tiles);
  if (KGn <= 1) hipLaunchKernelGGL((wgrad_fewc_k<1>), grid, dim3(256), lds, s, *d, g);
  else if (KGn == 2) hipLaunchKernelGGL((wgrad_fewc_k<2>), grid, dim3(256), lds, s, *d, g);
  else if (KGn == 3) hipLaunchKernelGGL((wgrad_fewc_k<3>), grid, dim3(256), lds, s, *d, g);
  else hipLaunchKernelGGL((wgrad_fewc_k<5>), grid, dim3(256), lds, s, *d, g);
  PMF_LAUNCH_CHECK();
  return (phase & 2) ? wg_reduce(d, g, s) : 0;
}

static int wgrad_phases(const pmf_wgrad_desc_t* d, pmf_stream_t st, int phase);
extern "C" int pmf_conv_wgrad(const pmf_wgrad_desc_t* d, pmf_stream_t st) { return wgrad_phases(d, st, 3); }
// the two stages separately: the partial-slab kernel, then the deterministic reduction into OIHW (+ bias fold).  A plan
// runs the reduction on its side stream: nothing downstream needs it before the optimiser (or the gradient all-reduce)
extern "C" int pmf_conv_wgrad_partial(const pmf_wgrad_desc_t* d, pmf_stream_t st) { return wgrad_phases(d, st, 1); }
extern "C" int pmf_conv_wgrad_reduce(const pmf_wgrad_desc_t* d, pmf_stream_t st) { return wgrad_phases(d, st, 2); }

static int wgrad_phases(const pmf_wgrad_desc_t* d, pmf_stream_t st, int phase) {
  hipStream_t s = (hipStream_t)st;
  if (!d || d->nsrc < 1 || d->nsrc > PMF_MAX_SRC || d->ntaps < 1 || d->ntaps > PMF_MAX_TAPS || d->nsplit < 1)
    return PMF_E_ARG;
  for (int i = 0; i < d->nsrc; ++i)
    if (d->src[i].C % 8 || d->src[i].ldc % 4) return PMF_E_ARG;
  if (d->gather && false) return PMF_E_ARG;
  if (wg_fewc(d)) return wg_launch_fewc(d, s, phase);
  if (wg_stream(d)) {
    WgGeom g;
    int lds;
    wg_geometry(d, 1, 32, &g, &lds);
    if (phase & 1) {
      if (d->nsplit % d->N) return PMF_E_ARG;               // (pmf_conv_wgrad_nsplit: N x S workgroups, one sample each)
      const int S = d->nsplit / d->N, KT = cdiv(g.Ktot, 32), NTL = g.Cout32 / 32;
      const dim3 grid(d->nsplit);
#define WS_CASE(kt, ntl) if (KT == kt && NTL == ntl) hipLaunchKernelGGL((wgrad_stream_k<kt, ntl>), grid, dim3(256), 0, s, *d, g.Ktot, g.Cout32, S)
      WS_CASE(1, 1); else WS_CASE(1, 2); else WS_CASE(2, 1); else WS_CASE(2, 2); else WS_CASE(3, 1); else WS_CASE(3, 2);
      else return PMF_E_UNSUPPORTED;
#undef WS_CASE
      PMF_LAUNCH_CHECK();
    }
    return (phase & 2) ? wg_reduce(d, g, s) : 0;
  }
  if (wg_direct_1x1(d)) {
    WgGeom g;
    int lds, kb, ob, nco;
    wg_geometry(d, 1, 32, &g, &lds);
    if (phase & 1) {
      if (((uintptr_t)d->dz & 7) != 0) return PMF_E_ARG;       // float2 loads of dz
      wg_direct_grid(d, &kb, &ob, &nco);
      static unsigned long long attr_set = 0ull;
      if (pmf_first_on_device(&attr_set)) {
        (void)hipFuncSetAttribute((const void*)wgrad_1x1_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_1x1_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      }
      const dim3 grid(d->nsplit, kb, ob);
      if (wg_direct_s3(d)) hipLaunchKernelGGL(wgrad_1x1_s3_k, grid, dim3(256), 4 * 16 * 64 * 4, s, *d, g.Ktot, g.Cout32);
      else if (nco == 2) hipLaunchKernelGGL(wgrad_1x1_k<2>, grid, dim3(256), 8 * 16 * 64 * 4, s, *d, g.Ktot, g.Cout32);
      else hipLaunchKernelGGL(wgrad_1x1_k<1>, grid, dim3(256), 4 * 16 * 64 * 4, s, *d, g.Ktot, g.Cout32);
      PMF_LAUNCH_CHECK();
    }
    return (phase & 2) ? wg_reduce(d, g, s) : 0;
  }
  int TB, NT;
  wg_config(d, &TB, &NT);
#define WG_CASE(tb, nt) if (TB == tb && NT == nt) return wg_launch<tb, nt>(d, s, phase)
  WG_CASE(9, 1); WG_CASE(9, 2);
  WG_CASE(4, 1); WG_CASE(4, 2);
  WG_CASE(1, 1); WG_CASE(1, 2); WG_CASE(1, 4);
#undef WG_CASE
  return PMF_E_UNSUPPORTED;
}
