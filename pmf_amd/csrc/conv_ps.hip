// Convolution on the bf16 matrix pipe with PRE-SPLIT operands: every byte the MFMAs read arrives in LDS by DMA.  gfx950 only.
//
// Same arithmetic as the split loops of conv_fwd.hip (fp32 = h1 + h2 + h3 in bf16, six v_mfma_f32_32x32x16_bf16 products
// per fp32 product, fp32 accumulate: fp32-class error), same output tile / accumulator layout / epilogue (conv_epi.h), but
// the input operands come as the three bf16 planes pmf_presplit wrote ([plane][c/8][pixel][8], view applied, split once
// per tensor instead of once per output-channel tile and use).  The K loop therefore has NO vector-ALU work and no
// register staging: per 16-channel stage the halo tile (6 x 16-byte slots per pixel: plane x channel half) and the weight
// fragments go global -> LDS with buffer_load ... lds / global_load_lds, and the waves only read fragments and multiply.
//   LDS input tile  [slab][plane][channel half][tile pixel] x 16 B: the 32 lanes of an MFMA row read 32 consecutive
//                   slots (conflict-free ds_read_b128); out-of-image pixels are zero-filled by the buffer range check;
//   LDS weights     [virtual tap][NT][plane] x 1 KiB fragments (pack format 1), two tap halves that ping-pong;
//   pipeline        stage s: wait+barrier | DMA B(s, half 1), DMA A(s+1) | MFMA half 0 | wait B + barrier |
//                   DMA B(s+1, half 0) | MFMA half 1        (input tile double-buffered, two barriers per stage).
#include "conv_epi.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define PS_TAPG 10      // virtual taps (taps x slabs) per stage
#define PS_NAMAX 14     // LDS-DMA instructions per wave for one input tile

struct PsGeom {
  int npixA;       // pixels of the halo tile
  int a_bytes;     // one input-tile buffer (na * 4 KiB)
  int na;          // DMA instructions per wave per input tile
  int p16;         // bytes of one (plane, 8-channel group) row: N*H*W*16
};

template <int I, int NM, int NR, int NF>
__device__ __forceinline__ void ps_sgb() {
  if constexpr (I < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (I < NF) {
      constexpr int k = (NR * (I + 1)) / NF - (NR * I) / NF;
      if constexpr (k > 0) __builtin_amdgcn_sched_group_barrier(0x100, k, 0);
    }
    ps_sgb<I + 1, NM, NR, NF>();
  }
}

// NTH virtual taps of one half: fragments of tap i+1 are read while the 6 MT NT MFMAs of tap i run
template <int BN, int MT, int NTH>
__device__ __forceinline__ void ps_half(f32x16 (&acc)[MT][BN / 32], f32x16& alt, const char* __restrict__ Ab,
                                        const char* __restrict__ Bh, const int (&abase)[MT], int pstride,
                                        const int (&aoff)[PS_TAPG], int t0, int lane) {
  constexpr int NT = BN / 32;
  bf16x8 a[2][MT][3], b[2][NT][3];
  const char* bp0 = Bh + lane * 16;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int p = 0; p < 3; ++p) a[0][m][p] = *(const bf16x8*)(Ab + abase[m] + aoff[t0] + p * pstride);
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p) b[0][u][p] = *(const bf16x8*)(bp0 + (u * 3 + p) * 1024);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int st = 0; st < NTH; ++st) {
    const int cur = st & 1, nxt = cur ^ 1;
#ifdef PS_ABL_NOREAD
    if (false) {
#else
    if (st + 1 < NTH) {
#endif
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[nxt][m][p] = *(const bf16x8*)(Ab + abase[m] + aoff[t0 + st + 1] + p * pstride);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p) b[nxt][u][p] = *(const bf16x8*)(bp0 + (((st + 1) * NT + u) * 3 + p) * 1024);
    }
    // smallest terms first; product-major so that back-to-back MFMAs hit different accumulators
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#ifdef PS_ABL_NOREAD
    constexpr int cur_ = 0;
#define cur cur_
#endif
#ifdef PS_ABL_NOMFMA
#define PS_NPR 1
#else
#define PS_NPR 6
#endif
    if constexpr (MT * NT == 1) {       // two accumulators: dependent MFMAs issue every ~64 cycles, not 32
#pragma unroll
      for (int pr = 0; pr < PS_NPR; ++pr) {
        if (pr & 1) alt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0][PA[pr]], b[cur][0][PB[pr]], alt, 0, 0, 0);
        else acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0][PA[pr]], b[cur][0][PB[pr]], acc[0][0], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int pr = 0; pr < PS_NPR; ++pr)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < NT; ++u)
          acc[m][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][m][PA[pr]], b[cur][u][PB[pr]], acc[m][u], 0, 0, 0);
    }
#ifdef PS_ABL_NOREAD
#undef cur
#endif
#if !defined(PS_NO_ILV) && !defined(PS_ABL_NOMFMA) && !defined(PS_ABL_NOREAD)
    if constexpr (MT * NT > 1) ps_sgb<0, 6 * MT * NT, 3 * (MT + NT), (6 * MT * NT * 2 + 2) / 3>();
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
}

// WS (wave specialisation): 512 threads.  Waves 0-3 are CONSUMERS -- one per SIMD, they read fragments and multiply, and
// meet the others only at the two barriers of a stage; waves 4-7 are PRODUCERS -- they sit next to a consumer on each
// SIMD and do nothing but issue the LDS-DMA of the stage ahead and wait for it to land.  An LDS-DMA instruction costs its
// issuing wave 60-100 cycles (measured: 17 % of the loop when the MFMA waves issue them themselves with one wave per
// SIMD); issued by a partner wave they run under the MFMAs.
template <int BN, int MT, int NTAPS, int SL, bool WS>
__global__ __launch_bounds__(WS ? 512 : 256) void conv_ps_k(const pmf_conv_desc_t d, const ConvGeom g, const PsGeom pg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  {
    PMF_SGPR_BATCH("s"(d.N), "s"(d.OH), "s"(d.OW), "s"(d.nsrc), "s"(d.ldw), "s"(d.w_s3), "s"(d.src[0].C), "s"(d.src[0].H),
                   "s"(d.src[0].W), "s"(g.segs_x_log2), "s"(g.th), "s"(g.tw), "s"(g.tiles_x), "s"(g.in_rows), "s"(g.in_cols),
                   "s"(g.dy_min), "s"(g.dx_min), "s"(g.Ktot), "s"(g.ksplit), "s"(pg.npixA), "s"(pg.a_bytes), "s"(pg.na),
                   "s"(pg.p16));
  }
  constexpr int NT = BN / 32, NV = NTAPS * SL, N0 = (NV + 1) / 2, N1 = NV - N0;
  static_assert(NV <= PS_TAPG, "too many virtual taps");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = WS && wave_all >= 4;
  const int wave = wave_all & 3;
  const int li = lane & 31, lh = lane >> 5;
  unsigned lin = blockIdx.x + gridDim.x * (blockIdx.z + gridDim.z * blockIdx.y);
  {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    if ((total & 7u) == 0u) lin = (lin & 7u) * (total >> 3) + (lin >> 3);
  }
  const int tile = lin % gridDim.x;
  const unsigned lin_r = lin / gridDim.x;
  const int bz = lin_r % gridDim.z, by = lin_r / gridDim.z;
  const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
  const int n = bz, n0 = (by / g.ksplit) * BN, ks = by % g.ksplit;
  const int oy0 = ty * g.th, ox0 = tx * g.tw;
  const int in_cols = g.in_cols, npixA = pg.npixA;
  int tri_ = 0;
  TR_START();
  TR();

  char* const A0 = (char*)smem;
  char* const Bh0 = A0 + 2 * pg.a_bytes;
  char* const Bh1 = Bh0 + N0 * NT * 3 * 1024;
  const int sH = d.src[0].H, sW = d.src[0].W;
  const int p16 = pg.p16, na = pg.na;

  // ---- stage iterator over (operand, 16 SL channels), split-K stages dealt round-robin
  int si = 0, c0 = 0, kb = 0, cn = 0;
  auto settle = [&]() {
    for (;;) {
      if (si >= d.nsrc) return false;
      if (c0 >= d.src[si].C) { kb += d.src[si].C; ++si; c0 = 0; continue; }
      if ((cn % g.ksplit) == ks) return true;
      ++cn; c0 += 16 * SL;
    }
  };

  // =========================================================================================== producer / DMA state
  // slot table of the input-tile DMA: instruction j of this wave fills slots (wave + 4 j) 64 + lane of the stage buffer
  // [slab][plane][half][pixel]; ga = byte offset inside the operand's planes without the plane term, -2^31 for slots that
  // read nothing (outside the image / past the tile: the buffer range check returns zeros)
  int ga[PS_NAMAX], gp[PS_NAMAX];
  const int KS = g.Ktot >> 4, CT = d.ldw >> 5;
  const size_t tap_stride = (size_t)KS * CT * 3 * 1024, slab_stride = (size_t)CT * 3 * 1024;
  __amdgpu_buffer_rsrc_t nrs;
  int nps = 0, nsoff = 0;
  const char* __restrict__ nw = nullptr;
  auto make_table = [&]() {
    const int ss = 6 * npixA;
    const float r_ss = 1.0f / (float)ss, r_np = 1.0f / (float)npixA, r_ic = 1.0f / (float)in_cols;
    auto fdiv = [](int a, int b, float rb) {   // a / b for 0 <= a < 2^22, exact after one correction step
      int q = (int)((float)a * rb);
      int r = a - q * b;
      q += r >= b ? 1 : 0; q -= r < 0 ? 1 : 0;
      return q;
    };
#pragma unroll
    for (int j = 0; j < PS_NAMAX; ++j) {
      const int s = (wave + 4 * j) * 64 + lane;
      const int sl = fdiv(s, ss, r_ss), r = s - sl * ss;
      const int pk = fdiv(r, npixA, r_np), pix = r - pk * npixA;
      const int row = fdiv(pix, in_cols, r_ic), col = pix - row * in_cols;
      const int iy = oy0 + g.dy_min + row, ix = ox0 + g.dx_min + col;
      const bool ok = sl < SL && iy >= 0 && iy < sH && ix >= 0 && ix < sW;
      ga[j] = ok ? (sl * 2 + (pk & 1)) * p16 + ((n * sH + iy) * sW + ix) * 16 : (int)0x80000000;
      gp[j] = ok ? (pk >> 1) : 0;
    }
  };
  auto head = [&]() {
    const void* sx = d.src[si].xs;
    const int sC = d.src[si].C;
    PMF_SGPR_BATCH("s"(sx), "s"(sC));
    nps = (sC >> 3) * p16;
    nrs = __builtin_amdgcn_make_buffer_rsrc((void*)sx, 0, 3 * nps, 0x00020000);
    nsoff = (c0 >> 3) * p16;
    nw = (const char*)d.w_s3 + ((size_t)((kb + c0) >> 4) * CT + (n0 >> 5)) * 3 * 1024;
  };
  auto dma_a = [&](char* __restrict__ dst) {
#ifdef PS_ABL_NODMA
    if (cn > 0) return;
#endif
#pragma unroll
    for (int j = 0; j < PS_NAMAX; ++j)
      if (j < na)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(nrs, (lds_ptr_t)(dst + (wave + 4 * j) * 1024), 16, ga[j] + gp[j] * nps, nsoff, 0, 0);
  };
  constexpr int NDMA = (N0 * NT * 3 + 3) / 4;
  auto dma_b = [&](const char* __restrict__ wsrc, char* __restrict__ dst, int va, int nth) {
#ifdef PS_ABL_NODMA
    if (cn > 0) return;
#endif
#pragma unroll
    for (int jj = 0; jj < NDMA; ++jj) {
      const int f = wave + 4 * jj;
      if (f < nth * NT * 3) {
        const int vl = f / (NT * 3), up = f - vl * (NT * 3);
        const int v = va + vl, t = v / SL, sl = v % SL;
        __builtin_amdgcn_global_load_lds((const float*)(wsrc + (size_t)t * tap_stride + sl * slab_stride + up * 1024 + lane * 16),
                                         (lds_ptr_t)(dst + f * 1024), 16, 0, 0);
      }
    }
  };
  auto wait_keep_a = [&]() {     // everything but the newest `na` DMA instructions (the next input tile) has landed
    switch (na) {
#define PS_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
      PS_W(1) PS_W(2) PS_W(3) PS_W(4) PS_W(5) PS_W(6) PS_W(7) PS_W(8) PS_W(9) PS_W(10) PS_W(11) PS_W(12) PS_W(13) PS_W(14)
#undef PS_W
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  if (producer) {     // ============================================================================== producer waves
    make_table();
    bool have = settle();
    int cur = 0;
    if (have) {
      head();
      dma_b(nw, Bh0, 0, N0);
      dma_a(A0);
    }
    while (have) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // A(s) and B(s, half 0) have landed
      __syncthreads();                                     // consumers are done with stage s-1
      const char* __restrict__ wcur = nw;
      ++cn; c0 += 16 * SL;
      have = settle();
      if (N1 > 0) dma_b(wcur, Bh1, N0, N1);
      if (have) {
        head();
        dma_a(A0 + (cur ^ 1) * pg.a_bytes);
      }
      if (N1 > 0) {
        if (have) wait_keep_a(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // B(s, half 1) visible; consumers finished half 0
      } else {
        __syncthreads();
      }
      if (have) dma_b(nw, Bh0, 0, N0);
      cur ^= 1;
    }
    if (g.ksplit <= 1 && d.stats) { __syncthreads(); __syncthreads(); }   // the barriers of the epilogue's statistics fold
    return;
  }

  // ================================================================================================ MFMA waves
  f32x16 acc[MT][NT], alt;
#pragma unroll
  for (int r = 0; r < 16; ++r) alt[r] = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
  int segrow[MT], segcol[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int s = wave * MT + m;
    segrow[m] = s >> g.segs_x_log2;
    segcol[m] = s & ((1 << g.segs_x_log2) - 1);
  }
  int aoff[PS_TAPG];
  {
    int tyv[9], txv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { tyv[t] = d.tdy[t]; txv[t] = d.tdx[t]; }
    PMF_SGPR_BATCH("s"(tyv[0]), "s"(tyv[1]), "s"(tyv[2]), "s"(tyv[3]), "s"(tyv[4]), "s"(tyv[5]), "s"(tyv[6]), "s"(tyv[7]),
                   "s"(tyv[8]), "s"(txv[0]), "s"(txv[1]), "s"(txv[2]), "s"(txv[3]), "s"(txv[4]), "s"(txv[5]), "s"(txv[6]),
                   "s"(txv[7]), "s"(txv[8]));
#pragma unroll
    for (int v = 0; v < PS_TAPG; ++v) {
      const int t = v / SL, sl = v % SL;
      aoff[v] = (v < NV && t < 9) ? ((tyv[t < 9 ? t : 0] - g.dy_min) * in_cols + (txv[t < 9 ? t : 0] - g.dx_min)) * 16 + sl * 6 * npixA * 16 : 0;
      asm volatile("" : "+v"(aoff[v]));     // wave-uniform, but SGPRs are the scarce file here
    }
  }
  const int pstride = 2 * npixA * 16;
  int abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) abase[m] = (segrow[m] * in_cols + segcol[m] * 32 + li + lh * npixA) * 16;
  if (!WS) make_table();
  TR();
  bool have = settle();
  int cur = 0;
  if (!WS && have) {
    head();
    dma_b(nw, Bh0, 0, N0);
    dma_a(A0);
  }
  TR();
  while (have) {
    if (!WS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of A(s) and B(s, half 0) has landed
    __syncthreads();                                     // ... everyone's; and everyone is done with stage s-1
    TR();
    const char* __restrict__ wcur = nw;
    const char* Ab = A0 + cur * pg.a_bytes;
    ++cn; c0 += 16 * SL;
    have = settle();
    if (!WS) {
      if (N1 > 0) dma_b(wcur, Bh1, N0, N1);
      if (have) {
        head();
        dma_a(A0 + (cur ^ 1) * pg.a_bytes);
      }
    }
    // (opaque copies: hipcc would otherwise hoist all taps x tiles x planes LDS addresses out of the stage loop into VGPRs)
#pragma unroll
    for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(abase[m]));
#pragma unroll
    for (int v = 0; v < PS_TAPG; ++v) asm volatile("" : "+v"(aoff[v]));
    __builtin_amdgcn_sched_barrier(0);
    ps_half<BN, MT, N0>(acc, alt, Ab, Bh0, abase, pstride, aoff, 0, lane);
    TR();
    if (N1 > 0) {
      if (!WS) { if (have) wait_keep_a(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __syncthreads();                                   // B(s, half 1) visible; everyone finished reading half 0
      TR();
      if (!WS && have) dma_b(nw, Bh0, 0, N0);
      ps_half<BN, MT, (N1 > 0 ? N1 : 1)>(acc, alt, Ab, Bh1, abase, pstride, aoff, N0, lane);
    } else {
      __syncthreads();
      TR();
      if (!WS && have) dma_b(nw, Bh0, 0, N0);
    }
    cur ^= 1;
    TR();
  }
  TR();
  if constexpr (MT * NT == 1) acc[0][0] += alt;
#ifdef PMF_CONV_TRACE
  conv_epilogue<BN, MT>(d, g, acc, segrow, segcol, n, n0, ks, oy0, ox0, tile, smem, tri_, tr_w0_);
#else
  conv_epilogue<BN, MT>(d, g, acc, segrow, segcol, n, n0, ks, oy0, ox0, tile, smem, tri_);
#endif
}

#ifdef PMF_CONV_TRACE
extern "C" int pmf_conv_ps_trace_set(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(pmf_trace_buf), &p, sizeof(p)); }
#endif

// ---- host side ---------------------------------------------------------------------------------------------------
int pmf_conv_geometry(int OH, int OW, int ntaps, const int8_t* tdy, const int8_t* tdx, int in_stride, int gather_req,
                      int BN, int MT, int kc_alloc, ConvGeom* g, int* gather_out);
int pmf_conv_choose_ksplit(const pmf_conv_desc_t* d, int blocks_mn, int nchunks, int mfma_per_chunk);
int pmf_conv_finish_launch(const pmf_conv_desc_t* d, const ConvGeom& g, hipStream_t s);

// slabs per stage: few-tap layers carry 2 / 4 slabs so that a stage holds enough MFMAs between its two barriers
static int ps_slabs(const pmf_conv_desc_t* d) {
  int sl = d->ntaps >= 5 ? 1 : (d->ntaps >= 3 ? 2 : 4);
  if (const char* e = getenv("PMF_PS_SL")) sl = atoi(e);
  for (; sl > 1; sl >>= 1) {
    bool ok = d->ntaps * sl <= PS_TAPG;
    for (int i = 0; i < d->nsrc; ++i) ok = ok && d->src[i].C % (16 * sl) == 0;
    if (ok) break;
  }
  return sl < 1 ? 1 : sl;
}

// geometry + LDS bytes of the pre-split launch for tile (BN, MT); 0 when the layer does not qualify
static int ps_plan(const pmf_conv_desc_t* d, int BN, int MT, ConvGeom* g, PsGeom* pg, int* sl_out) {
  if (!d->w_s3 || d->in_stride != 1 || d->gather || (d->ldw & 31)) return 0;
  if (d->ntaps != 1 && d->ntaps != 2 && d->ntaps != 3 && d->ntaps != 4 && d->ntaps != 9) return 0;
  int Ktot = 0;
  for (int i = 0; i < d->nsrc; ++i) {
    if (d->src[i].C % 16 || (d->src[i].flags & PMF_SRC_BCAST)) return 0;
    if (d->src[i].H != d->src[0].H || d->src[i].W != d->src[0].W) return 0;
    if (3 * (int64_t)(d->src[i].C / 8) * d->N * d->src[i].H * d->src[i].W * 16 >= (1ll << 31)) return 0;
    Ktot += d->src[i].C;
  }
  int gather;
  pmf_conv_geometry(d->OH, d->OW, d->ntaps, d->tdy, d->tdx, 1, 0, BN, MT, 16, g, &gather);
  if (gather) return 0;
  const int sl = ps_slabs(d);
  const int npix = g->in_rows * g->in_cols;
  const int na = cdiv(sl * 6 * npix, 256);
  if (na > PS_NAMAX) return 0;
  pg->npixA = npix; pg->na = na; pg->a_bytes = na * 4096;
  pg->p16 = d->N * d->src[0].H * d->src[0].W * 16;
  g->Ktot = Ktot;
  *sl_out = sl;
  int lds = 2 * pg->a_bytes + d->ntaps * sl * (BN / 32) * 3 * 1024;
  if (lds < 2 * 4 * 64 * 2 * 8) lds = 2 * 4 * 64 * 2 * 8;
  return lds <= 160 * 1024 ? lds : 0;
}

void pmf_conv_config_raw(const pmf_conv_desc_t* d, int* BN, int* MT);
// tile configuration of the pre-split launch: the caller's / the heuristic's (BN, MT), a 128-pixel tile where the
// 256-pixel one does not fit; 0 when the layer is not in this kernel's class
int pmf_conv_ps_config(const pmf_conv_desc_t* d, int* BN, int* MT) {
  static const bool off = getenv("PMF_NO_PS") != nullptr;
  if (off) return 0;
  pmf_conv_config_raw(d, BN, MT);
  ConvGeom g; PsGeom pg; int sl;
  if (ps_plan(d, *BN, *MT, &g, &pg, &sl)) return 1;
  if (*MT == 2 && ps_plan(d, *BN, 1, &g, &pg, &sl)) { *MT = 1; return 1; }
  if (*BN == 64 && ps_plan(d, 32, 1, &g, &pg, &sl)) { *BN = 32; *MT = 1; return 1; }
  return 0;
}

extern "C" int pmf_conv_ps_eligible(const pmf_conv_desc_t* d) {
  pmf_conv_desc_t t = *d;
  if (!t.w_s3) t.w_s3 = (const void*)1;
  if (!t.ldw) t.ldw = 64;
  ConvGeom g; PsGeom pg; int sl;
  return ps_plan(&t, 32, 1, &g, &pg, &sl) ? 1 : 0;
}

template <int BN, int MT>
static int launch_ps(const pmf_conv_desc_t* d, hipStream_t s) {
  ConvGeom g; PsGeom pg; int sl = 1;
  const int lds = ps_plan(d, BN, MT, &g, &pg, &sl);
  if (!lds) return PMF_E_UNSUPPORTED;
  if ((int64_t)d->N * d->out_H * d->out_W * d->out_ldc * 4 >= (1ll << 31)) return PMF_E_UNSUPPORTED;
  if (d->ep_relu_x && (int64_t)d->N * d->out_H * d->out_W * d->ep_relu_ldc * 4 >= (1ll << 31)) return PMF_E_UNSUPPORTED;
  if (d->ep_stat_mean && (!d->stats || !d->ep_relu_x)) return PMF_E_ARG;
  int nchunks = 0;
  for (int i = 0; i < d->nsrc; ++i) nchunks += d->src[i].C / (16 * sl);
  const int co_tiles = cdiv(d->Cout, BN);
  g.ksplit = pmf_conv_choose_ksplit(d, g.tiles_x * g.tiles_y * d->N * co_tiles, nchunks, d->ntaps * 8 * MT * (BN / 32));
  g.ws = d->splitk_ws;
  g.ws_ld = round_up(d->Cout, 4);
  dim3 grid(g.tiles_x * g.tiles_y, co_tiles * g.ksplit, d->N);
  static const bool ws = getenv("PMF_PS_NOWS") == nullptr;
#define PS_LAUNCH(NTAPS, SL)                                                                                          \
  do {                                                                                                                \
    static bool attr = false;                                                                                         \
    if (!attr) {                                                                                                      \
      (void)hipFuncSetAttribute((const void*)conv_ps_k<BN, MT, NTAPS, SL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      (void)hipFuncSetAttribute((const void*)conv_ps_k<BN, MT, NTAPS, SL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                    \
    }                                                                                                                 \
    if (ws) hipLaunchKernelGGL((conv_ps_k<BN, MT, NTAPS, SL, true>), grid, dim3(512), lds, s, *d, g, pg);             \
    else hipLaunchKernelGGL((conv_ps_k<BN, MT, NTAPS, SL, false>), grid, dim3(256), lds, s, *d, g, pg);               \
  } while (0)
  const int key = d->ntaps * 10 + sl;
  switch (key) {
    case 91: PS_LAUNCH(9, 1); break;
    case 41: PS_LAUNCH(4, 1); break;
    case 42: PS_LAUNCH(4, 2); break;
    case 31: PS_LAUNCH(3, 1); break;
    case 32: PS_LAUNCH(3, 2); break;
    case 21: PS_LAUNCH(2, 1); break;
    case 22: PS_LAUNCH(2, 2); break;
    case 24: PS_LAUNCH(2, 4); break;
    case 11: PS_LAUNCH(1, 1); break;
    case 12: PS_LAUNCH(1, 2); break;
    case 14: PS_LAUNCH(1, 4); break;
    default: return PMF_E_UNSUPPORTED;
  }
#undef PS_LAUNCH
  PMF_LAUNCH_CHECK();
  if (g.ksplit > 1) return pmf_conv_finish_launch(d, g, s);
  return 0;
}

// called by pmf_conv_fwd when every operand carries pre-split planes; PMF_E_UNSUPPORTED = take the staged kernels
int pmf_conv_ps_launch(const pmf_conv_desc_t* d, int BN, int MT, hipStream_t s) {
  if (BN == 64) return MT == 2 ? launch_ps<64, 2>(d, s) : launch_ps<64, 1>(d, s);
  return MT == 2 ? launch_ps<32, 2>(d, s) : launch_ps<32, 1>(d, s);
}

// partial-statistics rows / K stages of the pre-split launch (pmf_conv_fwd_stat_rows / pmf_conv_fwd_kstages)
int pmf_conv_ps_shape(const pmf_conv_desc_t* d, int BN, int MT, int* tiles, int* nchunks, int* ksplit) {
  ConvGeom g; PsGeom pg; int sl = 1;
  if (!ps_plan(d, BN, MT, &g, &pg, &sl)) return 0;
  *nchunks = 0;
  for (int i = 0; i < d->nsrc; ++i) *nchunks += d->src[i].C / (16 * sl);
  *tiles = g.tiles_x * g.tiles_y;
  *ksplit = pmf_conv_choose_ksplit(d, *tiles * d->N * cdiv(d->Cout, BN), *nchunks, d->ntaps * 8 * MT * (BN / 32));
  return 1;
}
