// Wave-scheduled split-bf16 convolution (round 5): the forward / input-gradient counterpart of conv_wgrad_s3n_k.
//
// conv_fwd_k's staged loop (conv_fwd.hip PIPE 5-10) gives every wave ALL output channels of one 32-pixel segment: per
// 16-channel stage a wave issues the transform + three-way split + LDS store of its share of the input tile (150-240 vector
// instructions), 7-14 LDS-DMA instructions for the weight fragments (60-100 cycles of issue each) and three barrier
// rendezvous around only 54-216 MFMAs; two workgroups per CU hide part of that in each other, and the matrix pipe ends up
// 37-43 % busy on the best layers (155-182 TFLOP/s fp32-equivalent where the weight-gradient kernel, scheduled by hand,
// runs 68 % busy).  Same recipe here:
//   * N-split: a workgroup owns a 4 x 32-pixel output tile and 32 NCO output channels; wave w owns output-channel tile
//     w % NCO for the 4 / NCO row segments of pixel group w / NCO, so ONE staged input tile feeds NCO x as many MFMAs per wave
//     (216 per 16-channel chunk at nine taps and NCO = 4) for the same staging work;
//   * weights never pass through LDS: a wave's B fragments are private to it (its own output-channel tile), and the packed
//     layout [tap][K/16][Cout/32][plane][lane][8 bf16] IS the MFMA B-fragment register layout -- one 16-byte buffer load per
//     lane and fragment, issued PD taps ahead into a ring of PD + 1 tap buffers (L2 hits: every workgroup of a launch reads
//     the same few hundred KiB); no LDS-DMA, no weight buffers in LDS;
//   * the input tile is double-buffered in LDS; tile t + 1 is transformed, split and stored by micro-ops placed in the MFMA
//     slots of tile t, the global loads of tile t + 2 follow them; ONE barrier per chunk;
//   * a chunk is a static schedule of NTAPS x MT x 6 slots = { one MFMA; at most one LDS read of the next group's A
//     fragments; at most one weight load; at most one micro-op of <= 6 instructions; sched_barrier(0) }.
// Arithmetic, operand views, zero padding and the epilogue (conv_epi.h) are those of conv_fwd_k; results differ from it by
// the summation order only (tap-major inside a chunk in both).  Eligibility: pmf_conv_ws_ok().
#include "conv_epi.h"
#include <stdlib.h>
#include <string.h>
#include <utility>

typedef __attribute__((ext_vector_type(8))) __bf16 wsb16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 wsb16x2;
typedef __attribute__((ext_vector_type(2))) float wsf32x2;
typedef __attribute__((ext_vector_type(2))) unsigned wsu32x2;
typedef __attribute__((ext_vector_type(4))) unsigned wsu32x4;
#define WS_APB 112        // bytes per staged pixel: [plane][16 bf16] + 16 (odd number of 16-byte slots: conflict-free b128 reads)
#define WS_KC 16

template <class F, int... Is>
__device__ __forceinline__ void ws_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ws_static_for(F&& f) {
  ws_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ unsigned ws_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(wsf32x2{a, b}, wsb16x2));
}
__device__ __forceinline__ float ws_lo16(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float ws_hi16(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ float ws_vmax(float a, float b) {   // plain v_max_f32 (fmaxf adds a canonicalising v_max)
  float r;
  asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// scalar-f32 forms that the SLP vectoriser cannot pack into v_pk_* (packed f32 next to MFMAs costs 22-26 cycles per
// instruction beyond its issue slot, MI355X_MICROARCH.md)
__device__ __forceinline__ float ws_fma(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float ws_mul(float a, float b) {
  float r;
  asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float ws_sub(float a, float b) {
  float r;
  asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct WsGeom {
  int segs_x_log2, th, tw, tiles_x, tiles_y;
  int in_rows, in_cols, dy_min, dx_min;
  int a_bytes;          // one input-tile buffer
  int KS, CT;           // 16-channel chunks in all operands; 32-channel output tiles in the packed weights (ldw / 32)
  int w_bytes;          // size of the packed weight array (buffer range)
};

// NTAPS taps, NCO output-channel tiles per workgroup, ASL staging slots (16 bytes) per thread, PD weight prefetch distance
template <int NTAPS, int NCO, int ASL, int PD>
__global__ __launch_bounds__(256) void conv_ws_k(const pmf_conv_desc_t d, const WsGeom g, const ConvGeom cg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = NCO;                        // row segments per wave: 4 / NCO pixel groups of NCO segments each
  constexpr int NB = PD + 1;                     // ring of tap buffers
  static_assert(NTAPS % NB == 0, "the ring index of a tap must not depend on the chunk");
  constexpr int NG = NTAPS * MT;                 // (tap, segment) groups per chunk
  constexpr int NMF = NG * 6;                    // MFMA slots per chunk
  constexpr int TF = NTAPS >= 3 ? NTAPS - 2 : NTAPS - 1;   // the tap whose slots carry the fetch of tile t + 2
  constexpr int NF = ASL + 3;                    // loads of a fetch: the tile slots + scale / shift / multiplier
  constexpr int NS_OPS = ASL * 10;               // micro-ops of the transform + split + store of one tile
  constexpr int S_SLOTS = TF * MT * 6;           // ... spread over the slots in front of tap TF
  static_assert(S_SLOTS >= 1, "schedule");
  char* __restrict__ A0 = (char*)smem;
  char* __restrict__ A1 = (char*)smem + g.a_bytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave % NCO, pg = wave / NCO;
  const int li = lane & 31, lh = lane >> 5;
  unsigned lin = blockIdx.x + gridDim.x * (blockIdx.z + gridDim.z * blockIdx.y);
  {   // XCD-aware tile order (see conv_fwd_k)
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    if ((total & 7u) == 0u) lin = (lin & 7u) * (total >> 3) + (lin >> 3);
  }
  const int tile = lin % gridDim.x;
  const unsigned lin_r = lin / gridDim.x;
  const int n = lin_r % gridDim.z, by = lin_r / gridDim.z;
  const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
  const int n0 = (by * NCO + cw) * 32;          // this wave's output-channel tile
  const int oy0 = ty * g.th, ox0 = tx * g.tw;
  const int in_cols = g.in_cols;
  const int sH = d.src[0].H, sW = d.src[0].W;
  int tri_ = 0;
  (void)tri_;

  f32x16 acc[MT][1];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;
  int segrow[MT], segcol[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int s = pg * MT + m;
    segrow[m] = s >> g.segs_x_log2;
    segcol[m] = s & ((1 << g.segs_x_log2) - 1);
  }

  // ---- staging slots of this thread: pixel -> global pixel index (or -1), LDS byte offset
  const int q = tid & 3;
  const int npixA = g.in_rows * in_cols;
  const int totalA = npixA * 4;
  int gA[ASL], oA[ASL];
  float mA[ASL];
  const float rcols = 1.f / (float)in_cols;
#pragma unroll
  for (int j = 0; j < ASL; ++j) {
    const int f = tid + 256 * j, pix = f >> 2;
    const int r = pmf_fdiv(pix, in_cols, rcols), c = pix - r * in_cols;
    const int iy = oy0 + g.dy_min + r, ix = ox0 + g.dx_min + c;
    const bool ok = f < totalA && iy >= 0 && iy < sH && ix >= 0 && ix < sW;
    gA[j] = ok ? (n * sH + iy) * sW + ix : -1;
    mA[j] = ok ? 1.f : 0.f;
    oA[j] = (f < totalA ? pix : npixA) * WS_APB + q * 8;          // slots beyond the tile write to the spare pixel
  }
  int aoff[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
    aoff[t] = (((int)d.tdy[t] - g.dy_min) * in_cols + ((int)d.tdx[t] - g.dx_min)) * WS_APB;
  int abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) abase[m] = (segrow[m] * in_cols + segcol[m] * 32 + li) * WS_APB + lh * 16;

  // ---- weights: B fragments straight from the packed array (L2), ring of NB tap buffers
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)d.w_s3, 0, g.w_bytes, 0x00020000);
  const int wvoff = ((n0 >> 5) * 3) * 1024 + lane * 16;
  const int tap_stride = g.KS * g.CT * 3 * 1024, chunk_stride = g.CT * 3 * 1024;
  wsu32x4 bq[NB][3];
  auto wload = [&](int slot, int plane, int soff) {
    bq[slot][plane] = __builtin_bit_cast(wsu32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, soff + plane * 1024, 0));
  };

  // ---- operand walk: chunk -> (operand, channel offset); the tile held in rA belongs to the chunk `f_*` describes
  int f_si = 0, f_c0 = 0;
  bool f_on = true;
  auto f_advance = [&]() {
    f_c0 += WS_KC;
    if (f_c0 >= d.src[f_si].C) { f_c0 = 0; ++f_si; }
    if (f_si >= d.nsrc) { f_on = false; f_si = d.nsrc - 1; }
  };
  f32x4 rA[ASL], sc4, sh4, cm4;
  float lo = 0.f;                                  // ReLU floor of the tile in rA
  float lo_next = 0.f;
  const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc((void*)d.src[0].x, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t xrs = zrs, prs_sc = zrs, prs_sh = zrs, prs_cm = zrs;
  int x_ld = 0, x_c = 0, cm_off = 0;
  // scalars of the fetch that comes next (tile t + 2 inside the loop)
  auto f_setup = [&]() {
    const pmf_src_t& sv = d.src[f_si];
    const int bytes = f_on ? d.N * sH * sW * sv.ldc * 4 : 0;
    xrs = __builtin_amdgcn_make_buffer_rsrc((void*)sv.x, 0, bytes, 0x00020000);
    prs_sc = __builtin_amdgcn_make_buffer_rsrc((void*)(sv.scale ? sv.scale : sv.x), 0, (f_on && sv.scale) ? sv.C * 4 : 0, 0x00020000);
    prs_sh = __builtin_amdgcn_make_buffer_rsrc((void*)(sv.shift ? sv.shift : sv.x), 0, (f_on && sv.scale) ? sv.C * 4 : 0, 0x00020000);
    prs_cm = __builtin_amdgcn_make_buffer_rsrc((void*)(sv.cmul ? sv.cmul : sv.x), 0, (f_on && sv.cmul) ? d.N * sv.cmul_ld * 4 : 0, 0x00020000);
    x_ld = sv.ldc; x_c = f_c0 + q * 4; cm_off = (n * sv.cmul_ld + x_c) * 4;
    lo_next = (sv.flags & PMF_SRC_RELU) ? 0.f : -__builtin_inff();
  };
  bool has_sc = false, has_cm = false, has_sc_next = false, has_cm_next = false;
  auto f_flags = [&]() { has_sc_next = f_on && d.src[f_si].scale != nullptr; has_cm_next = f_on && d.src[f_si].cmul != nullptr; };
  f32x4 rSc, rSh, rCm;                             // raw loads of the fetch (a missing array loads zeros: fixed up at rotate)
  auto fetch_op = [&](int k) {                     // load k of a fetch: ASL tile slots, then scale / shift / multiplier
    if (k < ASL) {
      const unsigned off = gA[k < ASL ? k : 0] >= 0 ? (unsigned)((gA[k < ASL ? k : 0] * x_ld + x_c) * 4) : 0x80000000u;
      rA[k < ASL ? k : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
    } else if (k == ASL) {
      rSc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs_sc, x_c * 4, 0, 0));
    } else if (k == ASL + 1) {
      rSh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs_sh, x_c * 4, 0, 0));
    } else {
      rCm = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs_cm, cm_off, 0, 0));
    }
  };
  auto rotate = [&]() {                            // the fetched tile becomes "the tile in rA" (after its loads have landed)
    has_sc = has_sc_next; has_cm = has_cm_next; lo = lo_next;
    sc4.x = has_sc ? rSc.x : 1.f; sc4.y = has_sc ? rSc.y : 1.f; sc4.z = has_sc ? rSc.z : 1.f; sc4.w = has_sc ? rSc.w : 1.f;
    sh4.x = has_sc ? rSh.x : 0.f; sh4.y = has_sc ? rSh.y : 0.f; sh4.z = has_sc ? rSh.z : 0.f; sh4.w = has_sc ? rSh.w : 0.f;
    cm4.x = has_cm ? rCm.x : 1.f; cm4.y = has_cm ? rCm.y : 1.f; cm4.z = has_cm ? rCm.z : 1.f; cm4.w = has_cm ? rCm.w : 1.f;
  };

  // ---- transform + split + store of slot j, as ten micro-ops (state in st_*)
  f32x4 st_t, st_m;
  unsigned st_l0 = 0, st_l1 = 0, st_l2 = 0, st_h0 = 0, st_h1 = 0, st_h2 = 0;
  float st_a = 0.f, st_b = 0.f, st_c = 0.f, st_d = 0.f;
  auto store_op = [&](int j, int k, char* __restrict__ Ad) {
    switch (k) {
      case 0: st_t.x = ws_fma(rA[j].x, sc4.x, sh4.x); st_t.y = ws_fma(rA[j].y, sc4.y, sh4.y);
              st_t.z = ws_fma(rA[j].z, sc4.z, sh4.z); st_t.w = ws_fma(rA[j].w, sc4.w, sh4.w); break;
      case 1: st_t.x = ws_vmax(st_t.x, lo); st_t.y = ws_vmax(st_t.y, lo); st_t.z = ws_vmax(st_t.z, lo); st_t.w = ws_vmax(st_t.w, lo); break;
      case 2: st_m.x = ws_mul(cm4.x, mA[j]); st_m.y = ws_mul(cm4.y, mA[j]); st_m.z = ws_mul(cm4.z, mA[j]); st_m.w = ws_mul(cm4.w, mA[j]); break;
      case 3: st_t.x = ws_mul(st_t.x, st_m.x); st_t.y = ws_mul(st_t.y, st_m.y); st_t.z = ws_mul(st_t.z, st_m.z); st_t.w = ws_mul(st_t.w, st_m.w); break;
      case 4: st_l0 = ws_pk(st_t.x, st_t.y); st_a = ws_sub(st_t.x, ws_lo16(st_l0)); st_b = ws_sub(st_t.y, ws_hi16(st_l0)); break;
      case 5: st_l1 = ws_pk(st_a, st_b); st_a = ws_sub(st_a, ws_lo16(st_l1)); st_b = ws_sub(st_b, ws_hi16(st_l1)); break;
      case 6: st_l2 = ws_pk(st_a, st_b);
              st_h0 = ws_pk(st_t.z, st_t.w); st_c = ws_sub(st_t.z, ws_lo16(st_h0)); st_d = ws_sub(st_t.w, ws_hi16(st_h0)); break;
      case 7: st_h1 = ws_pk(st_c, st_d); st_c = ws_sub(st_c, ws_lo16(st_h1)); st_d = ws_sub(st_d, ws_hi16(st_h1)); break;
      case 8: st_h2 = ws_pk(st_c, st_d); break;
      default: { char* o = Ad + oA[j];
                 *(wsu32x2*)(o) = wsu32x2{st_l0, st_h0};
                 *(wsu32x2*)(o + 32) = wsu32x2{st_l1, st_h1};
                 *(wsu32x2*)(o + 64) = wsu32x2{st_l2, st_h2}; } break;
    }
  };
  auto store_tile = [&](char* __restrict__ Ad) {
#pragma unroll
    for (int j = 0; j < ASL; ++j)
#pragma unroll
      for (int k = 0; k < 10; ++k) store_op(j, k, Ad);
  };

  // ---- A fragments of group (tap t, segment m): three 16-byte reads
  wsb16x8 af[2][MT][3];                              // [current / next tap][segment][plane]
  auto aread = [&](const char* __restrict__ Ac, int buf, int t, int m, int p) {
    af[buf][m][p] = *(const wsb16x8*)(Ac + abase[m] + aoff[t] + p * 32);
  };

  // ---- one chunk as a static schedule.  Vector-memory queue at the wait in front of tap t (t >= PD; the taps before it were
  // loaded by the previous chunk and are complete behind the wait at the top of the chunk): behind the loads of tap t come the
  // PD - 1 tap blocks issued since, and the fetch if tap TF lies among the issuing taps t - PD .. t - 1.
  auto w_off = [&](int chunk_base, int t) { return chunk_base + t * tap_stride; };

  // Slot s of tap t: product pr = s / MT, segment m = s % MT -- product-major, so that back-to-back MFMAs hit different
  // accumulators (a dependent MFMA waits for the whole pipeline depth of the one before it).
  auto run_chunk = [&](const char* __restrict__ Ac, char* __restrict__ An, int wb_cur, int wb_nxt) {
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
    constexpr int TS = MT * 6;                        // slots per tap
    constexpr int NR = MT * 3;                        // A reads of the next tap, one every second slot
    ws_static_for<NR>([&](auto R) { aread(Ac, 0, 0, decltype(R)::value / 3, decltype(R)::value % 3); });
    ws_static_for<NMF>([&](auto SI) {
      constexpr int s = decltype(SI)::value;
      constexpr int t = s / TS, ts = s % TS;          // tap, slot inside the tap
      constexpr int pr = ts / MT, m = ts % MT;
      constexpr int cur = t & 1, nxt = cur ^ 1;
      if constexpr (ts == 0 && t >= PD) {
        constexpr int n_after = 3 * (PD - 1) + ((t - PD <= TF && TF <= t - 1) ? NF : 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_after) : "memory");
      }
      acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][m][PA[pr]], __builtin_bit_cast(wsb16x8, bq[t % NB][PB[pr]]),
                                                         acc[m][0], 0, 0, 0);
      if constexpr ((ts & 1) == 0 && ts / 2 < NR && t + 1 < NTAPS) aread(Ac, nxt, t + 1, (ts / 2) / 3, (ts / 2) % 3);
      if constexpr ((ts & 1) == 1 && ts / 2 < 3) {    // one weight load: tap t + PD (of this chunk or of the next)
        constexpr int tw = t + PD, pl = ts / 2;
        if constexpr (tw < NTAPS) wload(tw % NB, pl, w_off(wb_cur, tw));
        else wload(tw % NB, pl, w_off(wb_nxt, tw - NTAPS));
      }
      if constexpr (s == 1) { f_advance(); f_setup(); f_flags(); }   // scalars of the fetch of tile t + 2 (used from tap TF on)
      if constexpr (s < S_SLOTS) {                    // store stream of tile t + 1, spread evenly over the slots before tap TF
        constexpr int k0 = (s * NS_OPS) / S_SLOTS, k1 = ((s + 1) * NS_OPS) / S_SLOTS;
        ws_static_for<k1 - k0>([&](auto KK) {
          constexpr int k = k0 + decltype(KK)::value;
          store_op(k / 10, k % 10, An);
        });
      }
      if constexpr (t == TF) {                        // fetch of tile t + 2: behind this tap's weight loads (slots 1, 3, 5)
        if constexpr (TS >= 6 + NF) {                 // one load per slot from slot 6 on
          if constexpr (ts >= 6 && ts < 6 + NF) fetch_op(ts - 6);
        } else {                                      // short taps: all of them in the tap's last slot
          if constexpr (ts == TS - 1) ws_static_for<NF>([&](auto KK) { fetch_op(decltype(KK)::value); });
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- prologue: tile 0 stored, tile 1 in flight, the first PD taps of chunk 0 in flight
  const int KS = g.KS;
  int wb = 0;                                         // byte offset of (tap 0) of the current chunk's fragments
  f_setup(); f_flags();
#pragma unroll
  for (int k = 0; k < NF; ++k) fetch_op(k);
#pragma unroll
  for (int t = 0; t < PD; ++t)
#pragma unroll
    for (int p = 0; p < 3; ++p) wload(t % NB, p, w_off(wb, t));
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PD) : "memory");
  rotate();
  store_tile(A0);
  f_advance(); f_setup(); f_flags();
#pragma unroll
  for (int k = 0; k < NF; ++k) fetch_op(k);

  int cur = 0;
  for (int c = 0; c < KS; ++c) {
    const char* Ac = cur ? A1 : A0;
    char* An = cur ? A0 : A1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tile c + 1 and the first PD taps of chunk c have landed
    __syncthreads();                                            // tile c complete in Ac; everyone finished reading An (tile c - 1)
    rotate();
    run_chunk(Ac, An, wb, wb + chunk_stride);                   // (advances the operand walk to tile c + 2 in its second slot)
    wb += chunk_stride;
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                              // (the epilogue's reductions reuse the tile buffers)
#ifdef PMF_CONV_TRACE
  const unsigned long long tr_w0_ = 0;
  conv_epilogue<32, MT, NCO>(d, cg, acc, segrow, segcol, n, n0, 0, oy0, ox0, tile, smem, tri_, tr_w0_);
#else
  conv_epilogue<32, MT, NCO>(d, cg, acc, segrow, segcol, n, n0, 0, oy0, ox0, tile, smem, tri_);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
static bool ws_geometry(const pmf_conv_desc_t* d, WsGeom* g, ConvGeom* cg, int* asl) {
  int OH = d->OH, OW = d->OW;
  int sx = 1;
  while (sx < 4 && (4 / sx) > OH) sx <<= 1;
  while (sx > 1 && (sx / 2) * 32 >= OW && (4 / (sx / 2)) <= OH) sx >>= 1;
  g->segs_x_log2 = sx == 4 ? 2 : (sx == 2 ? 1 : 0);
  g->th = 4 / sx; g->tw = 32 * sx;
  g->tiles_x = cdiv(OW, g->tw); g->tiles_y = cdiv(OH, g->th);
  int dy_min = 127, dy_max = -127, dx_min = 127, dx_max = -127;
  for (int t = 0; t < d->ntaps; ++t) {
    dy_min = d->tdy[t] < dy_min ? d->tdy[t] : dy_min; dy_max = d->tdy[t] > dy_max ? d->tdy[t] : dy_max;
    dx_min = d->tdx[t] < dx_min ? d->tdx[t] : dx_min; dx_max = d->tdx[t] > dx_max ? d->tdx[t] : dx_max;
  }
  g->dy_min = dy_min; g->dx_min = dx_min;
  g->in_rows = g->th + dy_max - dy_min; g->in_cols = g->tw + dx_max - dx_min;
  const int npix = g->in_rows * g->in_cols;
  if (npix * 4 > 256 * 5) return false;
  *asl = npix * 4 > 256 * 4 ? 5 : 4;
  g->a_bytes = round_up((npix + 1) * WS_APB, 16);
  int Ktot = 0;
  for (int i = 0; i < d->nsrc; ++i) Ktot += d->src[i].C;
  g->KS = Ktot / WS_KC; g->CT = d->ldw / 32;
  g->w_bytes = d->ntaps * g->KS * g->CT * 3 * 1024;
  memset(cg, 0, sizeof(*cg));
  cg->segs_x_log2 = g->segs_x_log2; cg->th = g->th; cg->tw = g->tw; cg->tiles_x = g->tiles_x; cg->tiles_y = g->tiles_y;
  cg->in_rows = g->in_rows; cg->in_cols = g->in_cols; cg->dy_min = dy_min; cg->dx_min = dx_min; cg->Ktot = Ktot;
  cg->ksplit = 1; cg->one = 0;
  return true;
}

// eligibility: split-bf16 weights, 9 or 4 taps, stride 1, one halo tile of <= 1280 staging slots, operands multiples of 16
// channels with the output's H x W, no broadcast, Cout a multiple of 32 (of 32 NCO for the chosen NCO)
extern "C" int pmf_conv_multi_ok(const pmf_conv_desc_t* d);
extern "C" int pmf_conv_ws_ok(const pmf_conv_desc_t* d) {
  if (!d->w_s3 || d->in_stride != 1 || d->gather || (d->ntaps != 9 && d->ntaps != 4) || (d->ldw & 31) || (d->Cout & 31)) return 0;
  if (d->out_sy != 1 && d->out_sy != 0) { /* strided outputs (parity classes of a stride-2 input gradient) are fine: epilogue */ }
  for (int i = 0; i < d->nsrc; ++i) {
    const pmf_src_t& s = d->src[i];
    if (s.C % 16 || (s.flags & PMF_SRC_BCAST) || s.H != d->src[0].H || s.W != d->src[0].W) return 0;
    if ((int64_t)d->N * s.H * s.W * s.ldc * 4 >= (1ll << 31)) return 0;
  }
  if (d->ndst > 0 && !pmf_conv_multi_ok(d)) return 0;     // (ranges start on 32-channel tiles and add up to Cout; no bias)
  WsGeom g; ConvGeom cg; int asl;
  if (!ws_geometry(d, &g, &cg, &asl)) return 0;
  if ((int64_t)d->N * d->out_H * d->out_W * d->out_ldc * 4 >= (1ll << 31)) return 0;
  int nco = d->Cout % 128 == 0 ? 4 : (d->Cout % 64 == 0 ? 2 : 1);
  // cfg bits 26-27 / PMF_CONV_WS_NCO: at most 1 / 2 / 4 tiles per workgroup; default: as many as the channel count allows while
  // the launch keeps >= 256 workgroups (a 16 x 512 map has 128 tiles: four tiles per workgroup would leave half the CUs idle)
  static const int env_nco = getenv("PMF_CONV_WS_NCO") ? atoi(getenv("PMF_CONV_WS_NCO")) : 0;
  const int code = ((d->cfg >> 25) & 1) ? ((d->cfg >> 26) & 3) : 0;     // (bits 26-27 mean something only next to bit 25)
  const int want = code ? (1 << (code - 1)) : env_nco;
  if (want == 1 || want == 2 || want == 4) { while (nco > want) nco >>= 1; return nco; }
  const int tiles = g.tiles_x * g.tiles_y * d->N;
  while (nco > 1 && tiles * (d->Cout / (32 * nco)) < 256) nco >>= 1;
  return nco;
}

extern "C" int pmf_conv_ws_rows(const pmf_conv_desc_t* d) {       // partial-statistics rows of a launch (one per tile and sample)
  WsGeom g; ConvGeom cg; int asl;
  if (!ws_geometry(d, &g, &cg, &asl)) return 0;
  return g.tiles_x * g.tiles_y * d->N;
}

template <int NTAPS, int NCO, int PD>
static int ws_launch_(const pmf_conv_desc_t* d, const WsGeom& g, const ConvGeom& cg, int asl, hipStream_t s) {
  const dim3 grid(g.tiles_x * g.tiles_y, d->Cout / (32 * NCO), d->N);
  int lds = 2 * g.a_bytes;
  if (lds < 2 * 4 * 64 * 2 * 8) lds = 2 * 4 * 64 * 2 * 8;
  static unsigned long long attr = 0ull;
  if (pmf_first_on_device(&attr)) {
    (void)hipFuncSetAttribute((const void*)conv_ws_k<NTAPS, NCO, 4, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_ws_k<NTAPS, NCO, 5, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (asl == 4) hipLaunchKernelGGL((conv_ws_k<NTAPS, NCO, 4, PD>), grid, dim3(256), lds, s, *d, g, cg);
  else hipLaunchKernelGGL((conv_ws_k<NTAPS, NCO, 5, PD>), grid, dim3(256), lds, s, *d, g, cg);
  PMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pmf_conv_ws_launch(const pmf_conv_desc_t* d, pmf_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  const int nco = pmf_conv_ws_ok(d);
  if (!nco) return PMF_E_UNSUPPORTED;
  WsGeom g; ConvGeom cg; int asl;
  ws_geometry(d, &g, &cg, &asl);
  if (d->ntaps == 9) {
    if (nco == 4) return ws_launch_<9, 4, 2>(d, g, cg, asl, s);
    if (nco == 2) return ws_launch_<9, 2, 2>(d, g, cg, asl, s);
    return ws_launch_<9, 1, 8>(d, g, cg, asl, s);
  }
  if (nco == 4) return ws_launch_<4, 4, 1>(d, g, cg, asl, s);
  if (nco == 2) return ws_launch_<4, 2, 3>(d, g, cg, asl, s);
  return ws_launch_<4, 1, 3>(d, g, cg, asl, s);
}
