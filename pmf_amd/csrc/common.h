// Internal helpers shared by the gfx950 kernels of libpmf_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pmf_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PMF_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e_ = hipGetLastError();             \
    if (e_ != hipSuccess) return (int)e_;          \
  } while (0)

#define PMF_COL_ROWS 512   // max partial rows written by the column-reduction kernels (one per workgroup)
extern int g_pmf_col_cap;      // tuning hook (pmf_debug_col): workgroup cap of the column kernels, <= the rows the caller sized
extern int g_pmf_col_unroll;   // ... and pixels per trip (4 or 8)

// true the first time it is called with `mask` on the current device: per-device one-time work (the dynamic-LDS function
// attribute is a property of the (function, device) pair)
static inline bool pmf_first_on_device(unsigned long long* mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 64) return true;
  const bool first = !((*mask >> dev) & 1ull);
  *mask |= 1ull << dev;
  return first;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

__device__ __forceinline__ float pmf_act(float v, int act) {
  if (act == PMF_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  if (act == PMF_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == PMF_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

// value of a view element (4 consecutive channels), zero padding is the caller's business
__device__ __forceinline__ f32x4 pmf_view_load4(const float* __restrict__ x, const float* __restrict__ scale,
                                                const float* __restrict__ shift, const float* __restrict__ cmul,
                                                int flags, size_t off, int c) {
  f32x4 v = *(const f32x4*)(x + off);
  if (scale) {
    f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
    v = v * sc + sh;
  }
  if (flags & PMF_SRC_RELU) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  if (cmul) v = v * *(const f32x4*)(cmul + c);
  return v;
}

// Scheduling recipe of one MFMA step (device code): MFMA i is followed by its share of the NR LDS reads that fetch the
// NEXT step's operands (and, behind the first NV MFMAs, one global load each).  Left alone -- or fenced into
// [reads][MFMAs] blocks with sched_barrier -- hipcc waits lgkmcnt(0) right behind reads it has just issued; inside the
// 64-cycle MFMA gaps the reads are free.  The builtin wants literal counts, hence the recursion.
template <int I, int NM, int NR, int NV>
__device__ __forceinline__ void pmf_sgb_seq() {
  if constexpr (I < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int k = (NR * (I + 1)) / NM - (NR * I) / NM;
    if constexpr (k > 0) __builtin_amdgcn_sched_group_barrier(0x100, k, 0);
    if constexpr (I < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    pmf_sgb_seq<I + 1, NM, NR, NV>();
  }
}

// a / b for 0 <= a < 2^22, b > 0 through the float reciprocal rb = 1.f / b with one correction step (exact): the slot tables
// of the conv prologues divide a dozen times per thread, and an integer division is ~40 instructions on this machine
__device__ __forceinline__ int pmf_fdiv(int a, int b, float rb) {
  int q = (int)((float)a * rb);
  const int r = a - q * b;
  q += r >= b ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}

// geometry shared by conv forward / weight-gradient host code
struct ConvGeom {
  int segs_x_log2;  // 32-pixel segments across the tile (log2)
  int th, tw;       // output tile (rows, cols)
  int tiles_x, tiles_y;
  int in_rows, in_cols;  // LDS input tile (halo mode: all taps; gather: one tap)
  int dy_min, dx_min;
  int Ktot;
  int kc_alloc;     // rows per tap reserved in the LDS weight tile
  int a_floats;     // floats reserved for the input tile
  int tap_group;    // taps per weight sub-stage
  int one;          // split loops: single operand and no K split -> the operand scalars are fetched once (conv_kloop_s3 ONE)
  int kchunk;       // direct 1x1 variant: 16-channel steps per weight chunk (two LDS buffers); 0 = all fragments resident
  int ksplit;       // split-K factor (K chunks dealt round-robin to ksplit workgroups; partials go to ws)
  int ws_ld;        // channel stride of the partial slabs
  float* ws;        // [ksplit][N*OH*OW][ws_ld] partial sums (deterministic split-K)
  unsigned* tickets;  // non-NULL: the last workgroup to arrive at an output tile combines the slabs in-kernel (conv_epi.h)
};
