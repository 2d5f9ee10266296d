// Operand pre-splitting for the split-bf16 matrix-pipe path (conv_ps.hip): fp32 view -> three bf16 planes.  gfx950 only.
//
// The convolution kernels of conv_fwd.hip split every fp32 operand into bf16(x), bf16(x - h1), bf16(x - h1 - h2) while
// they stage it -- once per output-channel tile and once per use (forward, weight gradient).  Here a tensor is split ONCE,
// with its view (BatchNorm scale / shift, ReLU, Dropout2d multiplier) applied, and stored in the order the consumers' LDS-DMA
// wants: dst[plane][c/8][pixel][8 bf16], i.e. 64 consecutive pixels of one 8-channel group = 1 KiB contiguous.
// HBM-bound: 4 B read + 6 B written per element.
#include "common.h"

typedef __attribute__((ext_vector_type(2))) __bf16 ps_bf16x2;
typedef __attribute__((ext_vector_type(2))) float ps_f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned ps_u32x4;

__device__ __forceinline__ unsigned ps_pk(ps_f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ps_bf16x2)); }
__device__ __forceinline__ ps_f32x2 ps_unpk(unsigned u) {
  return ps_f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
__device__ __forceinline__ void ps_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  ps_f32x2 r = {a, b};
  p0 = ps_pk(r);
  r = r - ps_unpk(p0);
  p1 = ps_pk(r);
  r = r - ps_unpk(p1);
  p2 = ps_pk(r);
}

// One workgroup = 64 consecutive pixels x all channel groups.  Reads are coalesced along the channels of a pixel (thread
// = (pixel, 8-channel group), group fastest); the three 16-byte results go through LDS so that the writes are coalesced
// along the pixels of a group (1 KiB runs).
template <int G>   // G = channel groups handled per pass (1, 2, 4, 8); C/8 is covered in passes of G
__global__ __launch_bounds__(256) void presplit_k(const pmf_view_t v, int64_t npix, int HW, int C, char* __restrict__ dst) {
  __shared__ ps_u32x4 sh[3][G][64 + 1];
  constexpr int PPW = 256 / G;                 // pixels per pass of the workgroup
  const int tid = threadIdx.x;
  const int gl = tid % G, pl = tid / G;
  const int ngroups = C >> 3;
  const int64_t plane_b = (int64_t)ngroups * npix * 16;
  for (int64_t p0 = (int64_t)blockIdx.x * PPW; p0 < npix; p0 += (int64_t)gridDim.x * PPW) {
    for (int g0 = 0; g0 < ngroups; g0 += G) {
      const int64_t p = p0 + pl;
      const int cg = g0 + gl, c = cg * 8;
      ps_u32x4 o0 = {0, 0, 0, 0}, o1 = {0, 0, 0, 0}, o2 = {0, 0, 0, 0};
      if (p < npix && cg < ngroups) {
        const int n = (int)(p / HW);
        const float* cm = v.cmul ? v.cmul + (size_t)n * v.cmul_ld : nullptr;
        const f32x4 a = pmf_view_load4(v.x, v.scale, v.shift, cm, v.flags, (size_t)p * v.ldc + c, c);
        const f32x4 b = pmf_view_load4(v.x, v.scale, v.shift, cm, v.flags, (size_t)p * v.ldc + c + 4, c + 4);
        unsigned x0, x1, x2, y0, y1, y2, z0, z1, z2, w0, w1, w2;
        ps_split2(a.x, a.y, x0, x1, x2);
        ps_split2(a.z, a.w, y0, y1, y2);
        ps_split2(b.x, b.y, z0, z1, z2);
        ps_split2(b.z, b.w, w0, w1, w2);
        o0 = ps_u32x4{x0, y0, z0, w0}; o1 = ps_u32x4{x1, y1, z1, w1}; o2 = ps_u32x4{x2, y2, z2, w2};
      }
      if (G == 1) {                            // already pixel-major
        if (p < npix) {
          char* o = dst + ((int64_t)cg * npix + p) * 16;
          *(ps_u32x4*)(o) = o0; *(ps_u32x4*)(o + plane_b) = o1; *(ps_u32x4*)(o + 2 * plane_b) = o2;
        }
      } else {
        __syncthreads();
        sh[0][gl][pl] = o0; sh[1][gl][pl] = o1; sh[2][gl][pl] = o2;
        __syncthreads();
        // writer: thread = (group, pixel), pixel fastest
        const int wp = tid % PPW, wg = tid / PPW;
        const int64_t pw = p0 + wp;
        if (pw < npix && g0 + wg < ngroups) {
          char* o = dst + ((int64_t)(g0 + wg) * npix + pw) * 16;
#pragma unroll
          for (int pl_ = 0; pl_ < 3; ++pl_) *(ps_u32x4*)(o + pl_ * plane_b) = sh[pl_][wg][wp];
        }
      }
    }
  }
}

extern "C" int64_t pmf_presplit_bytes(int64_t npix, int32_t C) { return 3 * (int64_t)(C / 8) * npix * 16; }

extern "C" int pmf_presplit(const pmf_view_t* v, int32_t N, int32_t HW, int32_t C, void* dst, pmf_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  if (!v || !v->x || !dst || C % 8 || v->ldc % 4 || N < 1 || HW < 1) return PMF_E_ARG;
  const int64_t npix = (int64_t)N * HW;
  const int ng = C / 8;
  const int G = ng >= 8 ? 8 : (ng >= 4 ? 4 : (ng >= 2 ? 2 : 1));
  int64_t blocks = cdiv64(npix, 256 / G);
  if (blocks > 16384) blocks = 16384;
  dim3 grid((unsigned)blocks);
  if (G == 8) hipLaunchKernelGGL(presplit_k<8>, grid, dim3(256), 0, s, *v, npix, HW, C, (char*)dst);
  else if (G == 4) hipLaunchKernelGGL(presplit_k<4>, grid, dim3(256), 0, s, *v, npix, HW, C, (char*)dst);
  else if (G == 2) hipLaunchKernelGGL(presplit_k<2>, grid, dim3(256), 0, s, *v, npix, HW, C, (char*)dst);
  else hipLaunchKernelGGL(presplit_k<1>, grid, dim3(256), 0, s, *v, npix, HW, C, (char*)dst);
  PMF_LAUNCH_CHECK();
  return 0;
}
