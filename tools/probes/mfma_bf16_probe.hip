// What does v_mfma_f32_32x32x16_bf16 do INSIDE one instruction?  (round 6: the six-product split is 2-3.5x noisier than the fp32-MFMA
// path at network level and its residual adds coherently in column sums -- docs/rounds/r06.md 1.)  Every A row = pattern[k], B = 1:
// all 1024 outputs are sum_k pattern[k] (+ C).  Build + run: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_bf16_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__global__ void probe_k(const float* __restrict__ pat, float c0, float* __restrict__ out) {
  const int lane = threadIdx.x, kh = lane >> 5;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)pat[kh * 8 + i]; b[i] = (__bf16)1.0f; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = c0;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];
}
__global__ void probe_f32_k(const float* __restrict__ pat, float c0, float* __restrict__ out) {   // v_mfma_f32_32x32x2_f32, 8 instructions
  const int lane = threadIdx.x, kh = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = c0;
  for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pat[2 * s + kh], 1.0f, acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];
}
static float run(bool bf, const float* pat, float c0) {
  float *dp, *dout, h;
  hipMalloc(&dp, 64); hipMalloc(&dout, 4);
  hipMemcpy(dp, pat, 64, hipMemcpyHostToDevice);
  if (bf) hipLaunchKernelGGL(probe_k, dim3(1), dim3(64), 0, 0, dp, c0, dout);
  else hipLaunchKernelGGL(probe_f32_k, dim3(1), dim3(64), 0, 0, dp, c0, dout);
  hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
  hipFree(dp); hipFree(dout);
  return h;
}
int main() {
  float pat[16];
  printf("# 1. one big product + 15 small ones of 2^-j (exact sum 1 + 15 * 2^-j): where do the small ones drop out of the inner sum?\n");
  for (int j = 8; j <= 40; j += 2) {
    pat[0] = 1.f; for (int k = 1; k < 16; ++k) pat[k] = ldexpf(1.f, -j);
    const double exact = 1.0 + 15.0 * ldexp(1.0, -j);
    const float g = run(true, pat, 0.f), g32 = run(false, pat, 0.f);
    printf("j=%2d  exact-1 %.6e  bf16-mfma-1 %.6e  f32-mfma-1 %.6e  (float(exact)-1 %.6e)\n", j, exact - 1.0, (double)g - 1.0, (double)g32 - 1.0,
           (double)(float)exact - 1.0);
  }
  printf("# 2. the same with NEGATIVE small products (1 - 15 * 2^-j): truncation toward zero or round to nearest?\n");
  for (int j = 20; j <= 30; j += 1) {
    pat[0] = 1.f; for (int k = 1; k < 16; ++k) pat[k] = -ldexpf(1.f, -j);
    const double exact = 1.0 - 15.0 * ldexp(1.0, -j);
    const float g = run(true, pat, 0.f), g32 = run(false, pat, 0.f);
    printf("j=%2d  exact-1 %.6e  bf16-mfma-1 %.6e  f32-mfma-1 %.6e  (float(exact)-1 %.6e)\n", j, exact - 1.0, (double)g - 1.0, (double)g32 - 1.0,
           (double)(float)exact - 1.0);
  }
  printf("# 3. accumulator add: C = 1, products sum to s: how is C + s rounded?\n");
  const double ss[] = {ldexp(1.0, -24), 1.5 * ldexp(1.0, -24), ldexp(1.0, -25), 3.0 * ldexp(1.0, -25), -ldexp(1.0, -25), -1.5 * ldexp(1.0, -25), -ldexp(1.0, -26)};
  for (double s : ss) {
    for (int k = 0; k < 16; ++k) pat[k] = 0.f;
    pat[0] = (float)s;                      // exactly representable in bf16 for the powers of two; 1.5 * 2^-24 = 2^-24 + 2^-25: two products
    if (s == 1.5 * ldexp(1.0, -24)) { pat[0] = ldexpf(1.f, -24); pat[1] = ldexpf(1.f, -25); }
    if (s == 3.0 * ldexp(1.0, -25)) { pat[0] = ldexpf(1.f, -24); pat[1] = ldexpf(1.f, -25); }
    if (s == -1.5 * ldexp(1.0, -25)) { pat[0] = -ldexpf(1.f, -25); pat[1] = -ldexpf(1.f, -26); }
    const float g = run(true, pat, 1.f), g32 = run(false, pat, 1.f);
    printf("s=%+.4e  float(1+s)-1 %.6e  bf16-mfma-1 %.6e  f32-mfma-1 %.6e\n", s, (double)(float)(1.0 + s) - 1.0, (double)g - 1.0, (double)g32 - 1.0);
  }
  return 0;
}
