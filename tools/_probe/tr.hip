#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(s16x4* o, int mode) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4* lp;
  int l = threadIdx.x;
  int idx;
  if (mode == 0) idx = l * 4;   // lane-linear
  else {  // row-major [pixel][64 elements pitch]: lane p of a 16-group -> row p>>2, cols 4*(p&3); group g -> +16 cols (g&1), +8 rows (g>>1)
    int p = l & 15, g = l >> 4;
    idx = ((p >> 2) + 8 * (g >> 1)) * 64 + 4 * (p & 3) + 16 * (g & 1);
  }
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + idx));
  o[l] = v;
}
int main() {
  s16x4* d; hipMalloc(&d, 64 * 8);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("l%2d: %5d %5d %5d %5d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l & 3) == 3 ? "\n" : " | "); }
  }
  return 0;
}
