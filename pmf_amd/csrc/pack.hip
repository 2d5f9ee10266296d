// Weight re-layout: PyTorch OIHW parameters -> packed implicit-GEMM slabs [tap][k][n] (gfx950).
// One batched launch re-packs every conv of the network (weights change every optimiser step), tiled through
// LDS so both the OIHW reads and the packed writes are coalesced.
#include "common.h"

#define PACK_L 320  // floats of one co-row slice held in LDS

__global__ __launch_bounds__(256) void pack_k(const pmf_pack_job_t* __restrict__ jobs, int njobs) {
  __shared__ float T[32][PACK_L + 1];
  // locate the job of this block (block_start is ascending)
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const pmf_pack_job_t& J = jobs[lo];
  const int b = blockIdx.x - J.block_start;
  const int tci = b % J.tiles_ci, tco = b / J.tiles_ci;
  const int co0 = tco * 32, ci0 = tci * J.CT;
  const int nco = min(32, J.Cout - co0), ct = min(J.CT, J.Cin - ci0);
  const int L = ct * J.KHW, wld = J.w_ld ? J.w_ld : J.Cin;
  for (int i = threadIdx.x; i < nco * L; i += 256) {
    const int col = i / L, j = i - col * L;
    T[col][j] = J.w[((size_t)(co0 + col) * wld + ci0) * J.KHW + j];
  }
  __syncthreads();
  if (J.format >= 1) {
    // split-bf16 fragments: GEMM element (tap t, k, n) -> plane p of fragment (t, k/16, n/32) at lane (n%32) + 32 ((k%16)/8),
    // element k%8 (the B-operand layout of v_mfma_f32_32x32x16_bf16); planes are 512 bf16 apart
    unsigned short* __restrict__ dst = (unsigned short*)J.dst;
    const int KS = J.K_pad >> 4, CTL = J.ldw >> 5;
    // whole k-octets inside this tile: one thread builds the 16 bytes a lane of the fragment holds (8 consecutive k of one
    // n), three 16-byte stores per thread, consecutive threads -> consecutive lanes of the fragment (coalesced)
    const int kdim = J.transpose ? nco : ct, ndim = J.transpose ? ct : nco;
    const int kbase = J.transpose ? co0 : ci0, nbase = J.transpose ? ci0 : co0;
    if (J.format == 1 && (kdim & 7) == 0 && (kbase & 7) == 0) {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const int noct = kdim >> 3, totalv = J.ntaps * noct * ndim;
      for (int i = threadIdx.x; i < totalv; i += 256) {
        const int nl = i % ndim, r = i / ndim, oc = r % noct, t = r / noct;
        const int tap = J.tap_idx[t];
        unsigned p0[4], p1[4], p2[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float x[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int kl = oc * 8 + e + h;
            x[h] = J.transpose ? T[kl][nl * J.KHW + tap] : T[nl][kl * J.KHW + tap];
          }
          unsigned u[2][3];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const unsigned u0 = __builtin_bit_cast(unsigned short, (__bf16)x[h]);
            const float r1 = x[h] - __builtin_bit_cast(float, u0 << 16);
            const unsigned u1 = __builtin_bit_cast(unsigned short, (__bf16)r1);
            const float r2 = r1 - __builtin_bit_cast(float, u1 << 16);
            const unsigned u2 = __builtin_bit_cast(unsigned short, (__bf16)r2);
            u[h][0] = u0; u[h][1] = u1; u[h][2] = u2;
          }
          p0[e >> 1] = u[0][0] | (u[1][0] << 16); p1[e >> 1] = u[0][1] | (u[1][1] << 16); p2[e >> 1] = u[0][2] | (u[1][2] << 16);
        }
        const int k = kbase + oc * 8, nn = nbase + nl;
        const size_t e0 = ((((size_t)t * KS + (k >> 4)) * CTL + (nn >> 5)) * 3) * 512 + ((nn & 31) + 32 * ((k & 15) >> 3)) * 8;
        *(u32x4*)(dst + e0) = u32x4{p0[0], p0[1], p0[2], p0[3]};
        *(u32x4*)(dst + e0 + 512) = u32x4{p1[0], p1[1], p1[2], p1[3]};
        *(u32x4*)(dst + e0 + 1024) = u32x4{p2[0], p2[1], p2[2], p2[3]};
      }
      return;
    }
    const int total = J.ntaps * nco * ct;
    for (int i = threadIdx.x; i < total; i += 256) {
      int col, cil, t;
      if (!J.transpose) { col = i % nco; const int r = i / nco; cil = r % ct; t = r / ct; }
      else { cil = i % ct; const int r = i / ct; col = r % nco; t = r / nco; }
      const float x = T[col][cil * J.KHW + J.tap_idx[t]];
      int k = J.transpose ? co0 + col : ci0 + cil;
      const int nn = J.transpose ? ci0 + cil : co0 + col;
      // format 2 (few-channel stem, conv_fwd.hip PIPE 14): ONE virtual tap whose K index is tap * 8 + channel, so that a
      // 16-deep MFMA step holds the 8 padded channels of TWO taps
      if (J.format == 2) { k += t * 8; t = 0; }
      const unsigned u0 = __builtin_bit_cast(unsigned short, (__bf16)x);
      const float r1 = x - __builtin_bit_cast(float, u0 << 16);
      const unsigned u1 = __builtin_bit_cast(unsigned short, (__bf16)r1);
      const float r2 = r1 - __builtin_bit_cast(float, u1 << 16);
      const unsigned u2 = __builtin_bit_cast(unsigned short, (__bf16)r2);
      const size_t e = ((((size_t)t * KS + (k >> 4)) * CTL + (nn >> 5)) * 3) * 512 + ((nn & 31) + 32 * ((k & 15) >> 3)) * 8 + (k & 7);
      dst[e] = (unsigned short)u0; dst[e + 512] = (unsigned short)u1; dst[e + 1024] = (unsigned short)u2;
    }
    return;
  }
  if (!J.transpose) {
    const int total = J.ntaps * ct * 32;
    for (int i = threadIdx.x; i < total; i += 256) {
      const int col = i & 31, r = i >> 5;
      const int cil = r % ct, t = r / ct;
      if (col < nco)
        J.dst[((size_t)t * J.K_pad + ci0 + cil) * J.ldw + co0 + col] = T[col][cil * J.KHW + J.tap_idx[t]];
    }
  } else {
    const int total = J.ntaps * nco * ct;
    for (int i = threadIdx.x; i < total; i += 256) {
      const int cil = i % ct, r = i / ct;
      const int col = r % nco, t = r / nco;
      J.dst[((size_t)t * J.K_pad + co0 + col) * J.ldw + ci0 + cil] = T[col][cil * J.KHW + J.tap_idx[t]];
    }
  }
}

extern "C" int pmf_pack_tile_ci(int32_t Cin, int32_t KHW) {
  int ct = PACK_L / KHW;
  if (ct < 1) ct = 1;
  if (ct > Cin) ct = Cin;
  if (ct >= 8) ct &= ~7;      // whole k-octets per tile (the vectorised split-bf16 writer)
  return ct;
}

extern "C" int pmf_pack_weights_batched(const pmf_pack_job_t* jobs_dev, int32_t njobs, int32_t total_blocks,
                                        pmf_stream_t s) {
  if (njobs < 1 || total_blocks < 1) return PMF_E_ARG;
  hipLaunchKernelGGL(pack_k, dim3(total_blocks), dim3(256), 0, (hipStream_t)s, jobs_dev, njobs);
  PMF_LAUNCH_CHECK();
  return 0;
}
