// Optimiser updates over a RANGE of the flat training state (pmf_net.py FlatState): AdamW with decoupled weight decay
// for the LiDAR stream, SGD with (Nesterov) momentum for the camera stream -- tasks/pmf/trainer.py:80-98 of the
// reference builds torch.optim.AdamW / torch.optim.SGD over those parameter lists; the arithmetic below is theirs
// (torch/optim/adamw.py, torch/optim/sgd.py), evaluated in float32 per element.
// A range launch is what lets the engine update the parameters whose gradients are final while the backward plan is still
// running (engine.py _optimise_behind_events): an optimiser over the whole buffer has to wait for the last gradient.
// HBM-bound: AdamW reads 16 B and writes 12 B per parameter, SGD reads 12 B and writes 8 B.
#include "common.h"
#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void adamw_range_k(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, int64_t n4, int64_t n,
                                                     float lr, float beta1, float beta2, float eps, float keep,
                                                     double beta1d, double beta2d, const float* __restrict__ step) {
  // bias corrections from the step counter on the device (the caller incremented it for this step): no host read
  const double st = (double)*step;
  const float bc1 = (float)(1.0 - pow(beta1d, st)), bc2 = (float)(1.0 - pow(beta2d, st));
  // 1 - beta in double first: float(1 - 0.999) and 1.f - float(0.999) differ by 5e-5 relative
  const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2), w1 = (float)(1.0 - beta1d), w2 = (float)(1.0 - beta2d);
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    pp = pp * keep;                            // param.mul_(1 - lr * weight_decay): the factor is formed in double by
                                               // Python and rounded once to float32 (the host passes it), as torch does
    mm = mm + w1 * (gg - mm);                  // exp_avg.lerp_(grad, 1 - beta1)
    vv = beta2 * vv + w2 * (gg * gg);          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step_size * (mm / denom);            // param.addcdiv_(exp_avg, denom, value=-step_size)
  };
  if (i < n4) {
    const f32x4 P = ((f32x4*)p)[i], M = ((f32x4*)m)[i], V = ((f32x4*)v)[i], G = ((const f32x4*)g)[i];
    float pa[4] = {P.x, P.y, P.z, P.w}, ma[4] = {M.x, M.y, M.z, M.w}, va[4] = {V.x, V.y, V.z, V.w};
    const float ga[4] = {G.x, G.y, G.z, G.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) upd(pa[k], ga[k], ma[k], va[k]);
    ((f32x4*)p)[i] = f32x4{pa[0], pa[1], pa[2], pa[3]};
    ((f32x4*)m)[i] = f32x4{ma[0], ma[1], ma[2], ma[3]};
    ((f32x4*)v)[i] = f32x4{va[0], va[1], va[2], va[3]};
  } else {
    const int64_t j = n4 * 4 + (i - n4);       // the (n % 4) tail elements, one thread each
    if (j < n) upd(p[j], g[j], m[j], v[j]);
  }
}

__global__ __launch_bounds__(256) void sgd_range_k(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ buf, int64_t n4, int64_t n, float lr, float momentum,
                                                   float dampening, float wd, int nesterov, int first) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  const float keep = 1.f - dampening;
  auto upd = [&](float& pp, float gg, float& bb) {
    if (wd != 0.f) gg = gg + wd * pp;                           // grad.add(param, alpha=weight_decay)
    if (momentum != 0.f) {
      bb = first ? gg : momentum * bb + keep * gg;              // buf = clone(grad) | buf.mul_(momentum).add_(grad, alpha=1-dampening)
      gg = nesterov ? gg + momentum * bb : bb;
    }
    pp -= lr * gg;
  };
  if (i < n4) {
    const f32x4 P = ((f32x4*)p)[i], B = ((f32x4*)buf)[i], G = ((const f32x4*)g)[i];
    float pa[4] = {P.x, P.y, P.z, P.w}, ba[4] = {B.x, B.y, B.z, B.w};
    const float ga[4] = {G.x, G.y, G.z, G.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) upd(pa[k], ga[k], ba[k]);
    ((f32x4*)p)[i] = f32x4{pa[0], pa[1], pa[2], pa[3]};
    if (momentum != 0.f) ((f32x4*)buf)[i] = f32x4{ba[0], ba[1], ba[2], ba[3]};
  } else {
    const int64_t j = n4 * 4 + (i - n4);
    if (j < n) { float bb = buf[j]; upd(p[j], g[j], bb); if (momentum != 0.f) buf[j] = bb; }
  }
}

static inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

extern "C" int pmf_adamw_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                               double beta1, double beta2, double eps, double weight_decay, const float* step,
                               pmf_stream_t s) {
  if (n < 0 || !step) return PMF_E_ARG;
  if (n == 0) return 0;
  if (!param || !grad || !exp_avg || !exp_avg_sq) return PMF_E_ARG;
  const int64_t n4 = (al16(param) && al16(grad) && al16(exp_avg) && al16(exp_avg_sq)) ? n / 4 : 0;
  const int64_t threads = n4 + (n - 4 * n4);
  hipLaunchKernelGGL(adamw_range_k, dim3((unsigned)cdiv64(threads, 256)), dim3(256), 0, (hipStream_t)s, param, grad, exp_avg,
                     exp_avg_sq, n4, n, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)(1.0 - lr * weight_decay), beta1, beta2,
                     step);
  PMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pmf_sgd_range(float* param, const float* grad, float* momentum_buffer, int64_t n, double lr, double momentum,
                             double dampening, double weight_decay, int32_t nesterov, int32_t first_step, pmf_stream_t s) {
  if (n < 0) return PMF_E_ARG;
  if (n == 0) return 0;
  if (!param || !grad || (momentum != 0.0 && !momentum_buffer)) return PMF_E_ARG;
  if (nesterov && (momentum <= 0.0 || dampening != 0.0)) return PMF_E_ARG;      // torch.optim.SGD raises ValueError
  float* buf = momentum_buffer ? momentum_buffer : param;                        // never touched when momentum == 0
  const int64_t n4 = (al16(param) && al16(grad) && al16(buf)) ? n / 4 : 0;
  const int64_t threads = n4 + (n - 4 * n4);
  hipLaunchKernelGGL(sgd_range_k, dim3((unsigned)cdiv64(threads, 256)), dim3(256), 0, (hipStream_t)s, param, grad, buf, n4, n,
                     (float)lr, (float)momentum, (float)dampening, (float)weight_decay, nesterov, first_step);
  PMF_LAUNCH_CHECK();
  return 0;
}

// ---- input normalisation of a training / validation batch, in place (tasks/pmf/trainer.py:291-295 of the reference:
// input_feature[:, 0:5] = (input_feature[:, 0:5] - mean) / std * mask.unsqueeze(1)) -- the reference's four element-wise
// passes as one launch; float32 subtract, IEEE divide, multiply in that order (fp contract off: same bits as torch).
__global__ __launch_bounds__(256) void normalise_k(float* __restrict__ x, const float* __restrict__ mask,
                                                   const float* __restrict__ mean, const float* __restrict__ stdv, int C,
                                                   int64_t stride_n, int64_t HW, int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;       // over N * C * HW
  if (i >= total) return;
  const int64_t hw = i % HW, nc = i / HW;
  const int c = (int)(nc % C);
  const int64_t n = nc / C;
  float* p = x + n * stride_n + (int64_t)c * HW + hw;
  const float d = *p - mean[c];
  const float q = d / stdv[c];
  *p = q * mask[n * HW + hw];
}

extern "C" int pmf_normalise_inplace(float* x, int64_t stride_n, const float* mask, const float* mean, const float* stdv,
                                     int32_t N, int32_t C, int64_t HW, pmf_stream_t s) {
  if (!x || !mask || !mean || !stdv || N < 0 || C < 1 || HW < 1 || stride_n < (int64_t)C * HW) return PMF_E_ARG;
  const int64_t total = (int64_t)N * C * HW;
  if (total == 0) return 0;
  hipLaunchKernelGGL(normalise_k, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)s, x, mask, mean, stdv, C,
                     stride_n, HW, total);
  PMF_LAUNCH_CHECK();
  return 0;
}
