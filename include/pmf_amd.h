/*
 * pmf_amd.h -- C ABI of libpmf_amd.so: MI355X (gfx950) kernels for the PMF dual-branch fusion hot path.
 *
 * The reference (ICEORY/PMF) has NO FFI / plugin layer: every device op it runs comes from stock
 * torch.nn modules.  This header is therefore the NEW drop-in boundary a maintainer
 * binds instead of ATen for this path; each entry point cites the reference lines whose arithmetic it
 * replaces.  Conventions:
 *   - plain C, caller-owned DEVICE pointers (PyTorch's caching allocator), no allocation inside;
 *   - every call enqueues on the given hipStream_t and returns without synchronising;
 *   - return 0 on success, a hipError_t (>0) from the launch, or a negative PMF_E_* argument error;
 *   - activations are NHWC fp32 (channel stride `ldc` floats per pixel); NCHW only at the model boundary;
 *   - re-entrant, no mutable globals (one process per GPU, DDP).
 */
#ifndef PMF_AMD_H
#define PMF_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* pmf_stream_t; /* hipStream_t */

#define PMF_E_ARG      (-1)
#define PMF_E_UNSUPPORTED (-2)

#define PMF_MAX_SRC  5
#define PMF_MAX_TAPS 49

enum { PMF_ACT_NONE = 0, PMF_ACT_LRELU = 1, PMF_ACT_RELU = 2, PMF_ACT_SIGMOID = 3 };
enum { PMF_SRC_RELU = 1, PMF_SRC_BCAST = 2 };

/* One operand of a (virtual) channel-concat.  Value seen by the consumer at (n, y, x, c):
 *     v = x[n,y,x,c] * scale[c] + shift[c]      (BatchNorm applied on load; optional)
 *     v = max(v, 0)                              (PMF_SRC_RELU; conv -> BN -> ReLU ordering)
 *     v = v * cmul[n*cmul_ld + c]                (Dropout2d (n,c) multiplier; optional)
 * and 0 outside the image (zero padding applies AFTER the transform).  PMF_SRC_BCAST: H = W = 1 map
 * broadcast over the output (ASPP image-level branch, pmf_net.py:122-125). */
typedef struct {
  const float* x;
  const float* scale;
  const float* shift;
  const float* cmul;
  int32_t C;        /* channels taken from this operand; multiple of 8 */
  int32_t ldc;      /* floats between consecutive pixels */
  int32_t H, W;
  int32_t flags;
  int32_t cmul_ld;
} pmf_src_t;

/* Generic convolution as an implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces nn.Conv2d forward AND (with transposed/flipped packed weights) its input-gradient for:
 * 1x1, 2x2 dil 2, 3x3 dil 1/2/6/12/18, 7x7; stride 1/2 (salsanext.py:12-19,44-59,118-130,186;
 * pmf_net.py:14-26,69-70,107-117,188-212; torchvision BasicBlock/Bottleneck).
 *   out[n, oy*out_sy+out_oy, ox*out_sx+out_ox, co] (+)= ep( act( sum_t sum_k in_t(n, oy*in_stride+tdy[t], ox*in_stride+tdx[t], k) * w[t][k][co] + bias[co] ) )
 * where k runs over the concatenated operands.  ep(): optional multiply by ep_cmul[n][co], optional ReLU mask
 * (ep_relu_x*scale+shift > 0 at the output position), both used by the input-gradient form.
 * stats (optional): per-channel sum and sum of squares of the activated output, accumulated with atomics
 * into the FLOAT64 array stats[0..Cout) / stats[Cout..2Cout) (train-mode BatchNorm after the activation). */
/* One destination of a multi-destination launch (pmf_conv_desc_t.ndst > 0): output channels [sum of the C of the
 * destinations before it, + C) go to this tensor with this epilogue.  Used for the input gradient of a convolution over
 * concatenated operands (torch.cat + Conv2d, salsanext.py:83-88,150-160; pmf_net.py:31-36): ONE pass over dz with all
 * transposed weight columns, instead of one launch per operand each re-reading dz. */
typedef struct {
  float* out;
  int32_t out_ldc, C;     /* C: channels of this destination; a multiple of 32 (64 for the 64-wide output tile) */
  int32_t accumulate, ep_cmul_ld;
  const float* ep_cmul;
  const float* ep_relu_x;
  const float* ep_relu_scale;
  const float* ep_relu_shift;
  int32_t ep_relu_ldc, ep_flags;
  double* stats;          /* [rows][2][C] of THIS destination (rows = pmf_conv_fwd_stat_rows of the launch) */
  const float* ep_stat_mean;
} pmf_conv_dst_t;

typedef struct {
  int32_t N, OH, OW;      /* GEMM M-space: output positions computed */
  int32_t Cout;
  int32_t nsrc;
  pmf_src_t src[PMF_MAX_SRC];
  int32_t ntaps;
  int8_t tdy[PMF_MAX_TAPS + 3], tdx[PMF_MAX_TAPS + 3];
  int32_t in_stride;
  int32_t gather;         /* 0: one LDS halo tile shared by all taps; 1: per-tap staging (big dilation) */
  const float* w;         /* packed [ntaps][Ktot][ldw], Ktot = sum of src[i].C */
  int32_t ldw;            /* >= Cout rounded up to 32; readable up to the tile edge */
  const float* bias;      /* [Cout] or NULL */
  int32_t act;
  float* out;
  int32_t out_ldc, out_H, out_W, out_sy, out_sx, out_oy, out_ox;
  int32_t accumulate;
  const float* ep_cmul;
  int32_t ep_cmul_ld;
  const float* ep_relu_x;
  const float* ep_relu_scale;
  const float* ep_relu_shift;
  int32_t ep_relu_ldc;
  double* stats;
  const float* ep_pmask;     /* optional per-pixel multiplier [N*out_H*out_W] applied after the activation and before the
                              * statistics (EPMF SparseVariantConv: output * dilated mask, epmf_net.py:44-49) */
  float* splitk_ws;          /* optional scratch for deterministic split-K on small maps (NULL = never split) */
  int64_t splitk_ws_bytes;
  int32_t cfg;               /* 0 = built-in heuristics; else tile configuration chosen by the caller's autotuner:
                              * BN (32|64) | MT (1|2) << 8 | K splits (1 = none) << 16, see pmf_conv_fwd_stat_rows;
                              * bit 24 (PMF_CFG_DIRECT_TAPS): a 2..9-tap launch with w_s3 runs the direct variant -- activations
                              * straight from global memory per tap, no input tile in LDS (conv_fwd.hip PIPE 13) -- where its
                              * weight fragments fit; ignored elsewhere */
  int32_t ep_flags;          /* PMF_EP_STAT_X_ONLY: ep_relu_x feeds ep_stat_mean only (no ReLU mask) */
  const float* ep_stat_mean; /* optional [Cout], needs stats + ep_relu_x: the second statistics column becomes
                              * sum out * (ep_relu_x - mean) instead of sum out^2 -- the BatchNorm-backward reduction
                              * (pmf_bn_bwd_reduce) of an input-gradient launch that is the LAST writer of that
                              * gradient map, folded by pmf_bn_bwd_fold */
  const void* w_s3;          /* optional, instead of w: the weights split into three bf16 planes in MFMA B-fragment order
                              * [ntaps][Ktot/16][ldw/32][3][64 lanes][8 bf16] (pmf_pack_job_t.format 1).  The launch then
                              * runs fp32 arithmetic on the bf16 matrix pipe (six split products, fp32 accumulate: see
                              * conv_fwd.hip PIPE 5); only for descriptors with pmf_conv_s3_eligible() != 0 */
  int32_t ndst;              /* 0: one destination (out / ep_* above).  > 0: dst[0..ndst) replace out, out_ldc, accumulate,
                              * ep_cmul, ep_relu_*, stats, ep_stat_mean and ep_flags per output-channel range; Cout = the sum
                              * of their C, no bias, no K split (pmf_conv_multi_ok) */
  pmf_conv_dst_t dst[PMF_MAX_SRC];
  uint32_t* splitk_tickets;  /* optional, with splitk_ws: u32[>= N * tiles * output-channel tiles] (<= 16384 entries), ZERO before the
                              * first launch and left zero by every launch (calls that share it must be stream-ordered).  A split-K
                              * launch then combines its partial slabs INSIDE the kernel: the last workgroup to arrive at an output
                              * tile's ticket sums the slabs in slab order (deterministic) and runs the epilogue -- no second
                              * launch (conv_finish_k); partial-statistics rows = pmf_conv_fwd_stat_rows() as without the split.
                              * NULL: the two-launch form. */
} pmf_conv_desc_t;
/* 1 when a descriptor with ndst > 0 can run as one launch (channel ranges on tile boundaries, no split-K workspace needed) */
int pmf_conv_multi_ok(const pmf_conv_desc_t* d);
#define PMF_EP_STAT_X_ONLY 1
#define PMF_CFG_DIRECT_TAPS (1 << 24)
/* cfg bit 25: a 4- or 9-tap stride-1 launch on split-bf16 weights runs the wave-scheduled N-split kernel (conv_ws.hip: the four
 * waves of a workgroup own different 32-channel output tiles of one staged 4 x 32-pixel input tile, weights as B fragments
 * straight from the packed array) where pmf_conv_ws_ok() != 0; ignored elsewhere */
#define PMF_CFG_WS (1 << 25)
int pmf_conv_ws_ok(const pmf_conv_desc_t* d);
/* 1 when the descriptor runs the software-pipelined K loop (stride 1 -- or a stride-2 3x3 with all nine taps --, one halo tile,
 * every operand a multiple of 16 channels, same H x W, no broadcast): the class pmf_conv_fwd accepts w_s3 for.
 * 3 for stem-class descriptors (ONE operand of 8 padded channels, 2..49 taps, e.g. the 7x7 RGB stem): the direct variant
 * with two taps per 16-deep MFMA step (conv_fwd.hip PIPE 14), weights in pmf_pack_job_t format 2;
 * 2 for one-tap descriptors (1x1 layers, any stride) that qualify for the direct variant (conv_fwd.hip PIPE 11:
 * activations straight from global memory, all weight fragments of an output-channel tile resident in LDS; operands
 * multiples of 16 channels, same H x W, no broadcast, Ktot * 32 * 6 bytes + tables within 160 KiB); 0 otherwise */
int pmf_conv_s3_eligible(const pmf_conv_desc_t* d);

int pmf_conv_fwd(const pmf_conv_desc_t* d, pmf_stream_t s);

/* Weight gradient (+ nothing else): dW[t][k][co] = sum_{n,oy,ox} in_t(n, oy*s+tdy, ox*s+tdx, k) * dz[n,oy,ox,co].
 * Two stages: per-block partial sums into `partial` (workspace, >= pmf_conv_wgrad_workspace() bytes),
 * then a deterministic reduction that writes the gradient in the PyTorch OIHW layout
 *   dw_oihw[(co*Cin_real + k)*KHW + tap_widx[t]]   (k < Cin_real; padded channels are dropped).
 * Replaces the weight half of aten::convolution_backward for every conv of the path. */
typedef struct {
  int32_t N, OH, OW, Cout;
  int32_t nsrc;
  pmf_src_t src[PMF_MAX_SRC];
  int32_t ntaps;
  int8_t tdy[PMF_MAX_TAPS + 3], tdx[PMF_MAX_TAPS + 3];
  int8_t tap_widx[PMF_MAX_TAPS + 3];
  int32_t in_stride;
  int32_t gather;
  const float* dz;
  int32_t dz_ldc;
  float* partial;
  int32_t nsplit;         /* pixel splits (grid.x); partial is [nsplit][ntaps][Ktot][Cout32] */
  float* dw_oihw;
  int32_t Cin_real, KHW;
  int32_t accumulate;     /* dw_oihw += (stride-2 parity classes share one weight) */
  const float* dbias_rows; /* optional: partial column sums of dz [dbias_nrows][dbias_ld] (pmf_bn_bwd_apply / */
  int32_t dbias_nrows, dbias_ld; /* pmf_act_bwd output); stage 2 folds them: dbias_out[co] += sum_r rows[r][co] */
  float* dbias_out;
  int32_t cfg;            /* 0 = built-in heuristics; else (autotuner) NT | kernel << 8: NT = 32-channel output tiles per
                           * workgroup (1, 2, 4); kernel 1 = pipelined (needs NT 1 and its shape conditions), 2 = unit-dealing */
  int32_t flags;          /* PMF_WGRAD_S3: run the pipelined kernel's products on the bf16 matrix pipe (operands split three
                           * ways, six products, fp32 accumulate -- fp32-class error, conv_wgrad.hip); ignored where the
                           * pipelined kernel's shape conditions do not hold */
} pmf_wgrad_desc_t;
#define PMF_WGRAD_S3 1

int pmf_conv_wgrad(const pmf_wgrad_desc_t* d, pmf_stream_t s);
/* its two stages separately (same descriptor): partial-slab kernel, then the deterministic reduction */
int pmf_conv_wgrad_partial(const pmf_wgrad_desc_t* d, pmf_stream_t s);
int pmf_conv_wgrad_reduce(const pmf_wgrad_desc_t* d, pmf_stream_t s);
int64_t pmf_conv_wgrad_workspace(const pmf_wgrad_desc_t* d);
/* stage 2 of many layers in ONE launch (a training plan queues the reductions of a run of layers: each needs its own
 * `partial` workspace then).  pmf_conv_wgrad_reduce_plan fills meta8 = {0, kind, Ktot, Cout32, KB, grid x, grid y, 0} for one
 * layer and returns its workgroup count; the caller sets meta8[0] = running sum of the counts (first workgroup of the
 * job), uploads the descriptors and the meta rows, and launches them with pmf_conv_wgrad_reduce_multi.  Results are
 * bit-identical to pmf_conv_wgrad_reduce per layer. */
int pmf_conv_wgrad_reduce_plan(const pmf_wgrad_desc_t* d, int32_t* meta8);
int pmf_conv_wgrad_reduce_multi(const pmf_wgrad_desc_t* jobs_dev, const int32_t* meta_dev, int32_t njobs,
                                int32_t total_blocks, pmf_stream_t s);

/* OIHW -> packed GEMM layout, all convs of the network in ONE launch (weights change every optimiser step).
 * Job j writes dst[t][k][n] (dst is [ntaps][K_pad][ldw], padding pre-zeroed by the caller once):
 *   transpose == 0 (forward):        n = co, k = ci :  w[co][ci][tap_idx[t]]
 *   transpose == 1 (input gradient): n = ci, k = co :  w[co][ci][tap_idx[t]]
 * Workgroup b of job j handles output-channel tile b / tiles_ci and input-channel tile b % tiles_ci
 * (CT = pmf_pack_tile_ci(Cin, KHW) channels each); block_start = first workgroup of the job (ascending). */
typedef struct {
  const float* w;
  float* dst;
  int32_t Cout, Cin, KHW, ntaps, transpose, K_pad, ldw, CT, tiles_ci, block_start;
  int8_t tap_idx[PMF_MAX_TAPS + 3];
  int32_t format;            /* 0: fp32 slabs [tap][K_pad][ldw]; 1: split-bf16 fragments (pmf_conv_desc_t.w_s3); 2: the same
                              * fragments for a stem-class layer (one operand of 8 padded channels, many taps): ONE virtual tap
                              * with K index = tap * 8 + channel, K_pad = ntaps * 8 rounded up to 16 (transpose 0 only) */
  int32_t w_ld;              /* input channels per output-channel row of w when the job packs a channel sub-range
                              * (w then points at its first channel, Cin = channels of the range); 0: Cin */
} pmf_pack_job_t;
int pmf_pack_tile_ci(int32_t Cin, int32_t KHW);
int pmf_pack_weights_batched(const pmf_pack_job_t* jobs_dev, int32_t njobs, int32_t total_blocks, pmf_stream_t s);

/* ---- BatchNorm2d (eps 1e-5, momentum 0.1; salsanext.py:17,21,...; torchvision bn) ------------------------
 * No atomics anywhere: producers write one partial row per workgroup, tiny fold kernels sum the rows in a fixed
 * order (deterministic; contended float64 atomics cost ~80 us per launch on MI355X).
 * train: stats = float64 [nrows][2][C] partial (sum, sumsq) rows from pmf_conv_fwd (nrows = pmf_conv_fwd_stat_rows)
 *        -> scale/shift for apply-on-load, saved mean/invstd, running_mean/var update (unbiased var).
 * eval:  running stats -> scale/shift. */
int pmf_conv_fwd_stat_rows(const pmf_conv_desc_t* d);
/* upper bound of pmf_conv_fwd_stat_rows over every tile configuration `cfg` may select (sizes `stats`) */
int pmf_conv_fwd_stat_rows_max(const pmf_conv_desc_t* d);
/* number of 16-channel K stages (after 64-channel stage merging) of this descriptor under cfg: bounds the K splits */
int pmf_conv_fwd_kstages(const pmf_conv_desc_t* d);
int pmf_bn_finalize(const double* stats, int32_t nrows, float count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                    float* save_mean, float* save_invstd, int32_t C, pmf_stream_t s);
int pmf_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* scale, float* shift, float* save_mean, float* save_invstd, int32_t C,
                       pmf_stream_t s);
/* small maps (pmf_bn_bwd_small_ok: <= 2048 pixels, C % 4 == 0): the whole train-mode BatchNorm backward of a layer --
 * column sums, dgamma += / dbeta +=, dz = coef-form of salsanext.py:27-33's BatchNorm2d backward times act'(a), and the
 * exact column sum of dz (the conv-bias gradient) as ONE row -- in one launch, one read of gy and a. */
int pmf_bn_bwd_small_ok(int64_t npix, int32_t C);
int pmf_bn_bwd_small(const float* gy, int32_t gy_ldc, const float* a, int32_t a_ldc, int64_t npix, int32_t C,
                     const float* save_mean, const float* gamma, const float* save_invstd, int32_t train, int32_t act,
                     float* dz, int32_t dz_ldc, float* dbias_row, float* dgamma, float* dbeta, pmf_stream_t s);
/* number of partial rows the column-reduction kernels (bn_bwd_reduce/apply, act_bwd) write for npix pixels, C channels */
int pmf_col_rows(int64_t npix, int32_t C);
/* tuning hook (tools/bench_elem.py): workgroup cap (> 0) and pixels per trip (4 / 8) of the column kernels; returns the cap */
int pmf_debug_col(int32_t cap, int32_t unroll);
/* backward pass 1 + fold: partial rows of sum gy, sum gy*(a-mean) into `part` (float64 [rows][2][C] scratch), then
 * dgamma += invstd*sum gy*(a-mean), dbeta += sum gy, coef[3][C] = {gamma*invstd, invstd^2*mean(gy*(a-mean)), mean(gy)}
 * (train == 0, eval-mode BN: coef[1] = coef[2] = 0). */
int pmf_bn_bwd_reduce(const float* gy, int32_t gy_ldc, const float* a, int32_t a_ldc, int64_t npix, int32_t C,
                      const float* save_mean, const float* gamma, const float* save_invstd, int32_t train,
                      double* part, float* coef, float* dgamma, float* dbeta, pmf_stream_t s);
/* the fold step of pmf_bn_bwd_reduce alone, over `nrows` partial rows [nrows][2][C] that an input-gradient launch with
 * ep_stat_mean wrote (pmf_conv_fwd_stat_rows rows): same dgamma / dbeta / coef results without re-reading gy and a */
int pmf_bn_bwd_fold(const double* part, int32_t nrows, int32_t C, int64_t npix, int32_t train, const float* gamma,
                    const float* save_invstd, float* coef, float* dgamma, float* dbeta, pmf_stream_t s);
/* backward pass 2: dz = coef0 * ((gy - coef2) - (a - mean) * coef1) * act'(a)   (act LRELU: slope from the sign of
 * a = lrelu(z); NONE for conv -> BN -> ReLU ordering).  dbias_rows (optional): partial column sums of dz
 * [pmf_col_rows(npix, C)][dbias_ld] for pmf_conv_wgrad to fold into the conv-bias gradient. */
int pmf_bn_bwd_apply(const float* gy, int32_t gy_ldc, const float* a, int32_t a_ldc, int64_t npix, int32_t C,
                     const float* coef, const float* save_mean, int32_t act, float* dz, int32_t dz_ldc,
                     float* dbias_rows, int32_t dbias_ld, pmf_stream_t s);

/* ---- element-wise / resampling ops (all NHWC fp32) ----------------------------------------------------- */
typedef struct { /* a tensor operand with optional affine-on-load */
  const float* x;
  const float* scale;
  const float* shift;
  const float* cmul;
  int32_t ldc, cmul_ld, flags;
} pmf_view_t;

/* out = act( va + vb )  with va/vb = affine views (vb optional).  Residual adds: salsanext.py:35,88;
 * torchvision BasicBlock `out += identity; relu`.  stats optional as in conv. */
int pmf_add_act(const pmf_view_t* a, const pmf_view_t* b, int32_t act, float* out, int32_t out_ldc, int64_t npix,
                int32_t HW, int32_t C, pmf_stream_t s);
/* backward of the above: g_in = g_out * act'(out) (relu mask from `out`), written or accumulated into ga / gb */
int pmf_add_act_bwd(const float* gout, int32_t g_ldc, const float* out, int32_t out_ldc, int32_t act, float* ga,
                    int32_t ga_ldc, int32_t ga_acc, float* gb, int32_t gb_ldc, int32_t gb_acc, int64_t npix,
                    int32_t C, pmf_stream_t s);
/* g *= act'(a) in place for convs not followed by BN (salsanext.py:24-25,70-71); act NONE leaves g untouched.
 * dbias_rows (optional): partial column sums of the result, [pmf_col_rows(npix, C)][dbias_ld] (see pmf_conv_wgrad). */
int pmf_act_bwd(float* g, int32_t g_ldc, const float* a, int32_t a_ldc, int32_t act, float* dbias_rows,
                int32_t dbias_ld, int64_t npix, int32_t C, pmf_stream_t s);
/* AvgPool2d(3, stride 2, pad 1, count_include_pad) of view (salsanext.py:65,96) and its gradient */
int pmf_avgpool3s2(const pmf_view_t* in, int32_t N, int32_t H, int32_t W, int32_t C, float* out, int32_t out_ldc,
                   pmf_stream_t s);
int pmf_avgpool3s2_bwd(const float* gout, int32_t g_ldc, int32_t N, int32_t H, int32_t W, int32_t C,
                       const float* cmul, int32_t cmul_ld, float* gin, int32_t gin_ldc, int32_t acc, pmf_stream_t s);
/* MaxPool2d(3, 2, 1) of relu(bn(x)) (torchvision stem, pmf_net.py:94-95); idx = window position of the max */
int pmf_maxpool3s2(const pmf_view_t* in, int32_t N, int32_t H, int32_t W, int32_t C, float* out, int32_t out_ldc,
                   uint8_t* idx, pmf_stream_t s);
int pmf_maxpool3s2_bwd(const float* gout, int32_t g_ldc, const uint8_t* idx, int32_t N, int32_t H, int32_t W,
                       int32_t C, const pmf_view_t* in, float* gin, int32_t gin_ldc, int32_t acc, pmf_stream_t s);
/* nn.Upsample(scale 2, bilinear, align_corners=False) of a view (pmf_net.py:191-210) and gradient */
int pmf_bilinear2x(const pmf_view_t* in, int32_t N, int32_t H, int32_t W, int32_t C, float* out, int32_t out_ldc,
                   pmf_stream_t s);
int pmf_bilinear2x_bwd(const float* gout, int32_t g_ldc, int32_t N, int32_t H, int32_t W, int32_t C, float* gin,
                       int32_t gin_ldc, int32_t acc, pmf_stream_t s);
/* PixelShuffle(2) of a view, times an output (n,c) multiplier (salsanext.py:137-139) and gradient */
int pmf_pixel_shuffle2(const pmf_view_t* in, int32_t N, int32_t H, int32_t W, int32_t Cout, const float* out_cmul,
                       int32_t out_cmul_ld, float* out, int32_t out_ldc, pmf_stream_t s);
int pmf_pixel_shuffle2_bwd(const float* gout, int32_t g_ldc, int32_t N, int32_t H, int32_t W, int32_t Cout,
                           const float* out_cmul, int32_t out_cmul_ld, const float* in_cmul, int32_t in_cmul_ld,
                           float* gin, int32_t gin_ldc, int32_t acc, pmf_stream_t s);
/* fusion gate: out = f * sigmoid(att) + pcd with f, att affine views (pmf_net.py:33-35) and gradient */
int pmf_fusion_gate(const pmf_view_t* f, const pmf_view_t* att, const float* pcd, int32_t pcd_ldc, float* out,
                    int32_t out_ldc, int64_t npix, int32_t C, pmf_stream_t s);
int pmf_fusion_gate_bwd(const float* gout, int32_t g_ldc, const pmf_view_t* f, const pmf_view_t* att, float* gf,
                        int32_t gf_ldc, int32_t gf_acc, float* gatt, int32_t gatt_ldc, float* gpcd, int32_t gpcd_ldc,
                        int32_t gpcd_acc, int64_t npix, int32_t C, pmf_stream_t s);
/* per-(n,c) spatial mean of a view (ASPP image pooling, pmf_net.py:122) and gradient (broadcast / HW) */
int pmf_global_mean(const pmf_view_t* in, int32_t N, int32_t HW, int32_t C, float* out, pmf_stream_t s);
/* out[n][p][c] = src[n][c] for the HW pixels of every sample (ASPP's image-level branch, pmf_net.py:124-125, as a materialised
 * operand); C, src_ldc, out_ldc multiples of 4.  Its backward is pmf_colsum_rows with nz = N. */
int pmf_broadcast_rows(const float* src, int32_t src_ldc, int32_t N, int64_t HW, int32_t C, float* out, int32_t out_ldc,
                       pmf_stream_t s);
int pmf_global_mean_bwd(const float* gout, int32_t N, int32_t HW, int32_t C, const float* cmul, int32_t cmul_ld,
                        float* gin, int32_t gin_ldc, int32_t acc, pmf_stream_t s);
/* column sums: out[z][c] += sum over the npix pixels of sample z of x[z][p][c], z < nz (x advances npix*ldc,
 * out advances C per sample).  Bias gradients (nz = 1) and broadcast-operand gradients (nz = N). */
int pmf_colsum(const float* x, int32_t ldc, int64_t npix, int32_t C, float* out, int32_t nz, pmf_stream_t s);
/* the same sums, deterministic (partial rows + fixed-order fold instead of float atomics); scratch: nz * PMF_COL_ROWS * C
 * floats the call may overwrite */
int pmf_colsum_rows(const float* x, int32_t ldc, int64_t npix, int32_t C, float* out, int32_t nz, float* scratch,
                    pmf_stream_t s);
/* per-pixel validity masks of EPMF's SparseVariantConv / ResContextBlock (epmf_net.py:30-50, 66-80):
 * mask[p] = (sum_c |x[p][c]| != 0); dilated mask = max-pool of the zero-padded mask with the conv's kernel/stride/
 * dilation; y = view(x) * mask and its gradient gx (+)= gy * mask (gx == gy, acc = 0: in place); out = a + b (b may be
 * NULL) for the layer's two bias vectors. */
int pmf_pmask_from(const pmf_view_t* in, int64_t npix, int32_t HW, int32_t C, float* mask, pmf_stream_t s);
int pmf_pmask_pool(const float* mask, int32_t N, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t dil, int32_t pad,
                   int32_t stride, float* out, int32_t OH, int32_t OW, pmf_stream_t s);
int pmf_pmask_mul(const pmf_view_t* in, const float* mask, int64_t npix, int32_t HW, int32_t C, float* out,
                  int32_t out_ldc, pmf_stream_t s);
int pmf_pmask_mul_bwd(const float* gy, int32_t gy_ldc, const float* mask, int64_t npix, int32_t C, float* gx,
                      int32_t gx_ldc, int32_t acc, pmf_stream_t s);
int pmf_vec_add(const float* a, const float* b, float* out, int32_t n, pmf_stream_t s);
/* softmax over channels of NHWC logits -> NCHW probabilities (pmf_net.py:176-178,221) and gradient
 * dlogit[n,y,x,c] = p*(g - sum_k g_k p_k) from NCHW p, g. */
int pmf_softmax_nhwc_to_nchw(const float* logits, int32_t ldc, int32_t N, int32_t HW, int32_t C, float* prob_nchw,
                             pmf_stream_t s);
int pmf_softmax_bwd_nchw_to_nhwc(const float* prob_nchw, const float* g_nchw, int32_t N, int32_t HW, int32_t C,
                                 float* dlogits, int32_t ldc, pmf_stream_t s);
/* the same two layout changes WITHOUT the softmax: SalsaNext(softmax=False) returns its logits
 * (pc_processor/models/salsanext.py:167,206-207); the backward is the transposed copy */
int pmf_logits_nhwc_to_nchw(const float* logits, int32_t ldc, int32_t N, int32_t HW, int32_t C, float* out_nchw,
                            pmf_stream_t s);
int pmf_logits_bwd_nchw_to_nhwc(const float* g_nchw, int32_t N, int32_t HW, int32_t C, float* dlogits, int32_t ldc,
                                pmf_stream_t s);
/* model-boundary layout change with channel padding; input may be a strided NCHW view (trainer.py:296-297) */
int pmf_nchw_to_nhwc(const float* x, int64_t stride_n, int64_t stride_c, int32_t N, int32_t C, int32_t HW,
                     float* out, int32_t out_ldc, pmf_stream_t s);
int pmf_fill(float* p, float v, int64_t n, pmf_stream_t s);

/* ---- KNN post-processing (pc_processor/postproc/knn.py:55-143) ------------------------------------------ */
/* labels[p] = vote over the `knn` nearest of the search x search window around (py[p], px[p]);
 * inv_gauss = (1 - gaussian) window weights [search*search] (host-computed, knn.py:12-34,103-105).
 * Integer result is bit-exact w.r.t. the oracle; ties broken by smaller window index. */
int pmf_knn_vote(const float* proj_range, const float* unproj_range, const int64_t* proj_argmax, const int64_t* px,
                 const int64_t* py, int32_t H, int32_t W, int64_t P, int32_t knn, int32_t search,
                 const float* inv_gauss, float cutoff, int32_t nclasses, int64_t* labels, pmf_stream_t s);
/* The same vote for all B frames of a batch in one launch (BASELINE configs[1] runs bs = 4; the reference's infer.py
 * calls knn.py once per frame, tasks/pmf_eval_semantickitti/infer.py:100-112): proj_range / proj_argmax are [B][H][W], the
 * points of all frames are concatenated and frame b owns [offsets[b], offsets[b+1]) (offsets: int64[B+1] on the device,
 * offsets[B] = P_total).  Labels bit-identical to B calls of pmf_knn_vote.  search <= 5: a workgroup of 256 consecutive
 * points stages the bounding box of its windows through LDS (sweep-file order: a few columns); a box above 4096 pixels and
 * search 7 use global gathers. */
int pmf_knn_vote_batch(const float* proj_range, const float* unproj_range, const int64_t* proj_argmax, const int64_t* px,
                       const int64_t* py, const int64_t* offsets, int32_t B, int32_t H, int32_t W, int64_t P_total,
                       int32_t knn, int32_t search, const float* inv_gauss, float cutoff, int32_t nclasses,
                       int64_t* labels, pmf_stream_t s);
/* The batched vote straight from the network's probability maps prob_nchw f32[B][nclasses][H][W] (replaces
 * `argmax = pred.argmax(dim=1)` + KNN of tasks/pmf_eval_semantickitti/infer.py:96-112): launch 1 writes the channel argmax
 * (ties -> lowest class, NaN wins: torch.argmax's rule) as an int32 map into argmax_ws i32[B*H*W] (caller's workspace),
 * launch 2 is the vote on it.  Labels bit-identical to torch.argmax + pmf_knn_vote_batch. */
int pmf_knn_vote_batch_prob(const float* proj_range, const float* unproj_range, const float* prob_nchw, const int64_t* px,
                            const int64_t* py, const int64_t* offsets, int32_t B, int32_t H, int32_t W, int64_t P_total,
                            int32_t knn, int32_t search, const float* inv_gauss, float cutoff, int32_t nclasses,
                            int32_t* argmax_ws, int64_t* labels, pmf_stream_t s);

/* ---- perspective projection + scatter (perspective_view_loader.py:77-135, parser.py:209-227) ------------- */
/* points f32[P][4], sem i32[P], image u8[h][w][3], proj f64[12] (device), lut i32[nlut].
 * Writes proj_out f32[10][h][w] (depth,x,y,z,i,r,g,b,mask,label), keep u8[P], the compacted row/col indices
 * (x_data,y_data i32[<=P]), depth f32[P] and *n_kept.  pix_idx i32[h*w] and blk_cnt i32[ceil(P/1024)+1] are
 * caller workspaces.  Last point in file order wins on duplicate pixels. */
int pmf_project_scatter(const float* points, const int32_t* sem, int64_t P, const uint8_t* image, int32_t h,
                        int32_t w, const double* proj, const int32_t* lut, int32_t nlut, float* proj_out,
                        uint8_t* keep, int32_t* x_data, int32_t* y_data, float* depth, int32_t* n_kept,
                        int32_t* pix_idx, int32_t* blk_cnt, pmf_stream_t s);
/* EPMF loader (perspective_view_loader_v2.py:42-157, parser.py:229-257 mapLidar2CameraCropYaw): keep = |xyz| > 0.5 and
 * fov_left <= -atan2(y,x) <= fov_right, no image-bounds filter; pass 1 compacts (source index, trunc row/col, float64
 * (v,u), depth) in file order and returns n_kept + bbox {row_min,row_max,col_min,col_max} (device); the caller sizes the
 * frame from the bbox (h = row_max-row_min+1, w = col_max-col_min+1), pass 2 writes proj_out f32[10][h][w] =
 * depth,x,y,z,i,r,g,b (image window at (row_min,col_min), zero outside),mask,label; last kept point wins per pixel. */
int pmf_project_v2_index(const float* points, int64_t P, const double* proj, float fov_left, float fov_right,
                         uint8_t* keep, int32_t* src_idx, int32_t* x_data, int32_t* y_data, double* xy_index,
                         float* depth, int32_t* n_kept, int32_t* bbox, int32_t* blk_cnt, pmf_stream_t s);
/* same with the row / column coordinates multiplied by img_scale first (training: the reference rescales the image by
 * a random factor in [1, 1.2] and the projected coordinates with it, perspective_view_loader_v2.py:50-57,74) */
int pmf_project_v2_index_scaled(const float* points, int64_t P, const double* proj, float fov_left, float fov_right,
                                double img_scale, uint8_t* keep, int32_t* src_idx, int32_t* x_data, int32_t* y_data,
                                double* xy_index, float* depth, int32_t* n_kept, int32_t* bbox, int32_t* blk_cnt,
                                pmf_stream_t s);
int pmf_project_v2_scatter(const float* points, const int32_t* sem, const int32_t* src_idx, const int32_t* x_data,
                           const int32_t* y_data, const float* depth, int32_t K, const uint8_t* image, int32_t ih,
                           int32_t iw, const int32_t* lut, int32_t nlut, int32_t x_min, int32_t y_min, int32_t h,
                           int32_t w, float* proj_out, int32_t* pix_idx, pmf_stream_t s);
/* pmf_project_scatter in TWO launches and without the per-call memset (the loader's per-frame path): pix_tag u32[h*w] and
 * slots u64[1 + ceil(P/1024)] (slots[0] = the block-ticket counter, left at 0 by every call; a call that finds it non-zero -- unzeroed
 * or foreign workspace -- writes nothing out of bounds, does not hang and reports *n_kept = -1) are PERSISTENT workspaces of the
 * caller, zeroed once and again whenever `generation` (1..4095, +1 per call on the same workspaces) wraps; calls that share
 * workspaces must be ordered (one stream); P <= 2^20.  Outputs bit-identical to pmf_project_scatter. */
int pmf_project_scatter2(const float* points, const int32_t* sem, int64_t P, const uint8_t* image, int32_t h, int32_t w,
                         const double* proj, const int32_t* lut, int32_t nlut, float* proj_out, uint8_t* keep,
                         int32_t* x_data, int32_t* y_data, float* depth, int32_t* n_kept, uint32_t* pix_tag, uint64_t* slots,
                         int32_t generation, pmf_stream_t s);
/* validation crop/pad (perspective_view_loader.py:71-74,138-141): dst[c][oh][ow] window copy with zero fill */
int pmf_crop_pad(const float* src, int32_t C, int32_t h, int32_t w, int32_t top, int32_t left, float* dst,
                 int32_t oh, int32_t ow, int32_t pad_top, int32_t pad_left, int32_t ch, int32_t cw, pmf_stream_t s);

/* Multi-camera merge of per-point predictions (replaces tasks/pmf_eval_nuscenes/infer.py:18-38 getMergePred): for each
 * of pc_size LiDAR points the label of the camera with the highest confidence; point_idx[j] / conf[j] / label[j] are
 * device arrays of counts[j] entries for camera j (host arrays of device pointers; counts on the host); a camera that
 * does not list a point counts as confidence 0 / label -1, ties go to the lowest camera index (torch.argmax);
 * keys: u64[pc_size] scratch; merged: i64[pc_size], -1 where no camera decides. */
int pmf_merge_pred(int32_t n_cams, const int64_t* const* point_idx, const float* const* conf,
                   const int64_t* const* label, const int64_t* counts, int64_t pc_size, uint64_t* keys, int64_t* merged,
                   pmf_stream_t s);
/* the same with a per-point fallback label (int64[pc_size], e.g. the argmax of a LiDAR-only SalsaNext) for the points
 * no camera sees, instead of -1 (more_experiment_config.md:10) */
int pmf_merge_pred_fallback(int32_t n_cams, const int64_t* const* point_idx, const float* const* conf,
                            const int64_t* const* label, const int64_t* counts, int64_t pc_size, const int64_t* fallback,
                            uint64_t* keys, int64_t* merged, pmf_stream_t s);
/* SalsaNext range-image loader (replaces pc_processor/dataset/preprocess/projection.py:31-86 RangeProjection.doProjection,
 * salsanext_loader.py:48-84 and augmentor.py:97-180).
 * pmf_points_transform: in place on points f32[P][C] -- flips (x <- -x, y <- -y), float32 translation, then
 *   xyz <- float32(float64(xyz) . R^T) with rot9 = R row-major float64[9] (NULL: no rotation).
 * pmf_range_project_index: per point depth / column / row (float32 arithmetic of the reference, arctan2 / arcsin
 *   rounded once from float64), optional uproj_x/uproj_y/uproj_depth [P] (all three or none), and per pixel the key
 *   (depth bits << 32 | point index) of its nearest point in keys u64[H][W] (all ones = empty); fov_* are the float32
 *   values |fov_left|, |fov_left|+|fov_right|, |fov_down|, |fov_up|+|fov_down| in radians.
 * pmf_range_project_gather: per pixel range f32[H][W] (-1 empty), idx i32[H][W] (-1 empty), mask i32[H][W] = idx > 0,
 *   label f32[H][W] = mapped_label[idx] * mask, feature f32[5][H][W] = ((range,x,y,z,i*(i != -1)) - mean5) / std5 * mask,
 *   proj_points f32[H][W][C] (-1 empty); every output pointer may be NULL. */
int pmf_points_transform(float* points, int64_t P, int32_t C, int32_t flipx, int32_t flipy, float tx, float ty, float tz,
                         const double* rot9, pmf_stream_t s);
int pmf_range_project_index(const float* points, int64_t P, int32_t C, float fov_left_abs, float fov_h,
                            float fov_down_abs, float fov_v, int32_t H, int32_t W, uint64_t* keys, int32_t* uproj_x,
                            int32_t* uproj_y, float* uproj_depth, pmf_stream_t s);
int pmf_range_project_gather(const float* points, int64_t P, int32_t C, const uint64_t* keys, int32_t H, int32_t W,
                             const int32_t* mapped_label, const float* mean5, const float* std5, float* feature,
                             float* label, int32_t* mask, float* range, int32_t* idx, float* proj_points,
                             pmf_stream_t s);

/* training-time tensor augmentation of the [C,h,w] frame (perspective_view_loader.py:63-69,138-141: torchvision
 * RandomHorizontalFlip -> RandomRotation(nearest, zero fill) -> RandomCrop -> Pad) as one gather.  matrix6 (HOST pointer):
 * the inverse affine matrix [a b c; d e f] torchvision builds for the drawn angle ([cos r, sin r, 0, -sin r, cos r, 0],
 * r = radians(-angle)); crop window (top, left, crop_h, crop_w) inside the rotated h x w image; the crop lands at
 * (pad_top, pad_left) of dst f32[C][oh][ow], zero elsewhere. */
int pmf_flip_rotate_crop(const float* src, int32_t C, int32_t h, int32_t w, int32_t flip, const float* matrix6,
                         int32_t top, int32_t left, int32_t crop_h, int32_t crop_w, int32_t pad_top, int32_t pad_left,
                         float* dst, int32_t oh, int32_t ow, pmf_stream_t s);

/* torchvision ColorJitter on the PIL image (perspective_view_loader.py:46-49,84-85; perspective_view_loader_v2.py:19-23,
 * 46-47), applied in place to the uint8 [h][w][3] frame on the device, bit for bit what Pillow computes (ImageEnhance =
 * Image.blend with a degenerate image; hue through Pillow's HSV conversion), each operation on the uint8 result of the
 * previous one.  HOST arrays: order4 = the drawn permutation of {0 brightness, 1 contrast, 2 saturation, 3 hue};
 * factor4[op] / enabled4[op] = the drawn factor of operation op / whether it is applied (a zero-width range draws
 * nothing).  scratch: one device uint64 (luma sum of the contrast step).  `image` must be 4-byte aligned (dword accesses):
 * PMF_E_ARG otherwise -- the Python wrapper sends an unaligned slice through an aligned device copy. */
int pmf_color_jitter(uint8_t* image, int32_t h, int32_t w, const int32_t* order4, const double* factor4,
                     const int32_t* enabled4, uint64_t* scratch, pmf_stream_t s);

/* ---- loss-side kernels ------------------------------------------------------------------------------------ */
/* Lovasz-softmax Jaccard gradient (pc_processor/loss/lovasz_softmax.py:56-68) for C class rows at once.
 * fg_sorted f32[C][P]: 0/1 foreground indicator, each row ordered by DESCENDING error with ignored pixels last;
 * n_valid: device int64, number of non-ignored pixels; bsum: f32[C][ceil(P/4096)] scratch.
 * grad[c][i] = jaccard_i - jaccard_{i-1}, jaccard_i = 1 - (G - cumsum(fg)_i) / (G + cumsum(1-fg)_i); 0 for i >= n_valid. */
int pmf_lovasz_grad(const float* fg_sorted, int32_t C, int64_t P, const int64_t* n_valid, float* bsum, float* grad,
                    pmf_stream_t s);

/* PMF training objective for both heads, value AND gradient w.r.t. the two probability maps (tasks/pmf/trainer.py:
 * 231-252 perception-aware loss, 303-332 total; pc_processor/loss/focal_softmax.py:37-63; lovasz_softmax.py:132-160,
 * classes='present', ignore=0):   total = foc + foc_cam + lambda * (lov + lov_cam) + gamma_per * per.
 * Step 1, pmf_loss_pixel: label histogram cnt u64[C]; per pixel the entropies, focal terms, perception-aware KL terms
 *   and their analytic gradients (weights NOT detached, as in the reference) -> grad_lidar / grad_camera f32[N][C][HW];
 *   Lovasz sort keys key f32[2C][N*HW] (|fg - p|, -1 for ignored pixels; rows C.. = camera head); partial sums
 *   rows f64[pmf_loss_rows(P)][4]; optional confusion matrices conf_*[pred][label] += 1 (u64[C][C], argmax vs label).
 * Caller: ONE descending sort of the 2C key rows (values + permutation; rocprim via torch.sort).
 * Step 2, pmf_loss_lovasz: Jaccard first differences along the permutation, value = dot(sorted errors, grad) per
 *   class, gradient lambda / n_present * grad * d|fg-p|/dp added to grad_* through the permutation (no atomics);
 *   bsum f32[2C][pmf_loss_chunks(P)], dots f64[2C][pmf_loss_chunks(P)] scratch;
 *   out8 = {total, foc, lov, foc_cam, lov_cam, per, per_p, per_q} (per = per_p + per_q: the two KL halves of
 *   trainer.py:247-250).  Deterministic (fixed summation order).
 * The *_w variants take the weights of the six terms from DEVICE memory, w6 = {foc, lov, foc_cam, lov_cam, per_p,
 * per_q}: total = sum_i w6[i] * term_i -- the EPMF multi-task objective (tasks/epmf/trainer.py:409-430 with
 * pc_processor/loss/multi_task_loss.py: w_i = 1 / (2 sigma_i^2)) without a host round trip for the learned sigmas. */
int pmf_loss_rows(int64_t P);
int pmf_loss_chunks(int64_t P);
int pmf_loss_pixel(const float* lidar_prob, const float* camera_prob, const int64_t* label, const float* alpha,
                   int32_t N, int32_t C, int64_t HW, float focal_gamma, float tau, float gamma_per,
                   unsigned long long* cnt, float* grad_lidar, float* grad_camera, float* key, double* rows,
                   unsigned long long* conf_lidar, unsigned long long* conf_camera, pmf_stream_t s);
int pmf_loss_lovasz(const int64_t* perm, const float* key_sorted, const int64_t* label, int32_t N, int32_t C, int64_t HW,
                    const unsigned long long* cnt, float lambda, float gamma_per, float* bsum, double* dots,
                    const double* rows, float* grad_lidar, float* grad_camera, float* out8, pmf_stream_t s);
int pmf_loss_pixel_w(const float* lidar_prob, const float* camera_prob, const int64_t* label, const float* alpha,
                     int32_t N, int32_t C, int64_t HW, float focal_gamma, float tau, const float* w6,
                     unsigned long long* cnt, float* grad_lidar, float* grad_camera, float* key, double* rows,
                     unsigned long long* conf_lidar, unsigned long long* conf_camera, pmf_stream_t s);
int pmf_loss_lovasz_w(const int64_t* perm, const float* key_sorted, const int64_t* label, int32_t N, int32_t C, int64_t HW,
                      const unsigned long long* cnt, const float* w6, float* bsum, double* dots, const double* rows,
                      float* grad_lidar, float* grad_camera, float* out8, pmf_stream_t s);
/* The Lovasz stage WITH its sort (lovasz_softmax.py:71-145: torch.sort of the per-class errors, descending): takes the
 * unsorted key matrix pmf_loss_pixel wrote and sorts every used row -- classes that occur in the batch, pixels with a label
 * (the count P - cnt[0] is read on the device) -- by four stable 8-bit counting passes over the 30 significant bits of the
 * errors; equal errors keep ascending pixel order.  workspace: pmf_loss_sort_workspace(C, P) bytes.  Same outputs as
 * pmf_loss_lovasz (the Lovasz value does not depend on the order of equal errors). */
int64_t pmf_loss_sort_workspace(int32_t C, int64_t P);
int pmf_loss_lovasz_sort(const float* key, const int64_t* label, int32_t N, int32_t C, int64_t HW,
                         const unsigned long long* cnt, float lambda, float gamma_per, void* workspace, float* bsum,
                         double* dots, const double* rows, float* grad_lidar, float* grad_camera, float* out8,
                         pmf_stream_t s);
int pmf_loss_lovasz_sort_w(const float* key, const int64_t* label, int32_t N, int32_t C, int64_t HW,
                           const unsigned long long* cnt, const float* w6, void* workspace, float* bsum, double* dots,
                           const double* rows, float* grad_lidar, float* grad_camera, float* out8, pmf_stream_t s);

/* ---- plan executor: a whole forward (or backward) pass = one call --------------------------------------- */
enum {
  PMF_OP_CONV = 1, PMF_OP_WGRAD, PMF_OP_PACK, PMF_OP_BN_FINALIZE, PMF_OP_BN_EVAL, PMF_OP_BN_BWD_REDUCE,
  PMF_OP_BN_BWD_APPLY, PMF_OP_ADD_ACT, PMF_OP_ADD_ACT_BWD, PMF_OP_ACT_BWD, PMF_OP_AVGPOOL, PMF_OP_AVGPOOL_BWD,
  PMF_OP_MAXPOOL, PMF_OP_MAXPOOL_BWD, PMF_OP_BILINEAR, PMF_OP_BILINEAR_BWD, PMF_OP_PSHUFFLE, PMF_OP_PSHUFFLE_BWD,
  PMF_OP_GATE, PMF_OP_GATE_BWD, PMF_OP_GMEAN, PMF_OP_GMEAN_BWD, PMF_OP_COLSUM, PMF_OP_SOFTMAX, PMF_OP_SOFTMAX_BWD,
  PMF_OP_NCHW2NHWC, PMF_OP_FILL, PMF_OP_PMASK_FROM, PMF_OP_PMASK_POOL, PMF_OP_PMASK_MUL, PMF_OP_PMASK_MUL_BWD,
  PMF_OP_VEC_ADD, PMF_OP_WGRAD_PART, PMF_OP_WGRAD_RED, PMF_OP_WGRAD_RED_MULTI, PMF_OP_BN_BWD_FOLD, PMF_OP_BN_BWD_SMALL,
  PMF_OP_BCAST
};

/* generic argument record for the small ops (slot meaning documented next to each dispatcher case in plan.cpp) */
typedef struct {
  void* p[12];
  int64_t l[4];
  int32_t i[16];
  float f[4];
  pmf_view_t v[3];
} pmf_small_args_t;

typedef struct {
  int32_t kind;
  int32_t pad_;    /* scheduling bits for pmf_plan_run: bits 0-1 = lane (0 = the caller's stream; a side lane forks from it
                    * at its first op of the range and is joined at the end of the range); bits 8-15 = e+1: the op's lane
                    * first waits for plan event e; bits 16-23 = e+1: record plan event e on the op's lane after the op;
                    * bits 24-30 = duration estimate in units of 4 us (0 = unknown): pmf_plan_run hands the ops to the lanes'
                    * streams in the order of a simulated parallel execution (per-lane order and events are kept; only the
                    * interleaving of the lanes changes, which decides how a captured graph replays) */
  union {
    pmf_conv_desc_t conv;
    pmf_wgrad_desc_t wgrad;
    pmf_small_args_t sm;
  } u;
} pmf_op_t;

/* runs ops[0..n) in order on stream s; returns 0 or the first error (index in *failed_at if non-NULL) */
int pmf_plan_run(const pmf_op_t* ops, int32_t n, pmf_stream_t s, int32_t* failed_at);
int pmf_plan_run_range(const pmf_op_t* ops, int32_t begin, int32_t end, pmf_stream_t s, int32_t* failed_at);
/* hipGraph capture of ops[begin..end): pmf_graph_launch replays the whole range (the plan allocates nothing, copies
 * nothing, never synchronises, and all its pointers are fixed).  Run the range eagerly once before capturing.  The
 * handle is bound to the pointer values inside `ops` at capture time.  It holds one linear hipGraph per stretch of a
 * lane between cross-lane edges; a replay launches them on the lanes' streams with stream events between them
 * (PMF_GRAPH_MODE=single: one multi-branch hipGraph instead, see csrc/plan.cpp for why that is not the default). */
int pmf_plan_capture(const pmf_op_t* ops, int32_t begin, int32_t end, void** graph_exec, int32_t* failed_at);
int pmf_graph_launch(void* graph_exec, pmf_stream_t s);
int pmf_graph_pieces(void* graph_exec);   /* linear hipGraphs behind the handle */
/* stream `s` waits for plan event `e` (pad_ bits 16-23 of the op that records it) as recorded by the most recently ENQUEUED
 * run / replay: how a caller hangs work of its own -- the gradient all-reduce -- behind one op of a running plan.
 * PMF_E_UNSUPPORTED: the event was never recorded (lanes off): order behind the whole range instead. */
int pmf_plan_event_wait(int32_t e, pmf_stream_t s);
/* lanes (pmf_op_t.pad_): on = 0 run every op on the caller's stream, 1 honour the lane bits (default; PMF_LANES=0 in the
 * environment starts with 0), < 0 query only; returns the previous setting */
int pmf_plan_lanes(int on);
/* the order in which pmf_plan_run_range / pmf_plan_capture hand ops [begin, end) to the lanes' streams (pmf_op_t.pad_
 * bits 24-30); out: end - begin op indices.  Host-only, launches nothing. */
int pmf_plan_issue_order(const pmf_op_t* ops, int32_t begin, int32_t end, int32_t* out);
int pmf_graph_destroy(void* graph_exec);
/* ---- optimiser updates over a range of the flat training state -------------------------------------------------------
 * Replaces: torch.optim.AdamW(lidar_stream.parameters()).step() and torch.optim.SGD(camera parameters, nesterov=True)
 * .step() of the reference (tasks/pmf/trainer.py:80-98 construction, :336-337 the two step() calls), element by element
 * the arithmetic of torch/optim/adamw.py / sgd.py in float32.  `n` consecutive floats starting at each pointer; a range
 * launch lets the caller update the parameters whose gradients are already final (and all-reduced) while the backward
 * plan is still running.  `step`: device float holding the 1-based count of THIS step (bias corrections are computed from
 * it on the device, no host read).  pmf_sgd_range: first_step != 0 initialises the momentum buffer with the
 * (weight-decayed) gradient as torch does; nesterov needs momentum > 0 and dampening == 0 (PMF_E_ARG otherwise, torch
 * raises ValueError). */
int pmf_adamw_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                    double beta1, double beta2, double eps, double weight_decay, const float* step, pmf_stream_t s);
int pmf_sgd_range(float* param, const float* grad, float* momentum_buffer, int64_t n, double lr, double momentum,
                  double dampening, double weight_decay, int32_t nesterov, int32_t first_step, pmf_stream_t s);
/* Replaces the four element-wise passes of tasks/pmf/trainer.py:291-295 (tasks/epmf/trainer.py likewise):
 *   input_feature[:, 0:C] = (input_feature[:, 0:C] - mean) / std * mask.unsqueeze(1)      in place, one launch.
 * x: [N, >= C, H, W] float32 with contiguous channel planes, stride_n floats between samples (8 * HW for the 8-channel
 * feature tensor, C = 5); mask [N, HW]; mean / std [C].  Same float32 operation order as torch (bit-identical). */
int pmf_normalise_inplace(float* x, int64_t stride_n, const float* mask, const float* mean, const float* stdv, int32_t N,
                          int32_t C, int64_t HW, pmf_stream_t s);
/* pixel splits pmf_conv_wgrad will use for this descriptor (sizes `partial`) */
int pmf_conv_wgrad_nsplit(const pmf_wgrad_desc_t* d);
/* sizeof() of the structs above, for bindings to self-check: 0 src, 1 conv, 2 wgrad, 3 view, 4 small, 5 op, 6 pack job */
int pmf_sizeof(int which);

const char* pmf_version(void);

#ifdef __cplusplus
}
#endif
#endif
