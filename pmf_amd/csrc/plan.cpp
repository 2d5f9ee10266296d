// Plan executor: a whole forward or backward pass of the network is a flat array of pmf_op_t built once on the
// host (static shapes, static buffers); running it is ONE C call that enqueues every kernel on the caller's
// HIP stream -- no Python between launches, and the array can be captured into a hipGraph by the caller.
#include <vector>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "../../include/pmf_amd.h"

extern "C" int pmf_pack_weights_batched(const pmf_pack_job_t*, int32_t, int32_t, pmf_stream_t);

// PMF_SKIP_OPS=kind,kind,... (pmf_op_kind_t numbers): these launches are left out -- a WHAT-IF measurement ("how long is
// the step without the weight gradients / the BatchNorm finalize launches?"), the results of such a run are garbage
static uint64_t skip_mask() {
  static const uint64_t m = [] {
    uint64_t v = 0;
    const char* e = getenv("PMF_SKIP_OPS");
    while (e && *e) {
      const char* e0 = e;
      const long k = strtol(e, (char**)&e, 10);
      if (e == e0) break;                       // not a number: stop (round 5: "none" looped forever here)
      if (k > 0 && k < 64) v |= 1ull << k;
      while (*e == ',' || *e == ' ') ++e;
    }
    return v;
  }();
  return m;
}

// PMF_DUP_OPS=kind,kind,...: these launches are issued TWICE -- the what-if in the other direction ("what does one more
// launch of this kind per layer cost the step?"), for idempotent kinds only (BatchNorm finalize would move the running
// statistics twice: a timing experiment, like PMF_SKIP_OPS; unlike it the data stay finite and non-zero, so the clocks do not
// change with it -- skipping the finalize launches zeroes every activation behind them, and a chip multiplying zeros runs faster)
static uint64_t dup_mask() {
  static const uint64_t m = [] {
    uint64_t v = 0;
    const char* e = getenv("PMF_DUP_OPS");
    while (e && *e) {
      const char* e0 = e;
      const long k = strtol(e, (char**)&e, 10);
      if (e == e0) break;
      if (k > 0 && k < 64) v |= 1ull << k;
      while (*e == ',' || *e == ' ') ++e;
    }
    return v;
  }();
  return m;
}

static int run_one_(const pmf_op_t& o, pmf_stream_t s);
static int run_one(const pmf_op_t& o, pmf_stream_t s) {
  if (dup_mask() >> (o.kind & 63) & 1ull) {
    const int rc = run_one_(o, s);
    if (rc) return rc;
  }
  return run_one_(o, s);
}
static int run_one_(const pmf_op_t& o, pmf_stream_t s) {
  const pmf_small_args_t& a = o.u.sm;
  const int32_t* i = a.i;
  if (skip_mask() >> (o.kind & 63) & 1ull) return 0;
  switch (o.kind) {
    case PMF_OP_CONV: return pmf_conv_fwd(&o.u.conv, s);
    case PMF_OP_WGRAD: return pmf_conv_wgrad(&o.u.wgrad, s);
    case PMF_OP_WGRAD_PART: return pmf_conv_wgrad_partial(&o.u.wgrad, s);
    case PMF_OP_WGRAD_RED: return pmf_conv_wgrad_reduce(&o.u.wgrad, s);
    case PMF_OP_WGRAD_RED_MULTI:  // p0 descriptors(dev)  p1 meta(dev)  i0 njobs  i1 total_blocks
      return pmf_conv_wgrad_reduce_multi((const pmf_wgrad_desc_t*)a.p[0], (const int32_t*)a.p[1], i[0], i[1], s);
    case PMF_OP_PACK:  // p0 jobs(dev)  i0 njobs  i1 total_blocks
      return pmf_pack_weights_batched((const pmf_pack_job_t*)a.p[0], i[0], i[1], s);
    case PMF_OP_BN_FINALIZE:  // p: stats gamma beta rm rv scale shift save_mean save_invstd | f: count mom eps | i: C nrows
      return pmf_bn_finalize((const double*)a.p[0], i[1], a.f[0], (const float*)a.p[1], (const float*)a.p[2],
                             (float*)a.p[3], (float*)a.p[4], a.f[1], a.f[2], (float*)a.p[5], (float*)a.p[6],
                             (float*)a.p[7], (float*)a.p[8], i[0], s);
    case PMF_OP_BN_EVAL:  // p: gamma beta rm rv scale shift save_mean save_invstd | f0 eps | i0 C
      return pmf_bn_eval_affine((const float*)a.p[0], (const float*)a.p[1], (const float*)a.p[2], (const float*)a.p[3],
                                a.f[0], (float*)a.p[4], (float*)a.p[5], (float*)a.p[6], (float*)a.p[7], i[0], s);
    case PMF_OP_BN_BWD_REDUCE:  // p: gy a save_mean gamma save_invstd part coef dgamma dbeta | i: gy_ldc a_ldc C train | l0 npix
      return pmf_bn_bwd_reduce((const float*)a.p[0], i[0], (const float*)a.p[1], i[1], a.l[0], i[2],
                               (const float*)a.p[2], (const float*)a.p[3], (const float*)a.p[4], i[3],
                               (double*)a.p[5], (float*)a.p[6], (float*)a.p[7], (float*)a.p[8], s);
    case PMF_OP_BN_BWD_FOLD:  // p: part gamma save_invstd coef dgamma dbeta | i: C nrows train | l0 npix
      return pmf_bn_bwd_fold((const double*)a.p[0], i[1], i[0], a.l[0], i[2], (const float*)a.p[1], (const float*)a.p[2],
                             (float*)a.p[3], (float*)a.p[4], (float*)a.p[5], s);
    case PMF_OP_BN_BWD_SMALL:  // p: gy a save_mean gamma save_invstd dz dbias_row dgamma dbeta | i: gy_ldc a_ldc C train act dz_ldc | l0 npix
      return pmf_bn_bwd_small((const float*)a.p[0], i[0], (const float*)a.p[1], i[1], a.l[0], i[2], (const float*)a.p[2],
                              (const float*)a.p[3], (const float*)a.p[4], i[3], i[4], (float*)a.p[5], i[5], (float*)a.p[6],
                              (float*)a.p[7], (float*)a.p[8], s);
    case PMF_OP_BN_BWD_APPLY:  // p: gy a coef save_mean dz dbias_rows | i: gy_ldc a_ldc C act dz_ldc dbias_ld | l0 npix
      return pmf_bn_bwd_apply((const float*)a.p[0], i[0], (const float*)a.p[1], i[1], a.l[0], i[2], (const float*)a.p[2],
                              (const float*)a.p[3], i[3], (float*)a.p[4], i[4], (float*)a.p[5], i[5], s);
    case PMF_OP_ADD_ACT:  // v0 a, v1 b | p0 out | i: act out_ldc HW has_b C | l0 npix
      return pmf_add_act(&a.v[0], i[3] ? &a.v[1] : nullptr, i[0], (float*)a.p[0], i[1], a.l[0], i[2], i[4], s);
    case PMF_OP_ADD_ACT_BWD:  // p: gout out ga gb | i: g_ldc out_ldc act ga_ldc ga_acc gb_ldc gb_acc C
      return pmf_add_act_bwd((const float*)a.p[0], i[0], (const float*)a.p[1], i[1], i[2], (float*)a.p[2], i[3], i[4],
                             (float*)a.p[3], i[5], i[6], a.l[0], i[7], s);
    case PMF_OP_ACT_BWD:  // p: g a dbias_rows | i: g_ldc a_ldc act C dbias_ld
      return pmf_act_bwd((float*)a.p[0], i[0], (const float*)a.p[1], i[1], i[2], (float*)a.p[2], i[4], a.l[0], i[3], s);
    case PMF_OP_AVGPOOL:  // v0 | p0 out | i: N H W C out_ldc
      return pmf_avgpool3s2(&a.v[0], i[0], i[1], i[2], i[3], (float*)a.p[0], i[4], s);
    case PMF_OP_AVGPOOL_BWD:  // p: gout cmul gin | i: g_ldc N H W C cmul_ld gin_ldc acc
      return pmf_avgpool3s2_bwd((const float*)a.p[0], i[0], i[1], i[2], i[3], i[4], (const float*)a.p[1], i[5],
                                (float*)a.p[2], i[6], i[7], s);
    case PMF_OP_MAXPOOL:  // v0 | p: out idx | i: N H W C out_ldc
      return pmf_maxpool3s2(&a.v[0], i[0], i[1], i[2], i[3], (float*)a.p[0], i[4], (uint8_t*)a.p[1], s);
    case PMF_OP_MAXPOOL_BWD:  // v0 | p: gout idx gin | i: g_ldc N H W C gin_ldc acc
      return pmf_maxpool3s2_bwd((const float*)a.p[0], i[0], (const uint8_t*)a.p[1], i[1], i[2], i[3], i[4], &a.v[0],
                                (float*)a.p[2], i[5], i[6], s);
    case PMF_OP_BILINEAR:  // v0 | p0 out | i: N H W C out_ldc
      return pmf_bilinear2x(&a.v[0], i[0], i[1], i[2], i[3], (float*)a.p[0], i[4], s);
    case PMF_OP_BILINEAR_BWD:  // p: gout gin | i: g_ldc N H W C gin_ldc acc
      return pmf_bilinear2x_bwd((const float*)a.p[0], i[0], i[1], i[2], i[3], i[4], (float*)a.p[1], i[5], i[6], s);
    case PMF_OP_PSHUFFLE:  // v0 | p: out_cmul out | i: N H W Cout out_cmul_ld out_ldc
      return pmf_pixel_shuffle2(&a.v[0], i[0], i[1], i[2], i[3], (const float*)a.p[0], i[4], (float*)a.p[1], i[5], s);
    case PMF_OP_PSHUFFLE_BWD:  // p: gout out_cmul in_cmul gin | i: g_ldc N H W Cout out_cmul_ld in_cmul_ld gin_ldc acc
      return pmf_pixel_shuffle2_bwd((const float*)a.p[0], i[0], i[1], i[2], i[3], i[4], (const float*)a.p[1], i[5],
                                    (const float*)a.p[2], i[6], (float*)a.p[3], i[7], i[8], s);
    case PMF_OP_GATE:  // v0 f, v1 att | p: pcd out | i: pcd_ldc out_ldc C | l0 npix
      return pmf_fusion_gate(&a.v[0], &a.v[1], (const float*)a.p[0], i[0], (float*)a.p[1], i[1], a.l[0], i[2], s);
    case PMF_OP_GATE_BWD:  // v0 v1 | p: gout gf gatt gpcd | i: g_ldc gf_ldc gf_acc gatt_ldc gpcd_ldc gpcd_acc C
      return pmf_fusion_gate_bwd((const float*)a.p[0], i[0], &a.v[0], &a.v[1], (float*)a.p[1], i[1], i[2],
                                 (float*)a.p[2], i[3], (float*)a.p[3], i[4], i[5], a.l[0], i[6], s);
    case PMF_OP_GMEAN:  // v0 | p0 out | i: N HW C
      return pmf_global_mean(&a.v[0], i[0], i[1], i[2], (float*)a.p[0], s);
    case PMF_OP_GMEAN_BWD:  // p: gout cmul gin | i: N HW C cmul_ld gin_ldc acc
      return pmf_global_mean_bwd((const float*)a.p[0], i[0], i[1], i[2], (const float*)a.p[1], i[3], (float*)a.p[2],
                                 i[4], i[5], s);
    case PMF_OP_COLSUM:  // p: x out [scratch] | i: ldc C nz | l0 npix
      if (a.p[2]) return pmf_colsum_rows((const float*)a.p[0], i[0], a.l[0], i[1], (float*)a.p[1], i[2], (float*)a.p[2], s);
      return pmf_colsum((const float*)a.p[0], i[0], a.l[0], i[1], (float*)a.p[1], i[2], s);
    case PMF_OP_SOFTMAX:  // p: logits prob | i: ldc N HW C ident (1: the logits go out as they are)
      if (i[4]) return pmf_logits_nhwc_to_nchw((const float*)a.p[0], i[0], i[1], i[2], i[3], (float*)a.p[1], s);
      return pmf_softmax_nhwc_to_nchw((const float*)a.p[0], i[0], i[1], i[2], i[3], (float*)a.p[1], s);
    case PMF_OP_SOFTMAX_BWD:  // p: prob g dlogits | i: N HW C ldc ident
      if (i[4]) return pmf_logits_bwd_nchw_to_nhwc((const float*)a.p[1], i[0], i[1], i[2], (float*)a.p[2], i[3], s);
      return pmf_softmax_bwd_nchw_to_nhwc((const float*)a.p[0], (const float*)a.p[1], i[0], i[1], i[2], (float*)a.p[2],
                                          i[3], s);
    case PMF_OP_NCHW2NHWC:  // p: x out | l: stride_n stride_c | i: N C HW out_ldc
      return pmf_nchw_to_nhwc((const float*)a.p[0], a.l[0], a.l[1], i[0], i[1], i[2], (float*)a.p[1], i[3], s);
    case PMF_OP_FILL:  // p0 | f0 value | l0 count
      return pmf_fill((float*)a.p[0], a.f[0], a.l[0], s);
    case PMF_OP_PMASK_FROM:  // v0 in | p: mask | i: HW C | l0 npix
      return pmf_pmask_from(&a.v[0], a.l[0], i[0], i[1], (float*)a.p[0], s);
    case PMF_OP_PMASK_POOL:  // p: mask out | i: N H W kh kw dil pad stride OH OW
      return pmf_pmask_pool((const float*)a.p[0], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], (float*)a.p[1], i[8], i[9], s);
    case PMF_OP_PMASK_MUL:  // v0 in | p: mask out | i: HW C out_ldc | l0 npix
      return pmf_pmask_mul(&a.v[0], (const float*)a.p[0], a.l[0], i[0], i[1], (float*)a.p[1], i[2], s);
    case PMF_OP_PMASK_MUL_BWD:  // p: gy mask gx | i: gy_ldc C gx_ldc acc | l0 npix
      return pmf_pmask_mul_bwd((const float*)a.p[0], i[0], (const float*)a.p[1], a.l[0], i[1], (float*)a.p[2], i[2], i[3], s);
    case PMF_OP_BCAST:  // p: src out | i: src_ldc N C out_ldc | l0 HW
      return pmf_broadcast_rows((const float*)a.p[0], i[0], i[1], a.l[0], i[2], (float*)a.p[1], i[3], s);
    case PMF_OP_VEC_ADD:  // p: a b out | i: n
      return pmf_vec_add((const float*)a.p[0], (const float*)a.p[1], (float*)a.p[2], i[0], s);
    default: return PMF_E_ARG;
  }
}

// Lanes (pmf_op_t.pad_): a plan may spread independent branches of the network (the camera stream and the LiDAR stream
// of PMFNet are independent between fusion points) over up to PMF_MAX_LANES HIP streams so that the ramp-up, tail and
// latency-bound small kernels of one branch run under the machine-filling kernels of the other.
//   bits 0-1   lane of the op (0 = the caller's stream)
//   bits 8-15  e+1: the op's lane first waits for plan event e      (recorded earlier in the SAME range by another lane)
//   bits 16-23 e+1: plan event e is recorded on the op's lane after the op
// A side lane forks from the main stream at its first op of the range (it sees everything issued on the main stream up to
// that point) and is joined to the main stream at the end of the range.  Events never reach across ranges.  hipGraph
// capture records the fork / join / event edges, so a replayed range is a DAG with one branch per lane.
// PMF_LANES=0 runs everything on the caller's stream.
#define PMF_MAX_LANES 4
#define PMF_MAX_EVENTS 255
static hipStream_t g_lane[PMF_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
static hipEvent_t g_fork = nullptr, g_join[PMF_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
static hipEvent_t g_ev[PMF_MAX_EVENTS];
static int g_nev = 0;
// true while the latest enqueued run recorded its plan events only as NODES of a multi-branch hipGraph ("single" mode): a
// stream wait on such an event would look at a stale or never-made stream record -- pmf_plan_event_wait refuses
static bool g_ev_graph_only = false;
// The lane streams and events above belong to ONE device (one process per GPU): they are created on the
// device that is current at the first lane use, and a plan run with another device current is refused instead of being
// enqueued on foreign streams.  (The lazily created objects are not guarded against concurrent first use either: plans
// are run from one thread per process.)
static int g_lane_dev = -1;
static int lane_device_ok() {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (g_lane_dev < 0) g_lane_dev = dev;
  return dev == g_lane_dev ? 0 : PMF_E_UNSUPPORTED;
}
static int lane_stream(int lane, hipStream_t* out) {
  if (int rc = lane_device_ok()) return rc;
  if (!g_lane[lane]) {
    hipError_t e = hipStreamCreateWithFlags(&g_lane[lane], hipStreamNonBlocking);
    if (e != hipSuccess) return (int)e;
    e = hipEventCreateWithFlags(&g_join[lane], hipEventDisableTiming);
    if (e != hipSuccess) return (int)e;
  }
  if (!g_fork) {
    hipError_t e = hipEventCreateWithFlags(&g_fork, hipEventDisableTiming);
    if (e != hipSuccess) return (int)e;
  }
  *out = g_lane[lane];
  return 0;
}
static int plan_event(int e, hipEvent_t* out) {
  if (int rc = lane_device_ok()) return rc;
  while (g_nev <= e) {
    hipError_t r = hipEventCreateWithFlags(&g_ev[g_nev], hipEventDisableTiming);
    if (r != hipSuccess) return (int)r;
    ++g_nev;
  }
  *out = g_ev[e];
  return 0;
}
static int g_lanes_on = -1;
static bool lanes_enabled() {
  if (g_lanes_on < 0) { const char* e = getenv("PMF_LANES"); g_lanes_on = (e && e[0] == '0') ? 0 : 1; }
  return g_lanes_on == 1;
}
// on = 0 / 1: run every op on the caller's stream / honour the lane bits; on < 0: query.  Returns the previous setting.
// (Per-op profiling switches the lanes off so that an event pair on the caller's stream brackets exactly one kernel.)
extern "C" int pmf_plan_lanes(int on) {
  const int prev = lanes_enabled() ? 1 : 0;
  if (on >= 0) g_lanes_on = on ? 1 : 0;
  return prev;
}

// Issue order of a range.  The lane bits and events define the dependencies; the ORDER in which the ops are handed to
// the streams is free as long as every lane keeps its own order and an event is recorded before it is awaited -- and it
// matters: hipGraph replay (ROCm 7.2) enqueues the nodes in capture order and resolves an edge between two of its
// internal streams against the TAIL of the source stream at the moment the dependent node is enqueued.  Captured in list
// order (the camera encoder's backward behind the whole LiDAR backward, a deferred weight-gradient batch in front of the
// home lane's next op), a consumer therefore waited for everything its producer's lane had been handed by then: the
// camera encoder's backward ran AFTER the LiDAR backward instead of under it (3.5 ms of latency-bound launches alone on
// the chip, profiles/r03_lanes_*.txt).  The ops are issued in the order of a simulated parallel execution instead:
// per-lane clocks advanced by the ops' cost hints (pad_ bits 24-30, units of 4 us, from the plan builder's flop / byte
// counts), an op starts at max(its lane's clock, the record time of the event it waits for), and the op with the
// earliest start goes next.  PMF_PLAN_ORDER=list keeps the list order (A/B).
static int issue_order(const pmf_op_t* ops, int32_t begin, int32_t end, bool lanes, std::vector<int32_t>& order) {
  const int32_t n = end - begin;
  order.resize(n > 0 ? n : 0);
  static const bool keep_list = [] { const char* e = getenv("PMF_PLAN_ORDER"); return e && e[0] == 'l'; }();
  if (!lanes || keep_list || n <= 0) {
    for (int32_t i = 0; i < n; ++i) order[i] = begin + i;
    return 0;
  }
  std::vector<int32_t> q[PMF_MAX_LANES];
  size_t cur[PMF_MAX_LANES] = {0, 0, 0, 0};
  double clock[PMF_MAX_LANES] = {0.0, 0.0, 0.0, 0.0};
  int32_t rec_at[PMF_MAX_EVENTS];
  bool rec_done[PMF_MAX_EVENTS] = {};
  double rec_clock[PMF_MAX_EVENTS] = {};
  for (int e = 0; e < PMF_MAX_EVENTS; ++e) rec_at[e] = -1;
  for (int32_t k = begin; k < end; ++k) {
    q[ops[k].pad_ & 3].push_back(k);
    const int r = ((ops[k].pad_ >> 16) & 0xff) - 1;
    if (r >= 0 && rec_at[r] < 0) rec_at[r] = k;
  }
  // fork of a side lane: it must see every main-lane op that precedes its first op in the list
  int32_t need_main[PMF_MAX_LANES] = {0, 0, 0, 0};
  double fork_clock[PMF_MAX_LANES] = {0.0, 0.0, 0.0, 0.0};
  bool forked[PMF_MAX_LANES] = {true, false, false, false};
  for (int l = 1; l < PMF_MAX_LANES; ++l) {
    if (q[l].empty()) continue;
    for (int32_t k : q[0]) if (k < q[l][0]) ++need_main[l];
    if (need_main[l] == 0) forked[l] = true;
  }
  for (int32_t out = 0; out < n; ++out) {
    int best = -1;
    double best_t = 0.0;
    for (int l = 0; l < PMF_MAX_LANES; ++l) {
      if (cur[l] >= q[l].size()) continue;
      const int32_t k = q[l][cur[l]];
      double t = clock[l];
      if (cur[l] == 0 && l > 0) {
        if (!forked[l]) continue;
        if (fork_clock[l] > t) t = fork_clock[l];
      }
      const int w = ((ops[k].pad_ >> 8) & 0xff) - 1;
      if (w >= 0 && rec_at[w] >= 0 && rec_at[w] < k) {     // (a wait for an event not recorded earlier in the range is void)
        if (!rec_done[w]) continue;
        if (rec_clock[w] > t) t = rec_clock[w];
      }
      if (best < 0 || t < best_t || (t == best_t && k < q[best][cur[best]])) { best = l; best_t = t; }
    }
    // a well-formed op array always has a ready op (the unissued op with the smallest list index); event bits that do
    // not come from the plan builder (a wait and its record on ops of one lane in the wrong order, a foreign binding) may
    // not: refuse instead of indexing q[-1]
    if (best < 0) return PMF_E_ARG;
    const int32_t k = q[best][cur[best]++];
    order[out] = k;
    const int cost = (ops[k].pad_ >> 24) & 0x7f;
    clock[best] = best_t + (cost ? cost : 1);
    const int r = ((ops[k].pad_ >> 16) & 0xff) - 1;
    if (r >= 0) { rec_done[r] = true; rec_clock[r] = clock[best]; }
    if (best == 0)
      for (int l = 1; l < PMF_MAX_LANES; ++l)
        if (!forked[l] && !q[l].empty() && (int32_t)cur[0] >= need_main[l]) { forked[l] = true; fork_clock[l] = clock[0]; }
  }
  return 0;
}

// the order in which pmf_plan_run_range / pmf_plan_capture hand ops [begin, end) to the streams (host-only: no launch)
extern "C" int pmf_plan_issue_order(const pmf_op_t* ops, int32_t begin, int32_t end, int32_t* out) {
  if (!ops || !out || begin < 0 || end < begin) return PMF_E_ARG;
  std::vector<int32_t> order;
  if (int rc = issue_order(ops, begin, end, lanes_enabled(), order)) return rc;
  for (size_t i = 0; i < order.size(); ++i) out[i] = order[i];
  return 0;
}

static int run_range(const pmf_op_t* ops, int32_t begin, int32_t end, hipStream_t main_s, int32_t* failed_at) {
  const bool lanes = lanes_enabled();
  bool used[PMF_MAX_LANES] = {true, false, false, false};
  bool ev_valid[PMF_MAX_EVENTS] = {};      // events recorded inside THIS range (a wait never reaches back further)
  hipStream_t st[PMF_MAX_LANES] = {main_s, nullptr, nullptr, nullptr};
  int rc = 0;
  std::vector<int32_t> order;
  if ((rc = issue_order(ops, begin, end, lanes, order)) != 0) { if (failed_at) *failed_at = begin; return rc; }
  int32_t rec_at[PMF_MAX_EVENTS];          // list position of an event's record op inside this range (-1: none)
  for (int i = 0; i < PMF_MAX_EVENTS; ++i) rec_at[i] = -1;
  for (int32_t j = begin; j < end; ++j) {
    const int r = ((ops[j].pad_ >> 16) & 0xff) - 1;
    if (r >= 0 && rec_at[r] < 0) rec_at[r] = j;
  }
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main_s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) g_ev_graph_only = false;  // eager: real records
  }
  int32_t k = begin;
  for (size_t oi = 0; oi < order.size() && rc == 0; ++oi) {
    k = order[oi];
    const int bits = ops[k].pad_;
    const int lane = lanes ? (bits & 3) : 0;
    const int wait_e = ((bits >> 8) & 0xff) - 1, rec_e = ((bits >> 16) & 0xff) - 1;
    if (lane && !used[lane]) {             // fork: the lane starts from the main stream's current position
      rc = lane_stream(lane, &st[lane]);
      if (rc == 0) rc = (int)hipEventRecord(g_fork, main_s);
      if (rc == 0) rc = (int)hipStreamWaitEvent(st[lane], g_fork, 0);
      if (rc) break;
      used[lane] = true;
    }
    // (a wait counts only for a record EARLIER IN LIST ORDER, the rule of issue_order and pmf_plan_capture: the eager run
    // and the captured replay build the same dependency graph)
    if (lanes && wait_e >= 0 && rec_at[wait_e] >= 0 && rec_at[wait_e] < k && ev_valid[wait_e]) {
      hipEvent_t ev;
      rc = plan_event(wait_e, &ev);
      if (rc == 0) rc = (int)hipStreamWaitEvent(st[lane], ev, 0);
      if (rc) break;
    }
    rc = run_one(ops[k], (pmf_stream_t)st[lane]);
    if (rc) break;
    if (lanes && rec_e >= 0) {
      hipEvent_t ev;
      rc = plan_event(rec_e, &ev);
      if (rc == 0) rc = (int)hipEventRecord(ev, st[lane]);
      if (rc) break;
      ev_valid[rec_e] = true;
    }
  }
  if (rc != 0 && failed_at) *failed_at = k;
  for (int l = 1; l < PMF_MAX_LANES; ++l) {      // join (also on failure: never leave a forked capture behind)
    if (!used[l]) continue;
    hipError_t e = hipEventRecord(g_join[l], st[l]);
    if (e == hipSuccess) e = hipStreamWaitEvent(main_s, g_join[l], 0);
    if (e != hipSuccess && rc == 0) rc = (int)e;
  }
  return rc;
}

extern "C" int pmf_plan_run(const pmf_op_t* ops, int32_t n, pmf_stream_t s, int32_t* failed_at) {
  return run_range(ops, 0, n, (hipStream_t)s, failed_at);
}

extern "C" int pmf_plan_run_range(const pmf_op_t* ops, int32_t begin, int32_t end, pmf_stream_t s, int32_t* failed_at) {
  return run_range(ops, begin, end, (hipStream_t)s, failed_at);
}

// ---- hipGraph capture of a plan range --------------------------------------------------------------------------
// A plan performs no allocation, memcpy or synchronisation and every pointer it uses is fixed at build time, so a
// range of ops can be captured ONCE (on a private capture stream: the legacy default stream cannot capture) and
// replayed.  Issuing the ~900 kernels of one training iteration one by one costs ~19 ms of host time per step on
// this stack -- about as much as the GPU needs to run them -- so without the graph the GPU starves whenever the host
// thread does anything else (optimiser, loss, next batch).
// The caller must have run the range eagerly once before (kernel attributes are set on first launch, which is not
// allowed inside a capture).
//
// Two forms (PMF_GRAPH_MODE):
//  * "single": the whole range, lanes included, is ONE multi-branch hipGraph.  Measured on ROCm 7.2: the replay resolves
//    an edge between two branches against whatever the source branch has been handed when the runtime reaches the
//    dependent node in ITS traversal of the DAG, which walks one branch to its end before the next -- the camera
//    encoder's backward (waiting for the image gradient of the deepest fusion block, a third of the way into the LiDAR
//    backward) started when the LiDAR backward had finished, and a deferred weight-gradient batch stalled the lane it
//    was taken off (profiles/r03_lanes_single.txt).
//  * "segments" (default): every lane is cut at its cross-lane edges (event waits, event records, fork points) into
//    LINEAR pieces; each piece is its own single-branch hipGraph, and a replay launches the pieces on the lanes' real
//    HIP streams with real events between them, in the issue order of the simulated parallel execution above.  A
//    linear graph is a batch of packets on one queue, and a stream event names exactly the piece it was recorded
//    behind, so the DAG executes as written.
struct PlanSeg {
  int lane;
  std::vector<int32_t> ops;
  hipGraphExec_t exec;
};
enum { ACT_START = 0, ACT_WAIT_FORK, ACT_WAIT_EV, ACT_LAUNCH, ACT_REC_EV, ACT_REC_FORK };
struct PlanAct { int kind, lane, arg; };
struct PlanProgram {
  std::vector<PlanSeg> segs;
  std::vector<PlanAct> acts;
  bool used[PMF_MAX_LANES];
  hipGraphExec_t single;     // "single" mode
};
static hipEvent_t g_forkev[PMF_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};

static void program_destroy(PlanProgram* p) {
  if (!p) return;
  for (auto& sg : p->segs) if (sg.exec) (void)hipGraphExecDestroy(sg.exec);
  if (p->single) (void)hipGraphExecDestroy(p->single);
  delete p;
}

static int capture_linear(const pmf_op_t* ops, const std::vector<int32_t>& idx, hipStream_t cap, hipGraphExec_t* out,
                          int32_t* failed_at) {
  hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return (int)e;
  int rc = 0;
  for (int32_t k : idx) {
    rc = run_one(ops[k], (pmf_stream_t)cap);
    if (rc) { if (failed_at) *failed_at = k; break; }
  }
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(cap, &graph);
  if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) return (int)e;
  e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  return (int)e;
}

extern "C" int pmf_plan_capture(const pmf_op_t* ops, int32_t begin, int32_t end, void** graph_exec, int32_t* failed_at) {
  static hipStream_t cap = nullptr;
  if (!graph_exec) return PMF_E_ARG;
  *graph_exec = nullptr;
  hipError_t e;
  if (!cap) {
    e = hipStreamCreateWithFlags(&cap, hipStreamNonBlocking);
    if (e != hipSuccess) return (int)e;
  }
  static const bool single = [] { const char* m = getenv("PMF_GRAPH_MODE"); return m && m[0] == 's' && m[1] == 'i'; }();
  PlanProgram* prog = new PlanProgram();
  prog->single = nullptr;
  for (int l = 0; l < PMF_MAX_LANES; ++l) prog->used[l] = (l == 0);
  if (single) {
    e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { delete prog; return (int)e; }
    int rc = run_range(ops, begin, end, cap, failed_at);   // side-lane ops fork / join inside the capture
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(cap, &graph);
    if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); delete prog; return rc; }
    if (e != hipSuccess) { delete prog; return (int)e; }
    e = hipGraphInstantiate(&prog->single, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { delete prog; return (int)e; }
    *graph_exec = (void*)prog;
    return 0;
  }
  // ---- cut the lanes into linear pieces, in issue order
  const bool lanes = lanes_enabled();
  std::vector<int32_t> order;
  if (int rc = issue_order(ops, begin, end, lanes, order)) { delete prog; return rc; }
  int32_t rec_at[PMF_MAX_EVENTS];
  for (int i = 0; i < PMF_MAX_EVENTS; ++i) rec_at[i] = -1;
  int32_t first_of[PMF_MAX_LANES] = {-1, -1, -1, -1}, fork_after[PMF_MAX_LANES] = {-1, -1, -1, -1};
  for (int32_t k = begin; k < end; ++k) {
    const int l = lanes ? (ops[k].pad_ & 3) : 0;
    if (first_of[l] < 0) first_of[l] = k;
    const int r = ((ops[k].pad_ >> 16) & 0xff) - 1;
    if (lanes && r >= 0 && rec_at[r] < 0) rec_at[r] = k;
  }
  for (int l = 1; l < PMF_MAX_LANES; ++l) {       // the main-lane op a side lane forks behind (list order), if any
    if (first_of[l] < 0) continue;
    for (int32_t k = begin; k < first_of[l]; ++k)
      if ((lanes ? (ops[k].pad_ & 3) : 0) == 0) fork_after[l] = k;
  }
  int open[PMF_MAX_LANES] = {-1, -1, -1, -1};
  prog->acts.push_back({ACT_START, 0, 0});
  for (int32_t k : order) {
    const int bits = ops[k].pad_;
    const int l = lanes ? (bits & 3) : 0;
    int w = lanes ? ((bits >> 8) & 0xff) - 1 : -1;
    const int r = lanes ? ((bits >> 16) & 0xff) - 1 : -1;
    if (w >= 0 && !(rec_at[w] >= 0 && rec_at[w] < k)) w = -1;     // (not recorded earlier in this range: void)
    if (l > 0 && !prog->used[l]) {
      prog->used[l] = true;
      prog->acts.push_back({ACT_WAIT_FORK, l, fork_after[l] >= 0 ? l : 0});
    }
    if (w >= 0 && open[l] >= 0) open[l] = -1;
    if (open[l] < 0) {
      if (w >= 0) prog->acts.push_back({ACT_WAIT_EV, l, w});
      open[l] = (int)prog->segs.size();
      prog->segs.push_back(PlanSeg{l, {}, nullptr});
      prog->acts.push_back({ACT_LAUNCH, l, open[l]});
    }
    prog->segs[open[l]].ops.push_back(k);
    if (r >= 0) {
      open[l] = -1;
      prog->acts.push_back({ACT_REC_EV, l, r});
    }
    if (l == 0)
      for (int sl = 1; sl < PMF_MAX_LANES; ++sl)
        if (fork_after[sl] == k) {
          open[0] = -1;
          prog->acts.push_back({ACT_REC_FORK, 0, sl});
        }
  }
  for (auto& sg : prog->segs) {
    int rc = capture_linear(ops, sg.ops, cap, &sg.exec, failed_at);
    if (rc) { program_destroy(prog); return rc; }
  }
  *graph_exec = (void*)prog;
  return 0;
}

extern "C" int pmf_graph_launch(void* graph_exec, pmf_stream_t s) {
  if (!graph_exec) return PMF_E_ARG;
  PlanProgram* prog = (PlanProgram*)graph_exec;
  hipStream_t main_s = (hipStream_t)s;
  if (prog->single) { g_ev_graph_only = true; return (int)hipGraphLaunch(prog->single, main_s); }
  g_ev_graph_only = false;
  hipStream_t st[PMF_MAX_LANES] = {main_s, nullptr, nullptr, nullptr};
  int rc = 0;
  for (int l = 1; l < PMF_MAX_LANES && rc == 0; ++l) {
    if (!prog->used[l]) continue;
    rc = lane_stream(l, &st[l]);
    if (rc == 0 && !g_forkev[l]) rc = (int)hipEventCreateWithFlags(&g_forkev[l], hipEventDisableTiming);
  }
  // the lane streams and their events belong to ONE device (the first that used a side lane in this process); a range
  // without side lanes (PMF_LANES=0, or a plan on a second GPU of the process) touches none of them and replays anywhere,
  // as its eager run does
  bool any_side = false;
  for (int l = 1; l < PMF_MAX_LANES; ++l) any_side = any_side || prog->used[l];
  if (rc == 0 && any_side && !g_fork) rc = (int)hipEventCreateWithFlags(&g_fork, hipEventDisableTiming);
  if (rc == 0 && any_side) rc = lane_device_ok();
  for (size_t i = 0; i < prog->acts.size() && rc == 0; ++i) {
    const PlanAct& a = prog->acts[i];
    hipEvent_t ev;
    switch (a.kind) {
      case ACT_START: if (any_side) rc = (int)hipEventRecord(g_fork, main_s); break;
      case ACT_WAIT_FORK: rc = (int)hipStreamWaitEvent(st[a.lane], a.arg ? g_forkev[a.arg] : g_fork, 0); break;
      case ACT_WAIT_EV:
        rc = plan_event(a.arg, &ev);
        if (rc == 0) rc = (int)hipStreamWaitEvent(st[a.lane], ev, 0);
        break;
      case ACT_LAUNCH: rc = (int)hipGraphLaunch(prog->segs[a.arg].exec, st[a.lane]); break;
      case ACT_REC_EV:
        rc = plan_event(a.arg, &ev);
        if (rc == 0) rc = (int)hipEventRecord(ev, st[a.lane]);
        break;
      case ACT_REC_FORK: rc = (int)hipEventRecord(g_forkev[a.arg], main_s); break;
      default: rc = PMF_E_ARG;
    }
  }
  for (int l = 1; l < PMF_MAX_LANES; ++l) {      // join (also on failure)
    if (!prog->used[l] || !st[l]) continue;
    hipError_t e = hipEventRecord(g_join[l], st[l]);
    if (e == hipSuccess) e = hipStreamWaitEvent(main_s, g_join[l], 0);
    if (e != hipSuccess && rc == 0) rc = (int)e;
  }
  return rc;
}

extern "C" int pmf_graph_destroy(void* graph_exec) {
  program_destroy((PlanProgram*)graph_exec);
  return 0;
}

// make stream `s` wait for plan event `e` as last recorded by a plan run / graph replay that has ALREADY been enqueued (the
// data-parallel engine gates the all-reduce of a gradient range on the event behind the launch that finalises it, instead
// of cutting the backward plan into segments).  PMF_E_UNSUPPORTED when the event has never been recorded (lanes off).
extern "C" int pmf_plan_event_wait(int32_t e, pmf_stream_t s) {
  if (e < 0 || e >= PMF_MAX_EVENTS) return PMF_E_ARG;
  if (e >= g_nev || !lanes_enabled() || g_ev_graph_only) return PMF_E_UNSUPPORTED;
  if (int rc = lane_device_ok()) return rc;
  return (int)hipStreamWaitEvent((hipStream_t)s, g_ev[e], 0);
}

// number of linear pieces of a captured range (1 in "single" mode); tests / diagnostics
extern "C" int pmf_graph_pieces(void* graph_exec) {
  if (!graph_exec) return PMF_E_ARG;
  PlanProgram* prog = (PlanProgram*)graph_exec;
  return prog->single ? 1 : (int)prog->segs.size();
}

// ABI self-check for language bindings: sizes of the structs a binding has to mirror
extern "C" int pmf_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(pmf_src_t);
    case 1: return (int)sizeof(pmf_conv_desc_t);
    case 2: return (int)sizeof(pmf_wgrad_desc_t);
    case 3: return (int)sizeof(pmf_view_t);
    case 4: return (int)sizeof(pmf_small_args_t);
    case 5: return (int)sizeof(pmf_op_t);
    case 6: return (int)sizeof(pmf_pack_job_t);
    default: return -1;
  }
}

extern "C" const char* pmf_version(void) { return "pmf_amd 0.1 (gfx950)"; }
