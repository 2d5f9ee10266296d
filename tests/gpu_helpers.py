"""Helpers for the -m gpu parity tests: call single C-ABI entry points on torch CUDA tensors and
compare whole plans against the CPU oracle tensor-by-tensor."""
import ctypes as C

import numpy as np
import torch

from pmf_amd import _lib as L


def nhwc(x, ldc=None):
    """NCHW cpu tensor -> contiguous NHWC cuda tensor with channel padding to ldc."""
    n, c, h, w = x.shape
    ldc = ldc or (c + 7) // 8 * 8
    out = torch.zeros(n, h, w, ldc)
    out[..., :c] = x.permute(0, 2, 3, 1)
    return out.cuda().contiguous()


def from_nhwc(t, c):
    return t[..., :c].permute(0, 3, 1, 2).contiguous().cpu()


def pack_fwd(w, k_pad, ldw):
    """OIHW -> [taps][k_pad][ldw] (host-side reference packing for single-op tests)."""
    co, ci, kh, kw = w.shape
    p = torch.zeros(kh * kw, k_pad, ldw)
    p[:, :ci, :co] = w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co)
    return p.cuda().contiguous()


def split_bf16(x):
    """fp32 tensor -> its three bf16 planes (round-to-nearest-even, exact residuals): x ~= h1 + h2 + h3."""
    h1 = x.to(torch.bfloat16)
    r1 = x - h1.float()
    h2 = r1.to(torch.bfloat16)
    r2 = r1 - h2.float()
    h3 = r2.to(torch.bfloat16)
    return h1, h2, h3


def pack_fwd_s3(w, k_pad, ldw):
    """OIHW -> split-bf16 MFMA B fragments [taps][k_pad/16][ldw/32][3][64 lanes][8] (pmf_conv_desc_t.w_s3): element
    (tap, k, n) sits in fragment (tap, k/16, n/32) at lane n%32 + 32*((k%16)/8), slot k%8."""
    co, ci, kh, kw = w.shape
    p = torch.zeros(kh * kw, k_pad, ldw)
    p[:, :ci, :co] = w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co)
    planes = torch.stack([h.view(torch.int16) for h in split_bf16(p)], 0)          # [3][t][k][n]
    f = planes.view(3, kh * kw, k_pad // 16, 2, 8, ldw // 32, 32)                   # p t ks kh ke ct nl
    f = f.permute(1, 2, 5, 0, 3, 6, 4).contiguous()                                # t ks ct p kh nl ke
    return f.cuda()


def pack_fwd_s3_stem(w, ldw):
    """stem class (one operand of 8 padded channels, many taps): pmf_pack_job_t format 2 -- ONE virtual tap whose K index is
    tap * 8 + channel, K padded to a multiple of 16"""
    co, ci, kh, kw = w.shape
    kp = (kh * kw * 8 + 15) // 16 * 16
    p = torch.zeros(1, kp, ldw)
    q = torch.zeros(kh * kw, 8, co)
    q[:, :ci, :] = w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co)
    p[0, :kh * kw * 8, :co] = q.reshape(kh * kw * 8, co)
    planes = torch.stack([h.view(torch.int16) for h in split_bf16(p)], 0)
    f = planes.view(3, 1, kp // 16, 2, 8, ldw // 32, 32).permute(1, 2, 5, 0, 3, 6, 4).contiguous()
    return f.cuda()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def conv_desc(srcs, wpk, ldw, bias, out, N, OH, OW, Cout, taps, stride=1, act=0, gather=0, stats=None):
    """srcs: list of dict(x=tensor NHWC, C=int, scale=, shift=, cmul=, relu=bool, bcast=bool)."""
    d = L.ConvDesc()
    d.N, d.OH, d.OW, d.Cout, d.nsrc = N, OH, OW, Cout, len(srcs)
    for i, s in enumerate(srcs):
        x = s["x"]
        d.src[i].x = x.data_ptr()
        d.src[i].C = s["C"]
        d.src[i].ldc = x.shape[-1]
        d.src[i].H, d.src[i].W = x.shape[1], x.shape[2]
        d.src[i].scale = s["scale"].data_ptr() if s.get("scale") is not None else None
        d.src[i].shift = s["shift"].data_ptr() if s.get("shift") is not None else None
        if s.get("cmul") is not None:
            d.src[i].cmul = s["cmul"].data_ptr()
            d.src[i].cmul_ld = s["cmul"].shape[1]
        d.src[i].flags = (L.SRC_RELU if s.get("relu") else 0) | (L.SRC_BCAST if s.get("bcast") else 0)
    d.ntaps = len(taps)
    for i, (dy, dx) in enumerate(taps):
        d.tdy[i], d.tdx[i] = dy, dx
    d.in_stride, d.gather = stride, gather
    d.w, d.ldw = wpk.data_ptr(), ldw
    d.bias = bias.data_ptr() if bias is not None else None
    d.act = act
    d.out, d.out_ldc, d.out_H, d.out_W = out.data_ptr(), out.shape[-1], OH, OW
    d.out_sy = d.out_sx = 1
    d.stats = stats.data_ptr() if stats is not None else None
    return d


def taps_of(kh, kw, dil, pad):
    return [(ky * dil - pad, kx * dil - pad) for ky in range(kh) for kx in range(kw)]


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1.0)).max())


def scale_err(a, b):
    """max |a-b| relative to the tensor's own scale (intermediate activations span orders of magnitude;
    the elementwise criterion is reserved for logits / probabilities)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1.0))


def capture_oracle(oracle):
    """register forward hooks on the oracle PMFNet; returns (dict filled during forward, handles).
    Keys are the plan's debug names (pmf_amd.plan.Plan.tensors / .views)."""
    cap, hs = {}, []

    def hook(name, fn=None):
        def h(mod, inp, out):
            cap[name] = (fn(out) if fn else out)
        return h
    enc, ls, dec = oracle.camera_stream_encoder, oracle.lidar_stream, oracle.camera_stream_decoder
    hs.append(enc.bn1.register_forward_hook(hook("enc.stem", lambda o: o.clamp_min(0))))
    for li in range(4):
        for bi, blk in enumerate(getattr(enc, "layer%d" % (li + 1))):
            hs.append(blk.register_forward_hook(hook("enc.layer%d.%d.out" % (li + 1, bi))))
    for nm in ("downCntx", "downCntx2", "downCntx3"):
        hs.append(getattr(ls, nm).register_forward_hook(hook(nm + ".out")))
    for i in range(1, 5):
        def h(mod, inp, out, i=i):
            cap["resBlock%d.pool" % i], cap["resBlock%d.resA" % i] = out
        hs.append(getattr(ls, "resBlock%d" % i).register_forward_hook(h))
        if hasattr(ls, "fusionblock_%d" % i):
            hs.append(getattr(ls, "fusionblock_%d" % i).register_forward_hook(hook("fusion%d.out" % i)))
            fb = getattr(ls, "fusionblock_%d" % i)
            hs.append(fb.fuse_conv.register_forward_hook(hook("fusion%d.f" % i)))
    hs.append(ls.resBlock5.register_forward_hook(hook("resBlock5.resA")))
    if hasattr(ls, "aspp"):
        hs.append(ls.aspp.register_forward_hook(hook("aspp.out")))
    for i in range(1, 5):
        hs.append(getattr(ls, "upBlock%d" % i).register_forward_hook(hook("upBlock%d.e" % i)))
    hs.append(ls.logits.register_forward_hook(hook("logits")))
    for i, nm in ((4, "up_4a"), (3, "up_3a"), (2, "up_2a"), (1, "up_1a")):
        hs.append(getattr(dec, nm).register_forward_hook(hook("dec.up%d.up" % i)))
    hs.append(dec.conv.register_forward_hook(hook("dec.logits")))
    return cap, hs


def compare_plan_to_oracle(plan, cap, skip=()):
    """[(name, rel_err)] in the oracle's execution order for every captured tensor the plan also holds."""
    rows = []
    for name, ref in cap.items():
        if name in skip:
            continue
        if name in plan.views:
            got = plan.read_view(plan.views[name]).cpu()
        elif name in plan.tensors:
            got = plan.read(plan.tensors[name]).cpu()
        else:
            continue
        ref = ref.detach()
        if got.shape != ref.shape:
            rows.append((name, float("inf")))
            continue
        rows.append((name, (rel_err if name in ("logits", "dec.logits") else scale_err)(got.numpy(), ref.numpy())))
    return rows


def masked_grad_rows(hip, plan, ref, drop_masks, pcd, rgb, upstream):
    """[(parameter, hip error, fp32-oracle error)]: relative L2 distance of every parameter gradient from the FLOAT64 oracle,
    with the HIP path's piecewise-linear decisions (Plan.act_decisions: LeakyReLU / ReLU signs, the stem pool's argmax) replayed
    by both oracle passes (oracle/act_masks.py) and both driven by ``upstream`` = the HIP path's (d loss / d lidar
    probabilities, d loss / d camera probabilities).  All three backward passes differentiate one piecewise-linear function:
    an activation on its kink cannot move a gradient by a whole term, what is left is rounding -- or a kernel defect.
    ``ref``: a PRISTINE oracle module (no forward pass run on it yet) holding the same parameters."""
    import copy
    from oracle import pmf_torch as O
    from oracle.act_masks import ActSites
    dec = plan.act_decisions(hip)
    grads = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m = copy.deepcopy(ref).to(dt).train()
        O.set_dropout_masks(m, {k: v.to(dt) for k, v in drop_masks.items()})
        with ActSites(m, inject=dec) as inj:
            a, b = m(pcd.to(dt), rgb.to(dt))
        assert not inj.unused, inj.unused
        torch.autograd.backward([a, b], [upstream[0].cpu().to(dt), upstream[1].cpu().to(dt)])
        grads[tag] = {k: p.grad.detach().double() for k, p in m.named_parameters()}
    rows = GradRows()
    for k, p in hip.named_parameters():
        g64 = grads["f64"][k]
        wk = k.rsplit(".", 1)[0] + ".weight"
        floor = 1e-6 * grads["f64"][wk].norm().item() if wk in grads["f64"] else 0.0
        den = max(g64.norm().item(), floor, 1e-30)
        rows.append((k, (p.grad.cpu().double() - g64).norm().item() / den, (grads["f32"][k] - g64).norm().item() / den))
        if p.dim() == 1:
            rows.one_d.add(k)
    return rows


class GradRows(list):
    """[(parameter, hip error, fp32-oracle error)] + the names of the 1-D parameters (conv bias, BatchNorm gamma / beta)"""

    def __init__(self, *a):
        super().__init__(*a)
        self.one_d = set()


def assert_masked_bar(rows, what=""):
    """EVERY parameter, no outlier allowance, relative L2 distance from the float64 oracle with the decisions injected:
      * weight tensors (>= 2-D):   hip <= max(4 x the fp32 CPU oracle's distance for THAT parameter, 2e-4) -- 4 x is the per-launch pin
        of the split products against the fp32-MFMA path (tests/test_gpu_parity.py); the headline shapes sit within 3.1 x
      * 1-D parameters (conv bias, BatchNorm gamma / beta): hip <= max(8 x ..., 5e-4).  Their gradients are column sums over all
        pixels, mostly under cancellation; the residual of the six-product bf16 split (dropped mid x lo / lo x lo terms, 6x longer
        fp32 accumulation chains in the MFMA) adds COHERENTLY in such sums where random rounding does not.  Measured
        (profiles/r06_masked_precision_class.txt, PMF-ResNet34 at 2 x 480 x 640): with PMF_CONV_F32=1 every parameter sits within
        2.7x of the fp32 CPU oracle; with the split products four 1-D parameters of resBlock4 / 5 / fusionblock_4 sit at 4.2-7.5x
        (2.4-4.1e-4), every weight tensor within 3.1x; PMF-ResNet50 at 2 x 480 x 640 after 47 training iterations: the whole
        camera encoder at 3.3-3.5x (1.4-1.6e-3 where the fp32 CPU oracle sits at 4.2-4.7e-4), 1.1x with PMF_CONV_F32=1.  That is
        the precision class DESIGN.md section 6 states, not a defect."""
    one_d = getattr(rows, "one_d", set())

    def bar(r):
        return max(8 * r[2], 5e-4) if r[0] in one_d else max(4 * r[2], 2e-4)
    bad = [r for r in rows if not r[1] <= bar(r)]
    assert not bad, "%s gradient error vs float64, decisions injected (hip, cpu-fp32):\n" % what + "\n".join(
        "%-55s %.3e %.3e" % r for r in sorted(bad, key=lambda t: -t[1])[:30])
