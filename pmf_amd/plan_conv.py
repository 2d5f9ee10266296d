"""Convolution layers of a plan (mixin of pmf_amd.plan.Plan): Conv2d (+bias) -> activation -> BatchNorm forward, and on the
tape the BatchNorm backward, the input gradients (one launch per operand / parity class, or ONE multi-destination launch for
a concatenated input) and the weight gradient (partial slabs + batched reduction on the weight-gradient lanes)."""
import ctypes as C
import os

import torch

from . import _lib as L
from .plan_graph import COL_ROWS, SPLITK_BYTES, _ru, T, V


class PlanConvMixin(object):
    def taps(self, kh, kw, dil, pad):
        out = []
        for ky in range(kh):
            for kx in range(kw):
                out.append((ky * dil - pad, kx * dil - pad, ky * kw + kx))
        return out

    def add_pack(self, weight, taps_widx, transpose, K_pad, ldw, fmt=0, cin=None):
        """register a pack job; returns the Buf of the packed slab [ntaps][K_pad][ldw] (fmt 0, fp32) or of the split-bf16
        fragments [ntaps][K_pad/16][ldw/32][3][64][8] (fmt 1, 6 bytes per weight; pmf_conv_desc_t.w_s3)."""
        # (fmt 2, the stem class: ONE virtual tap, K_pad = taps * 8 rounded up to 16)
        buf = self.persist.alloc((6 if fmt else 4) * (1 if fmt == 2 else len(taps_widx)) * K_pad * ldw)
        # forward packs: index of the conv op about to be emitted (the first reader); input-gradient packs: none (they are
        # read by the backward graph only)
        owner = len(self.fwd) if (not transpose and self.fwd is not None) else None
        # cin = (first input channel, count): pack that channel range only (its own column origin)
        self.pack_jobs.append((weight, buf, list(taps_widx), transpose, K_pad, ldw, self.lane, fmt, owner, cin))
        return buf

    def s3_ok(self, shape_fill):
        """True when this conv launch may run on the bf16 matrix pipe with split operands (conv_fwd.hip PIPE 5)."""
        if not self.s3:
            return False
        probe = L.ConvDesc()
        shape_fill(probe)
        if probe.ntaps == 1:
            # 1x1 layers: only the direct variant (activations straight from global memory, conv_fwd.hip PIPE 11), and only
            # where the map is large enough to fill the chip without a K split
            return (self.s3_direct_min_pix > 0 and probe.N * probe.OH * probe.OW >= self.s3_direct_min_pix
                    and L.lib().pmf_conv_s3_eligible(C.byref(probe)) == 2)
        if probe.ntaps < self.s3_min_taps:
            return False
        return int(L.lib().pmf_conv_s3_eligible(C.byref(probe)))     # (3: the stem class, weights in pack format 2)

    def conv(self, srcs, conv, act=L.ACT_NONE, bn=None, order="act_bn", relu_view=False, name="", pmask=None,
             extra_bias=None):
        """Conv2d (+bias) -> act -> [BatchNorm]  (order 'act_bn', SalsaNext style) or
        Conv2d -> BatchNorm -> [ReLU on the view]  (order 'bn_act', ResNet / attention style).
        Returns a V.  Registers the backward (BN backward, input gradients, weight gradient)."""
        kh, kw = conv.kernel_size
        dil, pad, stride = conv.dilation[0], conv.padding[0], conv.stride[0]
        Cout = conv.out_channels
        t0 = srcs[0].t
        N = t0.N
        inH = max(s.t.H for s in srcs)
        inW = max(s.t.W for s in srcs)
        OH = (inH + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        OW = (inW + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        out = T(self, N, OH, OW, Cout, name)
        # taps that can only ever read zero padding (|offset| beyond the map: the dilation-12/18 ASPP branches on a
        # 4-row map keep 3 of 9 taps) are dropped from forward, input gradient and weight gradient alike; their weight
        # gradient is exactly zero and stays at the zero the backward prologue writes
        all_taps = self.taps(kh, kw, dil, pad)
        taps = [(dy, dx, wi) for (dy, dx, wi) in all_taps
                if dy < inH and dy + (OH - 1) * stride >= 0 and dx < inW and dx + (OW - 1) * stride >= 0] or all_taps[:1]
        Ktot = sum(_ru(s.t.C, 8) for s in srcs)
        ldw = _ru(Cout, 64)
        # very large dilations: per-tap staging (the halo tile would not fit LDS)
        # (rows and columns separately: the dilation-12 / 18 ASPP branches keep one ROW of three taps on the 4-row map -- a
        # 4 x 56-pixel halo tile, not a 32 x 56 one)
        span_y = max(t[0] for t in taps) - min(t[0] for t in taps)
        span_x = max(t[1] for t in taps) - min(t[1] for t in taps)
        gather = 1 if (8 * stride + span_y) * (32 * stride + span_x) * 80 > 110 * 1024 else 0
        # a BatchNorm module in eval mode inside a training plan (frozen statistics, torch semantics): running statistics in
        # the forward pass, no statistics update, backward through the fixed affine map (dgamma / dbeta still flow)
        train_bn = bn is not None and self.training and bn.training
        has_bias = conv.bias is not None
        k_act = act if order == "act_bn" else L.ACT_NONE
        # EPMF SparseVariantConv: (conv + conv.bias + extra bias) * dilated mask; the mask multiplies after the
        # activation (LeakyReLU(0) = 0 and the mask is 0/1, so act(z*m) == act(z)*m) and before the BN statistics
        bsum = None
        if extra_bias is not None:
            if not has_bias:
                raise NotImplementedError("extra_bias needs a conv bias to add to")
            bsum = self.persist.alloc(4 * Cout)

            def fv(op):
                a = op.u.sm
                a.p[0], a.p[1], a.p[2] = conv.bias.data_ptr(), extra_bias.data_ptr(), bsum.ptr
                a.i[0] = Cout
            self.emit(self.fwd, L.OP_VEC_ADD, fv)

        def shape_fill(d):
            d.N, d.OH, d.OW, d.Cout, d.nsrc = N, OH, OW, Cout, len(srcs)
            for i, s in enumerate(srcs):
                d.src[i].C = _ru(s.t.C, 8)
                d.src[i].ldc, d.src[i].H, d.src[i].W = s.t.ldc, s.t.H, s.t.W
                # flags steer the kernel variant (hence the split-K decision the statistics-row probe must match)
                d.src[i].flags = (L.SRC_RELU if s.relu else 0) | (L.SRC_BCAST if s.bcast else 0)
            d.ntaps = len(taps)
            for i, (dy, dx, _) in enumerate(taps):
                d.tdy[i], d.tdx[i] = dy, dx
            d.in_stride, d.gather = stride, gather
            d.out_sy = d.out_sx = 1
            d.splitk_ws, d.splitk_ws_bytes = 1, SPLITK_BYTES      # non-NULL: same split decision as the real launch
            d.splitk_tickets = 1 if self.sk_fused else None          # (... and the same partial-statistics row count)
        fwd_s3 = self.s3_ok(shape_fill)
        if fwd_s3 == 3:
            # the 7x7 RGB stem: 8 padded channels x 49 taps, two taps per 16-deep MFMA step (conv_fwd.hip PIPE 14)
            wbuf = self.add_pack(conv.weight, [t[2] for t in taps], 0, _ru(len(taps) * 8, 16), ldw, 2)
        else:
            wbuf = self.add_pack(conv.weight, [t[2] for t in taps], 0, Ktot, ldw, int(bool(fwd_s3)))
        stat_rows = 0
        if train_bn:
            probe = L.ConvDesc()
            shape_fill(probe)
            probe.w_s3 = 1 if fwd_s3 else None
            stat_rows = L.lib().pmf_conv_fwd_stat_rows(C.byref(probe))
            # the autotuner (Plan.autotune) may pick another tile configuration: size the rows for any of them
            max_rows = max(stat_rows, L.lib().pmf_conv_fwd_stat_rows_max(C.byref(probe)))
        stats = self.act.alloc(16 * Cout * max_rows) if train_bn else None   # float64 [rows][2][Cout] partials
        bn_train_flag = int(train_bn)

        lane = self.lane

        def f(op, srcs=srcs):
            d = op.u.conv
            shape_fill(d)
            for i, s in enumerate(srcs):
                self.src_struct(s, d.src[i])
            d.ldw = ldw
            if fwd_s3:
                d.w_s3 = wbuf.ptr
            else:
                d.w = wbuf.ptr
            d.bias = (bsum.ptr if bsum is not None else conv.bias.data_ptr()) if has_bias else None
            d.act = k_act
            d.out, d.out_ldc, d.out_H, d.out_W = out.buf.ptr, out.ldc, OH, OW
            d.out_sy = d.out_sx = 1
            d.stats = stats.ptr if stats is not None else None
            d.ep_pmask = pmask.buf.ptr if pmask is not None else None
            d.splitk_ws, d.splitk_ws_bytes = self.sk_bufs[lane].ptr, self.sk_bufs[lane].nbytes
            d.splitk_tickets = self.sk_tickets[lane].ptr if self.sk_fused else None
        self.emit(self.fwd, L.OP_CONV, f)
        conv_fwd_index = len(self.fwd) - 1
        conv_flops = 2.0 * N * OH * OW * Cout * conv.in_channels * len(taps)   # algorithmic (SURVEY.md 8d rule)
        self.meta_fwd[len(self.fwd) - 1] = dict(family="conv_fwd", flops=conv_flops, name=name, shape="%dx%dx%d %d->%d t%d s%d" % (
            N, OH, OW, conv.in_channels, Cout, len(taps), stride))

        view = V(out)
        info = None
        if bn is not None:
            Cb = Cout
            scale, shift = self.persist.alloc(4 * Cb), self.persist.alloc(4 * Cb)
            smean, sinv = self.persist.alloc(4 * Cb), self.persist.alloc(4 * Cb)
            count = float(N * OH * OW)
            if train_bn:
                def fb(op):
                    a = op.u.sm
                    for i, p in enumerate((stats.ptr, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                           bn.running_mean.data_ptr(), bn.running_var.data_ptr(), scale.ptr,
                                           shift.ptr, smean.ptr, sinv.ptr)):
                        a.p[i] = p
                    a.f[0], a.f[1], a.f[2] = count, bn.momentum, bn.eps
                    a.i[0], a.i[1] = Cb, stat_rows
                self.emit(self.fwd, L.OP_BN_FINALIZE, fb)
                self._conv_fin[conv_fwd_index] = len(self.fwd) - 1      # its row count follows the conv's tile config
            else:
                def fb(op):
                    a = op.u.sm
                    for i, p in enumerate((bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                           bn.running_var.data_ptr(), scale.ptr, shift.ptr, smean.ptr, sinv.ptr)):
                        a.p[i] = p
                    a.f[0] = bn.eps
                    a.i[0] = Cb
                self.emit(self.fwd, L.OP_BN_EVAL, fb)
            info = dict(mean=smean, invstd=sinv, module=bn)
            view = V(out, scale, shift, relu=relu_view, bn=info)
            if train_bn:
                self.bn_modules.append(bn)
        if name:
            self.views[name] = view
            self.act_sites.append((conv, name, k_act, bool(relu_view)))
        if not self.training:
            return view

        # ------------------------------------------------------------------ backward
        def backward():
            # partial column sums of dz for the conv-bias gradient: one buffer PER LAYER (the weight-gradient op that
            # folds them runs on the side stream while the main stream already works on the next layer)
            dbr_ld = _ru(Cout, 4)
            dbr = self.act.alloc(COL_ROWS * dbr_ld * 4) if (has_bias and pmask is None) else None
            if bn is not None:
                if getattr(view, "_side", None):
                    if view.gy is None:
                        view.gy = T(self, N, OH, OW, Cout, name + ".gy", ldc=out.ldc)
                    view.gy_written = self._fold_side(view, view.gy, view.gy_written)
                    view._side = None
                    view._gy_last = None
                if view.gy is None:
                    raise RuntimeError("plan: no gradient reached BN output of %s" % name)
                if out.g is None:
                    out.g = T(self, N, OH, OW, Cout, name + ".dz", ldc=out.ldc)
                out.g_written = True
                coef = self.act.alloc(12 * Cout)                  # [3][Cout] per-channel backward coefficients
                dgam, dbet = self.pgrad(bn.weight), self.pgrad(bn.bias)
                gyt, dz = view.gy, out.g
                self._touch(gyt)
                self._touch(dz)
                self.colrows_max = max(self.colrows_max, _ru(Cout, 4))

                lw = getattr(view, "_gy_last", None)
                # small maps (<= 2048 pixels: the 4x128 stage and below, a fifth of the BatchNorm layers): column sums,
                # fold and apply in ONE launch (bn.hip bn_bwd_small_k) instead of three latency-bound ones; the
                # input-gradient epilogue then carries no partial sums either (its hook stays empty)
                small = (os.environ.get("PMF_BN_SMALL", "1") != "0" and Cout % 4 == 0
                         and bool(L.lib().pmf_bn_bwd_small_ok(out.npix, Cout)))
                if small:
                    one_row = has_bias and pmask is None

                    def rs(op):
                        a = op.u.sm
                        ps = (gyt.buf.ptr, out.buf.ptr, info["mean"].ptr, bn.weight.data_ptr(), info["invstd"].ptr,
                              dz.buf.ptr, dbr.ptr if (dbr is not None and one_row) else None, self.pgrad_buf.at(dgam),
                              self.pgrad_buf.at(dbet))
                        for i, p in enumerate(ps):
                            a.p[i] = p
                        a.i[0], a.i[1], a.i[2], a.i[3], a.i[4], a.i[5] = gyt.ldc, out.ldc, Cout, bn_train_flag, k_act, dz.ldc
                        a.l[0] = out.npix
                    self.emit(self.bwd, L.OP_BN_BWD_SMALL, rs)
                    self.note_bytes(self.bwd, "bn_bwd_small", 12.0 * out.npix * Cout)
                    self.grad_done[id(bn.weight)] = self.grad_done[id(bn.bias)] = len(self.bwd) - 1
                    dbias_rows = 1 if one_row else 0
                elif lw is not None and lw["lane"] == lane:
                    # the last writer of gy was an input-gradient launch on this lane: it carried the reduction
                    # (sum gy, sum gy*(a - mean) per tile row) in its epilogue; only the fold is left
                    probe = L.ConvDesc()
                    lw["shape"](probe)
                    nrows = L.lib().pmf_conv_fwd_stat_rows(C.byref(probe))
                    rows = self.act.alloc(16 * Cout * max(nrows, L.lib().pmf_conv_fwd_stat_rows_max(C.byref(probe))))
                    lw["hook"].update(rows=rows, mean=info["mean"])

                    def r1(op):
                        a = op.u.sm
                        ps = (rows.ptr, bn.weight.data_ptr(), info["invstd"].ptr, coef.ptr, self.pgrad_buf.at(dgam),
                              self.pgrad_buf.at(dbet))
                        for i, p in enumerate(ps):
                            a.p[i] = p
                        a.i[0], a.i[1], a.i[2] = Cout, nrows, bn_train_flag
                        a.l[0] = out.npix
                    self.emit(self.bwd, L.OP_BN_BWD_FOLD, r1)
                    self._conv_fold.setdefault(lw["index"], []).append(len(self.bwd) - 1)     # (a multi-destination
                    #                                                      launch feeds one fold per destination)
                else:
                    def r1(op):
                        a = op.u.sm
                        ps = (gyt.buf.ptr, out.buf.ptr, info["mean"].ptr, bn.weight.data_ptr(), info["invstd"].ptr,
                              self.bnpart_bufs[lane].ptr, coef.ptr, self.pgrad_buf.at(dgam), self.pgrad_buf.at(dbet))
                        for i, p in enumerate(ps):
                            a.p[i] = p
                        a.i[0], a.i[1], a.i[2], a.i[3] = gyt.ldc, out.ldc, Cout, bn_train_flag
                        a.l[0] = out.npix
                    self.emit(self.bwd, L.OP_BN_BWD_REDUCE, r1)
                    self.note_bytes(self.bwd, "bn_bwd_reduce", 8.0 * out.npix * Cout)
                if not small:
                    self.grad_done[id(bn.weight)] = self.grad_done[id(bn.bias)] = len(self.bwd) - 1

                def r2(op):
                    a = op.u.sm
                    ps = (gyt.buf.ptr, out.buf.ptr, coef.ptr, info["mean"].ptr, dz.buf.ptr,
                          dbr.ptr if dbr is not None else None)
                    for i, p in enumerate(ps):
                        a.p[i] = p
                    a.i[0], a.i[1], a.i[2], a.i[3], a.i[4], a.i[5] = gyt.ldc, out.ldc, Cout, k_act, dz.ldc, dbr_ld
                    a.l[0] = out.npix
                if not small:
                    self.emit(self.bwd, L.OP_BN_BWD_APPLY, r2)
                    self.note_bytes(self.bwd, "bn_bwd_apply", 12.0 * out.npix * Cout)
                    dbias_rows = L.lib().pmf_col_rows(out.npix, Cout) if (has_bias and pmask is None) else 0
            else:
                dz = self.tgrad(out)
                dbias_rows = 0
                if k_act != L.ACT_NONE or has_bias:
                    def r3(op):
                        a = op.u.sm
                        a.p[0], a.p[1] = dz.buf.ptr, out.buf.ptr
                        a.p[2] = dbr.ptr if dbr is not None else None
                        a.i[0], a.i[1], a.i[2], a.i[3], a.i[4] = dz.ldc, out.ldc, k_act, _ru(Cout, 4), dbr_ld
                        a.l[0] = out.npix
                    self.emit(self.bwd, L.OP_ACT_BWD, r3)
                    self.note_bytes(self.bwd, "act_bwd", 12.0 * out.npix * Cout)
                    dbias_rows = L.lib().pmf_col_rows(out.npix, _ru(Cout, 4)) if (has_bias and pmask is None) else 0
            dz = out.g
            if pmask is not None:
                # d/dz of act(z) * m: the activation derivative above was taken from a = act(z)*m (slope of the
                # masked-out zeros is irrelevant) -- multiply by the mask, then the bias gradients are plain column
                # sums of the masked dz (both bias vectors of SparseVariantConv receive the same gradient)
                def rm(op, dz=dz):
                    a = op.u.sm
                    a.p[0], a.p[1], a.p[2] = dz.buf.ptr, pmask.buf.ptr, dz.buf.ptr
                    a.i[0], a.i[1], a.i[2], a.i[3] = dz.ldc, _ru(Cout, 4), dz.ldc, 0
                    a.l[0] = dz.npix
                self.emit(self.bwd, L.OP_PMASK_MUL_BWD, rm)
                if has_bias:
                    boff = self.pgrad(conv.bias)

                    crows = self.act.alloc(COL_ROWS * _ru(Cout, 4) * 4)      # partial rows of the deterministic column sum

                    def rb(op, dz=dz, boff=boff, crows=crows):
                        a = op.u.sm
                        a.p[0], a.p[1], a.p[2] = dz.buf.ptr, self.pgrad_buf.at(boff), crows.ptr
                        a.i[0], a.i[1], a.i[2] = dz.ldc, Cout, 1
                        a.l[0] = dz.npix
                    self.emit(self.bwd, L.OP_COLSUM, rb)
                    self.grad_done[id(conv.bias)] = len(self.bwd) - 1
                    if extra_bias is not None:
                        eoff = self.pgrad(extra_bias)

                        def re(op, boff=boff, eoff=eoff):
                            a = op.u.sm
                            a.p[0], a.p[1], a.p[2] = self.pgrad_buf.at(boff), None, self.pgrad_buf.at(eoff)
                            a.i[0] = Cout
                        self.emit(self.bwd, L.OP_VEC_ADD, re)
                        self.grad_done[id(extra_bias)] = len(self.bwd) - 1
            self._dgrad(srcs, conv, dz, taps, stride, gather, name)
            self._wgrad(srcs, conv, dz, taps, stride, gather, name, dbias_rows, dbr, dbr_ld)
        self.on_backward(backward)
        return view

    def _dgrad(self, srcs, conv, dz, taps, stride, gather, name):
        """input gradients: the same implicit-GEMM kernel run over dz with transposed weights.
        stride 1:  dX[y] = sum_t dz[y - dy_t] W_t^T                      (one launch per operand)
        stride 2:  y = 2m + py:  dX[y] = sum_{t: (py-dy_t) even} dz[m + (py-dy_t)/2] W_t^T   (one launch per parity)"""
        Cout = conv.out_channels
        Kd = _ru(Cout, 8)                       # K of the input-gradient GEMM (padded channels of dz are zero)
        Cin_tot = sum(_ru(s.t.C, 8) for s in srcs)
        if not any(s.t.needs_grad for s in srcs):
            return
        ldwT = _ru(Cin_tot, 64) + 64
        lane = self.lane
        if stride == 1:
            classes = [(0, 0, [(-dy, -dx, wi) for (dy, dx, wi) in taps])]
        else:
            classes = []
            for py in range(2):
                for px in range(2):
                    sub = [((py - dy) // 2, (px - dx) // 2, wi) for (dy, dx, wi) in taps
                           if (py - dy) % 2 == 0 and (px - dx) % 2 == 0]
                    if sub:
                        classes.append((py, px, sub))
        # split-bf16 launches address the transposed weights by 32-column fragments: every operand that receives a
        # gradient must start on one
        offs, co_ = [], 0
        for s in srcs:
            offs.append(co_)
            co_ += _ru(s.t.C, 8)
        aligned = all(o % 32 == 0 for o, s in zip(offs, srcs) if s.t.needs_grad)
        ref = next(s for s in srcs if s.t.needs_grad)

        def class_probe(sub, py, px):
            def fill(d):
                H, W = ref.t.H, ref.t.W
                d.N = dz.N
                d.OH, d.OW = (H, W) if stride == 1 else ((H - py + 1) // 2, (W - px + 1) // 2)
                d.Cout, d.nsrc = ref.t.C, 1
                sv = d.src[0]
                sv.C, sv.ldc, sv.H, sv.W = Kd, dz.ldc, dz.H, dz.W
                d.ntaps = len(sub)
                for i, (dy, dx, _) in enumerate(sub):
                    d.tdy[i], d.tdx[i] = dy, dx
                d.in_stride, d.gather = 1, gather
            return fill
        if Kd % 16 and self.s3:
            # 20 output channels (the logits heads): K = 24 keeps the launch off the split kernels.  Rounded up to 32 the last
            # eight "channels" of a pixel are the first eight of the next pixel (finite values; past the end of the tensor the
            # buffer range check returns 0) against weight rows that are zero -- they add exactly 0
            Kd8, Kd = Kd, _ru(Cout, 16)
            if not all(self.s3_ok(class_probe(sub, py, px)) for (py, px, sub) in classes):
                Kd = Kd8
        dg_s3 = [Kd % 16 == 0 and self.s3_ok(class_probe(sub, py, px)) for (py, px, sub) in classes]
        # operands that do not start on a 32-column fragment (16 + 64 channels): every operand gets its own transposed
        # pack of its channel range, starting at column 0
        own_packs = not aligned and all(dg_s3) and all(_ru(s.t.C, 8) == s.t.C for s in srcs)
        if not aligned and not own_packs:
            dg_s3 = [False] * len(classes)
        packs = None if own_packs else [self.add_pack(conv.weight, [t[2] for t in sub], 1, Kd, ldwT, int(k3))
                                        for (_, _, sub), k3 in zip(classes, dg_s3)]
        # ---- one launch for all operands of a concatenated input (stride 1, split-bf16 path, every operand a whole number
        # of 32-channel fragments): dz is read ONCE and the output-channel ranges go to the operands' gradient tensors, each
        # with its own epilogue (accumulate / Dropout2d multiplier / ReLU mask / BatchNorm-backward partial sums)
        merged = (stride == 1 and len(srcs) >= 2 and not own_packs and aligned and all(dg_s3) and len(srcs) <= L.MAX_SRC
                  and all(s.t.needs_grad and not s.bcast and s.t.C % 32 == 0 for s in srcs)
                  and dz.N * dz.H * dz.W >= int(os.environ.get("PMF_DGRAD_MERGE_MINPIX", "1024")) and os.environ.get("PMF_DGRAD_MERGE", "1") != "0")
        if merged:
            # torch.cat of one tensor (or one root) twice: two destinations would be ONE gradient buffer, written with
            # accumulate 0 and 1 by workgroups of one launch in unspecified order -- such a concat keeps the per-operand
            # launches (stream order)
            holders = [(s.root() if s.root().bn is not None else s.root().t) for s in srcs]
            merged = len({id(h_) for h_ in holders}) == len(holders)
        if merged:
            # the library's own eligibility rule, on the shape-only descriptor (a mismatch would otherwise only surface as
            # PMF_E_ARG when the plan runs)
            probe = L.ConvDesc()
            probe.N, probe.OH, probe.OW = dz.N, ref.t.H, ref.t.W
            probe.Cout, probe.nsrc = sum(s.t.C for s in srcs), 1
            probe.src[0].C, probe.src[0].ldc, probe.src[0].H, probe.src[0].W = Kd, dz.ldc, dz.H, dz.W
            probe.ntaps = len(classes[0][2])
            for i, (dy, dx, _) in enumerate(classes[0][2]):
                probe.tdy[i], probe.tdx[i] = dy, dx
            probe.in_stride, probe.gather, probe.out_sy, probe.out_sx, probe.w_s3 = 1, gather, 1, 1, 1
            probe.ndst = len(srcs)
            for k_, s in enumerate(srcs):
                probe.dst[k_].C = s.t.C
            merged = bool(L.lib().pmf_conv_multi_ok(C.byref(probe)))
        if merged:
            (_, _, sub), wT = classes[0], packs[0]
            parts = []
            for s in srcs:
                r = s.root()
                tgt, acc = self.grad_of(s)
                parts.append(dict(s=s, r=r, tgt=tgt, acc=acc, relu_x=(r.t if r.relu else None), hook={}))
            Ctot = sum(pt["s"].t.C for pt in parts)

            def shape_only(d, sub=sub, parts=parts, Ctot=Ctot):
                d.N, d.OH, d.OW = dz.N, ref.t.H, ref.t.W
                d.Cout, d.nsrc = Ctot, 1
                sv = d.src[0]
                sv.C, sv.ldc, sv.H, sv.W = Kd, dz.ldc, dz.H, dz.W
                d.ntaps = len(sub)
                for i, (dy, dx, _) in enumerate(sub):
                    d.tdy[i], d.tdx[i] = dy, dx
                d.in_stride, d.gather = 1, gather
                d.out_sy = d.out_sx = 1
                d.w_s3 = 1
                d.ndst = len(parts)
                for k, pt in enumerate(parts):
                    d.dst[k].C = pt["s"].t.C

            def f(op, parts=parts, wT=wT, shape_only=shape_only, ldwT=ldwT):
                d = op.u.conv
                shape_only(d)
                d.src[0].x = dz.buf.ptr
                d.ldw = ldwT
                d.w_s3 = wT.ptr
                d.act = L.ACT_NONE
                d.out_H, d.out_W = ref.t.H, ref.t.W
                d.out, d.out_ldc = parts[0]["tgt"].buf.ptr, parts[0]["tgt"].ldc
                for k, pt in enumerate(parts):
                    e, s, r, tgt = d.dst[k], pt["s"], pt["r"], pt["tgt"]
                    e.out, e.out_ldc, e.accumulate = tgt.buf.ptr, tgt.ldc, pt["acc"]
                    if s.cmul is not None:
                        e.ep_cmul, e.ep_cmul_ld = self.masks_ptr + 4 * s.cmul, s.cmul_ld
                    if pt["relu_x"] is not None:
                        e.ep_relu_x, e.ep_relu_ldc = pt["relu_x"].buf.ptr, pt["relu_x"].ldc
                        e.ep_relu_scale = r.scale.ptr if r.scale is not None else None
                        e.ep_relu_shift = r.shift.ptr if r.shift is not None else None
                    if pt["hook"]:
                        e.stats, e.ep_stat_mean = pt["hook"]["rows"].ptr, pt["hook"]["mean"].ptr
                        if pt["relu_x"] is None:
                            e.ep_relu_x, e.ep_relu_ldc, e.ep_flags = r.t.buf.ptr, r.t.ldc, L.EP_STAT_X_ONLY
            self.emit(self.bwd, L.OP_CONV, f)
            for pt in parts:
                r = pt["r"]
                if r.bn is not None and pt["tgt"] is r.gy and self.bn_bwd_fused:
                    r._gy_last = dict(hook=pt["hook"], index=len(self.bwd) - 1, shape=shape_only, lane=lane)
            self.meta_bwd[len(self.bwd) - 1] = dict(
                family="conv_dgrad", flops=2.0 * dz.N * ref.t.H * ref.t.W * Ctot * Cout * len(sub), name=name,
                shape="%dx%dx%d %d->%s t%d s1" % (dz.N, ref.t.H, ref.t.W, Cout, "+".join(str(pt["s"].t.C) for pt in parts),
                                                  len(sub)))
            return
        coloff = 0
        for s in srcs:
            Cs = _ru(s.t.C, 8)
            if s.t.needs_grad:
                if own_packs:
                    ldwT = _ru(Cs, 64) + 64
                    packs = [self.add_pack(conv.weight, [t[2] for t in sub], 1, Kd, ldwT, 1, cin=(coloff, Cs))
                             for (_, _, sub) in classes]
                r = s.root()
                tmp = None
                if s.bcast:
                    # 1x1 map broadcast over the image: full-size gradient into a temp, then per-sample column sums
                    tmp = T(self, dz.N, dz.H, dz.W, s.t.C, name + ".bcast_tmp")
                    if r.t.g is None:
                        r.t.g = T(self, r.t.N, 1, 1, r.t.C, r.t.name + ".g", arena=self.zero_bwd, ldc=r.t.ldc)
                    r.t.g_written = True
                    self._touch(r.t.g)
                    tgt, acc = tmp, 0
                else:
                    tgt, acc = self.grad_of(s)
                    if stride != 1:
                        # parity classes only touch their own pixels: zero first, then accumulate
                        if not acc:
                            self.fill(self.bwd, tgt.buf, tgt.npix * tgt.ldc)
                        acc = 1
                relu_x = r.t if r.relu else None
                for (py, px, sub), wT, k3 in zip(classes, packs, dg_s3):
                    def shape_only(d, s=s, sub=sub, tgt=tgt, py=py, px=px, k3=k3):
                        H, W = tgt.H, tgt.W
                        d.N = dz.N
                        d.OH, d.OW = (H, W) if stride == 1 else ((H - py + 1) // 2, (W - px + 1) // 2)
                        d.Cout, d.nsrc = s.t.C, 1
                        sv = d.src[0]
                        sv.C, sv.ldc, sv.H, sv.W = Kd, dz.ldc, dz.H, dz.W
                        d.ntaps = len(sub)
                        for i, (dy, dx, _) in enumerate(sub):
                            d.tdy[i], d.tdx[i] = dy, dx
                        d.in_stride, d.gather = 1, gather
                        d.out_sy = d.out_sx = stride
                        d.splitk_ws, d.splitk_ws_bytes = 1, SPLITK_BYTES
                        d.splitk_tickets = 1 if self.sk_fused else None
                        d.w_s3 = 1 if k3 else None
                    # filled in by the BatchNorm backward of the layer that produced this operand when THIS launch is
                    # the last writer of its output gradient: the launch then also writes the BN-backward partial sums
                    hook = {}

                    def f(op, s=s, r=r, sub=sub, wT=wT, tgt=tgt, acc=acc, coloff=(0 if own_packs else coloff), py=py, px=px,
                          relu_x=relu_x, shape_only=shape_only, hook=hook, k3=k3, ldwT=ldwT):
                        d = op.u.conv
                        shape_only(d)
                        H, W = tgt.H, tgt.W
                        d.src[0].x = dz.buf.ptr
                        d.ldw = ldwT
                        if k3:      # fragment column coloff / 32 (3 planes x 1 KiB each)
                            d.w_s3 = wT.ptr + (coloff // 32) * 3 * 1024
                        else:
                            d.w = wT.at(coloff)
                        d.act = L.ACT_NONE
                        d.out, d.out_ldc, d.out_H, d.out_W = tgt.buf.ptr, tgt.ldc, H, W
                        d.out_sy = d.out_sx = stride
                        d.out_oy, d.out_ox = py, px
                        d.accumulate = acc
                        d.splitk_ws, d.splitk_ws_bytes = self.sk_bufs[lane].ptr, self.sk_bufs[lane].nbytes
                        d.splitk_tickets = self.sk_tickets[lane].ptr if self.sk_fused else None
                        if s.cmul is not None:
                            d.ep_cmul, d.ep_cmul_ld = self.masks_ptr + 4 * s.cmul, s.cmul_ld
                        if relu_x is not None:
                            d.ep_relu_x, d.ep_relu_ldc = relu_x.buf.ptr, relu_x.ldc
                            d.ep_relu_scale = r.scale.ptr if r.scale is not None else None
                            d.ep_relu_shift = r.shift.ptr if r.shift is not None else None
                        if hook:
                            d.stats, d.ep_stat_mean = hook["rows"].ptr, hook["mean"].ptr
                            if relu_x is None:
                                d.ep_relu_x, d.ep_relu_ldc, d.ep_flags = r.t.buf.ptr, r.t.ldc, L.EP_STAT_X_ONLY
                    self.emit(self.bwd, L.OP_CONV, f)
                    if (stride == 1 and tmp is None and r.bn is not None and tgt is r.gy and self.bn_bwd_fused):
                        r._gy_last = dict(hook=hook, index=len(self.bwd) - 1, shape=shape_only, lane=lane)
                    mh = tgt.H if stride == 1 else (tgt.H - py + 1) // 2
                    mw = tgt.W if stride == 1 else (tgt.W - px + 1) // 2
                    self.meta_bwd[len(self.bwd) - 1] = dict(
                        family="conv_dgrad", flops=2.0 * dz.N * mh * mw * s.t.C * Cout * len(sub), name=name,
                        shape="%dx%dx%d %d->%d t%d s%d" % (dz.N, mh, mw, Cout, s.t.C, len(sub), stride))
                if tmp is not None:
                    crows = self.act.alloc(COL_ROWS * tmp.N * _ru(r.t.g.ldc, 4) * 4)

                    def fc(op, tmp=tmp, g=r.t.g, crows=crows):
                        a = op.u.sm
                        a.p[0], a.p[1], a.p[2] = tmp.buf.ptr, g.buf.ptr, crows.ptr
                        a.i[0], a.i[1], a.i[2] = tmp.ldc, g.ldc, tmp.N
                        a.l[0] = tmp.H * tmp.W
                    self.emit(self.bwd, L.OP_COLSUM, fc)
            coloff += Cs

    def _wgrad(self, srcs, conv, dz, taps, stride, gather, name, dbias_rows=0, dbr=None, dbr_ld=0):
        Cout = conv.out_channels
        goff = self.pgrad(conv.weight)
        kh, kw = conv.kernel_size
        span = max(max(t[0] for t in taps) - min(t[0] for t in taps), max(t[1] for t in taps) - min(t[1] for t in taps))
        wg_gather = 1 if (gather or (3 * stride + 1 + span) * (31 * stride + 1 + span) * 128 > 100 * 1024) else 0

        def shape_fill(d):
            d.N, d.OH, d.OW, d.Cout, d.nsrc = dz.N, dz.H, dz.W, Cout, len(srcs)
            for i, s in enumerate(srcs):
                d.src[i].C = _ru(s.t.C, 8)
                d.src[i].ldc, d.src[i].H, d.src[i].W = s.t.ldc, s.t.H, s.t.W
            d.ntaps = len(taps)
            for i, (dy, dx, wi) in enumerate(taps):
                d.tdy[i], d.tdx[i], d.tap_widx[i] = dy, dx, wi
            d.in_stride, d.gather = stride, wg_gather
            d.Cin_real, d.KHW = conv.in_channels, kh * kw
        probe = L.WgradDesc()
        shape_fill(probe)
        probe.flags = L.WGRAD_S3 if self.s3 else 0       # (the kernel choice, hence the grid, depends on it)
        probe.nsplit = 1
        nsplit = L.lib().pmf_conv_wgrad_nsplit(C.byref(probe))
        probe.nsplit = nsplit
        self.wg_scratch = max(self.wg_scratch, L.lib().pmf_conv_wgrad_workspace(C.byref(probe)))

        def f(op):
            d = op.u.wgrad
            shape_fill(d)
            for i, s in enumerate(srcs):
                self.src_struct(s, d.src[i])
            d.dz, d.dz_ldc = dz.buf.ptr, dz.ldc
            d.partial = self.wg_bufs[lane].ptr
            d.nsplit = nsplit
            d.dw_oihw = self.pgrad_buf.at(goff)
            d.accumulate = 0
            d.flags = L.WGRAD_S3 if self.s3 else 0
            if dbias_rows:
                d.dbias_rows, d.dbias_nrows, d.dbias_ld = dbr.ptr, dbias_rows, dbr_ld
                d.dbias_out = self.pgrad_buf.at(boff)
        boff = self.pgrad(conv.bias) if dbias_rows else None
        home = self.lane
        lane = self._wgrad_lane_of(home) if (self.wgrad_lane and self._wgrad_lane_of(home) != home) else home
        self.n_wgrad += 1
        meta = dict(family="conv_wgrad", flops=2.0 * dz.N * dz.H * dz.W * Cout * conv.in_channels * len(taps), name=name,
                    shape="%dx%dx%d %d->%d t%d" % (dz.N, dz.H, dz.W, conv.in_channels, Cout, len(taps)))
        ws = None
        if self.flat is not None and self.batch_reds:
            # flat training state (the product path): the partial-slab kernel now, the reduction into OIHW later -- the
            # reductions of RED_BATCH consecutive layers are ONE launch (pmf_conv_wgrad_reduce_multi); every layer keeps
            # its own workspace until then (1.65 GB at 64x2048 bs 2: nothing next to 288 GB)
            ws = self.act.alloc(max(L.lib().pmf_conv_wgrad_workspace(C.byref(probe)), 256))

        def emit_ops():          # on the current lane
            if ws is not None:
                def fb(op, f=f, ws=ws):
                    f(op)
                    op.u.wgrad.partial = ws.ptr
                self.emit(self.bwd, L.OP_WGRAD_PART, fb)
                self.meta_bwd[len(self.bwd) - 1] = meta
                pend = self.pending_reds.setdefault(lane, [])
                pend.append((fb, [conv.weight] + ([conv.bias] if dbias_rows else [])))
                if len(pend) >= self.red_batch:
                    self.flush_reds(lane)
            else:
                # per-tensor gradients (tests, stock DistributedDataParallel): partial slabs, then the reduction into
                # OIHW, back to back on the op's lane through that lane's workspace
                self.emit(self.bwd, L.OP_WGRAD_PART, f)
                self.meta_bwd[len(self.bwd) - 1] = meta
                self.emit(self.bwd, L.OP_WGRAD_RED, f)
                self.grad_done[id(conv.weight)] = len(self.bwd) - 1
                if dbias_rows:
                    self.grad_done[id(conv.bias)] = len(self.bwd) - 1
        if lane == home:
            emit_ops()
        else:
            # weight-gradient lane: the ops are DEFERRED and emitted in batches behind ONE event of the home lane (every
            # dz stays alive in the arena until the end of the pass, so running a weight gradient late is always legal)
            q = self._wg_deferred.setdefault(home, [])
            q.append(emit_ops)
            if len(q) >= self.wgrad_batch:
                self.flush_wgrads(home)

    def flush_wgrads(self, home, final=False):
        q = self._wg_deferred.pop(home, [])
        if not q:
            return
        ready = self.record_event(self.bwd, lane=home)      # everything the batch reads is complete after this op
        self._emit_batch(home, q, ready)

    def _emit_batch(self, home, q, ready):
        prev = self.lane
        self.lane = self._wgrad_lane_of(home)
        self.wait_event(self.bwd, ready)
        for fn in q:
            fn()
        self.lane = prev

    def flush_reds(self, lane):
        """emit ONE stage-2 launch for the weight gradients queued on ``lane`` by _wgrad (flat training state only)"""
        pend = self.pending_reds.pop(lane, [])
        if not pend:
            return
        prev, self.lane = self.lane, lane

        def f(op, pend=pend):
            lib = L.lib()
            n = len(pend)
            descs = (L.WgradDesc * n)()
            meta = (C.c_int32 * (8 * n))()
            tmp = L.Op()
            blocks = 0
            for j, (fill, _) in enumerate(pend):
                C.memset(C.addressof(tmp), 0, C.sizeof(tmp))      # fills only set what they use (as on a fresh op)
                fill(tmp)
                C.memmove(C.addressof(descs[j]), C.addressof(tmp.u.wgrad), C.sizeof(L.WgradDesc))
                row = (C.c_int32 * 8)()
                nb = lib.pmf_conv_wgrad_reduce_plan(C.byref(descs[j]), row)
                if nb <= 0:
                    raise RuntimeError("pmf_conv_wgrad_reduce_plan failed: %d" % nb)
                row[0] = blocks
                meta[8 * j:8 * j + 8] = row[:]
                blocks += nb
            jd = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)
            md = torch.frombuffer(bytearray(bytes(meta)), dtype=torch.uint8).to(self.device)
            self._red_tables.append((jd, md))                  # keep the device tables alive with the plan
            a = op.u.sm
            a.p[0], a.p[1] = jd.data_ptr(), md.data_ptr()
            a.i[0], a.i[1] = n, blocks
        self.emit(self.bwd, L.OP_WGRAD_RED_MULTI, f)
        # data parallelism: the gradients this launch finalises (and everything listed before it) may be all-reduced as
        # soon as it AND the ops emitted so far on the two home lanes have run -- one event each; the engine makes its RCCL
        # side stream wait for them (pmf_plan_event_wait), no cut through the plan
        evs = [self._event_after(self._last_op[(id(self.bwd), lane)])[0]]
        for hl in (0, 1):
            ev = self.record_event(self.bwd, lane=hl)
            if ev is not None:
                evs.append(ev[0])
        self.dp_events.append((len(self.bwd), evs))
        self.lane = prev
        for _, params in pend:
            for p in params:
                self.grad_done[id(p)] = len(self.bwd) - 1
