"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- record / inject the piecewise-linear DECISIONS of an oracle forward pass.

The reference's backward is autograd through ``F.leaky_relu`` / ``F.relu`` / ``F.max_pool2d`` (pc_processor/models/
salsanext.py:27-33, pmf_net.py:20-29,94): every one of them is piecewise linear, so two fp32 paths whose pre-activations
differ by rounding agree in the forward pass and can differ by whole terms in the backward pass when a value sits on its
kink.  To tell such a flip from a kernel defect, the float64 / fp32 oracle passes can be run with the
decisions (sign masks, max-pool argmax positions) of ANOTHER path injected: all passes then differentiate the same piecewise
linear function, and what is left between them is rounding.

Sites are keyed without touching the oracle's module code: while the context is active the three functionals are replaced
in ``torch.nn.functional`` (``nn.LeakyReLU`` / ``nn.ReLU`` / ``nn.MaxPool2d`` resolve them at call time), and global forward
hooks track the last ``Conv2d`` / ``BatchNorm2d`` that ran:

    ("lrelu", conv)      F.leaky_relu right behind conv                     (conv -> LeakyReLU -> BN, salsanext.py:27-33)
    ("relu", conv)       F.relu on the output of the BatchNorm behind conv  (conv -> BN -> ReLU, pmf_net.py:20-29)
    ("relu_out", block)  F.relu on anything else: the residual sum of a ResNet block (torchvision BasicBlock / Bottleneck)
    ("maxpool", conv)    F.max_pool2d of a tensor that carries a gradient   (the stem's pool, pmf_net.py:94)

``conv`` / ``block`` are qualified module names (= state-dict prefixes, identical in the reference, this oracle and the HIP
product model).  Decisions: bool tensors (True = the slope-1 branch) for the activations, int64 flat input positions
(``return_indices`` layout) for the pool."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ActSites:
    """``with ActSites(net) as rec: net(...)`` records ``rec.decisions``; ``with ActSites(net, inject=d): net(...)`` runs the
    forward with the decisions ``d`` (every site met must be in ``d``: KeyError otherwise; ``rec.unused`` lists keys of ``d``
    no site asked for)."""

    def __init__(self, net, inject=None):
        self.names = {id(m): n for n, m in net.named_modules()}
        self.inject = inject
        self.decisions = {}
        self.last_conv, self.last_out, self.last_kind = None, None, None
        self.seen = []

    # ---- bookkeeping -----------------------------------------------------------------------------
    def _hook(self, mod, inp, out):
        if isinstance(mod, nn.Conv2d) and id(mod) in self.names:
            self.last_conv, self.last_out, self.last_kind = self.names[id(mod)], out, "conv"
        elif isinstance(mod, nn.BatchNorm2d) and id(mod) in self.names:
            self.last_out, self.last_kind = out, "bn"

    def _site(self, kind, key):
        k = (kind, key)
        if k in self.seen:
            raise RuntimeError("activation site %r met twice in one forward pass" % (k,))
        self.seen.append(k)
        return k

    # ---- the three functionals --------------------------------------------------------------------
    def _leaky_relu(self, x, negative_slope=0.01, inplace=False):
        k = self._site("lrelu", self.last_conv)
        if self.inject is None:
            self.decisions[k] = (x > 0).detach()
            return self._f["leaky_relu"](x, negative_slope, False)
        m = self.inject[k].to(x.device)
        return x * torch.where(m, torch.ones((), dtype=x.dtype), torch.full((), negative_slope, dtype=x.dtype))

    def _relu(self, x, inplace=False):
        behind_bn = self.last_kind == "bn" and x is self.last_out
        k = self._site("relu", self.last_conv) if behind_bn else self._site("relu_out", self.last_conv.rsplit(".", 1)[0])
        if self.inject is None:
            self.decisions[k] = (x > 0).detach()
            return self._f["relu"](x, False)
        return x * self.inject[k].to(x.device).to(x.dtype)

    def _max_pool2d(self, x, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False, return_indices=False):
        if return_indices or not (torch.is_grad_enabled() and x.requires_grad):
            return self._f["max_pool2d"](x, kernel_size, stride, padding, dilation, ceil_mode, return_indices)
        k = self._site("maxpool", self.last_conv)
        if self.inject is None:
            y, idx = self._f["max_pool2d"](x, kernel_size, stride, padding, dilation, ceil_mode, True)
            self.decisions[k] = idx.detach()
            return y
        idx = self.inject[k].to(x.device)
        return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)

    # ---- context ------------------------------------------------------------------------------------
    def __enter__(self):
        self._f = {n: getattr(F, n) for n in ("leaky_relu", "relu", "max_pool2d")}
        F.leaky_relu, F.relu, F.max_pool2d = self._leaky_relu, self._relu, self._max_pool2d
        self._h = nn.modules.module.register_module_forward_hook(self._hook)
        return self

    def __exit__(self, *exc):
        F.leaky_relu, F.relu, F.max_pool2d = self._f["leaky_relu"], self._f["relu"], self._f["max_pool2d"]
        self._h.remove()
        self.last_out = None
        return False

    @property
    def unused(self):
        return [] if self.inject is None else [k for k in self.inject if k not in self.seen]
