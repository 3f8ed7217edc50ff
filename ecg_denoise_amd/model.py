"""Host-side mirror of the reference's model boundary.

`RALENet` / `UNet` duck-type the nn.Module surface that `denoise_train.train` uses
(`model(x)`, `.parameters()`, `.train()/.eval()`, `.state_dict()/.load_state_dict()`,
reference: denoise_train.py:20-24,44,52,66,72,93; main.py:63-77) on top of
libralenet.so.  PyTorch is used only as the owner of device memory and streams: every
tensor handed to the library is a raw device pointer, all arithmetic is HIP.
"""
import ctypes as C
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

RW_LEN = (32, 16, 8, 4)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Engine:
    """One libralenet handle + the caller-owned flat buffers."""

    def __init__(self, variant, leads, L, max_batch, train, device):
        self.variant, self.leads, self.L = variant, int(leads), int(L)
        self.max_batch, self.trainable = int(max_batch), bool(train)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RalError("the RA-LENet path runs on a HIP device only (no CPU fallback)")
        self.cfg = _lib.make_config(variant, leads, L, max_batch, train)
        self.entries = _lib.layout(self.cfg)
        L_ = _lib.lib()
        self.nparam = L_.ral_param_floats(C.byref(self.cfg))
        self.nstate = L_.ral_state_floats(C.byref(self.cfg))
        with torch.cuda.device(self.device):
            z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.device)
            self.params = z(self.nparam)
            self.state = z(self.nstate)
            self.grads = z(self.nparam) if train else None
            self.adam_m = z(self.nparam) if train else None
            self.adam_v = z(self.nparam) if train else None
            self.bn_sums = z(L_.ral_bn_sums_doubles(C.byref(self.cfg)), torch.float64) if train else None
            h = C.c_void_p()
            _lib.check(L_.ral_create(C.byref(self.cfg), C.byref(h)))
            self.h = h
            _lib.check(L_.ral_bind(h, _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                   _ptr(self.state), _ptr(self.bn_sums)))
        self.counters = {e["name"]: 0 for e in self.entries if e["kind"] == _lib.KIND_COUNTER}

    def __del__(self):
        h = getattr(self, "h", None)
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            try:
                _lib._lib.ral_destroy(h)
            except Exception:
                pass
            self.h = None

    def view(self, buf, e):
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        return buf[e["offset"]:e["offset"] + n].view(e["shape"])


class _RalFunction(torch.autograd.Function):
    """Host glue with no arithmetic of its own: puts the library's forward / backward pair into a torch autograd graph, so that
    the reference loop runs as written (`pre = model(data); loss = F.mse_loss(pre, target); loss.backward();
    optimizer.step()`, denoise_train.py:51-57).  The saved state is the model handle (the library keeps the activations of
    its most recent training forward)."""

    @staticmethod
    def forward(ctx, x, anchor, model):
        ctx.model = model
        ctx.need_dx = bool(x.requires_grad)
        y = model._forward_raw(x)
        model._fwd_gen += 1
        ctx.gen = model._fwd_gen
        return y

    @staticmethod
    def backward(ctx, dy):
        m = ctx.model
        if ctx.gen != m._fwd_gen:
            raise _lib.RalError("backward through a forward that is not this model's most recent training forward "
                                "(the library keeps one set of activations)")
        return m._backward_autograd(dy, ctx.need_dx), None, None


class _AutogradMixin:
    """`autograd=True` at construction (or `enable_autograd()`): `model(x)` in training mode returns a tensor with a
    `grad_fn`, `parameters()` yields LEAF views into the flat parameter buffer whose `.grad` aliases the flat gradient
    buffer - `optim.Adam(model.parameters())`, `loss.backward()`, `optimizer.step()`, `optimizer.zero_grad()` work unchanged.
    This is the slow path (torch's optimiser and loss kernels instead of the fused ones); `train_step` is the fast one."""

    _autograd = False
    _fwd_gen = 0

    def enable_autograd(self, on=True):
        self._autograd = bool(on)
        self._leaves = None
        return self

    def _flat_pairs(self):
        """[(name, parameter view, gradient view)] over the flat buffers, in `parameters()` order"""
        raise NotImplementedError

    def _leaf_parameters(self):
        if getattr(self, "_leaves", None) is None:
            self._leaves = []
            for name, p, g in self._flat_pairs():
                leaf = p.detach()              # a view of the flat buffer: the optimiser's in-place update is the library's weight
                leaf.requires_grad_(True)
                self._leaves.append((name, leaf, g))
            self._anchor = torch.zeros(1, device=self._leaves[0][1].device, requires_grad=True)
        return self._leaves

    def _forward_autograd(self, x):
        self._leaf_parameters()
        return _RalFunction.apply(x, self._anchor, self)

    def _backward_autograd(self, dy, need_dx):
        leaves = self._leaf_parameters()
        # torch accumulates into .grad; the library overwrites its flat buffer.  A gradient that is still attached (the caller
        # did not zero it, or zeroed it in place) is kept and added back.
        keep = [(leaf, g, (leaf.grad.clone() if leaf.grad is not None and leaf.grad.data_ptr() == g.data_ptr() else None))
                for _, leaf, g in leaves if leaf.grad is not None]
        dx = self._backward_raw(dy.contiguous(), need_dx)
        for leaf, g, old in keep:
            if old is not None:
                g.add_(old)
            else:                               # a gradient tensor of the caller's own: accumulate into it, as autograd would
                leaf.grad.add_(g)
        for _, leaf, g in leaves:
            if leaf.grad is None:
                leaf.grad = g                   # alias: optimizer.step() reads the library's gradient buffer directly
        return dx


class _ModuleBase(_AutogradMixin):
    VARIANT = None

    def __init__(self, variant, leads=2, L=512, max_batch=32, train=True, device="cuda:0", seed=None, autograd=False):
        self.eng = _Engine(variant, leads, L, max_batch, train, device)
        self.training = bool(train)
        self.step_count = 0
        self._dy = None
        self._loss_sum = None
        self.reset_parameters(seed)
        self.enable_autograd(autograd)

    # ---- nn.Module surface ------------------------------------------------------
    def train(self, mode=True):
        if mode and not self.eng.trainable:
            raise _lib.RalError("model was created with train=False")
        self.training = bool(mode)
        self._params_changed()
        return self

    def _params_changed(self):
        """tell the library whether the parameters stand still (eval mode: the weight planes of one forward serve the next) - and, by
        telling it again, that they may have been rewritten (load_state_dict, reset_parameters).  Parameters edited in place in eval
        mode through `parameters()` / `eng.params` need this call too."""
        if self.eng.variant in ("nra", "full", "mlp") and getattr(self.eng, "h", None):
            _lib.check(_lib.lib().ral_set_option(self.eng.h, b"static_params", 0 if self.training else 1))

    def eval(self):
        return self.train(False)

    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    def named_parameters(self):
        if self._autograd:
            for name, leaf, _ in self._leaf_parameters():
                yield name, leaf
            return
        for e in self.eng.entries:
            if e["kind"] == _lib.KIND_PARAM:
                yield e["name"], self.eng.view(self.eng.params, e)

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def _flat_pairs(self):
        if self.eng.grads is None:
            raise _lib.RalError("model was created with train=False")
        return [(e["name"], self.eng.view(self.eng.params, e), self.eng.view(self.eng.grads, e))
                for e in self.eng.entries if e["kind"] == _lib.KIND_PARAM]

    def named_grads(self):
        return OrderedDict((e["name"], self.eng.view(self.eng.grads, e)) for e in self.eng.entries
                           if e["kind"] == _lib.KIND_PARAM)

    def num_parameters(self):
        return sum(int(np.prod(e["shape"])) for e in self.eng.entries if e["kind"] == _lib.KIND_PARAM)

    def state_dict(self):
        """Reference checkpoint contract: same keys, shapes and dtypes, same order."""
        sd = OrderedDict()
        for e in self.eng.entries:
            k = e["kind"]
            if k == _lib.KIND_PARAM:
                sd[e["name"]] = self.eng.view(self.eng.params, e).detach().clone()
            elif k == _lib.KIND_STATE:
                sd[e["name"]] = self.eng.view(self.eng.state, e).detach().clone()
            elif k == _lib.KIND_COUNTER:
                sd[e["name"]] = torch.tensor(self.eng.counters[e["name"]], dtype=torch.int64)
            else:  # relative_position_index (transformer.py:517-529): i - j + Len - 1
                n = e["shape"][0]
                i = torch.arange(n)
                sd[e["name"]] = (i[:, None] - i[None, :] + n - 1).to(torch.int64)
        return sd

    def load_state_dict(self, sd, strict=True):
        missing, unexpected = [], [k for k in sd if k not in {e["name"] for e in self.eng.entries}]
        for e in self.eng.entries:
            if e["name"] not in sd:
                if e["kind"] in (_lib.KIND_PARAM, _lib.KIND_STATE):
                    missing.append(e["name"])
                continue
            v = sd[e["name"]]
            if e["kind"] == _lib.KIND_PARAM or e["kind"] == _lib.KIND_STATE:
                buf = self.eng.params if e["kind"] == _lib.KIND_PARAM else self.eng.state
                dst = self.eng.view(buf, e)
                v = torch.as_tensor(v)
                if tuple(v.shape) != tuple(dst.shape):
                    raise _lib.RalError(f"size mismatch for {e['name']}: {tuple(v.shape)} vs {tuple(dst.shape)}")
                dst.copy_(v.to(dtype=torch.float32, device=dst.device))
            elif e["kind"] == _lib.KIND_COUNTER:
                self.eng.counters[e["name"]] = int(v)
        self._params_changed()
        if strict and (missing or unexpected):
            raise _lib.RalError(f"load_state_dict: missing {missing}, unexpected {unexpected}")
        return missing, unexpected

    def reset_parameters(self, seed=None):
        """PyTorch default initialisation (nothing is re-initialised by the reference):
        Linear/Conv weight and bias ~ U(+-1/sqrt(fan_in)); norm weight 1, bias 0; running stats 0/1;
        R-wave tables 0."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.eng.nparam, dtype=np.float32)
        fan = None
        # a 1-D parameter next to a running_mean buffer is a BatchNorm affine
        bn_prefix = {e["name"][:-len("running_mean")] for e in self.eng.entries if e["name"].endswith(".running_mean")}
        for e in self.eng.entries:
            if e["kind"] != _lib.KIND_PARAM:
                continue
            shp, name = e["shape"], e["name"]
            n = int(np.prod(shp))
            if "relative_position_bias_table" in name:
                a = np.zeros(n)
            elif len(shp) == 1 and (".norm" in name or ".bn." in name or name.startswith("conv1.2.")
                                    or name.startswith("bottleneck.2.") or name.startswith("bottleneck.5.")
                                    or name[:name.rfind(".") + 1] in bn_prefix):
                a = np.ones(n) if name.endswith("weight") else np.zeros(n)
            elif len(shp) >= 2:
                fan = int(np.prod(shp[1:]))
                a = rng.uniform(-1 / math.sqrt(fan), 1 / math.sqrt(fan), n)
            else:
                b = 1 / math.sqrt(fan) if fan else 0.1
                a = rng.uniform(-b, b, n)
            host[e["offset"]:e["offset"] + n] = a
        self.eng.params.copy_(torch.from_numpy(host))
        st = np.zeros(self.eng.nstate, dtype=np.float32)
        for e in self.eng.entries:
            if e["kind"] == _lib.KIND_STATE and e["name"].endswith("running_var"):
                st[e["offset"]:e["offset"] + int(np.prod(e["shape"]))] = 1.0
        self.eng.state.copy_(torch.from_numpy(st))
        for k in self.eng.counters:
            self.eng.counters[k] = 0
        self._params_changed()

    # ---- hot path -----------------------------------------------------------------
    def _check_x(self, x):
        e = self.eng
        if x.dim() != 3 or x.shape[1] != e.leads or x.shape[2] != e.L:
            raise _lib.RalError(f"expected input (B, {e.leads}, {e.L}), got {tuple(x.shape)}")
        if x.shape[0] > e.max_batch:
            raise _lib.RalError(f"batch {x.shape[0]} > max_batch {e.max_batch}")
        if x.dtype != torch.float32 or not x.is_cuda:
            raise _lib.RalError("input must be a float32 HIP tensor")
        return x.contiguous()

    def _forward_raw(self, x):
        x = self._check_x(x)
        y = torch.empty_like(x)
        self._x = x  # keep alive for backward (the stem gradient re-reads it)
        self._dy_fused = False   # (the sums `forward_loss` left for `backward()` belonged to the previous forward)
        _lib.check(_lib.lib().ral_forward(self.eng.h, _ptr(x), _ptr(y), x.shape[0], int(self.training), _stream()))
        if self.training:
            for k in self.eng.counters:
                self.eng.counters[k] += 1
        return y

    def _backward_raw(self, dy, want_dx):
        dx = torch.empty_like(dy) if want_dx else None
        self._dy_fused = False
        _lib.check(_lib.lib().ral_backward(self.eng.h, _ptr(dy), _ptr(dx), dy.shape[0], _stream()))
        return dx

    def forward(self, x):
        if self._autograd and self.training and torch.is_grad_enabled():
            return self._forward_autograd(x)
        return self._forward_raw(x)

    def __call__(self, x):
        return self.forward(x)

    def loss_and_metrics(self, pred, target, want_grad=True, global_windows=None):
        """F.mse_loss(pred, target), SNR(target, pred), RMSE(target, pred) of denoise_train.py:53,58-59
        in one pass; keeps d loss / d pred for `backward()`."""
        B = pred.shape[0]
        gw = int(global_windows or B)
        target = target.contiguous()
        snr = torch.empty(B, dtype=torch.float32, device=pred.device)
        rmse = torch.empty_like(snr)
        means = torch.empty(3, dtype=torch.float64, device=pred.device)   # loss share, sum SNR / gw, sum RMSE / gw
        loss = means[:1]
        dy = torch.empty_like(pred) if want_grad else None
        # (the means are finished on the device, ral_loss_means: no fill kernel before the launch, no division or metric
        # reduction after it)
        if getattr(self, "_loss_scratch", None) is None:
            self._loss_scratch = torch.zeros(64, dtype=torch.float64, device=pred.device)
        _lib.check(_lib.lib().ral_loss_means(_ptr(pred), _ptr(target), pred[0].numel(), B, gw, _ptr(dy), _ptr(snr), _ptr(rmse),
                                             _ptr(means), _ptr(self._loss_scratch), _stream()))
        self._means = means
        self._dy = dy
        self._dy_fused = False
        return loss, snr, rmse

    def forward_loss(self, x, target):
        """`pred = model(x); loss = criterion(pred, target)` and the step's SNR / RMSE (denoise_train.py:52-53,58-59) as ONE
        library call in training mode: the same numbers as `forward` + `loss_and_metrics`; the U-Net's output BatchNorm, its
        loss sums and the first sums of its backward pass then share one pass over the output (ral_forward_loss_means)."""
        if not self.training or type(self)._forward_raw is not _ModuleBase._forward_raw:   # (a subclass with bookkeeping of its own in forward)
            pred = self.forward(x)
            return (pred,) + tuple(self.loss_and_metrics(pred, target))
        x = self._check_x(x)
        target = target.contiguous()
        B = x.shape[0]
        y = torch.empty_like(x)
        self._x = x
        snr = torch.empty(B, dtype=torch.float32, device=x.device)
        rmse = torch.empty_like(snr)
        means = torch.empty(3, dtype=torch.float64, device=x.device)
        dy = torch.empty_like(x)
        if getattr(self, "_loss_scratch", None) is None:
            self._loss_scratch = torch.zeros(64, dtype=torch.float64, device=x.device)
        _lib.check(_lib.lib().ral_forward_loss_means(self.eng.h, _ptr(x), _ptr(target), _ptr(y), B, _ptr(dy), _ptr(snr), _ptr(rmse),
                                                     _ptr(means), _ptr(self._loss_scratch), _stream()))
        for k in self.eng.counters:
            self.eng.counters[k] += 1
        self._means = means
        self._dy = dy
        self._dy_fused = True
        return y, means[:1], snr, rmse

    def backward(self, dy=None, want_dx=False):
        """`loss.backward()` (denoise_train.py:56).  dy = None: the gradient the last loss call left (`self._dy`).  For the
        U-Net, `forward_loss` also left the first BatchNorm-backward sums of exactly that gradient, and `backward()` starts
        from them (a NULL dy at the C ABI); a dy passed explicitly - also `m.backward(m._dy.mul_(k))` - is always summed
        again, so rescaling it in place is safe."""
        fused = dy is None and getattr(self, "_dy_fused", False) and self.eng.variant == "unet"
        dy = self._dy if dy is None else dy.contiguous()
        dx = torch.empty_like(dy) if want_dx else None
        self._dy_fused = False
        _lib.check(_lib.lib().ral_backward(self.eng.h, C.c_void_p(0) if fused else _ptr(dy), _ptr(dx), dy.shape[0], _stream()))
        return dx

    def backward_input(self, dy):
        """d loss / d input with every weight frozen (`requires_grad = False`, ralenet_12leads.py:694-696): no weight
        gradient is formed."""
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.check(_lib.lib().ral_backward_input(self.eng.h, _ptr(dy), _ptr(dx), dy.shape[0], _stream()))
        return dx

    def zero_grad(self):
        pass  # ral_backward zeroes the flat gradient buffer itself

    def step(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        """torch.optim.Adam(lr=1e-3) of denoise_train.py:24 as one fused flat kernel."""
        self.step_count += 1
        _lib.check(_lib.lib().ral_adam_step(self.eng.h, lr, betas[0], betas[1], eps, self.step_count, grad_scale,
                                            _stream()))

    def train_step(self, x, target, lr=1e-3):
        """zero_grad -> forward -> mse -> backward -> Adam (denoise_train.py:51-57)."""
        pred, loss, snr, rmse = self.forward_loss(x, target)
        self.backward()
        self.step(lr)
        return {"loss": loss, "pred": pred, "snr": snr, "rmse": rmse}

    def debug_tensor(self, name):
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(_lib.lib().ral_debug_tensor(self.eng.h, name.encode(), C.byref(p), C.byref(n)))
        out = torch.empty(n.value, dtype=torch.float32, device=self.eng.device)
        torch.cuda.synchronize()
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(C.c_void_p(out.data_ptr()), p, n.value * 4, 3)  # hipMemcpyDeviceToDevice
        if rc != 0:
            raise _lib.RalError(f"hipMemcpy failed ({rc})")
        return out


class RALENet(_ModuleBase):
    """variant: "nra" = model/raletransformer.py::ralenet(), "full" =
    model/transformer.py::ralenet(high_level_enhence=True), "mlp" = ...(low_level_enhence=False)
    (main.py:69-77), generalised to `leads` in {1,2} and any `L` that is a multiple of 16 up to 1024 the way
    SURVEY §8c states (R-wave window stays centred in the level's tokens, Len constants unchanged).  Multiples of 256 are the
    fast shapes; other lengths run on padded token slots with the missing tokens masked (fp32 MFMA kernels)."""

    def __init__(self, variant="full", leads=2, L=512, max_batch=32, train=True, device="cuda:0", seed=None, autograd=False):
        if variant not in ("nra", "full", "mlp"):
            raise _lib.RalError(f"unknown RA-LENet variant {variant!r}")
        super().__init__(variant, leads, L, max_batch, train, device, seed, autograd)


class UNet(_ModuleBase):
    """model/UNet.py::UNet (main.py:63-65): conv U-Net baseline, BatchNorm after every conv
    (the reference's LazyBatchNorm1d materialised eagerly, quirk A15)."""

    def __init__(self, leads=2, L=512, max_batch=32, train=True, device="cuda:0", seed=None, autograd=False):
        super().__init__("unet", leads, L, max_batch, train, device, seed, autograd)


class ACDAE(_ModuleBase):
    """model/ACDAE.py::ACDAE (main.py:66-68): the attention-based convolutional denoising auto-encoder the reference
    compares against - Conv1d / MaxPool / LeakyReLU encoder, ConvTranspose1d / linear Upsample / LeakyReLU / ECA decoder
    with additive skips, 2 leads, no BatchNorm (train and eval forward are the same function)."""

    def __init__(self, L=512, max_batch=32, train=True, device="cuda:0", seed=None, autograd=False):
        super().__init__("acdae", 2, L, max_batch, train, device, seed, autograd)


class DANet(_ModuleBase):
    """model/DAM.py::Seq2Seq2 (main.py:67-68, test_cls.py:81-83), the "DANet" the reference compares against: four
    strided-conv encoder cells and four transposed-conv decoder cells, each conv -> APReLU -> BatchNorm1d, the first
    three decoder cells followed by a DAM (channel + spatial attention), decoder inputs = previous output + encoder
    feature.  36 BatchNorms; the two inside each DAM belong to ONE fcn that is applied to two batches per forward
    (`fcn1` and `fcn2` are the same modules): their state_dict entries appear under both names, their running
    statistics take two momentum updates per training forward and `num_batches_tracked` advances by 2."""

    def __init__(self, L=512, max_batch=32, train=True, device="cuda:0", seed=None, leads=2, autograd=False):
        super().__init__("danet", leads, L, max_batch, train, device, seed, autograd)      # (2 output channels: leads must be 2)

    def named_parameters(self):
        for k, v in super().named_parameters():
            if ".dam.fcn2." not in k:          # torch lists a shared parameter once, under its first name
                yield k, v

    def named_grads(self):
        return OrderedDict((k, v) for k, v in super().named_grads().items() if ".dam.fcn2." not in k)

    def _flat_pairs(self):
        return [t for t in super()._flat_pairs() if ".dam.fcn2." not in t[0]]

    def num_parameters(self):
        return sum(int(np.prod(e["shape"])) for e in self.eng.entries
                   if e["kind"] == _lib.KIND_PARAM and ".dam.fcn2." not in e["name"])

    def _forward_raw(self, x):
        y = super()._forward_raw(x)
        if self.training:
            for k in self.eng.counters:
                if ".dam.fcn" in k:
                    self.eng.counters[k] += 1   # the second batch through the shared fcn
        return y


class NewRALE(_AutogradMixin):
    """model/ralenet_12leads.py::newrale — 12-lead adapter around a pretrained, frozen RA-LENet
    (Transfer_learning.py:71-75): Conv1d 12->6->2 (k13, LeakyReLU 0.01), RA-LENet, Conv1d 2->6->12.
    Only the 2 210 adapter parameters train; the inner model keeps running in whatever mode
    `.train()` sets, so its BatchNorm still uses and updates batch statistics (reference quirk A16)."""

    SHAPES = OrderedDict([("conv1.weight", (6, 12, 13)), ("conv1.bias", (6,)), ("conv2.weight", (2, 6, 13)),
                          ("conv2.bias", (2,)), ("conv3.weight", (6, 2, 13)), ("conv3.bias", (6,)),
                          ("conv4.weight", (12, 6, 13)), ("conv4.bias", (12,))])

    def __init__(self, pretrained_rale_model, seed=None, autograd=False):
        self.rale = pretrained_rale_model
        self.enable_autograd(autograd)
        if self.rale.eng.leads != 2:
            raise _lib.RalError("newrale wraps a 2-lead RA-LENet")
        self.device, self.L = self.rale.eng.device, self.rale.eng.L
        self.off, cur = OrderedDict(), 0
        for k, shp in self.SHAPES.items():
            self.off[k] = cur
            cur += (int(np.prod(shp)) + 3) // 4 * 4
        z = lambda: torch.zeros(cur, dtype=torch.float32, device=self.device)
        self.params, self.grads, self.adam_m, self.adam_v = z(), z(), z(), z()
        self.training, self.step_count = True, 0
        rng = np.random.default_rng(seed)
        fan = None
        for k, shp in self.SHAPES.items():
            if len(shp) == 3:
                fan = shp[1] * shp[2]
            b = 1.0 / math.sqrt(fan)
            self._view(self.params, k).copy_(torch.from_numpy(rng.uniform(-b, b, shp).astype(np.float32)))

    def _view(self, buf, k):
        n = int(np.prod(self.SHAPES[k]))
        return buf[self.off[k]:self.off[k] + n].view(self.SHAPES[k])

    def train(self, mode=True):
        self.training = bool(mode); self.rale.train(mode); return self

    def eval(self):
        return self.train(False)

    def cuda(self, device=None):
        return self

    def named_parameters(self):
        if self._autograd:
            return [(k, leaf) for k, leaf, _ in self._leaf_parameters()]
        return [(k, self._view(self.params, k)) for k in self.SHAPES]

    def parameters(self):            # trainable parameters only (requires_grad filter of the reference)
        return [p for _, p in self.named_parameters()]

    def _flat_pairs(self):
        return [(k, self._view(self.params, k), self._view(self.grads, k)) for k in self.SHAPES]

    def named_grads(self):
        return OrderedDict((k, self._view(self.grads, k)) for k in self.SHAPES)

    def state_dict(self):
        sd = OrderedDict()
        for k in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias"):
            sd[k] = self._view(self.params, k).clone()
        for k, v in self.rale.state_dict().items():
            sd["rale." + k] = v
        for k in ("conv3.weight", "conv3.bias", "conv4.weight", "conv4.bias"):
            sd[k] = self._view(self.params, k).clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        for k in self.SHAPES:
            if k in sd:
                self._view(self.params, k).copy_(torch.as_tensor(sd[k]).to(self.device, torch.float32))
            elif strict:
                raise _lib.RalError(f"missing {k}")
        inner = OrderedDict((k[5:], v) for k, v in sd.items() if k.startswith("rale."))
        if inner:
            self.rale.load_state_dict(inner, strict)

    def _conv(self, name, x, cout, lrelu):
        y = torch.empty(x.shape[0], cout, x.shape[2], dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().ral_conv13_forward(_ptr(x), _ptr(self._view(self.params, name + ".weight")),
                                                 _ptr(self._view(self.params, name + ".bias")), _ptr(y), x.shape[0],
                                                 x.shape[1], cout, x.shape[2], int(lrelu), _stream()))
        return y

    # forward and backward are written as (adapter convs in front) / (inner RA-LENet) / (adapter convs behind) so that the
    # data-parallel adapter (dp.NewRALEEngineAdapter) can cut the inner model at its BatchNorm reduction points
    def _forward_pre(self, x):
        x = x.contiguous()
        if x.dim() != 3 or x.shape[1] != 12 or x.shape[2] != self.L or not x.is_cuda:
            raise _lib.RalError(f"expected a HIP tensor (B, 12, {self.L}), got {tuple(x.shape)}")
        a1 = self._conv("conv1", x, 6, True)
        a2 = self._conv("conv2", a1, 2, True)
        self._saved = [x, a1, a2]
        return a2

    def _forward_post(self, r):
        a3 = self._conv("conv3", r, 6, True)
        y = self._conv("conv4", a3, 12, False)
        self._saved += [r, a3, y]
        return y

    def _forward_raw(self, x):
        return self._forward_post(self.rale._forward_raw(self._forward_pre(x)))

    def _backward_raw(self, dy, want_dx):
        if want_dx:
            raise _lib.RalError("newrale: the input gradient is not provided")
        self.backward(dy)
        return None

    def forward(self, x):
        if self._autograd and self.training and torch.is_grad_enabled():
            return self._forward_autograd(x)
        return self._forward_raw(x)

    def __call__(self, x):
        return self.forward(x)

    def loss_and_metrics(self, pred, target, want_grad=True, global_windows=None):
        """F.mse_loss / SNR / RMSE over the flattened 12 x L windows (denoise_train.py:53,58-59) in the fused loss kernel."""
        B, n = pred.shape[0], pred[0].numel()
        gw = int(global_windows or B)
        pred, target = pred.contiguous(), target.contiguous()
        snr = torch.empty(B, dtype=torch.float32, device=pred.device)
        rmse = torch.empty_like(snr)
        means = torch.empty(3, dtype=torch.float64, device=pred.device)   # loss share, sum SNR / gw, sum RMSE / gw
        loss = means[:1]
        dy = torch.empty_like(pred) if want_grad else None
        if getattr(self, "_loss_scratch", None) is None:
            self._loss_scratch = torch.zeros(64, dtype=torch.float64, device=pred.device)
        _lib.check(_lib.lib().ral_loss_means(_ptr(pred), _ptr(target), n, B, gw, _ptr(dy), _ptr(snr), _ptr(rmse),
                                             _ptr(means), _ptr(self._loss_scratch), _stream()))
        self._means = means
        self._dy = dy
        return loss, snr, rmse

    def _conv_bwd(self, name, x, y, dy, lrelu, want_dx):
        dx = torch.empty_like(x) if want_dx else None
        _lib.check(_lib.lib().ral_conv13_backward(_ptr(x), _ptr(y), _ptr(dy), _ptr(self._view(self.params, name + ".weight")),
                                                  _ptr(self._view(self.grads, name + ".weight")),
                                                  _ptr(self._view(self.grads, name + ".bias")), _ptr(dx), x.shape[0],
                                                  x.shape[1], y.shape[1], x.shape[2], int(lrelu), _stream()))
        return dx

    def _backward_pre(self, dy=None):
        dy = (self._dy if dy is None else dy).contiguous()
        x, a1, a2, r, a3, y = self._saved
        self.grads.zero_()
        d3 = self._conv_bwd("conv4", a3, y, dy, False, True)
        return self._conv_bwd("conv3", r, a3, d3, True, True)

    def _backward_post(self, d2):
        x, a1, a2, r, a3, y = self._saved
        d1 = self._conv_bwd("conv2", a1, a2, d2, True, True)
        self._conv_bwd("conv1", x, a1, d1, True, False)

    def backward(self, dy=None):
        dr = self._backward_pre(dy)
        self._backward_post(self.rale.backward_input(dr))        # frozen weights: input gradient only, no dW kernels

    def step(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        self.step_count += 1
        _lib.check(_lib.lib().ral_adam_flat(_ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                            self.params.numel(), lr, betas[0], betas[1], eps, self.step_count, grad_scale,
                                            _stream()))

    def train_step(self, x, target, lr=1e-3):
        pred = self.forward(x)
        loss, snr, rmse = self.loss_and_metrics(pred, target)
        self.backward()
        self.step(lr)
        return {"loss": loss, "pred": pred, "snr": snr, "rmse": rmse}


def ralenet(high_level_enhence=False, low_level_enhence=True, **kw):
    """Constructor spelled like model/transformer.py::ralenet (main.py:71-77)."""
    return RALENet("full" if high_level_enhence else "mlp", **kw)


def pe_table_host(L, level, leads=2, variant="full"):
    cfg = _lib.make_config(variant, leads, L, 1, 0)
    C_ = 8 << level
    n = L >> level
    out = np.empty((n, C_), dtype=np.float32)
    _lib.check(_lib.lib().ral_pe_table(C.byref(cfg), level, out.ctypes.data_as(C.c_void_p), out.size))
    return out
