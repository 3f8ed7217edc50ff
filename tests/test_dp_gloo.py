"""Data-parallel host logic on CPU: world_size-2 `gloo` processes driving the trainer of
ecg_denoise_amd/dp.py with a CPU stand-in engine (built from the oracle, split at the two
BatchNorm reduction points exactly like the C ABI) must reproduce the single-process step
on the concatenated batch: same loss, same gradients, same parameters after Adam."""
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

import ralenet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Same call surface as dp.HipEngineAdapter; fp64 torch-CPU arithmetic."""

    def __init__(self, variant, leads, seed):
        self.variant = variant
        self.p = OrderedDict((k, v.double()) for k, v in O.init_params(O.ralenet_param_shapes(variant, leads), seed).items())
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in self.p.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in self.p.items())
        self.bn_sums = torch.zeros(64, dtype=torch.float64)
        n = sum(v.numel() for v in self.p.values())
        self.grads = torch.zeros(n, dtype=torch.float64)
        self.step = 0
        self.running = O.new_bn_state(8, torch.float64)

    def forward_begin(self, x):
        self.x = x.double()
        self.leaf = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.p.items())
        a = F.leaky_relu(F.conv1d(self.x, self.leaf["conv1.0.weight"], self.leaf["conv1.0.bias"], padding=1), 0.2)
        self.a0 = a
        self.bn_sums[:8] = a.detach().sum((0, 2)); self.bn_sums[8:16] = (a.detach() ** 2).sum((0, 2))

    def forward_end(self, G):
        cnt = G * self.x.shape[2]
        mean = self.bn_sums[:8] / cnt
        var = self.bn_sums[8:16] / cnt - mean ** 2
        self.mean, self.rstd, self.cnt = mean, 1.0 / torch.sqrt(var + 1e-5), cnt
        # x0 as a LEAF: the global-statistics BatchNorm backward is applied by hand in backward_end
        xhat = (self.a0.detach() - mean[None, :, None]) * self.rstd[None, :, None]
        self.xhat = xhat
        self.x0 = (xhat * self.p["conv1.2.weight"][None, :, None] + self.p["conv1.2.bias"][None, :, None]).requires_grad_(True)
        p = dict(self.leaf)
        self.pred = _rest_of_network(p, self.x0, self.variant)
        return self.pred.detach()

    def loss(self, pred, target, G):
        n = target[0].numel()
        self.l = ((self.pred - target.double()) ** 2).sum() / (G * n)
        return self.l.detach().reshape(1), O.snr(target.double(), pred), O.rmse(target.double(), pred)

    def backward_begin(self):
        names = [k for k in self.leaf if not k.startswith("conv1.")]
        gs = torch.autograd.grad(self.l, [self.x0] + [self.leaf[k] for k in names], allow_unused=True)
        self.gx0 = gs[0]
        self.gdict = {k: (g if g is not None else torch.zeros_like(self.leaf[k])) for k, g in zip(names, gs[1:])}
        self.bn_sums[32:40] = self.gx0.sum((0, 2)); self.bn_sums[40:48] = (self.gx0 * self.xhat).sum((0, 2))
        # like the C ABI: the decoder half of the flat gradient buffer is final here (gradient bucket 1)
        o1 = self.grad_buckets()[1][0]
        self.grads[o1:] = torch.cat([self.gdict[k].reshape(-1) for k in self._decoder_keys()])

    def _decoder_keys(self):
        ks = list(self.p)
        return ks[next(i for i, k in enumerate(ks) if k.startswith("utransformer4.")):]

    def grad_buckets(self):
        o1 = sum(v.numel() for k, v in self.p.items() if k not in set(self._decoder_keys()))
        return [(0, o1), (o1, self.grads.numel() - o1)]

    def bucket_stream(self):
        return None          # CPU: no streams; the early all-reduce is still issued asynchronously

    def bucket_wait(self, k, stream):
        pass

    def backward_end(self, G):
        gam = self.p["conv1.2.weight"]
        s1, s2 = self.bn_sums[32:40] / self.cnt, self.bn_sums[40:48] / self.cnt
        da = gam[None, :, None] * self.rstd[None, :, None] * (self.gx0 - s1[None, :, None] - self.xhat * s2[None, :, None])
        gw, gb = torch.autograd.grad(self.a0, [self.leaf["conv1.0.weight"], self.leaf["conv1.0.bias"]], da)
        self.gdict["conv1.0.weight"], self.gdict["conv1.0.bias"] = gw, gb
        # affine grads use the ALL-REDUCED sums on every rank; the later gradient all-reduce (sum) would
        # count them world_size times, so each rank contributes its 1/world share
        w = dist.get_world_size() if dist.is_initialized() else 1
        self.gdict["conv1.2.weight"] = self.bn_sums[40:48].clone() / w
        self.gdict["conv1.2.bias"] = self.bn_sums[32:40].clone() / w
        dec = set(self._decoder_keys())     # bucket 0 only: bucket 1 may already have been all-reduced
        o1 = self.grad_buckets()[1][0]
        self.grads[:o1] = torch.cat([self.gdict[k].reshape(-1) for k in self.p if k not in dec])

    def replica_state(self):
        tensors = list(self.p.values()) + list(self.m.values()) + list(self.v.values()) + \
            [t for t in self.running.values() if torch.is_tensor(t)]

        def set_counters(c):
            self.step = int(c[0])
        return tensors, (lambda: [self.step]), set_counters

    def adam(self, lr):
        self.step += 1
        off = 0
        g = OrderedDict()
        for k, v in self.p.items():
            g[k] = self.grads[off:off + v.numel()].view_as(v); off += v.numel()
        O.adam_step(self.p, g, self.m, self.v, self.step, lr)


def _rest_of_network(p, x0, variant):
    """ralenet_forward from the BatchNorm output on (same code path as the oracle)."""
    le, rw, _ = O.variant_flags(variant)
    L = x0.shape[2]
    biases = [None] * 5
    if rw:
        for i, ln in enumerate(O.RW_LEN):
            biases[i + 1] = O.rwave_bias(p[f"rwattn{i+1}.relative_position_bias_table"], ln, L >> i)

    def stage(t, name, rwi):
        for i in range(2):
            t = O.transformer_block(t, p, O.block_prefix(variant, name, i), le, biases[rwi] if rwi else None)
        return t
    t = x0.permute(0, 2, 1)
    x1 = O.patch_merge(stage(t, "dtransformer1", 1), p, "pm1")
    x2 = O.patch_merge(stage(x1, "dtransformer2", 2), p, "pm2")
    x3 = O.patch_merge(stage(x2, "dtransformer3", 3), p, "pm3")
    x4 = O.patch_merge(stage(x3, "dtransformer34", 4), p, "pm4")
    xm = stage(x4, "transformer", 0) + x4
    d = O.patch_separate(stage(xm, "utransformer4", 0), p, "ps4") + x3
    d = O.patch_separate(stage(d, "utranformer3", 4), p, "ps3") + x2
    d = O.patch_separate(stage(d, "utransformer2", 3), p, "ps2") + x1
    d = O.patch_separate(stage(d, "utransformer1", 2), p, "ps1")
    d = d.permute(0, 2, 1) + x0
    return F.conv1d(d, p["transconv.0.weight"], p["transconv.0.bias"], padding=1)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from ecg_denoise_amd.dp import DataParallelTrainer
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 2, 256, generator=g); y = torch.randn(4, 2, 256, generator=g)
    # every rank draws DIFFERENT initial weights (what the model constructors do without a seed) and rank 1 pretends to
    # have taken steps already: the trainer must make rank 0's replica everybody's before the first step
    eng = OracleEngine("full", 2, 1234 + 1000 * rank)
    eng.step = 5 * rank
    tr = DataParallelTrainer(eng)
    chk = torch.stack([sum(v.sum() for v in eng.p.values()), torch.tensor(float(eng.step), dtype=torch.float64)])
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    assert all(torch.equal(b, both[0]) for b in both), "replicas differ after DataParallelTrainer.__init__"
    sh = slice(rank * 2, rank * 2 + 2)
    losses = [tr.train_step(x[sh], y[sh])["loss"].item() for _ in range(2)]
    if rank == 0:
        q.put((losses, {k: v.numpy() for k, v in eng.p.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_concatenated_batch():
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 2, 256, generator=g); y = torch.randn(4, 2, 256, generator=g)
    # single process reference: plain oracle train steps (fp64)
    p = OrderedDict((k, v.double()) for k, v in O.init_params(O.ralenet_param_shapes("full", 2), 1234).items())
    m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items()); v = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
    bn = O.new_bn_state(8, torch.float64)
    fwd = lambda pp, xx: O.ralenet_forward(pp, xx, "full", True, bn)
    ref_losses = [O.train_step(p, x.double(), y.double(), fwd, m, v, s)["loss"].item() for s in (1, 2)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [pr.start() for pr in procs]
    losses, params = q.get(timeout=600)
    [pr.join(60) for pr in procs]
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-9)
    for k in p:
        if k.endswith("to_kv.bias"):
            continue   # key-bias gradient is pure rounding noise (softmax shift invariance)
        np.testing.assert_allclose(params[k], p[k].numpy(), rtol=1e-6, atol=1e-9, err_msg=k)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 4: the 12-lead transfer-learning model (`newrale`) under data parallelism
# ---------------------------------------------------------------------------------------------------------------------
class OracleNewRALEEngine:
    """Same call surface as dp.NewRALEEngineAdapter, fp64 torch-CPU arithmetic: adapter convs with autograd, the FROZEN
    inner RA-LENet cut at its stem BatchNorm (which still normalises with batch statistics and updates its running
    ones: reference quirk A16, ralenet_12leads.py:694-696 + denoise_train.py:44)."""

    def __init__(self, seed_adapter, seed_inner, variant="full"):
        self.variant = variant
        self.pa = OrderedDict((k, v.double()) for k, v in O.init_params(O.newrale_param_shapes(), seed_adapter).items())
        self.p = OrderedDict((k, v.double()) for k, v in O.init_params(O.ralenet_param_shapes(variant, 2), seed_inner).items())
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in self.pa.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in self.pa.items())
        self.bn_sums = torch.zeros(64, dtype=torch.float64)
        self.grads = torch.zeros(sum(v.numel() for v in self.pa.values()), dtype=torch.float64)
        self.running = O.new_bn_state(8, torch.float64)
        self.tracked = torch.zeros(1, dtype=torch.float64)
        self.step = 0

    def forward_begin(self, x):
        self.leaf = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.pa.items())
        a = F.leaky_relu(F.conv1d(x.double(), self.leaf["conv1.weight"], self.leaf["conv1.bias"], padding=6), 0.01)
        a2 = F.leaky_relu(F.conv1d(a, self.leaf["conv2.weight"], self.leaf["conv2.bias"], padding=6), 0.01)
        self.L = x.shape[2]
        self.a0 = F.leaky_relu(F.conv1d(a2, self.p["conv1.0.weight"], self.p["conv1.0.bias"], padding=1), 0.2)
        self.bn_sums[:8] = self.a0.detach().sum((0, 2)); self.bn_sums[8:16] = (self.a0.detach() ** 2).sum((0, 2))

    def forward_end(self, G):
        cnt = G * self.L
        mean = self.bn_sums[:8] / cnt
        var = self.bn_sums[8:16] / cnt - mean ** 2
        self.rstd, self.cnt = 1.0 / torch.sqrt(var + 1e-5), cnt
        self.running["running_mean"].mul_(0.9).add_(0.1 * mean)
        self.running["running_var"].mul_(0.9).add_(0.1 * var * cnt / (cnt - 1))
        self.tracked += 1
        self.xhat = (self.a0.detach() - mean[None, :, None]) * self.rstd[None, :, None]
        self.x0 = (self.xhat * self.p["conv1.2.weight"][None, :, None] + self.p["conv1.2.bias"][None, :, None]).requires_grad_(True)
        r = _rest_of_network(dict(self.p), self.x0, self.variant)
        a3 = F.leaky_relu(F.conv1d(r, self.leaf["conv3.weight"], self.leaf["conv3.bias"], padding=6), 0.01)
        self.pred = F.conv1d(a3, self.leaf["conv4.weight"], self.leaf["conv4.bias"], padding=6)
        return self.pred.detach()

    def loss(self, pred, target, G):
        self.l = ((self.pred - target.double()) ** 2).sum() / (G * target[0].numel())
        return self.l.detach().reshape(1), O.snr(target.double(), pred), O.rmse(target.double(), pred)

    def backward_begin(self):
        late = ["conv3.weight", "conv3.bias", "conv4.weight", "conv4.bias"]
        gs = torch.autograd.grad(self.l, [self.x0] + [self.leaf[k] for k in late])
        self.gx0, self.gd = gs[0], dict(zip(late, gs[1:]))
        self.bn_sums[32:40] = self.gx0.sum((0, 2)); self.bn_sums[40:48] = (self.gx0 * self.xhat).sum((0, 2))

    def backward_end(self, G):
        gam = self.p["conv1.2.weight"]
        s1, s2 = self.bn_sums[32:40] / self.cnt, self.bn_sums[40:48] / self.cnt
        da = gam[None, :, None] * self.rstd[None, :, None] * (self.gx0 - s1[None, :, None] - self.xhat * s2[None, :, None])
        early = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias"]
        self.gd.update(zip(early, torch.autograd.grad(self.a0, [self.leaf[k] for k in early], da)))
        self.grads[:] = torch.cat([self.gd[k].reshape(-1) for k in self.pa])

    def adam(self, lr):
        self.step += 1
        off, g = 0, OrderedDict()
        for k, v in self.pa.items():
            g[k] = self.grads[off:off + v.numel()].view_as(v); off += v.numel()
        O.adam_step(self.pa, g, self.m, self.v, self.step, lr)

    def replica_state(self):
        tensors = list(self.pa.values()) + list(self.m.values()) + list(self.v.values()) + list(self.p.values()) + \
            [self.running["running_mean"], self.running["running_var"], self.tracked]

        def set_counters(c):
            self.step = int(c[0])
        return tensors, (lambda: [self.step]), set_counters


def _worker_newrale(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from ecg_denoise_amd.dp import DataParallelTrainer
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 12, 256, generator=g); y = torch.randn(4, 12, 256, generator=g)
    eng = OracleNewRALEEngine(77 + 10 * rank, 1234 + 1000 * rank)      # ranks start from different adapters AND inner models
    tr = DataParallelTrainer(eng)
    sh = slice(rank * 2, rank * 2 + 2)
    losses = [tr.train_step(x[sh], y[sh])["loss"].item() for _ in range(2)]
    if rank == 0:
        q.put((losses, {k: v.numpy() for k, v in eng.pa.items()}, {k: v.numpy() for k, v in eng.running.items() if torch.is_tensor(v)},
               float(eng.tracked)))
    dist.barrier()
    dist.destroy_process_group()


def test_newrale_two_ranks_equal_one_process_on_the_concatenated_batch():
    """BASELINE config 4's split: two ranks x 2 twelve-lead windows == one process on the 4 windows, including the inner
    model's BatchNorm running statistics (the frozen inner model still trains those)."""
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 12, 256, generator=g).double(); y = torch.randn(4, 12, 256, generator=g).double()
    pa = OrderedDict((k, v.double()) for k, v in O.init_params(O.newrale_param_shapes(), 77).items())
    p = OrderedDict((k, v.double()) for k, v in O.init_params(O.ralenet_param_shapes("full", 2), 1234).items())
    m = OrderedDict((k, torch.zeros_like(v)) for k, v in pa.items()); v = OrderedDict((k, torch.zeros_like(t)) for k, t in pa.items())
    bn = O.new_bn_state(8, torch.float64)
    fwd = lambda pp, xx: O.newrale_forward(pp, p, xx, "full", True, bn)
    ref_losses = [O.train_step(pa, x, y, fwd, m, v, s)["loss"].item() for s in (1, 2)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_newrale, args=(r, 2, port, q)) for r in range(2)]
    [pr.start() for pr in procs]
    losses, params, running, tracked = q.get(timeout=600)
    [pr.join(60) for pr in procs]
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-9)
    for k in pa:
        np.testing.assert_allclose(params[k], pa[k].numpy(), rtol=1e-6, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(running["running_mean"], bn["running_mean"].numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(running["running_var"], bn["running_var"].numpy(), rtol=1e-9)
    assert tracked == bn["num_batches_tracked"] == 2


# ---------------------------------------------------------------------------------------------------------------------
# Collectives per step, the logging interval, and U-Net with per-rank BatchNorm statistics
# ---------------------------------------------------------------------------------------------------------------------
class _StubEngine:
    """call surface of dp.HipEngineAdapter (gradient buckets) / dp.NewRALEEngineAdapter (buckets=False); no arithmetic"""

    def __init__(self, buckets):
        self.bn_sums = torch.zeros(64, dtype=torch.float64)
        self.grads = torch.zeros(100)
        if buckets:
            self.grad_buckets = lambda: [(0, 60), (60, 40)]
            self.bucket_stream = lambda: None
            self.bucket_wait = lambda k, s: None

    def forward_begin(self, x): self.x = x
    def forward_end(self, G): return self.x
    def loss(self, pred, target, G): return torch.full((1,), 1.0 / G * pred.shape[0]), torch.ones(pred.shape[0]), torch.ones(pred.shape[0])
    def backward_begin(self): pass
    def backward_end(self, G): pass
    def adam(self, lr): pass


class _StubLayerEngine:
    """call surface of dp.UNetEngineAdapter: one reduction point per BatchNorm layer, or the fused local step"""

    def __init__(self):
        self.bn_sums = torch.zeros(128 * 10, dtype=torch.float64)
        self.grads = torch.zeros(100)
        self.local_calls = 0

    def forward_iter(self, x, G):
        for bn in range(10):
            yield self.bn_sums[128 * bn:128 * bn + 64]
        self.pred = x

    def backward_iter(self, G):
        for bn in range(9, -1, -1):
            yield self.bn_sums[128 * bn + 64:128 * bn + 128]

    def forward_local(self, x):
        self.local_calls += 1
        self.pred = x
        return x

    def backward_local(self): self.local_calls += 1
    def loss(self, pred, target, G): return torch.full((1,), 1.0 / G * pred.shape[0]), torch.ones(pred.shape[0]), torch.ones(pred.shape[0])
    def adam(self, lr): pass


def _worker_counts(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from ecg_denoise_amd import dp
    n = [0]
    real = dist.all_reduce

    def counting(*a, **k):
        n[0] += 1
        return real(*a, **k)
    dp.dist.all_reduce = counting
    x = torch.zeros(2, 2, 16)
    out = {}

    def per_step(tr, steps):
        res = []
        for _ in range(steps):
            n[0] = 0
            tr.train_step(x, x)
            res.append(n[0])
        return res
    out["ralenet_sync"] = per_step(dp.DataParallelTrainer(_StubEngine(True), sync_state=False, log_every=4), 4)
    out["ralenet_local_bn"] = per_step(dp.DataParallelTrainer(_StubEngine(True), sync_bn=False, sync_state=False, log_every=4), 4)
    out["newrale_sync"] = per_step(dp.DataParallelTrainer(_StubEngine(False), sync_state=False, log_every=4), 4)
    out["ralenet_log1"] = per_step(dp.DataParallelTrainer(_StubEngine(True), sync_state=False), 2)
    eng = _StubLayerEngine()
    out["unet_sync"] = per_step(dp.DataParallelTrainer(eng, sync_state=False, log_every=4), 4)
    assert eng.local_calls == 0
    eng = _StubLayerEngine()
    tr = dp.DataParallelTrainer(eng, sync_bn=False, sync_state=False, log_every=4)
    out["unet_local_bn"] = per_step(tr, 4)
    assert eng.local_calls == 8                      # sync_bn=False is honoured: the fused local step, no per-layer cut
    m = tr.metrics()
    out["metrics"] = (float(m["loss"]), float(m["snr"]), int(m["steps"]))
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_collectives_per_step_and_logging_interval():
    """RA-LENet: 2 BatchNorm reductions + 2 gradient buckets per step (2 without sync-BN), newrale 2 + 1, U-Net 10 + 10 + 1
    with global statistics and ONE with per-rank statistics; the metrics reduction is issued once per logging interval."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_counts, args=(r, 2, port, q)) for r in range(2)]
    [pr.start() for pr in procs]
    out = q.get(timeout=300)
    [pr.join(60) for pr in procs]
    assert out["ralenet_sync"] == [4, 4, 4, 5]        # the 4th step closes the logging interval
    assert out["ralenet_local_bn"] == [2, 2, 2, 3]
    assert out["newrale_sync"] == [3, 3, 3, 4]
    assert out["ralenet_log1"] == [5, 5]
    assert out["unet_sync"] == [21, 21, 21, 22]
    assert out["unet_local_bn"] == [1, 1, 1, 2]
    assert max(out["unet_local_bn"]) <= 12 and max(out["ralenet_sync"]) <= 12
    loss, snr, steps = out["metrics"]
    assert steps == 4 and abs(loss - 1.0) < 1e-12 and abs(snr - 1.0) < 1e-12     # global means over the interval


class OracleUNetEngine:
    """dp.UNetEngineAdapter's per-rank-statistics surface in fp64 torch-CPU arithmetic (autograd backward)"""

    def __init__(self, seed):
        self.p = OrderedDict((k, v.double()) for k, v in O.init_params(O.unet_param_shapes(2), seed).items())
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in self.p.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in self.p.items())
        self.bn = O.unet_bn_state(self.p, torch.float64)
        self.grads = torch.zeros(sum(v.numel() for v in self.p.values()), dtype=torch.float64)
        self.step = 0

    def forward_iter(self, x, G):
        raise NotImplementedError("global-batch statistics are covered on the GPU (tests/test_gpu_dp.py)")

    def forward_local(self, x):
        self.leaf = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.p.items())
        self.pred = O.unet_forward(self.leaf, x.double(), True, self.bn)
        return self.pred.detach()

    def loss(self, pred, target, G):
        self.l = ((self.pred - target.double()) ** 2).sum() / (G * target[0].numel())
        return self.l.detach().reshape(1), O.snr(target.double(), pred), O.rmse(target.double(), pred)

    def backward_local(self):
        gs = torch.autograd.grad(self.l, list(self.leaf.values()), allow_unused=True)
        self.grads[:] = torch.cat([(g if g is not None else torch.zeros_like(v)).reshape(-1) for g, v in zip(gs, self.p.values())])

    def adam(self, lr):
        self.step += 1
        off, g = 0, OrderedDict()
        for k, v in self.p.items():
            g[k] = self.grads[off:off + v.numel()].view_as(v); off += v.numel()
        O.adam_step(self.p, g, self.m, self.v, self.step, lr)


def _worker_unet_local(rank, world, port, q, nwin):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from ecg_denoise_amd.dp import DataParallelTrainer
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2 * nwin, 2, 256, generator=g); y = torch.randn(2 * nwin, 2, 256, generator=g)
    eng = OracleUNetEngine(321)
    tr = DataParallelTrainer(eng, sync_bn=False, sync_state=False)
    sh = slice(rank * nwin, (rank + 1) * nwin)
    out = tr.train_step(x[sh], y[sh])
    if rank == 0:
        q.put((out["loss"].item(), eng.grads.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_unet_per_rank_batchnorm_statistics_stay_within_1e_3_of_the_global_batch_step():
    """`sync_bn=False`: every rank normalises with its own 64 windows ("per-rank stats with a documented <= 1e-3
    deviation").  Measured against the single-process step on the 128 windows: the loss moves by well under 1e-3
    relative and the gradient by ~1e-2 of its norm - the price of the 21 -> 1 collectives per step; the exact mode stays
    the default."""
    torch.set_num_threads(1)
    nwin = 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2 * nwin, 2, 256, generator=g).double(); y = torch.randn(2 * nwin, 2, 256, generator=g).double()
    p = OrderedDict((k, v.double()) for k, v in O.init_params(O.unet_param_shapes(2), 321).items())
    m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items()); v = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
    bn = O.unet_bn_state(p, torch.float64)
    ref = O.train_step(p, x, y, lambda pp, xx: O.unet_forward(pp, xx, True, bn), m, v, 1)
    gref = torch.cat([ref["grads"][k].reshape(-1) for k in p]).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_unet_local, args=(r, 2, port, q, nwin)) for r in range(2)]
    [pr.start() for pr in procs]
    loss, grads = q.get(timeout=300)
    [pr.join(60) for pr in procs]
    dl = abs(loss - ref["loss"].item()) / ref["loss"].item()
    dg = np.linalg.norm(grads - gref) / np.linalg.norm(gref)
    print(f"per-rank statistics vs global batch: loss {dl:.2e} relative, gradient {dg:.2e} of its norm")
    assert dl < 1e-3, dl
    assert dg < 5e-2, dg
    assert dl > 0.0          # (it IS a different step: equality would mean the statistics were shared after all)
