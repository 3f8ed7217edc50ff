"""Data-parallel training over the GPUs of one node: one process per GPU, windows of the
global batch sharded evenly, parameters / Adam state replicated.

Collectives per step (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm):
  1. BatchNorm forward sums   (16 doubles)   -> exact global-batch statistics (sync-BN)
  2. BatchNorm backward sums  (16 doubles)
  3. flat gradient buffer     (4.35 MB fp32) in two buckets: the decoder half (utransformer4 .. transconv) is
     final about half-way through the backward pass and is all-reduced on a side stream WHILE the bottleneck and
     encoder backward kernels run (`ral_grad_bucket` / `ral_grad_bucket_wait`); the other half follows the stem
  4. metrics (loss, SNR, RMSE: three doubles) once per LOGGING INTERVAL (`log_every` steps), on the communication
     stream: the compute stream never waits for it (`DataParallelTrainer.metrics()` reads the result)
That is 4 collectives per step with sync-BN (2 with `sync_bn=False`: per-rank statistics, the PyTorch-DDP default), plus
one per logging interval.  No collective touches activations; inference needs none (replicas).
`NewRALEEngineAdapter` runs the 12-lead transfer-learning model the same way: the frozen inner RA-LENet is cut at the same
two reduction points (its BatchNorm still uses batch statistics), only the 8.8 KB adapter gradient is all-reduced.
U-Net (a BatchNorm after every conv): with `sync_bn=True` `UNetEngineAdapter` cuts the step at every layer, 10 + 10
reductions of 64 doubles per step, for exact global-batch statistics.  They cannot be packed: layer l + 1 normalises with
the COMPLETE statistics of layer l, and the backward sums of layer l are formed from the gradient that left layer l + 1 -
each reduction is a true dependency of the next stage, so a 0.5 ms step pays ~20 latency-bound collectives.  With
`sync_bn=False` every rank normalises with its own shard's statistics (what `torch.nn.parallel.DistributedDataParallel`
does without SyncBatchNorm) and the step is the fused single-GPU forward / backward plus ONE gradient all-reduce;
tests/test_dp_gloo.py measures how far that is from the global-batch step (1e-3 class at 64 windows per rank).

The trainer only sequences engine calls and collectives, so it is testable on CPU with
`gloo` and any engine exposing the same methods (tests/test_dp_gloo.py).
"""
import torch
import torch.distributed as dist


class HipEngineAdapter:
    """Splits RALENet's train step at the two BatchNorm reduction points."""

    def __init__(self, model):
        import ctypes as C
        from . import _lib
        from .model import _ptr, _stream
        self.m, self._lib, self._ptr, self._stream, self._C = model, _lib, _ptr, _stream, C
        self.bn_sums = model.eng.bn_sums
        self.grads = model.eng.grads

    def forward_begin(self, x):
        self.x = self.m._check_x(x)
        self._lib.check(self._lib.lib().ral_forward_begin(self.m.eng.h, self._ptr(self.x), x.shape[0], self._stream()))

    def forward_end(self, global_windows):
        self.y = torch.empty_like(self.x)
        self._lib.check(self._lib.lib().ral_forward_end(self.m.eng.h, self._ptr(self.y), self.x.shape[0],
                                                        global_windows, self._stream()))
        for k in self.m.eng.counters:
            self.m.eng.counters[k] += 1
        return self.y

    def loss(self, pred, target, global_windows):
        return self.m.loss_and_metrics(pred, target, True, global_windows)   # loss already / global_windows

    def backward_begin(self):
        dy = self.m._dy
        self._lib.check(self._lib.lib().ral_backward_begin(self.m.eng.h, self._ptr(dy), dy.shape[0], self._stream()))

    def backward_end(self, global_windows):
        self._lib.check(self._lib.lib().ral_backward_end(self.m.eng.h, self._C.c_void_p(0), self.x.shape[0],
                                                         global_windows, self._stream()))

    def adam(self, lr):
        self.m.step(lr)

    def replica_state(self):
        return _replica_state(self.m)

    # ---- gradient buckets (overlap of the all-reduce with the backward pass) ----
    def grad_buckets(self):
        """[(offset, count)] of bucket 0 (final after backward_end) and bucket 1 (final inside backward_begin)"""
        C = self._C
        out = []
        for k in (0, 1):
            off, cnt = C.c_int64(), C.c_int64()
            self._lib.check(self._lib.lib().ral_grad_bucket(self.m.eng.h, k, C.byref(off), C.byref(cnt)))
            out.append((off.value, cnt.value))
        return out

    def bucket_stream(self):
        if getattr(self, "_comm", None) is None:
            self._comm = torch.cuda.Stream(device=self.m.eng.device)
            from . import hw_queues_set_too_late
            if hw_queues_set_too_late():
                import warnings
                warnings.warn("ecg_denoise_amd was imported after the GPU was initialised and GPU_MAX_HW_QUEUES is not set: "
                              "the early gradient bucket's all-reduce will run AFTER the backward pass, not under it. "
                              "Export GPU_MAX_HW_QUEUES=8 (or import ecg_denoise_amd before the first GPU call).", RuntimeWarning)
        return self._comm

    def bucket_wait(self, k, stream):
        """make `stream` wait until bucket k holds its final gradients (call after the engine call producing it)"""
        self._lib.check(self._lib.lib().ral_grad_bucket_wait(self.m.eng.h, k, stream.cuda_stream))


class UNetEngineAdapter:
    """U-Net under data parallelism with exact global-batch BatchNorm: the model has a BatchNorm after every conv
    (UNet.py:46-141), so the step is cut at every layer.  `forward_iter` / `backward_iter` run one stage per
    iteration and yield the slice of `bn_sums` (64 doubles) that must be summed over the ranks before the next
    stage may run; the trainer all-reduces whatever is yielded."""

    def __init__(self, model):
        import ctypes as C
        from . import _lib
        from .model import _ptr, _stream
        self.m, self._lib, self._ptr, self._stream, self._C = model, _lib, _ptr, _stream, C
        self.bn_sums = model.eng.bn_sums
        self.grads = model.eng.grads
        self.pred = None

    def forward_iter(self, x, global_windows):
        L, h = self._lib.lib(), self.m.eng.h
        self.x = self.m._check_x(x)
        B = x.shape[0]
        for si in range(11):
            self._lib.check(L.ral_unet_forward_stage(h, self._ptr(self.x), B, 1, si, global_windows, self._stream()))
            bn = L.ral_unet_stage_bn(si)
            if bn >= 0:
                yield self.bn_sums[128 * bn:128 * bn + 64]
        self.pred = torch.empty_like(self.x)
        self._lib.check(L.ral_unet_forward_finish(h, self._ptr(self.pred), B, 1, global_windows, self._stream()))
        for k in self.m.eng.counters:
            self.m.eng.counters[k] += 1

    def loss(self, pred, target, global_windows):
        return self.m.loss_and_metrics(pred, target, True, global_windows)

    def backward_iter(self, global_windows):
        L, h = self._lib.lib(), self.m.eng.h
        dy = self.m._dy
        B = dy.shape[0]
        self._lib.check(L.ral_unet_backward_start(h, self._ptr(dy), B, global_windows, self._stream()))
        for si in range(10, -1, -1):
            bn = L.ral_unet_stage_bn(si)
            if bn >= 0:       # the BatchNorm-backward sums of this stage's output, accumulated by its consumers
                yield self.bn_sums[128 * bn + 64:128 * bn + 128]
            self._lib.check(L.ral_unet_backward_stage(h, B, si, global_windows, self._stream()))
        self._lib.check(L.ral_unet_backward_finish(h, B, global_windows, self._stream()))

    # ---- per-rank BatchNorm statistics: the fused single-GPU entry points, no reduction point inside the step
    def forward_local(self, x):
        self.pred = self.m.forward(x)
        return self.pred

    def backward_local(self):
        self.m.backward()

    def adam(self, lr):
        self.m.step(lr)

    def replica_state(self):
        return _replica_state(self.m)


class NewRALEEngineAdapter:
    """The 12-lead transfer-learning model (`newrale`, ralenet_12leads.py:680-709; caller Transfer_learning.py:71-82)
    under data parallelism - BASELINE config 4.  Only the 2 210 adapter parameters train, but the frozen inner RA-LENet
    is left in train mode by `model.train()` (denoise_train.py:44), so its stem BatchNorm normalises with BATCH
    statistics and keeps updating its running ones (quirk A16).  For the N-rank step to be the single-process step on
    the concatenated batch the inner model is therefore cut exactly like a trainable one:
      forward : conv1, conv2, inner stem conv   -> all-reduce of 16 doubles (sum x, sum x^2)  -> rest, conv3, conv4
      backward: conv4, conv3, inner data-gradient chain (`ral_backward_input_begin`: no weight-gradient kernels)
                -> all-reduce of 16 doubles (sum dy, sum dy x^) -> stem input gradient, conv2, conv1
      then ONE all-reduce of the 8.8 KB adapter gradient and the flat Adam kernel on the adapter buffer.
    Replica state = adapter parameters and moments AND the inner model's parameters and running statistics."""

    def __init__(self, model):
        import ctypes as C
        from . import _lib
        from .model import _ptr, _stream
        self.m, self._lib, self._ptr, self._stream, self._C = model, _lib, _ptr, _stream, C
        self.inner = model.rale
        self.bn_sums = self.inner.eng.bn_sums
        self.grads = model.grads

    def forward_begin(self, x):
        self.a2 = self.m._forward_pre(x)
        self.inner._x = self.a2
        self._lib.check(self._lib.lib().ral_forward_begin(self.inner.eng.h, self._ptr(self.a2), self.a2.shape[0],
                                                          self._stream()))

    def forward_end(self, global_windows):
        r = torch.empty_like(self.a2)
        self._lib.check(self._lib.lib().ral_forward_end(self.inner.eng.h, self._ptr(r), self.a2.shape[0], global_windows,
                                                        self._stream()))
        for k in self.inner.eng.counters:
            self.inner.eng.counters[k] += 1
        return self.m._forward_post(r)

    def loss(self, pred, target, global_windows):
        return self.m.loss_and_metrics(pred, target, True, global_windows)

    def backward_begin(self):
        self.dr = self.m._backward_pre()
        self._lib.check(self._lib.lib().ral_backward_input_begin(self.inner.eng.h, self._ptr(self.dr), self.dr.shape[0],
                                                                 self._stream()))

    def backward_end(self, global_windows):
        d2 = torch.empty_like(self.dr)
        self._lib.check(self._lib.lib().ral_backward_input_end(self.inner.eng.h, self._ptr(d2), self.dr.shape[0],
                                                               global_windows, self._stream()))
        self.m._backward_post(d2)

    def adam(self, lr):
        self.m.step(lr)

    def replica_state(self):
        m, ie = self.m, self.inner.eng
        tensors = [m.params, m.adam_m, m.adam_v, ie.params, ie.state]

        def get_counters():
            return [m.step_count] + [ie.counters[k] for k in sorted(ie.counters)]

        def set_counters(v):
            m.step_count = int(v[0])
            for k, c in zip(sorted(ie.counters), v[1:]):
                ie.counters[k] = int(c)
            self.inner._params_changed()      # (see _replica_state)
        return tensors, get_counters, set_counters


def _replica_state(model):
    """everything a replica owns that must be identical on every rank before the first step: flat device tensors
    (broadcast in place) and a getter / setter for the host-side counters"""
    e = model.eng
    tensors = [e.params, e.state, e.adam_m, e.adam_v]

    def get_counters():
        return [model.step_count] + [e.counters[k] for k in sorted(e.counters)]

    def set_counters(v):
        model.step_count = int(v[0])
        for k, c in zip(sorted(e.counters), v[1:]):
            e.counters[k] = int(c)
        if hasattr(model, "_params_changed"):
            model._params_changed()           # (the broadcast rewrote the parameter buffer: cached weight planes of an eval-mode model are stale)
    return tensors, get_counters, set_counters


class DataParallelTrainer:
    """`sync_state=True` (default) makes rank 0's replica the replica of every rank when the trainer is built:
    parameters, BatchNorm running statistics, Adam moments, the step count and `num_batches_tracked` are broadcast
    (the model constructors draw their weights from OS entropy unless given a seed, and only gradients are ever
    all-reduced, so replicas that start apart stay apart).  Call `sync_state()` again after loading a checkpoint on
    one rank.

    `sync_bn`: True = BatchNorm statistics of the global batch (the N-rank step IS the single-process step on the
    concatenated batch); False = every rank's own statistics (no BatchNorm collective).  With False the BatchNorm RUNNING
    statistics of the ranks drift apart (DistributedDataParallel would re-broadcast rank 0's buffers every forward; this
    trainer does not): call `sync_state()` before evaluating or writing a checkpoint, otherwise both depend on the rank.
    The gradient of a step then differs from the global-batch gradient by ~1e-2 of its norm (tests/test_dp_gloo.py).
    `log_every`: the loss / SNR / RMSE means of the last `log_every` steps are all-reduced once per interval on the
    communication stream; `metrics()` returns them.  With `log_every == 1` (default) `train_step` also waits for the
    reduction and returns the global loss, as the reference loop prints it every step (denoise_train.py:54-64); with a
    longer interval `train_step` returns this rank's own mean (`loss_is_global` False) and never waits."""

    def __init__(self, engine, group=None, sync_bn=True, sync_state=True, log_every=1, force_collectives=False):
        self.e, self.group, self.sync_bn, self.log_every = engine, group, sync_bn, max(1, int(log_every))
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # `force_collectives`: issue every collective, the bucket stream and its events although the group has ONE rank (an
        # all-reduce over one rank is the identity): the way a box with one GPU runs the real RCCL path (tests/test_gpu_dp_procs.py)
        self._coll = self.world > 1 or (bool(force_collectives) and dist.is_initialized())
        self._hist, self._pending, self._last = [], None, None
        # collectives issued by the last train_step: data path (BatchNorm sums, gradient buckets) / metric reductions
        self._ncoll, self._nmet, self.collectives_last_step, self.metric_collectives_last_step = 0, 0, None, None
        if sync_state:
            self.sync_state()

    def sync_state(self, src=0):
        if self.world <= 1 or not hasattr(self.e, "replica_state"):
            return
        tensors, get_counters, set_counters = self.e.replica_state()
        # `src` is a rank of `group` (group_src), so sub-groups work without translating to global ranks
        for t in tensors:
            if t is not None:
                dist.broadcast(t, group=self.group, group_src=src)
        c = torch.tensor(get_counters(), dtype=torch.int64, device=tensors[0].device)
        dist.broadcast(c, group=self.group, group_src=src)
        set_counters(c.tolist())

    def _allreduce(self, t):
        if self._coll:
            self._ncoll += 1
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _comm_stream(self):
        return self.e.bucket_stream() if hasattr(self.e, "bucket_stream") else None

    # ---- metrics: one reduction per logging interval, off the compute stream -------------------------------------
    def _log(self, loss, snr, rmse, G):
        """loss: this rank's share of the global mean ((sse / n) / G); snr, rmse: per local window"""
        # the HIP engines' loss kernel leaves {loss share, sum SNR / G, sum RMSE / G} in one three-double tensor
        # (ral_loss_means; `loss` is a view of its first entry): nothing is reduced here then, and with log_every == 1 the
        # step has no metric kernel at all
        means = getattr(getattr(self.e, "m", None), "_means", None)
        if means is not None and means.data_ptr() == loss.data_ptr():
            self._hist.append(means)
            if len(self._hist) < self.log_every:
                return None
            k = len(self._hist)
            if k > 1:
                buf = torch.stack(self._hist).sum(0) / k
            else:     # (a copy where the all-reduce below would otherwise overwrite the tensor `loss` is a view of)
                buf = self._hist[0].clone() if self._coll else self._hist[0]
        else:
            self._hist.append((loss, snr, rmse))
            if len(self._hist) < self.log_every:
                return None
            k = len(self._hist)
            buf = torch.stack([torch.stack([l.reshape(()).double() for l, _, _ in self._hist]).sum(),
                               torch.stack([s.double().sum() for _, s, _ in self._hist]).sum() / G,
                               torch.stack([r.double().sum() for _, _, r in self._hist]).sum() / G]) / k
        self._hist = []
        work = None
        if self._coll:
            self._nmet += 1
            if self._pending is not None and self._pending[1] is not None:
                self._pending[1].wait()          # a reduction nobody asked for (metrics() not called): finish it before its buffer goes
            comm = self._comm_stream()
            if comm is not None:
                comm.wait_stream(torch.cuda.current_stream(buf.device))
                if buf.is_cuda:
                    buf.record_stream(comm)      # allocated on the compute stream, used on the communication stream
                with torch.cuda.stream(comm):
                    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending = (buf, work, k)
        return self._pending

    def metrics(self):
        """global means over the last completed logging interval: {"loss", "snr", "rmse", "steps"} (None before the first)"""
        if self._pending is not None:
            buf, work, k = self._pending
            if work is not None:
                work.wait()
            self._last = {"loss": buf[0], "snr": buf[1], "rmse": buf[2], "steps": k}
            self._pending = None
        return self._last

    def _finish(self, loss, snr, rmse, pred, G):
        logged = self._log(loss, snr, rmse, G)
        self.collectives_last_step, self.metric_collectives_last_step, self._ncoll, self._nmet = self._ncoll, self._nmet, 0, 0
        if self.log_every == 1 and logged is not None:
            return {"loss": self.metrics()["loss"].reshape(1), "loss_is_global": True, "snr": snr, "rmse": rmse, "pred": pred}
        return {"loss": loss * self.world, "loss_is_global": self.world == 1, "snr": snr, "rmse": rmse, "pred": pred}

    def train_step(self, x_local, target_local, lr=1e-3):
        """x_local: this rank's shard (equal shard sizes).  Returns loss (see `log_every`), snr, rmse (local)."""
        B = x_local.shape[0]
        G = B * self.world
        e = self.e
        if hasattr(e, "forward_iter"):     # an engine with one reduction point per layer (U-Net)
            if self.sync_bn:
                for t in e.forward_iter(x_local, G):
                    self._allreduce(t)
                pred = e.pred
                loss, snr, rmse = e.loss(pred, target_local, G)
                for t in e.backward_iter(G):
                    self._allreduce(t)
            else:                          # per-rank statistics: the fused step, no reduction point inside it
                pred = e.forward_local(x_local)
                loss, snr, rmse = e.loss(pred, target_local, G)
                e.backward_local()
            self._allreduce(e.grads)       # dy carries 1 / G: the sum of the local gradients is the global-mean gradient
            e.adam(lr)
            return self._finish(loss, snr, rmse, pred, G)
        e.forward_begin(x_local)
        if self.sync_bn:
            self._allreduce(e.bn_sums[:32])
        pred = e.forward_end(G if self.sync_bn else B)
        loss, snr, rmse = e.loss(pred, target_local, G)
        e.backward_begin()
        early = None
        if self._coll and hasattr(e, "grad_buckets"):
            # decoder half of the gradients: all-reduced on the engine's communication stream, which only waits
            # for the kernels that write it, so the collective runs under the rest of the backward pass
            (o0, n0), (o1, n1) = e.grad_buckets()
            comm = e.bucket_stream()
            e.bucket_wait(1, comm)
            self._ncoll += 1
            if comm is not None:
                with torch.cuda.stream(comm):
                    early = dist.all_reduce(e.grads[o1:o1 + n1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                early = dist.all_reduce(e.grads[o1:o1 + n1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.sync_bn:
            self._allreduce(e.bn_sums[32:64])
        e.backward_end(G if self.sync_bn else B)
        if early is not None:
            self._allreduce(e.grads[o0:o0 + n0])
            early.wait()                  # the current stream waits for the early bucket
        else:
            self._allreduce(e.grads)      # dy already carries 1/G: the sum IS the global-mean gradient
        e.adam(lr)
        return self._finish(loss, snr, rmse, pred, G)
