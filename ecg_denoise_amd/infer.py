"""Inference path: hipGraph-captured forward + streaming of long records (SURVEY §8f rank 3;
BASELINE config 4: 30-minute records, batches of 4096 windows, one MI355X).

The reference only has non-overlapping fixed-length chunking of the 650 000-sample MIT-BIH records
(local_utils/local_utils.py:116-130, 256-sample chunks, z-score per chunk group).  Here a record is cut
into windows of the model's length with an optional overlap; every window is z-scored per lead
(np_norm, local_utils/local_utils.py:261-266), denoised in eval mode (BatchNorm running statistics),
de-normalised and stitched back (overlapping regions keep the centre of each window)."""
import torch

from . import _lib
from .model import _ptr, _stream


class GraphedForward:
    """model(x) for a fixed batch size captured once into a hipGraph (torch.cuda.CUDAGraph drives the
    capture; all kernels inside are libralenet launches on the capture stream and its forked lanes)."""

    def __init__(self, model, batch):
        e = model.eng
        if batch > e.max_batch:
            raise _lib.RalError(f"batch {batch} > max_batch {e.max_batch}")
        self.model, self.batch = model, batch
        model.eval()
        self.x = torch.zeros(batch, e.leads, e.L, dtype=torch.float32, device=e.device)
        side = torch.cuda.Stream(device=e.device)
        side.wait_stream(torch.cuda.current_stream(e.device))
        with torch.cuda.stream(side):          # warm-up outside capture (lazy LDS-size attributes, lanes)
            for _ in range(2):
                self.y = model(self.x)
        torch.cuda.current_stream(e.device).wait_stream(side)
        torch.cuda.synchronize(e.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.y = model(self.x)

    def __call__(self, x):
        self.x.copy_(x)
        self.graph.replay()
        return self.y


class StreamingDenoiser:
    """Long records through the eval-mode forward, everything on the device: `ral_stream_windows` (window gather +
    per-window z-score, statistics kept in a side buffer), the model in batches of `batch` windows, `ral_stream_stitch`
    (de-normalise + keep-the-centre stitching).  With `use_graph` the whole pipeline of a record group of a given
    shape (R, leads, T) is ONE hipGraph: replaying it costs one launch per group."""

    def __init__(self, model, batch=4096, overlap=0, use_graph=True, max_plans=4):
        self.model, self.L, self.leads = model, model.eng.L, model.eng.leads
        self.max_plans = max(1, int(max_plans))      # plans (buffers + hipGraph per record-group shape) kept, LRU
        self.batch = min(batch, model.eng.max_batch)
        if overlap < 0 or overlap >= self.L or overlap % 2:
            raise _lib.RalError("overlap must be an even number of samples in [0, L)")
        self.overlap, self.hop = overlap, self.L - overlap
        self.use_graph = use_graph
        self.plans = {}
        model.eval()

    def windows_per_record(self, T):
        if T < self.L:
            raise _lib.RalError(f"record shorter than one window ({T} < {self.L})")
        n_reg = (T - self.L) // self.hop + 1
        return n_reg + (1 if (T - self.L) % self.hop else 0)

    def window_starts(self, T):
        n_reg = (T - self.L) // self.hop + 1
        st = [k * self.hop for k in range(n_reg)]
        return st + ([T - self.L] if (T - self.L) % self.hop else [])

    def _run(self, p):
        """enqueue the pipeline of one record group on the current stream (captured or eager)"""
        lib, e = _lib.lib(), self.model.eng
        R, T, nw_all = p["R"], p["T"], p["nw"]
        for w0 in range(0, nw_all, self.batch):
            nw = min(self.batch, nw_all - w0)
            _lib.check(lib.ral_stream_windows(_ptr(p["rec"]), R, T, self.leads, self.L, self.hop, w0, nw, _ptr(p["win"]),
                                              _ptr(p["stats"]), _stream()))
            _lib.check(lib.ral_forward(e.h, _ptr(p["win"]), _ptr(p["y"][w0:]), nw, 0, _stream()))
        _lib.check(lib.ral_stream_stitch(_ptr(p["y"]), _ptr(p["stats"]), R, T, self.leads, self.L, self.hop, _ptr(p["out"]),
                                         _stream()))

    def _plan(self, R, T):
        key = (R, T)
        if key in self.plans:
            self.plans[key] = self.plans.pop(key)       # most recently used last
            return self.plans[key]
        while len(self.plans) >= self.max_plans:        # drop the least recently used shape (its buffers and graph)
            self.plans.pop(next(iter(self.plans)))
        dev = self.model.eng.device
        nw = R * self.windows_per_record(T)
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        p = {"R": R, "T": T, "nw": nw, "rec": z(R, self.leads, T), "win": z(min(self.batch, nw), self.leads, self.L),
             "y": z(nw, self.leads, self.L), "stats": z(nw * self.leads * 2), "out": z(R, self.leads, T), "graph": None}
        if self.use_graph:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):          # warm-up outside capture (lazy LDS-size attributes, lane streams)
                self._run(p)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            p["graph"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(p["graph"]):
                self._run(p)
        self.plans[key] = p
        return p

    @torch.no_grad()
    def denoise(self, record, copy=True):
        """record: (leads, T) or a group (R, leads, T), host or device -> denoised record(s) of the same shape on the
        device.  The result is a fresh tensor; `copy=False` returns a view of the plan's output buffer instead, which the
        next call with the same shape overwrites (throughput loops that consume each result before the next call)."""
        dev = self.model.eng.device
        rec = torch.as_tensor(record, dtype=torch.float32)
        single = rec.dim() == 2
        if single:
            rec = rec[None]
        if rec.dim() != 3 or rec.shape[1] != self.leads:
            raise _lib.RalError(f"expected a record of shape ({self.leads}, T) or (R, {self.leads}, T)")
        p = self._plan(rec.shape[0], rec.shape[2])
        p["rec"].copy_(rec, non_blocking=True)
        if p["graph"] is not None:
            p["graph"].replay()
        else:
            self._run(p)
        out = p["out"][0] if single else p["out"]
        return out.clone() if copy else out
