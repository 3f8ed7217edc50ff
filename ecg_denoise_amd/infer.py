"""Inference path: hipGraph-captured forward + streaming of long records (SURVEY §8f rank 3;
BASELINE config 4: 30-minute records, batches of 4096 windows, one MI355X).

The reference only has non-overlapping fixed-length chunking of the 650 000-sample MIT-BIH records
(local_utils/local_utils.py:116-130, 256-sample chunks, z-score per chunk group).  Here a record is cut
into windows of the model's length with an optional overlap; every window is z-scored per lead
(np_norm, local_utils/local_utils.py:261-266), denoised in eval mode (BatchNorm running statistics),
de-normalised and stitched back (overlapping regions keep the centre of each window)."""
import numpy as np
import torch

from . import _lib


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
    def __init__(self, model, batch=4096, overlap=0, use_graph=True):
        self.model, self.L, self.leads = model, model.eng.L, model.eng.leads
        self.batch = min(batch, model.eng.max_batch)
        if overlap < 0 or overlap >= self.L or overlap % 2:
            raise _lib.RalError("overlap must be an even number of samples in [0, L)")
        self.overlap, self.hop = overlap, self.L - overlap
        model.eval()
        self.fwd = GraphedForward(model, self.batch) if use_graph else None

    def windows(self, record):
        """record (leads, T) -> (n, leads, L) windows (last one right-aligned), start offsets"""
        T = record.shape[-1]
        if T < self.L:
            raise _lib.RalError(f"record shorter than one window ({T} < {self.L})")
        starts = list(range(0, T - self.L + 1, self.hop))
        if starts[-1] != T - self.L:
            starts.append(T - self.L)
        idx = torch.as_tensor(starts)[:, None] + torch.arange(self.L)[None, :]
        return record[:, idx].permute(1, 0, 2).contiguous(), starts

    @torch.no_grad()
    def denoise(self, record):
        """record: (leads, T) float tensor/array on host or device -> denoised (leads, T) on the device"""
        dev = self.model.eng.device
        rec = torch.as_tensor(record, dtype=torch.float32).to(dev)
        if rec.dim() != 2 or rec.shape[0] != self.leads:
            raise _lib.RalError(f"expected a record of shape ({self.leads}, T)")
        w, starts = self.windows(rec)
        mu = w.mean(-1, keepdim=True)
        sd = w.std(-1, unbiased=False, keepdim=True).clamp_min(1e-6)
        w = (w - mu) / sd
        out = torch.empty_like(w)
        for i in range(0, w.shape[0], self.batch):
            chunk = w[i:i + self.batch]
            if self.fwd is not None and chunk.shape[0] == self.batch:
                out[i:i + self.batch] = self.fwd(chunk)
            else:
                out[i:i + chunk.shape[0]] = self.model(chunk.contiguous())
        out = out * sd + mu
        # keep the centre of every window; the edges of the record keep the full window.  The sample -> (window,
        # offset) map depends only on the record length: built once on the host, applied as one device gather
        T = rec.shape[-1]
        src = self._stitch_map(T, starts, dev)
        return out.permute(1, 0, 2).reshape(self.leads, -1)[:, src]

    def _stitch_map(self, T, starts, dev):
        cache = self.__dict__.setdefault("_maps", {})
        if T not in cache:
            idx = np.empty(T, dtype=np.int64)
            h = self.overlap // 2
            for k, s in enumerate(starts):
                a = 0 if k == 0 else h
                b = self.L if k == len(starts) - 1 else self.L - h
                if k == len(starts) - 1 and k > 0:
                    a = max(h, starts[k - 1] + self.L - h - s)
                idx[s + a:s + b] = k * self.L + np.arange(a, b)
            cache[T] = torch.as_tensor(idx, device=dev)
        return cache[T]
