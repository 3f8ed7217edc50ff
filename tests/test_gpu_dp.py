"""Data-parallel plumbing on one GPU: the split train step of dp.HipEngineAdapter equals the fused one, and the early
gradient bucket (decoder half, all-reduced under the rest of the backward pass) is final when its event fires."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B", [64, 256])     # 256: two lanes + weight-gradient side streams (the default schedule)
def test_early_gradient_bucket_is_final_when_its_event_fires(B):
    from ecg_denoise_amd import RALENet
    from ecg_denoise_amd.dp import HipEngineAdapter
    m = RALENet("full", leads=2, L=256, max_batch=B, device=DEV, seed=4)
    x = torch.randn(B, 2, 256, device=DEV); t = torch.randn(B, 2, 256, device=DEV)
    m.train()
    e = HipEngineAdapter(m)
    (o0, n0), (o1, n1) = e.grad_buckets()
    assert o0 == 0 and o1 == n0 and o1 + n1 == m.eng.grads.numel() and n1 > 0
    names = [k for k, _ in m.named_parameters()]
    first_dec = next(k for k in names if k.startswith("utransformer4."))
    assert m.named_grads()[first_dec].data_ptr() == m.eng.grads.data_ptr() + 4 * o1      # the split is at the decoder
    for _ in range(3):      # several times: a premature event would show up as a race
        e.forward_begin(x); pred = e.forward_end(B)
        e.loss(pred, t, B)
        e.backward_begin()
        comm = e.bucket_stream()
        e.bucket_wait(1, comm)
        with torch.cuda.stream(comm):
            early = m.eng.grads[o1:o1 + n1].clone()
        e.backward_end(B)
        e.bucket_wait(0, comm)
        with torch.cuda.stream(comm):
            late = m.eng.grads[:o1].clone()
        torch.cuda.synchronize()
        assert early.abs().sum().item() > 0
        assert torch.equal(early, m.eng.grads[o1:o1 + n1])
        assert torch.equal(late, m.eng.grads[:o1])
    # and the split step is the fused step
    g_split = m.eng.grads.clone()
    y = m(x); m.loss_and_metrics(y, t); m.backward()
    torch.testing.assert_close(g_split, m.eng.grads, rtol=1e-4, atol=1e-7)


def _lockstep(adapters, xs, ts, G):
    """drive several engines like ranks of one job: whatever they yield is summed across them (the all-reduce)"""
    def reduce(gens):
        while True:
            parts = [next(g, None) for g in gens]     # every engine advances, also into its tail after the last yield
            if all(p is None for p in parts):
                return
            assert all(p is not None for p in parts), "ranks must reach the same reduction points"
            tot = sum(p.clone() for p in parts)
            for p in parts:
                p.copy_(tot)
    reduce([a.forward_iter(x, G) for a, x in zip(adapters, xs)])
    for a, t in zip(adapters, ts):
        a.loss(a.pred, t, G)
    reduce([a.backward_iter(G) for a in adapters])


def test_unet_sync_batchnorm_two_ranks_equal_one_process():
    """Two U-Net engines with half the batch each, run stage by stage with their BatchNorm sums added after every
    layer (what the data-parallel trainer's all-reduces do), reproduce one engine on the whole batch: outputs,
    running statistics, and gradients (sum of the two) -- i.e. exact global-batch BatchNorm."""
    from ecg_denoise_amd import UNet
    from ecg_denoise_amd.dp import UNetEngineAdapter
    B, L = 16, 512
    torch.manual_seed(0)
    x = torch.randn(B, 2, L, device=DEV); t = torch.randn(B, 2, L, device=DEV)
    ref = UNet(leads=2, L=L, max_batch=B, device=DEV, seed=21); ref.train()
    y = ref(x); ref.loss_and_metrics(y, t); ref.backward()
    halves = [UNet(leads=2, L=L, max_batch=B // 2, device=DEV, seed=21) for _ in range(2)]
    for h in halves:
        h.train()
    # ref's state_dict already holds the post-forward running statistics: reset the halves to a fresh model's
    fresh = UNet(leads=2, L=L, max_batch=1, device=DEV, seed=21)
    for h in halves:
        h.load_state_dict(fresh.state_dict())
    ads = [UNetEngineAdapter(h) for h in halves]
    _lockstep(ads, [x[:B // 2], x[B // 2:]], [t[:B // 2], t[B // 2:]], B)
    torch.cuda.synchronize()
    pred = torch.cat([a.pred for a in ads])
    torch.testing.assert_close(pred, y, rtol=2e-5, atol=2e-6)
    g = halves[0].eng.grads + halves[1].eng.grads
    torch.testing.assert_close(g, ref.eng.grads, rtol=2e-4, atol=2e-7)
    sd_ref, sd_h = ref.state_dict(), halves[0].state_dict()
    for k in sd_ref:
        if "running" in k:
            torch.testing.assert_close(sd_h[k], sd_ref[k], rtol=1e-5, atol=1e-7)
    # per-rank statistics would NOT match: the halves' own monolithic forward differs from the global one
    solo = UNet(leads=2, L=L, max_batch=B // 2, device=DEV, seed=21); solo.load_state_dict(fresh.state_dict()); solo.train()
    assert (solo(x[:B // 2]) - y[:B // 2]).abs().max().item() > 1e-3


def test_ralenet_sync_batchnorm_two_ranks_equal_one_process():
    """The same emulation for RA-LENet's split step (forward_begin / _end, backward_begin / _end with the two
    BatchNorm sum exchanges): two half-batch engines reproduce the whole-batch engine."""
    from ecg_denoise_amd import RALENet
    from ecg_denoise_amd.dp import HipEngineAdapter
    B, L = 8, 256
    torch.manual_seed(1)
    x = torch.randn(B, 2, L, device=DEV); t = torch.randn(B, 2, L, device=DEV)
    ref = RALENet("full", leads=2, L=L, max_batch=B, device=DEV, seed=31); ref.train()
    fresh = {k: v.clone() for k, v in ref.state_dict().items()}
    y = ref(x); ref.loss_and_metrics(y, t); ref.backward()
    halves = [RALENet("full", leads=2, L=L, max_batch=B // 2, device=DEV, seed=31) for _ in range(2)]
    for h in halves:
        h.load_state_dict(fresh); h.train()
    ads = [HipEngineAdapter(h) for h in halves]
    xs, ts = [x[:B // 2], x[B // 2:]], [t[:B // 2], t[B // 2:]]

    def exchange(lo, hi):
        tot = ads[0].bn_sums[lo:hi] + ads[1].bn_sums[lo:hi]
        for a in ads:
            a.bn_sums[lo:hi] = tot

    for a, xx in zip(ads, xs):
        a.forward_begin(xx)
    exchange(0, 32)
    preds = [a.forward_end(B) for a in ads]
    for a, p, tt in zip(ads, preds, ts):
        a.loss(p, tt, B)
        a.backward_begin()
    exchange(32, 64)
    for a in ads:
        a.backward_end(B)
    torch.cuda.synchronize()
    torch.testing.assert_close(torch.cat(preds), y, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(halves[0].eng.grads + halves[1].eng.grads, ref.eng.grads, rtol=3e-4, atol=3e-7)
    for k, v in ref.state_dict().items():
        if "running" in k:
            torch.testing.assert_close(halves[0].state_dict()[k], v, rtol=1e-5, atol=1e-7)


def test_newrale_sync_batchnorm_two_ranks_equal_one_process():
    """BASELINE config 4's split on one GPU: two half-batch `newrale` engines (12 leads x 1024 samples) driven through
    dp.NewRALEEngineAdapter with the inner model's BatchNorm sums added where the all-reduces sit reproduce the
    whole-batch model: outputs, adapter gradients (sum of the two), and the frozen inner model's running statistics."""
    from ecg_denoise_amd import NewRALE, RALENet
    from ecg_denoise_amd.dp import NewRALEEngineAdapter
    B, L = 8, 1024
    torch.manual_seed(3)
    x = torch.randn(B, 12, L, device=DEV); t = torch.randn(B, 12, L, device=DEV)

    def make(batch):
        inner = RALENet("full", leads=2, L=L, max_batch=batch, device=DEV, seed=41)
        for k, v in inner.named_parameters():
            if "relative_position_bias_table" in k:
                v.copy_(0.2 * torch.randn(v.shape, generator=torch.Generator().manual_seed(9)).to(DEV))
        return NewRALE(inner, seed=42).train()
    ref = make(B)
    y = ref(x); ref.loss_and_metrics(y, t); ref.backward()
    halves = [make(B // 2) for _ in range(2)]
    ads = [NewRALEEngineAdapter(h) for h in halves]
    xs, ts = [x[:B // 2], x[B // 2:]], [t[:B // 2], t[B // 2:]]

    def exchange(lo, hi):
        tot = ads[0].bn_sums[lo:hi] + ads[1].bn_sums[lo:hi]
        for a in ads:
            a.bn_sums[lo:hi] = tot
    for a, xx in zip(ads, xs):
        a.forward_begin(xx)
    exchange(0, 32)
    preds = [a.forward_end(B) for a in ads]
    for a, p, tt in zip(ads, preds, ts):
        a.loss(p, tt, B)
        a.backward_begin()
    exchange(32, 64)
    for a in ads:
        a.backward_end(B)
    torch.cuda.synchronize()
    torch.testing.assert_close(torch.cat(preds), y, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(halves[0].grads + halves[1].grads, ref.grads, rtol=3e-4, atol=3e-7)
    for k, v in ref.rale.state_dict().items():
        if "running" in k or "num_batches" in k:
            torch.testing.assert_close(halves[0].rale.state_dict()[k], v, rtol=1e-5, atol=1e-7)
    # per-rank statistics would NOT reproduce the whole batch
    solo = make(B // 2)
    assert (solo(xs[0]) - y[:B // 2]).abs().max().item() > 1e-4


def test_padded_window_length_two_ranks_equal_one_process():
    """A window length that is not a multiple of 256 (L = 320: padded token slots, masked) under the data-parallel split: two
    RA-LENet engines with half the batch each, their stem BatchNorm sums added where the all-reduces sit - the statistics count
    B x L existing samples, not the padded slots - reproduce the whole-batch engine: outputs, running statistics, summed
    gradients."""
    from ecg_denoise_amd import RALENet
    from ecg_denoise_amd.dp import HipEngineAdapter
    B, L = 16, 320
    torch.manual_seed(1)
    x = torch.randn(B, 2, L, device=DEV); t = torch.randn(B, 2, L, device=DEV)
    ref = RALENet("full", leads=2, L=L, max_batch=B, device=DEV, seed=31); ref.train()
    y = ref(x); ref.loss_and_metrics(y, t); ref.backward()
    halves = [RALENet("full", leads=2, L=L, max_batch=B // 2, device=DEV, seed=31) for _ in range(2)]
    ads = [HipEngineAdapter(h) for h in halves]
    xs, ts = [x[:B // 2].contiguous(), x[B // 2:].contiguous()], [t[:B // 2].contiguous(), t[B // 2:].contiguous()]
    for h in halves:
        h.train()
    for a, xx in zip(ads, xs):
        a.forward_begin(xx)
    tot = sum(a.bn_sums[:32].clone() for a in ads)
    for a in ads:
        a.bn_sums[:32].copy_(tot)
    preds = [a.forward_end(B) for a in ads]
    for a, p, tt in zip(ads, preds, ts):
        a.loss(p, tt, B)
        a.backward_begin()
    tot = sum(a.bn_sums[32:64].clone() for a in ads)
    for a in ads:
        a.bn_sums[32:64].copy_(tot)
    for a in ads:
        a.backward_end(B)
    torch.cuda.synchronize()
    assert torch.allclose(torch.cat(preds), y, rtol=0, atol=2e-6)
    torch.testing.assert_close(halves[0].eng.state, ref.eng.state, rtol=1e-5, atol=1e-7)
    g = halves[0].eng.grads + halves[1].eng.grads
    ng = {e["name"]: (e["offset"], int(torch.tensor(e["shape"]).prod())) for e in ref.eng.entries if e["kind"] == 0}
    for k, (o, n) in ng.items():
        a_, b_ = g[o:o + n].double(), ref.eng.grads[o:o + n].double()
        assert (a_ - b_).norm().item() <= 2e-4 * b_.norm().item() + 1e-8, k
