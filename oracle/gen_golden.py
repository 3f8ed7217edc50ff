"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE
(/root/reference, build container only) and running it on seeded inputs with the
build-owned deterministic weights of `ralenet_oracle.init_params`.

    python oracle/gen_golden.py            # writes tests/golden/*.npz

Fixtures are data only (inputs, expected outputs, loss, gradient summaries, BN
running statistics, post-Adam parameter summaries).  The reference never leaves this
container; tests replay the fixtures against the oracle (CPU) and the HIP path (GPU).

Import recipe: stub `torchvision.ops` (imported, unused) and
`global_utils.torch_utils.log_utils` (`train_log`, `mkdir`); `ralenet_12leads.py`
does not parse (empty `if __name__` body) so only its `newrale` class text is exec'd.
"""
import contextlib
import importlib
import io
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ralenet_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
NSAMP = 16


def import_reference():
    for name in ["torchvision", "torchvision.ops", "global_utils", "global_utils.torch_utils",
                 "global_utils.torch_utils.log_utils"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    lu = sys.modules["global_utils.torch_utils.log_utils"]
    lu.train_log = lambda *a, **k: (lambda f: f)
    lu.mkdir = lambda p: os.makedirs(p, exist_ok=True)
    sys.path.insert(0, REF)
    ra = importlib.import_module("model.raletransformer")
    tr = importlib.import_module("model.transformer")
    un = importlib.import_module("model.UNet")
    src = open(os.path.join(REF, "model", "ralenet_12leads.py")).read()
    a = src.index("class newrale")
    b = src.index('if __name__ == "__main__":')
    ns = {"nn": torch.nn, "torch": torch}
    exec(src[a:b], ns)
    dt = importlib.import_module("denoise_train")
    ev = importlib.import_module("local_utils.evaluate")
    return ra, tr, un, ns["newrale"], dt, ev


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build_ralenet(ra, tr, variant, leads, L, seed):
    """Reference model at (leads, L) using only the constructor-level patches of
    SURVEY §8c: PE max_len, rwattn.whole_length, conv1/transconv lead count."""
    mod = ra if variant == "nra" else tr
    if variant == "nra":
        m = ra.ralenet()
    elif variant == "full":
        m = quiet(tr.ralenet, high_level_enhence=True)
    else:
        m = quiet(tr.ralenet, low_level_enhence=False)
    if L > 1000:   # lift the max_len=1000 cap: rebuild each table with the reference class
        for sub in m.modules():
            if isinstance(sub, mod.AbsPositionalEncoding):
                sub.P = mod.AbsPositionalEncoding(sub.P.shape[-1], max_len=L).P
    if variant != "nra":
        for i in range(4):
            getattr(m, f"rwattn{i+1}").whole_length = L >> i
    if leads != 2:
        m.conv1[0] = torch.nn.Conv1d(leads, 8, 3, padding=1)
        m.transconv[0] = torch.nn.Conv1d(8, leads, 3, padding=1)
    p = O.init_params(O.ralenet_param_shapes(variant, leads), seed)
    missing, unexpected = m.load_state_dict(p, strict=False)
    assert not unexpected, unexpected
    assert all(("running_" in k or "num_batches" in k or "relative_position_index" in k)
               for k in missing), missing
    return m, p


def sample_idx(n):
    return np.unique(np.linspace(0, n - 1, NSAMP).astype(np.int64))


def summarize(prefix, named, out):
    """per-tensor L2 norm + NSAMP evenly spaced entries (full tensor if tiny)."""
    norms, samp = [], []
    for k, t in named.items():
        a = t.detach().double().reshape(-1).numpy()
        norms.append(np.sqrt((a * a).sum()))
        samp.append(a[sample_idx(a.size)] if a.size > NSAMP else np.pad(a, (0, NSAMP - a.size)))
    out[prefix + "_norm"] = np.array(norms, dtype=np.float64)
    out[prefix + "_samp"] = np.stack([np.pad(s, (0, NSAMP - s.size)) for s in samp]).astype(np.float64)


def model_case(name, m, p, x, tgt, bn_prefixes, steps=3):
    """forward train/eval, loss, grads, BN running stats, params after 1 and 3 Adam steps."""
    out = {"x": x.numpy(), "target": tgt.numpy()}
    keys = list(p.keys())
    named = dict(m.named_parameters())
    m.train()
    y = m(x)
    loss = torch.nn.functional.mse_loss(y, tgt)
    m.zero_grad()
    loss.backward()
    out["y_train"] = y.detach().numpy()
    out["loss"] = np.float64(loss.item())
    grads = OrderedDict((k, named[k].grad if named[k].grad is not None else torch.zeros_like(named[k]))
                        for k in keys)
    summarize("grad", grads, out)
    small = [k for k in keys if named[k].numel() <= 256]
    for k in ("conv1.0.weight", "conv1.2.weight", "conv1.2.bias", "transconv.0.weight",
              "rwattn1.relative_position_bias_table", "rwattn4.relative_position_bias_table"):
        if k in grads:
            out["gradfull_" + k] = grads[k].numpy()
    sd = m.state_dict()
    for bp in bn_prefixes:
        out["bn_mean_" + bp] = sd[bp + ".running_mean"].numpy().copy()
        out["bn_var_" + bp] = sd[bp + ".running_var"].numpy().copy()
    m.eval()
    with torch.no_grad():
        out["y_eval"] = m(x).numpy()
    # Adam trajectory: restart from the initial weights/BN state
    m.load_state_dict(p, strict=False)
    for bp in bn_prefixes:
        getattr_path(m, bp).reset_running_stats()
    opt = torch.optim.Adam([named[k] for k in keys if named[k].requires_grad], lr=1e-3)
    m.train()
    losses = []
    for s in range(steps):
        opt.zero_grad()
        l = torch.nn.functional.mse_loss(m(x), tgt)
        l.backward()
        opt.step()
        losses.append(l.item())
        if s in (0, steps - 1):
            summarize(f"adam{s+1}", OrderedDict((k, named[k]) for k in keys), out)
    out["adam_losses"] = np.array(losses, dtype=np.float64)
    out["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss {out['loss']:.6f}  |y| {np.abs(out['y_train']).mean():.4f}")


def getattr_path(m, path):
    for part in path.split("."):
        m = m[int(part)] if part.isdigit() else getattr(m, part)
    return m


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ra, tr, un, newrale, dt, ev = import_reference()

    # ---- PE table parity (oracle restatement vs reference module) -----------
    for C in O.CHANNELS:
        ref = ra.AbsPositionalEncoding(C).P[0]
        assert torch.equal(ref[:512], O.pe_table(512, C)), C
    pe = {f"C{C}": ra.AbsPositionalEncoding(C).P[0, :64].numpy() for C in O.CHANNELS}
    np.savez_compressed(os.path.join(OUT, "pe_tables.npz"), **pe)

    # ---- G1: TransformerBlock ------------------------------------------------
    g = torch.Generator().manual_seed(11)
    blk = {}
    for (C, N) in [(8, 64), (16, 32), (128, 16)]:
        for le in (0, 1):
            for msk in (0, 1):
                tag = f"C{C}_N{N}_le{le}_m{msk}"
                b = quiet(tr.TransformerBlock, C, C // 4, local_enhence=bool(le))
                shapes = OrderedDict((k, tuple(v.shape)) for k, v in b.named_parameters())
                p = O.init_params(OrderedDict(("blk." + k, s) for k, s in shapes.items()), 100 + C + le)
                b.load_state_dict(OrderedDict((k[4:], v) for k, v in p.items()))
                x = torch.randn(2, N, C, generator=g, requires_grad=True)
                Len = min(8, N)
                table = torch.randn(2 * Len - 1, C // 4, generator=g) * 0.5
                bias = O.rwave_bias(table, Len, N).unsqueeze(0) if msk else None
                y = b(x, bias) if msk else b(x)
                w = torch.randn(y.shape, generator=g)
                (y * w).sum().backward()
                blk[tag + "_x"] = x.detach().numpy(); blk[tag + "_y"] = y.detach().numpy()
                blk[tag + "_w"] = w.numpy(); blk[tag + "_dx"] = x.grad.numpy()
                blk[tag + "_table"] = table.numpy()
                nm = dict(b.named_parameters())
                blk[tag + "_gnorm"] = np.array([nm[k[4:]].grad.double().norm().item() for k in p])
                blk[tag + "_keys"] = np.array(list(p.keys()))
    np.savez_compressed(os.path.join(OUT, "g1_blocks.npz"), **blk)
    print("g1_blocks: %d arrays" % len(blk))

    # ---- G2: patch merge / separate / R-wave expansion -------------------------
    g2 = {}
    pm = ra.PatchMerging(16); ps = ra.PatchSeparate(32)
    for nm_, mod in (("pm", pm), ("ps", ps)):
        shapes = OrderedDict((nm_ + "." + k, tuple(v.shape)) for k, v in mod.named_parameters())
        p = O.init_params(shapes, 7)
        mod.load_state_dict(OrderedDict((k[3:], v) for k, v in p.items()))
    x = torch.randn(2, 32, 16, generator=g, requires_grad=True)
    y = pm(x); w = torch.randn(y.shape, generator=g); (y * w).sum().backward()
    g2.update(pm_x=x.detach().numpy(), pm_y=y.detach().numpy(), pm_w=w.numpy(), pm_dx=x.grad.numpy())
    x = torch.randn(2, 16, 32, generator=g, requires_grad=True)
    y = ps(x); w = torch.randn(y.shape, generator=g); (y * w).sum().backward()
    g2.update(ps_x=x.detach().numpy(), ps_y=y.detach().numpy(), ps_w=w.numpy(), ps_dx=x.grad.numpy())
    rp = tr.RelativePositionEmbedding(8, 32, 4)
    rp.relative_position_bias_table.data = torch.randn(15, 4, generator=g)
    g2.update(rw_table=rp.relative_position_bias_table.data.numpy(), rw_bias=rp().detach().numpy()[0])
    np.savez_compressed(os.path.join(OUT, "g2_modules.npz"), **g2)

    # ---- G3: whole models ---------------------------------------------------------
    cases = [("nra", 2, 512, 4), ("nra", 2, 256, 4), ("full", 2, 256, 4), ("mlp", 2, 256, 4),
             ("full", 2, 512, 4), ("full", 1, 512, 4), ("full", 2, 1024, 2)]
    for (variant, leads, L, B) in cases:
        gg = torch.Generator().manual_seed(2023)
        x = torch.randn(B, leads, L, generator=gg)
        tgt = torch.randn(B, leads, L, generator=gg)
        m, p = build_ralenet(ra, tr, variant, leads, L, seed=1234)
        model_case(f"g3_{variant}_l{leads}_L{L}", m, p, x, tgt, ["conv1.2"])

    gg = torch.Generator().manual_seed(2023)
    x = torch.randn(4, 2, 512, generator=gg); tgt = torch.randn(4, 2, 512, generator=gg)
    u = un.UNet(); u(x)
    p = O.init_params(O.unet_param_shapes(), 1234)
    u.load_state_dict(p, strict=False)
    for k in O.UNET_BN:
        getattr_path(u, k).reset_running_stats()
    model_case("g3_unet_l2_L512", u, p, x, tgt, O.UNET_BN)

    # newrale: frozen inner model (BN still in train mode, A16)
    inner, pin = build_ralenet(ra, tr, "full", 2, 256, seed=1234)
    nr = newrale(inner)
    pa = O.init_params(O.newrale_param_shapes(), 77)
    nr.load_state_dict(pa, strict=False)
    gg = torch.Generator().manual_seed(2023)
    x = torch.randn(2, 12, 256, generator=gg); tgt = torch.randn(2, 12, 256, generator=gg)
    nr.train()
    y = nr(x); loss = torch.nn.functional.mse_loss(y, tgt); loss.backward()
    nm = dict(nr.named_parameters())
    out = {"x": x.numpy(), "target": tgt.numpy(), "y_train": y.detach().numpy(), "loss": np.float64(loss.item())}
    for k in pa:
        out["grad_" + k] = nm[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g3_newrale_L256.npz"), **out)
    print("newrale loss", loss.item())

    # ---- G4: metrics known answers ---------------------------------------------
    gg = torch.Generator().manual_seed(5)
    yv = torch.randn(6, 2, 128, generator=gg); pv = yv + 0.3 * torch.randn(6, 2, 128, generator=gg)
    np.savez_compressed(os.path.join(OUT, "g4_metrics.npz"), y=yv.numpy(), pred=pv.numpy(),
                        snr=ev.SNR(yv, pv).numpy(), rmse=ev.RMSE(yv, pv).numpy(),
                        snr_09=ev.SNR(yv, 0.9 * yv).numpy())

    # ---- G5: denoise_train.train trace --------------------------------------------
    gg = torch.Generator().manual_seed(99)
    clean = torch.randn(128, 2, 256, generator=gg)
    noisy = clean + 0.5 * torch.randn(128, 2, 256, generator=gg)
    ds_tr = torch.utils.data.TensorDataset(noisy[:96], clean[:96])
    ds_te = torch.utils.data.TensorDataset(noisy[96:], clean[96:])
    m, p = build_ralenet(ra, tr, "nra", 2, 256, seed=4321)
    cwd = os.getcwd()
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                res = dt.train(epochs=10, model=m, batch_size=32,
                               train_loader=torch.utils.data.DataLoader(ds_tr, 32, shuffle=False),
                               test_loader=torch.utils.data.DataLoader(ds_te, 32, shuffle=False),
                               use_gpu=False, model_name="ralenet_nra", noise_name="emb", noise_intensity=0)
            line = open("output.txt").read()
            saved = sorted(os.listdir("model_save/ralenet_nra"))
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "g5_train_trace.npz"), clean=clean.numpy(), noisy=noisy.numpy(),
                        train_snr=np.array(res[0]), test_snr=np.array(res[1]),
                        train_rmse=np.array(res[2]), test_rmse=np.array(res[3]),
                        output_line=np.array(line), saved=np.array(saved))
    print("g5:", line.strip(), saved)


if __name__ == "__main__":
    main()
