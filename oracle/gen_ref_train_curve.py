"""Build-container only: train the REFERENCE model (imported from /root/reference) with the reference's own
denoise_train.train on the synthetic arrays of ecg_denoise_amd/synth.py, and record the per-epoch SNR curve as a
fixture (tests/golden/g6_ref_train_curve.npz) for the SNR-improvement comparison of the HIP path."""
import contextlib, io, os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))   # (this file lives in oracle/)
import gen_golden as G
import ralenet_oracle as O
from ecg_denoise_amd import synth

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
variant = sys.argv[2] if len(sys.argv) > 2 else "full"
torch.set_num_threads(int(os.environ.get("REF_THREADS", "6")))
ra, tr, un, newrale, dt, ev = G.import_reference()
noisy, clean = synth.make_dataset(10000, 2, 256, "emb", 0.0, seed=2023)
(trn, trc), (ten, tec) = synth.split_8000_2000(noisy, clean)
m, p = G.build_ralenet(ra, tr, variant, 2, 256, seed=777)
# reference default init scale: R-wave tables start at zero, norm affines at (1, 0)
sd = O.init_params(O.ralenet_param_shapes(variant, 2), 777)
for k in sd:
    if "relative_position_bias_table" in k: sd[k].zero_()
    elif ".norm" in k or k.startswith("conv1.2."): sd[k].fill_(1.0 if k.endswith("weight") else 0.0)
m.load_state_dict(sd, strict=False)
mk = lambda a, b: torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.tensor(a), torch.tensor(b)), 32, shuffle=False)
in_snr = ev.SNR(torch.tensor(tec), torch.tensor(ten)).mean().item()
t0 = time.time()
with tempfile.TemporaryDirectory() as td:
    cwd = os.getcwd(); os.chdir(td)
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            res = dt.train(epochs=epochs, model=m, batch_size=32, train_loader=mk(trn, trc), test_loader=mk(ten, tec),
                           use_gpu=False, model_name="ralenet", noise_name="emb", noise_intensity=0)
    finally:
        os.chdir(cwd)
# REF_TAG names a repeat of the same run (same data, same initial weights) at another intra-op thread count: only the
# summation order inside the BLAS / reduction kernels changes, which is what the run-to-run spread experiment needs
tag = os.environ.get("REF_TAG", "")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"g6_ref_train_curve_{variant}{tag}.npz"), threads=torch.get_num_threads(),
                    train_snr=np.array(res[0]), test_snr=np.array(res[1]), train_rmse=np.array(res[2]),
                    test_rmse=np.array(res[3]), input_snr_test=np.float64(in_snr), epochs=epochs, seed=777,
                    seconds=time.time() - t0)
print("done", epochs, "epochs in", time.time() - t0, "s; final test SNR", res[1][-1], "input SNR", in_snr)
