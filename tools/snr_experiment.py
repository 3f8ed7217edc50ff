"""SNR-improvement experiment on the synthetic ECG set: the main.py protocol — 10 000 windows,
8000/2000 split, batch 32, Adam 1e-3, 100 epochs, emb noise at 0 dB — run through ecg_denoise_amd.train.train on
one MI355X, compared with the per-epoch curve the REFERENCE produced on the same arrays
(tests/golden/g6_ref_train_curve_full.npz, made by oracle/gen_ref_train_curve.py in the build container)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ecg_denoise_amd import RALENet, synth
from ecg_denoise_amd.train import train

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
variant = sys.argv[2] if len(sys.argv) > 2 else "full"
init_seed = int(sys.argv[3]) if len(sys.argv) > 3 else 777   # 777 = the seed of the reference curve fixture
run_tag = sys.argv[4] if len(sys.argv) > 4 else ""           # repeats of one configuration (the fp32 atomics make runs differ)
noisy, clean = synth.make_dataset(10000, 2, 256, "emb", 0.0, seed=2023)
(trn, trc), (ten, tec) = synth.split_8000_2000(noisy, clean)
batches = lambda a, b: [(a[i:i + 32], b[i:i + 32]) for i in range(0, len(a), 32)]
m = RALENet(variant, leads=2, L=256, max_batch=32, device="cuda:0", seed=777)


def reference_curve_init(model, seed):
    """The initial weights oracle/gen_ref_train_curve.py gave the reference model: the build's seeded init rule (one
    numpy Generator, parameters visited in state_dict order) with the reference's default R-wave tables (0) and
    norm affines (1, 0); the draws those tensors consume in the rule are made and discarded."""
    import math
    from collections import OrderedDict
    rng = np.random.default_rng(seed)
    sd, fan = OrderedDict(), None
    for name, p in model.named_parameters():
        shp = tuple(p.shape)
        if "relative_position_bias_table" in name:
            rng.standard_normal(shp); a = np.zeros(shp)
        elif len(shp) == 1 and (".norm" in name or name.startswith("conv1.2.")):
            rng.standard_normal(shp); a = np.ones(shp) if name.endswith("weight") else np.zeros(shp)
        elif len(shp) >= 2:
            fan = int(np.prod(shp[1:])); a = rng.uniform(-1.0 / math.sqrt(fan), 1.0 / math.sqrt(fan), shp)
        else:
            b = 1.0 / math.sqrt(fan) if fan else 0.1; a = rng.uniform(-b, b, shp)
        sd[name] = torch.tensor(a, dtype=torch.float32)
    model.load_state_dict(sd, strict=False)


reference_curve_init(m, init_seed)
in_snr = float(np.mean(10 * np.log10((tec ** 2).mean((1, 2)) / ((ten - tec) ** 2).mean((1, 2)))))
t0 = time.time()
res = train(epochs=epochs, model=m, batch_size=32, train_loader=batches(trn, trc), test_loader=batches(ten, tec), use_gpu=True,
            model_name="ralenet", noise_name="emb", noise_intensity=0, out_dir=os.path.join(ROOT, "gpurun_out", "snr_run"),
            log=lambda *_: None)
dt = time.time() - t0
out = {"variant": variant, "init_seed": init_seed, "epochs": epochs, "seconds": round(dt, 1), "input_snr_test_db": round(in_snr, 4),
       "final_test_snr_db": res[1][-1], "snr_improvement_db": res[1][-1] - in_snr, "test_snr_curve": res[1], "train_snr_curve": res[0]}
ref_path = os.path.join(ROOT, "tests", "golden", f"g6_ref_train_curve_{variant}.npz")
if os.path.exists(ref_path):
    g = np.load(ref_path)
    n = min(len(g["test_snr"]), epochs)
    out["reference_final_test_snr_db"] = float(g["test_snr"][n - 1])
    out["reference_snr_improvement_db"] = float(g["test_snr"][n - 1] - g["input_snr_test"])
    out["delta_vs_reference_db"] = res[1][n - 1] - float(g["test_snr"][n - 1])
    out["reference_epochs"] = int(g["epochs"])
    out["reference_test_snr_curve"] = [float(v) for v in g["test_snr"][:n]]
    out["mean_last10_db"] = float(np.mean(res[1][n - 10:n])); out["reference_mean_last10_db"] = float(np.mean(g["test_snr"][n - 10:n]))
    out["first5_delta_db"] = [round(res[1][i] - float(g["test_snr"][i]), 4) for i in range(min(5, n))]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"snr_experiment_{variant}_seed{init_seed}{run_tag}.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "curve" not in k}))
print("test SNR by epoch (every 10th):", [round(v, 3) for v in res[1][9::10]])
