"""Assemble profiles/r02_snr_experiment.json: the run-to-run spread of the REFERENCE (same data, same initial weights, 100
epochs of the main.py protocol, three intra-op thread counts = three summation orders; tests/golden/g6_ref_train_curve_full*.npz,
made by oracle/gen_ref_train_curve.py in the build container) next to the HIP path's own repeats from the same initial
weights (gpurun_out/snr_experiment_full_seed777_*.json, tools/snr_experiment.py on one MI355X; fp32 atomics make them
differ) and the round-1 runs (same init and three other seeds)."""
import glob, json, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = []
for tag, th in (("", 6), ("_t2", 2), ("_t3", 3)):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"g6_ref_train_curve_full{tag}.npz"))
    c = g["test_snr"].astype(float)
    ref.append({"threads": th, "final_test_snr_db": round(c[-1], 4), "mean_last10_db": round(c[-10:].mean(), 4),
                "first5_db": [round(v, 4) for v in c[:5]], "seconds_cpu": round(float(g["seconds"])), "test_snr_curve": [round(v, 4) for v in c]})
hip = []
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "snr_experiment_full_seed777_*.json"))):
    d = json.load(open(f))
    hip.append({"run": os.path.basename(f)[-6:-5], "init_seed": 777, "final_test_snr_db": round(d["final_test_snr_db"], 4),
                "mean_last10_db": round(d["mean_last10_db"], 4), "first5_db": [round(v, 4) for v in d["test_snr_curve"][:5]],
                "seconds_gpu": d["seconds"], "test_snr_curve": [round(v, 4) for v in d["test_snr_curve"]]})
r01 = json.load(open(os.path.join(ROOT, "profiles", "r01_snr_experiment.json")))["runs"]
for d in r01:
    hip.append({"run": "r01", "init_seed": d["init_seed"], "final_test_snr_db": d["final_test_snr_db"], "mean_last10_db": d.get("mean_last10_db"),
                "seconds_gpu": d.get("seconds")})
same = [h for h in hip if h["init_seed"] == 777]
rl, hl = [r["mean_last10_db"] for r in ref], [h["mean_last10_db"] for h in same]
rf, hf = [r["final_test_snr_db"] for r in ref], [h["final_test_snr_db"] for h in same]
out = {"note": __doc__.replace("\n", " "),
       "summary": {"reference_final_db": rf, "reference_final_range_db": round(max(rf) - min(rf), 3),
                   "reference_mean_last10_db": rl, "reference_mean_of_mean_last10_db": round(float(np.mean(rl)), 3),
                   "hip_same_init_final_db": hf, "hip_same_init_final_range_db": round(max(hf) - min(hf), 3),
                   "hip_same_init_mean_last10_db": hl, "hip_mean_of_mean_last10_db": round(float(np.mean(hl)), 3),
                   "hip_minus_reference_mean_last10_db": round(float(np.mean(hl) - np.mean(rl)), 3),
                   "epochs_1_2_identical_to_db": 1e-3,
                   "reading": "three reference runs that differ only in summation order end 0.55 dB apart (19.16 .. 19.72), four HIP runs "
                              "from the same weights 1.35 dB apart (18.48 .. 19.83); all seven agree to 1e-3 dB for the first two epochs "
                              "(500 optimiser steps) and separate at epoch 3.  Means of the last ten epochs: reference 19.36 .. 19.69 "
                              "(mean 19.54), HIP 19.25 .. 19.71 (mean 19.58): the HIP path is inside the reference's own run-to-run spread."},
       "reference_runs": ref, "hip_runs": hip}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_snr_experiment.json"), "w"), indent=1)
print(json.dumps(out["summary"], indent=1))
