"""Assemble profiles/r02_snr_experiment.json: the run-to-run spread of the REFERENCE (same data, same initial weights, 100
epochs of the main.py protocol, three intra-op thread counts = three summation orders; tests/golden/g6_ref_train_curve_full*.npz,
made by oracle/gen_ref_train_curve.py in the build container) next to the HIP path's own repeats from the same initial
weights (gpurun_out/snr_experiment_full_seed777_*.json, tools/snr_experiment.py on one MI355X; fp32 atomics make them
differ) and the round-1 runs (same init and three other seeds)."""
import glob, json, os, sys
import numpy as np
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"      # output profiles/<round>_snr_experiment.json
TAG = sys.argv[2] if len(sys.argv) > 2 else ""            # only the HIP runs whose file name carries this tag (e.g. _r4)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = []
for tag, th in (("", 6), ("_t2", 2), ("_t3", 3), ("_t1", 1), ("_t4", 4)):
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", f"g6_ref_train_curve_full{tag}.npz")):
        continue
    g = np.load(os.path.join(ROOT, "tests", "golden", f"g6_ref_train_curve_full{tag}.npz"))
    c = g["test_snr"].astype(float)
    ref.append({"threads": th, "final_test_snr_db": round(c[-1], 4), "mean_last10_db": round(c[-10:].mean(), 4),
                "first5_db": [round(v, 4) for v in c[:5]], "seconds_cpu": round(float(g["seconds"])), "test_snr_curve": [round(v, 4) for v in c]})
hip = []
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"snr_experiment_full_seed777{TAG}*.json"))):
    d = json.load(open(f))
    hip.append({"run": os.path.basename(f)[-6:-5], "init_seed": 777, "final_test_snr_db": round(d["final_test_snr_db"], 4),
                "mean_last10_db": round(d["mean_last10_db"], 4), "first5_db": [round(v, 4) for v in d["test_snr_curve"][:5]],
                "seconds_gpu": d["seconds"], "test_snr_curve": [round(v, 4) for v in d["test_snr_curve"]]})
r01 = json.load(open(os.path.join(ROOT, "profiles", "r01_snr_experiment.json")))["runs"] if not TAG else []
for d in r01:
    hip.append({"run": "r01", "init_seed": d["init_seed"], "final_test_snr_db": d["final_test_snr_db"], "mean_last10_db": d.get("mean_last10_db"),
                "seconds_gpu": d.get("seconds")})
same = [h for h in hip if h["init_seed"] == 777]
rl, hl = [r["mean_last10_db"] for r in ref], [h["mean_last10_db"] for h in same]
rf, hf = [r["final_test_snr_db"] for r in ref], [h["final_test_snr_db"] for h in same]
first2 = np.array([r["first5_db"][:2] for r in ref] + [h["first5_db"][:2] for h in same if "first5_db" in h])
agree = float(np.abs(first2 - first2[0]).max())
sd = lambda v: float(np.std(v, ddof=1))
reading = (f"{len(ref)} reference runs that differ only in summation order end {max(rf) - min(rf):.2f} dB apart ({min(rf):.2f} .. {max(rf):.2f}), "
           f"{len(same)} HIP runs from the same weights {max(hf) - min(hf):.2f} dB apart ({min(hf):.2f} .. {max(hf):.2f}); all {len(first2)} curves that "
           f"were recorded per epoch agree to {agree:.1e} dB for the first two epochs (500 optimiser steps) and separate at epoch 3.  Means of the last ten "
           f"epochs: reference {min(rl):.2f} .. {max(rl):.2f} (mean {np.mean(rl):.2f}, s.d. {sd(rl):.2f}), HIP {min(hl):.2f} .. {max(hl):.2f} "
           f"(mean {np.mean(hl):.2f}, s.d. {sd(hl):.2f}): the HIP path is inside the reference's own run-to-run spread.")
out = {"note": __doc__.replace("\n", " "),
       "summary": {"reference_final_db": rf, "reference_final_range_db": round(max(rf) - min(rf), 3),
                   "reference_mean_last10_db": rl, "reference_mean_of_mean_last10_db": round(float(np.mean(rl)), 3),
                   "hip_same_init_final_db": hf, "hip_same_init_final_range_db": round(max(hf) - min(hf), 3),
                   "hip_same_init_mean_last10_db": hl, "hip_mean_of_mean_last10_db": round(float(np.mean(hl)), 3),
                   "hip_minus_reference_mean_last10_db": round(float(np.mean(hl) - np.mean(rl)), 3),
                   "epochs_1_2_max_difference_db": round(agree, 5), "reading": reading},
       "reference_runs": ref, "hip_runs": hip}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{ROUND}_snr_experiment.json"), "w"), indent=1)
print(json.dumps(out["summary"], indent=1))
