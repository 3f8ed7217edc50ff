"""gpurun_out/<round>/ (made by tools/collect_profiles.sh on the GPU box) -> the committed summaries profiles/<round>_*.

    python tools/summarise_profiles.py r02
"""
import csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
SRC, DST = os.path.join(ROOT, "gpurun_out", R), os.path.join(ROOT, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)   # (a re-collection merges into the same directories:
    if not f:                                                                  #  the newest file of a pattern is the current one)
        raise SystemExit(f"missing {pattern} under {SRC}")
    return f[-1]


def copy(pattern, name):
    shutil.copy(one(pattern), os.path.join(DST, f"{R}_{name}"))


copy("ks_default/*/*kernel_stats.csv", "kernel_stats_default_b2048.csv")
copy("ks_serial/*/*kernel_stats.csv", "kernel_stats_serialised_b2048.csv")
copy("unet_ks/*/*kernel_stats.csv", "unet_kernel_stats_b2048.csv")
copy("unet_staged_ks/*/*kernel_stats.csv", "unet_staged_kernel_stats_b2048.csv")
copy("unet_train_ks/*/*kernel_stats.csv", "unet_train_kernel_stats_b2048.csv")
for src, name in (("attn_bench.log", "attn_bench.jsonl"), ("valu_probe.log", "valu_probe.txt"), ("bench.json", "bench.json"),
                  ("bench_kinds.log", "bench_kinds.txt"), ("config_bench.log", "config_bench.jsonl"),
                  ("baselines_bench.log", "baselines_bench.jsonl"), ("bench_newrale.json", "bench_newrale.json"),
                  ("bench_unet.json", "bench_unet.json"), ("unet_train_timeline.txt", "unet_train_timeline.txt"),
                  ("dp2_overlap.txt", "dp2_overlap.txt"), ("cpu_baseline_b2048.json", "cpu_baseline_b2048.json"),
                  ("attn_bench_fp32.log", "attn_bench_fp32.jsonl"), ("step_timeline_events.txt", "step_timeline_events.txt"),
                  ("step_timeline_events_serial.txt", "step_timeline_events_serial.txt")):
    if os.path.exists(os.path.join(SRC, src)):
        lines = [l for l in open(os.path.join(SRC, src)) if "amdgpu.ids" not in l]
        open(os.path.join(DST, f"{R}_{name}"), "w").writelines(lines)
py = sys.executable
subprocess.check_call([py, os.path.join(ROOT, "tools", "hbm_traffic.py"), one("pmc_fetch/*/*counter_collection.csv"),
                       one("pmc_write/*/*counter_collection.csv"), os.path.join(DST, f"{R}_hbm_traffic")], stdout=subprocess.DEVNULL)
subprocess.check_call([py, os.path.join(ROOT, "tools", "sq_counters.py"), one("pmc_sq/*/*counter_collection.csv"),
                       os.path.join(DST, f"{R}_sq_counters.json")], stdout=subprocess.DEVNULL)


# U-Net forward: HBM bytes per launch of the fused kernel and of the staged kernels (FETCH doubled, KB units)
SCRATCH = {}        # kernel -> Scratch_Size (bytes per lane) as rocprofv3 reports it with every counter row


def per_kernel(path, counter, scale):
    tot, cnt = {}, {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        n = r["Kernel_Name"].replace("void ", "")
        n = n[:n.index("(")] if "(" in n else n
        tot[n] = tot.get(n, 0.0) + float(r["Counter_Value"]) * 1024.0 * scale
        cnt[n] = cnt.get(n, 0) + 1
        if r.get("Scratch_Size") not in (None, ""):
            SCRATCH[n] = int(float(r["Scratch_Size"]))
    return {k: tot[k] / cnt[k] for k in tot}


out = {"note": "U-Net forward at batch 2048 x 2 x 512 (tools/unet_bench.py): HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / "
               "WRITE_SIZE (separate passes; KB x 1024; FETCH doubled per MI355X_MICROARCH.md).  Algorithmic bytes: the fused "
               "kernel reads one 4 KB window and writes one (16.8 MB per launch + 42 KB of packed weights); a staged conv "
               "stage reads and writes one 4 KB tensor per window each (16.8 MB + 8.4 MB per skip)."}
for tag, key in (("unet", "fused"), ("unet_staged", "staged")):
    f = per_kernel(one(f"{tag}_fetch/*/*counter_collection.csv"), "FETCH_SIZE", 2.0)
    w = per_kernel(one(f"{tag}_write/*/*counter_collection.csv"), "WRITE_SIZE", 1.0)
    pick = (lambda n: "k_unet_infer" in n or "k_unet_pack" in n) if key == "fused" else (lambda n: "k_unet_fwd_t" in n or n.startswith("k_unet_out"))
    out[key] = {n: {"fetch_bytes": int(f[n]), "write_bytes": int(w.get(n, 0)), "scratch_bytes_per_lane": SCRATCH.get(n)}
                for n in sorted(f) if pick(n)}
json.dump(out, open(os.path.join(DST, f"{R}_unet_hbm_traffic.json"), "w"), indent=1)
# one default-schedule step as a timeline: who is the critical path (lane chains or weight-gradient side streams)
with open(os.path.join(DST, f"{R}_step_timeline.txt"), "w") as fo:
    subprocess.check_call([py, os.path.join(ROOT, "tools", "diag", "step_timeline.py"), os.path.join(SRC, "ks_default")], stdout=fo)
print("written:", sorted(os.path.basename(p) for p in glob.glob(os.path.join(DST, f"{R}_*"))))
