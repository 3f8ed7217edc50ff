"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/rNN_hbm_traffic.{json,csv}: HBM bytes per launch, averaged per kernel family and per bench.py kernel kind.

    python tools/hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> profiles/r01_hbm_traffic

Units: rocprofv3 reports both counters in KB; FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide
coalesced reads at 64 bytes, same guide)."""
import collections, csv, json, sys

KIND_OF = {"k_qkv_fwd": "qkv_fwd", "k_attn_fwd": "attn_fwd", "k_mlp_fwd": "mlp_fwd", "k_resample_fwd": "resample_fwd",
           "k_resample_bwd": "resample_bwd", "k_dw": "dw", "k_mlp_bwd": "mlp_bwd", "k_mlp_bwd_s": "mlp_bwd",
           "k_attn_bwd": "attn_bwd", "k_qkv_bwd": "qkv_bwd", "k_attn_fwd_v": "attn_fwd", "k_attn_bwd_vq": "attn_bwd",
           "k_attn_bwd_vkv": "attn_bwd", "k_attn_table_reduce": "attn_bwd",
           "k_attn_bwd_w": "attn_bwd", "k_attn_bwd_h": "attn_bwd", "k_attn_bwd_m": "attn_bwd", "k_attn_bwd_mh": "attn_bwd", "k_attn_fwd_w": "attn_fwd",   # (k_attn_tpart_reduce: 0.1 MB
           # behind 14 of the 18 attention launches - not a launch of its own in the per-launch average)
           "k_mlp_fwd_w": "mlp_fwd", "k_mlp_fwd_wh": "mlp_fwd", "k_mlp_fwd_h": "mlp_fwd", "k_mlp_bwd_h": "mlp_bwd", "k_mlp_bwd_w": "mlp_bwd", "k_mlp_bwd_w2": "mlp_bwd",
           "k_qkv_fwd_ws": "qkv_fwd",
           "k_qkv_fwd_h": "qkv_fwd", "k_qkv_bwd_h": "qkv_bwd",
           "k_unet_infer": "unet_infer_fused", "k_unet_fwd_t": "unet_fwd_stage", "k_unet_out": "unet_fwd_stage"}


def family(name):
    n = name.replace("void ", "")
    if n.startswith("_Z"):                      # a mangled name rocprofv3 did not demangle: _Z<len><name>I...
        import re
        m = re.match(r"_Z(\d+)", n)
        if m:
            k = int(m.group(1))
            return n[m.end():m.end() + k]
    for sep in ("<", "("):
        if sep in n:
            n = n[:n.index(sep)]
    return n.strip()


def per_kernel(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        f = family(r["Kernel_Name"])
        tot[f] += float(r["Counter_Value"]) * 1024.0
        cnt[f] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    ft, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wc = per_kernel(write_csv, "WRITE_SIZE")
    rows, kinds = [], collections.defaultdict(lambda: [0.0, 0.0, 0])
    for f in sorted(ft, key=lambda k: -ft[k]):
        fb, wb = 2.0 * ft[f] / fc[f], (wt[f] / wc[f] if wc[f] else 0.0)
        rows.append((f, fc[f], int(fb), int(wb)))
        k = KIND_OF.get(f)
        if k:
            kinds[k][0] += 2.0 * ft[f]; kinds[k][1] += wt[f]; kinds[k][2] += fc[f]
    with open(out + ".csv", "w") as fh:
        fh.write("kernel,launches,fetch_bytes_per_launch_corrected,write_bytes_per_launch\n")
        for r in rows:
            fh.write("%s,%d,%d,%d\n" % r)
    js = {"note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units x 1024; "
                  "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950); bench.py "
                  "workload, RAL_LANES=1 RAL_NO_SIDE_STREAM=1, averaged over the launches of one kind",
          "per_launch_bytes": {k: {"fetch": int(v[0] / v[2]), "write": int(v[1] / v[2]), "total": int((v[0] + v[1]) / v[2]),
                                   "launches_sampled": v[2]} for k, v in kinds.items()}}
    json.dump(js, open(out + ".json", "w"), indent=1)
    print(json.dumps(js["per_launch_bytes"], indent=1))


if __name__ == "__main__":
    main()
