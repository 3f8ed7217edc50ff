"""Sum rocprofv3 PMC counters per kernel name: python tools/diag/pmc_sum.py <dir> [substring]"""
import collections, csv, glob, sys
acc = collections.defaultdict(collections.Counter); n = collections.Counter()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if sub not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"   {name:36s} {v / n[(k, name)]:16.0f} per launch ({n[(k, name)]} launches)")
