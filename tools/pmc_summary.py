"""Summarise rocprofv3 --pmc counter_collection.csv files under a directory (developer tool): median per kernel."""
import csv, glob, collections, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
keys = sys.argv[2:] or ["k_stft", "k_ssim"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in keys):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, vals in sorted(v.items()):
        vals = sorted(vals)
        print("    %-24s n=%-3d median=%.5g" % (c, len(vals), vals[len(vals) // 2]))
