"""Summarise rocprofv3 --pmc counter_collection.csv files under a directory (developer tool)."""
import csv, glob, collections, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
keys = sys.argv[2:] or ["k_stft<double, 11", "k_ssim"]
for f in sorted(glob.glob(root + "/pmc_*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for k in keys:
            if k in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, vals in sorted(v.items()):
            vals = sorted(vals)
            print("%-22s %-24s n=%-3d median=%.4g" % (k, c, len(vals), vals[len(vals) // 2]))
