"""Summarise gpurun_out/prof_<tag>/ into profiles/<tag>_*.{csv,json} (run in the build container after gpurun)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof_%s" % tag
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(src + "/stats/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
keep = [r for r in rows if any(k in r["Name"] for k in ("k_stft", "k_ssim", "k_finalize", "k_rows"))]
with open("profiles/%s_bench_kernel_stats.csv" % tag, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows[:12]:
        r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
summary = {"source": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py (1024 pairs per launch)",
           "kernels": {}}
for r in keep:
    summary["kernels"][r["Name"][:60]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(src + "/pmc_%s/*counter_collection.csv" % c)
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        for k in ("k_stft<double, 11", "k_ssim"):
            if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = sorted(v)
        summary.setdefault("pmc_kb_per_launch", {}).setdefault(k, {})[c] = v[len(v) // 2]
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads exactly 1/2 of a coalesced streaming read; x2.
# WRITE_SIZE is taken as reported (it matches the 3.16 GB of magnitudes the STFT kernel is known to write).
pm = summary.get("pmc_kb_per_launch", {})
tr = {}
for k, d in pm.items():
    tr[k] = {"fetch_bytes_corrected": 2 * d.get("FETCH_SIZE", 0) * 1024, "write_bytes": d.get("WRITE_SIZE", 0) * 1024}
    tr[k]["total_bytes"] = tr[k]["fetch_bytes_corrected"] + tr[k]["write_bytes"]
summary["traffic_bytes_per_launch"] = tr
bj = src + "/bench_under_rocprof.json"
if os.path.exists(bj) and os.path.getsize(bj):
    summary["bench_line_under_rocprof"] = json.load(open(bj))
json.dump(summary, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
print(json.dumps(summary["kernels"], indent=1)); print(json.dumps(tr, indent=1))
