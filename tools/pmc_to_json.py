"""Summarise gpurun_out/prof_<tag>/ into profiles/<tag>_*.{csv,json} (run in the build container after
tools/collect_profiles.sh ran on the GPU box)."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = "gpurun_out/prof_%s" % tag
os.makedirs("profiles", exist_ok=True)
summary = {"source": "rocprofv3 --kernel-trace --stats -- python bench.py --config <cfg> --steps 5 --warmup 2 --no-cpu-baseline "
                     "(cfg2 with its side figures, the others --no-side); PMC: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                     "--kernel-trace passes per config and for tools/exp_api_true.py / tools/exp_sinc.py (tools/collect_profiles_r03.sh)",
           "kernels": {}}
if tag >= "r06":
    summary["source"] = ("rocprofv3 --kernel-trace --stats -- python bench.py --config <cfg> --steps 5 --warmup 2 --no-cpu-baseline --no-side, ONE bench "
                         "command per summary (cfg2only = --config cfg2: the three kernels of the headline step; apitrue = --config apitrue: "
                         "AudioMetrics(48000) 2229 / 480; cfg3 = the product default low-pass engine, cfg3f64 = --lowpass-engine segments); PMC: separate "
                         "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace passes of the same commands (tools/collect_profiles_r06.sh); "
                         "k_tl_* traffic = the MEAN over the launches of a step (seven inverse products at seven cuts)")
elif tag >= "r05":
    summary["source"] = summary["source"].replace("collect_profiles_r03.sh", "collect_profiles_r05.sh") + \
        "; cfg3 = the product default engine (conv, ONE ssr_fft_lowpass_multi call per step); cfg3f64 = bench.py --config cfg3 --lowpass-engine segments; " \
        "k_tl_* traffic = the MEAN over the launches of a step (seven inverse products at seven cuts)"
elif tag >= "r04":
    summary["source"] = summary["source"].replace("collect_profiles_r03.sh", "collect_profiles_r04.sh") + "; cfg3conv = bench.py --config cfg3 --lowpass-engine conv"
for stats in sorted(glob.glob(src + "/stats_*/*kernel_stats.csv")):
    cfg = os.path.basename(os.path.dirname(stats)).replace("stats_", "")
    rows = list(csv.DictReader(open(stats)))
    with open("profiles/%s_%s_kernel_stats.csv" % (tag, cfg), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows[:(24 if cfg == 'cfg2' else 12)]:
            r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
    summary["kernels"][cfg] = {r["Name"][:70]: {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1),
                                                 "pct": float(r["Percentage"])}
                               for r in rows if r["Name"].startswith(("void k_", "k_"))}
    bj = src + "/bench_under_rocprof_%s.json" % cfg
    if os.path.exists(bj) and os.path.getsize(bj):
        summary.setdefault("bench_line_under_rocprof", {})[cfg] = json.load(open(bj))
KEYS = ("k_stft_wave<double, false", "k_stft_wave<double, true", "k_ssim", "k_stft<double, 11")
MORE = {"cfg3": ("k_lowpass_wave", "k_ola_paired", "k_ola(", "k_specred_wave") if tag < "r05" else ("k_tl_fwd", "k_tl_inv", "k_tl_fold", "k_tl_pad", "k_specred_wave"),
        "cfg3f64": ("k_lowpass_wave", "k_ola_paired", "k_ola("),
        "cfg3fused": ("k_lowpass_group",), "cfg5": ("k_resample<", "k_resample_rc", "k_resample_chain"),
        "cfg3conv": ("k_tl_gemm<0>", "k_tl_gemm<2>", "k_tl_fold", "k_tl_pad"),
        "api": ("k_stft_r3_rot<double, false", "k_stft_r3_rot<double, true"), "sinc": ("k_resample_sinc",)}
pm = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for sub, keys in [("", KEYS), ("_cfg2", KEYS)] + [("_" + cfg, ks) for cfg, ks in MORE.items()]:
        f = glob.glob(src + "/pmc_%s%s/*counter_collection.csv" % (c, sub))
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            for k in keys:
                if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                    agg[k].append((int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"])))
        for k, v in agg.items():
            vals = [x for _, x in sorted(v)]
            if k in ("k_resample<", "k_resample_rc"):   # cfg-5's two stages: 441/160, then 160/147
                # (with the fused chain in the step the stages run in bench.py's report only: n x stage 1, then n x stage 2; before
                # that they alternated inside the step)
                halves = "k_resample_chain" in agg
                parts = (vals[:len(vals) // 2], vals[len(vals) // 2:]) if halves else (vals[0::2], vals[1::2])
                for name, part in ((k.rstrip("<") + " stage 1", parts[0]), (k.rstrip("<") + " stage 2", parts[1])):
                    if part:
                        part = sorted(part)
                        pm.setdefault(name, {})[c] = part[len(part) // 2]
                continue
            if k == "k_resample_sinc":        # one launch per rate pair of tools/exp_sinc.py: report the first pair (44.1 -> 48 kHz)
                pm.setdefault(k + " 44.1->48 kHz, 128 files", {})[c] = vals[0]
                continue
            if k.startswith("k_tl_"):             # launches of a step differ (seven cuts): the mean is what a step's sum needs
                pm.setdefault(k, {})[c] = sum(vals) / len(vals)
                continue
            vals = sorted(vals)
            pm.setdefault(k, {})[c] = vals[len(vals) // 2]
summary["pmc_kb_per_launch"] = pm
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads exactly 1/2 of a coalesced streaming read: x2.
# WRITE_SIZE is taken as reported (it matches the 3.16 GB of magnitudes the STFT kernel is known to write).  Both in KiB.
tr = {}
for k, d in pm.items():
    tr[k] = {"fetch_bytes_corrected": 2 * d.get("FETCH_SIZE", 0) * 1024, "write_bytes": d.get("WRITE_SIZE", 0) * 1024}
    tr[k]["total_bytes"] = tr[k]["fetch_bytes_corrected"] + tr[k]["write_bytes"]
summary["traffic_bytes_per_launch"] = tr
json.dump(summary, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
print(json.dumps({c: {k: v["avg_us"] for k, v in d.items() if v["pct"] > 1} for c, d in summary["kernels"].items()}, indent=1))
print(json.dumps(tr, indent=1))
