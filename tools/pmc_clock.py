"""Developer tool (run ON the GPU box after a `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` pass): average shader clock per
kernel = GRBM_GUI_ACTIVE cycles / dispatch duration.  rocprofv3 reports the counter SUMMED over the chip's XCDs (eight on an
MI355X: a memory-bound torch kernel reads 19.26 "GHz" = 8 x 2.408), so the sum is divided by SSR_XCDS (default 8).
usage: pmc_clock.py <dir>"""
import collections, csv, glob, os, sys
XCDS = int(os.environ.get("SSR_XCDS", "8"))
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
        continue
    ns, name = dur[r["Dispatch_Id"]]
    if ns < 200000:
        continue
    a = agg[name[:60]]
    a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
for name, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s launches %4d  avg %.3f ms  clock %.0f MHz" % (name, n, ns / n / 1e6, cyc / ns * 1e3 / XCDS))
