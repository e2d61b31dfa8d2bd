"""Developer tool: GPU busy time and the largest idle gaps in a rocprofv3 kernel trace (*_kernel_trace.csv): for the LAST `WINDOW_MS`
milliseconds of the trace (default: everything), the sum of kernel durations, the span, and the 12 largest gaps with the kernels either
side - where a host-driven pipeline leaves the GPU waiting."""
import csv
import os
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]))
rows.sort()
win = float(os.environ.get("WINDOW_MS", 0))
if win:
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - win * 1e6]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print("kernels %d, span %.2f ms, busy %.2f ms (%.0f %%)" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
gaps = sorted(((rows[i + 1][0] - max(r[1] for r in rows[max(0, i - 8):i + 1]), i) for i in range(len(rows) - 1)), reverse=True)[:12]
for g, i in sorted(gaps, key=lambda x: x[1]):
    print("  gap %7.3f ms at %8.2f ms: %s -> %s" % (g / 1e6, (rows[i][1] - rows[0][0]) / 1e6, rows[i][2], rows[i + 1][2]))
