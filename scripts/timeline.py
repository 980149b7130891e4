"""Kernel-trace CSV of `bench.py --steps N` (rocprofv3 --kernel-trace) -> the launches of ONE steady-state step with the idle
gap in front of each (what the GPU waits for the host): profiles/r05_step_timeline.txt.  Usage: timeline.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
big = [i for i, r in enumerate(rows) if "tica_sym_f32_kernel" in r[2]]
if len(big) < 3:
    sys.exit("fewer than three MFMA launches in the trace")
a, b = big[-2], big[-1]          # from the end of the second-to-last step's MFMA kernel to the end of the last one
t0 = rows[a][1]
prev = t0
gaps = 0.0
for s, e, name in rows[a + 1:b + 1]:
    gap = max(0, s - prev) / 1e3
    gaps += gap
    short = name.split("(")[0].replace("void ", "")[:60]
    print("%10.1f us  +gap %7.1f  dur %9.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, short))
    prev = max(prev, e)
print("from the end of one step's MFMA kernel to the end of the next: %.1f us; idle between launches: %.1f us" % ((rows[b][1] - t0) / 1e3, gaps))
