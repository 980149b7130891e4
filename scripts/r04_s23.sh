#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s23
mkdir -p $OUT
cd $ROOT
python scripts/share8.py 2>&1 | grep -v amdgpu > $OUT/share8.txt; cat $OUT/share8.txt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -o t -- python $ROOT/scripts/solveprof.py > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(out, "tr", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
for f in glob.glob(os.path.join(out, "tr", "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Name", "")))
ev.sort()
# the last solve: from the last tica_finalise_kernel on
idx = [i for i, e in enumerate(ev) if "tica_export_sym_kernel" in e[2] or "tica_export_kernel" in e[2]]
start = idx[-2] if len(idx) >= 2 else 0
# find the start of the second-last solve to bound one solve
fin = [i for i, e in enumerate(ev) if "tica_finalise_kernel" in e[2]]
a, b = fin[-2], fin[-1]
# include the exports before finalise
while a > 0 and ("tica_export" in ev[a - 1][2] or "tica_unshift" in ev[a-1][2] or "COPY" in ev[a-1][2]): a -= 1
bb = b
while bb > 0 and ("tica_export" in ev[bb - 1][2] or "tica_unshift" in ev[bb-1][2] or "COPY" in ev[bb-1][2]): bb -= 1
t0 = ev[a][0]
with open(os.path.join(out, "solve_timeline.txt"), "w") as fo:
    prev_end = t0
    busy = 0
    for s, e, n in ev[a:bb]:
        fo.write("%9.1f us  +gap %6.1f  dur %6.1f  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n))
        busy += e - s
        prev_end = max(prev_end, e)
    fo.write("one solve: span %.1f us, busy %.1f us, %d events\n" % ((prev_end - t0) / 1e3, busy / 1e3, bb - a))
print(open(os.path.join(out, "solve_timeline.txt")).read()[-6000:])
PY
rm -rf $OUT/tr
