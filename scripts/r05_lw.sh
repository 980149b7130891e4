#!/bin/bash
# kernel-level split of one wide labelling call (scripts/labelwide.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lw; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $ROOT/scripts/labelwide.py > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
t=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: everything after the second-to-last inertia kernel
idx = [i for i, r in enumerate(rows) if "inertia" in r["Kernel_Name"]]
seg = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = int(rows[idx[-2]]["End_Timestamp"])
print("gap since the previous call's last kernel: %.1f us" % ((t0 - prev_end) / 1e3))
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60]))
    prev_end = e
PY
