#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_l64
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python $ROOT/scripts/label64scan.py > $OUT/log.txt 2>&1 < /dev/null
tail -2 $OUT/log.txt
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "label64" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
for i in range(0, len(d), 20):
    print("group %d: min %.1f us  median %.1f us  (LDS %s, grid %s x %s)" % (i // 20, min(d[i:i+20]), sorted(d[i:i+20])[10], rows[i].get("LDS_Block_Size"), rows[i].get("Grid_Size_X"), rows[i].get("Grid_Size_Y")))
PY
rm -f $OUT/*kernel_trace.csv $OUT/*.db $OUT/*agent_info.csv
