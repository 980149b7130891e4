#!/bin/bash
# scripts/mbkgaps.sh -- gaps between the dependent kernels of the MiniBatchKMeans step loop (rocprofv3 kernel trace)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_mbkgaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python $ROOT/scripts/mbkprof_small.py > $OUT/log.txt 2>&1 < /dev/null
grep "MBKM" $OUT/log.txt | tail -1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, statistics as st
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return "label" if "small_label" in n else "update" if "small_update" in n else "gather" if "gather" in n else "other"
gaps = {}
for a, b in zip(rows, rows[1:]):
    k = (short(a["Kernel_Name"]), short(b["Kernel_Name"]))
    gaps.setdefault(k, []).append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:6]:
    print("%s -> %s: n=%d median gap %.1f us, p10 %.1f, p90 %.1f" % (k[0], k[1], len(v), st.median(v), sorted(v)[len(v)//10], sorted(v)[9*len(v)//10]))
PY
rm -f $OUT/*kernel_trace.csv $OUT/*.db $OUT/*agent_info.csv
