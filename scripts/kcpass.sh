#!/bin/bash
# scripts/kcpass.sh -- per-pass durations of the k-centers pass kernel over one KCenters.fit (rocprofv3 kernel trace)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_kcpass
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python $ROOT/scripts/kcperf.py > $OUT/log.txt 2>&1 < /dev/null
grep -v rocprofv3 $OUT/log.txt | tail -3
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ("kcenters_pass_kernel" in r["Kernel_Name"] or "kcenters_screen_pass_kernel" in r["Kernel_Name"])]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("pass kernels:", len(d))
fit = d[:200]
print("first fit, every 10th pass (us):", " ".join("%.0f" % fit[i] for i in range(0, 200, 10)))
print("sum first fit %.2f ms" % (sum(fit) / 1e3))
last = d[-200:]
print("last fit (white noise), every 10th (us):", " ".join("%.0f" % last[i] for i in range(0, 200, 10)))
PY
rm -f $OUT/*kernel_trace.csv $OUT/*.db $OUT/*agent_info.csv
