#!/bin/bash
# scripts/pmc_any.sh <tag> "<counters...>" <python script> [args] -- one rocprofv3 PMC pass over any script (kernel-trace only)
TAG=${1:-pmc}; shift
CTRS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --kernel-include-regex "msm::" --output-format csv -d $OUT -o pmc -- python $ROOT/"$@" > $OUT/run.log 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print("   %-32s per-dispatch=%.6g (n=%d)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
else
tail -5 $OUT/run.log
fi
rm -f $OUT/*kernel_trace.csv $OUT/*.db $OUT/*counter_collection.csv
