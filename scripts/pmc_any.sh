#!/bin/bash
# scripts/pmc_any.sh <tag> "<counters...>" <kernel regex> -- <command...>: one rocprofv3 PMC pass (own run, kernel-trace
# only) over any command; prints per-kernel totals and per-dispatch averages.
TAG=$1; CTRS=$2; RE=$3; shift 3; [ "$1" = "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
CMD=()
for a in "$@"; do case "$a" in /*|-*|[0-9]*) CMD+=("$a");; *) if [ -e "$ROOT/$a" ]; then CMD+=("$ROOT/$a"); else CMD+=("$a"); fi;; esac; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --kernel-include-regex "$RE" --output-format csv -d $OUT -o pmc -- "${CMD[@]}" > $OUT/cmd.log 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
echo "== $TAG: $CTRS"
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
        print("   %-32s total=%.6g  per-dispatch=%.6g (n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
PY
else
  tail -5 $OUT/cmd.log
fi
rm -f $OUT/*kernel_trace.csv $OUT/*.db $OUT/*counter_collection.csv
