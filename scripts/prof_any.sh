#!/bin/bash
# scripts/prof_any.sh <tag> <script.py relative to the repo root> [args...] -- rocprofv3 kernel-trace + stats of any repo script on the GPU box.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $ROOT/"$@" > $OUT/log.txt 2>&1 < /dev/null
grep -v rocprofv3 $OUT/log.txt | tail -5
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -10 "$f" | cut -c1-170; else echo "no kernel stats produced"; fi
rm -f $OUT/*kernel_trace.csv $OUT/*.db $OUT/*agent_info.csv
