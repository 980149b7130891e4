#!/bin/bash
# scripts/profile.sh <tag> [bench args...] -- rocprofv3 kernel-trace + stats of bench.py on the GPU box.
# Output lands in gpurun_out/prof_<tag>/ (scratch); copy the *_kernel_stats.csv into profiles/.
TAG=${1:-run}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-mbk "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-400
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f"
rm -f $OUT/*kernel_trace.csv $OUT/*.db $OUT/*agent_info.csv  # large; the stats file is what we keep
