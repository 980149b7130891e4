#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s22
mkdir -p $OUT
cd $ROOT
bash scripts/prof_any.sh share8 scripts/share8.py 6 > $OUT/share8_prof.txt 2>&1
f=$(find $ROOT/gpurun_out/prof_share8 -name "*kernel_stats.csv" | head -1); grep "msm::k\|msm::ksc" $f | cut -c1-220 > $OUT/share8_kernels.csv; cat $OUT/share8_kernels.csv | head -30
