#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s19
mkdir -p $OUT
cd $ROOT
python scripts/share8.py 2>&1 | grep -v amdgpu > $OUT/share8.txt; cat $OUT/share8.txt
bash scripts/prof_any.sh share8 scripts/share8.py 6 > $OUT/share8_prof.txt 2>&1
f=$(find $ROOT/gpurun_out/prof_share8 -name "*kernel_stats.csv" | head -1); grep "msm::" $f | cut -c1-220 > $OUT/share8_kernels.csv; cat $OUT/share8_kernels.csv | head -30
python scripts/solvetime.py 2>&1 | grep -v amdgpu > $OUT/solvetime.txt; cat $OUT/solvetime.txt
bash scripts/prof_any.sh solve scripts/solveprof.py > $OUT/solve_prof.txt 2>&1
f=$(find $ROOT/gpurun_out/prof_solve -name "*kernel_stats.csv" | head -1); grep -v "at::native" $f | cut -c1-220 > $OUT/solve_kernels.csv; head -40 $OUT/solve_kernels.csv
