#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s24
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_toppairs.py tests/test_gpu_tica.py -x -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
python scripts/solvetime.py 2>&1 | grep -v amdgpu > $OUT/solvetime.txt; cat $OUT/solvetime.txt
bash scripts/prof_any.sh solve scripts/solveprof.py > $OUT/solve_prof.txt 2>&1
f=$(find $ROOT/gpurun_out/prof_solve -name "*kernel_stats.csv" | head -1); grep "msm::" $f | cut -c1-200 > $OUT/solve_kernels.csv; head -12 $OUT/solve_kernels.csv
