#!/bin/bash
# scripts/r04_all.sh -- the whole round-4 evidence in one session: the GPU tier, then scripts/r04_final.sh (all sections)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
bash scripts/r04_final.sh smoke pmc bench stats c5 all
