#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s21
mkdir -p $OUT
cd $ROOT
python scripts/share8.py 2>&1 | grep -v amdgpu > $OUT/share8.txt; cat $OUT/share8.txt
for m in step fit sleep; do python scripts/share_steps.py $m 2>&1 | grep -v amdgpu > $OUT/steps_$m.txt; cat $OUT/steps_$m.txt; done
