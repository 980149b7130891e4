#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s25
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_toppairs.py tests/test_gpu_tica.py tests/test_gpu_tica_seams.py tests/test_gpu_configs.py tests/test_gpu_workflow.py tests/test_gpu_tica_uncentred.py -x -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
python scripts/solvetime.py 2>&1 | grep -v amdgpu > $OUT/solvetime.txt; cat $OUT/solvetime.txt
python scripts/solvetime.py 1024 2>&1 | grep -v amdgpu > $OUT/solvetime1024.txt; cat $OUT/solvetime1024.txt
python scripts/solvetime.py 256 2>&1 | grep -v amdgpu > $OUT/solvetime256.txt; cat $OUT/solvetime256.txt
