#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s27
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_workflow.py tests/test_gpu_fullsize.py tests/test_gpu_kmeans.py tests/test_gpu_configs.py -x -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
python scripts/c3stress.py 2>&1 | grep -v amdgpu > $OUT/c3.txt; cat $OUT/c3.txt
