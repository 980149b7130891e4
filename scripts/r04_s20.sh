#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s20
mkdir -p $OUT
cd $ROOT
python scripts/share8.py 2>&1 | grep -v amdgpu > $OUT/share8.txt; cat $OUT/share8.txt
python scripts/kcperf.py 2>&1 | grep -v amdgpu > $OUT/kcperf.txt; cat $OUT/kcperf.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_libdistance.py tests/test_gpu_assign_screen.py tests/test_gpu_transition.py -x -q -k "kcenters or KCenters or distributed or sharded or world" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
