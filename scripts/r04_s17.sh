#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s17
timeout 1200 python -m pytest tests/test_gpu_libdistance.py tests/test_gpu_fullsize.py tests/test_gpu_assign_screen.py -x -q -m gpu > gpurun_out/s17/pytest.txt 2>&1
tail -6 gpurun_out/s17/pytest.txt
timeout 300 python scripts/c3stress.py > gpurun_out/s17/c3.txt 2>&1; cat gpurun_out/s17/c3.txt
