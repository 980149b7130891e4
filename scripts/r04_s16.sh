#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s16
timeout 1200 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_configs.py tests/test_gpu_workflow.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/s16/pytest.txt 2>&1
tail -8 gpurun_out/s16/pytest.txt
timeout 300 python scripts/mbk65536.py > gpurun_out/s16/mbk65536.txt 2>&1; cat gpurun_out/s16/mbk65536.txt
BATCH=1024 timeout 300 python scripts/mbk65536.py > gpurun_out/s16/mbk1024.txt 2>&1; cat gpurun_out/s16/mbk1024.txt
