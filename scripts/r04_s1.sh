#!/bin/bash
# round 4, GPU session 1: the ping-pong MFMA kernel alone + the tests around the re-ordered hand-offs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s1
{
  timeout 120 scripts/micro/img_mfma 2048 1048576 5 8192
  timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536
  timeout 120 scripts/micro/img_mfma 512 4194304 3 8192
  timeout 120 scripts/micro/img_mfma 300 200000 2 8192
} > gpurun_out/s1/micro.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_toppairs.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/s1/pytest.txt 2>&1
timeout 200 python scripts/mbkprof_small.py > gpurun_out/s1/mbk.txt 2>&1
tail -30 gpurun_out/s1/micro.txt; tail -5 gpurun_out/s1/pytest.txt; cat gpurun_out/s1/mbk.txt
