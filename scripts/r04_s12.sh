#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s12
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_tica.py tests/test_gpu_tica_fold.py tests/test_gpu_tica_uncentred.py tests/test_gpu_tica_seams.py -x -q -m gpu > gpurun_out/s12/pytest.txt 2>&1
tail -5 gpurun_out/s12/pytest.txt
timeout 300 python scripts/config5.py 2>&1 | grep "f32" > gpurun_out/s12/config5.txt; cat gpurun_out/s12/config5.txt
timeout 300 python scripts/ablate.py 512 2048 1536 1024 640 > gpurun_out/s12/ablate.txt 2>&1; tail -12 gpurun_out/s12/ablate.txt
