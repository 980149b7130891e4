#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s5
{
  echo "== F=2048 lag 0/1/2"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 519 0
} > gpurun_out/s5/micro.txt 2>&1
grep -v "check\|running" gpurun_out/s5/micro.txt
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_tica.py tests/test_gpu_tica_fold.py tests/test_gpu_tica_uncentred.py -x -q -m gpu -k "bf16 or config5 or fold" > gpurun_out/s5/pytest.txt 2>&1
tail -15 gpurun_out/s5/pytest.txt
timeout 300 python scripts/config5.py > gpurun_out/s5/config5_768.txt 2>&1; cat gpurun_out/s5/config5_768.txt
MSM_TICA_IMG_RING_MB=128 timeout 300 python scripts/config5.py > gpurun_out/s5/config5_128.txt 2>&1; cat gpurun_out/s5/config5_128.txt
MSM_TICA_IMG_RING_MB=2048 timeout 300 python scripts/config5.py > gpurun_out/s5/config5_2048.txt 2>&1; cat gpurun_out/s5/config5_2048.txt
