#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s8
{
  echo "== F=2048: whole cohorts (bit-exact check)"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 7 0 0
  echo "== F=2048: 256 workgroups (remainder cohort)"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 7 0 256
  echo "== F=512: 256 workgroups"; timeout 120 scripts/micro/img_mfma 512 2097152 3 65536 7 0 256
  echo "== F=1000: 256 workgroups"; timeout 120 scripts/micro/img_mfma 1000 1048576 2 65536 7 0 256
} > gpurun_out/s8/micro.txt 2>&1
grep -v "running" gpurun_out/s8/micro.txt
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_tica.py tests/test_gpu_tica_fold.py tests/test_gpu_tica_uncentred.py -x -q -m gpu -k "bf16 or config5 or fold" > gpurun_out/s8/pytest.txt 2>&1
tail -5 gpurun_out/s8/pytest.txt
for cfg in "2048 0" "1024 0" "2048 40"; do
  set -- $cfg
  echo "== ring $1 MB, producer CUs $2"
  MSM_TICA_IMG_RING_MB=$1 MSM_TICA_IMG_PRODUCER_CUS=$2 timeout 300 python scripts/config5.py 2>&1 | grep "bf16"
done > gpurun_out/s8/config5.txt 2>&1
cat gpurun_out/s8/config5.txt
