#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s11
{
  echo "== F=2048: whole cohorts (bit-exact check)"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 7 0 0
  echo "== F=2048: 256 workgroups"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 7 0 256
  echo "== F=300: whole cohorts"; timeout 120 scripts/micro/img_mfma 300 200000 2 8192 7 0 0
} > gpurun_out/s11/micro.txt 2>&1
grep -v "running" gpurun_out/s11/micro.txt
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_tica.py tests/test_gpu_tica_fold.py tests/test_gpu_tica_uncentred.py tests/test_gpu_tica_seams.py -x -q -m gpu > gpurun_out/s11/pytest.txt 2>&1
tail -5 gpurun_out/s11/pytest.txt
timeout 300 python scripts/config5.py 2>&1 | grep "bf16\|f32" > gpurun_out/s11/config5.txt; cat gpurun_out/s11/config5.txt
