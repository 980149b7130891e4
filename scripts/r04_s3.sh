#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s3
{
  echo "== F=2048 wrap=16 steps (image L2-resident): the MFMA-side ceiling of each structure"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 7 16
  echo "== F=2048 wrap=512 steps (MALL-resident)"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 7 512
  echo "== F=2048 no wrap"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 7 0
} > gpurun_out/s3/micro.txt 2>&1
cat gpurun_out/s3/micro.txt
bash scripts/pmc_any.sh s3_fetch "FETCH_SIZE" "tica_img" -- scripts/micro/img_mfma 2048 1048576 1 65536 7 0 > gpurun_out/s3/pmc_fetch.txt 2>&1
bash scripts/pmc_any.sh s3_hit "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "tica_img" -- scripts/micro/img_mfma 2048 1048576 1 65536 7 0 > gpurun_out/s3/pmc_hit.txt 2>&1
bash scripts/pmc_any.sh s3_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "tica_img" -- scripts/micro/img_mfma 2048 1048576 1 65536 7 0 > gpurun_out/s3/pmc_sq.txt 2>&1
cat gpurun_out/s3/pmc_fetch.txt gpurun_out/s3/pmc_hit.txt gpurun_out/s3/pmc_sq.txt
