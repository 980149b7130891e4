#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s7
for cfg in "2048 40 1" "2048 40 0" "2048 64 1" "768 40 1"; do
  set -- $cfg
  echo "== ring $1 MB, producer CUs $2, mask layout $3"
  MSM_TICA_IMG_RING_MB=$1 MSM_TICA_IMG_PRODUCER_CUS=$2 MSM_TICA_IMG_MASK_LAYOUT=$3 timeout 300 python scripts/config5.py 2>&1 | grep "F=2048 bf16"
done > gpurun_out/s7/config5.txt 2>&1
cat gpurun_out/s7/config5.txt
