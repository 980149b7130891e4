#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s6
for cfg in "2048 40" "768 40" "2048 0" "2048 24" "768 24" "384 40"; do
  set -- $cfg
  echo "== ring $1 MB, producer CUs $2"
  MSM_TICA_IMG_RING_MB=$1 MSM_TICA_IMG_PRODUCER_CUS=$2 timeout 300 python scripts/config5.py 2>&1 | grep "F=2048 bf16\|F=512 bf16"
done > gpurun_out/s6/config5.txt 2>&1
cat gpurun_out/s6/config5.txt
