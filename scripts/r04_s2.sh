#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2
{
  echo "== F=512"; timeout 60 scripts/micro/img_mfma 512 2097152 3 8192 7
  echo "== F=2048"; timeout 120 scripts/micro/img_mfma 2048 1048576 5 8192 7
  echo "== F=2048, rare flush"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 7
  echo "== F=300"; timeout 60 scripts/micro/img_mfma 300 200000 2 8192 7
} > gpurun_out/s2/micro.txt 2>&1
cat gpurun_out/s2/micro.txt
