#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s4
{
  echo "== ablations, F=2048, rare flush"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 263 0
  echo "== ablations, F=2048, wrap 16"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 263 16
} > gpurun_out/s4/micro.txt 2>&1
grep -v "check\|running" gpurun_out/s4/micro.txt
