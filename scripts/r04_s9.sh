#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s9
{
  echo "== F=2048: whole cohorts (bit-exact check)"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 1031 0 0
  echo "== F=2048: 256 workgroups (remainder cohort)"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 1031 0 256
  echo "== F=2048: 256 workgroups, wrap 16"; timeout 120 scripts/micro/img_mfma 2048 1048576 3 65536 1031 16 256
  echo "== F=300: whole cohorts"; timeout 120 scripts/micro/img_mfma 300 200000 2 8192 1031 0 0
} > gpurun_out/s9/micro.txt 2>&1
grep -v "running" gpurun_out/s9/micro.txt
