#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s13
{
echo "== default"; N_SEQ=125 timeout 120 python scripts/foldperf.py 1 0
echo "== pacing off"; MSM_TICA_COHORT_PACING=0 N_SEQ=125 timeout 120 python scripts/foldperf.py 1 0
echo "== kflush 4096"; MSM_TICA_KFLUSH=4096 N_SEQ=125 timeout 120 python scripts/foldperf.py 1
echo "== kflush 32768"; MSM_TICA_KFLUSH=32768 N_SEQ=125 timeout 120 python scripts/foldperf.py 1
echo "== N_SEQ=250"; N_SEQ=250 timeout 120 python scripts/foldperf.py 1
echo "== N_SEQ=1000"; N_SEQ=1000 timeout 120 python scripts/foldperf.py 1
} > gpurun_out/s13/small.txt 2>&1
cat gpurun_out/s13/small.txt
