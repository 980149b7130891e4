#!/bin/bash
O=gpurun_out/r05_s5
mkdir -p $O
cd /root/repo
( timeout 600 python -m pytest tests/test_gpu_toppairs.py -x -q ) > $O/pytest_toppairs.txt 2>&1
echo "rc=$?" >> $O/pytest_toppairs.txt
( timeout 300 python scripts/solvetime.py 512 10 ) > $O/solvetime_queued.txt 2>&1
( MSM_SOLVE_QUEUED=0 timeout 300 python scripts/solvetime.py 512 10 ) > $O/solvetime_hostdriven.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace -o tr -- python /root/repo/scripts/solvetime.py 512 10 > /root/repo/$O/trace.log 2>&1
cd /root/repo
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -40 $f > $O/solve_kernel_stats.csv; rm -rf $O/trace
tail -6 $O/pytest_toppairs.txt; head -6 $O/solvetime_queued.txt; head -6 $O/solvetime_hostdriven.txt; cut -c1-150 $O/solve_kernel_stats.csv | head -30
