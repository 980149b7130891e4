#!/bin/bash
O=gpurun_out/r05_s8
mkdir -p $O
cd /root/repo
timeout 300 python scripts/packperf.py 2>&1 | grep "KC=.*fused=0" > $O/packperf_nt.txt
cat $O/packperf_nt.txt
