#!/bin/bash
O=gpurun_out/r05_s4
mkdir -p $O
cd /root/repo
for KC in 2048 512 256; do MSM_TICA_IMG_KC=$KC timeout 300 python scripts/packperf.py 2>&1 | grep "KC=.*fused=0" ; done > $O/packperf2.txt
cat $O/packperf2.txt
