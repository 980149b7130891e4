#!/bin/bash
O=gpurun_out/r05_s3
mkdir -p $O
cd /root/repo
( timeout 300 scripts/micro/img_fused 2048 100 10000 100 3 1 ) > $O/img_fused_2048.txt 2>&1
echo "rc=$?" >> $O/img_fused_2048.txt
( timeout 120 scripts/micro/img_fused 512 40 5000 7 3 0 ) > $O/img_fused_512.txt 2>&1
echo "rc=$?" >> $O/img_fused_512.txt
cat $O/img_fused_2048.txt; tail -12 $O/img_fused_512.txt
