#!/bin/bash
# round 5, GPU session 1: the fused bf16 kernel alone (exactness + times + ablations), then through the library
O=gpurun_out/r05_s1
mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
( timeout 300 scripts/micro/img_fused 2048 100 10000 100 5 1 ) > $O/img_fused_2048.txt 2>&1
echo "rc=$?" >> $O/img_fused_2048.txt
( timeout 120 scripts/micro/img_fused 512 40 5000 7 3 0 ) > $O/img_fused_512.txt 2>&1
echo "rc=$?" >> $O/img_fused_512.txt
( timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "fused or config5_bf16_image" ) > $O/pytest_fused.txt 2>&1
echo "rc=$?" >> $O/pytest_fused.txt
( timeout 600 python scripts/config5.py ) > $O/config5.txt 2>&1
echo "rc=$?" >> $O/config5.txt
tail -30 $O/img_fused_2048.txt; tail -8 $O/img_fused_512.txt; tail -15 $O/pytest_fused.txt; tail -30 $O/config5.txt
