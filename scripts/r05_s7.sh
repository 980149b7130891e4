#!/bin/bash
O=gpurun_out/r05_s7
mkdir -p $O
cd /root/repo
( timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1
echo "rc=$?" >> $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
