#!/bin/bash
O=gpurun_out/r05_s5
mkdir -p $O
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_toppairs.py tests/test_gpu_tica.py tests/test_gpu_tica_fold.py tests/test_gpu_tica_seams.py tests/test_gpu_workflow.py tests/test_gpu_tica_uncentred.py -x -q ) > $O/pytest_tica.txt 2>&1
echo "rc=$?" >> $O/pytest_tica.txt
( timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-mbk --no-cpu-baseline ) > $O/bench_short.txt 2>&1
tail -3 $O/pytest_tica.txt; tail -1 $O/bench_short.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'], d['roofline']['kernel_ms'])"
