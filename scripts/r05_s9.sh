#!/bin/bash
O=gpurun_out/r05_s9
mkdir -p $O
cd /root/repo
( MSM_TICA_RS=1 timeout 900 python -m pytest tests/test_gpu_tica.py tests/test_gpu_tica_fold.py tests/test_gpu_tica_uncentred.py tests/test_gpu_tica_seams.py -x -q ) > $O/pytest_rs.txt 2>&1
echo "rc=$?" >> $O/pytest_rs.txt
tail -5 $O/pytest_rs.txt
for RS in 0 1 0 1; do
MSM_TICA_RS=$RS timeout 300 python bench.py --steps 4 --warmup 2 --no-extras --no-mbk --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('RS=$RS', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['top_eigenvalues'])" | tee -a $O/bench_rs.txt
done
