#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s14
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -x -q -m gpu -k "world_of_eight or stress or self_launches or two_ranks or rccl_world" > gpurun_out/s14/pytest.txt 2>&1
tail -8 gpurun_out/s14/pytest.txt
PROFILE=1 timeout 300 python scripts/mbk65536.py > gpurun_out/s14/mbk65536.txt 2>&1; head -60 gpurun_out/s14/mbk65536.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s14/mbktrace -o tr -- python $GRAFT_REPO_ROOT/scripts/mbk65536.py > $GRAFT_REPO_ROOT/gpurun_out/s14/mbktrace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/s14/mbktrace -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-200
cp "$f" gpurun_out/s14/mbk65536_kernel_stats.csv; rm -rf gpurun_out/s14/mbktrace
bash scripts/r04_pmc_all.sh > gpurun_out/s14/pmc_all.log 2>&1; tail -40 gpurun_out/s14/pmc_all.log
