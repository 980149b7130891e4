#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_libdistance.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_workflow.py tests/test_gpu_assign_screen.py tests/test_gpu_transition.py -x -q 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-mbk 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['phases_ms'], d['roofline']['kernel_ms'])"
