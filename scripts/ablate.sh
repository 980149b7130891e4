#!/bin/bash
# MFMA-kernel ablation (profiling only): prints kernel ms for each MSM_TICA_ABLATE mask
for a in 0 1 2 3 4 5 7; do
  MSM_TICA_ABLATE=$a python bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames ${1:-4000000} 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ablate=$a kernel_ms', round(d['roofline']['kernel_ms'],2), 'frac', round(d['roofline']['frac'],3))"
done
