#!/bin/bash
# scripts/r02_profiles.sh -- one GPU session: bench line, rocprofv3 kernel stats and PMC passes (own runs, kernel-trace only)
# for the bench step and for the secondary kernels.  Summaries land in gpurun_out/r02/ ; copy them to profiles/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
agg() {  # aggregate a counter_collection csv per kernel
python - "$1" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    print(k)
    for c, v in d.items():
        print("   %-32s total=%.6g  per-dispatch=%.6g (n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
PY
}
stats() {  # $1 tag, rest: command
  local tag=$1; shift
  rm -rf $OUT/tmp_$tag; mkdir -p $OUT/tmp_$tag
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$tag -o t -- "$@" > $OUT/${tag}_log.txt 2>&1 < /dev/null
  f=$(find $OUT/tmp_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${tag}_kernel_stats.csv && head -12 "$f" | cut -c1-200
  rm -rf $OUT/tmp_$tag
}
pmc() {  # $1 tag, $2 counters, rest: command
  local tag=$1; local ctr=$2; shift; shift
  rm -rf $OUT/tmp_$tag; mkdir -p $OUT/tmp_$tag
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "msm::" --output-format csv -d $OUT/tmp_$tag -o p -- "$@" > $OUT/${tag}_log.txt 2>&1 < /dev/null
  f=$(find $OUT/tmp_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then { echo "# rocprofv3 --kernel-trace --pmc $ctr -- $*"; agg "$f"; } > $OUT/${tag}.txt; head -40 $OUT/${tag}.txt; fi
  rm -rf $OUT/tmp_$tag
}
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-mbk --no-extras"
python $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 600 $OUT/bench_line.json; echo
stats bench $B
pmc pmc_bench_fetch "FETCH_SIZE" $B
pmc pmc_bench_write "WRITE_SIZE" $B
pmc pmc_bench_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" $B
K="python $ROOT/scripts/r02_kernels.py"
stats kernels $K
pmc pmc_kernels_fetch "FETCH_SIZE" $K
pmc pmc_kernels_write "WRITE_SIZE" $K
pmc pmc_kernels_busy "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" $K
grep -h "ms" $OUT/kernels_log.txt | grep -v rocprof | head -20
# second pass of the round: the exact wide assign kernel's issue mix, the MiniBatchKMeans step kernels, per-pass k-centers times
W="python $ROOT/scripts/assignwide.py"
pmc pmc_wide_assign_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" $W
pmc pmc_wide_assign_b "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" $W
stats mbk_step_F10 python $ROOT/scripts/mbkprof_small.py
stats mbk_step_F512 python $ROOT/scripts/mbkprof512.py
python $ROOT/scripts/assignperf.py 2>&1 | grep assign_nearest > $OUT/assignperf.txt; cat $OUT/assignperf.txt

