#!/bin/bash
# scripts/r03_profiles.sh [sections...] -- round-3 evidence session on the GPU box: bench line, rocprofv3 kernel stats and PMC
# passes (own runs, kernel-trace only).  Sections: bench stats pmc solve kc assign (default: all).  Summaries land in
# gpurun_out/r03/ ; the ones quoted in DESIGN.md are copied to profiles/r03_*.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
SECTIONS=${@:-bench stats pmc solve kc assign}
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
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$tag -o t -- "$@" > $OUT/${tag}_log.txt 2>&1 < /dev/null
  f=$(find $OUT/tmp_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${tag}_kernel_stats.csv && head -14 "$f" | cut -c1-180
  rm -rf $OUT/tmp_$tag
}
pmc() {  # $1 tag, $2 counters, rest: command
  local tag=$1; local ctr=$2; shift; shift
  rm -rf $OUT/tmp_$tag; mkdir -p $OUT/tmp_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "msm::" --output-format csv -d $OUT/tmp_$tag -o p -- "$@" > $OUT/${tag}_log.txt 2>&1 < /dev/null
  f=$(find $OUT/tmp_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then { echo "# rocprofv3 --kernel-trace --pmc $ctr -- $*"; agg "$f"; } > $OUT/${tag}.txt; grep -A4 -E "sym_f32|screen_pass|colsum|project_mfma" $OUT/${tag}.txt | head -40; fi
  rm -rf $OUT/tmp_$tag
}
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-mbk --no-extras"
for sec in $SECTIONS; do case $sec in
bench) timeout 600 python $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 400 $OUT/bench_line.json; echo ;;
stats) stats bench $B ;;
pmc)   pmc pmc_bench_fetch "FETCH_SIZE" $B
       pmc pmc_bench_write "WRITE_SIZE" $B
       pmc pmc_bench_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" $B ;;
solve) stats solve python $ROOT/scripts/solveprof.py
       timeout 200 python $ROOT/scripts/solvetime.py > $OUT/solvetime.txt 2>&1; grep -v amdgpu $OUT/solvetime.txt ;;
kc)    pmc pmc_kc_valu "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" python $ROOT/scripts/kcperf.py
       pmc pmc_kc_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" python $ROOT/scripts/kcperf.py ;;
bf16)  stats bf16 python $ROOT/scripts/bf16prof.py bf16
       pmc pmc_bf16_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" python $ROOT/scripts/bf16prof.py bf16
       pmc pmc_bf16_wait "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" python $ROOT/scripts/bf16prof.py bf16
       pmc pmc_bf16_clk "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" python $ROOT/scripts/bf16prof.py bf16
       grep -A7 "img_mfma" $OUT/pmc_bf16_lds.txt $OUT/pmc_bf16_wait.txt $OUT/pmc_bf16_clk.txt ;;
assign) timeout 200 python $ROOT/scripts/assignperf.py 2>&1 | grep assign_nearest > $OUT/assignperf.txt; cat $OUT/assignperf.txt ;;
esac; done
