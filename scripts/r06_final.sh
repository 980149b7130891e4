#!/bin/bash
# scripts/r06_final.sh [sections...] -- round-6 evidence in ONE session on the GPU box.  Sections (default: all):
#   smoke   __graft_entry__.smoke()
#   pmc     FETCH_SIZE / WRITE_SIZE / MFMA-busy passes over the bench step -> profiles/traffic.json refreshed (with the sha of the
#           kernel source), so that the bench line of THIS session carries the traffic measured in this session
#   bench   the full default bench line
#   stats   rocprofv3 --kernel-trace --stats of the bench step
#   c5      configs[4]: scripts/config5.py plain, then FETCH_SIZE / WRITE_SIZE passes over it -> r06_pmc_config5.txt
#   all     scripts/r06_pmc_all.sh (every bench leg)
# Everything lands in gpurun_out/r06/; the files DESIGN.md quotes are copied to profiles/r06_*.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
SECTIONS=${@:-smoke pmc bench stats c5 all}
agg() {
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
  t=$(find $OUT/tmp_$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && [ "$tag" = "bench" ] && python $ROOT/scripts/timeline.py "$t" > $OUT/r06_step_timeline.txt 2>&1 && tail -1 $OUT/r06_step_timeline.txt
  rm -rf $OUT/tmp_$tag
}
pmc() {  # $1 tag, $2 counters, rest: command
  local tag=$1; local ctr=$2; shift; shift
  rm -rf $OUT/tmp_$tag; mkdir -p $OUT/tmp_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "msm::" --output-format csv -d $OUT/tmp_$tag -o p -- "$@" > $OUT/${tag}_log.txt 2>&1 < /dev/null
  f=$(find $OUT/tmp_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then { echo "# rocprofv3 --kernel-trace --pmc $ctr -- $*"; agg "$f"; } > $OUT/${tag}.txt; fi
  rm -rf $OUT/tmp_$tag
}
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-mbk --no-extras"
for sec in $SECTIONS; do case $sec in
smoke) (cd $ROOT && timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt ;;
pmc)   pmc pmc_bench_fetch "FETCH_SIZE" $B
       pmc pmc_bench_write "WRITE_SIZE" $B
       pmc pmc_bench_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" $B
       { echo "# bench step (10M x 512 fp32, lag 100): FETCH_SIZE (KB, x2 on gfx950 for bytes), WRITE_SIZE (KB), MFMA busy -- separate passes"
         cat $OUT/pmc_bench_fetch.txt $OUT/pmc_bench_write.txt $OUT/pmc_bench_mfma.txt; } > $OUT/r06_pmc_bench.txt
       python - <<PY
import json, re, hashlib
root, out = "$ROOT", "$OUT"
def per_dispatch(path, counter):
    txt = open(path).read()
    m = re.search(r"tica_sym_f32_kernel<false, true[^\n]*\n(?:.*\n)*?\s+%s\s+total=\S+\s+per-dispatch=(\S+)" % counter, txt)
    return float(m.group(1))
f = per_dispatch(out + "/pmc_bench_fetch.txt", "FETCH_SIZE")
w = per_dispatch(out + "/pmc_bench_write.txt", "WRITE_SIZE")
p = root + "/profiles/traffic.json"
d = json.load(open(p))
k = d["tica_sym_f32_kernel"]
old = {x: k[x] for x in ("fetch_size_kb_raw", "write_size_kb_raw", "bytes_per_launch", "tica_hip_sha16", "source")}
if old["tica_hip_sha16"] != hashlib.sha256(b"".join(open(root + "/msmbuilder_amd/csrc/" + f_, "rb").read() for f_ in ("tica_common_dev.h", "tica_cg_dev.h", "tica_sym_dev.h"))).hexdigest()[:16]:
    k["round4"] = old
k["fetch_size_kb_raw"], k["write_size_kb_raw"] = f, w
k["bytes_per_launch"] = int((2 * f + w) * 1024)
k["tica_hip_sha16"] = hashlib.sha256(b"".join(open(root + "/msmbuilder_amd/csrc/" + f_, "rb").read() for f_ in ("tica_common_dev.h", "tica_cg_dev.h", "tica_sym_dev.h"))).hexdigest()[:16]
k["source"] = "profiles/r06_pmc_bench.txt (round 6 final session, scripts/r06_final.sh: FETCH_SIZE x2 + WRITE_SIZE, separate passes, then the bench line of the same session)"
json.dump(d, open(p, "w"), indent=2)
open(out + "/traffic.json", "w").write(open(p).read())
print("traffic.json:", f, w, k["bytes_per_launch"], k["tica_hip_sha16"])
PY
       ;;
bench) (cd $ROOT && timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt)
       python -c "
import json
d=json.loads(open('$OUT/bench_line.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['phases_ms'], d['roofline']['frac'], d['roofline']['traffic'], [(k, round(v['modelled_step_ms'], 2), round(v['modelled_speedup'], 2)) for k, v in d['strong_scaling_model']['ladder'].items()], d['config4_label_wide']['label_plus_inertia_ms'], {m: round(v['accumulate_ms'], 2) for m, v in d['config5_width']['modes'].items()})
" ;;
stats) stats bench $B ;;
c5)    timeout 300 python $ROOT/scripts/config5.py 2>&1 | grep -v amdgpu > $OUT/config5.txt; cat $OUT/config5.txt
       pmc pmc_c5_fetch "FETCH_SIZE" python $ROOT/scripts/legs.py c5
       pmc pmc_c5_write "WRITE_SIZE" python $ROOT/scripts/legs.py c5
       pmc pmc_c5_hit "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" python $ROOT/scripts/legs.py c5
       stats c5 python $ROOT/scripts/legs.py c5
       MSM_TICA_IMG_CARRY=0 timeout 300 python $ROOT/scripts/config5.py 2>&1 | grep "F=2048" > $OUT/config5_prepass.txt
       MSM_TICA_IMG_FUSED=1 timeout 300 python $ROOT/scripts/config5.py 2>&1 | grep "input bfloat16" > $OUT/config5_fused.txt
       export MSM_TICA_IMG_FUSED=1
       pmc pmc_c5f_fetch "FETCH_SIZE" python $ROOT/scripts/legs.py c5
       pmc pmc_c5f_write "WRITE_SIZE" python $ROOT/scripts/legs.py c5
       unset MSM_TICA_IMG_FUSED
       { echo "# BASELINE configs[4] width: 1,000,000 x 2048 bfloat16-STORED rows (4.096e9 B), one fit per mode (bf16, then bf16x2)"
         echo "# FETCH_SIZE is in KB and counts 64 B per 128-B request on gfx950: bytes = value x 2048.  WRITE_SIZE: bytes = value x 1024."
         grep -A3 "tica_img" $OUT/pmc_c5_fetch.txt; grep -A3 "tica_img" $OUT/pmc_c5_write.txt; grep -A5 "tica_img" $OUT/pmc_c5_hit.txt
         echo "# kernel durations of the same command (rocprofv3 --kernel-trace --stats)"; grep "tica_img\|Name" $OUT/c5_kernel_stats.csv | cut -c1-200
         echo "# accumulate times outside the profiler (scripts/config5.py)"; cat $OUT/config5.txt
         echo "# the FUSED kernel on the same input (MSM_TICA_IMG_FUSED=1: tica_img_fused_kernel + a column-sum pass, no image)"
         grep -A3 "tica_img\|tica_colsum" $OUT/pmc_c5f_fetch.txt; grep -A3 "tica_img\|tica_colsum" $OUT/pmc_c5f_write.txt; cat $OUT/config5_fused.txt; } > $OUT/r06_pmc_config5.txt
       ;;
all)   bash $ROOT/scripts/r05_pmc_all.sh > $OUT/pmc_all_stdout.txt 2>&1; cp $ROOT/gpurun_out/pmc_all/r05_pmc_all.txt $OUT/ 2>/dev/null; tail -40 $OUT/r06_pmc_all.txt ;;
esac; done
