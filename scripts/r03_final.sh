#!/bin/bash
# scripts/r03_final.sh -- end-of-round evidence in ONE session on the GPU box: PMC passes of the bench step, profiles/traffic.json
# refreshed from them (with the sha of the kernel source they were taken on), then the full bench line (whose roofline.traffic
# then comes from this very session), then the kernel-trace stats.  Everything lands in gpurun_out/r03/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
bash $ROOT/scripts/r03_profiles.sh pmc > $OUT/final_pmc_stdout.txt 2>&1
python - <<PY
import json, re, hashlib
root = "$ROOT"
def per_dispatch(path, counter):
    txt = open(path).read()
    m = re.search(r"tica_sym_f32_kernel<false, true>\n(?:.*\n)*?\s+%s\s+total=\S+\s+per-dispatch=(\S+)" % counter, txt)
    return float(m.group(1))
f = per_dispatch(root + "/gpurun_out/r03/pmc_bench_fetch.txt", "FETCH_SIZE")
w = per_dispatch(root + "/gpurun_out/r03/pmc_bench_write.txt", "WRITE_SIZE")
p = root + "/profiles/traffic.json"
d = json.load(open(p))
k = d["tica_sym_f32_kernel"]
k["fetch_size_kb_raw"], k["write_size_kb_raw"] = f, w
k["bytes_per_launch"] = int((2 * f + w) * 1024)
k["tica_hip_sha16"] = hashlib.sha256(open(root + "/msmbuilder_amd/csrc/tica.hip", "rb").read()).hexdigest()[:16]
json.dump(d, open(p, "w"), indent=2)
open(root + "/gpurun_out/r03/traffic.json", "w").write(open(p).read())
print("traffic.json:", f, w, k["bytes_per_launch"], k["tica_hip_sha16"])
PY
timeout 600 python $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt
python -c "
import json
d=json.load(open('$OUT/bench_line.json'))
print(d['ms_per_step'], d['value'], d['phases_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['strong_scaling_model']['modelled_speedup_at_8'])
"
bash $ROOT/scripts/r03_profiles.sh stats > $OUT/final_stats_stdout.txt 2>&1
grep "tica_sym\|batch_pass\|kcb_select\|assign_screen\|colsum\|project_mfma" $OUT/bench_kernel_stats.csv | cut -c1-150
