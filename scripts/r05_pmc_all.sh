#!/bin/bash
# scripts/r05_pmc_all.sh -- HBM-side counters for EVERY leg the bench prints: FETCH_SIZE and WRITE_SIZE in their
# own rocprofv3 passes (TCC slots), a kernel-trace pass for the durations, one file: gpurun_out/pmc_all/r05_pmc_all.txt
# (copied to profiles/).  FETCH_SIZE on gfx950 counts 64 B per 128-B request: DOUBLE it before comparing with bytes
# (MI355X_MICROARCH.md, HBM); Infinity-Cache hits are included.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_all
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "msm::" --output-format csv -d $OUT/$C -o pmc -- python $ROOT/scripts/legs.py > $OUT/legs_$C.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tr -- python $ROOT/scripts/legs.py > $OUT/legs_trace.log 2>&1
python - "$OUT" <<'PY' > $OUT/r05_pmc_all.txt
import csv, sys, glob, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:64]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
dur = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0].replace("void ", "")[:64]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
print("kernel | launches | FETCH_SIZE x 2 [GB] (KB counter x 2048) | WRITE_SIZE [GB] | total ms (kernel-trace pass) | (fetch x 2 + write) / ms = TB/s")
for k in sorted(agg, key=lambda k: -agg[k].get("FETCH_SIZE", 0)):
    fz = agg[k].get("FETCH_SIZE", 0.0) * 1024 * 2 / 1e9; wz = agg[k].get("WRITE_SIZE", 0.0) * 1024 / 1e9
    n = cnt[(k, "FETCH_SIZE")] or cnt[(k, "WRITE_SIZE")]
    d = dur.get(k)
    print("%-64s %6d  %10.3f  %10.3f  %10s  %s" % (k, n, fz, wz, "%.3f" % d[1] if d else "-", "%.2f" % ((fz + wz) / d[1]) if d and d[1] > 0 else "-"))
PY
echo >> $OUT/r05_pmc_all.txt
grep -A 20 "ALGORITHMIC" $OUT/legs_trace.log >> $OUT/r05_pmc_all.txt
cat $OUT/r05_pmc_all.txt
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/trace
