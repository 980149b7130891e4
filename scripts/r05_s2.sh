#!/bin/bash
# round 5, GPU session 2: why are the fused kernel's global loads slow?  16-byte loads, cache-resident input, PMC
O=gpurun_out/r05_s2
mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
( timeout 300 scripts/micro/img_fused 2048 100 10000 100 3 1 ) > $O/img_fused_2048.txt 2>&1
( timeout 300 scripts/micro/img_fused 2048 100 10000 100 3 1 16 ) > $O/img_fused_2048_wrap16.txt 2>&1
cd /tmp
rocprofv3 -L > /root/repo/$O/counters.txt 2>&1
for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /root/repo/$O/pmc_$T -o pmc -- /root/repo/scripts/micro/img_fused 2048 100 10000 100 1 0 > /root/repo/$O/pmc_$T.log 2>&1
done
cd /root/repo
python - $O <<'PY' > $O/pmc_summary.txt
import csv, sys, glob, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-36s per-dispatch=%.6g (n=%d)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
rm -rf $O/pmc_*/
tail -12 $O/img_fused_2048.txt; tail -12 $O/img_fused_2048_wrap16.txt | head -8; cat $O/pmc_summary.txt | head -80
