#!/bin/bash
O=gpurun_out/r05_s6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "kmeans_label" --output-format csv -d /root/repo/$O/pmc_$C -o pmc -- python /root/repo/scripts/labelwide.py > /root/repo/$O/pmc_$C.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace -o tr -- python /root/repo/scripts/labelwide.py > /root/repo/$O/trace.log 2>&1
cd /root/repo
python - $O <<'PY' > $O/labelwide_pmc.txt
import csv, sys, glob, os, collections
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "label_v4" in r["Kernel_Name"]]
        print(c, "per dispatch of kmeans_label_v4_kernel, in launch order (KB; FETCH x2 on gfx950):", [round(float(r["Counter_Value"])) for r in rows], "grid", [r.get("Grid_Size") for r in rows])
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "label_v4" in r["Kernel_Name"]]
    print("durations (us), launch order:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows])
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/trace
cat $O/labelwide_pmc.txt
