#!/bin/bash
# per-dispatch durations of the fused top-K kernel (grid size tells the chunk: strips x slices)
export TMPDIR=/tmp
rm -rf /tmp/tkc
timeout -k 10 150 rocprofv3 --kernel-trace -d /tmp/tkc --output-format csv -- python3 tools/bench_topk.py --full $TOPK_ARGS > /dev/null 2>&1
python3 - <<EOF2
import csv, glob
f = glob.glob("/tmp/tkc/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "topk_scores_filter" in r["Kernel_Name"]:
        print(r["Grid_Size_X"], r["Grid_Size_Y"], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1), "us")
EOF2
