#!/bin/bash
# top-K: throughput, kernel trace and MFMA busy counters of tools/bench_topk.py
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 tools/bench_topk.py "$@" > $out/line.json 2> $out/line.err
cat $out/line.json
rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 tools/bench_topk.py "$@" > $out/kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $out/mfma --output-format csv -- python3 tools/bench_topk.py "$@" > $out/mfma.log 2>&1 || true
python3 - <<EOF2
import csv, glob, collections
f = glob.glob("$out/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", r["Percentage"], "%")
fs = glob.glob("$out/mfma/**/*counter_collection.csv", recursive=True)
if fs:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "topk_scores_filter" in r["Kernel_Name"] or "topk_prefilter_bf16" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        print(k, {c: round(x) for c, x in m.items()})
        if m.get("SQ_BUSY_CU_CYCLES"):
            print("   MFMA busy / (4 SIMDs x CU-busy cycles) =", round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_BUSY_CU_CYCLES"]), 3))
EOF2
find $out -name "*_kernel_trace.csv" -delete 2>/dev/null || true
find $out -name "*agent_info.csv" -delete 2>/dev/null || true
