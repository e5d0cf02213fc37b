#!/bin/bash
# stage timing of the fused top-K kernel: kernel-trace average of topk_scores_filter_* for the named library variants
export TMPDIR=/tmp
P=laplace-gnn-recommendation_amd
cp $P/liblaplace_hip.so $P/liblaplace_hip_full.so
for v in full "$@"; do
  cp $P/liblaplace_hip_$v.so $P/liblaplace_hip.so
  rm -rf /tmp/tk_$v
  rocprofv3 --kernel-trace --stats -d /tmp/tk_$v --output-format csv -- python3 tools/bench_topk.py --full > /dev/null 2>&1
  python3 - <<EOF2
import csv, glob
f = glob.glob("/tmp/tk_$v/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "topk_scores_filter" in r["Name"]:
        print("$v", r["Name"][30:70], round(float(r["AverageNs"])/1e3,1), "us avg")
EOF2
done
cp $P/liblaplace_hip_full.so $P/liblaplace_hip.so
