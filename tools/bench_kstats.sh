#!/bin/bash
# per-kernel durations of the C2 bench step under a variant library: tools/bench_kstats.sh <variant|now> [pattern]
v=$1; pat=${2:-bpr}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/bk_$v; rm -rf $out; mkdir -p $out
if [ $v != now ]; then export LAPLACE_HIP_LIB=$GRAFT_REPO_ROOT/laplace-gnn-recommendation_amd/liblaplace_hip_$v.so; fi
rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --config c2 --no-pmc --no-c4 --no-ranker --no-pinsage --no-e2e --no-map --no-cpu-baseline --no-plain-leg > $out/kt.log 2>&1
python3 - <<EOF2
import csv, glob
f = glob.glob("$out/kt/**/*kernel_stats.csv", recursive=True)[0]
print("== $v")
for r in csv.DictReader(open(f)):
    if "$pat" in r["Name"] or "gather_rows" in r["Name"] or "fill" in r["Name"].lower():
        print(r["Name"][:60], r["Calls"], "avg", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1))
EOF2
find $out -name "*_kernel_trace.csv" -delete
