#!/bin/bash
# per-kernel durations of tools/bench_topk.py under a variant library: tools/topk_kstats.sh <variant|now> [bench args]
v=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ks_$v; rm -rf $out; mkdir -p $out
if [ $v != now ]; then export LAPLACE_HIP_LIB=$GRAFT_REPO_ROOT/laplace-gnn-recommendation_amd/liblaplace_hip_$v.so; fi
rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 tools/bench_topk.py "$@" > $out/kt.log 2>&1
python3 - <<EOF2
import csv, glob
f = glob.glob("$out/kt/**/*kernel_stats.csv", recursive=True)[0]
print("== $v")
for r in list(csv.DictReader(open(f)))[:7]:
    print(r["Name"][:60], r["Calls"], "avg", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1))
EOF2
find $out -name "*_kernel_trace.csv" -delete
