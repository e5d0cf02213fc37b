#!/bin/bash
# per-kernel durations of the top-K prefilter path for the product library and the named variant libraries
# (liblaplace_hip_<v>.so, e.g. built with tools/build_variant.sh pre1 "-DMI_PRE_PROBE=1" topk — probes give wrong results):
# tools/pre_probe.sh [variant ...]
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/laplace-gnn-recommendation_amd
for v in now "$@"; do
  if [ "$v" = now ]; then export LAPLACE_HIP_LIB=$P/liblaplace_hip.so; else export LAPLACE_HIP_LIB=$P/liblaplace_hip_$v.so; fi
  rm -rf /tmp/pp_$v
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/pp_$v --output-format csv -- python3 tools/bench_topk.py --full --pre-only --users 32768 $PRE_PROBE_ARGS > /tmp/pp_$v.log 2>&1
  grep -h '"workload"' /tmp/pp_$v.log | cut -c1-400
  python3 - <<EOF2
import csv, glob
f = glob.glob("/tmp/pp_$v/**/*kernel_stats.csv", recursive=True)[0]
print("== $v")
for r in list(csv.DictReader(open(f)))[:6]:
    print("  ", r["Name"][:70], r["Calls"], "avg", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1))
EOF2
done
