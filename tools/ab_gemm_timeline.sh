#!/bin/bash
# per-launch timeline of the ranker iteration for two libraries (variant name, then the tree's own)
set -e
export TMPDIR=/tmp
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 100 --warmup 20 --pipelined"
for v in ${1:-nopipe} now; do
  out=$GRAFT_REPO_ROOT/gpurun_out/tl_$v; rm -rf $out; mkdir -p $out
  if [ $v = now ]; then unset LAPLACE_HIP_LIB; else export LAPLACE_HIP_LIB=$GRAFT_REPO_ROOT/laplace-gnn-recommendation_amd/liblaplace_hip_$v.so; fi
  rocprofv3 --kernel-trace -d $out/kt --output-format csv -- python3 $A > $out/kt.log 2>&1
  python3 tools/iter_timeline.py $out/kt > $out/timeline.txt
  find $out -name "*.csv" -delete
done
