#!/bin/bash
# A/B of MI_SPMM_ROWS_RPS (rows a sub-group walks in sequence) on the ranker iteration: variants built with
# tools/build_variant.sh rps2 "-DMI_SPMM_ROWS_RPS=2" spmm (round 3: 1 -> 0.547 ms, 2 -> 0.656 ms, 4 -> 0.702 ms per iteration).
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 400 --warmup 50 --pipelined"
for v in now rps2 rps4 now; do
  if [ $v = now ]; then unset LAPLACE_HIP_LIB; else export LAPLACE_HIP_LIB=$PWD/laplace-gnn-recommendation_amd/liblaplace_hip_$v.so; fi
  python3 $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_iteration'],4), 'ms/iter', round(d['positive_edges_per_s']), 'pos-edges/s')"
done
