#!/bin/bash
# ranker training loop at BASELINE configs[2] scale, sampling overlapped: ms per iteration, three runs
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 300 --warmup 50 --pipelined"
for i in 1 2 3; do
  timeout -k 10 200 python3 $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/iteration', round(d['ms_per_iteration'],4), 'pos-edges/s', round(d['positive_edges_per_s']))"
done
