#!/bin/bash
# Same-box A/B of the ranker iteration: round-1 tree (.ab_r1/, exported with git archive) vs this tree, alternating;
# the box's host cores are shared, so each side is run several times and the fastest run of each is what counts.
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 400 --warmup 50 --pipelined"
for i in 1 2 3; do
  (cd .ab_r1 && python3 $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('r1  ', round(d['ms_per_iteration'],3), 'ms/iter', round(d['positive_edges_per_s']), 'pos-edges/s')")
  python3 $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('now ', round(d['ms_per_iteration'],3), 'ms/iter', round(d['positive_edges_per_s']), 'pos-edges/s')"
done
