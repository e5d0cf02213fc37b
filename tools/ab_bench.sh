#!/bin/bash
# bench.py step time and dense-launch time, three runs (A/B of propagate kernel variants)
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-map --no-plain-leg --no-pmc --no-c4 --no-ranker --no-pinsage --no-e2e 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'], 3), 'dense launch ms', round(d['roofline']['avg_launch_ms'], 4), 'loss', d.get('loss'))"
done
