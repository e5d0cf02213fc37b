#!/bin/bash
# A/B of environment switches / extra bench flags on BASELINE configs[3] at N = 1.
# usage: tools/ab_c4_env.sh "LABEL|ENV1=V1 ENV2=V2|extra bench args" ...      (empty env / args allowed)
for spec in "$@"; do
  IFS='|' read -r label envs extra <<< "$spec"
  log=/tmp/ab_c4env_$$.log
  env $envs timeout -k 10 300 python3 bench.py --config c4 --steps 8 --warmup 3 --no-cpu-baseline --no-pmc --no-map \
      --no-plain-leg --no-side --no-ranker --no-pinsage --no-e2e --no-topk $extra > $log 2>&1
  python3 - "$label" "$log" <<'EOF2'
import json, sys
label, log = sys.argv[1], sys.argv[2]
line = [l for l in open(log) if l.startswith("{")]
if not line:
    print(f"{label}: no line", open(log).read()[-600:])
else:
    d = json.loads(line[-1]); r = d["roofline"]
    f = lambda x: "-" if x is None else f"{x:.3f}"
    print(f"{label}: {d['ms_per_step']:.2f} ms/step, dense launch {f(r['avg_launch_ms'])} ms, sparse {f(r['sparse_launch_avg_ms'])} ms, "
          f"with Adam {f(r['dense_with_adam_epilogue_avg_ms'])} ms, loss {d['loss']:.6f}")
EOF2
done
