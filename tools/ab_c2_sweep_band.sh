#!/bin/bash
# A/B of the sweep form's band width (columns) on BASELINE configs[1]: tools/ab_c2_sweep_band.sh 1024 2048 4096 ...
for v in "$@"; do
  LAPLACE_SWEEP_BAND=$v timeout -k 10 240 python3 bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-map \
      --no-plain-leg --no-c4 --no-ranker --no-pinsage --no-e2e --no-topk > /tmp/ab_c2s_$v.log 2>&1
  python3 - <<EOF2
import json
line = [l for l in open("/tmp/ab_c2s_$v.log") if l.startswith("{")]
if not line:
    print("sweep band=$v: no line", open("/tmp/ab_c2s_$v.log").read()[-600:])
else:
    d = json.loads(line[-1]); r = d["roofline"]
    print(f"sweep band=$v: {d['ms_per_step']:.3f} ms/step, dense launch {r['avg_launch_ms']:.3f} ms, sparse {r['sparse_launch_avg_ms']:.3f} ms, with Adam {r['dense_with_adam_epilogue_avg_ms']:.3f} ms, loss {d['loss']:.6f}")
EOF2
done
