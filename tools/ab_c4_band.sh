#!/bin/bash
# A/B of the split rows' band rule on BASELINE configs[3] at N = 1 (bench.py --config c4): LAPLACE_SPMM_BAND_MIN_PER = entries a
# band cut of a split row should hold on average (csrc/spmm.hip, plan_seg_flags_kernel).  tools/ab_c4_band.sh 1 8 32 ...
for v in "$@"; do
  LAPLACE_SPMM_BAND_MIN_PER=$v timeout -k 10 240 python3 bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-map \
      --no-plain-leg --no-ranker --no-pinsage --no-e2e --no-topk > /tmp/ab_c4_$v.log 2>&1
  python3 - <<EOF2
import json
line = [l for l in open("/tmp/ab_c4_$v.log") if l.startswith("{")]
if not line:
    print("min_per=$v: no line", open("/tmp/ab_c4_$v.log").read()[-600:])
else:
    d = json.loads(line[-1]); r = d["roofline"]
    print(f"min_per=$v: {d['ms_per_step']:.2f} ms/step, dense launch {r['avg_launch_ms']:.3f} ms, sparse {r['sparse_launch_avg_ms']:.3f} ms, with Adam {r['dense_with_adam_epilogue_avg_ms']:.3f} ms, loss {d['loss']:.6f}")
EOF2
done
