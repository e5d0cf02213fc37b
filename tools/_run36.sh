export TMPDIR=/tmp
out=gpurun_out/r04y; mkdir -p $out
timeout -k 10 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo bench rc=$?
python3 - <<EOF2
import json
d=json.loads(open("$out/bench_line.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_ms"], r["sparse_launch_avg_ms"], r["dense_with_adam_epilogue_avg_ms"], r["traffic"])
print({k:(v.get("ms_per_step") or v.get("ms_per_iteration")) for k,v in d.items() if isinstance(v,dict) and ("ms_per_step" in v or "ms_per_iteration" in v)})
print(d["wall_s"])
EOF2
