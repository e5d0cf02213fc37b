#!/bin/bash
# Everything the round's closing numbers come from, in one GPU call: full GPU suite, smoke, the bench line, the one-stream C4
# kernel + PMC profile, the ranker timeline.  Outputs under gpurun_out/final/.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $out
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $out/gpu_tests.log 2>&1; echo tests rc=$?; tail -n 3 $out/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout -k 10 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo bench rc=$?
LAPLACE_SPMM_TWO_STREAMS=0 timeout -k 10 500 bash tools/prof_bench.sh r04_bench_c4_one_stream_final --config c4 --steps 5 > $out/prof_c4.log 2>&1; echo prof rc=$?; cp gpurun_out/r04_bench_c4_one_stream_final/r04_bench_c4_one_stream_final.md $out/ 2>/dev/null; tail -n 2 $out/prof_c4.log
python3 - <<EOF2
import json
d=json.loads(open("$out/bench_line.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_ms"], r["sparse_launch_avg_ms"], r["traffic"])
print({k:(v.get("ms_per_step") or v.get("ms_per_iteration")) for k,v in d.items() if isinstance(v,dict) and ("ms_per_step" in v or "ms_per_iteration" in v)}, d["wall_s"]["total"])
EOF2
