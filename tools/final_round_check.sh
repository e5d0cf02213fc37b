#!/bin/bash
# Everything the round's closing numbers come from, in one GPU call: full GPU suite, smoke, the bench line, the C2 / C4
# kernel + PMC profiles, the top-K kernel table, the H&M-scale end-to-end run.  Outputs under gpurun_out/final/.
out=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $out/gpu_tests.log 2>&1; tail -n 3 $out/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout -k 10 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo bench rc=$?
bash tools/pre_probe.sh > $out/topk_kernels.txt 2>&1; tail -n 8 $out/topk_kernels.txt
timeout -k 10 400 bash tools/prof_bench.sh r03_bench_c2 > $out/prof_c2.log 2>&1; cp gpurun_out/r03_bench_c2/r03_bench_c2.md gpurun_out/r03_bench_c2/traffic.json $out/ 2>/dev/null; tail -n 2 $out/prof_c2.log
timeout -k 10 600 bash tools/prof_bench.sh r03_bench_c4 --config c4 --steps 5 > $out/prof_c4.log 2>&1; cp gpurun_out/r03_bench_c4/r03_bench_c4.md $out/ 2>/dev/null; cp gpurun_out/r03_bench_c4/traffic.json $out/traffic_c4.json 2>/dev/null; tail -n 2 $out/prof_c4.log
timeout -k 10 600 python tools/e2e_hm_scale.py 2>/dev/null | tail -n 1 > $out/e2e.json; cut -c1-400 $out/e2e.json
python3 - <<EOF2
import json
d=json.loads(open("$out/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["ranker_c3"]["ms_per_step"], d["ranker_c3"]["value"], d["pinsage_c5"]["ms_per_iteration"], d["c4_n1"]["ms_per_step"], d["topk_a10"]["k12_users_per_s"], d["topk_a10"]["k256_users_per_s"], d["wall_s"]["total"])
EOF2
