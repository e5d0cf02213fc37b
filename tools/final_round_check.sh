#!/bin/bash
# Everything the round's closing numbers come from, in one GPU call: full GPU suite, smoke, the bench line, the ranker
# profile + timeline, the H&M-scale end-to-end run and the PinSAGE line.  Outputs under gpurun_out/final/.
out=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $out/gpu_tests.log 2>&1; tail -n 3 $out/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout -k 10 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo bench rc=$?
bash tools/prof_ranker.sh r03_ranker_native > $out/prof_ranker.log 2>&1; cp gpurun_out/r03_ranker_native/*.json gpurun_out/r03_ranker_native/r03_ranker_native.md $out/ 2>/dev/null
bash tools/ranker_iter.sh > $out/ranker_iter.txt 2>&1; cat $out/ranker_iter.txt
bash tools/ab_gemm_timeline.sh now > $out/abtl.log 2>&1; cp gpurun_out/tl_now/timeline.txt $out/ranker_timeline.txt; tail -n 1 $out/ranker_timeline.txt
timeout -k 10 600 python tools/e2e_hm_scale.py 2>/dev/null | tail -n 1 > $out/e2e.json; cut -c1-400 $out/e2e.json
timeout -k 10 300 python tools/bench_pinsage.py --iters 300 2>/dev/null | tail -n 1 > $out/pinsage.json; cut -c300-520 $out/pinsage.json
python3 - <<EOF2
import json
d=json.loads(open("$out/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["ranker_c3"]["ms_per_step"], d["ranker_c3"]["value"], d["pinsage_c5"]["ms_per_iteration"], d["c4_n1"]["ms_per_step"], d["wall_s"]["total"])
EOF2
