export TMPDIR=/tmp
mkdir -p gpurun_out/r04o
timeout -k 10 900 python bench.py > gpurun_out/r04o/bench_line.json 2> gpurun_out/r04o/bench.err; echo bench rc=$?
tail -c 400 gpurun_out/r04o/bench.err
python3 - <<'EOF2'
import json
d=json.loads(open("gpurun_out/r04o/bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "wall", d["wall_s"])
print("c2", d["c2"]["ms_per_step"], d["c2"]["roofline"]["frac"], d["c2"].get("plain_step_ms"))
print("ranker", d["ranker_c3"]["ms_per_step"], d["ranker_c3"]["value"])
print("pinsage", {k: d["pinsage_c5"][k] for k in ("ms_per_iteration","positive_pairs_per_s")}, d["pinsage_c5"]["reference_default_walk_length_2"]["ms_per_iteration"])
print("e2e", d["e2e_c3"]["map_at_12"], d["e2e_c3"]["stage_s"])
print("topk", d["topk_a10"]["k12_users_per_s"], d["topk_a10"]["k256_users_per_s"])
print("map", d["map_at_12"]["value"], d["map_at_12"]["popularity_predictor_map_at_12"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:80])
EOF2
