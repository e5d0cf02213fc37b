export TMPDIR=/tmp
mkdir -p gpurun_out/r04s
timeout -k 10 1000 python -m pytest tests/test_gpu_lightgcn.py tests/test_gpu_dist.py tests/test_gpu_acceptance.py -x -q > gpurun_out/r04s/tests_a.log 2>&1; echo rc=$?; tail -n 6 gpurun_out/r04s/tests_a.log
bash tools/ab_c4_env.sh "with x_bits||" > gpurun_out/r04s/ab.txt 2>&1; cat gpurun_out/r04s/ab.txt
timeout -k 10 300 python3 bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-map --no-plain-leg --no-side --no-ranker --no-pinsage --no-e2e --no-topk 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c2', d['ms_per_step'], r['avg_launch_ms'], r['sparse_launch_avg_ms'], r['dense_with_adam_epilogue_avg_ms'])"
