export TMPDIR=/tmp
mkdir -p gpurun_out/r04p
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 128 --device-sampler --steps 200 --warmup 30 --pipelined"
for i in 1 2; do timeout -k 10 200 python3 $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('b128 ms/iteration', round(d['ms_per_iteration'],4), 'pos-edges/s', round(d['positive_edges_per_s']))"; done
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r04p/tests.log 2>&1; echo rc=$?; tail -n 5 gpurun_out/r04p/tests.log
