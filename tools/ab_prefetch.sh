A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 200 --warmup 20 --pipelined"
for i in 1 2 3; do
LAPLACE_SAMPLER_PREFETCH=1 python3 $A 2>>gpurun_out/ab_err.log | python3 -c "import sys,json; print('main  ', json.loads(sys.stdin.read())['ms_per_iteration'])"
LAPLACE_SAMPLER_PREFETCH=thread python3 $A 2>>gpurun_out/ab_err.log | python3 -c "import sys,json; print('thread', json.loads(sys.stdin.read())['ms_per_iteration'])"
done
