#!/bin/bash
# A/B: sampler host side on the calling thread (two batches ahead) vs on a thread of its own, at several interpreter switch intervals
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 300 --warmup 50 --pipelined"
run() { timeout -k 10 200 python3 $A 2>>gpurun_out/ab_err.log | python3 -c "import sys,json; print('$1', round(json.loads(sys.stdin.read())['ms_per_iteration'], 4))"; }
for i in 1 2 3; do
LAPLACE_SAMPLER_PREFETCH=1 run main
LAPLACE_SAMPLER_PREFETCH=thread LAPLACE_SAMPLER_SWITCH=1e-4 run thread_100us
LAPLACE_SAMPLER_PREFETCH=thread LAPLACE_SAMPLER_SWITCH=2e-5 run thread_20us
LAPLACE_SAMPLER_PREFETCH=thread LAPLACE_SAMPLER_SWITCH=5e-6 run thread_5us
done
