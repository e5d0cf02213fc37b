export TMPDIR=/tmp
mkdir -p gpurun_out/r04n
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 200 --warmup 50 --pipelined"
rocprofv3 --kernel-trace -d gpurun_out/r04n/kt --output-format csv -- python3 $A > gpurun_out/r04n/kt.log 2>&1
tail -c 300 gpurun_out/r04n/kt.log
python3 tools/iter_gap.py gpurun_out/r04n/kt > gpurun_out/r04n/gap.txt 2>&1; cat gpurun_out/r04n/gap.txt
python3 tools/iter_timeline.py gpurun_out/r04n/kt > gpurun_out/r04n/timeline.txt 2>&1; tail -3 gpurun_out/r04n/timeline.txt
find gpurun_out/r04n -name "*_kernel_trace.csv" -delete
