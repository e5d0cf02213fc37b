export TMPDIR=/tmp
mkdir -p gpurun_out/r04f
timeout -k 10 900 python -m pytest tests/test_gpu_ranker.py tests/test_gpu_native_vs_oracle.py -x -q > gpurun_out/r04f/tests_a.log 2>&1; echo rc=$?; tail -n 8 gpurun_out/r04f/tests_a.log
bash tools/ranker_iter.sh > gpurun_out/r04f/ranker_iter.txt 2>&1; cat gpurun_out/r04f/ranker_iter.txt
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 200 --warmup 50 --pipelined"
rocprofv3 --kernel-trace -d gpurun_out/r04f/kt --output-format csv -- python3 $A > gpurun_out/r04f/kt.log 2>&1
python3 tools/iter_timeline.py gpurun_out/r04f/kt > gpurun_out/r04f/timeline.txt 2>&1; cat gpurun_out/r04f/timeline.txt
find gpurun_out/r04f -name "*_kernel_trace.csv" -delete
timeout -k 10 900 python -m pytest tests/test_gpu_lightgcn.py -x -q -k "hybrid or sweep or plan" > gpurun_out/r04f/tests_b.log 2>&1; echo rc=$?; tail -n 8 gpurun_out/r04f/tests_b.log
