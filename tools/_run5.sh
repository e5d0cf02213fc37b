export TMPDIR=/tmp
mkdir -p gpurun_out/r04e
timeout -k 10 900 python -m pytest tests/test_gpu_ranker.py tests/test_gpu_native_vs_oracle.py tests/test_gpu_pinsage_device.py -x -q > gpurun_out/r04e/tests_a.log 2>&1; echo rc=$?; tail -n 15 gpurun_out/r04e/tests_a.log
bash tools/ranker_iter.sh > gpurun_out/r04e/ranker_iter.txt 2>&1; cat gpurun_out/r04e/ranker_iter.txt
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 200 --warmup 50 --pipelined"
rocprofv3 --kernel-trace -d gpurun_out/r04e/kt --output-format csv -- python3 $A > gpurun_out/r04e/kt.log 2>&1
python3 tools/iter_timeline.py gpurun_out/r04e/kt > gpurun_out/r04e/timeline.txt 2>&1; cat gpurun_out/r04e/timeline.txt
find gpurun_out/r04e -name "*_kernel_trace.csv" -delete
timeout -k 10 1000 python -m pytest tests/test_gpu_full_size.py -x -q -k "c5 or c3" > gpurun_out/r04e/tests_b.log 2>&1; echo rc=$?; tail -n 15 gpurun_out/r04e/tests_b.log
