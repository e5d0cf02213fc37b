export TMPDIR=/tmp
mkdir -p gpurun_out/r04r
timeout -k 10 900 python -m pytest tests/test_gpu_pinsage_device.py tests/test_pinsage.py tests/test_gpu_native_vs_oracle.py -x -q > gpurun_out/r04r/tests_a.log 2>&1; echo rc=$?; tail -n 8 gpurun_out/r04r/tests_a.log
for L in 3 2; do timeout -k 10 300 python tools/bench_pinsage.py --iters 400 --walk-length $L 2>/dev/null | cut -c1-500; done
timeout -k 10 300 python tools/bench_pinsage.py --iters 400 --walk-length 3 --batch 1024 2>/dev/null | cut -c1-500
