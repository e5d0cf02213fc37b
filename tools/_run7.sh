export TMPDIR=/tmp
mkdir -p gpurun_out/r04g
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04g/gpu_tests.log 2>&1; echo rc=$?; tail -n 12 gpurun_out/r04g/gpu_tests.log
timeout -k 10 200 python tools/bench_topk.py --full --users 16384 > gpurun_out/r04g/topk.json 2> gpurun_out/r04g/topk.err; tail -c 1500 gpurun_out/r04g/topk.json
