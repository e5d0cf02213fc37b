export TMPDIR=/tmp
mkdir -p gpurun_out/r04u
timeout -k 10 900 python -m pytest tests/test_gpu_topk_gemm.py tests/test_gpu_lightgcn.py tests/test_gpu_matching.py tests/test_gpu_end_to_end.py -x -q > gpurun_out/r04u/tests_a.log 2>&1; echo rc=$?; tail -n 5 gpurun_out/r04u/tests_a.log
timeout -k 10 200 python tools/bench_topk.py --full --users 16384 --pre-only | cut -c1-420
timeout -k 10 200 python tools/bench_topk.py --full --users 65536 --pre-only | cut -c1-420
timeout -k 10 200 python tools/bench_topk.py --full --users 16384 | cut -c100-900
