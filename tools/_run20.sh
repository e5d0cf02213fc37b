export TMPDIR=/tmp
mkdir -p gpurun_out/r04t
timeout -k 10 1000 python -m pytest tests/test_gpu_lightgcn.py tests/test_gpu_dist.py tests/test_gpu_acceptance.py -x -q > gpurun_out/r04t/tests_a.log 2>&1; echo rc=$?; tail -n 6 gpurun_out/r04t/tests_a.log
bash tools/ab_c4_env.sh "skip zero partial rows||" > gpurun_out/r04t/ab.txt 2>&1; cat gpurun_out/r04t/ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -k "c4_full" > gpurun_out/r04t/tests_b.log 2>&1; echo rc=$?; tail -n 4 gpurun_out/r04t/tests_b.log
