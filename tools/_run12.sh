export TMPDIR=/tmp
mkdir -p gpurun_out/r04l
timeout -k 10 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_ranker.py tests/test_gpu_topk_gemm.py -x -q > gpurun_out/r04l/tests_a.log 2>&1; echo rc=$?; tail -n 6 gpurun_out/r04l/tests_a.log
bash tools/ranker_iter.sh > gpurun_out/r04l/ranker_iter.txt 2>&1; cat gpurun_out/r04l/ranker_iter.txt
