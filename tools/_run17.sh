export TMPDIR=/tmp
mkdir -p gpurun_out/r04q
timeout -k 10 1000 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_ranker.py tests/test_gpu_end_to_end.py tests/test_gpu_reference_fixtures.py tests/test_gpu_full_size.py -x -q -k "not c4_full" > gpurun_out/r04q/tests_a.log 2>&1; echo rc=$?; tail -n 8 gpurun_out/r04q/tests_a.log
bash tools/ranker_iter.sh > gpurun_out/r04q/ranker_iter.txt 2>&1; cat gpurun_out/r04q/ranker_iter.txt
timeout -k 10 200 python tools/bench_topk.py --full --users 16384 --pre-only --ws-gib 4 | cut -c1-400
timeout -k 10 200 python tools/bench_topk.py --full --users 16384 --pre-only | cut -c1-400
