export TMPDIR=/tmp
mkdir -p gpurun_out/r04j
timeout -k 10 900 python -m pytest tests/test_gpu_ranker.py tests/test_gpu_sampler.py tests/test_gpu_topk_gemm.py tests/test_gpu_lightgcn.py -x -q -k "not full_size" > gpurun_out/r04j/tests_a.log 2>&1; echo rc=$?; tail -n 6 gpurun_out/r04j/tests_a.log
bash tools/ranker_iter.sh > gpurun_out/r04j/ranker_iter.txt 2>&1; cat gpurun_out/r04j/ranker_iter.txt
PRE_PROBE_ARGS="--ks 256" bash tools/pre_probe.sh radixt2 rp1 rp2 rp3 rp4 > gpurun_out/r04j/refine_probe.txt 2>&1; grep "==\|refine\|workload" gpurun_out/r04j/refine_probe.txt | cut -c1-260
timeout -k 10 200 python tools/bench_topk.py --full --users 16384 --pre-only > gpurun_out/r04j/topk.json 2> gpurun_out/r04j/topk.err; cat gpurun_out/r04j/topk.json
timeout -k 10 200 python tools/bench_topk.py --full --users 65536 --pre-only > gpurun_out/r04j/topk64k.json 2> gpurun_out/r04j/topk.err; cat gpurun_out/r04j/topk64k.json
