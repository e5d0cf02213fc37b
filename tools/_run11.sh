export TMPDIR=/tmp
mkdir -p gpurun_out/r04k
timeout -k 10 900 python -m pytest tests/test_gpu_topk_gemm.py tests/test_gpu_lightgcn.py -x -q -k "topk or predictions or top_k" > gpurun_out/r04k/tests_a.log 2>&1; echo rc=$?; tail -n 6 gpurun_out/r04k/tests_a.log
PRE_PROBE_ARGS="--ks 256" bash tools/pre_probe.sh noslab rp3 rp4 > gpurun_out/r04k/refine_probe.txt 2>&1; grep "==\|refine\|workload" gpurun_out/r04k/refine_probe.txt | cut -c1-260
timeout -k 10 200 python tools/bench_topk.py --full --users 16384 --pre-only > gpurun_out/r04k/topk.json 2> gpurun_out/r04k/topk.err; cat gpurun_out/r04k/topk.json
timeout -k 10 200 python tools/bench_topk.py --full --users 65536 --pre-only > gpurun_out/r04k/topk64k.json 2> gpurun_out/r04k/topk.err; cat gpurun_out/r04k/topk64k.json
timeout -k 10 400 python3 tools/prof_host_native.py --hm > gpurun_out/r04k/host_native_hm.txt 2>&1; head -8 gpurun_out/r04k/host_native_hm.txt
