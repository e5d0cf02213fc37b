export TMPDIR=/tmp
mkdir -p gpurun_out/r04h
timeout -k 10 900 python -m pytest tests/test_gpu_ranker.py tests/test_gpu_native_vs_oracle.py tests/test_gpu_matching.py -x -q > gpurun_out/r04h/tests_a.log 2>&1; echo rc=$?; tail -n 6 gpurun_out/r04h/tests_a.log
bash tools/ranker_iter.sh > gpurun_out/r04h/ranker_iter.txt 2>&1; cat gpurun_out/r04h/ranker_iter.txt
for v in wg192 wg256; do echo "== $v"; LAPLACE_HIP_LIB=$PWD/laplace-gnn-recommendation_amd/liblaplace_hip_$v.so bash tools/ranker_iter.sh; done > gpurun_out/r04h/ranker_iter_wg.txt 2>&1; cat gpurun_out/r04h/ranker_iter_wg.txt
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 200 --warmup 50 --pipelined"
rocprofv3 --kernel-trace -d gpurun_out/r04h/kt --output-format csv -- python3 $A > gpurun_out/r04h/kt.log 2>&1
python3 tools/iter_timeline.py gpurun_out/r04h/kt > gpurun_out/r04h/timeline.txt 2>&1; cat gpurun_out/r04h/timeline.txt
find gpurun_out/r04h -name "*_kernel_trace.csv" -delete
PRE_PROBE_ARGS="--ks 256" bash tools/pre_probe.sh rp1 rp2 rp3 rp4 > gpurun_out/r04h/refine_probe.txt 2>&1; grep -v "^ *$" gpurun_out/r04h/refine_probe.txt | grep "==\|refine\|prefilter_bf16" 
