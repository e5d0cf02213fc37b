set -x
mkdir -p gpurun_out/r04a
timeout -k 10 600 python bench.py > gpurun_out/r04a/bench_line.json 2> gpurun_out/r04a/bench.err; echo bench rc=$?
tail -c 600 gpurun_out/r04a/bench.err
LAPLACE_SPMM_TWO_STREAMS=0 timeout -k 10 500 bash tools/prof_bench.sh r04_bench_c4_one_stream --config c4 --steps 5 > gpurun_out/r04a/prof_c4.log 2>&1; tail -n 3 gpurun_out/r04a/prof_c4.log
bash tools/ab_c4_env.sh "default||" "persistent_rows|LAPLACE_PERSISTENT_ROWS=1|" "plain_step||--plain-step" "one_stream|LAPLACE_SPMM_TWO_STREAMS=0|" > gpurun_out/r04a/ab.txt 2>&1
cat gpurun_out/r04a/ab.txt
