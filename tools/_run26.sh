export TMPDIR=/tmp
out=gpurun_out/r04z; mkdir -p $out
timeout -k 10 500 bash tools/prof_tcc.sh r04z/tcc tools/exp_c4.py --half both --n 3 > $out/tcc.log 2>&1; echo tcc rc=$?; tail -n 14 $out/tcc.log
LAPLACE_SPMM_TWO_STREAMS=0 timeout -k 10 500 bash tools/prof_bench.sh r04_bench_c4_one_stream --config c4 --steps 5 > $out/prof_c4.log 2>&1; echo prof rc=$?; tail -n 3 $out/prof_c4.log
