export TMPDIR=/tmp
L=$PWD/laplace-gnn-recommendation_amd
timeout -k 10 900 python -m pytest tests/test_gpu_lightgcn.py tests/test_gpu_full_size.py -x -q -k "not c5 and not pinsage" > gpurun_out/r04x/tests3.log 2>&1; echo rc=$?; tail -n 4 gpurun_out/r04x/tests3.log
bash tools/kt.sh r04x/q1 "spmm" tools/exp_c4_xmap.py --rare 1 --n 3
LAPLACE_SPMM_PACK=0 bash tools/kt.sh r04x/q2 "spmm" tools/exp_c4_xmap.py --rare 1 --n 3
bash tools/ab_c4_env.sh "packed||" "unpacked|LAPLACE_SPMM_PACK=0|" "packed-2||"
