export TMPDIR=/tmp
mkdir -p gpurun_out/r04x
timeout -k 10 900 python -m pytest tests/test_gpu_lightgcn.py tests/test_gpu_full_size.py -x -q -k "not c5 and not pinsage" > gpurun_out/r04x/tests2.log 2>&1; echo rc=$?; tail -n 4 gpurun_out/r04x/tests2.log
bash tools/ab_c4_env.sh "grouped-U4||" "U2|LAPLACE_HIP_LIB=$PWD/laplace-gnn-recommendation_amd/liblaplace_hip_xu2.so|" "hint-off|LAPLACE_X_RARE=0|" "grouped-U4-one-stream|LAPLACE_SPMM_TWO_STREAMS=0|" "hint-off-one-stream|LAPLACE_SPMM_TWO_STREAMS=0 LAPLACE_X_RARE=0|"
