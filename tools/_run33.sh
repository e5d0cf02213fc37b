export TMPDIR=/tmp
mkdir -p gpurun_out/r04x
timeout -k 10 900 python -m pytest tests/test_gpu_lightgcn.py tests/test_gpu_full_size.py -x -q -k "not c5 and not pinsage" > gpurun_out/r04x/tests3.log 2>&1; echo rc=$?; tail -n 6 gpurun_out/r04x/tests3.log
