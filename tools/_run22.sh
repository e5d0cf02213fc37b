export TMPDIR=/tmp
mkdir -p gpurun_out/r04v
timeout -k 10 800 python tools/e2e_sweep.py > gpurun_out/r04v/e2e_sweep.log 2>&1; echo rc=$?; grep -v "^$" gpurun_out/r04v/e2e_sweep.log | cut -c1-400 | tail -n 12
