export TMPDIR=/tmp
out=gpurun_out/r04y; mkdir -p $out
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=12 > $out/gpu_tests.log 2>&1; echo tests rc=$?; tail -n 18 $out/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
