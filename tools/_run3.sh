export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
for v in "--hybrid 1 --pace 500" "--hybrid 1 --pace 700" "--hybrid 1 --pace 1000" "--hybrid 1 --sweep-band 4096 --pace 1400" "--hybrid 1 --sweep-band 1024 --pace 350" "--hybrid 0 --pace 700"; do
  timeout -k 10 300 python3 tools/exp_c4.py --half items $v 2>&1 | grep "C4 half" >> gpurun_out/r04c/times.txt || echo "FAILED $v" >> gpurun_out/r04c/times.txt
done
cat gpurun_out/r04c/times.txt
timeout -k 10 500 bash tools/prof3.sh r04c_items_hyb1_pace700 tools/exp_c4.py --half items --hybrid 1 --pace 700 > gpurun_out/r04c/p_hyb1_pace.txt 2>&1
cat gpurun_out/r04c/p_hyb1_pace.txt
