set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
for v in "--half items" "--half items --hybrid 0" "--half items --hybrid 1" "--half items --slices 2" "--half items --slices 2 --band 32768" "--half users" "--half users --slices 2"; do
  timeout -k 10 300 python3 tools/exp_c4.py $v 2>&1 | grep -v "^$" | tail -3 >> gpurun_out/r04b/times.txt || echo "FAILED $v" >> gpurun_out/r04b/times.txt
done
cat gpurun_out/r04b/times.txt
timeout -k 10 500 bash tools/prof3.sh r04b_items_base tools/exp_c4.py --half items > gpurun_out/r04b/p_base.txt 2>&1
timeout -k 10 500 bash tools/prof3.sh r04b_items_hyb0 tools/exp_c4.py --half items --hybrid 0 > gpurun_out/r04b/p_hyb0.txt 2>&1
timeout -k 10 500 bash tools/prof3.sh r04b_items_hyb1 tools/exp_c4.py --half items --hybrid 1 > gpurun_out/r04b/p_hyb1.txt 2>&1
cat gpurun_out/r04b/p_base.txt gpurun_out/r04b/p_hyb0.txt gpurun_out/r04b/p_hyb1.txt
