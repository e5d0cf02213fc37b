export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
for v in "--chunk 512" "--chunk 1024" "--chunk 2048" "--chunk 1024 --band 32768" "--chunk 4096"; do
  timeout -k 10 300 python3 tools/exp_c4.py --half items $v 2>&1 | grep "C4 half\|^rows" >> gpurun_out/r04d/times.txt || echo "FAILED $v" >> gpurun_out/r04d/times.txt
done
cat gpurun_out/r04d/times.txt
