mkdir -p gpurun_out/r04i
timeout -k 10 400 python3 tools/prof_host_native.py > gpurun_out/r04i/host_native.txt 2>&1; head -60 gpurun_out/r04i/host_native.txt
