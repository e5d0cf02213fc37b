L=$PWD/laplace-gnn-recommendation_amd
timeout -k 10 300 python tools/exp_c4_xmap.py --check 1 --n 1 2>&1 | grep -v amdgpu.ids | tail -8
LAPLACE_HIP_LIB=$L/liblaplace_hip_xu2.so timeout -k 10 300 python tools/exp_c4_xmap.py --check 1 --n 1 2>&1 | grep -v amdgpu.ids | tail -8
