L=$PWD/laplace-gnn-recommendation_amd
timeout -k 10 300 python tools/exp_c4_xmap.py --check 1 --n 1 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/kt.sh r04x/k1 "spmm" tools/exp_c4_xmap.py --rare 1
LAPLACE_HIP_LIB=$L/liblaplace_hip_xu2.so bash tools/kt.sh r04x/k2 "spmm" tools/exp_c4_xmap.py --rare 1
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -k "c4" 2>&1 | tail -3
