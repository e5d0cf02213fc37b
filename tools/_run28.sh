L=$PWD/laplace-gnn-recommendation_amd
bash tools/kt.sh r04x/k1 "spmm" tools/exp_c4_xmap.py --rare 1
LAPLACE_HIP_LIB=$L/liblaplace_hip_xu2.so bash tools/kt.sh r04x/k2 "spmm" tools/exp_c4_xmap.py --rare 1
bash tools/kt.sh r04x/k3 "spmm" tools/exp_c4_xmap.py --rare 0
