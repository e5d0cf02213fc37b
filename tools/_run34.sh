L=$PWD/laplace-gnn-recommendation_amd
for v in xs1 xs2 xsu2 xsu8; do echo variant $v; LAPLACE_HIP_LIB=$L/liblaplace_hip_$v.so bash tools/kt.sh r04x/$v "xscan" tools/exp_c4_xmap.py --rare 1 --n 3; done
