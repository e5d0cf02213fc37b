L=$PWD/laplace-gnn-recommendation_amd
for i in 1 2 3 4 5; do echo probe $i; LAPLACE_HIP_LIB=$L/liblaplace_hip_xp$i.so bash tools/kt.sh r04x/p$i "xmap|fixup" tools/exp_c4_xmap.py --rare 1 --n 3; done
