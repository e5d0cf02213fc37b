L=$PWD/laplace-gnn-recommendation_amd
bash tools/ab_c4_env.sh "scan-U4||" "scan-U8|LAPLACE_HIP_LIB=$L/liblaplace_hip_xsu8.so|" "hint-off|LAPLACE_X_RARE=0|" "hint-off-unpacked|LAPLACE_X_RARE=0 LAPLACE_SPMM_PACK=0|" "scan-U4-2||" "scan-U8-2|LAPLACE_HIP_LIB=$L/liblaplace_hip_xsu8.so|"
