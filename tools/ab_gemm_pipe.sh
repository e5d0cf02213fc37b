A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 400 --warmup 50 --pipelined"
for i in 1 2 3; do
for v in nopipe now; do
  if [ $v = nopipe ]; then export LAPLACE_HIP_LIB=$PWD/laplace-gnn-recommendation_amd/liblaplace_hip_nopipe.so; else unset LAPLACE_HIP_LIB; fi
  python3 $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_iteration'],4), 'ms/iter', round(d['positive_edges_per_s']), 'pos-edges/s')"
done; done
