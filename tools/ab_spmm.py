import os, sys, time, json, shutil
sys.path.insert(0, '/root/repo')
variant = sys.argv[1]
pkg = '/root/repo/laplace-gnn-recommendation_amd'
if variant != 'base':
    shutil.copy(f'{pkg}/liblaplace_hip_{variant}.so', f'{pkg}/liblaplace_hip.so')
import torch as t
from laplace_amd import ops, synthetic as S
from laplace_amd.interactions import Interactions
ei = S.generate(S.C2).to('cuda')
inter = Interactions(ei, S.C2.num_users, S.C2.num_items)
adj, _ = inter.adjacency('bipartite').gcn_normalized(False)
n, d = adj.n_rows, 128
X = t.randn(n, d, device='cuda') * 0.1
Y = t.empty(n, d, device='cuda')
for _ in range(3): ops.spmm(adj, X, Y=Y)
t.cuda.synchronize()
ts = []
for _ in range(5):
    s, e = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.spmm(adj, X, Y=Y)
    e.record(); t.cuda.synchronize(); ts.append(s.elapsed_time(e) / 10)
print(variant, 'spmm ms: min %.4f med %.4f' % (min(ts), sorted(ts)[2]), 'checksum %.6f' % float(Y.double().sum()))
