"""Lean SpMM launch loop on the C2 graph for rocprofv3 (kernel trace or --pmc passes):
python3 tools/prof_spmm.py [n_launches] [colsort]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('AB_LIB'):
    import shutil
    _pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'laplace-gnn-recommendation_amd')
    os.environ['LAPLACE_HIP_LIB'] = f"{_pkg}/liblaplace_hip_{os.environ['AB_LIB']}.so"  # never overwrite the product library
import torch as t
from laplace_amd import ops, synthetic as S
from laplace_amd.interactions import Interactions
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ei = S.generate(S.C2).to('cuda')
inter = Interactions(ei, S.C2.num_users, S.C2.num_items)
adj, _ = inter.adjacency('bipartite').gcn_normalized(False)
if os.environ.get('AB_BAND'):
    adj.plan = ops.build_spmm_plan(adj, band=int(os.environ['AB_BAND']))
print('graph ready', flush=True)
n, d = adj.n_rows, 128
X = t.randn(n, d, device='cuda') * 0.1
A = t.randn(n, d, device='cuda') * 0.1
Y = t.empty(n, d, device='cuda'); Sx = t.empty(n, d, device='cuda')
for _ in range(n_launch): ops.spmm(adj, X, Y=Y, addend=A, S=Sx, scale=0.5)
t.cuda.synchronize()
print('done', flush=True)
