"""A/B harness for SpMM variants on the C2 graph: python tools/ab_spmm.py VARIANT [BAND]
VARIANT = base | name of laplace-gnn-recommendation_amd/liblaplace_hip_<name>.so (tools/build_variant.sh);
BAND = columns per band of the split-row plan (0 = row-major; default: ops.DEFAULT_BAND)."""
import os, sys, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
variant = sys.argv[1]
band = int(sys.argv[2]) if len(sys.argv) > 2 else None
pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'laplace-gnn-recommendation_amd')
if variant != 'base':
    os.environ['LAPLACE_HIP_LIB'] = f'{pkg}/liblaplace_hip_{variant}.so'  # never overwrite the product library
import torch as t
from laplace_amd import ops, synthetic as S
from laplace_amd.interactions import Interactions
ei = S.generate(S.C2).to('cuda')
inter = Interactions(ei, S.C2.num_users, S.C2.num_items)
adj, _ = inter.adjacency('bipartite').gcn_normalized(False)
adj.plan = ops.build_spmm_plan(adj, band=band)
n, d = adj.n_rows, 128
g = t.Generator(device='cuda').manual_seed(1)
X = t.randn(n, d, device='cuda', generator=g) * 0.1
A = t.randn(n, d, device='cuda', generator=g) * 0.1
Y = t.empty(n, d, device='cuda'); Sx = t.empty(n, d, device='cuda')
for _ in range(3): ops.spmm(adj, X, Y=Y, addend=A, S=Sx, scale=0.5)
t.cuda.synchronize()
rows = t.cat([t.randint(0, n, (300,), device='cuda', generator=g), t.arange(S.C2.num_users, S.C2.num_users + 20, device='cuda')])
err = 0.0
for r in rows.tolist():
    b, e = int(adj.rowptr[r]), int(adj.rowptr[r + 1])
    ref = (adj.val[b:e].double()[:, None] * X[adj.col[b:e].long()].double()).sum(0)
    err = max(err, float((Y[r].double() - ref).abs().max()), float((Sx[r].double() - 0.5 * (A[r].double() + ref)).abs().max()))
ts = []
for _ in range(5):
    s, e = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.spmm(adj, X, Y=Y, addend=A, S=Sx, scale=0.5)
    e.record(); t.cuda.synchronize(); ts.append(s.elapsed_time(e) / 10)
print(variant, 'band', band, 'items', adj.plan.n_items, 'launch', adj.plan.struct.n_launch, 'spmm ms: min %.4f med %.4f' % (min(ts), sorted(ts)[2]), 'max err vs f64 %.2e' % err, 'checksum %.6f' % float(Y.double().sum()), flush=True)
