"""Locality experiments on the C2 propagate (round 2): python3 tools/exp_locality.py [--reorder none|cold|cold2] [--slices 1|2|4] [--n 10]
  --reorder cold : items renumbered by degree (hot first), users ordered by their coldest item
  --slices s     : the product is run as s launches over D/s-wide column slices (ldx stays D)
Run under rocprofv3 (kernel trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE) for per-kernel time and traffic."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('AB_LIB'):
    import shutil
    _pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'laplace-gnn-recommendation_amd')
    os.environ['LAPLACE_HIP_LIB'] = f"{_pkg}/liblaplace_hip_{os.environ['AB_LIB']}.so"  # never overwrite the product library
import numpy as np
import torch as t
from laplace_amd import ops, synthetic as S
from laplace_amd.interactions import Interactions

ap = argparse.ArgumentParser()
ap.add_argument('--reorder', default='none')
ap.add_argument('--slices', type=int, default=1)
ap.add_argument('--n', type=int, default=10)
ap.add_argument('--band', type=int, default=-1)
ap.add_argument('--pace', type=int, default=0, help='sweep time table: ticks (10 ns) per band')
args = ap.parse_args()
ops.SWEEP_PACE = args.pace

spec = S.C2
ei = S.generate(spec).numpy()
U, I = spec.num_users, spec.num_items
u, i = ei[0], ei[1]
if args.reorder != 'none':
    ideg = np.bincount(i, minlength=I)
    order = np.argsort(-ideg, kind='stable')         # hot items first
    inew = np.empty(I, dtype=np.int64); inew[order] = np.arange(I)
    i2 = inew[i]
    cold = np.zeros(U, dtype=np.int64)
    np.maximum.at(cold, u, i2)                        # coldest item of each user = largest new id
    if args.reorder == 'cold2':                       # tie-break by the second coldest
        key = u * I + i2
        ks = np.sort(key)
        uu, ii = ks // I, ks % I
        last = np.r_[uu[1:] != uu[:-1], True]
        second = np.zeros(U, dtype=np.int64)
        prev_same = np.r_[False, uu[1:] == uu[:-1]]
        idx_last = np.nonzero(last)[0]
        has2 = prev_same[idx_last]
        second[uu[idx_last[has2]]] = ii[idx_last[has2] - 1]
        uorder = np.lexsort((second, cold))
    else:
        uorder = np.argsort(cold, kind='stable')
    unew = np.empty(U, dtype=np.int64); unew[uorder] = np.arange(U)
    u, i = unew[u], i2
ei = t.from_numpy(np.stack([u, i])).to('cuda')
inter = Interactions(ei, U, I)
adj, _ = inter.adjacency('bipartite').gcn_normalized(False)
if args.band >= 0:
    adj.plan = ops.build_spmm_plan(adj, band=args.band)
print('graph ready; plan items', adj.plan.n_items, 'long rows', adj.plan.n_long_rows, flush=True)
n, d = adj.n_rows, 128
g = t.Generator(device='cuda').manual_seed(1)
X = t.randn(n, d, device='cuda', generator=g) * 0.1
A = t.randn(n, d, device='cuda', generator=g) * 0.1
Y = t.empty(n, d, device='cuda'); Sx = t.empty(n, d, device='cuda')
w = d // args.slices

def product():
    for s in range(args.slices):
        c = slice(s * w, (s + 1) * w)
        ops.spmm(adj, X[:, c], Y=Y[:, c], addend=A[:, c], S=Sx[:, c], scale=0.5)

for _ in range(2): product()
t.cuda.synchronize()
rows = t.cat([t.randint(0, n, (200,), device='cuda', generator=g), t.arange(U, U + 20, device='cuda')])
err = 0.0
for r in rows.tolist():
    b, e = int(adj.rowptr[r]), int(adj.rowptr[r + 1])
    ref = (adj.val[b:e].double()[:, None] * X[adj.col[b:e].long()].double()).sum(0)
    err = max(err, float((Y[r].double() - ref).abs().max()), float((Sx[r].double() - 0.5 * (A[r].double() + ref)).abs().max()))
ts = []
for _ in range(3):
    s_, e_ = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(args.n): product()
    e_.record(); t.cuda.synchronize(); ts.append(s_.elapsed_time(e_) / args.n)
print(f'reorder={args.reorder} slices={args.slices} band={args.band}: product ms min {min(ts):.4f} med {sorted(ts)[1]:.4f}  max err vs f64 {err:.2e}', flush=True)
