"""Propagate experiments on BASELINE configs[3] (C4: 8 M x 100 K, 100 M edges) at N = 1, under the product's locality order:
    python3 tools/exp_c4.py [--slices 1|2|4] [--band B] [--half users|items|both] [--n 5] [--edges E --users U]
  --slices s  : the product is run as s launches over D/s-wide column slices of X / Y (ldx stays D)
  --band B    : columns per band of the banded work-item plan (default: the product's)
  --half      : which row half of the adjacency is timed (the item rows = the split rows, the user rows = the short rows)
One-stream mode (kernels back to back).  Run under tools/prof3.sh for per-kernel time and FETCH / WRITE traffic."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from laplace_amd import ops, synthetic as S
from laplace_amd.interactions import Interactions

ap = argparse.ArgumentParser()
ap.add_argument('--slices', type=int, default=1)
ap.add_argument('--band', type=int, default=-1)
ap.add_argument('--half', default='both')
ap.add_argument('--n', type=int, default=5)
ap.add_argument('--blocks', type=int, default=S.C4_BLOCKS, help='user blocks of C4 to generate (64 = all of it)')
ap.add_argument('--chunk', type=int, default=ops.DEFAULT_CHUNK)
ap.add_argument('--pace', type=int, default=0, help='sweep time table: ticks (10 ns) per band')
ap.add_argument('--sweep-band', type=int, default=0)
ap.add_argument('--hybrid', type=int, default=-1, help='>= 0: hybrid plan (sweep for the hub rows + tail mode: 0 banded, 1 whole rows)')
args = ap.parse_args()
ops.SPMM_TWO_STREAMS = 0
ops.SWEEP_PACE = args.pace
if args.sweep_band:
    ops.SWEEP_BAND = args.sweep_band

spec = S.C4
ei = S.generate_blocks(spec, S.C4_BLOCKS, 0, args.blocks).to('cuda')
U, I = spec.num_users * args.blocks // S.C4_BLOCKS, spec.num_items
inter = Interactions(ei, U, I)
order = inter.locality_order()
inter = inter.permuted(order)
adj, _ = inter.adjacency('bipartite').gcn_normalized(False)
del ei
n, d = adj.n_rows, 128
a_users, a_items = ops.row_slice(adj, 0, U), ops.row_slice(adj, U, n)
parts = {'users': [a_users], 'items': [a_items], 'both': [adj]}[args.half]
for a in parts:
    if args.hybrid >= 0:
        a.plan = ops.build_hybrid_plan(a, chunk=args.chunk, tail_whole=bool(args.hybrid), band=None if args.band < 0 else args.band)
    elif args.band >= 0 or args.chunk != ops.DEFAULT_CHUNK:
        a.plan = ops.build_spmm_plan(a, chunk=args.chunk, band=None if args.band < 0 else args.band)
    else:
        a.plan = ops.build_spmm_plan(a)
    p = a.plan
    print(f'rows {a.n_rows}: plan items {p.n_items} long rows {p.n_long_rows} launch {int(p.struct.n_launch)} band {int(p.struct.band)} '
          f'sweep {p.sweep is not None}', flush=True)
g = t.Generator(device='cuda').manual_seed(1)
X = t.randn(n, d, device='cuda', generator=g) * 0.1
Y = t.empty(n, d, device='cuda')
w = d // args.slices
r0 = {'users': 0, 'items': U, 'both': 0}[args.half]

def product():
    for a in parts:
        for s in range(args.slices):
            c = slice(s * w, (s + 1) * w)
            ops.spmm(a, X[:, c], Y=Y[r0:r0 + a.n_rows, c])

for _ in range(2): product()
t.cuda.synchronize()
rows = t.cat([t.randint(r0, r0 + parts[0].n_rows, (100,), device='cuda', generator=g),
              t.arange(U, U + 20, device='cuda') if args.half != 'users' else t.arange(0, 20, device='cuda')])
err = 0.0
for r in rows.tolist():
    b, e = int(adj.rowptr[r]), int(adj.rowptr[r + 1])
    ref = t.zeros(d, dtype=t.float64, device='cuda')
    for lo in range(b, e, 1 << 20):
        hi = min(e, lo + (1 << 20))
        ref += (adj.val[lo:hi].double()[:, None] * X[adj.col[lo:hi].long()].double()).sum(0)
    err = max(err, float((Y[r].double() - ref).abs().max()))
ts = []
for _ in range(3):
    s_, e_ = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(args.n): product()
    e_.record(); t.cuda.synchronize(); ts.append(s_.elapsed_time(e_) / args.n)
print(f'C4 half={args.half} slices={args.slices} band={args.band} chunk={args.chunk} hybrid={args.hybrid} pace={args.pace} sweep_band={ops.SWEEP_BAND}: product ms min {min(ts):.4f} med {sorted(ts)[1]:.4f}  '
      f'max err vs f64 {err:.2e}', flush=True)
