"""The first backward product of the fused step on BASELINE configs[3] at N = 1, alone: the whole adjacency times the compact
batch gradient through x_map (131 072 sampled positive edges' users + their items + uniform negatives), one stream.
    python3 tools/exp_c4_xmap.py [--rare 0|1] [--n 5] [--blocks B]
Run under rocprofv3 --kernel-trace --stats for the per-kernel times (spmm_items_xmap_kernel / spmm_items_kernel<…, true>,
spmm_fixup_kernel<…, true, …>, spmm_rows_kernel<…, true, …>)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from laplace_amd import ops, synthetic as S
from laplace_amd.interactions import Interactions

ap = argparse.ArgumentParser()
ap.add_argument('--rare', type=int, default=1)
ap.add_argument('--n', type=int, default=5)
ap.add_argument('--blocks', type=int, default=S.C4_BLOCKS)
ap.add_argument('--batch', type=int, default=131072)
ap.add_argument('--streams', type=int, default=0)
ap.add_argument('--check', type=int, default=0)
args = ap.parse_args()
ops.SPMM_TWO_STREAMS = args.streams
spec = S.C4
ei = S.generate_blocks(spec, S.C4_BLOCKS, 0, args.blocks).to('cuda')
U, I = spec.num_users * args.blocks // S.C4_BLOCKS, spec.num_items
inter = Interactions(ei, U, I)
inter = inter.permuted(inter.locality_order())
adj, _ = inter.adjacency('bipartite').gcn_normalized(False)
g = t.Generator(device='cuda').manual_seed(1)
pick = t.randint(0, ei.shape[1], (args.batch,), device='cuda', generator=g)
del ei
ei2 = inter.edge_index if hasattr(inter, 'edge_index') else None
rows = ops.expand_rows(adj)                      # row of every entry; the first half of the entries are the user rows'
e = pick % int(adj.rowptr[U])                    # entries of user rows: (user, item + U)
users, pos = rows[e].long(), adj.col[e].long() - U
neg = t.randint(0, I, (args.batch,), device='cuda', generator=g)
n, d = adj.n_rows, 128
gmap, nodes, cnt = ops.batch_nodes(users, pos, neg, U, n)
adj.plan = ops.build_spmm_plan(adj)
p = adj.plan
print(f'plan items {p.n_items} long rows {p.n_long_rows} band {int(p.struct.band)}; batch nodes {cnt.tolist()}', flush=True)
Xc = t.randn(3 * args.batch, d, device='cuda', generator=g) * 0.1
out = t.empty(n, d, device='cuda')

def product():
    ops.spmm(adj, Xc, addend=Xc, S=out, x_map=gmap, addend_map=gmap, x_rare=bool(args.rare))

if args.check:
    outs = []
    for rare in (0, 1, 1):
        o = t.full((n, d), float('nan'), device='cuda')
        ops.spmm(adj, Xc, addend=Xc, S=o, x_map=gmap, addend_map=gmap, x_rare=bool(rare))
        t.cuda.synchronize()
        outs.append(o)
    deg = (adj.rowptr[1:] - adj.rowptr[:-1])
    for name, a_, b_ in (('rare vs plain', outs[1], outs[0]), ('rare vs rare', outs[2], outs[1])):
        bad = ((a_ != b_) & ~(a_.isnan() & b_.isnan())).any(1).nonzero().view(-1)
        print(f'{name}: {bad.numel()} rows differ', flush=True)
        if bad.numel():
            dd = deg[bad]
            print('  first rows', bad[:8].tolist(), 'degrees', dd[:8].tolist(), 'min/max degree', int(dd.min()), int(dd.max()),
                  'max |diff|', float((a_[bad] - b_[bad]).abs().max()), flush=True)
            r = int(bad[0]); li = int(adj.plan.long_index[r]) if adj.plan.long_index is not None else -1
            if li >= 0:
                sb, se = int(adj.plan.item_ptr[li]), int(adj.plan.item_ptr[li + 1])
                print('  row', r, 'slots', sb, se, flush=True)
product()
t.cuda.synchronize()
ev = [t.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(args.n):
    product()
ev[1].record()
t.cuda.synchronize()
print(f'rare {args.rare}: {ev[0].elapsed_time(ev[1]) / args.n:.3f} ms per product, checksum {float(out.double().sum()):.6e}', flush=True)
