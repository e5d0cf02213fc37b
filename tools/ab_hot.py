"""A/B of the LDS hot-row cache on the C2 propagate under the product's locality order (round 3):
    python3 tools/ab_hot.py [--configs "rows:threads:waves,..."] [--n 20]
rows = rows offered to the cache (-1 = plain launch, 0 = persistent launch without cache), threads = workgroup size, waves = wavefronts per CU (0 = default).
Prints the dense-launch time per configuration (HIP events, min / median of 5 x n launches) and checks bitwise equality
with the plain launch.  Run under rocprofv3 --kernel-trace --stats with ONE configuration for per-kernel times."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from laplace_amd import ops, synthetic as S
from laplace_amd.interactions import Interactions

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="-1:0:0,0:256:20,0:256:16,0:512:16,63:256:20,32:256:20,128:512:16,304:1024:16")
ap.add_argument("--n", type=int, default=20)
ap.add_argument("--config", default="c2")
args = ap.parse_args()
spec = S.C4 if args.config == "c4" else S.C2
ei = (S.generate_blocks(spec, S.C4_BLOCKS, 0, S.C4_BLOCKS) if args.config == "c4" else S.generate(spec)).to("cuda")
U, I = spec.num_users, spec.num_items
inter = Interactions(ei, U, I)
order = inter.locality_order()
inter = inter.permuted(order)
adj, _ = inter.adjacency("bipartite").gcn_normalized(False)
n, d = adj.n_rows, 128
g = t.Generator(device="cuda").manual_seed(1)
X = t.randn(n, d, device="cuda", generator=g) * 0.1
A = t.randn(n, d, device="cuda", generator=g) * 0.1
Y = t.empty(n, d, device="cuda"); Sx = t.empty(n, d, device="cuda")
ref = None
for cfg in args.configs.split(","):
    rows, threads, waves = (int(x) for x in cfg.split(":"))
    adj.hot = (U, rows) if rows > 0 else None
    ops.PERSISTENT_ROWS = rows >= 0
    ops.HOT_THREADS = threads | (waves << 12)
    for _ in range(3):
        ops.spmm(adj, X, Y=Y, addend=A, S=Sx, scale=0.5)
    t.cuda.synchronize()
    if ref is None:
        ref = (Y.clone(), Sx.clone())
    same = t.equal(Y, ref[0]) and t.equal(Sx, ref[1])
    ts = []
    for _ in range(5):
        s_, e_ = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(args.n):
            ops.spmm(adj, X, Y=Y, addend=A, S=Sx, scale=0.5)
        e_.record(); t.cuda.synchronize(); ts.append(s_.elapsed_time(e_) / args.n)
    print(f"hot rows {rows:4d} threads {threads:4d} waves/CU {waves:2d}: dense launch ms min {min(ts):.4f} med {sorted(ts)[2]:.4f}  bitwise == plain: {same}", flush=True)

# ---- where the time goes: user-row half and item-row half as separate launches (row slices with their own plans) ----
if os.environ.get("AB_HOT_SLICES", "1") == "1":
    a_u, a_i = ops.row_slice(adj, 0, U), ops.row_slice(adj, U, n)
    a_u.plan, a_i.plan = ops.build_spmm_plan(a_u), ops.build_spmm_plan(a_i)
    print("user slice: long rows", a_u.plan.n_long_rows, "items", a_u.plan.n_items, "| item slice: long rows", a_i.plan.n_long_rows,
          "items", a_i.plan.n_items, "sweep" if a_i.plan.sweep is not None else "work items", flush=True)

    def timed(a, lo, hi, label):
        for _ in range(3):
            ops.spmm(a, X, Y=Y[lo:hi], addend=A[lo:hi], S=Sx[lo:hi], scale=0.5)
        t.cuda.synchronize()
        if os.environ.get("AB_HOT_YONLY") == "1":   # one output stream, no addend: the forward layers 1..K-1 of the trainer
            for _ in range(3):
                ops.spmm(a, X, Y=Y[lo:hi])
            t.cuda.synchronize()
            ts = []
            for _ in range(5):
                s_, e_ = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
                s_.record()
                for _ in range(args.n):
                    ops.spmm(a, X, Y=Y[lo:hi])
                e_.record(); t.cuda.synchronize(); ts.append(s_.elapsed_time(e_) / args.n)
            print(f"  {label} [Y only]: ms min {min(ts):.4f} med {sorted(ts)[2]:.4f}", flush=True)
        ts = []
        for _ in range(5):
            s_, e_ = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(args.n):
                ops.spmm(a, X, Y=Y[lo:hi], addend=A[lo:hi], S=Sx[lo:hi], scale=0.5)
            e_.record(); t.cuda.synchronize(); ts.append(s_.elapsed_time(e_) / args.n)
        print(f"  {label}: ms min {min(ts):.4f} med {sorted(ts)[2]:.4f}", flush=True)

    for cfg in args.configs.split(","):
        rows, threads, waves = (int(x) for x in cfg.split(":"))
        a_u.hot = (U, rows) if rows > 0 else None
        ops.PERSISTENT_ROWS = rows >= 0
        ops.HOT_THREADS = threads | (waves << 12)
        timed(a_u, 0, U, f"user rows, hot {rows} threads {threads} waves {waves}")
        a_i.hot = None
        timed(a_i, U, n, f"item rows (short rows + split rows), persistent {rows >= 0} threads {threads} waves {waves}")
