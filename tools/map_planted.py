"""MAP@12 on a planted-structure synthetic graph (synthetic.SyntheticSpec.communities): layer-0 predictor (the
reference's, F8) vs propagated embeddings vs the popularity predictor, at checkpoints of the training run.
    python3 tools/map_planted.py --users 50000 --items 5000 --edges 1000000 --batch 16384 --lr 0.01 --steps 50,100,200"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from dataclasses import replace
import bench
from laplace_amd import synthetic as S
from laplace_amd.interactions import Interactions
from laplace_amd.model.lightgcn import LightGCN
from laplace_amd.trainer import LightGCNTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=943)
ap.add_argument("--items", type=int, default=1682)
ap.add_argument("--edges", type=int, default=100000)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--layers", type=int, default=2)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--lr", type=float, default=1e-2)
ap.add_argument("--communities", type=int, default=8)
ap.add_argument("--mix", type=float, default=0.85)
ap.add_argument("--steps", default="0,50,100,200,400")
ap.add_argument("--eval-users", type=int, default=20000)
ap.add_argument("--deg-min", type=int, default=1)
args = ap.parse_args()
spec = S.SyntheticSpec(args.users, args.items, args.edges, seed=5, communities=args.communities, community_mix=args.mix,
                       deg_min=args.deg_min, deg_max=min(2000, args.items // 2))
ei = S.generate(spec)
held = S.heldout_edges(spec, ei, args.eval_users).to("cuda")
t.manual_seed(0)
model = LightGCN(args.users, args.items, args.dim, args.layers).to("cuda")
inter = Interactions(ei.to("cuda"), args.users, args.items)
tr = LightGCNTrainer(model, inter.adjacency("bipartite"), inter, lr=args.lr, Lambda=1e-6, batch_size=args.batch, seed=7)
done = 0
for cp in [int(x) for x in args.steps.split(",")]:
    out = bench.map_at_12(model, tr, inter, held, cp - done)
    done = cp
    print(cp, json.dumps({k: round(v, 4) if isinstance(v, float) else v for k, v in out.items() if "map" in k or k == "value"}), flush=True)
