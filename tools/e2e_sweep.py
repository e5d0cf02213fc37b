#!/usr/bin/env python3
"""MAP@12 of the configs[3] chain against training length (one graph, several runs of tools/e2e_hm_scale.run)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_hm_scale as E


def main():
    from laplace_amd import synthetic as S
    users, items, edges = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (1_371_980, 105_542, 31_800_000)))
    spec = S.SyntheticSpec(users, items, edges, seed=2, zipf_s=1.0, communities=32, community_mix=0.9)
    graph = S.generate_hetero(spec, feature_signal=True)
    for lg, rk, lr in ((300, 300, 0.05), (1500, 300, 0.05), (300, 1500, 0.05), (1500, 1500, 0.05), (1500, 4000, 0.05), (4000, 1500, 0.02)):
        out = E.run(users=users, items=items, edges=edges, lightgcn_steps=lg, ranker_iters=rk, lightgcn_lr=lr, graph=graph)
        print(json.dumps({"lightgcn_steps": lg, "ranker_iters": rk, "lr": lr, "map": out["map_at_12"], "loss": out["ranker_loss_last"],
                          "lg_s": out["stage_s"]["lightgcn_train_s"], "rk_s": out["stage_s"]["ranker_train_s"]}), flush=True)


if __name__ == "__main__":
    main()
