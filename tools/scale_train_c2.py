"""The reference-shaped LightGCN pipeline (`run_pipeline_lightgcn.train`: split, three adjacencies, training iterations,
validation every 100, test) at BASELINE configs[1] scale — a scale rehearsal, not a benchmark.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
import torch as t
from laplace_amd import synthetic as S
from laplace_amd.config import lightgcn_config
from laplace_amd.run_pipeline_lightgcn import train
ei = S.generate(S.C2)
cfg = replace(lightgcn_config, epochs=101, k=12, hidden_layer_size=128, learning_rate=1e-3, batch_size=16384, num_iterations=3,
              eval_every=100, lr_decay_every=100, Lambda=1e-6, num_recommendations=256, save_model=False, show_graph=False)
t0 = time.perf_counter()
stats = train(cfg, edge_index=ei, num_users=S.C2.num_users, num_articles=S.C2.num_items, compat="bipartite", device="cuda", seed=0, verbose=True)
t.cuda.synchronize()
print(json.dumps({"total_s": round(time.perf_counter() - t0, 1), "stats": str(stats)}))
