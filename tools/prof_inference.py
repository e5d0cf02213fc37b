#!/usr/bin/env python3
"""Where a ranker INFERENCE batch's time goes (run_submission.make_predictions on device-built evaluation samples, H&M shape at
1/4 scale by default): sampling alone, the model's eval forward alone, the per-customer top-k selection alone, and the loop."""
import os, sys, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as t
from laplace_amd import run_submission as RS, synthetic as S
from laplace_amd.data.device_sampler import DeviceGraphSampler
from laplace_amd.data.matching import PopularItemsMatcher
from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
from laplace_amd.utils.constants import Constants
from laplace_amd.utils.get_info import get_feature_info, select_properties

users, items, edges = 343_000, 26_400, 7_950_000
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
spec = S.SyntheticSpec(users, items, edges, seed=2, zipf_s=1.0, communities=32, community_mix=0.9)
hetero, users_adj, articles_adj = S.generate_hetero(spec, feature_signal=True)
cfg = SimpleNamespace(k=12, num_neighbors=64, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0, batch_size=BATCH,
                      num_gnn_layers=2, hidden_layer_size=128, encoder_layer_output_size=64, conv_agg_type="add", num_linear_layers=2,
                      heterogeneous_prop_agg_type="sum", batch_norm=True, p_dropout_edges=0.0, p_dropout_features=0.3)
dev = "cuda"
matchers = [PopularItemsMatcher.from_adjacency(articles_adj, 150)]
ev = DeviceGraphSampler(cfg, hetero, users_adj, articles_adj, device=dev, seed=4, train=False, matchers=matchers, shuffle=False)
first = ev.sample(t.arange(BATCH), step=0)
model = Encoder_Decoder_Model(get_SAGEConv_layers(2, 128, 64, "add"), get_linear_layers(2, 128, 128, 1), get_feature_info(hetero),
                              first.metadata(), True, "sum", True, 0.0, 0.3).to(dev)
model.initialize_encoder_input_size(first)
model.eval()
sync = t.cuda.synchronize
N = 60
eval_u = t.arange(BATCH * N)

def timed(f, n=N):
    f(0); sync(); t0 = time.perf_counter()
    for i in range(n):
        f(i)
    sync(); return 1e3 * (time.perf_counter() - t0) / n

bs = [None]
print("sampling (ev.sample, one batch at a time): %.3f ms / batch" % timed(lambda i: bs.__setitem__(0, ev.sample(eval_u[BATCH * i:BATCH * i + BATCH], step=i))))
b = bs[0]
x, eid, eli, el = select_properties(b)
from laplace_amd.ranker_native import NativeRankerForward
nf = NativeRankerForward(model)
with t.no_grad():
    got = nf.logits(x, eid, eli)
    print("native forward:", "declined: %s" % nf.declined if got is None else "taken", "| message-passing edges", int(b[Constants.edge_key].edge_index.shape[1]),
          "label edges", int(eli.shape[1]))
    if got is not None:
        print("native eval forward on a fixed batch: %.3f ms" % timed(lambda i: nf.logits(x, eid, eli)))
    print("eval forward on a fixed batch: %.3f ms" % timed(lambda i: model(dict(x), eid, eli)))
    print("make_predictions on a fixed batch (forward + selection + .cpu()): %.3f ms" % timed(lambda i: RS.make_predictions(model, [b], k=12, device=dev)))
    t0 = time.perf_counter()
    c, p = RS.make_predictions(model, (ev.sample(eval_u[BATCH * i:BATCH * i + BATCH], step=i) for i in range(N)), k=12, device=dev)
    sync(); print("the loop, one ev.sample per batch: %.3f ms / batch; nodes per batch %s" % (1e3 * (time.perf_counter() - t0) / N, {k: v.shape[0] for k, v in b.x_dict.items()}))
    ev.step = 0
    t0 = time.perf_counter()
    c2, p2 = RS.make_predictions(model, ev.iter_users(eval_u), k=12, device=dev)
    sync(); print("the loop, ev.iter_users (sampling pipelined): %.3f ms / batch = %.0f users/s; same predictions: %s" % (
        1e3 * (time.perf_counter() - t0) / N, BATCH * N / (time.perf_counter() - t0), bool(t.equal(p, p2))))
