"""Host side of the NATIVE ranker training loop (one C call per iteration + the device sampler's iterator): who sets the pace?
Prints host ms / iteration (issue time, no sync) against the drained time, for the whole loop, the sampler alone and the step
alone on a fixed batch, then a cProfile of the loop.  usage: python3 tools/prof_host_native.py [--hm]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from types import SimpleNamespace
from laplace_amd import synthetic as S
from laplace_amd.data.device_sampler import DeviceGraphSampler
from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
from laplace_amd.ranker_native import NativeRankerStep
from laplace_amd.utils.constants import Constants
from laplace_amd.utils.get_info import get_feature_info
dev = "cuda"
spec = S.SyntheticSpec(1_371_980, 105_542, 31_800_000, seed=2, zipf_s=1.0) if "--hm" in sys.argv else S.SyntheticSpec(200_000, 50_000, 4_000_000, seed=2, zipf_s=1.0)
graph, users, articles = S.generate_hetero(spec)
cfg = SimpleNamespace(k=12, num_neighbors=64, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0, batch_size=24)
loader = DeviceGraphSampler(cfg, graph, users, articles, device=dev, seed=0)
t.manual_seed(0)
it = iter(loader)
first = next(it)
model = Encoder_Decoder_Model(get_SAGEConv_layers(2, 128, 64, "add"), get_linear_layers(2, 128, 128, 1),
                              get_feature_info(graph), first.metadata(), True, "sum", True, 0.0, 0.3).to(dev)
model.initialize_encoder_input_size(first.to(dev))
opt = t.optim.Adam(model.parameters(), lr=0.01)
model.train()
native = NativeRankerStep(model, opt)

def step(batch):
    lab = batch[Constants.edge_key]
    loss = native.step(batch.x_dict, batch.edge_index_dict, lab.edge_label_index, lab.edge_label)
    assert loss is not None, native.declined
    return loss

for _ in range(20): step(next(it))
t.cuda.synchronize()

def timed(tag, fn, n=300):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); t.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag:42s} host issue {1e3*(t1-t0)/n:.3f} ms/iter, drained {1e3*(t2-t0)/n:.3f} ms/iter", flush=True)

timed("loop: next(it) + native.step", lambda: step(next(it)))
timed("loop: next(it) + native.step", lambda: step(next(it)))
fixed = next(it)
timed("native.step on one fixed batch", lambda: step(fixed))
timed("next(it) alone (sampler iterator)", lambda: next(it))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step(next(it))
pr.disable(); t.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:6500])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(25); print(s.getvalue()[:5000])
