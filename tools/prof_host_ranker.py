"""cProfile of the ranker training loop's HOST side (small graph, device sampler, pipelined iterator)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from types import SimpleNamespace
from laplace_amd import synthetic as S
from laplace_amd.data.device_sampler import DeviceGraphSampler
from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
from laplace_amd.utils.get_info import get_feature_info, select_properties
dev = "cuda"
spec = S.SyntheticSpec(200_000, 50_000, 4_000_000, seed=2, zipf_s=1.0)
graph, users, articles = S.generate_hetero(spec)
cfg = SimpleNamespace(k=12, num_neighbors=64, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0, batch_size=24)
loader = DeviceGraphSampler(cfg, graph, users, articles, device=dev, seed=0)
t.manual_seed(0)
it = iter(loader)
first = next(it)
model = Encoder_Decoder_Model(get_SAGEConv_layers(2, 128, 64, "add"), get_linear_layers(2, 128, 128, 1),
                              get_feature_info(graph), first.metadata(), True, "sum", True, 0.2, 0.3).to(dev)
model.initialize_encoder_input_size(first.to(dev))
opt = t.optim.Adam(model.parameters(), lr=0.01, fused=True)  # one multi-tensor launch, same update
crit = t.nn.BCEWithLogitsLoss()
model.train()
from laplace_amd.ranker_step import FusedRankerStep
fused = None if os.environ.get("LAPLACE_RANKER_AUTOGRAD") == "1" else FusedRankerStep(model, opt)
def step(batch):
    x, ei, eli, y = select_properties(batch)
    if fused is not None:
        loss = fused.step(x, ei, eli, y)
        if loss is not None:
            return loss
    opt.zero_grad()
    loss = crit(model(x, ei, eli).view(-1), y)
    loss.backward()
    opt.step()
    return loss
for _ in range(10): step(next(it))
t.cuda.synchronize()
import time
def timed(tag, n=200):
    t0 = time.perf_counter()
    for _ in range(n): step(next(it))
    t1 = time.perf_counter(); t.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag}: host {1e3*(t1-t0)/n:.3f} ms/iter, with drain {1e3*(t2-t0)/n:.3f} ms/iter", flush=True)
timed("multithreaded autograd (default)")
with t.autograd.set_multithreading_enabled(False):
    timed("autograd in the calling thread  ")
timed("multithreaded autograd (default)")
with t.autograd.set_multithreading_enabled(False):
    timed("autograd in the calling thread  ")
# forward only / no-grad forward, to split the host cost
def fwd_only(batch):
    x, ei, eli, y = select_properties(batch)
    return crit(model(x, ei, eli).view(-1), y)
t0 = time.perf_counter()
for _ in range(200): fwd_only(next(it))
t.cuda.synchronize(); print(f"sampler + forward + loss only: {1e3*(time.perf_counter()-t0)/200:.3f} ms/iter")
t0 = time.perf_counter()
for _ in range(200): next(it)
t.cuda.synchronize(); print(f"sampler only: {1e3*(time.perf_counter()-t0)/200:.3f} ms/iter", flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step(next(it))
pr.disable(); t.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(40); print(s.getvalue()[:7000])
