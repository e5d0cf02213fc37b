"""Host-side cost per launch of the op wrappers (tiny operands: the kernel is shorter than the call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from laplace_amd import ops
from laplace_amd.model.layers import Linear, SAGEConv, BipartiteGraph
dev = 'cuda'
A = t.randn(64, 64, device=dev); B = t.randn(64, 64, device=dev); C = t.empty(64, 64, device=dev)
ei = t.randint(0, 64, (2, 256), device=dev)
g = BipartiteGraph(ei, 64, 64)
a = ops.DeviceCSR(64, 64, g.by_dst.rowptr, g.by_dst.col, t.ones(256, device=dev), None)
def bench(name, fn, n=2000):
    for _ in range(50): fn()
    t.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); t.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:40s} host {1e6*(t1-t0)/n:7.2f} us/call   (+drain {1e6*(t2-t1)/n:6.2f})", flush=True)
bench("torch add", lambda: t.add(A, B, out=C))
bench("torch empty", lambda: t.empty(64, 64, device=dev))
bench("torch mm", lambda: t.mm(A, B, out=C))
bench("ops.gemm out=", lambda: ops.gemm(A, B, out=C))
bench("ops.gemm alloc", lambda: ops.gemm(A, B))
bench("ops.spmm Y=", lambda: ops.spmm(a, A, Y=C))
lin = Linear(64, 64).to(dev)
x = A.clone().requires_grad_(True)
bench("Linear fwd (autograd fn)", lambda: lin(x))
def fb():
    y = lin(x); y.sum().backward()
bench("Linear fwd+bwd", fb, 500)
conv = SAGEConv((64, 64), 64, aggr="add").to(dev)
xs = A.clone().requires_grad_(True); xd = B.clone().requires_grad_(True)
bench("SAGEConv fwd", lambda: conv((xs, xd), g), 500)
def cfb():
    y = conv((xs, xd), g); y.sum().backward()
bench("SAGEConv fwd+bwd", cfb, 300)
bn = t.nn.BatchNorm1d(64).to(dev)
def bnfb():
    y = bn(x); y.sum().backward()
bench("BatchNorm1d fwd+bwd", bnfb, 500)
