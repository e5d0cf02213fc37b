#!/usr/bin/env python3
"""mi_sage_wgrad_f32 against the grouped GEMM on the ranker's weight-gradient shapes (24 users / batch at the H&M shape:
~30 000 article rows, ~2 000 customer rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from laplace_amd import ops
from laplace_amd.model.layers import _run_products, _ones4

def bench(fn, n=50):
    for _ in range(5): fn()
    t.cuda.synchronize()
    s, e = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); t.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n

for name, shapes in (("layer 0 (m=128, 84 | 84)", [(30000, 128, 84, 84), (2100, 128, 84, 84)]),
                     ("layer 1 (m=64, 128 | 128)", [(30000, 64, 128, 128), (2100, 64, 128, 128)])):
    g = t.Generator(device="cuda").manual_seed(0)
    probs, specs = [], []
    for (k, m, n1, n2) in shapes:
        dy, mk = t.randn(k, m, device="cuda", generator=g), t.randn(k, m, device="cuda", generator=g)
        b1, b2 = t.randn(k, n1, device="cuda", generator=g), t.randn(k, n2, device="cuda", generator=g)
        gw1, gb, gw2 = t.empty(m, n1, device="cuda"), t.empty(m, device="cuda"), t.empty(m, n2, device="cuda")
        probs.append(dict(dy=dy, mask=mk, b1=b1, b2=b2, gw1=gw1, gb=gb, gw2=gw2))
        db4 = t.empty(m, 4, device="cuda")
        specs += [dict(A=dy, B=b1, out=t.empty(m, n1, device="cuda"), trans_a=True, trans_b=False, mask=mk),
                  dict(A=dy, B=_ones4(k, "cuda"), out=db4, trans_a=True, trans_b=False, mask=mk),
                  dict(A=dy, B=b2, out=t.empty(m, n2, device="cuda"), trans_a=True, trans_b=False, mask=mk)]
    assert ops.sage_wgrad(probs)
    _run_products(specs)
    t.cuda.synchronize()
    err = float((probs[0]["gw1"] - specs[0]["out"]).abs().max()) / float(specs[0]["out"].abs().max())
    print(f"{name}: wgrad {bench(lambda: ops.sage_wgrad(probs)):.1f} us, grouped GEMM {bench(lambda: _run_products(specs)):.1f} us, rel diff {err:.1e}", flush=True)
