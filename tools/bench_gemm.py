#!/usr/bin/env python3
"""fp32 MFMA GEMM (mi_gemm_f32) on the shapes the hot path produces: python tools/bench_gemm.py [VARIANT]
VARIANT = name of laplace-gnn-recommendation_amd/liblaplace_hip_<name>.so built by tools/build_variant.sh NAME "-D..." gemm."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
if variant != "base":   # selected through LAPLACE_HIP_LIB (laplace_amd/_lib.py): the product library is never overwritten
    os.environ["LAPLACE_HIP_LIB"] = os.path.join(ROOT, "laplace-gnn-recommendation_amd", f"liblaplace_hip_{variant}.so")
import torch as t
from laplace_amd import ops

# (m, n, k): top-K score block; ranker lin_l on a 128-user batch; its output layer; a 24-user batch; a square reference
for (m, n, k) in [(2621, 100000, 128), (169000, 128, 84), (169000, 64, 128), (32000, 128, 128), (4096, 4096, 4096)]:
    A, B, C = t.randn(m, k, device="cuda"), t.randn(n, k, device="cuda"), t.empty(m, n, device="cuda")
    for _ in range(3):
        ops.gemm(A, B, out=C)
    t.cuda.synchronize()
    s, e = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.gemm(A, B, out=C)
    e.record()
    t.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(variant, (m, n, k), "%.3f ms  %.1f TF/s" % (ms, 2 * m * n * k / ms / 1e9), flush=True)
