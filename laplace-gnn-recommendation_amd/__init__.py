"""laplace_amd — MI355X-native hot path of the laplace GNN link-prediction engine.

Layout mirrors the reference's flat module tree (config, model/, data/, utils/, training,
run_pipeline_lightgcn) so call sites read the same; the arithmetic is in csrc/*.hip behind
the C ABI of include/laplace_hip.h, loaded by `_lib`.  There is no CPU fallback: ops raise
if the library is missing or a tensor is not on the GPU.
"""
__version__ = "0.1.0"
