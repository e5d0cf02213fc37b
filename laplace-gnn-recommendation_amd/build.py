"""Build recipe for liblaplace_hip.so (gfx950 only) and the C oracle.

hipcc cross-compiles without a GPU; objects are cached by source mtime so rebuilds are
incremental.  The shared library is built IN-TREE next to this file so that it travels
with a source snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB_PATH = os.path.join(PKG_DIR, "liblaplace_hip.so")
HEADER = os.path.join(ROOT, "include", "laplace_hip.h")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ARCH = "gfx950"
CXXFLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
    "-Wall", "-Wno-unused-function", "-Wno-unused-result",
    "-DNDEBUG",
]


# per-file additions.  topk.hip: the bf16 prefilter kernel (csrc/topk_prefilter.hpp) compares its accumulators with per-lane
# thresholds; with the accumulators in AGPRs (the allocator's default for an MFMA result when the kernel needs > 256 registers)
# every compared value costs a v_accvgpr_read first — 85 per item panel beside 96 MFMAs.  The VGPR form of the MFMA keeps the
# accumulators in VGPRs and the query fragments (MFMA operands only) in AGPRs: 480 -> 372 registers, no copies.  The other
# kernels of the file fit 256 VGPRs and compile to the same code either way.
EXTRA_FLAGS = {"topk.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    deps = [HEADER] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    return max(os.path.getmtime(d) for d in deps)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
        os.path.getmtime(src), _deps_mtime())
    if stale:
        cmd = [HIPCC, *CXXFLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip for gfx950 and link liblaplace_hip.so. Returns its path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    need_link = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs)
    if need_link:
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[build] {LIB_PATH} ({os.path.getsize(LIB_PATH)} bytes)")
    return LIB_PATH


def build_oracle(verbose: bool = False) -> None:
    """Compile oracle/'s C restatement (test infrastructure, never loaded by the product)."""
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"oracle build failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("[build] oracle ok")


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv, verbose=True)
    build_oracle(verbose=True)
