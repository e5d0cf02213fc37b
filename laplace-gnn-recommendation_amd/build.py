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


# ---- host-side sanitizer build ("-fsanitize=address host build of the C++ layer") --------------------------------
# The HOST half of every csrc/*.hip — the C ABI, the native executors' descriptor walks (COUNT / CHECK / LAUNCH passes), the
# plan builders' host code — compiled with AddressSanitizer + UndefinedBehaviorSanitizer (-Xarch_host: the gfx950 device code
# is compiled as in the product; GPU sanitizers are not available on this pool).  A separate library, never loaded by the
# product; tests/test_host_sanitizers.py runs tests/asan_driver.py against it on the CPU (LD_PRELOAD = the ASan runtime).
ASAN_OBJ_DIR = os.path.join(PKG_DIR, "build", "asan")
ASAN_LIB_PATH = os.path.join(PKG_DIR, "build", "liblaplace_hip_asan.so")
ASAN_FLAGS = [f"--offload-arch={ARCH}", "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DNDEBUG",
              "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer"]


def asan_runtime() -> str:
    """Path of the shared ASan runtime of hipcc's clang (to LD_PRELOAD into the python that loads the library)."""
    clang = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    r = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = r.stdout.strip()
    if r.returncode != 0 or not os.path.isabs(path) or not os.path.exists(path):
        raise RuntimeError(f"ASan runtime not found ({clang}: {r.stdout} {r.stderr})")
    return path


def build_asan(force: bool = False, verbose: bool = False) -> str:
    """liblaplace_hip_asan.so: host code under ASan + UBSan, device code as in the product.  Returns its path."""
    os.makedirs(ASAN_OBJ_DIR, exist_ok=True)
    dep = _deps_mtime()

    def one(src: str) -> str:
        obj = os.path.join(ASAN_OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep):
            cmd = [HIPCC, *ASAN_FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc (asan) failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, _sources()))
    if force or not os.path.exists(ASAN_LIB_PATH) or any(os.path.getmtime(o) > os.path.getmtime(ASAN_LIB_PATH) for o in objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan",
               "-o", ASAN_LIB_PATH, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link (asan) failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[build] {ASAN_LIB_PATH} ({os.path.getsize(ASAN_LIB_PATH)} bytes; host side under ASan + UBSan)")
    return ASAN_LIB_PATH


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
    if "--asan" in sys.argv:
        build_asan(force="--force" in sys.argv, verbose=True)
