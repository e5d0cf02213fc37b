"""One process per GPU without an external launcher: the parent starts N workers BEFORE it touches the GPU (an exec or a
fork after HIP initialisation takes the machine down on this pool), hands rank 0's result line through and fails if any
worker fails.  Workers find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, exactly as
under torch.distributed.run."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Optional


def launched() -> bool:
    return "WORLD_SIZE" in os.environ


def self_launch(script: str, argv: List[str], n: int, result_marker: str = '"metric"', timeout_s: float = 900.0) -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line: Optional[str] = None
    rc = 0
    try:
        for raw in procs[0].stdout:
            txt = raw.decode(errors="replace").rstrip("\n")
            if txt.startswith("{") and result_marker in txt:
                line = txt
            elif txt:
                print(txt, file=sys.stderr, flush=True)
        deadline = time.time() + timeout_s
        for p in procs:
            try:
                p.wait(timeout=max(1.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                rc = rc or 124
            rc = rc or (p.returncode or 0)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if line is None:
        print(f"{os.path.basename(script)}: rank 0 produced no result line", file=sys.stderr)
        return rc or 1
    print(line, flush=True)
    return rc


def init_distributed():
    """(rank, world, device) of a worker; initialises torch.distributed when world > 1 (backend nccl = RCCL, or
    LAPLACE_BENCH_BACKEND=gloo with LAPLACE_BENCH_ONE_GPU=1 to rehearse on a one-GPU box)."""
    import torch as t
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("LAPLACE_BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    t.cuda.set_device(local)
    dev = t.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LAPLACE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    return rank, world, dev
