"""One process per GPU without an external launcher: the parent starts N workers BEFORE it touches the GPU (an exec or a
fork after HIP initialisation takes the machine down on this pool), hands rank 0's result line through and fails if any
worker fails.  Workers find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, exactly as
under torch.distributed.run.

The parent never blocks on one child: rank 0's stdout is drained by a reader thread while the main loop polls every
worker against ONE deadline that starts at launch.  The first worker that exits non-zero (or the deadline) ends the job:
the others are killed and the parent returns non-zero at once — a rank that dies before the rendezvous can no longer leave
rank 0 (and the parent with it) waiting in a collective."""
from __future__ import annotations

import datetime
import os
import socket
import subprocess
import sys
import threading
import time
from typing import List, Optional, Sequence

INIT_TIMEOUT_S = 180.0   # rendezvous / any collective: a missing rank is an error after three minutes, not a hang


def launched() -> bool:
    return "WORLD_SIZE" in os.environ


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def supervise(cmds: Sequence[Sequence[str]], envs: Sequence[dict], result_marker: str = '"metric"',
              timeout_s: float = 900.0, poll_s: float = 0.1, name: str = "launch") -> int:
    """Run cmds[r] under envs[r]; rank 0's stdout is scanned for the result line (a JSON object holding
    `result_marker`), which is printed once every worker has exited 0.  Returns 0, the first failing worker's exit
    code, or 124 on the deadline."""
    t_start = time.time()
    procs: List[subprocess.Popen] = []
    for r, (cmd, env) in enumerate(zip(cmds, envs)):
        procs.append(subprocess.Popen(list(cmd), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    found: List[str] = []

    def drain():
        for raw in procs[0].stdout:
            txt = raw.decode(errors="replace").rstrip("\n")
            if txt.startswith("{") and result_marker in txt:
                found.append(txt)
            elif txt:
                print(txt, file=sys.stderr, flush=True)

    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print(f"{name}: worker {r} exited with code {c}; stopping the other workers", file=sys.stderr, flush=True)
                rc = c if c > 0 else 128 - c   # a signal death (negative) reads as 128 + signo
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t_start > timeout_s:
                print(f"{name}: deadline of {timeout_s:.0f} s reached; stopping the workers", file=sys.stderr, flush=True)
                rc = 124
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
    reader.join(timeout=5)
    if rc:
        return rc
    if not found:
        print(f"{name}: rank 0 produced no result line", file=sys.stderr)
        return 1
    print(found[-1], flush=True)
    return 0


def self_launch(script: str, argv: List[str], n: int, result_marker: str = '"metric"', timeout_s: float = 900.0) -> int:
    port = _free_port()
    cmds, envs = [], []
    for r in range(n):
        envs.append(dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                         MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0"))
        cmds.append([sys.executable, os.path.abspath(script)] + list(argv))
    return supervise(cmds, envs, result_marker, timeout_s, name=os.path.basename(script))


def backend_name() -> str:
    """The torch.distributed backend a worker uses: "nccl" (= RCCL on ROCm) unless LAPLACE_BENCH_BACKEND says gloo."""
    return os.environ.get("LAPLACE_BENCH_BACKEND", "nccl")


def init_distributed():
    """(rank, world, device) of a worker; initialises torch.distributed when world > 1 (backend nccl = RCCL, or
    LAPLACE_BENCH_BACKEND=gloo with LAPLACE_BENCH_ONE_GPU=1 to rehearse on a one-GPU box).  The rendezvous and every
    collective carry a timeout of INIT_TIMEOUT_S: a rank that never arrives raises instead of hanging."""
    import torch as t
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("LAPLACE_BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    t.cuda.set_device(local)
    dev = t.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        to = datetime.timedelta(seconds=float(os.environ.get("LAPLACE_DIST_TIMEOUT_S", INIT_TIMEOUT_S)))
        if backend_name() == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=to)
        else:
            dist.init_process_group(backend_name(), timeout=to)
    return rank, world, dev
