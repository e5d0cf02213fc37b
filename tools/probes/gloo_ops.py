"""Timing of the collectives of the PinSAGE data-parallel exchange under gloo with CUDA tensors (two ranks, one card)."""
import os, sys, time
import torch as t, torch.distributed as dist, torch.multiprocessing as mp

def work(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda"
    words = 4 + 2 * 1536 + 1536 * 64 + 96
    send = t.zeros(words, dtype=t.int32, device=dev)
    gathered = t.zeros(world, words, dtype=t.int32, device=dev)
    flat = t.zeros(13000, device=dev)
    def timed(name, fn, n=20):
        fn(); t.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        t.cuda.synchronize()
        if rank == 0: print(f"{name}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms", flush=True)
    timed("all_gather list of rows (int32 cuda)", lambda: dist.all_gather([gathered[r] for r in range(world)], send))
    timed("all_gather_into_tensor (int32 cuda)", lambda: dist.all_gather_into_tensor(gathered.view(-1), send))
    sf, gf = send.view(t.float32), gathered.view(t.float32)
    timed("all_gather_into_tensor (float32 view)", lambda: dist.all_gather_into_tensor(gf.view(-1), sf))
    timed("all_reduce flat small (float cuda)", lambda: dist.all_reduce(flat))
    hs, hg = send.cpu(), gathered.cpu()
    timed("all_gather_into_tensor (int32 cpu)", lambda: dist.all_gather_into_tensor(hg.view(-1), hs))
    dist.destroy_process_group()

if __name__ == "__main__":
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(work, args=(2, port), nprocs=2, join=True)
