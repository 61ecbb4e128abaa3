"""Chain-parallel multi-GPU layer: one process per GPU, chains sharded contiguously over ranks, no communication
while stepping (chains are independent: scripts/utils.py:351-363 runs them sequentially), and ONE gather of
samples per sampling segment over RCCL (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" on CPU tests)."""
import os
import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    rank, local_rank, world = env_rank()
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:  # CHMC_DIST_BACKEND=gloo: CPU-collective rehearsal of the multi-rank path
                backend = os.environ.get("CHMC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            kw = {}
            if backend == "nccl":
                dev = int(os.environ.get("CHMC_BENCH_DEVICE", local_rank))
                torch.cuda.set_device(dev)
                kw["device_id"] = torch.device("cuda", dev)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_chains(total_chains, rank, world):
    """Contiguous chain range [offset, offset + count) of this rank."""
    base, rem = divmod(total_chains, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def gather_samples(local, dst=0, equal_shards=False):
    """The single gather of a sampling segment: `local` is this rank's [B_local, n] sample block (numpy array or
    torch tensor on the rank's device).  Returns the [B_total, n] array on rank `dst`, None elsewhere.
    equal_shards=True (every rank holds the same number of chains, as in bench.py) skips the exchange of the shard
    sizes: the segment then costs exactly one collective."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.detach().cpu().numpy() if isinstance(local, torch.Tensor) else np.asarray(local)
    backend = dist.get_backend()
    t = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local))
    if backend == "nccl" and not t.is_cuda:
        t = t.cuda()
    if backend == "gloo" and t.is_cuda:
        t = t.cpu()
    world, rank = dist.get_world_size(), dist.get_rank()
    if equal_shards:
        counts = [t.shape[0]] * world
    else:
        counts = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
        counts = [int(c.item()) for c in counts]
    if len(set(counts)) == 1:  # equal shards: one gather collective
        out = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, out, dst=dst)
    else:  # ragged shards: pad to the largest
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        outp = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, outp, dst=dst)
        out = [o[:n] for o, n in zip(outp, counts)] if rank == dst else None
    if rank != dst:
        return None
    return torch.cat(out, 0).cpu().numpy()


def world_size():
    """World size read back from the initialised process group (1 without one)."""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def backend_name():
    """Collective backend in use ("nccl" is RCCL on ROCm); None for a single process."""
    import torch.distributed as dist
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def max_over_ranks(value):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values):
    import torch
    import torch.distributed as dist
    v = np.asarray(values, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return v
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(v, dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
