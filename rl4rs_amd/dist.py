"""Multi-GPU plumbing: one process per GPU, env batches sharded by rank, no collective on the env path.

Every env row is independent (rl4rs/env/base.py:157-170 has no cross-row term), so rank r owns its own batch
of envs, its own log shard / RNG stream and a replica of catalogue + weights.  The only collectives are
(i) the barrier + max-over-ranks timing of bench.py and (ii) the policy-gradient all-reduce of a training
loop (``allreduce_mean_``), both over ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo" on CPU).
"""
import os


def dist_env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init(backend=None):
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_seed(base_seed, rank):
    """Per-rank seed of the synthetic log / sampling stream (seed = 1000 + rank)."""
    return int(base_seed) + int(rank)


def shard_rows(n_rows, rank, world):
    """Contiguous row block [lo, hi) of rank ``rank`` when ONE batch of n_rows envs is split (strong scaling)."""
    base, rem = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the timed region of bench.py)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def world_size():
    """Size of the initialised process group (1 when torch.distributed is not in use)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def allreduce_mean_(flat):
    """In-place mean all-reduce of ONE flat fp32 gradient buffer (a single fused collective per optimiser step:
    the mask-model gradient is 140 KB, latency-bound, so bucketing would only add launches)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    return flat


def barrier():
    import torch
    import torch.distributed as dist
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
