"""Multi-GPU plumbing: one process per GPU, env batches sharded by rank, no collective on the env path.

Every env row is independent (rl4rs/env/base.py:157-170 has no cross-row term), so rank r owns its own batch
of envs, its own log shard / RNG stream and a replica of catalogue + weights.  The only collectives are
(i) the barrier + max-over-ranks timing of bench.py and (ii) the policy-gradient all-reduce of a training
loop (``allreduce_mean_``), both over ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo" on CPU).
"""
import os


# RL4RS_DIST_FORCE=1 (or set_force(True)): an initialised process group of ONE rank still runs every collective below instead of
# short-circuiting, and init() creates that one-rank group.  tests/test_gpu_nccl_one_rank.py executes the RCCL ("nccl")
# device-memory branch of every wrapper this way on a 1-GPU box.
FORCE_COLLECTIVES = os.environ.get('RL4RS_DIST_FORCE', '0') == '1'


def set_force(flag):
    global FORCE_COLLECTIVES
    FORCE_COLLECTIVES = bool(flag)


def collectives_active():
    """True when the wrappers below really enter a collective: a process group of more than one rank, or any initialised group
    while FORCE_COLLECTIVES is set."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or FORCE_COLLECTIVES


def dist_env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init(backend=None):
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if (world > 1 or FORCE_COLLECTIVES) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def ranks_sharing_device():
    """How many ranks of this node run on the SAME GPU as this one: 1 in the product configuration (one process per GPU over RCCL);
    more only in the dry runs that put several gloo ranks on one GPU (tests/test_gpu_bench_multirank.py: eight ranks on a 1-GPU
    box).  A persistent kernel with a grid barrier (k_ppo_pass) needs all of its workgroups resident at once, which a GPU shared by
    several processes does not promise: the trainer divides the kernel's co-residency budget by this number."""
    import torch
    _, _, world = dist_env()
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev <= 0 or local_world <= n_dev:
        return 1
    return -(-local_world // n_dev)


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
    if not collectives_active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=None if dist.get_backend() == 'gloo' else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not collectives_active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=None if dist.get_backend() == 'gloo' else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(value, device=None):
    """Every rank's python float, in rank order, on every rank (bench.py: per-rank rates beside the max-over-ranks time)."""
    import torch
    import torch.distributed as dist
    if not collectives_active():
        return [float(value)]
    W = dist.get_world_size()
    dev = None if dist.get_backend() == 'gloo' else device
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    out = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(W)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def world_size():
    """Size of the initialised process group (1 when torch.distributed is not in use)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def rank():
    """Rank in the initialised process group (0 when torch.distributed is not in use)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    return 0


def _needs_host_staging(t):
    """gloo moves host memory: a CUDA tensor goes through a pinned host copy (2 ranks on ONE GPU in the GPU tests, CPU
    tests); RCCL ("nccl") works on device memory directly."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == 'gloo'


def allreduce_mean_(flat):
    """In-place mean all-reduce of ONE flat fp32 gradient buffer (a single fused collective per optimiser step:
    the mask-model gradient is 140 KB, latency-bound, so bucketing would only add launches)."""
    import torch.distributed as dist
    if not collectives_active():
        return flat
    if _needs_host_staging(flat):
        h = flat.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        flat.copy_(h.div_(dist.get_world_size()))
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    return flat


def allreduce_sum_(t):
    """In-place SUM all-reduce of a (small) device tensor, enqueued like any other collective: nothing is read on the host
    (PPO's KL statistic of a data-parallel train call: every rank later derives the same kl_coeff from the same sum)."""
    import torch.distributed as dist
    if not collectives_active():
        return t
    if _needs_host_staging(t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t, src=0):
    """In-place broadcast of rank ``src``'s tensor (initial parameters / optimiser state of a data-parallel trainer)."""
    import torch.distributed as dist
    if not collectives_active():
        return t
    if _needs_host_staging(t):
        h = t.detach().cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
        return t
    dist.broadcast(t, src=src)
    return t


LAST_ROWS_PATH = None          # 'sparse' / 'dense': which exchange the last allreduce_rows_mean_ call of this process took (tests)


_OVERFLOW = {}            # device -> bool scalar: a sparse-row exchange since the last take_row_overflow saw more distinct rows than its cap
ROW_OVERFLOW_MESSAGE = ("a minibatch touched more distinct embedding rows than the static cap of the sparse gradient exchange "
                        "(dist.calibrate_row_cap, measured on the first minibatch): the exchanged rows were NaN and have been stepped "
                        "into the tables - rebuild the trainer (a larger headroom, or cap=None for the id-slot count)")


def take_row_overflow(device):
    """The accumulated overflow flag of ``allreduce_rows_mean_`` on ``device`` (a device bool scalar, or None when no sparse exchange
    ran) - and reset it.  The trainers fold it into the status word of their deferred statistics, so an overflow raises at the next
    settle instead of leaving NaN tables behind silently."""
    import torch
    return _OVERFLOW.pop(torch.device(device) if not isinstance(device, torch.device) else device, None)


def calibrate_row_cap(ids, table_rows, headroom=4, floor=1024):
    """A static distinct-row bound for ``allreduce_rows_mean_`` measured ONCE on a trainer's first minibatch and agreed across the
    ranks: ``headroom`` x the largest distinct-id count any rank saw, rounded up to a multiple of ``floor``, at most the number
    of id slots / table rows.  (One host read of a device scalar + one MAX all-reduce, at start-up only; the id-slot count
    itself - 256 x 64 x 2 = 32768 for the raw-state policy's sequence table - made every W >= 2 fall back to the dense 51 MB
    all-reduce)  A later minibatch with more distinct rows than the cap turns the exchanged rows into NaN (loud)."""
    import torch
    distinct = int(torch.unique(ids.reshape(-1)).numel())
    most = int(round(max_over_ranks(float(distinct), device=ids.device if ids.is_cuda else None)))
    cap = ((headroom * max(most, 1) + floor - 1) // floor) * floor
    return int(min(cap, ids.numel(), table_rows))


def allreduce_rows_mean_(table_grad, ids, cap=None):
    """Mean all-reduce of a SPARSE-ROW gradient: ``table_grad`` [H, E] is zero outside the rows named by ``ids`` (int64, in
    [0, H), duplicates allowed) that this rank's minibatch touched (embedding tables of the raw-state policy: 2 x 100000 x 128
    floats of which a 256-sample minibatch touches a few thousand rows).  Ranks exchange (distinct ids, rows) with ONE
    all-gather each instead of all-reducing the 51 MB table; every rank then rebuilds the mean in the same fixed rank order,
    so the result is bit-identical across ranks.

    Nothing in here reads a device value on the host (no ``.item()``, no ``torch.unique`` - both drain the queue once per
    minibatch): the distinct ids are found by a fixed-shape sort + first-occurrence scatter into buffers of the STATIC size
    ``cap`` (upper bound on the distinct rows; default: the number of id slots, at most H), padded with -1.  More distinct
    rows than ``cap`` (only possible with a caller-supplied hint that is wrong) turn the exchanged rows into NaN: a loud
    failure instead of silently dropped rows.  ``cap`` (or, without it, ``ids.numel()``) must be the same on every rank -
    it sizes the all-gather buffers; a trainer's minibatch shape is.  When ``cap`` rows per rank are not a small part of the table (a static
    decision) the dense all-reduce is used."""
    import torch
    import torch.distributed as dist
    if not collectives_active():
        return table_grad
    W = dist.get_world_size()
    H, E = table_grad.shape
    ids = ids.reshape(-1)
    n = int(ids.numel())
    cap = min(H, n if cap is None else int(cap))
    global LAST_ROWS_PATH
    if n == 0 or cap * W * 2 > H:
        LAST_ROWS_PATH = 'dense'
        return allreduce_mean_(table_grad.view(-1)).view(H, E)
    LAST_ROWS_PATH = 'sparse'
    dev = table_grad.device
    srt, _ = ids.to(device=dev, dtype=torch.int64).sort()
    first = torch.ones(n, dtype=torch.bool, device=dev)
    first[1:] = srt[1:] != srt[:-1]
    pos = first.cumsum(0) - 1                                     # rank of every id among the distinct ones
    overflow = pos[-1] >= cap                                     # device scalar, never read on the host
    slot = torch.where(first & (pos < cap), pos, torch.full_like(pos, cap))          # repeats / overflow -> trash slot `cap`
    buf = torch.full((cap + 1,), -1, dtype=torch.int64, device=dev)
    buf.scatter_(0, slot, srt)
    my_ids = buf[:cap].contiguous()
    flagged = torch.cat([my_ids, overflow.to(torch.int64).reshape(1)])        # the rank's overflow bit travels with its ids
    valid = my_ids >= 0
    safe = my_ids.clamp_min(0)
    my_rows = table_grad.index_select(0, safe) * valid[:, None].to(table_grad.dtype)
    my_rows = torch.where(overflow, torch.full_like(my_rows, float('nan')), my_rows)
    stage = _needs_host_staging(table_grad)
    if stage:
        flagged, my_rows = flagged.cpu(), my_rows.cpu()
    all_ids = [torch.empty_like(flagged) for _ in range(W)]
    all_rows = [torch.empty_like(my_rows) for _ in range(W)]
    dist.all_gather(all_ids, flagged)
    dist.all_gather(all_rows, my_rows)
    table_grad.index_fill_(0, safe, 0.0)                          # (padding names row 0: zero already unless touched, and then listed)
    inv = 1.0 / W
    # the overflow flag is GLOBAL: every rank's bit rides behind its ids, so all ranks raise ROW_OVERFLOW_MESSAGE at the same
    # settle (a rank-local flag left the others carrying NaN tables into a collective the raising rank had already left - ADVICE r5)
    for r in range(W):                       # fixed order: every rank performs the same sequence of additions
        got = all_ids[r].to(dev)
        overflow = overflow | (got[cap] != 0)
        i_r = got[:cap].clamp_min(0)                              # padded entries carry zero rows: they add 0 to row 0
        table_grad.index_add_(0, i_r, all_rows[r].to(dev) * inv)
    _OVERFLOW[dev] = overflow if _OVERFLOW.get(dev) is None else (_OVERFLOW[dev] | overflow)
    return table_grad


def barrier():
    import torch
    import torch.distributed as dist
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if collectives_active():
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
