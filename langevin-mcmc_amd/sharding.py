"""Chain sharding across ranks: contiguous global chain-id ranges, seeds stay global
(RNG(chainId + seedOffset), mlt.cpp:61-62), one all-reduce of the film + the normalisation scalar."""
import numpy as np


def chain_range(rank, world, chains_per_rank):
    """[begin, end) of the global chain ids resident on `rank` (weak scaling: fixed chains per rank)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * chains_per_rank, (rank + 1) * chains_per_rank


def split_total(total_chains, world):
    """Strong-scaling split of a fixed number of chains: first (total % world) ranks get one more."""
    base, extra = divmod(total_chains, world)
    out, b = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((b, b + n))
        b += n
    return out


def group_ranges(total_chains, world):
    """The partition lmc_group_chains_init (and a job of RCCL ranks that follows the same rule) gives a fixed total: rank r holds the
    global chain ids [total r / world, total (r + 1) / world) -- host/context.cpp lmc_group_chains_init."""
    return [(total_chains * r // world, total_chains * (r + 1) // world) for r in range(world)]


def allreduce_film(film, normalization, dist, device=None):
    """Sums the per-rank indirect film buffers (the only data-path collective) and checks that every rank used
    the same normalisation scalar.  `dist` is torch.distributed (backend nccl == RCCL on ROCm, gloo on CPU)."""
    import torch

    t = torch.from_numpy(np.ascontiguousarray(film, np.float32))
    n = torch.tensor([float(normalization)], dtype=torch.float64)
    if device is not None:
        t, n = t.to(device), n.to(device)
    dist.all_reduce(t)
    nmax = n.clone()
    dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.MIN)
    if float(nmax.item()) != float(n.item()):
        raise RuntimeError("ranks disagree on the normalisation scalar: %r vs %r" % (n.item(), nmax.item()))
    return t.cpu().numpy(), float(n.item())
