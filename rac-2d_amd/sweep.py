"""Cell-sweep sharding: the caller-side partition of a grid of cells over the GPUs of one node.

Cells are independent initial-value problems once their input records are frozen,
so a sweep is a static block partition plus ONE gather of the end-state abundances (RCCL over xGMI when
the process group is "nccl"; "gloo" in the CPU tests).  No other collective is on the path.
Reference seam: the serial loop over cells in do_chemical_stuff, reference src/disk.f90:864-938.
"""
import numpy as np


def partition(ncell, world, rank):
    """Contiguous block [lo, hi) of rank; block sizes differ by at most one cell."""
    base, rem = divmod(ncell, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def interleaved_order(cost):
    """Permutation that deals cells round-robin by descending expected cost (e.g. n_gas), so that contiguous
    blocks of the permuted list carry similar work (steps per cell vary 2-3x with density and temperature)."""
    return np.argsort(-np.asarray(cost), kind="stable")


def solve_sharded(solve_local, cells, y, dist=None, device=None):
    """Solve rows [lo, hi) of (cells, y) with ``solve_local(cells_block, y_block) -> y_end_block`` and gather.

    ``dist`` is torch.distributed (initialised) or None for a single process.  Returns the full
    [ncell, nSpecies] end-state array on every rank, in the original cell order."""
    import torch
    ncell = cells.shape[0]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return solve_local(cells, y)
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = partition(ncell, world, rank)
    y_loc = solve_local(cells[lo:hi], y[lo:hi])
    nS = y.shape[1]
    maxn = -(-ncell // world)
    t_loc = torch.zeros((maxn, nS), dtype=torch.float64, device=device)
    t_loc[:hi - lo] = torch.as_tensor(np.ascontiguousarray(y_loc), dtype=torch.float64, device=device)
    out = torch.empty((world * maxn, nS), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t_loc) if hasattr(dist, "all_gather_into_tensor") and device is not None else \
        dist.all_gather(list(out.view(world, maxn, nS).unbind(0)), t_loc)
    out = out.view(world, maxn, nS).cpu().numpy()
    full = np.empty((ncell, nS))
    for r in range(world):
        a, b = partition(ncell, world, r)
        full[a:b] = out[r, :b - a]
    return full
