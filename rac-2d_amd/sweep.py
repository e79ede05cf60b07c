"""Cell-sweep sharding: the caller-side partition of a grid of cells over the GPUs of one node.

Cells are independent initial-value problems once their input records are frozen,
so a sweep is a static partition plus ONE gather of the results -- end-state abundances, t_final, quality and the
per-cell counters travel together in one block per rank (RCCL over xGMI when the process group is "nccl";
"gloo" in the CPU tests).  No other collective is on the path.
Reference seam: the serial loop over cells in do_chemical_stuff, reference src/disk.f90:864-938.
"""
import numpy as np


def partition(ncell, world, rank):
    """Contiguous block [lo, hi) of rank; block sizes differ by at most one cell."""
    base, rem = divmod(ncell, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def interleaved_order(cost, world):
    """Permutation that deals the cells round-robin over ``world`` ranks by descending expected cost (e.g. n_gas, or the
    step count of the previous global iteration): after it, the contiguous blocks of ``partition`` hold cells
    k, k + world, k + 2 world, ... of the cost ranking, so every rank gets the same mix of expensive and cheap cells
    (steps per cell vary 2-3x with density and temperature, a few cells by 10x)."""
    idx = np.argsort(-np.asarray(cost, dtype=np.float64), kind="stable")
    return np.concatenate([idx[r::world] for r in range(world)])


def solve_sharded(solve_local, cells, y, dist=None, device=None, cost=None):
    """Solve the cells of this rank with ``solve_local(cells_block, y_block) -> dict(y, t_final, quality, stats)`` and gather.

    ``dist`` is torch.distributed (initialised) or None for a single process.  ``cost`` (optional, [ncell]) balances
    the ranks through ``interleaved_order``; without it the partition is contiguous in the caller's cell order.
    Returns dict(y [ncell, nS], t_final [ncell], quality [ncell] int32, stats [ncell, NSTAT] int64) on every rank, in the
    original cell order."""
    import torch
    ncell = cells.shape[0]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = solve_local(cells, y)
        return dict(y=np.asarray(out["y"]), t_final=np.asarray(out["t_final"]), quality=np.asarray(out["quality"], np.int32),
                    stats=np.asarray(out["stats"], np.int64))
    world, rank = dist.get_world_size(), dist.get_rank()
    order = interleaved_order(cost, world) if cost is not None else np.arange(ncell)
    lo, hi = partition(ncell, world, rank)
    mine = order[lo:hi]
    res = solve_local(np.ascontiguousarray(cells[mine]), np.ascontiguousarray(y[mine]))
    nS = y.shape[1]
    st = np.asarray(res["stats"], np.int64).reshape(hi - lo, -1)
    nstat = st.shape[1]
    ncol = nS + 2 + nstat
    maxn = -(-ncell // world)
    blk = np.zeros((maxn, ncol))
    blk[:hi - lo, :nS] = res["y"]
    blk[:hi - lo, nS] = res["t_final"]
    blk[:hi - lo, nS + 1] = res["quality"]
    blk[:hi - lo, nS + 2:] = st  # counters stay far below 2^53: exact in f64
    t_loc = torch.as_tensor(blk, dtype=torch.float64, device=device)
    out = torch.empty((world * maxn, ncol), dtype=torch.float64, device=device)
    if hasattr(dist, "all_gather_into_tensor") and device is not None:
        dist.all_gather_into_tensor(out, t_loc)
    else:
        dist.all_gather(list(out.view(world, maxn, ncol).unbind(0)), t_loc)
    out = out.view(world, maxn, ncol).cpu().numpy()
    full = np.empty((ncell, ncol))
    for r in range(world):
        a, b = partition(ncell, world, r)
        full[order[a:b]] = out[r, :b - a]
    return dict(y=np.ascontiguousarray(full[:, :nS]), t_final=full[:, nS].copy(), quality=full[:, nS + 1].astype(np.int32),
                stats=np.rint(full[:, nS + 2:]).astype(np.int64))


def solve_by_layers(solve, cells, y, layer, update=None, update_surface=True):
    """The caller's sweep in dependency order: the reference solves a cell only after the cells above it, because the
    self-shielding factors in its record are integrals over what those cells ended with (update_params_above_alt, reference
    src/disk.f90:1823-1883; the list of cells that may be solved next is kept by update_calculating_cells, :1937).  Here every
    LAYER is one batch: layer 0 (the surface) first; before a layer is solved as one batch ``update(k, idx, cells, y_done, done)``
    may rewrite its records ``cells[idx]`` from the end states ``y_done`` of all cells solved so far (``done``: their indices; empty
    for the surface layer, which is skipped with ``update_surface=False``).  Layers of a few hundred cells do not fill the GPU with one wave per
    cell; the engine hands them to four-wave teams by itself.

    solve(cells_block, y_block) -> dict(y, t_final, quality, stats) (e.g. a closure over Network.evol_solve_batch);
    cells [ncell, NPAR] (modified in place by ``update``), y [ncell, nS] start abundances, layer [ncell] ints.
    Returns dict(y, t_final, quality, stats) in the caller's cell order."""
    cells = np.asarray(cells)
    y_out = np.array(y, dtype=np.float64, copy=True)
    layer = np.asarray(layer)
    ncell = cells.shape[0]
    t_final = np.zeros(ncell)
    quality = np.zeros(ncell, np.int32)
    stats = None
    done = np.zeros(0, dtype=np.int64)
    for k in np.unique(layer):
        idx = np.nonzero(layer == k)[0]
        # (the surface layer too, with nothing above it: update_params_above_alt runs for every cell, column densities 0 there)
        if update is not None and (done.size or update_surface):
            update(int(k), idx, cells, y_out, done)
        res = solve(np.ascontiguousarray(cells[idx]), np.ascontiguousarray(y_out[idx]))
        y_out[idx] = res["y"]
        t_final[idx] = res["t_final"]
        quality[idx] = res["quality"]
        st = np.asarray(res["stats"], np.int64).reshape(idx.size, -1)
        if stats is None:
            stats = np.zeros((ncell, st.shape[1]), np.int64)
        stats[idx] = st
        done = np.concatenate([done, idx])
    return dict(y=y_out, t_final=t_final, quality=quality, stats=stats if stats is not None else np.zeros((0, 0), np.int64))
