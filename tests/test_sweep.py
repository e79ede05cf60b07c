"""Sharding of a cell sweep over ranks: partition arithmetic, and a world_size-2 gloo run of the gather."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _sweep():
    sys.path.insert(0, ROOT)
    return importlib.import_module("rac-2d_amd.sweep")


def test_partition_covers_everything_once():
    sw = _sweep()
    for ncell in (0, 1, 7, 8, 9, 20001):
        for world in (1, 2, 3, 8):
            blocks = [sw.partition(ncell, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == ncell
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_interleaved_order_deals_round_robin():
    sw = _sweep()
    rng = np.random.default_rng(3)
    for ncell, world in ((11, 2), (20000, 8), (5, 8)):
        cost = rng.lognormal(0.0, 1.5, ncell)
        order = sw.interleaved_order(cost, world)
        assert sorted(order.tolist()) == list(range(ncell))  # a permutation
        sums = []
        for r in range(world):
            lo, hi = sw.partition(ncell, world, r)
            mine = order[lo:hi]
            sums.append(cost[mine].sum())
            # rank r holds ranks r, r + world, ... of the cost ranking
            ranking = np.argsort(-cost, kind="stable")
            np.testing.assert_array_equal(mine, ranking[r::world])
        if ncell >= 1000:
            assert max(sums) / min(sums) < 1.1  # balanced, unlike a contiguous split of the sorted list


def _fake_solve(c, yy):
    n = len(c)
    return dict(y=yy * 2.0 + c[:, :1], t_final=c[:, 1] + 0.5, quality=(c[:, 0] % 3).astype(np.int32),
                stats=(np.arange(n * 20).reshape(n, 20) + c[:, :1].astype(np.int64)))


def _worker(rank, world, port, ncell, nS, use_cost, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sw = _sweep()
    cells = np.arange(ncell * 28, dtype=np.float64).reshape(ncell, 28)
    y = np.arange(ncell * nS, dtype=np.float64).reshape(ncell, nS)
    cost = np.cos(np.arange(ncell)) if use_cost else None

    def local(c, yy):  # stats depend on the cell only (not on the position in the block), so the gather can be checked
        out = _fake_solve(c, yy)
        out["stats"] = np.repeat(c[:, :1].astype(np.int64), 20, axis=1) * 7 + np.arange(20)
        return out
    full = sw.solve_sharded(local, cells, y, dist=dist, device=None, cost=cost)
    q.put((rank, full))
    dist.destroy_process_group()


@pytest.mark.parametrize("use_cost", [False, True])
def test_sharded_sweep_world2_gloo(use_cost):
    ncell, nS, world = 11, 5, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ncell, nS, use_cost, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cells = np.arange(ncell * 28, dtype=np.float64).reshape(ncell, 28)
    y = np.arange(ncell * nS, dtype=np.float64).reshape(ncell, nS)
    for r in range(world):  # every rank holds everything, in the caller's cell order
        np.testing.assert_array_equal(res[r]["y"], y * 2.0 + cells[:, :1])
        np.testing.assert_array_equal(res[r]["t_final"], cells[:, 1] + 0.5)
        np.testing.assert_array_equal(res[r]["quality"], (cells[:, 0] % 3).astype(np.int32))
        np.testing.assert_array_equal(res[r]["stats"], np.repeat(cells[:, :1].astype(np.int64), 20, axis=1) * 7 + np.arange(20))
