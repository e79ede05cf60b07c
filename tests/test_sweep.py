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


def _worker(rank, world, port, ncell, nS, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sw = _sweep()
    cells = np.arange(ncell * 28, dtype=np.float64).reshape(ncell, 28)
    y = np.arange(ncell * nS, dtype=np.float64).reshape(ncell, nS)
    full = sw.solve_sharded(lambda c, yy: yy * 2.0 + c[:, :1], cells, y, dist=dist, device=None)
    q.put((rank, full))
    dist.destroy_process_group()


def test_sharded_sweep_world2_gloo():
    ncell, nS, world = 11, 5, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ncell, nS, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cells = np.arange(ncell * 28, dtype=np.float64).reshape(ncell, 28)
    y = np.arange(ncell * nS, dtype=np.float64).reshape(ncell, nS)
    expect = y * 2.0 + cells[:, :1]
    for r in range(world):
        np.testing.assert_array_equal(res[r], expect)
