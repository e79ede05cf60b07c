"""Developer experiment (CPU): does the ORDER in which chem_ode_f's sums are taken decide whether a hot cell stalls?
The C restatement (oracle/, reference order) against the same restatement with the GPU engine's scatter order (ORC_F_ORDER=scatter),
on the hot cells (T > 500 K) of a configs[1]-like batch.  usage: oracle_f_order.py <scatter|reference> [ncells] [seed]"""
import os, sys, time
mode = sys.argv[1]
if mode == "scatter":
    os.environ["ORC_F_ORDER"] = "scatter"
import importlib
import numpy as np
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
ncell = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20240601


def work(args):
    import oracle_ctypes as O
    i, cell = args
    net = O.Network(os.path.join(ROOT, "data", "rate06_dipole_reformated_again_withoutgrain.dat"))
    y0 = net.initial_abundances(os.path.join(ROOT, "data", "ini_abund_waterice_loMetal.dat"))
    p = O.default_params()
    t0 = time.time()
    o = net.solve_cell(p, cell, y0)
    return i, o["nst"], o["nje"], o["quality"], o["t_final"], time.time() - t0


if __name__ == "__main__":
    R = importlib.import_module("rac-2d_amd")
    allc = R.cells.synth_batch(10000, seed=seed)
    hot = np.nonzero(allc[:, 0] > 500.0)[0][:ncell]
    with Pool(8) as pool:
        res = pool.map(work, [(int(i), allc[i]) for i in hot], chunksize=4)
    nst = np.array([r[1] for r in res]); nje = np.array([r[2] for r in res]); q = np.array([r[3] for r in res])
    bad = [(r[0], r[1], r[2], r[3], "%.3g" % r[4]) for r in res if r[3] != 0 or r[2] > r[1] / 4]
    print(mode, "cells", len(res), "flagged", int((q != 0).sum()), "NJE > NST/4:", int((nje > nst / 4).sum()), "total steps", int(nst.sum()), "total NJE", int(nje.sum()))
    print("  stalling:", bad[:12])
