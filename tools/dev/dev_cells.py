"""Developer aid (not a test): solve selected cells of the synthetic Andrews grid and print their counters."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
R = importlib.import_module("rac-2d_amd")
netfile = "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat"
idx = [int(a) for a in sys.argv[1:]] or [19233, 18437, 20, 8696, 15952]
net = R.Network("data/" + netfile)
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
cells = R.cells.andrews_grid()[idx]
p = R.default_params()
out = net.evol_solve_batch(p, cells, net.init_abundances(y0, cells))
st = out["stats"]
for k, i in enumerate(idx):
    print("cell %5d: T=%.1f n=%.2e NST=%d NFE=%d NJE=%d NLU=%d NERR=%d nrec=%d q=%d tf=%.3g" % (i, cells[k, 0], cells[k, 2], st[k, 0], st[k, 1], st[k, 2], st[k, 3], st[k, 4], st[k, 5], out["quality"][k], out["t_final"][k]))
