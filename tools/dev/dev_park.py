"""Developer aid (not a test): a handful of cells with the end-of-pass hand-over to teams on and off."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
R = importlib.import_module("rac-2d_amd")
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
p = R.default_params(); p.t_max = 1e2
cells = R.cells.synth_batch(7, seed=5)
mode = sys.argv[1] if len(sys.argv) > 1 else "on"
net.set_team_threshold(-1.0 if mode == "off" else 0.5)
print("mode", mode, flush=True)
out = net.evol_solve_batch(p, cells, net.init_abundances(y0, cells))
print("done", out["stats"][:, 0], "parked", net.last_parked_cells(), flush=True)
