"""Developer aid (not a test): step trace of one synthetic cell on the GPU (stderr) for comparison with the oracle."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
R = importlib.import_module("rac-2d_amd")
idx = int(sys.argv[1]); nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
cell = R.cells.synth_batch(nbatch)[idx:idx + 1]
p = R.default_params()
out = net.evol_solve_batch(p, cell, net.init_abundances(y0, cell))
print("GPU:", out["stats"][0][:8].tolist(), out["quality"], out["t_final"])
