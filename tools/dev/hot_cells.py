"""Developer aid: the configs[1] cells that stall on the GPU, steps and outcome; RACGPU_LIB selects the build."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
R = importlib.import_module("rac-2d_amd")
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
allc = R.cells.synth_batch(10000)
idx = [2054, 2957, 8262, 1054, 9248, 6485, 3038, 5313]
p = R.default_params()
out = net.evol_solve_batch(p, allc[idx], net.init_abundances(y0, allc[idx]))
st = out["stats"]
for k, i in enumerate(idx):
    print("cell %5d T %.0f n %.1e: NST %6d NFE %6d NJE %5d NLU %6d NERR %d quality %d t_final %.3g" % (i, allc[i, 0], allc[i, 2], st[k, 0], st[k, 1], st[k, 2], st[k, 3], st[k, 4], out["quality"][k], out["t_final"][k]))
