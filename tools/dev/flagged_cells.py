"""Developer aid: which cells of configs[1] end flagged, and the costliest ones."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
R = importlib.import_module("rac-2d_amd")
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
allc = R.cells.synth_batch(10000)
p = R.default_params()
out = net.evol_solve_batch(p, allc, net.init_abundances(y0, allc))
st = out["stats"]
bad = np.nonzero(out["quality"] != 0)[0]
top = np.argsort(-st[:, 8])[:8]
for i in list(bad) + [t for t in top if t not in bad]:
    print("cell %5d T %.0f Td %.0f n %.2e: NST %6d NFE %6d NJE %5d NLU %6d NERR %d quality %d t_final %.4g cyc %.2fe9" % (i, allc[i, 0], allc[i, 1], allc[i, 2], st[i, 0], st[i, 1], st[i, 2], st[i, 3], st[i, 4], out["quality"][i], out["t_final"][i], st[i, 8] / 1e9))
print("kernel ms", out["kernel_ms"], "total steps", st[:, 0].sum())
