"""Developer aid (not a test): cycles of one LU factorisation + one pair of triangular solves, in isolation.
RACGPU_DEBUG_REPEAT=R makes racgpu_newton_solve repeat both R times per cell and report per-part cycle counts."""
import importlib, sys, os
import numpy as np
os.environ.setdefault("RACGPU_DEBUG_REPEAT", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
R = importlib.import_module("rac-2d_amd")
d = np.load("tools/dev/dev_state9565.npz")
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
p = R.default_params(); nS = net.nSpecies
for ncell in [int(a) for a in sys.argv[1:]] or [256, 2048]:
    y = np.repeat(d["y"][None, :], ncell, 0); cell = np.repeat(d["cell"][None, :], ncell, 0)
    f = net.ode_f(p, cell[:1], y[:1])[0]
    b = np.repeat((30.0 * f)[None, :], ncell, 0)
    x = net.newton_solve(p, cell, y, 30.0, b)
    print(ncell, "cells: solutions identical across cells:", bool(np.all(x == x[0])), " |x|max %.3e" % np.abs(x[0]).max(), flush=True)
