"""Developer aid (GPU box): time of the factorisation alone at full occupancy (3072 cells of the configs[2] grid, 9 factorisations each).
RACGPU_LIB selects the build."""
import os, sys, importlib, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
R = importlib.import_module("rac-2d_amd")
DATA = os.path.join(ROOT, "data")
net = R.Network(os.path.join(DATA, "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat"))
p = R.default_params()
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
cells = R.cells.andrews_grid()[::6][:ncell]
y0 = net.load_initial_abundances(os.path.join(DATA, "ini_abund_waterice_loMetal.dat"))
y = net.init_abundances(y0, cells)
f = net.ode_f(p, cells, y)
for mode in (1, 2):
    os.environ["RACGPU_DEBUG_REPEAT"] = "9"; os.environ["RACGPU_DEBUG_REPEAT_MODE"] = str(mode)
    net.newton_solve(p, cells, y, 1e6, f * 1e6)
