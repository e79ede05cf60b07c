"""Developer aid (GPU box, under rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE): the engine's phases one kernel launch each, on 3072 cells
of the configs[2] grid (the persistent kernel's wave count, so the same 1.2 GB footprint), so that the counters give bytes per phase call.
Launch order (the parser keys on it): k_rates, k_rhs, k_jac, k_newton x3 (1 LU + 1 solve; 9 LU + 1 solve; 1 LU + 9 solves)."""
import os, sys, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
R = importlib.import_module("rac-2d_amd")
DATA = os.path.join(ROOT, "data")
net = R.Network(os.path.join(DATA, "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat"))
p = R.default_params()
cells = R.cells.andrews_grid()[::6][:3072]
y0 = net.load_initial_abundances(os.path.join(DATA, "ini_abund_waterice_loMetal.dat"))
y = net.init_abundances(y0, cells)
n = len(cells)
net.cal_rates(p, cells)
f = net.ode_f(p, cells, y)
net.ode_jac(p, cells, y)
for rep, mode in ((1, 0), (9, 1), (9, 2)):
    os.environ["RACGPU_DEBUG_REPEAT"] = str(rep); os.environ["RACGPU_DEBUG_REPEAT_MODE"] = str(mode)
    net.newton_solve(p, cells, y, 1e6, f * 1e6)
print("phase_traffic done", n, "cells; nS %d nR %d nnzJ %d nzl %d nzu %d" % (net.nSpecies, net.nReactions, net.nnzJ, net.nzl, net.nzu))
