"""Developer aid: accuracy of the device LDU solve at a recorded state, against a dense pivoted solve."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
R = importlib.import_module("rac-2d_amd")
d = np.load("tools/dev/dev_state9565.npz")
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
p = R.default_params(); nS = net.nSpecies
y = d["y"][None, :]; cell = d["cell"][None, :]
colptr, rowidx = net.jac_pattern()
J = net.ode_jac(p, cell, y)[0]
Jd = np.zeros((nS, nS))
for j in range(nS):
    sl = slice(colptr[j] - 1, colptr[j + 1] - 1); Jd[rowidx[sl] - 1, j] = J[sl]
rng = np.random.default_rng(3)
f = net.ode_f(p, cell, y)[0]
for gamma in (1e-3, 1.0, 30.0, 60.0, 100.0, 1e3, 1e5):
    P = np.eye(nS) - gamma * Jd
    for name, b in (("rhs=h*f", gamma * f), ("random", rng.standard_normal(nS) * np.abs(y[0]))):
        x = net.newton_solve(p, cell, y, gamma, b[None, :])[0]
        xr = np.linalg.solve(P, b)
        res = np.abs(P @ x - b).max() / (np.abs(P) @ np.abs(x)).max()
        print("gamma %.0e %-8s  |x-xref|/|xref| = %.2e   backward err %.2e   cond(P) %.1e" % (
            gamma, name, np.linalg.norm(x - xr) / np.linalg.norm(xr), res, np.linalg.cond(P)), flush=True)
