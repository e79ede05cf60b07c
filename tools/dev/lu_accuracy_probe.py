"""Developer probe (GPU box): how accurate is the engine's pivot-free sparse LDU of P = I - gamma J at states where the GPU run needs
more steps than the reference (RTOL 1e-8, hot dense cells)?  For each probed cell: y = the reference's RTOL 1e-8 end state, J from the
engine's own Jacobian hook, b = gamma * f(y); x from racgpu_newton_solve against scipy's pivoted sparse LU, both measured in the
integrator's weighted RMS norm (weights 1 / (rtol |y| + atol)), plus one step of iterative refinement to see what it would buy.
    python tools/dev/lu_accuracy_probe.py > gpurun_out/lu_accuracy.txt"""
import importlib, os, sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
R = importlib.import_module("rac-2d_amd")
g = np.load(os.path.join(ROOT, "tests", "golden", "grid64_grain.npz"))
net = R.Network(os.path.join(ROOT, "data", str(g["network_file"])))
nS = net.nSpecies
colptr, rowidx = net.jac_pattern()
idx = list(g["grid_idx"])
p = R.default_params(); p.RTOL = 1e-8
dump = {}
for ci in (39, 351, 9711, 2223, 663, 12207):
    k = idx.index(ci)
    cell = g["cells"][k:k + 1]
    y = np.ascontiguousarray(g["yend_tight"][k][:nS])[None, :]
    vals = net.ode_jac(p, cell, y)[0]
    J = sp.csc_matrix((vals, rowidx - 1, colptr - 1), shape=(nS, nS))
    f = net.ode_f(p, cell, y)[0]
    rtol, atol = net.set_solver_flags_alt(p, 1, float(cell[0, R.cells.P_D2H]))
    w = 1.0 / (rtol[:nS] * np.abs(y[0]) + atol[:nS])
    wrms = lambda v: float(np.sqrt(np.mean((v * w) ** 2)))
    for gamma in (1e-2, 1.0, 1e2, 1e4, 1e6):
        P = sp.identity(nS, format="csc") - gamma * J
        b = gamma * f
        x = net.newton_solve(p, cell, y, gamma, b[None, :])[0]
        xs = spl.splu(P).solve(b)
        r = P @ x - b
        # one refinement step through the engine's own factors
        dx = net.newton_solve(p, cell, y, gamma, (-r)[None, :])[0]
        x2 = x + dx
        if ci in (39, 2223) and gamma in (1.0, 1e2, 1e4):
            dump["c%d_g%g" % (ci, gamma)] = np.concatenate([x, b])
            dump["c%d_J" % ci] = vals; dump["c%d_y" % ci] = y[0]; dump["c%d_w" % ci] = w
        print("cell %5d gamma %.0e  |x|w %.2e  |x - x_piv|w %.2e  |Px-b|w %.2e   refined: |x2 - x_piv|w %.2e  |Px2-b|w %.2e   pivoted |Px-b|w %.2e  max|x| %.2e" % (
            ci, gamma, wrms(xs), wrms(x - xs), wrms(r), wrms(x2 - xs), wrms(P @ x2 - b), wrms(P @ xs - b), np.max(np.abs(xs))))
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "lu_accuracy_dump.npz"), colptr=colptr, rowidx=rowidx, **dump)
