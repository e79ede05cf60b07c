"""Developer aid (not a test): where a configs[1] cell that the GPU leaves flagged stalls, and how accurate the device's unpivoted
LDU of P = I - gamma*J is at that state (against a dense pivoted solve in numpy)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
R = importlib.import_module("rac-2d_amd")
cellno = int(sys.argv[1]) if len(sys.argv) > 1 else 1054
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
cell = R.cells.synth_batch(10000)[cellno:cellno + 1]
p = R.default_params()
out = net.evol_solve_batch(p, cell, net.init_abundances(y0, cell), record=True)
nrr = int(out["stats"][0, R.S_NREC_REAL]); touts = out["touts"][0]
print("cell", cellno, "q", out["quality"][0], "tf %.4g" % out["t_final"][0], "NST", out["stats"][0, 0], "NJE", out["stats"][0, 2], "NERR", out["stats"][0, 4], "nrec_real", nrr)
nS = net.nSpecies
colptr, rowidx = net.jac_pattern()
for rec in (nrr - 1, max(nrr - 20, 1), nrr // 2):
    y = out["record"][0, rec, :nS].copy()
    J = net.ode_jac(p, cell, y[None, :])[0]
    Jd = np.zeros((nS, nS))
    for j in range(nS):
        sl = slice(colptr[j] - 1, colptr[j + 1] - 1)
        Jd[rowidx[sl] - 1, j] = J[sl]
    f = net.ode_f(p, cell, y[None, :])[0]
    print("record %d t=%.4g  min y %.3e  #neg %d  |f|max %.3e" % (rec, touts[rec], y.min(), (y < 0).sum(), np.abs(f).max()))
    for gamma in (1e-2, 1e0, 1e2, 1e4, 1e6):
        P = np.eye(nS) - gamma * Jd
        b = gamma * f
        x_np = np.linalg.solve(P, b)
        x_gpu = net.newton_solve(p, cell, y[None, :], gamma, b[None, :])[0]
        # weighted like the integrator: ewt = rtol*|y| + atol
        ewt = 1e-4 * np.abs(y) + 1e-30
        err = np.sqrt(np.mean(((x_gpu - x_np) / ewt) ** 2)); nrm = np.sqrt(np.mean((x_np / ewt) ** 2))
        res = np.abs(P @ x_gpu - b).max() / (np.abs(P) @ np.abs(x_gpu) + np.abs(b)).max()
        print("   gamma %.0e: cond %.2e  wrms(x_gpu - x_dense) %.3e  wrms(x) %.3e  backward err %.2e" % (gamma, np.linalg.cond(P), err, nrm, res))
