"""Developer probe (GPU box): how accurate is f(y) of the engine at states where its Newton iteration struggles at RTOL 1e-8?  States from
the oracle (tools/dev/states_tight.npz: cells 39 and 2223 of the grid at t = 10 and 100 yr).  f is recomputed in extended precision
(numpy longdouble) from the ENGINE'S OWN rate coefficients and compared with the engine's and the oracle's f, species by species, in
units of the sum of |flux| touching the species (the scale of the rounding error of any summation order)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
R = importlib.import_module("rac-2d_amd")
S = np.load(os.path.join(ROOT, "tools", "dev", "states_tight.npz"))
net = R.Network(os.path.join(ROOT, "data", "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat"))
nS = net.nSpecies
rx = net.reactions(); rows = net.reaction_rows()
p = R.default_params()
LD = np.longdouble
for ci in (39, 2223):
    cell = S["c%d_cell" % ci][None, :]
    k = net.cal_rates(p, cell)[0]
    nsite = cell[0, R.cells.P_D2H] * cell[0, R.cells.P_SITES]
    for tm in (10.0, 100.0):
        y = S["c%d_t%g_y" % (ci, tm)]
        yd_gpu = net.ode_f(p, cell, y[None, :])[0]
        yd_gpu2 = net.ode_f(p, cell, y[None, :])[0]
        yd_orc = S["c%d_t%g_ydot" % (ci, tm)][:nS]
        yl = y.astype(LD)
        exact = np.zeros(nS, LD); scale = np.zeros(nS, LD)
        for r in range(net.nReactions):
            it = rx["itype"][r]; a = rx["reac"][r, 0] - 1; b = rx["reac"][r, 1] - 1
            kk = LD(k[r])
            if it in (5, 6, 21, 64):
                f = kk * yl[a] * yl[b]
                if yl[a] < 0 and yl[b] < 0: f = -f
            elif it in (1, 2, 3, 13, 61, 20, 0):
                f = kk * yl[a]
            elif it in (62, 75):
                t1 = LD(nsite) * (LD(rows["ABC"][r, 2]) if it == 75 else LD(1))
                if t1 <= 0: f = kk
                else:
                    t = yl[a] / t1
                    f = kk * t if t <= 1e-4 else kk * (1 - np.exp(-t))
            elif it == 63:
                f = kk * yl[a] * yl[a]
                if yl[a] < 0: f = -f
            else:
                continue
            for s in range(rx["n_reac"][r]): exact[rx["reac"][r, s] - 1] -= f; scale[rx["reac"][r, s] - 1] += abs(f)
            for s in range(rx["n_prod"][r]): exact[rx["prod"][r, s] - 1] += f; scale[rx["prod"][r, s] - 1] += abs(f)
        sc = np.maximum(scale.astype(np.float64), 1e-300)
        eg = np.abs(yd_gpu - exact.astype(np.float64)) / sc; eo = np.abs(yd_orc - exact.astype(np.float64)) / sc
        print("cell %d t=%g: repeat call identical %s | engine: max err/scale %.2e (species %s) median %.1e | oracle (its own rates): max %.2e (species %s) median %.1e" % (
            ci, tm, np.array_equal(yd_gpu, yd_gpu2), eg.max(), net.names[int(eg.argmax())], np.median(eg), eo.max(), net.names[int(eo.argmax())], np.median(eo)))
        worst = np.argsort(-eg)[:5]
        for i in worst:
            print("     %-8s y %.3e ydot %.6e exact %.6e scale %.3e  err/scale engine %.2e oracle %.2e" % (net.names[i], y[i], yd_gpu[i], float(exact[i]), sc[i], eg[i], eo[i]))
