"""Developer probe (GPU box): the engine's Jacobian against the oracle's (the reference's chem_ode_jac restated, bit-identical to it on the
fixtures) at mid-trajectory states of grid cells 39 and 2223 (tools/dev/states_tight.npz), entry by entry."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
R = importlib.import_module("rac-2d_amd")
S = np.load(os.path.join(ROOT, "tools", "dev", "states_tight.npz"))
G = np.load(os.path.join(ROOT, "tests", "golden", "rate06_grain.npz"))
net = R.Network(os.path.join(ROOT, "data", "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat"))
nS = net.nSpecies
colptr, rowidx = net.jac_pattern()
IA, JA = G["IA"], G["JA"]
p = R.default_params()
for ci in (39, 2223):
    cell = S["c%d_cell" % ci][None, :]
    for tm in (10.0, 100.0):
        y = S["c%d_t%g_y" % (ci, tm)]
        vals = net.ode_jac(p, cell, y[None, :])[0]
        ref = S["c%d_t%g_jac" % (ci, tm)]
        dref = {}
        for j in range(nS):
            for q in range(IA[j] - 1, IA[j + 1] - 1):
                if JA[q] <= nS:
                    dref[(JA[q], j + 1)] = ref[q]
        worst = []
        seen = set()
        for j in range(nS):
            for q in range(colptr[j] - 1, colptr[j + 1] - 1):
                key = (rowidx[q], j + 1); seen.add(key)
                r = dref.get(key, 0.0)
                d = abs(vals[q] - r)
                rel = d / max(abs(r), 1e-300) if r != 0 else (0.0 if vals[q] == 0 else np.inf)
                worst.append((rel, d, key, vals[q], r))
        missing = [(k, v) for k, v in dref.items() if k not in seen and v != 0.0]
        worst.sort(reverse=True)
        print("cell %d t=%g: entries %d, max rel diff %.2e, nonzero reference entries outside the engine's pattern: %d" % (ci, tm, len(worst), worst[0][0], len(missing)))
        for rel, d, key, v, r in worst[:6]:
            print("     (%s <- d/d %s) engine %.15e oracle %.15e rel %.2e   y_row %.2e y_col %.2e" % (net.names[key[0] - 1], net.names[key[1] - 1], v, r, rel, y[key[0] - 1], y[key[1] - 1]))
        for k, v in missing[:5]:
            print("     MISSING (%s <- %s) = %.6e" % (net.names[k[0] - 1], net.names[k[1] - 1], v))
        print("     negative abundances in the state: %d (min %.2e)" % (int((y < 0).sum()), y.min()))
