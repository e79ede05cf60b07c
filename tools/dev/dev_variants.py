"""Developer aid: ulp-variants of given synthetic cells on the GPU (is a slow cell intrinsically slow, or a rare event?)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
R = importlib.import_module("rac-2d_amd")
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
cells = R.cells.synth_batch(10000)
p = R.default_params(); p.max_steps_per_cell = 20000
for idx in [int(a) for a in sys.argv[1:]]:
    var = []
    for k in range(64):
        c = cells[idx].copy()
        for _ in range(k):
            c[2] = np.nextafter(c[2], np.inf)
        c[5] = c[2] * c[6]
        var.append(c)
    var = np.array(var)
    out = net.evol_solve_batch(p, var, net.init_abundances(y0, var))
    st = out["stats"]
    print(idx, "NST:", st[:, 0].tolist(), flush=True)
    print(idx, "NJE:", st[:, 2].tolist(), "quality", out["quality"].tolist(), flush=True)
