"""Developer aid: how many cells of three configs[1]-like batches (other seeds) end flagged, with the electron's rate taken from the charge
balance (default) and with the scatter's own value (RACGPU_NO_CHARGE_BALANCE=1)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
R = importlib.import_module("rac-2d_amd")
y0 = None
for seed in (11, 12, 13):
    net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
    y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
    cells = R.cells.synth_batch(10000, seed=seed)
    out = net.evol_solve_batch(R.default_params(), cells, net.init_abundances(y0, cells))
    st = out["stats"]
    bad = np.nonzero(out["quality"] != 0)[0]
    print("seed %d: flagged %d %s  steps %d  NJE %d  cells with NJE > NST/4: %d  kernel %.0f ms" % (seed, len(bad), list(bad[:8]), st[:, 0].sum(), st[:, 2].sum(), (st[:, 2] > st[:, 0] / 4).sum(), out["kernel_ms"]), flush=True)
    net.close()
