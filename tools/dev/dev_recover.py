"""Developer aid (not a test): the cells of the configs[1] batch (10 000 random cells, seed 20240601) that a single chem_evol_solve
pass leaves flagged, and what the caller's local-iteration loop (racgpu_calc_cells, nlocal_iter = 4) makes of them."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
R = importlib.import_module("rac-2d_amd")
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
cells = R.cells.synth_batch(10000)
p = R.default_params()
t0 = time.time()
out = net.evol_solve_batch(p, cells, net.init_abundances(y0, cells))
st = out["stats"]
print("single pass: wall %.1fs kernel %.0f ms, steps %d -> %.0f steps/s; flagged %d" % (time.time() - t0, out["kernel_ms"], st[:, 0].sum(), st[:, 0].sum() / out["kernel_ms"] * 1e3, (out["quality"] != 0).sum()))
print("cycles/cell percentiles 50/99/max (1e9):", np.percentile(st[:, 8], [50, 99, 100]) / 1e9, " sum/3072 %.2fe9" % (st[:, 8].sum() / 3072 / 1e9))
for c in (3414, 1324, 9741, 2412):
    print("round-1 cell %d: NST %d NJE %d NERR %d q %d tf %.3g" % (c, st[c, 0], st[c, 2], st[c, 4], out["quality"][c], out["t_final"][c]))
flag = np.nonzero(out["quality"] != 0)[0]
if len(flag):
    t0 = time.time()
    r = net.calc_cells(p, cells[flag], net.init_abundances(y0, cells[flag]), nlocal_iter=4)
    print("calc_cells on the %d flagged cells: %.1fs" % (len(flag), time.time() - t0))
    for k, c in enumerate(flag):
        print("cell %5d: single pass q %d tf %.3g NST %d | loop: iterations %d q %d tf %.3g NST %d" % (
            c, out["quality"][c], out["t_final"][c], st[c, 0], r["stats"][k, R.S_NITER], r["quality"][k], r["t_final"][k], r["stats"][k, 0]))
