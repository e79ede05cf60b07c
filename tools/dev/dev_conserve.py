"""Developer aid: conservation statistics of a synthetic batch (see tests/test_gpu_fullsize.py)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import oracle_ctypes as O
R = importlib.import_module("rac-2d_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NET = "data/rate06_dipole_reformated_again_withoutgrain.dat"
net = R.Network(NET); onet = O.Network(NET)
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
cells = R.cells.synth_batch(n); p = R.default_params()
yin = net.init_abundances(y0, cells)
out = net.evol_solve_batch(p, cells, yin)
el = onet.elements.astype(float)
good = (out["quality"] == 0) & (out["t_final"] == p.t_max)
before = yin @ el; after = out["y"] @ el
tot = np.maximum(np.abs(yin) @ np.abs(el), np.abs(out["y"]) @ np.abs(el))
for e in range(el.shape[1]):
    if not np.any(el[:, e]): continue
    d = np.abs(after[good, e] - before[good, e])
    rel = d / np.maximum(tot[good, e], 1e-300)
    w = np.argmax(d)
    idx = np.flatnonzero(good)[w]
    print("col %2d: abs drift p50 %.1e p99 %.1e max %.1e | rel p99 %.1e max %.1e | worst cell %d NST %d NJE %d T %.0f n %.1e min(y) %.2e" % (
        e, np.percentile(d, 50), np.percentile(d, 99), d.max(), np.percentile(rel, 99), rel.max(), idx, out["stats"][idx, 0], out["stats"][idx, 2],
        cells[idx, 0], cells[idx, 2], out["y"][idx].min()))
