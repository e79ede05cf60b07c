"""Developer scan (not a test): per-cell statistics of a synthetic batch, saved for analysis."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
R = importlib.import_module("rac-2d_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
maxsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
cells = R.cells.synth_batch(n)
p = R.default_params(); p.max_steps_per_cell = maxsteps
t0 = time.time()
out = net.evol_solve_batch(p, cells, net.init_abundances(y0, cells))
dt = time.time() - t0
st = out["stats"]
print("n=%d wall %.1fs kernel %.0f ms  total steps %d  -> %.0f steps/s" % (n, dt, out["kernel_ms"], st[:, 0].sum(), st[:, 0].sum() / dt))
print("NST percentiles 50/90/99/max:", np.percentile(st[:, 0], [50, 90, 99, 100]))
print("cycles/cell percentiles 50/90/99/max (1e9):", np.percentile(st[:, 8], [50, 90, 99, 100]) / 1e9)
print("quality!=0:", (out["quality"] != 0).sum(), " NERR>0:", (st[:, 4] > 0).sum(), " tfinal<tmax:", (out["t_final"] < 1e6).sum())
worst = np.argsort(-st[:, 8])[:8]
for w in worst:
    print("cell %4d: T=%.1f Td=%.1f n=%.2e Av=%.3g G0=%.2e  NST=%d NFE=%d NJE=%d NLU=%d NERR=%d nrec=%d q=%d tf=%.3g fails=%d cyc=%.2fe9" % (
        w, cells[w, 0], cells[w, 1], cells[w, 2], cells[w, 12], cells[w, 15], st[w, 0], st[w, 1], st[w, 2], st[w, 3], st[w, 4], st[w, 5],
        out["quality"][w], out["t_final"][w], st[w, 7], st[w, 8] / 1e9))
np.savez_compressed("gpurun_out/scan_%d.npz" % n, cells=cells, stats=st, quality=out["quality"], t_final=out["t_final"])
