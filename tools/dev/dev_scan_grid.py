"""Developer scan (not a test): per-cell statistics of the synthetic Andrews grid (BASELINE configs[2]), saved for analysis."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
R = importlib.import_module("rac-2d_amd")
netfile = sys.argv[1] if len(sys.argv) > 1 else "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat"
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1
net = R.Network("data/" + netfile)
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
cells = R.cells.andrews_grid()[::stride]
n = len(cells)
p = R.default_params()
t0 = time.time()
out = net.evol_solve_batch(p, cells, net.init_abundances(y0, cells))
dt = time.time() - t0
st = out["stats"]
print("n=%d wall %.1fs kernel %.0f ms  total steps %d  -> %.0f steps/s (kernel)" % (n, dt, out["kernel_ms"], st[:, 0].sum(), st[:, 0].sum() / out["kernel_ms"] * 1e3))
print("NST percentiles 50/90/99/max:", np.percentile(st[:, 0], [50, 90, 99, 100]))
print("cycles/cell percentiles 50/90/99/max (1e9):", np.percentile(st[:, 8], [50, 90, 99, 100]) / 1e9, " sum/2048 %.2fe9" % (st[:, 8].sum() / 2048 / 1e9))
print("quality!=0:", (out["quality"] != 0).sum(), " NERR>0:", (st[:, 4] > 0).sum(), " tfinal<tmax:", (out["t_final"] < cells[:, 27]).sum())
print("cycles per step %.3fM; phase shares rhs %.3f jac %.3f lu %.3f solve %.3f | lu: scatter %.3f lds/phases %.3f dense %.3f" % (st[:, 8].sum() / st[:, 0].sum() / 1e6, *[st[:, k].sum() / st[:, 8].sum() for k in (9, 10, 11, 12, 13, 14, 15)]))
worst = np.argsort(-st[:, 8])[:10]
for w in worst:
    print("cell %5d: T=%.1f Td=%.1f n=%.2e AvS=%.3g G0=%.2e tmax=%.2e NST=%d NFE=%d NJE=%d NLU=%d NERR=%d nrec=%d q=%d tf=%.3g cyc=%.2fe9" % (
        w, cells[w, 0], cells[w, 1], cells[w, 2], cells[w, 13], cells[w, 15], cells[w, 27], st[w, 0], st[w, 1], st[w, 2], st[w, 3], st[w, 4], st[w, 5],
        out["quality"][w], out["t_final"][w], st[w, 8] / 1e9))
np.savez_compressed("gpurun_out/scan_grid_%s_%d.npz" % (netfile[:20], n), cells=cells, stats=st, quality=out["quality"], t_final=out["t_final"])
