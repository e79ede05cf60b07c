"""Developer probe (not a test): one cell, increasing horizons, GPU vs oracle, with wall times."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
R = importlib.import_module("rac-2d_amd")
import oracle_ctypes as O
D = "data/"
net = R.Network(D + "rate06_dipole_reformated_again_withoutgrain.dat")
onet = O.Network(D + "rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances(D + "ini_abund_waterice_loMetal.dat")
cell = R.cells.make_cell(50.0, 40.0, 1e8, 5.0, 1e3)
for tmax in [float(x) for x in sys.argv[1:]] or [1e-6, 1e-3, 1.0]:
    p = R.default_params(); p.t_max = tmax; p.mxstep_per_interval = 500
    op = O.default_params(); op.t_max = tmax; op.mxstep_per_interval = 500
    t0 = time.time()
    out = net.evol_solve_batch(p, cell[None, :], net.init_abundances(y0, cell[None, :]), record=True)
    dt = time.time() - t0
    o = onet.solve_cell(op, cell, y0, record=True)
    nS = net.nSpecies
    m = o["y"][:nS] >= 1e-10
    err = np.max(np.abs(out["y"][0][m] - o["y"][:nS][m]) / o["y"][:nS][m])
    print("t_max %.1e: gpu %.2fs kernel %.1f ms  stats gpu %s oracle nst=%d nfe=%d nje=%d nlu=%d  q %d/%d nerr %d/%d tf %g/%g  err(X>=1e-10) %.2e" % (
        tmax, dt, out["kernel_ms"], out["stats"][0][:6].tolist(), o["nst"], o["nfe"], o["nje"], o["nlu"], out["quality"][0], o["quality"],
        out["stats"][0][4], o["nerr"], out["t_final"][0], o["t_final"], err), flush=True)
    # first record where they part ways
    rec = out["record"][0][:, :nS]; orec = o["record"][:, :nS]
    for k in range(rec.shape[0]):
        mm = orec[k] >= 1e-12
        e = np.max(np.abs(rec[k][mm] - orec[k][mm]) / orec[k][mm])
        if e > 1e-6:
            print("   first record differing >1e-6: k=%d t=%g/%g err %.2e" % (k, out["touts"][0][k], o["touts"][k], e), flush=True)
            break
