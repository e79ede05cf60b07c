"""Developer aid: GPU record of one synthetic cell vs the oracle's, species-wise, around a given time."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
R = importlib.import_module("rac-2d_amd")
import oracle_ctypes as O
idx = int(sys.argv[1]); tprobe = float(sys.argv[2])
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
onet = O.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
cell = R.cells.synth_batch(10000)[idx:idx + 1]
p = R.default_params()
out = net.evol_solve_batch(p, cell, net.init_abundances(y0, cell), record=True)
o = onet.solve_cell(O.default_params(), cell[0], y0, record=True)
nS = net.nSpecies
tg, to = out["touts"][0], o["touts"]
for k in range(len(to)):
    if to[k] >= tprobe * 0.3 and to[k] <= tprobe * 1.2:
        a, b = out["record"][0][k][:nS], o["record"][k][:nS]
        neg_g = [(net.names[i], a[i]) for i in np.where(a < -1e-25)[0]]
        neg_o = [(net.names[i], b[i]) for i in np.where(b < -1e-25)[0]]
        m = np.abs(b) > 1e-12
        e = np.abs(a[m] - b[m]) / np.abs(b[m])
        w = np.argsort(-e)[:4]
        names = np.array(net.names)[m]
        print("k=%d t gpu %.6g oracle %.6g  max rel diff (|X|>1e-12) %.2e  worst %s   neg gpu %s  neg oracle %s" % (
            k, tg[k], to[k], e.max(), [(names[i], "%.3e" % b[m][i], "%.3e" % a[m][i]) for i in w], neg_g[:6], neg_o[:6]), flush=True)
