"""Developer probe: step counts of selected grid64 cells against RTOL.  python tools/dev/nst_vs_rtol.py gpu|oracle"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
g = np.load(os.path.join(ROOT, "tests", "golden", "grid64_grain.npz"))
idx = list(g["grid_idx"]); sel = [idx.index(c) for c in (39, 351, 663, 1911, 2223, 2535, 4719, 7215, 9711, 12207)]
cells = g["cells"][sel]
netf = os.path.join(ROOT, "data", str(g["network_file"])); inif = os.path.join(ROOT, "data", str(g["initial_file"]))
if sys.argv[1] == "gpu":
    R = importlib.import_module("rac-2d_amd")
    net = R.Network(netf); y0 = net.load_initial_abundances(inif)
    for rtol in (1e-5, 1e-6, 1e-7, 1e-8):
        p = R.default_params(); p.RTOL = rtol; p.max_runtime_allowed = 0.0
        out = net.evol_solve_batch(p, cells, net.init_abundances(y0, cells))
        print("gpu rtol %g NST %s NJE %s" % (rtol, out["stats"][:, 0].tolist(), out["stats"][:, 2].tolist()))
else:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_ctypes as O
    onet = O.Network(netf); y0 = onet.initial_abundances(inif)
    for rtol in (1e-5, 1e-6, 1e-7, 1e-8):
        op = O.default_params(); op.RTOL = rtol
        r = [onet.solve_cell(op, c, y0) for c in cells]
        print("oracle rtol %g NST %s NJE %s" % (rtol, [o["nst"] for o in r], [o["nje"] for o in r]))
