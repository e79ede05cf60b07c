"""Developer probe: step-by-step trace of one grid64 cell.  GPU (RACGPU_DEBUG_TRACE) or oracle (ORC_TRACE):
    python tools/dev/trace_cell.py gpu 39 1e-8 4000 2> trace.txt      |     python tools/dev/trace_cell.py oracle 39 1e-8 2> trace.txt"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
which, ci, rtol = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
g = np.load(os.path.join(ROOT, "tests", "golden", "grid64_grain.npz"))
k = list(g["grid_idx"]).index(ci)
cell = g["cells"][k:k + 1]
netf = os.path.join(ROOT, "data", str(g["network_file"])); inif = os.path.join(ROOT, "data", str(g["initial_file"]))
if which == "gpu":
    os.environ["RACGPU_DEBUG_TRACE"] = sys.argv[4]
    R = importlib.import_module("rac-2d_amd")
    net = R.Network(netf)
    net.set_team_threshold(-1.0)  # one wave, no hand-over: the trace belongs to the wave that starts the cell
    y0 = net.load_initial_abundances(inif)
    p = R.default_params(); p.RTOL = rtol; p.max_runtime_allowed = 0.0
    out = net.evol_solve_batch(p, cell, net.init_abundances(y0, cell))
    print("gpu stats", out["stats"][0, :8], out["t_final"], out["quality"])
else:
    os.environ["ORC_TRACE"] = "1"
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_ctypes as O
    onet = O.Network(netf)
    y0 = onet.initial_abundances(inif)
    op = O.default_params(); op.RTOL = rtol
    o = onet.solve_cell(op, cell[0], y0)
    print("oracle", {kk: o[kk] for kk in o if kk not in ("y", "record", "touts")})
