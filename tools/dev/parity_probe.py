"""Developer probe (GPU box): the 64 grid cells of tests/golden/grid64_grain.npz at RTOL 1e-4 and 1e-8, per-cell worst species against the
reference's runs of the same cells.  python tools/dev/parity_probe.py > gpurun_out/parity_probe.txt"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
R = importlib.import_module("rac-2d_amd")
g = np.load(os.path.join(ROOT, "tests", "golden", "grid64_grain.npz"))
net = R.Network(os.path.join(ROOT, "data", str(g["network_file"])))
y0 = net.load_initial_abundances(os.path.join(ROOT, "data", str(g["initial_file"])))
nS = net.nSpecies
names = net.names


def worst(y, ref):
    m = ref >= 1e-6
    e = np.zeros(nS); e[m] = np.abs(y[m] - ref[m]) / ref[m]
    return float(e.max()), int(e.argmax())


for rtol, key in ((1e-4, "yend"), (1e-8, "yend_tight")):
    p = R.default_params(); p.RTOL = rtol; p.max_runtime_allowed = 0.0  # (the reference runs of the fixture had their wall-clock guards off)
    charge = net.species_attrs()["charge"].astype(float); elH = net.species_elements()[:, 3].astype(float)
    yin = net.init_abundances(y0, g["cells"])
    out = net.evol_solve_batch(p, g["cells"], net.init_abundances(y0, g["cells"]))
    print("== RTOL %g" % rtol)
    for c in range(len(g["cells"])):
        ref = g[key][c][:nS]
        e, sp = worst(out["y"][c], ref)
        if rtol == 1e-4:
            fl, _ = worst(g["yend_ulp"][c][:nS], ref)
            sc = g["scalars"][c]
        else:
            fl, _ = worst(ref, g["yend_tighter"][c][:nS])
            sc = g["scalars_tight"][c]
        et, spt = worst(out["y"][c], g["yend_tighter"][c][:nS])
        ec = int(out["stats"][c, R.S_ERRCODES])
        print("   conservation: charge gpu %.2e ref %.2e | H nuclei drift gpu %.2e ref %.2e" % (charge @ out["y"][c], charge @ ref, elH @ out["y"][c] - elH @ yin[c], elH @ ref - elH @ yin[c]))
        print("cell %5d T %7.1f n %.2e  err %.2e (%-8s X=%.1e)  ref-own %.2e  vs-1e-10 %.2e (%s)  NERR gpu %d [%d %d %d %d] ref %d  NST %d NFE %d NJE %d NLU %d fail %d  q %d/%d  tf %s" % (
            g["grid_idx"][c], g["cells"][c, 0], g["cells"][c, 2], e, names[sp], ref[sp], fl, et, names[spt], out["stats"][c, R.S_NERR],
            ec & 0xffff, (ec >> 16) & 0xffff, (ec >> 32) & 0xffff, (ec >> 48) & 0xffff, sc[2], out["stats"][c, 0], out["stats"][c, 1], out["stats"][c, 2], out["stats"][c, 3], out["stats"][c, 7], out["quality"][c], sc[1],
            out["t_final"][c] == sc[0]))
