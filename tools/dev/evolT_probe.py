"""Developer aid: the evolT bench's exception cells, reference (here, CPU) against the GPU (on the box).
  here:     python tools/dev/evolT_probe.py ref 6747 10101 ...   -> tools/dev/evolT_probe_ref.npz
  GPU box:  python tools/dev/evolT_probe.py gpu                   (reads that file)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib
import bench as B
R = importlib.import_module("rac-2d_amd")
OUT = os.path.join(ROOT, "tools", "dev", "evolT_probe_ref.npz")
network, initial = B.NETWORKS["grain"], "ini_abund_waterice_loMetal.dat"
cells_h, r_au, z_au = R.cells.andrews_grid(return_geometry=True)
hc_h = R.cells.andrews_grid_hc(cells_h, r_au, z_au)
params = R.default_params()

if sys.argv[1] == "ref":
    idx = np.array([int(a) for a in sys.argv[2:]])
    ref, dt, cores = B.run_reference(cells_h[idx], network, initial, params, hc=hc_h[idx])
    np.savez(OUT, idx=idx, yend=np.array([r["yend"] for r in ref]), scalars=np.array([r["scalars"] for r in ref]),
             errcodes=np.array([r["errcodes"] for r in ref]))
    for k, r in enumerate(ref):
        print(idx[k], "scalars", r["scalars"], "T end", r["yend"][-1], "errcodes", r["errcodes"], "T0", cells_h[idx[k], 0], "nH", cells_h[idx[k], 2])
else:
    Z = np.load(OUT)
    idx = Z["idx"]
    net = R.Network(os.path.join(B.DATA, network))
    net.load_heating_cooling(B.DATA)
    y0 = net.load_initial_abundances(os.path.join(B.DATA, initial))
    g = net.evolT_solve_batch(params, cells_h[idx], hc_h[idx], net.init_abundances(y0, cells_h[idx]), record=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "evolT_probe_gpu.npz"), **{k: np.asarray(v) for k, v in g.items()})
    for k in range(len(idx)):
        st, co = g["stats"][k], g["cell_out"][k]
        yr = Z["yend"][k]; yg = np.r_[g["y"][k], co[R.O_TGAS]]
        m = yr >= 1e-6
        e = np.abs(yg[m] - yr[m]) / yr[m]
        w = np.nonzero(m)[0][np.argsort(-e)[:4]]
        print(idx[k], "ref scalars", Z["scalars"][k], "errc", Z["errcodes"][k], "| gpu t_final %.17g quality %d nerr %d (codes %x) nst %d T %.10g evolT_end %g | ref T %.10g"
              % (g["t_final"][k], g["quality"][k], st[R.S_NERR], st[R.S_ERRCODES], st[R.S_NST], co[R.O_TGAS], co[R.O_EVOLT_END], yr[-1]))
        rec, to = g["record"][k], g["touts"][k]
        nr = int(st[R.S_NREC_REAL])
        print("    T(t) every 10th record:", " ".join("%.3e:%.2f" % (to[i], rec[i, -1]) for i in range(0, nr, max(1, nr // 12))))
        print("    worst:", [(int(i), (net.names[i] if i < net.nSpecies else "T"), "%.3e" % (abs(yg[i] - yr[i]) / yr[i]), "%.4e" % yr[i]) for i in w])
