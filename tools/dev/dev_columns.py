"""Developer aid (not a test): the configs[2] grid in dependency order on the device (racgpu_column_sweep)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
R = importlib.import_module("rac-2d_amd")
net = R.Network("data/rate06_dipole_reformated_again_withgrain_lowH2Bind.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
ncol, nz = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), 100
grid = R.cells.andrews_grid(ncol=ncol, nz=nz)
ncell = grid.shape[0]
col_cells = np.concatenate([np.arange(c * nz, (c + 1) * nz)[::-1] for c in range(ncol)])  # andrews_grid runs upwards within a column
col_ptr = np.arange(ncol + 1) * nz
dz = np.full(ncell, 1e12)
p = R.default_params()
for rep in range(1):
    t0 = time.time(); out = net.column_sweep(p, grid, net.init_abundances(y0, grid), col_ptr, col_cells, dz); t1 = time.time()
    print("column sweep: %.2f s (kernel %.2f s), %d cell-steps -> %.0f cell-steps/s; flagged cells %d" % (
        t1 - t0, net.last_kernel_ms() / 1e3, out["stats"][:, 0].sum(), out["stats"][:, 0].sum() / (t1 - t0), (out["quality"] != 0).sum()), flush=True)
