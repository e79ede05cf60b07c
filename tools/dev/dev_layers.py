"""Developer aid (not a test): the configs[2] grid solved in the reference's dependency order (sweep.solve_by_layers, 100 layers
of 200 cells, H2/H2O/OH shielding towards the surface recomputed from the layers above) against the one-batch sweep."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
R = importlib.import_module("rac-2d_amd")
C = R.cells
net = R.Network("data/rate06_dipole_reformated_again_withgrain_lowH2Bind.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
ncol, nz = 200, 100
grid = C.andrews_grid(ncol=ncol, nz=nz)
ncell = grid.shape[0]
column = np.repeat(np.arange(ncol), nz); layer = (nz - 1) - np.tile(np.arange(nz), ncol)
dz = np.full(ncell, 1e12)
p = R.default_params()
iH2, iH2O, iOH = (net.species_index(nm) - 1 for nm in ("H2", "H2O", "OH"))

def update(k, idx, cells_, y_done, done):
    n = np.zeros(ncell)
    for sp, slot, f in ((iH2, C.P_FSS_ISM_H2, lambda N: C.h2_self_shielding(N, 1e5)),
                        (iH2O, C.P_FSS_ISM_H2O, lambda N: C.lya_self_shielding(N, C.LYA_CROSS_H2O)),
                        (iOH, C.P_FSS_ISM_OH, lambda N: C.lya_self_shielding(N, C.LYA_CROSS_OH))):
        n[:] = 0.0
        n[done] = cells_[done, C.P_NGAS] * y_done[done, sp]
        cells_[idx, slot] = f(C.column_density_above(n, dz, column, layer)[idx])

t0 = time.time(); one = net.evol_solve_batch(p, grid, net.init_abundances(y0, grid)); t1 = time.time()
print("one batch: %.2f s, %d cell-steps" % (t1 - t0, one["stats"][:, 0].sum()), flush=True)
kms = []
def solve(cb, yb):
    r = net.evol_solve_batch(p, cb, yb); kms.append(net.last_kernel_ms()); return r
t0 = time.time(); out = R.sweep.solve_by_layers(solve, grid.copy(), net.init_abundances(y0, grid), layer, update); t1 = time.time()
print("by layers: %.2f s (kernels %.2f s), %d cell-steps, %d layers; per layer min/median/max %.0f/%.0f/%.0f ms" % (
    t1 - t0, sum(kms) / 1e3, out["stats"][:, 0].sum(), len(kms), min(kms), np.median(kms), max(kms)))
