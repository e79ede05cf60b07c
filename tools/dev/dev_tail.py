"""Developer aid (not a test): configs[1] with and without four-wave teams for its costliest cells (racgpu_set_team_threshold),
and those cells on their own."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
R = importlib.import_module("rac-2d_amd")
net = R.Network("data/rate06_dipole_reformated_again_withoutgrain.dat")
y0 = net.load_initial_abundances("data/ini_abund_waterice_loMetal.dat")
allc = R.cells.synth_batch(10000)
p = R.default_params()

def run(cells, tag):
    t0 = time.time(); out = net.evol_solve_batch(p, cells, net.init_abundances(y0, cells)); t1 = time.time()
    print("%-34s %.2f s  kernel %.0f ms  team cells %d  total steps %d" % (tag, t1 - t0, net.last_kernel_ms(), net.last_team_cells(), out["stats"][:, 0].sum()))
    return out

SHORT = len(sys.argv) > 1 and sys.argv[1] == 'short'
HINTED_ONLY = len(sys.argv) > 1 and sys.argv[1] == 'hinted'
UNHINTED_ONLY = len(sys.argv) > 1 and sys.argv[1] == 'unhinted'
if SHORT:
    idx = [8262, 1054, 9248, 6485]
    cells = allc[idx]
    for frac in (-1.0, 1e-9):
        net.set_team_threshold(frac); net.set_cost_hints(np.ones(len(idx)))
        o3 = run(cells, "costliest alone, team frac %g" % frac)
        st = o3["stats"]
        for k, i in enumerate(idx):
            print("cell %5d: cyc=%.2fe9 | per LU %.2fM (scatter %.2f rect %.2f dense %.2f fin %.2f)" % (i, st[k, 8] / 1e9, st[k, 11] / st[k, 3] / 1e6, st[k, 13] / st[k, 3] / 1e6, st[k, 14] / st[k, 3] / 1e6, st[k, 15] / st[k, 3] / 1e6, (st[k, 11] - st[k, 13] - st[k, 14] - st[k, 15]) / st[k, 3] / 1e6))
    sys.exit(0)
if UNHINTED_ONLY:
    net.set_team_threshold(0.5)
    for rep in range(2):
        o1 = run(allc, "full batch, queue order, hand-over"); print("   parked", net.last_parked_cells())
    sys.exit(0)
net.set_team_threshold(-1.0)
out = run(allc, "full batch, queue order, no teams")
net.set_team_threshold(0.5)
o1 = run(allc, "full batch, queue order, hand-over")
print("   parked", net.last_parked_cells())
assert np.array_equal(o1["y"], out["y"]) and np.array_equal(o1["stats"][:, :8], out["stats"][:, :8])
cost = out["stats"][:, 8].astype(float)
idx = [int(i) for i in np.argsort(-cost)[:8]]
for frac in ((0.5, 0.5) if HINTED_ONLY else (-1.0, 0.0, 0.5)):
    net.set_team_threshold(frac); net.set_cost_hints(cost)
    o2 = run(allc, "full batch, hinted, team frac %g" % frac)
    print("   parked", net.last_parked_cells())
    assert np.array_equal(o2["y"], out["y"]) and np.array_equal(o2["stats"][:, :8], out["stats"][:, :8])
if HINTED_ONLY:
    sys.exit(0)
cells = allc[idx]
for frac in (0.0, 1e-9):
    net.set_team_threshold(frac); net.set_cost_hints(np.ones(len(idx)))
    o3 = run(cells, "8 costliest alone, team frac %g" % frac)
st = o3["stats"]
for k, i in enumerate(idx):
    print("cell %5d: NST=%d NLU=%d cyc=%.2fe9 (%.2f s) | rhs %.3f jac %.3f lu %.3f solve %.3f | per LU %.2fM (scatter %.2f rect %.2f dense %.2f fin %.2f) per jac %.2fM per solve %.3fM per f %.3fM" % (
        i, st[k, 0], st[k, 3], st[k, 8] / 1e9, st[k, 8] / 2.4e9, *[st[k, c] / st[k, 8] for c in (9, 10, 11, 12)], st[k, 11] / max(st[k, 3], 1) / 1e6,
        st[k, 13] / max(st[k, 3], 1) / 1e6, st[k, 14] / max(st[k, 3], 1) / 1e6, st[k, 15] / max(st[k, 3], 1) / 1e6, (st[k, 11] - st[k, 13] - st[k, 14] - st[k, 15]) / max(st[k, 3], 1) / 1e6,
        st[k, 10] / max(st[k, 2], 1) / 1e6, st[k, 12] / max(st[k, 1], 1) / 1e6, st[k, 9] / max(st[k, 1], 1) / 1e6))
