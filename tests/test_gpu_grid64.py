"""GPU parity on the bench workload itself: 64 cells of the configs[2] grid (every fourth cell of bench.py's 256-cell parity sample)
against the reference's end states for exactly these cells (tests/golden/grid64_grain.npz, made by tests/golden/make_golden.py
grid64 from the unmodified reference):

  yend          the reference at the template settings (RTOL 1e-4)
  yend_ulp      the same with n_gas moved by ONE ulp: that cell's own rounding-noise floor
  yend_tight    the reference at RTOL 1e-8
  yend_tighter  the reference at RTOL 1e-10: how converged its own RTOL 1e-8 answer is (median 7e-9, max 2.5e-6 over these cells)

Bounds (species with X >= 1e-6):
  * RTOL 1e-4: BASELINE.json's bar 1e-4, or 3x the cell's own floor where the reference itself moves by more than that;
  * RTOL 1e-8: 5e-6 + 3x the reference's own 1e-8 <-> 1e-10 distance on that cell (both runs sit within their truncation error of
    the exact solution; the reference's is up to 2.5e-6 here);
  * t_final and quality equal; NERR within what the reference's 1-ulp twin shows against the reference itself (up to 2 per
    cell, all of them ISTATE -4 / -5: which step fails its error test is decided at rounding level).
"""
import numpy as np
import pytest

from conftest import DATA, load_golden, major_relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def grid64(racgpu):
    g = load_golden("grid64_grain")
    net = racgpu.Network(f"{DATA}/{g['network_file']}")
    y0 = net.load_initial_abundances(f"{DATA}/{g['initial_file']}")
    return g, net, y0


def _solve(racgpu, net, y0, cells, rtol):
    p = racgpu.default_params()
    p.RTOL = rtol
    return net.evol_solve_batch(p, cells, net.init_abundances(y0, cells))


def test_grid_cells_end_state_within_the_references_own_floor(racgpu, grid64):
    """The floor of a cell = the largest move of the reference's own end state under seven one-ulp changes of an input (n_gas +1, -1, +2
    ulp, Tgas +1, -1 ulp, zeta_CR +1 ulp: `yend_ulp` and `floor_twins` of the fixture; one twin alone underestimates a chaotic quantity:
    cell 10959 moves by 9e-6 under the first and by 7e-4 under the second).  Where either side took an ISTATE -4/-5 return the error
    policy has loosened the offending species' RTOL by up to 1e-3 (ode_solver_error_handling, reference src/chemistry.f90:297-377): the
    bar there is that cap."""
    g, net, y0 = grid64
    nS = net.nSpecies
    out = _solve(racgpu, net, y0, g["cells"], 1e-4)
    bad = []
    for c in range(len(g["cells"])):
        ref, twin = g["yend"][c][:nS], g["yend_ulp"][c][:nS]
        floor = max(major_relerr(twin, ref), float(g["floor_twins"][c].max()))
        err = major_relerr(out["y"][c], ref)
        bound = max(1e-4, 3.0 * floor)
        if out["stats"][c, racgpu.S_NERR] > 0 or g["scalars"][c, 2] > 0:
            bound = max(bound, 3e-3)
        if err > bound:
            bad.append((int(g["grid_idx"][c]), err, floor))
    assert not bad, bad
    assert (out["t_final"] == g["scalars"][:, 0]).all()
    assert (out["quality"] == g["scalars"][:, 1].astype(int)).all()


def test_grid_cells_error_returns_are_the_references_kind(racgpu, grid64):
    g, net, y0 = grid64
    out = _solve(racgpu, net, y0, g["cells"], 1e-4)
    nerr = out["stats"][:, racgpu.S_NERR]
    codes = out["stats"][:, racgpu.S_ERRCODES]
    by_code = np.array([[(int(c) >> s) & 0xffff for s in (0, 16, 32, 48)] for c in codes])
    assert (by_code.sum(axis=1) == nerr).all()
    assert (by_code[:, 0] == 0).all() and (by_code[:, 3] == 0).all()  # only error-test (-4) and convergence (-5) failures, as in the reference
    assert (g["errcodes"][:, 0] == 0).all() and (g["errcodes"][:, 3] == 0).all()
    # the reference against its own 1-ulp twin differs by up to 2 error returns on a cell, 6 in total on either side
    twin_spread = int(np.max(np.abs(g["scalars"][:, 2] - g["scalars_ulp"][:, 2])))
    assert twin_spread == 2
    assert np.max(np.abs(nerr - g["scalars"][:, 2])) <= twin_spread
    assert abs(int(nerr.sum()) - int(g["scalars"][:, 2].sum())) <= 6


def test_grid_cells_tight_tolerance_pair(racgpu, grid64):
    """RTOL 1e-8 on both sides, the reference's wall-clock guards and the engine's modelled ones off: the pin that trajectory noise does
    not limit.  Measured: max 1.5e-6 (H+ at 2900 K), median 9e-9 over the 64 cells; the reference's own RTOL 1e-8 <-> 1e-10 distance is up
    to 2.5e-6.  (Before dev_rhs balanced the grain number exactly the hot cells were up to 6e-5 off in H+: DESIGN.md section 2.)"""
    g, net, y0 = grid64
    nS = net.nSpecies
    p = racgpu.default_params()
    p.RTOL = 1e-8; p.max_runtime_allowed = 0.0
    out = net.evol_solve_batch(p, g["cells"], net.init_abundances(y0, g["cells"]))
    bad = []
    for c in range(len(g["cells"])):
        ref, truth = g["yend_tight"][c][:nS], g["yend_tighter"][c][:nS]
        own = major_relerr(ref, truth)  # the reference's RTOL 1e-8 run against its RTOL 1e-10 run
        err = major_relerr(out["y"][c], ref)
        if err > 5e-6 + 3.0 * own:
            bad.append((int(g["grid_idx"][c]), float(g["cells"][c, 0]), err, own))
        # and against the better truth the engine's RTOL 1e-8 run is as good as the reference's own
        assert major_relerr(out["y"][c], truth) <= 5e-6 + 3.0 * own, (int(g["grid_idx"][c]), major_relerr(out["y"][c], truth), own)
    assert not bad, bad
    same_nerr = out["stats"][:, racgpu.S_NERR] == g["scalars_tight"][:, 2]
    assert (out["t_final"][same_nerr] == g["scalars_tight"][same_nerr, 0]).all()
    assert (out["quality"] == g["scalars_tight"][:, 1].astype(int)).all()
    assert np.max(np.abs(out["stats"][:, racgpu.S_NERR] - g["scalars_tight"][:, 2])) <= 1  # (a handful of the 64 cells differ by one on either side)
    # the engine needs no more steps than the reference's restatement does on these cells (oracle: 3545 ... 9435 on the ten hardest)
    assert out["stats"][:, racgpu.S_NST].max() < 9000


def test_grid_cells_rate_coefficients(racgpu, grid64):
    """chem_cal_rates at the grid's own temperatures (9 ... 3300 K): 1e-12 relative, same zero pattern (duplicate pruning, ranges)."""
    g, net, y0 = grid64
    sub = g["rates_cells"]
    k = net.cal_rates(racgpu.default_params(), g["cells"][sub])
    ref = g["rates"]
    assert ((k == 0) == (ref == 0)).all()
    nz = ref != 0
    assert np.max(np.abs(k[nz] - ref[nz]) / np.abs(ref[nz])) <= 1e-12


def test_grid_cells_rhs_at_the_end_states(racgpu, grid64):
    """chem_ode_f at the reference's end state of every cell: within 1e-9 of the largest flux touching each species."""
    g, net, y0 = grid64
    nS = net.nSpecies
    p = racgpu.default_params()
    rx = net.reactions()
    y = np.ascontiguousarray(g["yend"][:, :nS])
    yd = net.ode_f(p, g["cells"], y)
    k = net.cal_rates(p, g["cells"])
    for c in range(len(g["cells"])):
        ya = np.where(rx["reac"][:, 0] > 0, y[c][np.maximum(rx["reac"][:, 0] - 1, 0)], 0.0)
        yb = np.where((rx["reac"][:, 1] > 0) & np.isin(rx["itype"], (5, 6, 21, 64)), y[c][np.maximum(rx["reac"][:, 1] - 1, 0)], 1.0)
        fl = np.abs(k[c] * ya * yb)
        scale = np.zeros(nS)
        for cols in (rx["reac"], rx["prod"]):
            for s in range(cols.shape[1]):
                m = cols[:, s] > 0
                np.maximum.at(scale, cols[m, s] - 1, fl[m])
        err = np.abs(yd[c] - g["ydotend"][c][:nS])
        assert (err <= 1e-9 * np.maximum(scale, 1e-300) + 1e-300).all(), (int(g["grid_idx"][c]), float(err.max()))
