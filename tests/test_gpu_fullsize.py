"""BASELINE configs[1] (10 000 synthetic cells, rate06 no-grain) and configs[2] (the 20 000-cell Andrews grid, rate06 with grains:
the headline workload) at their full sizes on the GPU, checked through properties that do not need a reference run of every cell:

* element conservation and charge neutrality, for the columns of the reference's element matrix that every
  reaction of the network balances.  A BDF step with a Newton corrector preserves such linear invariants up to the
  error of the linear solves, and those are LU factorisations without pivoting of matrices with condition numbers
  up to 1e25: the reference's own end states (tests/golden) drift by up to 2e-10 relative per element and 6e-8 in
  total mass, the GPU's by up to 5e-7 (abundances are per H nucleus).  The bound, 1e-6 of the sum of
  |count|*|abundance| plus 1e-7, is far below what a wrong stoichiometry or a lost flux would produce (>= the
  abundance of the species involved);
* every cell reaches its t_max with quality 0, or was stopped by the run-time guard past 0.5 t_max (still quality 0 in
  the reference's bookkeeping), or carries the reference's quality flags; the last two kinds are rare;
* the result of a cell does not depend on where in the batch, or in which order, it is solved: the first 64 cells
  solved alone, and the whole batch solved again costliest-first (racgpu_set_cost_hints), give the same bits.
"""
import numpy as np
import pytest

from conftest import DATA

pytestmark = pytest.mark.gpu
NET = "rate06_dipole_reformated_again_withoutgrain.dat"
NCELL = 10000


@pytest.fixture(scope="module")
def full(racgpu):
    net = racgpu.Network(f"{DATA}/{NET}")
    y0 = net.load_initial_abundances(f"{DATA}/ini_abund_waterice_loMetal.dat")
    cells = racgpu.cells.synth_batch(NCELL)  # the bench workload (seed 20240601)
    p = racgpu.default_params()
    yin = net.init_abundances(y0, cells)
    out = net.evol_solve_batch(p, cells, yin)
    return net, y0, cells, p, yin, out


def test_every_cell_finishes_or_is_flagged(full):
    net, y0, cells, p, yin, out = full
    q, tf, st = out["quality"], out["t_final"], out["stats"]
    good = q == 0
    assert good.sum() >= 0.995 * NCELL, "flagged cells: %d" % (~good).sum()
    # a cell stopped by the run-time guard after more than half of t_max keeps quality 0 (reference
    # src/chemistry.f90:480-491 exits the loop; flag 2 is only set for t <= 0.5 t_max, :581-582)
    early = good & (tf < p.t_max)
    assert np.all(tf[early] > 0.5 * p.t_max) and early.sum() <= 0.005 * NCELL, (early.sum(), tf[early].min() if early.any() else None)
    assert np.all(np.isin(q, [0, 1, 2, 3, 256, 257, 258, 259, 512, 513, 514, 515]))  # sums of the reference's flags
    assert np.all(np.isfinite(out["y"][good]))
    assert np.all(st[:, 0] > 0) and np.all(st[:, 1] >= st[:, 0]) and np.all(st[:, 3] >= st[:, 2])  # NST, NFE >= NST, NLU >= NJE


def test_elements_and_charge_are_conserved(full, oracle):
    net, y0, cells, p, yin, out = full
    onet = oracle.Network(f"{DATA}/{NET}")
    el = onet.elements.astype(np.float64)  # [nS, 20]: charge, then element counts (reference getElements)
    # which columns does the network itself balance?  (reactions as parsed; photons and cosmic rays are not species)
    reac, prod = onet.reac, onet.prod  # 1-based, 0 = empty
    balance = np.zeros((onet.nR, el.shape[1]))
    for k in range(3):
        m = reac[:, k] > 0
        balance[m] -= el[reac[m, k] - 1]
    for k in range(4):
        m = prod[:, k] > 0
        balance[m] += el[prod[m, k] - 1]
    balanced = [e for e in range(el.shape[1]) if np.any(el[:, e]) and not np.any(balance[:, e])]
    assert 0 in balanced and len(balanced) >= 8, balanced  # charge and the abundant elements at least
    good = (out["quality"] == 0) & (out["t_final"] == p.t_max)
    before = yin @ el
    after = out["y"] @ el
    tot = np.maximum(np.abs(yin) @ np.abs(el), np.abs(out["y"]) @ np.abs(el))  # scale: sum of |count| * |abundance|
    for e in balanced:  # column 0 is the charge (total 0: neutrality), the others are elements
        drift = np.abs(after[good, e] - before[good, e])
        bound = 1e-6 * tot[good, e] + 1e-7
        assert np.all(drift <= bound), (e, float(drift.max()), float((drift / bound).max()))


def test_a_cell_does_not_care_where_or_when_it_is_solved(full, racgpu):
    net, y0, cells, p, yin, out = full
    alone = net.evol_solve_batch(p, cells[:64], yin[:64])
    np.testing.assert_array_equal(alone["y"], out["y"][:64])
    net.set_cost_hints(out["stats"][:, 8].astype(np.float64))  # cycles of the first pass: costliest first
    try:
        again = net.evol_solve_batch(p, cells, yin)
    finally:
        net.set_cost_hints(None)
    np.testing.assert_array_equal(again["y"], out["y"])
    np.testing.assert_array_equal(again["stats"][:, :8], out["stats"][:, :8])
    np.testing.assert_array_equal(again["quality"], out["quality"])


# ---- the other three networks of BASELINE.json (charged grains, itype 21; 484-524 species), 256 synthetic cells each --------
OTHER = [("rate06_dipole_reformated_again_withgrain_lowH2Bind.dat", "ini_abund_waterice_loMetal.dat"),
         ("rate06_withgrain_lowH2Bind_hiOBind_lowCObind.dat", "ini_abund_waterice_loMetal_CO.dat"),
         ("rate12_withGrain_lowH2Bind_hiObind.dat", "ini_abund_waterice_loMetal.dat")]


@pytest.mark.parametrize("netfile,inifile", OTHER, ids=["rate06_grain", "rate06_default", "rate12_grain"])
def test_other_networks_batch_properties(racgpu, oracle, netfile, inifile):
    """Same properties on the with-grain networks: completion or flags, conservation of every balanced column of the
    element matrix (charge included: the grain charge states take part), independence of the batch position."""
    net = racgpu.Network(f"{DATA}/{netfile}")
    y0 = net.load_initial_abundances(f"{DATA}/{inifile}")
    n = 256
    cells = racgpu.cells.synth_batch(n, seed=77)
    p = racgpu.default_params()
    yin = net.init_abundances(y0, cells)
    out = net.evol_solve_batch(p, cells, yin)
    q, tf = out["quality"], out["t_final"]
    done = (q == 0) & (tf == p.t_max)
    assert done.sum() >= 0.97 * n, (int(done.sum()), np.unique(q, return_counts=True))
    assert np.all(tf[(q == 0) & ~done] > 0.5 * p.t_max)
    onet = oracle.Network(f"{DATA}/{netfile}")
    el = onet.elements.astype(np.float64)
    balance = np.zeros((onet.nR, el.shape[1]))
    for k in range(3):
        m = onet.reac[:, k] > 0
        balance[m] -= el[onet.reac[m, k] - 1]
    for k in range(4):
        m = onet.prod[:, k] > 0
        balance[m] += el[onet.prod[m, k] - 1]
    balanced = [e for e in range(el.shape[1]) if np.any(el[:, e]) and not np.any(balance[:, e])]
    assert len(balanced) >= 8, balanced
    before, after = yin @ el, out["y"] @ el
    tot = np.maximum(np.abs(yin) @ np.abs(el), np.abs(out["y"]) @ np.abs(el))
    for e in balanced:
        drift = np.abs(after[done, e] - before[done, e])
        bound = 1e-6 * tot[done, e] + 1e-7
        assert np.all(drift <= bound), (e, float(drift.max()), float((drift / bound).max()))
    alone = net.evol_solve_batch(p, cells[100:108], yin[100:108])
    np.testing.assert_array_equal(alone["y"], out["y"][100:108])


# ---- BASELINE configs[2], the headline workload: the full 20 000-cell Andrews grid on the rate06+grain network ---------------------
@pytest.fixture(scope="module")
def grid(racgpu):
    net = racgpu.Network(f"{DATA}/rate06_dipole_reformated_again_withgrain_lowH2Bind.dat")
    y0 = net.load_initial_abundances(f"{DATA}/ini_abund_waterice_loMetal.dat")
    cells = racgpu.cells.andrews_grid()
    p = racgpu.default_params()
    yin = net.init_abundances(y0, cells)
    out = net.evol_solve_batch(p, cells, yin)
    return net, cells, p, yin, out


def test_grid_every_cell_reaches_its_own_tmax(grid, racgpu):
    net, cells, p, yin, out = grid
    assert len(cells) == 20000
    tmax = cells[:, racgpu.cells.P_TMAX]
    assert tmax.min() < 1e4 and (tmax == p.t_max).sum() > 5000  # the orbit rule gives the inner columns shorter runs
    assert np.all(out["quality"] == 0), np.unique(out["quality"], return_counts=True)
    st = out["stats"]
    # a cell without error returns ends exactly at its t_max; one that took an ISTATE < 0 return lags by what that interval lost
    # and may run out of records a fraction of a per cent early (chem_evol_solve's loop, reference src/chemistry.f90:440-565: the
    # reference's own cells do the same, on other cells)
    clean = st[:, racgpu.S_NERR] == 0
    assert clean.sum() > 0.85 * len(cells)
    assert np.all(out["t_final"][clean] == tmax[clean])
    short = out["t_final"] != tmax
    assert short.sum() <= 30 and np.all(out["t_final"] >= 0.9 * tmax), (int(short.sum()), float((out["t_final"] / tmax).min()), st[short][:, [racgpu.S_NERR]].ravel())
    assert np.all(st[:, racgpu.S_ISAV] == st[:, racgpu.S_NREC_REAL]) and np.all(st[:, racgpu.S_NREC_REAL] == st[:, racgpu.S_NREC])
    assert np.all(np.isfinite(out["y"])) and np.all(np.isfinite(out["cell_out"]))
    # n_mol_on_grain is the sum of the surface species per grain (get_ice_coverage, reference src/chemistry.f90:989-1003)
    at_grain = np.array([nm.startswith("g") for nm in net.names])
    np.testing.assert_allclose(out["cell_out"][:, racgpu.O_N_MOL_ON_GRAIN], out["y"][:, at_grain].sum(axis=1) / cells[:, racgpu.cells.P_D2H], rtol=1e-12)
    # no cell is a tail: the costliest one stays within a few times the mean
    cyc = st[:, racgpu.S_CYC_TOTAL].astype(float)
    assert cyc.max() < 4.0 * cyc.mean()


def test_grid_elements_and_charge_are_conserved(grid, oracle):
    net, cells, p, yin, out = grid
    onet = oracle.Network(f"{DATA}/rate06_dipole_reformated_again_withgrain_lowH2Bind.dat")
    el = onet.elements.astype(np.float64)
    balance = np.zeros((onet.nR, el.shape[1]))
    for k in range(3):
        m = onet.reac[:, k] > 0
        balance[m] -= el[onet.reac[m, k] - 1]
    for k in range(4):
        m = onet.prod[:, k] > 0
        balance[m] += el[onet.prod[m, k] - 1]
    balanced = [e for e in range(el.shape[1]) if np.any(el[:, e]) and not np.any(balance[:, e])]
    assert 0 in balanced and len(balanced) >= 8, balanced
    before, after = yin @ el, out["y"] @ el
    tot = np.maximum(np.abs(yin) @ np.abs(el), np.abs(out["y"]) @ np.abs(el))
    for e in balanced:
        drift = np.abs(after[:, e] - before[:, e])
        bound = 1e-6 * tot[:, e] + 1e-7
        assert np.all(drift <= bound), (e, float(drift.max()), float((drift / bound).max()))


def test_grid_cells_do_not_care_where_or_when_they_are_solved(grid, racgpu):
    net, cells, p, yin, out = grid
    pick = np.arange(0, 20000, 313)  # 64 cells from all over the grid, solved as a batch of their own
    alone = net.evol_solve_batch(p, cells[pick], yin[pick])
    np.testing.assert_array_equal(alone["y"], out["y"][pick])
    np.testing.assert_array_equal(alone["t_final"], out["t_final"][pick])
    net.set_cost_hints(out["stats"][:, racgpu.S_CYC_TOTAL].astype(np.float64))
    try:
        again = net.evol_solve_batch(p, cells, yin)
    finally:
        net.set_cost_hints(None)
    np.testing.assert_array_equal(again["y"], out["y"])
    np.testing.assert_array_equal(again["stats"][:, :8], out["stats"][:, :8])
    # the caller's loop has nothing to redo on this workload: one local iteration everywhere, same bits
    loop = net.calc_cells(p, cells[pick], yin[pick], nlocal_iter=4)
    np.testing.assert_array_equal(loop["y"], out["y"][pick])
    assert np.all(loop["stats"][:, racgpu.S_NITER] == 1)
