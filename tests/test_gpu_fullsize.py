"""BASELINE configs[1] at its full size (10 000 synthetic cells, rate06 no-grain) on the GPU, checked through
properties that do not need a reference run of 10 000 cells:

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
