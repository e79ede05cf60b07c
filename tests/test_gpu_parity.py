"""GPU parity tests: the HIP path, called through the C ABI (ctypes -> libracgpu.so), against
(1) the golden vectors produced by the unmodified reference and (2) the CPU oracle on seeded inputs.

Tolerances (stated per check):
  * rates: 1e-12 relative -- same formulas, different libm (device pow/exp vs glibc).
  * ydot:  1e-9 of the largest |flux| touching the species -- ydot is a sum of cancelling fluxes whose order
           of accumulation differs (LDS atomics vs the reference's reaction order).
  * Jacobian: 1e-12 relative per entry (same accumulation order as the reference).
  * end-state abundances: BASELINE.json's bar, <= 1e-4 relative on species with X >= 1e-6, EXCEPT where the
    reference's own answer moves by more than that when one input is perturbed by one ulp (its noise floor,
    recorded in the fixture as yend_ulp); then the bound is 3x that floor.
"""
import numpy as np
import pytest

import os

from conftest import DATA, GOLDEN, GOLDEN_TAGS, load_golden, major_relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=GOLDEN_TAGS)
def case(request, racgpu):
    g = load_golden(request.param)
    net = racgpu.Network(f"{DATA}/{g['network_file']}")
    p = racgpu.default_params()
    p.RTOL = float(g["rtol"]); p.t_max = float(g["t_max"])
    return request.param, g, net, p


def _y0(g, net, cells):
    return net.init_abundances(g["y0"], cells)


def test_rates_match_reference(case):
    tag, g, net, p = case
    k = net.cal_rates(p, g["cells"])
    ref = g["rates"]
    assert ((k == 0) == (ref == 0)).all(), "a rate the reference zeroes (duplicate pruning / range) is not zero"
    nz = ref != 0
    assert np.max(np.abs(k[nz] - ref[nz]) / np.abs(ref[nz])) <= 1e-12


def test_rhs_matches_reference(case):
    tag, g, net, p = case
    nS = net.nSpecies
    y0 = _y0(g, net, g["cells"])
    yd = net.ode_f(p, g["cells"], y0)
    rx = net.reactions()
    for c in range(len(g["cells"])):
        for y, ref in ((y0[c], g["ydot0"][c]), (g["yend"][c][:nS], g["ydotend"][c])):
            ydc = net.ode_f(p, g["cells"][c:c + 1], y[None, :])[0] if y is not y0[c] else yd[c]
            # scale: the largest single flux magnitude feeding each species
            k = g["rates"][c]
            ya = np.where(rx["reac"][:, 0] > 0, y[np.maximum(rx["reac"][:, 0] - 1, 0)], 0.0)
            yb = np.where((rx["reac"][:, 1] > 0) & np.isin(rx["itype"], (5, 6, 21, 64)), y[np.maximum(rx["reac"][:, 1] - 1, 0)], 1.0)
            fl = np.abs(k * ya * yb)
            scale = np.zeros(nS)
            for cols in (rx["reac"], rx["prod"]):
                for s in range(cols.shape[1]):
                    m = cols[:, s] > 0
                    np.maximum.at(scale, cols[m, s] - 1, fl[m])
            err = np.abs(ydc - ref[:nS])
            assert (err <= 1e-9 * np.maximum(scale, 1e-300) + 1e-300).all(), (tag, c, float(err.max()))


def test_jacobian_matches_reference(case):
    tag, g, net, p = case
    nS = net.nSpecies
    colptr, rowidx = net.jac_pattern()
    y0 = _y0(g, net, g["cells"][:1])
    vals = net.ode_jac(p, g["cells"][:1], y0)[0]
    # reference CSC (NEQ x NEQ, 1-based) -> dict
    IA, JA, ref = g["IA"], g["JA"], g["jac0"]
    dense = {}
    for j in range(nS):
        for q in range(IA[j] - 1, IA[j + 1] - 1):
            if JA[q] <= nS:
                dense[(JA[q], j + 1)] = ref[q]
    seen = set()
    for j in range(nS):
        for q in range(colptr[j] - 1, colptr[j + 1] - 1):
            key = (int(rowidx[q]), j + 1)
            seen.add(key)
            r = dense.get(key, 0.0)
            assert abs(vals[q] - r) <= 1e-12 * abs(r), (tag, key, vals[q], r)
    # everything the reference has outside our reduced pattern must be an explicit zero
    for key, r in dense.items():
        if key not in seen:
            assert r == 0.0, (tag, key, r)


def test_newton_solve_residual(case):
    """P x = b through the device LDU, checked by multiplying back with the device's own Jacobian values."""
    tag, g, net, p = case
    nS = net.nSpecies
    colptr, rowidx = net.jac_pattern()
    y = g["yend"][:1, :nS].copy()
    J = net.ode_jac(p, g["cells"][:1], y)[0]
    rng = np.random.default_rng(7)
    b = rng.standard_normal((1, nS)) * np.abs(y) + 1e-20
    for gamma in (1e-3, 1e2):
        x = net.newton_solve(p, g["cells"][:1], y, gamma, b)[0]
        Px = x.copy()
        for j in range(nS):
            sl = slice(colptr[j] - 1, colptr[j + 1] - 1)
            np.subtract.at(Px, rowidx[sl] - 1, gamma * J[sl] * x[j])
        # backward error relative to |P||x| row sums
        absPx = np.abs(x).copy()
        for j in range(nS):
            sl = slice(colptr[j] - 1, colptr[j + 1] - 1)
            np.add.at(absPx, rowidx[sl] - 1, gamma * np.abs(J[sl]) * abs(x[j]))
        assert (np.abs(Px - b[0]) <= 1e-9 * absPx + 1e-300).all(), (tag, gamma, float(np.max(np.abs(Px - b[0]) / absPx)))


def test_end_state_matches_reference(case):
    tag, g, net, p = case
    nS = net.nSpecies
    out = net.evol_solve_batch(p, g["cells"], _y0(g, net, g["cells"]))
    for c in range(len(g["cells"])):
        ref = g["yend"][c][:nS]
        floor = major_relerr(g["yend_ulp"][c][:nS], ref)
        err = major_relerr(out["y"][c], ref)
        bound = max(1e-4, 3.0 * floor)
        print(f"{tag} cell {c}: GPU vs reference {err:.2e} (reference 1-ulp noise floor {floor:.2e}); "
              f"NST {out['stats'][c, 0]} NFE {out['stats'][c, 1]} NJE {out['stats'][c, 2]} NLU {out['stats'][c, 3]}")
        assert out["t_final"][c] == g["scalars"][c, 0]
        assert out["quality"][c] == int(g["scalars"][c, 1])
        # NERR counts discrete error returns of the integrator; like the step count it is trajectory-noise
        # sensitive (the reference itself takes a different number of steps against its 1-ulp twin)
        assert abs(out["stats"][c, 4] - int(g["scalars"][c, 2])) <= 2
        assert err <= bound, (tag, c, err, floor)
        # also against the RTOL = 1e-8 reference: must be as close to the truth as the reference itself is
        truth = g["yend_tight"][c][:nS]
        assert major_relerr(out["y"][c], truth) <= max(1e-4, 3.0 * max(major_relerr(ref, truth), floor))


def test_tight_tolerance_run_matches_the_reference_truth(case, racgpu):
    """RTOL = 1e-8 on both sides: the reference's own "truth" run (yend_tight) against the GPU at the same tolerance.
    At this tolerance the trajectory noise that limits the RTOL = 1e-4 comparison is gone, so this is the tightest
    pin of the whole path (rates, RHS, Jacobian, linear algebra, step control, output grid) against the reference:
    bound 1e-5 on species with X >= 1e-6, measured 1e-9 ... 1e-6."""
    tag, g, net, p0 = case
    nS = net.nSpecies
    p = racgpu.default_params()
    for f in ("t_max", "ATOL", "dt_first_step", "ratio_tstep", "mxstep_per_interval", "steps_reset_solver"):
        setattr(p, f, getattr(p0, f))
    p.RTOL = 1e-8
    out = net.evol_solve_batch(p, g["cells"], _y0(g, net, g["cells"]))
    for c in range(len(g["cells"])):
        truth = g["yend_tight"][c][:nS]
        err = major_relerr(out["y"][c], truth)
        print(f"{tag} cell {c}: GPU(1e-8) vs reference(1e-8) {err:.2e}; NST {out['stats'][c, 0]}")
        assert out["quality"][c] == 0 and out["t_final"][c] == p.t_max
        assert err <= 1e-5, (tag, c, err)


def test_without_solver_resets(case, racgpu):
    """steps_reset_solver = 9999999 (no ISTATE = 1 restarts; the reference's yend_noreset for cell 0): exercises the long
    uninterrupted history (order and step size carried across all 315 output intervals)."""
    tag, g, net, p0 = case
    nS = net.nSpecies
    p = racgpu.default_params()
    for f in ("t_max", "RTOL", "ATOL", "dt_first_step", "ratio_tstep", "mxstep_per_interval"):
        setattr(p, f, getattr(p0, f))
    p.steps_reset_solver = 9999999
    out = net.evol_solve_batch(p, g["cells"][:1], _y0(g, net, g["cells"][:1]))
    ref = g["yend_noreset"][:nS]
    floor = major_relerr(g["yend_ulp"][0][:nS], g["yend"][0][:nS])
    err = major_relerr(out["y"][0], ref)
    print(f"{tag}: GPU vs reference without resets {err:.2e} (1-ulp floor of the cell with resets {floor:.2e}); NST {out['stats'][0, 0]} "
          f"(reference {int(g['stats_noreset'][0])})")
    assert out["quality"][0] == 0 and out["t_final"][0] == p.t_max
    assert err <= max(1e-4, 3.0 * floor), (tag, err, floor)
    assert abs(out["stats"][0, 0] - g["stats_noreset"][0]) <= 0.1 * g["stats_noreset"][0]  # same number of steps to 10 %


def test_record_and_touts(case):
    tag, g, net, p = case
    if tag != "rate06_nograin":
        pytest.skip("record layout checked on one network")
    out = net.evol_solve_batch(p, g["cells"][:1], _y0(g, net, g["cells"][:1]), record=True)
    np.testing.assert_allclose(out["touts"][0], g["touts"][0], rtol=1e-13, atol=0)
    nS = net.nSpecies
    assert out["record"].shape == (1, len(g["touts"][0]), nS + 1)
    np.testing.assert_array_equal(out["record"][0, -1, :nS], out["y"][0])
    assert out["record"][0, 0, nS] == g["cells"][0][0]  # T slot carries Tgas


def test_gpu_vs_oracle_random_cells(racgpu, oracle):
    """Seeded synthetic cells (the BASELINE config-2 recipe), GPU vs the CPU oracle, short horizon."""
    net = racgpu.Network(f"{DATA}/rate06_dipole_reformated_again_withoutgrain.dat")
    onet = oracle.Network(f"{DATA}/rate06_dipole_reformated_again_withoutgrain.dat")
    y0 = net.load_initial_abundances(f"{DATA}/ini_abund_waterice_loMetal.dat")
    cells = racgpu.cells.synth_batch(6, seed=11)
    p = racgpu.default_params(); p.t_max = 1e3
    op = oracle.default_params(); op.t_max = 1e3
    out = net.evol_solve_batch(p, cells, net.init_abundances(y0, cells))
    for c in range(len(cells)):
        o = onet.solve_cell(op, cells[c], y0)
        err = major_relerr(out["y"][c], o["y"][:net.nSpecies])
        print(f"cell {c}: T={cells[c][0]:.1f} n={cells[c][2]:.2e} GPU vs oracle {err:.2e}; NST gpu {out['stats'][c, 0]} oracle {o['nst']}")
        assert out["quality"][c] == o["quality"] and out["t_final"][c] == o["t_final"]
        # no 1-ulp twin of the reference exists for these cells; the floor seen on the fixtures reaches 7e-5 already at
        # X >= 1e-6, so allow 3e-4 here (the fixture tests above carry the 1e-4 bar with measured floors)
        assert err <= 3e-4


def test_H2_form_use_moeq_rate_coefficients(racgpu):
    """chemsol_params%H2_form_use_moeq = .true. (reference src/chemistry.f90:876-881): the gH + gH coefficient by the rate equation's
    steady state.  Rate coefficients of four cells against the reference's (tests/golden/moeq_grain.npz: <= 1e-12, and only that
    coefficient differs from the default branch).  chem_ode_f / chem_ode_jac change form with the switch as well (src/disk.f90:4625-4630,
    4828-4840): that part is not built, and the integrator refuses the switch."""
    g = np.load(os.path.join(GOLDEN, "moeq_grain.npz"))
    net = racgpu.Network(os.path.join(DATA, str(g["network_file"])))
    p = racgpu.default_params()
    k0 = net.cal_rates(p, g["cells"])
    p.H2_form_use_moeq = 1
    k = net.cal_rates(p, g["cells"])
    ref = g["rates"]
    assert ((k == 0) == (ref == 0)).all()
    nz = ref != 0
    assert np.max(np.abs(k[nz] - ref[nz]) / np.abs(ref[nz])) <= 1e-12
    changed = np.nonzero((k != k0).any(axis=0))[0]
    assert len(changed) == 1 and (g["rates_default_cell0"] != ref[0]).sum() == 1 and g["rates_default_cell0"][changed[0]] != ref[0][changed[0]]
    y0 = net.load_initial_abundances(os.path.join(DATA, str(g["initial_file"])))
    with pytest.raises(racgpu.RacgpuError, match="not implemented"):
        net.evol_solve_batch(p, g["cells"][:1], net.init_abundances(y0, g["cells"][:1]))
