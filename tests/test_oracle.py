"""The CPU oracle against the golden vectors of the unmodified reference (tests/golden/*.npz).

This is what pins the oracle: network parse, species attributes, Jacobian pattern, initial abundances, rate
coefficients, tolerances, ydot at two compositions and the Jacobian values must be BIT-IDENTICAL to the
reference's; the integrated end state must agree to the reference's own rounding-noise floor (its answer
moves by `yend_ulp` when one input moves by one ulp), and fill of the sparse LU must match YSMP's to 3 %."""
import numpy as np
import pytest

from conftest import DATA, GOLDEN_TAGS, load_golden, major_relerr


@pytest.fixture(scope="module", params=GOLDEN_TAGS)
def case(request, oracle):
    g = load_golden(request.param)
    net = oracle.Network(f"{DATA}/{g['network_file']}")
    p = oracle.default_params()
    p.RTOL = float(g["rtol"]); p.t_max = float(g["t_max"])
    return request.param, g, net, p


def test_network_tables_bit_identical(case):
    tag, g, net, p = case
    assert net.names == list(g["species"])
    for k in ("reac", "prod", "n_reac", "n_prod", "itype"):
        np.testing.assert_array_equal(getattr(net, k), g[k])
    np.testing.assert_array_equal(np.diff(net.dupli_ptr), g["n_dupli"])
    np.testing.assert_array_equal(net.mass_num, g["mass_num"])
    np.testing.assert_array_equal(net.vib_freq, g["vib_freq"])
    np.testing.assert_array_equal(net.Edesorb, g["Edesorb"])
    np.testing.assert_array_equal(net.counterpart, g["counterpart"])
    np.testing.assert_array_equal(net.charge, g["charge"])
    np.testing.assert_array_equal(net.IA, g["IA"])
    np.testing.assert_array_equal(net.JA, g["JA"])
    y0 = net.initial_abundances(f"{DATA}/{g['initial_file']}")
    np.testing.assert_array_equal(y0, g["y0"])


def test_rates_tolerances_rhs_jacobian_bit_identical(case):
    tag, g, net, p = case
    y0 = g["y0"]
    for c, cell in enumerate(g["cells"]):
        k = net.rates(p, cell)
        np.testing.assert_array_equal(k, g["rates"][c])
        rt, at = net.tolerances(p, 1, cell[6])
        np.testing.assert_array_equal(rt, g["rtols"][c]); np.testing.assert_array_equal(at, g["atols"][c])
        y = net.initial_state(y0, cell)
        np.testing.assert_array_equal(net.rhs(p, cell, k, y), g["ydot0"][c])
        np.testing.assert_array_equal(net.rhs(p, cell, k, g["yend"][c]), g["ydotend"][c])
    cell = g["cells"][0]
    k = net.rates(p, cell); y = net.initial_state(y0, cell)
    J = net.jac_csc(p, cell, k, y)
    np.testing.assert_array_equal(J, g["jac0"])
    # the fast whole-matrix assembly equals the reference's column-at-a-time form
    for j in (1, 2, net.index("E-"), net.index("H2"), net.nS):
        col = net.jac_col(p, cell, k, y, j)
        np.testing.assert_array_equal(J[net.IA[j - 1] - 1:net.IA[j] - 1], col[net.JA[net.IA[j - 1] - 1:net.IA[j] - 1] - 1])


def test_end_state_within_reference_noise_floor(case):
    tag, g, net, p = case
    nS = net.nS
    for c, cell in enumerate(g["cells"]):
        if tag == "rate12_grain" and c > 0:
            continue  # keep the CPU suite short; the GPU suite covers every fixture cell
        s = net.solve_cell(p, cell, g["y0"])
        ref = g["yend"][c][:nS]
        floor = major_relerr(g["yend_ulp"][c][:nS], ref)
        err = major_relerr(s["y"][:nS], ref)
        print(f"{tag} cell {c}: oracle vs reference {err:.2e}; reference 1-ulp twin {floor:.2e}; NST {s['nst']} NFE {s['nfe']}")
        assert s["rc"] == 0 and s["t_final"] == g["scalars"][c, 0] and s["quality"] == int(g["scalars"][c, 1])
        assert abs(s["nerr"] - int(g["scalars"][c, 2])) <= 2  # discrete error returns are trajectory-noise sensitive
        assert err <= max(1e-4, 3.0 * floor)
        # sparse LU fill against YSMP's (IWORK(19,25,26) of the reference run)
        assert s["nnz"] == int(g["stats"][c, 4])
        assert abs(s["nzl"] + s["nzu"] - (g["stats"][c, 5] + g["stats"][c, 6])) <= 0.03 * (g["stats"][c, 5] + g["stats"][c, 6])


def test_step_counts_close_to_reference_without_resets(oracle):
    """With steps_reset_solver = infinity DLSODES' own counters cover the whole run: the restatement must take
    the same number of steps to within the spread the reference shows against its 1-ulp twin (a few %)."""
    g = load_golden("rate06_nograin")
    net = oracle.Network(f"{DATA}/{g['network_file']}")
    p = oracle.default_params(); p.steps_reset_solver = 9999999
    s = net.solve_cell(p, g["cells"][0], g["y0"])
    nst, nfe, nje, nlu = g["stats_noreset"][:4]
    assert abs(s["nst"] - nst) <= 0.05 * nst and abs(s["nfe"] - nfe) <= 0.06 * nfe
    assert abs(s["nje"] - nje) <= 3 and abs(s["nlu"] - nlu) <= 0.08 * nlu
    assert major_relerr(s["y"][:net.nS], g["yend_noreset"][:net.nS]) <= 1e-4


def test_n_record(oracle):
    p = oracle.default_params()
    assert oracle.lib().orc_n_record(__import__("ctypes").byref(p), 0.0, 1e6) == 316
