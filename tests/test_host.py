"""Host-side (no GPU) checks of the product library: it loads, exports the whole C ABI, and its parser,
species bookkeeping, tolerance policy and symbolic factorisation agree with the reference fixtures.
No compute entry point is called here; without a GPU they must fail loudly, which is also checked."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest

from conftest import DATA, GOLDEN_TAGS, ROOT, load_golden


def test_abi_exports_every_declared_symbol(racgpu):
    lib = racgpu.lib()
    hdr = open(os.path.join(ROOT, "include", "racgpu.h")).read()
    declared = set(re.findall(r"\b(racgpu_[A-Za-z0-9_]+)\s*\(", hdr))
    assert declared == set(racgpu.ABI_SYMBOLS), declared ^ set(racgpu.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert ctypes.sizeof(racgpu.ChemsolParams) == 8 * 8 + 6 * 4 + 8 + 3 * 8


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_parser_matches_reference(racgpu, tag):
    g = load_golden(tag)
    net = racgpu.Network(f"{DATA}/{g['network_file']}")
    assert net.names == list(g["species"])
    rx = net.reactions()
    for k in ("reac", "prod", "n_reac", "n_prod", "itype", "n_dupli"):
        np.testing.assert_array_equal(rx[k], g[k])
    at = net.species_attrs()
    for k in ("mass_num", "vib_freq", "Edesorb", "counterpart", "charge"):
        np.testing.assert_array_equal(at[k], g[k])
    np.testing.assert_array_equal(net.load_initial_abundances(f"{DATA}/{g['initial_file']}"), g["y0"])
    p = racgpu.default_params(); p.RTOL = float(g["rtol"])
    for c, cell in enumerate(g["cells"]):
        rt, at_ = net.set_solver_flags_alt(p, 1, cell[6])
        np.testing.assert_array_equal(rt, g["rtols"][c]); np.testing.assert_array_equal(at_, g["atols"][c])
    y = net.init_abundances(g["y0"], g["cells"])
    i0 = net.species_index("Grain0")
    for c, cell in enumerate(g["cells"]):
        exp = g["y0"].copy()
        if i0:
            exp[i0 - 1] = cell[6]
        np.testing.assert_array_equal(y[c], exp)


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_pattern_is_reference_pattern_minus_structural_zeros(racgpu, tag):
    g = load_golden(tag)
    net = racgpu.Network(f"{DATA}/{g['network_file']}")
    nS = net.nSpecies
    colptr, rowidx = net.jac_pattern()
    ours = {(int(rowidx[q]), j + 1) for j in range(nS) for q in range(colptr[j] - 1, colptr[j + 1] - 1)}
    ref = {(int(g["JA"][q]), j + 1) for j in range(nS) for q in range(g["IA"][j] - 1, g["IA"][j + 1] - 1) if g["JA"][q] <= nS}
    assert ours - ref <= {(i, i) for i in range(1, nS + 1)}  # we only add missing diagonals (DPREP does too)
    assert all(r == c or True for r, c in ours)
    # everything of the reference we drop carries an exact zero in its Jacobian fixture (cell 0)
    pos = {(int(g["JA"][q]), j + 1): q for j in range(nS) for q in range(g["IA"][j] - 1, g["IA"][j + 1] - 1)}
    for key in ref - ours:
        assert g["jac0"][pos[key]] == 0.0
    # fill of our ordering vs YSMP's on the reference pattern (IWORK(25), IWORK(26)).  Ours has fewer rows/cols but
    # pads the trailing block (<= 128 wide, >= 90 % dense) to fully dense with explicit zeros: allow 8 %.
    assert net.nzl + net.nzu <= 1.08 * (g["stats"][0, 5] + g["stats"][0, 6])


def test_n_record_and_defaults(racgpu):
    p = racgpu.default_params()
    assert (p.RTOL, p.ATOL, p.t_max, p.dt_first_step, p.ratio_tstep) == (1e-4, 1e-30, 1e6, 1e-8, 1.1)
    assert (p.mxstep_per_interval, p.steps_reset_solver) == (6000, 50)
    assert racgpu.lib().racgpu_n_record(ctypes.byref(p), 0.0, 1e6) == 316


def test_errors_are_reported(racgpu):
    with pytest.raises(racgpu.RacgpuError):
        racgpu.Network("/nonexistent/network.dat")
    net = racgpu.Network(f"{DATA}/rate06_dipole_reformated_again_withoutgrain.dat")
    with pytest.raises(racgpu.RacgpuError):
        net.load_initial_abundances("/nonexistent/abund.dat")


def test_compute_fails_loudly_without_gpu(racgpu):
    if racgpu.device_count() > 0:
        pytest.skip("a GPU is visible")
    net = racgpu.Network(f"{DATA}/rate06_dipole_reformated_again_withoutgrain.dat")
    cell = racgpu.cells.make_cell(50.0, 40.0, 1e8, 5.0, 1e3)
    with pytest.raises(racgpu.RacgpuError, match="no CPU fallback|HIP"):
        net.cal_rates(racgpu.default_params(), cell)


def test_ragged_and_comment_rows(racgpu, tmp_path):
    """Edge cases of the row format: comment/blank lines, D exponents, blank numeric fields, PHOTON/CRP slots."""
    rows = [
        "! a comment",
        "",
        "H2          PHOTON                  H           H                                    4D-11        0.00      2.6    10 41000  3 C PH M",
        "H           CRP                     H+          E-                                   5.98e-18     0.00      0.0    10 41000  1 C CP M",
        "H                                   gH                                               1.0                                    61",
        "gH                                  H                                                1.0                  450.0             62      !HH93",
    ]
    f = tmp_path / "mini.dat"
    f.write_text("\n".join(rows) + "\n")
    net = racgpu.Network(str(f))
    assert net.names == ["H2", "H", "H+", "E-", "gH"]
    rx = net.reactions()
    np.testing.assert_array_equal(rx["n_reac"], [1, 1, 1, 1])
    np.testing.assert_array_equal(rx["n_prod"], [2, 2, 1, 1])
    np.testing.assert_array_equal(rx["itype"], [3, 1, 61, 62])
    at = net.species_attrs()
    assert at["Edesorb"][4] == 450.0 and at["counterpart"][4] == 2 and at["counterpart"][1] == 5
    assert at["charge"].tolist() == [0, 0, 1, -1, 0]


def test_per_cell_tmax_orbit_rule(racgpu):
    """reference src/disk.f90:2078-2085: min(t_max0, max(100, nOrbit_tmax * 2 pi / Omega / yr)) unless use_fixed_tmax."""
    yr = 3600.0 * 24.0 * 365.0
    omega_1au = 2 * np.pi / (365.25 * 86400.0)  # ~1.99e-7 rad/s
    t = racgpu.cells.tmax_this(np.array([omega_1au, omega_1au * 1e6, omega_1au * 1e-3]), t_max0=1e6, n_orbit_tmax=1e5)
    assert t[0] == pytest.approx(1e5 * 365.25 * 86400.0 / yr) and t[0] < 1e6   # 1e5 orbits at 1 AU ~ 1.0007e5 yr
    assert t[1] == 1e2                                                           # floor t_min = 100 yr
    assert t[2] == 1e6                                                           # capped by t_max0
    assert np.all(racgpu.cells.tmax_this(np.array([omega_1au]), t_max0=3e5, use_fixed_tmax=True) == 3e5)


def test_multi_gpu_dealing_rule_is_a_balanced_permutation(racgpu):
    """racgpu_multi_deal (host only): what racgpu_multi_calc_cells does before the devices start -- every cell gets one owner and one
    position, device batches differ by at most one cell, positions are 0..n_d-1 per device; with cost hints the cells go round-robin by
    descending cost (sweep.interleaved_order's rule), without them round-robin in cell order; merging by (owner, position) restores the
    caller's order."""
    sweep = importlib.import_module("rac-2d_amd.sweep")
    rng = np.random.default_rng(5)
    for ndev, ncell in ((1, 7), (2, 9), (3, 10), (8, 20000), (4, 3)):
        for cost in (None, rng.uniform(1.0, 100.0, ncell)):
            owner, pos = racgpu.multi_deal(ndev, ncell, cost)
            assert owner.min() >= 0 and owner.max() < ndev
            counts = np.bincount(owner, minlength=ndev)
            assert counts.max() - counts.min() <= 1
            for d in range(ndev):
                assert sorted(pos[owner == d]) == list(range(counts[d]))
            order = np.arange(ncell) if cost is None else np.argsort(-cost, kind="stable")
            np.testing.assert_array_equal(owner[order], np.arange(ncell) % ndev)       # k-th costliest cell -> device k mod ndev
            np.testing.assert_array_equal(pos[order], np.arange(ncell) // ndev)        # ... as its (k div ndev)-th cell
            if cost is not None:  # the same sets as the torch.distributed path deals (rank r: cells r, r + world, ... of the cost ranking)
                il = sweep.interleaved_order(cost, ndev)
                for d in range(ndev):
                    lo, hi = sweep.partition(ncell, ndev, d)
                    assert set(il[lo:hi]) == set(np.nonzero(owner == d)[0])
            # merge: per-device result rows back into cell order
            rows = [np.full((counts[d], 2), -1.0) for d in range(ndev)]
            for c in range(ncell):
                rows[owner[c]][pos[c]] = (c, 10.0 * c)
            merged = np.array([rows[owner[c]][pos[c]] for c in range(ncell)])
            np.testing.assert_array_equal(merged[:, 0], np.arange(ncell))
    with pytest.raises(racgpu.RacgpuError):
        racgpu.multi_deal(0, 5)
