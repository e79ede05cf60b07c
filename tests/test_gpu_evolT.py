"""Gas temperature co-evolving with the chemistry (chemsol_params%evolT, the reference's production default whenever a cell gains
energy): the HIP path through the C ABI against tests/golden/evolT_grain.npz -- eight cells of the configs[2] grid with heating/cooling
records (rac-2d_amd/cells.py::andrews_grid_hc), run through the UNMODIFIED reference with evolT = .true. (oracle/_ref/ref_driver,
tests/golden/make_golden.py evolT; the README template's heating_cooling_configure switches).

Tolerances:
  * the 28 heating/cooling terms: 1e-12 relative each (same formulas, device libm), same zero pattern; the net rate and dT/dt: 1e-12
    of the sum of the terms' magnitudes (the net is a difference of terms that cancel to 1e-6 in places);
  * dy/dt of the species at the reference's states: 1e-9 of the largest flux touching each species;
  * the finite-difference T row / T column of the Jacobian: a difference quotient of the above, so its error is the terms' error
    divided by the step: 1e-7 relative, floor 1e-9 |dT/dt| / step;
  * the run: T and the species with X >= 1e-6 at t_final within max(1e-4, 3 x the cell's 1-ulp floor) (BASELINE.json's bar; the
    floor is the reference against its own twin with n_gas moved by one ulp); t_final, quality and "T still evolving at the end" equal;
    at RTOL 1e-8 within 5e-5 (T within 1e-5).
"""
import numpy as np
import pytest

from conftest import DATA, load_golden, major_relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["evolT_grain", "evolT_default"])
def ev(racgpu, request):
    """evolT_grain: eight configs[2] grid cells on the rate06+grain network; evolT_default: four of them on the README-default network (484
    species, 5830 reactions, its own initial abundances)."""
    g = load_golden(request.param)
    net = racgpu.Network(f"{DATA}/{g['network_file']}")
    net.load_heating_cooling(DATA)
    return g, net


def _terms_close(got, ref, what):
    assert ((got[1:] == 0) == (ref[1:] == 0)).all(), (what, np.nonzero((got[1:] == 0) != (ref[1:] == 0))[0])
    nz = ref[1:] != 0
    rel = np.abs(got[1:][nz] - ref[1:][nz]) / np.abs(ref[1:][nz])
    assert rel.max() <= 1e-12, (what, int(np.argmax(rel)), float(rel.max()))
    scale = np.sum(np.abs(ref[1:]))
    assert abs(got[0] - ref[0]) <= 1e-12 * scale, (what, got[0], ref[0], scale)


def test_heating_cooling_terms_and_rhs_match_reference(racgpu, ev):
    g, net = ev
    nS = net.nSpecies
    p = racgpu.default_params()
    kB, spy = 1.3806503e-16, 3600.0 * 24.0 * 365.0
    rx = net.reactions()
    for tag_y, tag_f, tag_h in (("y0", "ydot0", "hc0"), ("yend", "ydotend", "hcend")):
        if tag_y == "y0":
            y = np.hstack([net.init_abundances(g["y0"], g["cells"]), g["cells"][:, :1]])
        else:
            y = g["yend"]
        out = net.ode_f_evolT(p, g["cells"], g["hc"], y)
        for c in range(len(g["cells"])):
            _terms_close(out["terms"][c], g[tag_h][c], (tag_h, int(g["grid_idx"][c])))
            scale = np.sum(np.abs(g[tag_h][c][1:])) * spy / (g["cells"][c, 2] * kB)
            assert abs(out["ydot"][c, nS] - g[tag_f][c][nS]) <= 1e-12 * scale, (tag_f, c, out["ydot"][c, nS], g[tag_f][c][nS])
            # species part: within 1e-9 of the largest flux touching each species (the bound of test_gpu_parity)
            k = net.cal_rates(p, np.r_[g["cells"][c][:0], [y[c, nS]], g["cells"][c][1:]][None, :])[0]
            ya = np.where(rx["reac"][:, 0] > 0, y[c][np.maximum(rx["reac"][:, 0] - 1, 0)], 0.0)
            yb = np.where((rx["reac"][:, 1] > 0) & np.isin(rx["itype"], (5, 6, 21, 64)), y[c][np.maximum(rx["reac"][:, 1] - 1, 0)], 1.0)
            fl = np.abs(k * ya * yb)
            scale = np.zeros(nS)
            for cols in (rx["reac"], rx["prod"]):
                for s_ in range(cols.shape[1]):
                    m = cols[:, s_] > 0
                    np.maximum.at(scale, cols[m, s_] - 1, fl[m])
            err = np.abs(out["ydot"][c, :nS] - g[tag_f][c][:nS])
            assert (err <= 1e-9 * np.maximum(scale, 1e-300) + 1e-300).all(), (tag_f, int(g["grid_idx"][c]), float(err.max()))


def test_jacobian_T_row_and_column_match_reference(racgpu, ev):
    g, net = ev
    nS = net.nSpecies
    p = racgpu.default_params()
    y = np.hstack([net.init_abundances(g["y0"], g["cells"]), g["cells"][:, :1]])
    out = net.ode_f_evolT(p, g["cells"], g["hc"], y, jac_border=True)
    for c in range(len(g["cells"])):
        Tdot = abs(g["ydot0"][c][nS])
        d2h = g["cells"][c, 6]
        for k in range(10):
            j = int(g["idx10"][k]) - 1
            dy = y[c, j] * 1e-2 + d2h * 1e-6
            ref = g["trow0"][c][k]
            assert abs(out["trow"][c, k] - ref) <= 1e-7 * abs(ref) + 1e-9 * Tdot / dy, (int(g["grid_idx"][c]), k, out["trow"][c, k], ref)
        dT = y[c, nS] * 1e-2 + 1.0
        ref = g["tcol0"][c]
        tol = 1e-7 * np.abs(ref) + 1e-9 * np.r_[np.full(nS, np.max(np.abs(g["ydot0"][c][:nS]))), Tdot] / dT
        bad = np.nonzero(np.abs(out["tcol"][c] - ref) > tol)[0]
        assert bad.size == 0, (int(g["grid_idx"][c]), bad[:5], out["tcol"][c][bad[:5]], ref[bad[:5]])


def test_T_row_by_blocks_is_the_T_row_by_full_evaluations(racgpu, ev):
    """The production path re-evaluates, for each of the ten T-row species, only the blocks of the heating/cooling function that read its
    abundance (kHcRowMask, engine_hc.hpp).  Against ten FULL evaluations (developer switch): the same bits, at the start state, at the end
    state and at randomly disturbed states of the eight fixture cells -- a block missing from a mask would change its row entry."""
    g, net = ev
    nS = net.nSpecies
    p = racgpu.default_params()
    rng = np.random.default_rng(7)
    y0 = np.hstack([net.init_abundances(g["y0"], g["cells"]), g["cells"][:, :1]])
    yend = g["yend"][:, :nS + 1].copy()
    states = [y0, yend]
    for _ in range(3):
        yy = yend.copy()
        yy[:, :nS] *= 10.0 ** rng.uniform(-1.0, 1.0, (len(yy), nS))     # every abundance moved by up to a factor of ten
        yy[:, nS] *= rng.uniform(0.5, 2.0, len(yy))
        states.append(yy)
    for y in states:
        a = net.ode_f_evolT(p, g["cells"], g["hc"], y, jac_border=True)
        b = net.ode_f_evolT(p, g["cells"], g["hc"], y, jac_border=True, full_rows=True)
        np.testing.assert_array_equal(a["trow"], b["trow"])
        np.testing.assert_array_equal(a["tcol"], b["tcol"])
        assert np.isfinite(a["trow"]).all() and (a["trow"] != 0).any()


def test_heat_reactions_are_the_references(racgpu, ev):
    g, net = ev
    rx, ht = net.heat_reactions()
    assert np.array_equal(rx, g["heat_rxn"])
    assert np.array_equal(ht, g["heat_val"])


def _run(racgpu, net, g, rtol, record=False):
    p = racgpu.default_params()
    p.RTOL = rtol
    return net.evolT_solve_batch(p, g["cells"], g["hc"], net.init_abundances(g["y0"], g["cells"]), record=record)


def test_T_and_abundances_at_the_end_of_the_run(racgpu, ev):
    g, net = ev
    nS = net.nSpecies
    out = _run(racgpu, net, g, 1e-4)
    bad = []
    for c in range(len(g["cells"])):
        ref, twin = g["yend"][c], g["yend_ulp"][c]
        floor = max(major_relerr(twin[:nS], ref[:nS]), abs(twin[nS] - ref[nS]) / ref[nS], float(g["floor_twins"][c].max()))
        err = max(major_relerr(out["y"][c], ref[:nS]), abs(out["cell_out"][c, racgpu.O_TGAS] - ref[nS]) / ref[nS])
        if err > max(1e-4, 3.0 * floor):
            bad.append((int(g["grid_idx"][c]), err, floor, out["cell_out"][c, racgpu.O_TGAS], ref[nS]))
    assert not bad, bad
    assert (out["t_final"] == g["scalars"][:, 0]).all()
    assert (out["quality"] == g["scalars"][:, 1].astype(int)).all()
    assert (out["cell_out"][:, racgpu.O_EVOLT_END] == g["evolTend"]).all()


def test_T_history_and_tight_tolerance_run(racgpu, ev):
    g, net = ev
    nS = net.nSpecies
    out = _run(racgpu, net, g, 1e-8, record=True)
    for c in range(len(g["cells"])):
        ref = g["yend_tight"][c]
        m = ref[:nS] >= 1e-6
        e = np.zeros(nS); e[m] = np.abs(out["y"][c][m] - ref[:nS][m]) / ref[:nS][m]
        # (cell 20, which heats from 1025 K to 1867 K, sits at 2e-5 in its worst species; the others far below)
        assert e.max() <= 5e-5, (int(g["grid_idx"][c]), float(e.max()), net.names[int(e.argmax())], float(ref[int(e.argmax())]))
        assert abs(out["cell_out"][c, racgpu.O_TGAS] - ref[nS]) <= 1e-5 * ref[nS]
        nrec = int(out["stats"][c, racgpu.S_NREC])
        Tref = g["Trecord_tight"][c][:nrec]
        Tgpu = out["record"][c, :nrec, nS]
        assert np.max(np.abs(Tgpu - Tref) / Tref) <= 1e-4, (int(g["grid_idx"][c]), float(np.max(np.abs(Tgpu - Tref) / Tref)))
    assert (out["cell_out"][:, racgpu.O_EVOLT_END] == g["evolTend_tight"]).all()


def test_cells_without_energy_gain_keep_their_temperature(racgpu, ev):
    """en_gain_tot <= 0 switches T evolution off for the cell (src/disk.f90:2071): the run is the fixed-T run, bit for bit."""
    g, net = ev
    hc = g["hc"].copy()
    hc[:, 0] = 0.0
    p = racgpu.default_params()
    y0 = net.init_abundances(g["y0"], g["cells"])
    a = net.evolT_solve_batch(p, g["cells"][:3], hc[:3], y0[:3])
    b = net.evol_solve_batch(p, g["cells"][:3], y0[:3])
    assert np.array_equal(a["y"], b["y"]) and np.array_equal(a["t_final"], b["t_final"])
    assert (a["cell_out"][:, racgpu.O_TGAS] == g["cells"][:3, 0]).all()


def test_four_waves_on_an_evolT_cell_give_the_bits_of_one(racgpu, ev):
    """With cost hints the costliest cells of an evolT batch start on a team of four waves (k_solve_team_T: the factorisation and the
    Jacobian shared out, f(y), the 28 terms and the border on wave 0).  Abundances, temperatures, times, counters: the bits of the
    one-wave run."""
    g, net = ev
    if len(g["cells"]) < 8:
        pytest.skip("the eight-cell fixture only")
    p = racgpu.default_params()
    y0 = net.init_abundances(g["y0"], g["cells"])
    net.set_cost_hints(None)
    a = net.evolT_solve_batch(p, g["cells"], g["hc"], y0)
    assert net.last_team_cells() == 0
    cost = np.ones(len(g["cells"])); cost[[1, 4, 6]] = 1e6          # three cells far above the team threshold
    net.set_cost_hints(cost)
    b = net.evolT_solve_batch(p, g["cells"], g["hc"], y0)
    net.set_cost_hints(None)
    assert net.last_team_cells() == 3
    for k in ("y", "t_final", "quality", "cell_out"):
        np.testing.assert_array_equal(a[k], b[k])
    S = racgpu
    for col in (S.S_NST, S.S_NFE, S.S_NJE, S.S_NLU, S.S_NERR, S.S_QSUM, S.S_ERRCODES):
        np.testing.assert_array_equal(a["stats"][:, col], b["stats"][:, col])


def test_local_iterations_with_T_evolving(racgpu, ev):
    """Network.evolT_calc_cells: the caller's loop of calc_this_cell over racgpu_evolT_solve_batch.  With mxstep so small that every
    interval returns ISTATE = -1, a cell ends flagged early and is continued from its hand-off record (abundances AND temperature) with the
    next tolerance policy: each iteration gets further, the loop is the by-hand sequence of evolT_solve_batch calls bit for bit, and cells
    that finish in the first iteration are what one evolT_solve_batch call returns."""
    g, net = ev
    if len(g["cells"]) < 8:
        pytest.skip("the eight-cell fixture only")
    p = racgpu.default_params()
    cells = g["cells"][:4]; hc = g["hc"][:4]
    y0 = net.init_abundances(g["y0"], cells)
    full = net.evolT_calc_cells(p, cells, hc, y0, nlocal_iter=3)
    one = net.evolT_solve_batch(p, cells, hc, y0, tol_j=1)
    done = one["quality"] == 0
    assert done.any()
    np.testing.assert_array_equal(full["y"][done], one["y"][done])
    np.testing.assert_array_equal(full["tgas"][done], one["cell_out"][done, racgpu.O_TGAS])
    assert (full["niter"][done] == 1).all()
    p.mxstep_per_interval = 6                                      # every interval ends in ISTATE = -1: flagged, stops early
    loop = net.evolT_calc_cells(p, cells[:2], hc[:2], y0[:2], nlocal_iter=3)
    assert (loop["niter"] >= 2).all()
    # by hand for cell 0
    C = racgpu.cells
    c0 = cells[:1].copy(); h0 = hc[:1].copy(); yy = y0[:1].copy(); t = np.zeros(1)
    for j in range(1, int(loop["niter"][0]) + 1):
        o = net.evolT_solve_batch(p, c0, h0, yy, t0=None if j == 1 else t, tol_j=j, rectify=j > 1)
        assert j == 1 or o["cell_out"][0, racgpu.O_T_END] > t[0]
        yy = o["y"]; t = o["t_final"].copy(); c0[0, C.P_TGAS] = o["cell_out"][0, racgpu.O_TGAS]
    np.testing.assert_array_equal(loop["y"][0], yy[0])
    assert loop["t_final"][0] == t[0] and loop["tgas"][0] == c0[0, C.P_TGAS] and t[0] > 0.0
