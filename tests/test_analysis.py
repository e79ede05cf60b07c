"""Analysis outputs against the reference's own: chem_ode_f_alt fluxes, production / destruction ranking,
elemental reservoirs, and the rows of the per-cell rate dump -- captured from the unmodified reference at the end state of a
configs[2] grid cell (tests/golden/policy_grain.npz, keys ana_*).  Host side only: no GPU needed (the rate coefficients and the
end state are the reference's; on the GPU box the same functions take Network.cal_rates and the solver's output)."""
import io

import numpy as np

import os
import pytest
from conftest import DATA, GOLDEN

G = np.load(f"{GOLDEN}/policy_grain.npz")


def _net(racgpu):
    return racgpu.Network(f"{DATA}/{G['network_file']}")


def test_reaction_fluxes_match_chem_ode_f_alt(racgpu):
    net = _net(racgpu)
    y = G["ana_yend"][:net.nSpecies]
    flux = racgpu.analysis.reaction_fluxes(net, G["ana_rates"], y, G["policy_cells"][0])
    ref = G["ana_flux"]
    assert ((flux == 0) == (ref == 0)).all()
    it = net.reactions()["itype"]
    surf = np.isin(it, (62, 75))  # k (1 - exp(-t)) with t down to 1e-9: one ulp of exp is 1e-7 of the result
    for m, tol in ((~surf, 1e-14), (surf, 1e-6)):
        nz = (ref != 0) & m
        assert np.max(np.abs(flux[nz] - ref[nz]) / np.abs(ref[nz])) <= tol


def test_contribution_ranking_matches_get_contribution_each(racgpu):
    net = _net(racgpu)
    y = G["ana_yend"][:net.nSpecies]
    species = [int(s) for s in G["ana_produ_species"]]
    con = racgpu.analysis.contributions(net, G["ana_rates"], y, G["policy_cells"][0], species)
    for sp in species:
        for which, lst in zip(("produ", "destr"), con[sp]):
            ref = G[f"ana_{which}_{sp}"].reshape(-1, 2)  # (reaction, contribution), the reference's top 20
            got = lst[:len(ref)]
            np.testing.assert_allclose([c for _, c in got], ref[:, 1], rtol=1e-6, atol=0)
            # same reactions wherever the contributions differ (ties may be ordered differently by the two sorts)
            for (i0, c), (r0, rc) in zip(got, ref):
                if i0 != int(r0):
                    assert abs(c - dict(lst)[int(r0)]) <= 1e-6 * abs(c), (sp, which, i0, int(r0))


def test_elemental_residence_matches_reference(racgpu):
    net = _net(racgpu)
    y = G["ana_yend"][:net.nSpecies]
    res = racgpu.analysis.elemental_residence(net, y)
    for e in range(20):
        ref = G[f"ana_eleres_{e + 1}"].reshape(-1, 3)
        got = res[e]
        assert len(got) == len(ref), (e, got, ref)
        for (i0, frac, accu), (r0, rf, ra) in zip(got, ref):
            if abs(rf) > 0:
                assert i0 == int(r0)
            np.testing.assert_allclose([frac, accu], [rf, ra], rtol=1e-13, atol=1e-300)


def test_rate_dump_rows_are_the_references(racgpu, tmp_path):
    net = _net(racgpu)
    racgpu.analysis.write_rate_dump(tmp_path / "reac_rates_cell_0001.dat", net, G["ana_rates"])
    got = open(tmp_path / "reac_rates_cell_0001.dat").read().splitlines()
    ref = str(G["ana_ratedump"]).splitlines()
    assert len(got) == len(ref) == net.nReactions
    bad = [(k, g, r) for k, (g, r) in enumerate(zip(got, ref)) if g != r]
    assert not bad, bad[:3]


def test_snapshot_blocks_are_written(racgpu):
    net = _net(racgpu)
    y = G["ana_yend"][:net.nSpecies]
    f = io.StringIO()
    racgpu.analysis.write_elements(f, net, 1e6, y, 73.3)
    racgpu.analysis.write_contributions(f, net, 1e6, y, G["ana_rates"], G["policy_cells"][0], [net.species_index("CO"), net.species_index("H2O")])
    txt = f.getvalue()
    assert "Total net charge" in txt and "Production" in txt and "Destruction" in txt and "CO  " in txt


def test_iter_dat_writer_reproduces_the_references_file(racgpu):
    """analysis.write_iter_dat against the reference's own writer (tests/golden/iter_probe_grain.dat: write_header +
    disk_save_results_write on a cell whose k-th printed field holds k + k/1000, abundance i = i * 1e-3), byte for byte; and the file
    reads back under the parsing rules of the reference's reader (first line without its first character split on blanks = keys,
    numpy.loadtxt(comments='!') columns = values)."""
    A = racgpu.analysis
    net = racgpu.Network(os.path.join(DATA, "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat"))
    ref = open(os.path.join(GOLDEN, "iter_probe_grain.dat")).read().splitlines()
    cols = {}
    for k, name in enumerate(A.ITER_INT_COLUMNS + A.ITER_REAL_COLUMNS, start=1):
        cols[name] = np.array([float(k) if k <= 6 else k + k * 1e-3])
    cols["cvg"] = np.array([1.0]); cols["cr_count"] = np.array([0.0])  # (converged = .true.; no optical record: cr_count prints 0)
    y = (np.arange(1, net.nSpecies + 1) * 1e-3)[None, :]
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "iter_0001.dat")
        A.write_iter_dat(fn, net.names, cols, y)
        mine = open(fn).read().splitlines()
        d = A.load_iter_dat(fn)
    assert mine[0].rstrip() == ref[0].rstrip()
    assert mine[1] == ref[1]
    assert len(d) == 148 + net.nSpecies and d["w_Kep"][0] == pytest.approx(142.142) and d["gH2O"][0] == pytest.approx((net.names.index("gH2O") + 1) * 1e-3)


def test_chem_analyse_driver_on_a_time_record(racgpu, tmp_path):
    """analysis.chem_analyse = the reference's chem_analyse loop (src/disk.f90:4136-4300) over a cell's record: which records it visits
    (every incr-th, skipping those that moved less than a tenth of the relative time step), the evol_ file (header, one ES14.4E4 row per
    record), one elemental block per visited record, one ranking block per visited record and species.  Host side only: a synthetic record
    around the reference's end state of the fixture cell, its own rate coefficients."""
    A = racgpu.analysis
    net = _net(racgpu)
    nS = net.nSpecies
    yend = G["ana_yend"]
    nrec = 60
    touts = 1e-3 * 1.3 ** np.arange(nrec)
    rec = np.tile(np.r_[yend[:nS], 50.0], (nrec, 1))
    rec[:40, :nS] *= (1.0 + 2.0 * np.exp(-np.arange(40) / 12.0))[:, None]       # moving at first, then flat: the flat part is skipped
    visited = A.analysed_records(touts, rec, nrec)
    want = []
    for k in range(1, nrec + 1, 1 + nrec // 20):                                # (the reference's loop, src/disk.f90:4196-4208)
        if k >= 2:
            dy = np.max(np.abs(rec[k - 1] - rec[k - 2]) / (rec[k - 1] + rec[k - 2] + 1e-15))
            if dy < 0.1 * (touts[k - 1] - touts[k - 2]) / (touts[k - 1] + touts[k - 2]):
                continue
        want.append(k)
    assert visited == want and visited[0] == 1 and 3 <= len(visited) < len(range(1, nrec + 1, 1 + nrec // 20))
    assert A.analysed_records(touts, rec, nrec, incr=1)[:3] == [1, 2, 3]
    cell = G["policy_cells"][0]
    sp = [net.species_index("CO"), net.species_index("H2O")]
    paths, vis = A.chem_analyse(net, str(tmp_path), 7, cell, touts, rec, nrec, lambda T: G["ana_rates"], species=sp, geometry=(1.5, 2.0, 0.25, 0.5))
    assert vis == visited and os.path.basename(paths[0]) == "evol_0007_rz_1.500000_.250000_iter_001.dat"
    ev = open(paths[0]).read().splitlines()
    assert len(ev) == nrec + 1 and ev[0].startswith("!Time_(yr)    ") and ev[0].endswith("  Tgas        ") and len(ev[0]) == 14 * (nS + 2)
    row = np.array([float(ev[5][14 * k:14 * (k + 1)]) for k in range(nS + 2)])
    np.testing.assert_allclose(row, np.r_[touts[4], rec[4]], rtol=1e-4)
    el = open(paths[1]).read()
    assert el.count("Time = ") == len(visited) and el.count("Total net charge") == len(visited)
    co = open(paths[2]).read()
    assert co.count("Time = ") == len(visited) and co.count("  Production") == 2 * len(visited) and co.count("Tgas = ") == len(visited)
    # a block is what write_contributions writes for that snapshot
    buf = io.StringIO(); A.write_contributions(buf, net, touts[0], rec[0, :nS], G["ana_rates"], cell, sp)
    assert co.split("\n", 1)[1].startswith(buf.getvalue())
