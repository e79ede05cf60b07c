"""The Fortran ISO_C_BINDING host end to end on the GPU: reads the reference-style &chemistry_configure namelist,
solves a small cell table through libracgpu.so and writes the reference-layout outputs."""
import importlib
import os
import subprocess

import numpy as np
import pytest

from conftest import DATA, ROOT, load_golden, major_relerr

HOST = os.path.join(ROOT, "rac-2d_amd", "fortran", "racgpu_host")


@pytest.mark.gpu
def test_fortran_host_matches_reference(tmp_path, racgpu):
    if not os.path.exists(HOST):
        pytest.skip("racgpu_host not built")
    g = load_golden("rate06_nograin")
    cells = g["cells"][:2]
    np.savetxt(tmp_path / "cells.txt", cells, fmt="%.17e")
    out = subprocess.run([HOST, os.path.join(ROOT, "tests", "fortran_host", "configure_chemistry.dat"),
                          str(tmp_path / "cells.txt"), str(tmp_path / "out")], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    nS = 464
    rec = np.fromfile(tmp_path / "out.bin", dtype=np.float64).reshape(2, nS + 20)
    assert (rec[:, nS:] == 0).all()  # the 20 column densities belong to the caller
    for c in range(2):
        ref = g["yend"][c][:nS]
        floor = major_relerr(g["yend_ulp"][c][:nS], ref)
        assert major_relerr(rec[c, :nS], ref) <= max(1e-4, 3 * floor)
    # <out>.dat is the reference's iter_NNNN.dat: header byte for byte what analysis.write_iter_dat (pinned to the reference's own writer,
    # tests/test_analysis.py) writes for this network, rows parsed by the reference reader's rules
    A = racgpu.analysis
    rows = open(tmp_path / "out.dat").read().splitlines()
    assert len(rows) == 3
    cols = {k: np.zeros(2) for k in A.ITER_INT_COLUMNS + A.ITER_REAL_COLUMNS}
    A.write_iter_dat(tmp_path / "py.dat", list(g["species"]), cols, np.zeros((2, nS)))
    assert rows[0].rstrip() == open(tmp_path / "py.dat").read().splitlines()[0].rstrip()
    d = A.load_iter_dat(tmp_path / "out.dat")
    assert len(d) == 148 + nS
    np.testing.assert_allclose(d["n_gas"], cells[:, racgpu.cells.P_NGAS], rtol=1e-5)
    np.testing.assert_allclose(d["Tgas"], cells[:, racgpu.cells.P_TGAS], rtol=1e-5)
    np.testing.assert_allclose(d["f_CO_S"], cells[:, racgpu.cells.P_FSS_STAR_CO], rtol=1e-5)
    np.testing.assert_allclose(np.array([d[nm] for nm in g["species"]]).T, rec[:, :nS], rtol=1.1e-5, atol=1e-290)
    assert (d["t_final"] > 0).all() and (d["qual"] == 0).all()
    # restart from the .bin just written (the reference's use_backup_chemical_data path), local-iteration loop on: cells
    # that finished with quality 0 stay as they are handed in (t0 = 0 again, so this is a second full run from the end state)
    out2 = subprocess.run([HOST, os.path.join(ROOT, "tests", "fortran_host", "configure_chemistry.dat"), str(tmp_path / "cells.txt"),
                           str(tmp_path / "out2"), "4", str(tmp_path / "out.bin")], cwd=ROOT, capture_output=True, text=True)
    assert out2.returncode == 0 and "Abundances taken from" in out2.stdout, out2.stdout + out2.stderr
    rec2 = np.fromfile(tmp_path / "out2.bin", dtype=np.float64).reshape(2, nS + 20)
    assert np.isfinite(rec2).all() and rec2.shape == rec.shape
    row2 = open(tmp_path / "out2.counters").read().splitlines()[1]
    assert int(row2.split()[1]) == 1  # one local iteration sufficed


@pytest.mark.gpu
def test_fortran_host_writes_the_time_series_of_chem_evol_solve(tmp_path, racgpu):
    """flag_chem_evol_save = .true.: the file chem_evol_solve writes while integrating (reference
    src/chemistry.f90:404-413, 476-478): '! Time', names, 'Tgas' in A14; rows (t, y(1:NEQ)) in ES14.4E4 for the records
    2..n_record_real.  Checked against the reference's own record of the same cell (golden touts and end state)."""
    if not os.path.exists(HOST):
        pytest.skip("racgpu_host not built")
    g = load_golden("rate06_nograin")
    np.savetxt(tmp_path / "cells.txt", g["cells"][:1], fmt="%.17e")
    conf = open(os.path.join(ROOT, "tests", "fortran_host", "configure_chemistry.dat")).read()
    conf = conf.replace("flag_chem_evol_save         = .false.", "flag_chem_evol_save         = .true. ")
    assert ".true." in conf
    (tmp_path / "conf.dat").write_text(conf)
    out = subprocess.run([HOST, str(tmp_path / "conf.dat"), str(tmp_path / "cells.txt"), str(tmp_path / "out")],
                         cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = open(tmp_path / "out_cell000001_chem_evol_tmp.dat").read().splitlines()
    nS = 464
    hdr = [rows[0][14 * k:14 * (k + 1)] for k in range(nS + 2)]
    assert hdr[0] == "! Time        " and hdr[-1] == "   Tgas       " and [h.strip() for h in hdr[1:-1]] == list(g["species"])
    touts = g["touts"][0]
    assert len(rows) - 1 == len(touts) - 1 == 315  # records 2..316 (the initial state is not written)
    assert all(len(r) == 14 * (nS + 2) for r in rows[1:])
    data = np.array([[float(r[14 * k:14 * (k + 1)]) for k in range(nS + 2)] for r in rows[1:]])
    np.testing.assert_allclose(data[:, 0], touts[1:], rtol=6e-5)   # ES14.4E4: five significant digits
    assert np.all(data[:, -1] == g["cells"][0, 0])                    # the T slot stays at Tgas (fixed-T branch)
    ref = g["yend"][0][:nS]
    m = ref >= 1e-6
    floor = major_relerr(g["yend_ulp"][0][:nS], ref)
    assert np.max(np.abs(data[-1, 1:-1][m] - ref[m]) / ref[m]) <= max(1e-4, 3 * floor) + 6e-5


def test_fortran_host_reads_reference_namelist_without_gpu(tmp_path, racgpu):
    """No GPU: the host must parse the namelist and then refuse loudly (exit code 1, message)."""
    if racgpu.device_count() > 0 or not os.path.exists(HOST):
        pytest.skip("GPU visible or host not built")
    np.savetxt(tmp_path / "cells.txt", racgpu.cells.make_cell(50.0, 40.0, 1e8, 5.0, 1e3)[None, :], fmt="%.17e")
    out = subprocess.run([HOST, os.path.join(ROOT, "tests", "fortran_host", "configure_chemistry.dat"),
                          str(tmp_path / "cells.txt"), str(tmp_path / "out")], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 1 and "no HIP device" in out.stdout + out.stderr


@pytest.mark.gpu
def test_fortran_host_through_the_multi_gpu_entry_point(tmp_path, racgpu):
    """ndev = 1 on the one-GPU box: the cells go through racgpu_multi_calc_cells (dealing, one host thread per device, the RCCL
    all-gather of the result rows over a one-rank communicator) and must come back exactly as the single-device call returns them."""
    if not os.path.exists(HOST):
        pytest.skip("racgpu_host not built")
    g = load_golden("rate06_nograin")
    np.savetxt(tmp_path / "cells.txt", g["cells"][:3], fmt="%.17e")
    conf = os.path.join(ROOT, "tests", "fortran_host", "configure_chemistry.dat")
    a = subprocess.run([HOST, conf, str(tmp_path / "cells.txt"), str(tmp_path / "one"), "2"], cwd=ROOT, capture_output=True, text=True)
    b = subprocess.run([HOST, conf, str(tmp_path / "cells.txt"), str(tmp_path / "multi"), "2", "-", "1"], cwd=ROOT, capture_output=True, text=True)
    assert a.returncode == 0 and b.returncode == 0, a.stdout + a.stderr + b.stdout + b.stderr
    assert "one RCCL all-gather" in b.stdout
    assert (tmp_path / "one.bin").read_bytes() == (tmp_path / "multi.bin").read_bytes()
    ra, rb = open(tmp_path / "one.dat").read().splitlines(), open(tmp_path / "multi.dat").read().splitlines()
    assert ra == rb


@pytest.mark.gpu
def test_fortran_host_with_the_gas_temperature_evolving(tmp_path, racgpu):
    """hc.txt given: chemsol_params%evolT for every cell with en_gain_tot > 0; checked against the reference's own evolT run of the same
    cells (tests/golden/evolT_grain.npz)."""
    if not os.path.exists(HOST):
        pytest.skip("racgpu_host not built")
    g = load_golden("evolT_grain")
    sel = [1, 5]
    np.savetxt(tmp_path / "cells.txt", g["cells"][sel], fmt="%.17e")
    np.savetxt(tmp_path / "hc.txt", g["hc"][sel], fmt="%.17e")
    conf = open(os.path.join(ROOT, "tests", "fortran_host", "configure_chemistry.dat")).read()
    conf = conf.replace("rate06_dipole_reformated_again_withoutgrain.dat", str(g["network_file"]))
    assert str(g["network_file"]) in conf
    conf += """&heating_cooling_configure
  heating_cooling_config%dir_transition_rates    = 'DATADIR/'
  heating_cooling_config%use_analytical_CII_OI   = .true.
  heating_cooling_config%IonCoolingWithLut       = .true.
  heating_cooling_config%filename_NII            = 'N+_LUT.bin'
  heating_cooling_config%filename_SiII           = 'Si+_LUT.bin'
  heating_cooling_config%filename_FeII           = 'Fe+_LUT.bin'
  heating_cooling_config%solve_method            = 2
  heating_cooling_config%use_mygasgraincooling      = .true.
  heating_cooling_config%use_chemicalheatingcooling = .true.
  heating_cooling_config%use_Xray_heating           = .true.
  heating_cooling_config%heating_Xray_en            = 0.0D0
  heating_cooling_config%heating_eff_chem           = 0.3D0
  heating_cooling_config%heating_eff_H2form         = 0.5D0
  heating_cooling_config%heating_eff_phd_H2         = 1D0
  heating_cooling_config%heating_eff_phd_H2O        = 0.5D0
  heating_cooling_config%heating_eff_phd_OH         = 0.5D0
  heating_cooling_config%cooling_gg_coeff           = 1D0
/
""".replace("DATADIR", DATA)
    (tmp_path / "conf.dat").write_text(conf)
    out = subprocess.run([HOST, str(tmp_path / "conf.dat"), str(tmp_path / "cells.txt"), str(tmp_path / "out"), "1", "-", "0", str(tmp_path / "hc.txt")],
                         cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    nS = len(g["species"])
    rec = np.fromfile(tmp_path / "out.bin", dtype=np.float64).reshape(2, nS + 20)
    rows = open(tmp_path / "out.dat").read().splitlines()
    for k, c in enumerate(sel):
        ref, twin = g["yend"][c], g["yend_ulp"][c]
        floor = max(major_relerr(twin[:nS], ref[:nS]), abs(twin[nS] - ref[nS]) / ref[nS])
        assert major_relerr(rec[k, :nS], ref[:nS]) <= max(1e-4, 3 * floor)
        T = float(racgpu.analysis.load_iter_dat(tmp_path / "out.dat")["Tgas"][k])
        assert abs(T - ref[nS]) <= max(1e-4, 3 * floor) * ref[nS] + 5e-6 * ref[nS]  # (ES14.5E3: six digits)
