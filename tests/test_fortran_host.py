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
    rows = open(tmp_path / "out.dat").read().splitlines()
    assert len(rows) == 3 and len(rows[0]) == 14 * (nS + 3)
    hdr = [rows[0][14 * k:14 * (k + 1)].strip() for k in range(nS + 3)]
    assert hdr[3:] == list(g["species"])


def test_fortran_host_reads_reference_namelist_without_gpu(tmp_path, racgpu):
    """No GPU: the host must parse the namelist and then refuse loudly (exit code 1, message)."""
    if racgpu.device_count() > 0 or not os.path.exists(HOST):
        pytest.skip("GPU visible or host not built")
    np.savetxt(tmp_path / "cells.txt", racgpu.cells.make_cell(50.0, 40.0, 1e8, 5.0, 1e3)[None, :], fmt="%.17e")
    out = subprocess.run([HOST, os.path.join(ROOT, "tests", "fortran_host", "configure_chemistry.dat"),
                          str(tmp_path / "cells.txt"), str(tmp_path / "out")], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 1 and "no HIP device" in out.stdout + out.stderr
