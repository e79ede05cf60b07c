"""Build-time guard (CPU only; hipcc cross-compiles gfx950 without a GPU): the register, scratch and LDS use of the hot kernel.

Round 1 lost a GPU run to a build of k_solve that spilled VGPRs to scratch ("Hazards"), and the kernel sat
at the 256-VGPR edge where one added variable halves the occupancy.  This test compiles rac-2d_amd/csrc/engine.hip with the
Makefile's own flags plus -Rpass-analysis=kernel-resource-usage and fails when k_solve uses scratch, accumulator registers
(hipcc's spill space once the 256 architectural VGPRs are exhausted), or more VGPRs than 3 waves per SIMD allow."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_k_solve_has_no_scratch_and_fits_three_waves_per_simd():
    csrc = os.path.join(ROOT, "rac-2d_amd", "csrc")
    out = subprocess.run(["make", "-s", "-C", csrc, "resources"], capture_output=True, text=True, timeout=600)
    text = out.stdout + out.stderr
    kernels = {}
    cur = None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); kernels[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur:
            kernels[cur][m.group(1).strip()] = m.group(2)
    ks = {k: v for k, v in kernels.items() if "k_solve" in k}
    # k_solve (one wave per cell); k_solve_team, k_solve_team_resume and k_solve_columns (four waves per cell); k_solve_T, k_solve_team_T (evolT)
    assert len(ks) == 6, (list(kernels), text[-2000:])
    for name, k in ks.items():
        # never: see the module docstring (k_solve_T: the 8-byte call frame of the one non-inlined function, dev_heating_cooling; no spills)
        assert int(k["ScratchSize"]) <= (8 if ("k_solve_T" in name or "k_solve_team_T" in name) else 0), (name, k)
        assert int(k["AGPRs"]) == 0, (name, k)                 # AGPR spills mean the 256 VGPRs ran out
        assert int(k["VGPRs Spill"]) == 0, (name, k)
        if "k_solve_T" in name or "k_solve_team_T" in name:  # the heating/cooling terms cost registers: two waves per SIMD (evolT)
            assert int(k["VGPRs"]) <= 256 and int(k["Occupancy"]) >= 2, (name, k)
            continue
        if "resume" in name or "columns" in name:  # run on an otherwise idle chip: two teams per CU
            assert int(k["VGPRs"]) <= 256, (name, k)
            continue
        if "team" in name:    # one wave of a team next to two single ones on a SIMD: 176 + 2 * 168 = 512 registers
            assert int(k["VGPRs"]) <= 176, (name, k)
            continue
        assert int(k["VGPRs"]) <= 168, (name, k)   # 3 waves per SIMD (MI355X_MICROARCH.md, register files)
        assert int(k["Occupancy"]) >= 3, (name, k)
    for name, v in kernels.items():                # no kernel of the library may use scratch beyond that call frame
        assert int(v["ScratchSize"]) <= (8 if ("k_solve_T" in name or "k_solve_team_T" in name or "evolT_hooks" in name) else 0), (name, v)
        assert int(v["VGPRs Spill"]) == 0, (name, v)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/amdflang"), reason="no amdflang")
def test_fortran_binding_module_compiles(tmp_path):
    """The ISO_C_BINDING module is part of the boundary: every interface block must compile (the GPU tests run a prebuilt host
    binary, which a broken module would leave stale)."""
    src = os.path.join(ROOT, "rac-2d_amd", "fortran", "racgpu_mod.f90")
    out = subprocess.run(["/opt/rocm/bin/amdflang", "-O0", "-c", src, "-o", str(tmp_path / "racgpu_mod.o"), "-J", str(tmp_path)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
