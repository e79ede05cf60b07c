#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference and `make -C oracle ref`).  The reference
itself cannot travel; what is committed is DATA: inputs (cell records) and the reference's outputs
(rates, ydot, Jacobian values, end-state abundances, step statistics, output times).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

For every network x cell the reference is run three times:
  * "cfg"   : the configuration's own settings (RTOL below, steps_reset_solver = 50)
  * "ulp"   : identical, but n_gas moved by ONE ulp -- the reference's own sensitivity to a
              rounding-level perturbation, i.e. the noise floor of any "matches DLSODES" claim
  * "tight" : RTOL = 1e-8 (truth), steps_reset_solver = 50
and cell 0 additionally with steps_reset_solver = 9999999 ("noreset") so that IWORK(11..13,21)
count the whole trajectory (every ISTATE=1 zeroes them).
"""
import math
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
INP = "/root/reference/inp/"
MP = 1.67262158e-24

sys.path.insert(0, ROOT)
import importlib  # noqa: E402
make_cell = importlib.import_module("rac-2d_amd.cells").make_cell  # same recipe as bench.py and the tests

CELLS = [  # Tgas, Tdust, n_gas, Av, G0_star   (T 10-2000 K, n 1e5-1e12, Av 0.01-50)
    (50.0, 40.0, 1e8, 5.0, 1e3),
    (300.0, 300.0, 1e10, 5.0, 1e3),
    (2000.0, 1500.0, 1e5, 0.01, 1e6),
    (10.0, 10.0, 1e12, 50.0, 1e3),
]
NETWORKS = [  # tag, network file, initial abundances, RTOL, t_max, cell subset
    ("rate06_nograin", "rate06_dipole_reformated_again_withoutgrain.dat", "ini_abund_waterice_loMetal.dat", 1e-4, 1e6, [0, 1, 2, 3]),
    ("rate06_grain", "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat", "ini_abund_waterice_loMetal.dat", 1e-4, 1e6, [0, 2]),
    ("rate06_default", "rate06_withgrain_lowH2Bind_hiOBind_lowCObind.dat", "ini_abund_waterice_loMetal_CO.dat", 1e-4, 1e6, [0, 3]),
    ("rate12_grain", "rate12_withGrain_lowH2Bind_hiObind.dat", "ini_abund_waterice_loMetal.dat", 1e-6, 1e7, [0, 1]),
]


def read_cell(fn):
    d, cur = {}, None
    for line in open(fn):
        if line.startswith("#"):
            p = line.split()
            cur = p[1] + ("_" + p[3] if len(p) > 3 else "")
            d[cur] = []
        else:
            d[cur].append(float(line))
    return {k: np.array(v) for k, v in d.items()}


def run_ref(network, initial, cells, rtol, t_max, steps_reset, dump_jac, solve=1):
    with tempfile.TemporaryDirectory() as td:
        np.savetxt(os.path.join(td, "cells.txt"), cells, fmt="%.17e")
        with open(os.path.join(td, "run.nml"), "w") as f:
            f.write("&ref_run\n chem_dir='%s'\n network='%s'\n initial='%s'\n out_dir='%s'\n cell_file='%s'\n"
                    " ncell=%d\n rtol=%.17e\n atol=1D-30\n dt_first_step=1D-8\n ratio_tstep=1.1D0\n t_max=%.17e\n"
                    " mxstep=6000\n steps_reset=%d\n dump_jac=%d\n solve=%d\n/\n"
                    % (INP, network, initial, td, os.path.join(td, "cells.txt"), len(cells), rtol, t_max,
                       steps_reset, dump_jac, solve))
        subprocess.run([DRIVER, os.path.join(td, "run.nml")], stdout=subprocess.DEVNULL, check=True)
        out = [read_cell(os.path.join(td, "cell_%04d.txt" % (i + 1))) for i in range(len(cells))]
        meta = dict(
            species=[l.rstrip("\n") for l in open(os.path.join(td, "species.txt"))],
            network=np.loadtxt(os.path.join(td, "network.txt"), skiprows=1, dtype=np.int32),
            attr=np.loadtxt(os.path.join(td, "species_attr.txt")),
            pattern=np.loadtxt(os.path.join(td, "pattern.txt"), dtype=np.int32),
            y0=np.loadtxt(os.path.join(td, "y0.txt")),
        )
    return out, meta


def main():
    for tag, network, initial, rtol, t_max, subset in NETWORKS:
        cells = np.array([make_cell(*CELLS[i]) for i in subset])
        cfg, meta = run_ref(network, initial, cells, rtol, t_max, 50, 1)
        cells_ulp = cells.copy()
        cells_ulp[:, 2] = np.nextafter(cells_ulp[:, 2], np.inf)
        cells_ulp[:, 5] = cells_ulp[:, 2] * cells_ulp[:, 6]
        ulp, _ = run_ref(network, initial, cells_ulp, rtol, t_max, 50, 0)
        tight, _ = run_ref(network, initial, cells, 1e-8, t_max, 50, 0)
        noreset, _ = run_ref(network, initial, cells[:1], rtol, t_max, 9999999, 0)
        nS = len(meta["species"])
        NEQ = nS + 1
        out = dict(
            network_file=network, initial_file=initial, rtol=rtol, t_max=t_max,
            species=np.array(meta["species"]),
            reac=meta["network"][:, 0:3], prod=meta["network"][:, 3:7],
            n_reac=meta["network"][:, 7], n_prod=meta["network"][:, 8], itype=meta["network"][:, 9],
            n_dupli=meta["network"][:, 10],
            mass_num=meta["attr"][:, 0], vib_freq=meta["attr"][:, 1], Edesorb=meta["attr"][:, 2],
            counterpart=meta["attr"][:, 3].astype(np.int32), charge=meta["attr"][:, 4].astype(np.int32),
            IA=meta["pattern"][:NEQ + 1], JA=meta["pattern"][NEQ + 1:], y0=meta["y0"],
            cells=cells, cells_ulp=cells_ulp,
            rates=np.array([c["rates"] for c in cfg]),
            rtols=np.array([c["rtol"] for c in cfg]), atols=np.array([c["atol"] for c in cfg]),
            ydot0=np.array([c["ydot0"] for c in cfg]),
            ydotend=np.array([c["ydotend"] for c in cfg]),
            jac0=cfg[0]["jac0"],  # CSC values on IA/JA for cell 0 (the others only differ in rates)
            yend=np.array([c["yend"] for c in cfg]),
            scalars=np.array([c["scalars"][:3] for c in cfg]),  # t_final, quality, NERR
            stats=np.array([c["stats"] for c in cfg]),  # last-segment NST NFE NJE NLU, NNZ NZL NZU, n_record, n_record_real
            touts=np.array([c["touts"] for c in cfg]),
            yend_ulp=np.array([c["yend"] for c in ulp]),
            yend_tight=np.array([c["yend"] for c in tight]),
            stats_tight=np.array([c["stats"] for c in tight]),
            yend_noreset=noreset[0]["yend"], stats_noreset=noreset[0]["stats"],
            ref_cpu_seconds=np.array([c["scalars"][3] for c in cfg]),
        )
        fn = os.path.join(HERE, tag + ".npz")
        np.savez_compressed(fn, **out)
        ye, yu = out["yend"][:, :nS], out["yend_ulp"][:, :nS]
        for i in range(len(cells)):
            m = ye[i] >= 1e-6
            print("%-16s cell %d  NST(last seg)=%d  ref-vs-ulp-twin max rel (X>=1e-6): %.2e   cpu %.2fs" % (
                tag, i, out["stats"][i, 0], np.max(np.abs(ye[i][m] - yu[i][m]) / ye[i][m]), out["ref_cpu_seconds"][i]))
        print("wrote", fn, os.path.getsize(fn) // 1024, "KiB")


if __name__ == "__main__":
    main()
