#!/usr/bin/env python3
"""Extract the DATA TABLES the reference compiles into itself and write them as plain-text data files under data/.

    python tools/extract_reference_tables.py           (build container only: reads /root/reference/src as text)

What is taken are numbers: the published tables of Neufeld & Kaufman 1993 / Neufeld, Lepp & Melnick 1995 (cooling functions of H2,
H2O, CO; reference src/load_Neufeld_cooling_{H2,H2O,CO}.f90) and of Visser, van Dishoeck & Black 2009 (12CO self-shielding at
Tex = 50 K, b = 0.3 km/s; src/load_Visser_CO_selfshielding.f90).  They are private module data in the reference, so no driver can
read them out of the compiled modules; the array constructors are parsed from the source text instead, each literal converted
the way the Fortran compiler converts it (a literal without a D exponent is single precision and is widened to double: the
Visser axes such as 15.20002927 and all of its table values are of that kind).  No code is taken.

The Visser table is then CHECKED through the compiled reference (oracle/_ref/ref_shielding, which calls the reference's own
get_12CO_shielding): at every node the function must return the extracted value, and at 2000 random points the bilinear
interpolation of ln f on the extracted table (the product's arithmetic, rac-2d_amd/cells.py::co_shielding) must agree to 1e-12.
The Neufeld tables are checked term by term through the heating/cooling fixtures (tests/golden/make_golden.py evolT).

Output format (data/neufeld_cooling_tables.dat, data/visser2009_co_shielding.dat): for every array a header line
"# name n1 [n2]" followed by its values in Fortran (column-major) order, one per line, 17 significant digits.
"""
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src"


def literal(tok):
    """value of a Fortran real literal as the compiler stores it in a double precision entity"""
    t = tok.strip().upper()
    if "D" in t:
        return float(t.replace("D", "E"))
    return float(np.float32(float(t)))  # default-kind real literal: single precision, then widened


def strip_comments(text):
    out = []
    for line in text.splitlines():
        s = line.strip()
        if s.startswith("!"):
            continue
        # a trailing comment (none of these files has '!' inside a string)
        out.append(line.split("!")[0])
    return "\n".join(out)


def arrays_of(path, int_params):
    """{name: ndarray} of every `name = (/ ... /)` / `name = reshape((/ ... /), (/n1, n2/))` in the (comment-free) text"""
    text = strip_comments(open(path).read()).replace("&", " ")
    text = re.sub(r"\s+", " ", text)
    res = {}
    for m in re.finditer(r"(\w+)\s*=\s*(reshape\s*\(\s*)?\(/(.*?)/\)(\s*,\s*\(/(.*?)/\)\s*\))?", text):
        name, is_reshape, body, _, shape = m.groups()
        toks = [t for t in body.split(",") if t.strip()]
        try:
            vals = np.array([literal(t) for t in toks])
        except ValueError:
            continue
        if is_reshape:
            dims = [int_params[d.strip()] if d.strip() in int_params else int(d) for d in shape.split(",")]
            # RESHAPE takes the leading elements when the source is longer than the shape (the CO high-temperature tables list eleven
            # column-density rows for an axis of ten: the compiled reference uses the first ten, and so does this file)
            assert int(np.prod(dims)) <= vals.size, (name, dims, vals.size)
            vals = vals[:int(np.prod(dims))].reshape(dims, order="F")
        res[name] = vals
    return res


def int_parameters(path):
    text = strip_comments(open(path).read()).replace("&", " ")
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"(\w+)\s*=\s*(\d+)\s*[,\n]", text)}


def write_tables(fn, header, groups):
    with open(fn, "w") as f:
        for h in header:
            f.write("! " + h + "\n")
        for prefix, arrs in groups:
            for name, a in arrs.items():
                a = np.asarray(a, dtype=np.float64)
                f.write("# %s%s %s\n" % (prefix, name, " ".join(str(d) for d in a.shape)))
                for v in a.ravel(order="F"):
                    f.write("%.17e\n" % v)


def main():
    # ---- Neufeld cooling tables -------------------------------------------------------------------------------------------
    groups = []
    for mol in ("H2", "H2O", "CO"):
        p = os.path.join(SRC, "load_Neufeld_cooling_%s.f90" % mol)
        arrs = arrays_of(p, int_parameters(p))
        groups.append((mol + ".", arrs))
        print(mol, {k: v.shape for k, v in arrs.items()})
    write_tables(os.path.join(ROOT, "data", "neufeld_cooling_tables.dat"),
                 ["Cooling-function tables of Neufeld & Kaufman 1993 (ApJ 418, 263) and Neufeld, Lepp & Melnick 1995 (ApJS 100, 132) as the",
                  "reference (rac-2d) tabulates them: axes T [K] or log10 T, log10 N~ [cm-2 km-1 s]; values -log10 L0, -log10 L_LTE, -log10 n_1/2, alpha.",
                  "Written by tools/extract_reference_tables.py; arrays in column-major order."], groups)
    # ---- Visser 2009 CO self-shielding ----------------------------------------------------------------------------------------
    p = os.path.join(SRC, "load_Visser_CO_selfshielding.f90")
    arrs = arrays_of(p, int_parameters(p))
    lh, lc, f = arrs["logN_H2"], arrs["logN_12CO"], arrs["f_12CO"]
    assert f.shape == (lc.size, lh.size), f.shape
    write_tables(os.path.join(ROOT, "data", "visser2009_co_shielding.dat"),
                 ["12CO self-shielding factors of Visser, van Dishoeck & Black 2009 (A&A 503, 323), Tex(CO) = 50 K, as the reference (rac-2d)",
                  "tabulates them: f_12CO(logN_12CO, logN_H2), axes log10 of the column densities [cm-2] (first node 0 = no column).",
                  "Written by tools/extract_reference_tables.py; arrays in column-major order."],
                 [("", {"logN_H2": lh, "logN_12CO": lc, "f_12CO": f})])
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_shielding")
    if os.path.exists(exe):
        def call(nh2, nco):
            inp = "".join("CO %.17e %.17e\n" % (a, b) for a, b in zip(nh2, nco))
            out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split()
            return np.array([float(v) for v in out])
        gh, gc = np.meshgrid(lh, lc)  # [ncol, nrow]
        fn = call(10.0 ** gh.ravel(), 10.0 ** gc.ravel()).reshape(gh.shape)
        e_nodes = float(np.max(np.abs(fn - f) / f))
        sys.path.insert(0, ROOT)
        import importlib
        cells = importlib.import_module("rac-2d_amd.cells")
        rng = np.random.default_rng(20240608)
        ph = 10.0 ** rng.uniform(13.0, 23.5, 2000); pc = 10.0 ** rng.uniform(8.0, 19.5, 2000)
        ref = call(ph, pc)
        mine = cells.co_shielding((lh, lc, f), ph, pc)
        e_pts = float(np.max(np.abs(mine - ref) / ref))
        print("Visser table through the compiled reference: nodes max rel %.2e, 2000 random points max rel %.2e" % (e_nodes, e_pts))
        assert e_nodes <= 1e-12 and e_pts <= 1e-12
    # ---- Bethell & Bergin 2011, Table 2: X-ray cross sections per H of gas and dust ---------------------------------------------
    p = os.path.join(SRC, "load_Bethell_Xray.f90")
    arrs = arrays_of(p, int_parameters(p))
    assert arrs["E_r"].shape == (2, 16) and arrs["c_g"].shape == (3, 16) and arrs["c_d"].shape == (3, 16)
    write_tables(os.path.join(ROOT, "data", "bethell2011_xray_cross.dat"),
                 ["X-ray photoabsorption cross sections per H nucleus of Bethell & Bergin 2011 (ApJ 740, 7), Table 2, as the reference (rac-2d)",
                  "tabulates them: 16 energy bands E_r(2, band) [keV], polynomial coefficients c_g(3, band) (gas) and c_d(3, band) (dust) of",
                  "sigma = 1e-24 cm^2 / E^3 (c1 + c2 E + c3 E^2).  Written by tools/extract_reference_tables.py; arrays in column-major order."],
                 [("", {"E_r": arrs["E_r"], "c_g": arrs["c_g"], "c_d": arrs["c_d"]})])
    print("wrote data/neufeld_cooling_tables.dat, data/visser2009_co_shielding.dat, data/bethell2011_xray_cross.dat")


if __name__ == "__main__":
    main()
