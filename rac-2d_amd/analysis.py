"""Analysis outputs of the reference around the per-cell solve, from the arrays the engine returns (host side, numpy).

  reaction_fluxes        chem_ode_f_alt            reference src/chemistry.f90:1792-1854  (flux of every reaction at a composition)
  contributions          get_species_produ_destr + get_contribution_each   :1643-1789       (which reactions make / destroy a species)
  elemental_residence    chem_elemental_residence  :1593-1640                              (which species hold an element)
  write_rate_dump        save_chem_rates           src/disk.f90:3555-3592                  (per-cell rate dump; the cell-record namelist
                                                                                            that precedes the rows there is the caller's)
  write_contributions / write_elements            the per-snapshot blocks chem_analyse writes, src/disk.f90:4196-4262

`net` is a rac-2d_amd.Network; `rates` the rate coefficients of the cell (Network.cal_rates), `y` abundances [nS]."""
import numpy as np

ELEMENT_NAMES = ["+-", "E", "Grain", "H", "D", "He", "C", "N", "O", "Si", "S", "Fe", "Na", "Mg", "Cl", "P", "F", "Ne", "Ar", "K"]
ELE_RESI_NMAX, ELE_FRAC_TO_SUM, ELE_FRAC_TO_MAX = 10, 0.999, 1e-5   # src/chemistry.f90:175-177


def reaction_fluxes(net, rates, y, cell):
    """chem_ode_f_alt: note that it is NOT chem_ode_f (no sign rules, surface forms linearised below 1e-9 instead of 1e-4)."""
    rx = net.reactions()
    it = rx["itype"]; a = rx["reac"][:, 0] - 1; b = rx["reac"][:, 1] - 1
    ya = np.where(a >= 0, y[np.maximum(a, 0)], 0.0); yb = np.where(b >= 0, y[np.maximum(b, 0)], 0.0)
    r = np.zeros(net.nReactions)
    two = np.isin(it, (5, 6, 21, 64)); one = np.isin(it, (1, 2, 3, 13, 61, 0, 20)); sq = it == 63
    r[two] = rates[two] * ya[two] * yb[two]
    r[one] = rates[one] * ya[one]
    r[sq] = rates[sq] * ya[sq] * ya[sq]
    nsite = cell[6] * cell[7]
    abc3 = net.reaction_rows()["ABC"][:, 2]
    for kind, den in ((62, np.full(net.nReactions, nsite)), (75, nsite * abc3)):
        m = it == kind
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ya[m] / den[m]
            r[m] = np.where(t <= 1e-9, rates[m] * t, rates[m] * (1.0 - np.exp(-t)))
    return r


def _produ_destr(net):
    """get_species_produ_destr: per species the reactions it is a product / reactant of (each once), and how often it appears."""
    rx = net.reactions()
    nS = net.nSpecies
    produ = [dict() for _ in range(nS)]; destr = [dict() for _ in range(nS)]
    for i in range(net.nReactions):
        for j in range(rx["n_reac"][i]):
            s = rx["reac"][i, j] - 1
            destr[s][i] = destr[s].get(i, 0) + 1
        for j in range(rx["n_prod"][i]):
            s = rx["prod"][i, j] - 1
            produ[s][i] = produ[s].get(i, 0) + 1
    return produ, destr


def contributions(net, rates, y, cell, species, tables=None):
    """get_contribution_each for the listed species (1-based indices): {species: (produ, destr)}, each a list of
    (reaction (1-based), contribution) sorted by decreasing contribution."""
    flux = reaction_fluxes(net, rates, y, cell)
    produ, destr = tables or _produ_destr(net)
    out = {}
    for sp in species:
        res = []
        for tab in (produ[sp - 1], destr[sp - 1]):
            lst = [(i + 1, n * flux[i]) for i, n in tab.items()]
            lst.sort(key=lambda e: -e[1])
            res.append(lst)
        out[sp] = tuple(res)
    return out


def elemental_residence(net, y):
    """chem_elemental_residence: per element (20 of them) the species holding most of it: list of (species (1-based), fraction,
    accumulated fraction), at most 10, cut where 99.9 % is reached or a fraction falls below 1e-5 of the largest."""
    el = net.species_elements().astype(np.float64)
    out = []
    for e in range(20):
        v = y * el[:, e]
        order = np.argsort(-np.abs(v), kind="stable")
        tot = np.abs(v).sum()
        rows, accum = [], 0.0
        for j in range(ELE_RESI_NMAX):
            i0 = order[j]
            accum += abs(v[i0])
            if abs(v[i0]) >= 1e-90:
                frac, accu = v[i0] / tot, accum / tot
            else:
                frac, accu = 0.0, 0.0
            rows.append((int(i0) + 1, frac, accu))
            if accum >= ELE_FRAC_TO_SUM * tot or abs(frac) <= ELE_FRAC_TO_MAX * rows[0][1]:
                break
        out.append(rows)
    return out


def _double2str(x, nw, np_):
    """double2str of the reference (src/sub_trivials.f90:1278-1289): F<nw>.<np>, or ES<nw>.<np> when that overflows."""
    s = "%*.*f" % (nw, np_, x)
    if len(s) > nw:
        m, e = ("%.*E" % (np_, x)).split("E")
        s = "%sE%+03d" % (m, int(e))
        s = s.rjust(nw)
    return s


def _es(x, w, d, e=2):
    """Fortran ESw.dEe."""
    if x == 0.0 or not np.isfinite(x):
        m, ex = "%.*f" % (d, 0.0 if x == 0 else x), 0
    else:
        m, ex = ("%.*E" % (d, x)).split("E"); ex = int(ex)
    return ("%sE%s%0*d" % (m, "+" if ex >= 0 else "-", e, abs(ex))).rjust(w)


def write_rate_dump(path, net, rates):
    """The reaction rows of save_chem_rates: '(7(A12), ES9.2, F9.2, A9, 2I6, I3, X, A1, X, A2, ES16.6E3)'."""
    rr = net.reaction_rows()
    it = net.reactions()["itype"]
    with open(path, "w") as f:
        for k in range(net.nReactions):
            f.write("".join(rr["names"][k]) + _es(rr["ABC"][k, 0], 9, 2) + "%9.2f" % rr["ABC"][k, 1] + _double2str(rr["ABC"][k, 2], 9, 1)
                    + "%6d%6d" % (int(rr["T_range"][k, 0]), int(rr["T_range"][k, 1])) + "%3d" % it[k] + " " + rr["reliability"][k] + " "
                    + rr["ctype"][k] + _es(rates[k], 16, 6, 3) + "\n")


def write_elements(f, net, t, y, Tgas):
    """One snapshot block of chem_analyse's ele_*.dat (src/disk.f90:4206-4226)."""
    el = net.species_elements()
    f.write("Time = %s\n" % _es(t, 14, 4)); f.write("Tgas = %s\n" % _es(Tgas, 14, 4))
    f.write("    Total net charge: %s\n" % _es(float((y * el[:, 0]).sum()), 10, 2))
    f.write("    Total free charge: %s\n" % _es(float((y * np.abs(el[:, 0])).sum()) / 2.0, 10, 2))
    for e, rows in enumerate(elemental_residence(net, y)):
        f.write("    %-8s\n" % ELEMENT_NAMES[e])
        for i0, frac, accu in rows:
            f.write("      %-12s%s%s%s\n" % (net.names[i0 - 1], _es(y[i0 - 1], 10, 2), _es(frac, 10, 2), _es(accu, 10, 2)))


def write_contributions(f, net, t, y, rates, cell, species, tables=None):
    """One snapshot block of chem_analyse's contri_*.dat (src/disk.f90:4228-4262) for the listed species."""
    rr = net.reaction_rows()
    f.write("Time = %s\n" % _es(t, 14, 4))
    con = contributions(net, rates, y, cell, species, tables)
    for sp in species:
        f.write("%-12s%s\n" % (net.names[sp - 1], _es(y[sp - 1], 12, 2)))
        for title, lst in zip(("Production", "Destruction"), con[sp]):
            tot = sum(c for _, c in lst) + 1e-100
            f.write("  %s  %s\n" % (title, _es(tot, 12, 2)))
            accum = 0.0
            for j, (i0, c) in enumerate(lst[:20]):
                accum += c
                k = i0 - 1
                f.write("    %4d%s%s%8.2f%s  %s%s%9.2f%9.2f%8.1f%8.1f\n" % (
                    j + 1, _es(c, 12, 2), _es(accum, 12, 2), accum / tot, _es(rates[k], 12, 2), "".join(rr["names"][k][0:2] + rr["names"][k][3:7]),
                    _es(rr["ABC"][k, 0], 12, 2), rr["ABC"][k, 1], rr["ABC"][k, 2], rr["T_range"][k, 0], rr["T_range"][k, 1]))
                if c <= lst[0][1] * 1e-6:
                    break
