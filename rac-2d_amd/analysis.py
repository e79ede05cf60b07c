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


def analysed_records(touts, record, n_record_real, incr=0, frac=0.1):
    """The records chem_analyse visits (reference src/disk.f90:4176-4208): every incr-th of the first n_record_real (incr <= 0: 1 +
    n_record_real / 20), skipping a record whose largest relative change of y (and T) against the record BEFORE it is below frac times the
    relative step of t.  1-based record numbers."""
    touts = np.asarray(touts, dtype=np.float64); record = np.asarray(record, dtype=np.float64)
    if incr <= 0:
        incr = 1 + n_record_real // 20
    out = []
    for k in range(1, n_record_real + 1, incr):
        if k >= 2:
            a, b = record[k - 1], record[k - 2]
            dy_y = float(np.max(np.abs(a - b) / (a + b + 1e-15)))
            dt_t = (touts[k - 1] - touts[k - 2]) / (touts[k - 1] + touts[k - 2])
            if dy_y < frac * dt_t:
                continue
        out.append(k)
    return out


def chem_analyse(net, out_dir, cell_id, cell, touts, record, n_record_real, rates_of, species=(), iteration=1, geometry=(0.0, 0.0, 0.0, 0.0),
                 incr=0, tables=None):
    """chem_analyse (reference src/disk.f90:4136-4300) on a cell's time record as the engine returns it (Network.evol_solve_batch /
    evolT_solve_batch with record=True): the three files of a_disk_ana_params%analyse_out_dir,
      evol_<id>_rz_<xmin>_<ymin>_iter_<it>.dat    '!Time_(yr)', names, 'Tgas' in A14; one ES14.4E4 row (t, y, T) per record
      ele_...dat, contri_...dat                   a header line with the cell's conditions, then per visited record (analysed_records)
                                                  the elemental-residence block / the production-destruction ranking of `species`
    rates_of(Tgas) -> rate coefficients at that temperature (e.g. lambda T: net.cal_rates(params, cell_at(T))[0]): the reference
    re-evaluates nothing there with T fixed, and with T evolving the record's last column is the temperature of the snapshot.
    The ranking of reactions by chemical heat at the end of each contri block needs the heating/cooling module's per-reaction heats and is
    written by the reference only; everything else follows it line by line.  Returns the three paths and the visited records."""
    import os
    cell = np.asarray(cell, dtype=np.float64); touts = np.asarray(touts, dtype=np.float64); record = np.asarray(record, dtype=np.float64)
    nS = net.nSpecies
    xmin, xmax, ymin, ymax = geometry
    pre = "%04d_rz_%s_%s_iter_%03d" % (cell_id, ("%.6f" % xmin).lstrip("0") or "0", ("%.6f" % ymin).lstrip("0") or "0", iteration)  # (F0.6)
    paths = [os.path.join(out_dir, "%s_%s.dat" % (k, pre)) for k in ("evol", "ele", "contri")]
    with open(paths[0], "w") as f:
        f.write("!Time_(yr)    " + "".join("  " + ("%-12s" % nm)[:12] for nm in net.names) + "  Tgas        \n")  # (A14 of character(12) names: two blanks, then the name)
        for k in range(n_record_real):
            f.write(_es(touts[k], 14, 4, 4) + "".join(_es(v, 14, 4, 4) for v in record[k, :nS + 1]) + "\n")
    from . import cells as Cc
    head = "%10.1f%10.1f%s%s%s%5d%5d%s%s%s%s\n" % (cell[Cc.P_TGAS], cell[Cc.P_TDUST], _es(cell[Cc.P_NGAS], 12, 2), _es(cell[Cc.P_AV_STAR], 12, 2),
                                                   _es(cell[Cc.P_AV_ISM], 12, 2), cell_id, iteration, _es(xmin, 16, 6), _es(xmax, 16, 6), _es(ymin, 16, 6), _es(ymax, 16, 6))
    visited = analysed_records(touts, record[:, :nS + 1], n_record_real, incr)
    with open(paths[1], "w") as f1, open(paths[2], "w") as f2:
        f1.write(head); f2.write(head)
        for k in visited:
            y = record[k - 1, :nS]; T = record[k - 1, nS]
            write_elements(f1, net, touts[k - 1], y, T)
            if len(species) == 0:
                f2.write("Time = %s\n" % _es(touts[k - 1], 14, 4))
                continue
            c = cell.copy(); c[Cc.P_TGAS] = T
            write_contributions(f2, net, touts[k - 1], y, rates_of(T), c, list(species), tables)
            f2.write("Tgas = %s\n" % _es(T, 14, 4))
    return paths, visited


# ---------------------------------------------------------------------------------------------------------
# iter_NNNN.dat: the per-cell ASCII table of a global iteration (reference write_header / disk_save_results_write,
# src/disk.f90:2745-2902, 2905-3073).  Row = 2I5, 4I14, 142 ES14.5E3, nSpecies ES14.5E3; header = '!' + the column names right-aligned
# in the same widths (cvg 4, qual 5, everything else 14).  The reference's readers (utils_python/draw/misc.py::load_data_as_dic) take
# the header's whitespace-separated words as keys and numpy.loadtxt(comments='!') columns as values.
# ---------------------------------------------------------------------------------------------------------
ITER_INT_COLUMNS = ("cvg", "qual", "cr_count", "abc_dus", "scc_HI", "abc_wat")
ITER_REAL_COLUMNS = (
    "t_final", "rmin", "rmax", "zmin", "zmax", "n_gas", "Tgas", "Tdust", "Tdust1", "Tdust2", "Tdust3", "Tdust4", "ndust_1", "ndust_2", "ndust_3",
    "ndust_4", "ndust_t", "rhodus_1", "rhodus_2", "rhodus_3", "rhodus_4", "sigdus_1", "sigdus_2", "sigdus_3", "sigdus_4", "sigd_av", "d2gmas",
    "d2gnum", "deplet", "mg_cell", "md_cell", "presr_t", "presr_g", "egain_d", "egain_ab", "egain_e", "egain_d1", "egain_e1", "egain_d2",
    "egain_e2", "egain_d3", "egain_e3", "egain_d4", "egain_e4", "flx_tot", "flx_Xray", "G0_UV", "flx_Lya", "flx_Vis", "flx_NIR", "flx_MIR",
    "flx_FIR", "vr_tot", "vz_tot", "ani_tot", "vr_Xray", "vz_Xray", "ani_Xray", "vr_UV", "vz_UV", "ani_UV", "vr_Lya", "vz_Lya", "ani_Lya", "vr_Vis",
    "vz_Vis", "ani_Vis", "vr_NIR", "vz_NIR", "ani_NIR", "vr_MIR", "vz_MIR", "ani_MIR", "vr_FIR", "vz_FIR", "ani_FIR", "Av_ISM", "Av_Star", "UV_G0_I",
    "UV_G0_S", "LyAG0_a", "LyANF0", "zeta_X", "Ncol_I", "Ncol_S", "N_H2_I", "N_H2O_I", "N_OH_I", "N_CO_I", "N_H2_S", "N_H2O_S", "N_OH_S", "N_CO_S",
    "f_H2_I", "f_H2O_I", "f_OH_I", "f_CO_I", "f_H2_S", "f_H2O_S", "f_OH_S", "f_CO_S", "R_H2_fo", "hc_net", "h_ph_gr", "h_fo_H2", "h_cosmi", "h_vi_H2",
    "h_io_CI", "h_ph_H2", "h_ph_wa", "h_ph_OH", "h_Xray", "h_visco", "h_chem", "c_el_gr", "c_vi_H2", "c_gg_co", "c_OI", "c_CII", "c_NII", "c_SiII",
    "c_FeII", "c_OH_ro", "c_wa_ro", "c_wa_vi", "c_CO_ro", "c_CO_vi", "c_H2_ro", "c_LyAlp", "c_fb", "c_ff", "alpha", "am", "ion_cha", "v_Kep", "w_Kep",
    "dv_dr", "c_sound", "dv_turb", "l_coher", "nsit_gr", "nmol_gr")
assert len(ITER_REAL_COLUMNS) == 142
# where the 29 values of type_heating_cooling_rates_list (HC_TERM_NAMES order) sit in the row: the writer's order differs from the type's
_HC_ROW_ORDER = ("hc_net", "h_ph_gr", "h_fo_H2", "h_cosmi", "h_vi_H2", "h_io_CI", "h_ph_H2", "h_ph_wa", "h_ph_OH", "h_Xray", "h_visco", "h_chem",
                 "c_el_gr", "c_vi_H2", "c_gg_co", "c_OI", "c_CII", "c_wa_ro", "c_wa_vi", "c_CO_ro", "c_CO_vi", "c_H2_ro", "c_LyAlp", "c_fb", "c_ff",
                 "c_NII", "c_SiII", "c_FeII", "c_OH_ro")


def iter_file_columns(cells, y, t_final, quality, cell_out, names, hc=None, hc_terms=None, geometry=None, col_den=None, converged=None):
    """The columns of iter_NNNN.dat this path owns or is handed, as a dict name -> [ncell] array; everything that belongs to subsystems
    out of scope here (photon counters, fluxes and their anisotropies, masses, pressures, dust energy exchange) is what the reference
    writes for a cell those subsystems have not touched: zero.
      cells [ncell, NPAR], y [ncell, nS] (handed-back abundances), t_final, quality, cell_out [ncell, NOUT] of a solve call;
      hc [ncell, NHC] heating/cooling records (dust components, omega_Kepler, ...), hc_terms [ncell, 29] (Network.ode_f_evolT terms);
      geometry = (rmin, rmax, zmin, zmax) arrays [AU]; col_den = dict with N_H2_I ... N_CO_S; converged [ncell] 0/1."""
    from . import cells as Cc
    cells = np.asarray(cells, dtype=np.float64); n = cells.shape[0]
    col = {k: np.zeros(n) for k in ITER_INT_COLUMNS + ITER_REAL_COLUMNS}
    col["cvg"] = np.zeros(n) if converged is None else np.asarray(converged, dtype=np.float64)
    col["qual"] = np.asarray(quality, dtype=np.float64)
    col["t_final"] = np.asarray(t_final, dtype=np.float64)
    if geometry is not None:
        col["rmin"], col["rmax"], col["zmin"], col["zmax"] = (np.asarray(a, dtype=np.float64) for a in geometry)
    co = np.asarray(cell_out, dtype=np.float64)
    col["n_gas"] = cells[:, Cc.P_NGAS]; col["Tgas"] = np.where(np.isfinite(co[:, 3]), co[:, 3], cells[:, Cc.P_TGAS]); col["Tdust"] = cells[:, Cc.P_TDUST]
    col["ndust_t"] = cells[:, Cc.P_NDUST]; col["sigd_av"] = cells[:, Cc.P_SIGDUST]; col["d2gnum"] = cells[:, Cc.P_D2H]
    col["Av_ISM"] = cells[:, Cc.P_AV_ISM]; col["Av_Star"] = cells[:, Cc.P_AV_STAR]; col["UV_G0_I"] = cells[:, Cc.P_G0_ISM]; col["UV_G0_S"] = cells[:, Cc.P_G0_STAR]
    col["LyANF0"] = cells[:, Cc.P_LYA]; col["LyAG0_a"] = cells[:, Cc.P_LYA] / 6e7  # G0_Lya_atten = phflux_Lya / phy_Habing_photon_flux_CGS (src/disk.f90:1880)
    col["zeta_X"] = cells[:, Cc.P_ZETA_X]; col["Ncol_I"] = cells[:, Cc.P_NCOL_ISM]
    for k, slot in (("f_H2_I", Cc.P_FSS_ISM_H2), ("f_H2O_I", Cc.P_FSS_ISM_H2O), ("f_OH_I", Cc.P_FSS_ISM_OH), ("f_CO_I", Cc.P_FSS_ISM_CO),
                    ("f_H2_S", Cc.P_FSS_STAR_H2), ("f_H2O_S", Cc.P_FSS_STAR_H2O), ("f_OH_S", Cc.P_FSS_STAR_OH), ("f_CO_S", Cc.P_FSS_STAR_CO)):
        col[k] = cells[:, slot]
    col["nsit_gr"] = cells[:, Cc.P_SITES]; col["nmol_gr"] = np.nan_to_num(co[:, 1])
    # R_H2_form_rate = get_H2_form_rate(R_H2_form_rate_coeff, X(gH), X(H), n_gas) (src/disk.f90:4302-4315)
    names = list(names); y = np.asarray(y, dtype=np.float64)
    coeff = np.nan_to_num(co[:, 0])
    if "gH" in names:
        col["R_H2_fo"] = coeff * y[:, names.index("gH")] ** 2 * cells[:, Cc.P_NGAS]
    elif "H" in names:
        col["R_H2_fo"] = coeff * y[:, names.index("H")] * cells[:, Cc.P_NGAS]
    if hc is not None:
        hc = np.asarray(hc, dtype=np.float64)
        for i in range(4):
            col["Tdust%d" % (i + 1)] = hc[:, Cc.H_TDUSTS + i]; col["ndust_%d" % (i + 1)] = hc[:, Cc.H_N_DUSTS + i]
            col["sigdus_%d" % (i + 1)] = hc[:, Cc.H_SIG_DUSTS + i]; col["egain_d%d" % (i + 1)] = hc[:, Cc.H_EN_GAINS + i]
        col["egain_d"] = hc[:, Cc.H_EN_GAIN_TOT]; col["deplet"] = hc[:, Cc.H_DUST_DEPL]; col["Ncol_S"] = hc[:, Cc.H_NCOL_STAR]
        col["w_Kep"] = hc[:, Cc.H_OMEGA_K]; col["dv_turb"] = hc[:, Cc.H_DV_TURB]; col["l_coher"] = hc[:, Cc.H_COHERENT]
    if hc_terms is not None:
        ht = np.asarray(hc_terms, dtype=np.float64)
        for k, name in enumerate(_HC_ROW_ORDER):
            col[name] = ht[:, k]
    if col_den is not None:
        for k, v in col_den.items():
            col[k] = np.asarray(v, dtype=np.float64)
    return col


def _f_es14(v):
    """Fortran ES14.5E3 of one value"""
    if not np.isfinite(v):
        return "%14s" % ("NaN" if np.isnan(v) else ("Inf" if v > 0 else "-Inf"))
    s = "%.5E" % v
    mant, ex = s.split("E")
    return "%14s" % ("%sE%s%03d" % (mant, ex[0], int(ex[1:])))


def write_iter_dat(path, names, columns, y):
    """iter_NNNN.dat in the reference's layout (write_header + one disk_save_results_write row per cell)."""
    names = list(names); y = np.asarray(y, dtype=np.float64)
    with open(path, "w") as f:
        f.write("!" + "cvg".rjust(4) + "qual".rjust(5) + "".join(k.rjust(14) for k in ITER_INT_COLUMNS[2:] + ITER_REAL_COLUMNS)
                + "".join("  " + ("%-12s" % nm)[:12] for nm in names) + "\n")  # (A14 of a character(12) name: two blanks, then the name)
        for c in range(y.shape[0]):
            f.write("%5d%5d" % (int(columns["cvg"][c]), int(columns["qual"][c])) + "".join("%14d" % int(columns[k][c]) for k in ITER_INT_COLUMNS[2:])
                    + "".join(_f_es14(float(columns[k][c])) for k in ITER_REAL_COLUMNS) + "".join(_f_es14(float(v)) for v in y[c]) + "\n")


def load_iter_dat(path):
    """The parsing rules of the reference's own reader (utils_python/draw/misc.py::load_data_as_dic): keys = the whitespace-separated
    words of the first line without its first character, values = the columns of numpy.loadtxt(comments='!')."""
    data = np.loadtxt(path, comments="!", ndmin=2)
    with open(path) as f:
        keys = f.readline()[1:].split()
    return {k: data[:, i] for i, k in enumerate(keys)}
