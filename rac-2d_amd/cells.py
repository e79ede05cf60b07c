"""Synthetic per-cell input records (the A4 fields of SURVEY.md section 8(a)).

A cell record is NPAR = 28 float64 values in the order declared in include/racgpu.h (RACGPU_P_*),
mirroring the fields of the reference's type_cell_rz_phy_basic that the fixed-temperature path reads
(reference: src/data_struct.f90:316-442).  Derived dust quantities follow the reference's own formulas:
GrainRadius = sqrt(sigma/pi), SitesPerGrain = 4 sigma 1e15, ratioDust2HnucNum = ndust/n_gas
(src/vertical_structure.f90:205-206,219) and the single-cell test recipe src/test_cases_bak.f90:47-56.
"""
import math

import numpy as np

NPAR = 28
MP_CGS = 1.67262158e-24
(P_TGAS, P_TDUST, P_NGAS, P_GRAIN_RADIUS, P_SIGDUST, P_NDUST, P_D2H, P_SITES, P_ALBEDO, P_ZETA_CR,
 P_ZETA_X, P_NCOL_ISM, P_AV_ISM, P_AV_STAR, P_G0_ISM, P_G0_STAR, P_G0_H2PHD, P_G0_PHOTODES, P_LYA,
 P_FSS_ISM_H2, P_FSS_ISM_CO, P_FSS_ISM_H2O, P_FSS_ISM_OH, P_FSS_STAR_H2, P_FSS_STAR_CO,
 P_FSS_STAR_H2O, P_FSS_STAR_OH, P_TMAX) = range(NPAR)


def make_cell(Tgas, Tdust, n_gas, Av, G0_star, a_cm=1e-5, t_max=0.0, f_H2=1e-4, f_CO=1e-2):
    """One record: 0.1 um grains of 2 g cm^-3, dust/gas mass ratio 0.01, mean weight 1.4."""
    sig = math.pi * a_cm * a_cm
    d2h = 0.01 * 1.4 * MP_CGS / (4.0 * math.pi / 3.0 * a_cm ** 3 * 2.0)
    att = G0_star * math.exp(-2.6 * Av)
    return np.array([Tgas, Tdust, n_gas, a_cm, sig, n_gas * d2h, d2h, 4.0 * sig * 1e15, 0.5, 1.36e-17, 0.0,
                     Av / 5.3e-22, Av, Av, 1.0, G0_star, att, att, 0.0,
                     f_H2, f_CO, 1.0, 1.0, f_H2, f_CO, 1.0, 1.0, t_max], dtype=np.float64)


def synth_batch(ncell, seed=20240601):
    """BASELINE config 2: log-uniform T in [10, 3000] K, n_H in [1e3, 1e12] cm^-3.

    Tdust = min(T, 1500); log10 G0_star ~ U[-2, 6]; Av = 10^U[-2, 2]; H2/CO shielding factors fall with
    Av as min(1, 1e-4 / Av) and min(1, 1e-2 / Av) (a stated closed form, not a physical model)."""
    rng = np.random.default_rng(seed)
    T = 10.0 ** rng.uniform(1.0, math.log10(3000.0), ncell)
    n = 10.0 ** rng.uniform(3.0, 12.0, ncell)
    g0 = 10.0 ** rng.uniform(-2.0, 6.0, ncell)
    av = 10.0 ** rng.uniform(-2.0, 2.0, ncell)
    out = np.empty((ncell, NPAR))
    for i in range(ncell):
        out[i] = make_cell(T[i], min(T[i], 1500.0), n[i], av[i], g0[i],
                           f_H2=min(1.0, 1e-4 / av[i]), f_CO=min(1.0, 1e-2 / av[i]))
    return out


def tmax_this(omega_kepler, t_max0=1e6, n_orbit_tmax=1e5, use_fixed_tmax=False):
    """Per-cell integration time of the caller's sweep (reference set_initial_condition_4solver, src/disk.f90:2017-2018,
    2078-2085): t_max0 itself with use_fixed_tmax, else min(t_max0, max(100 yr, nOrbit_tmax orbital periods)); omega_kepler
    in rad/s (par%omega_Kepler), result in years (the reference's year is 365 days).  Goes into slot P_TMAX of the cell
    record; 0 there means "params.t_max"."""
    omega = np.asarray(omega_kepler, dtype=np.float64)
    if use_fixed_tmax:
        return np.full_like(omega, float(t_max0))
    seconds_per_year = 3600.0 * 24.0 * 365.0
    two_pi = 6.283185307179586476925
    return np.minimum(float(t_max0), np.maximum(1e2, n_orbit_tmax * two_pi / omega / seconds_per_year))


# ---------------------------------------------------------------------------------------------------------
# BASELINE configs[2]/[3]: a synthetic Andrews-2009-like (r, z) grid
# ---------------------------------------------------------------------------------------------------------
MSUN_CGS = 1.9891e33          # reference src/sub_global_variables.f90:38
AU_CM = 1.49597871e13         # :55
G_CGS = 6.67428e-8            # :27
SEC_PER_YEAR = 3600.0 * 24.0 * 365.0


def andrews_density(r_au, z_au, Md=2e-2, rin=0.1, rout=200.0, rc=80.0, hc=10.0, gam=1.5, psi=1.0,
                    r0_in_exp=3.5, rs_in_exp=1e2, p_in_exp=1.0, f_in_exp=1e-5, particlemass=1.4 * MP_CGS):
    """n_H [cm^-3] of the reference's analytic gas disk, reference src/grid.f90:1741-1818 (Andrews_dens) with the
    README's `a_disk%andrews_gas` values (README.md:225-235): Md 0.02 Msun, rc 80 AU, hc 10 AU, gamma 1.5, psi 1,
    inner taper below 3.5 AU (factor 1e-5), no outer taper, no flattening."""
    r = np.asarray(r_au, dtype=np.float64)
    z = np.asarray(z_au, dtype=np.float64)
    sigma_c = (2.0 - gam) * Md / (2.0 * math.pi * rc ** 2) / (math.exp(-(rin / rc) ** (2.0 - gam)) - math.exp(-(rout / rc) ** (2.0 - gam)))
    rrc = r / rc
    taper = np.where(r < r0_in_exp, np.exp(-((r0_in_exp - r) / rs_in_exp) ** p_in_exp) * f_in_exp, 1.0)
    sigma = sigma_c * rrc ** (-gam) * np.exp(-rrc ** (2.0 - gam)) * taper  # Msun / AU^2
    h = hc * rrc ** psi
    n = sigma / (math.sqrt(2.0 * math.pi) * h) * np.exp(-0.5 * (z / h) ** 2) * MSUN_CGS / (AU_CM ** 3 * particlemass)
    return np.where((r < rin) | (r > rout), 0.0, n), h


def andrews_grid(ncol=200, nz=100, rmin=0.1, rmax=200.0, zr_max=0.6, star_mass_msun=0.6, t_max0=1e6,
                 n_orbit_tmax=1e5, use_fixed_tmax=False, a_cm=1e-5, return_geometry=False, Md=2e-2):
    """BASELINE configs[2]: ~20 k frozen cell records of a full 2-D disk grid.

    The real grid needs the reference's Monte-Carlo radiative transfer (out of scope), so every field below is a
    STATED CLOSED FORM of the analytic Andrews gas disk -- a workload with the reference grid's structure
    (columns inner -> outer, cells top -> bottom as the sweep visits them, src/disk.f90:885-937, 2512-2520) and its
    dynamic range (n_H 1e3...1e14, Av 1e-4...1e5, t_max by the orbit rule), not a physical model:

      geometry   ncol columns log-spaced in r in [rmin, rmax] AU; nz cells per column at z/r = (k+1/2)/nz * zr_max,
                 ordered top -> bottom within a column, columns inner -> outer
      n_gas      andrews_density(r, z), floored at 1e2 cm^-3
      columns    N_toISM(r,z) = int_z^inf n dz' (vertical, analytic erfc); N_toStar(r,z) = int along the ray from the star
                 at constant z/r (h ~ r for psi = 1, so the Gaussian factor is constant along the ray)
      Av         N * 5.3e-22 (both directions; the reference's calc_Av_*_from_Ncol conversion)
      UV         G0_UV_toISM = 1; G0_UV_toStar = 1e3 (100 AU / d)^2, d^2 = r^2 + z^2; G0_UV_H2phd =
                 G0_UV_toStar_photoDesorb = G0_UV_toStar e^{-2.6 Av_toStar}; phflux_Lya = 1e7 G0_UV_toStar/1e3 e^{-2.6 Av_toStar}
      ionisation zeta_CR = 1.36e-17; zeta_Xray = 1e-13 (1 AU / d)^2 e^{-N_toStar / 1e24}
      dust       0.1 um grains, dust/gas mass ratio 0.01 (make_cell), albedo 0.5
      Tdust      T_mid = 120 K (r / 1 AU)^-0.5 (floor 8 K) in the shadow, T_thin = 280 K (d / 1 AU)^-0.5 unshadowed:
                 Tdust = T_mid + (T_thin - T_mid) e^{-Av_toStar}, clipped to [5, 1500] K
      Tgas       Tdust (1 + 9 e^{-2.6 Av_toStar} / (1 + n_gas / 1e7)), clipped to <= 5000 K (hot, thin, irradiated surface)
      shielding  f_H2 = Draine-Bertoldi 1996 eq. 37 (reference src/disk.f90:1887-1897, b5 = 1) of N_H2 = 0.1 N;
                 f_CO = min(1, (1e-5 N / 1e15)^-0.75); f_H2O = min(1, (1e-6 N / 1e17)^-0.5);
                 f_OH = min(1, (1e-7 N / 1e17)^-0.5), N the column in the respective direction
      t_max      tmax_this(omega_Kepler) (orbit rule, src/disk.f90:2078-2085) unless use_fixed_tmax
    Md scales the gas mass (bench.py gives every rank of a weak-scaling run a disk of its own).
    """
    r = rmin * (rmax / rmin) ** ((np.arange(ncol) + 0.5) / ncol)            # column centres
    mu = zr_max * (nz - 0.5 - np.arange(nz)) / nz                            # z/r, top -> bottom
    R, MU = np.meshgrid(r, mu, indexing="ij")                                 # [ncol, nz]
    Z = R * MU
    n, H = andrews_density(R, Z, Md=Md)
    n_mid, _ = andrews_density(r, 0.0 * r, Md=Md)
    n = np.maximum(n, 1e2)
    from math import erfc
    erfc_v = np.vectorize(erfc)
    N_ism = n_mid[:, None] * H * AU_CM * math.sqrt(math.pi / 2.0) * erfc_v(Z / (math.sqrt(2.0) * H))
    # towards the star: int_{rin}^{r} n_mid(r') dr' * sqrt(1 + mu^2) * exp(-(mu r'/h(r'))^2 / 2), the last factor constant for psi = 1
    edges = rmin * (rmax / rmin) ** (np.arange(ncol + 1) / ncol)
    cum = np.concatenate([[0.0], np.cumsum(n_mid * np.diff(edges))]) * AU_CM   # up to the outer edge of each column
    N_in = 0.5 * (cum[:-1] + cum[1:])                                           # up to the column centre
    gauss = np.exp(-0.5 * (MU * (80.0 / 10.0)) ** 2)
    N_star = N_in[:, None] * np.sqrt(1.0 + MU ** 2) * gauss
    Av_ism, Av_star = N_ism * 5.3e-22, N_star * 5.3e-22
    d2 = R * R + Z * Z
    g0 = 1e3 * (100.0 ** 2) / d2
    att = g0 * np.exp(-np.minimum(2.6 * Av_star, 700.0))
    lya = 1e7 * att / 1e3
    zx = 1e-13 / d2 * np.exp(-np.minimum(N_star / 1e24, 700.0))
    t_mid = np.maximum(120.0 * R ** -0.5, 8.0)
    t_thin = 280.0 * d2 ** -0.25
    tdust = np.clip(t_mid + (t_thin - t_mid) * np.exp(-np.minimum(Av_star, 700.0)), 5.0, 1500.0)
    tgas = np.minimum(tdust * (1.0 + 9.0 * np.exp(-np.minimum(2.6 * Av_star, 700.0)) / (1.0 + n / 1e7)), 5000.0)

    def f_h2(N):
        x = 0.1 * N / 5e14
        s = np.sqrt(1.0 + x)
        return 0.965 / (1.0 + x) ** 2 + 0.035 / s * np.exp(-8.5e-4 * s)

    def f_pow(N, x_mol, n0, p):
        return np.minimum(1.0, np.maximum(x_mol * N / n0, 1e-300) ** (-p))

    sig = math.pi * a_cm * a_cm
    d2h = 0.01 * 1.4 * MP_CGS / (4.0 * math.pi / 3.0 * a_cm ** 3 * 2.0)
    omega = np.sqrt(G_CGS * star_mass_msun * MSUN_CGS / (R * AU_CM) ** 3)
    tmax = tmax_this(omega, t_max0, n_orbit_tmax, use_fixed_tmax)
    out = np.empty((ncol, nz, NPAR))
    out[..., P_TGAS] = tgas; out[..., P_TDUST] = tdust; out[..., P_NGAS] = n; out[..., P_GRAIN_RADIUS] = a_cm
    out[..., P_SIGDUST] = sig; out[..., P_NDUST] = n * d2h; out[..., P_D2H] = d2h; out[..., P_SITES] = 4.0 * sig * 1e15
    out[..., P_ALBEDO] = 0.5; out[..., P_ZETA_CR] = 1.36e-17; out[..., P_ZETA_X] = zx; out[..., P_NCOL_ISM] = N_ism
    out[..., P_AV_ISM] = Av_ism; out[..., P_AV_STAR] = Av_star; out[..., P_G0_ISM] = 1.0; out[..., P_G0_STAR] = g0
    out[..., P_G0_H2PHD] = att; out[..., P_G0_PHOTODES] = att; out[..., P_LYA] = lya
    out[..., P_FSS_ISM_H2] = f_h2(N_ism); out[..., P_FSS_ISM_CO] = f_pow(N_ism, 1e-5, 1e15, 0.75)
    out[..., P_FSS_ISM_H2O] = f_pow(N_ism, 1e-6, 1e17, 0.5); out[..., P_FSS_ISM_OH] = f_pow(N_ism, 1e-7, 1e17, 0.5)
    out[..., P_FSS_STAR_H2] = f_h2(N_star); out[..., P_FSS_STAR_CO] = f_pow(N_star, 1e-5, 1e15, 0.75)
    out[..., P_FSS_STAR_H2O] = f_pow(N_star, 1e-6, 1e17, 0.5); out[..., P_FSS_STAR_OH] = f_pow(N_star, 1e-7, 1e17, 0.5)
    out[..., P_TMAX] = tmax
    cells = np.ascontiguousarray(out.reshape(ncol * nz, NPAR))
    if return_geometry:
        return cells, np.ascontiguousarray(R.reshape(-1)), np.ascontiguousarray(Z.reshape(-1))
    return cells


# ---------------------------------------------------------------------------------------------------------
# The fields only the heating/cooling terms read (gas temperature co-evolving, evolT): racgpu.h RACGPU_H_*
# ---------------------------------------------------------------------------------------------------------
NHC = 28
(H_EN_GAIN_TOT, H_NCOL_STAR, H_PAH, H_MMW, H_OMEGA_K, H_DV_TURB, H_COHERENT, H_NEUFELD_G, H_NEUFELD_DVDZ, H_DUST_DEPL, H_VOLUME,
 H_NDUSTCOMPO) = range(12)
H_SIG_DUSTS, H_N_DUSTS, H_TDUSTS, H_EN_GAINS = 12, 16, 20, 24


def make_hc_record(cell, omega_kepler=2e-9, coherent_length=1e13, dv_turb=1e4, volume=1e39, en_gain=1e30, pah=1.6e-7):
    """One heating/cooling record to go with a cell record: a single dust component that IS the cell's dust (sig_dusts =
    sigdust_ave, n_dusts = ndust_tot, Tdusts = Tdust), Ncol_toStar from Av_toStar (5.3e-22 mag cm^2), ISM PAH abundance, mean
    molecular weight 1.4, Neufeld G = 1 and dv/dz = omega_Kepler in km s^-1 cm^-1, en_gain_tot > 0 (T evolves)."""
    cell = np.asarray(cell, dtype=np.float64)
    h = np.zeros(cell.shape[:-1] + (NHC,))
    h[..., H_EN_GAIN_TOT] = en_gain
    h[..., H_NCOL_STAR] = cell[..., P_AV_STAR] / 5.3e-22
    h[..., H_PAH] = pah; h[..., H_MMW] = 1.4; h[..., H_OMEGA_K] = omega_kepler; h[..., H_DV_TURB] = dv_turb
    h[..., H_COHERENT] = coherent_length; h[..., H_NEUFELD_G] = 1.0; h[..., H_NEUFELD_DVDZ] = np.asarray(omega_kepler) * 1e-5
    h[..., H_DUST_DEPL] = 1.0; h[..., H_VOLUME] = volume; h[..., H_NDUSTCOMPO] = 1.0
    h[..., H_SIG_DUSTS] = cell[..., P_SIGDUST]; h[..., H_N_DUSTS] = cell[..., P_NDUST]; h[..., H_TDUSTS] = cell[..., P_TDUST]
    h[..., H_EN_GAINS] = en_gain
    return h


def andrews_grid_hc(cells, r_au, z_au, star_mass_msun=0.6, nz=100, zr_max=0.6):
    """Heating/cooling records for andrews_grid(return_geometry=True), again STATED CLOSED FORMS in place of what the reference's
    Monte-Carlo radiative transfer and grid would supply: omega_Kepler from r; coherent length = the gas scale height h(r) = r/8;
    turbulent width 0.1 of the isothermal sound speed at Tdust; cell volume 2 pi r dr dz of the log-spaced grid; the dust's energy
    gain = sigma_SB Tdust^4 emission balance, 4 sig_dust n_dust sigma_SB Tdust^4 V (what the cap of the gas-grain term is measured
    against); one dust component; ISM PAH abundance scaled down by 0.1 (settled disk)."""
    cells = np.asarray(cells, dtype=np.float64)
    r = np.asarray(r_au, dtype=np.float64) * AU_CM
    omega = np.sqrt(G_CGS * star_mass_msun * MSUN_CGS / r ** 3)
    cs = np.sqrt(1.3806503e-16 * cells[:, P_TDUST] / (1.4 * MP_CGS))
    ncol = 200
    dlnr = math.log(200.0 / 0.1) / ncol
    vol = 2.0 * math.pi * r * (r * dlnr) * (r * zr_max / nz)
    gain = 4.0 * cells[:, P_SIGDUST] * cells[:, P_NDUST] * 5.670373e-5 * cells[:, P_TDUST] ** 4 * vol
    return make_hc_record(cells, omega_kepler=omega, coherent_length=r / 8.0, dv_turb=0.1 * cs, volume=vol, en_gain=gain, pah=1.6e-8)


# ---------------------------------------------------------------------------------------------------------
# Self-shielding factors of a cell from the column densities above it: the part of the caller's update_params_above_alt
# (reference src/disk.f90:1823-1883) that only needs numbers, for sweeps that refresh the records of a layer from the layers
# solved before it (sweep.solve_by_layers).  The column densities themselves come from the caller's grid (the reference traces
# rays through its quadtree, calc_Ncol_from_cell_to_point: out of scope); column_density_above does the vertical sum of a
# regular column grid.
# ---------------------------------------------------------------------------------------------------------
LYA_CROSS_H2O = 1.2e-17   # const_LyAlpha_cross_H2O, reference src/sub_global_variables.f90:82
LYA_CROSS_OH = 1.8e-18    # const_LyAlpha_cross_OH, :83


def h2_self_shielding(N_H2, dv_turb):
    """min(1, Draine & Bertoldi 1996 eq. 37): get_H2_self_shielding, reference src/disk.f90:1887-1897, as update_params_above_alt
    caps it (:1840-1843).  N_H2 [cm^-2], dv_turb [cm/s].  The second coefficient is a single-precision literal in the
    reference (0.035 without D0)."""
    x = np.asarray(N_H2, dtype=np.float64) / 5e14
    b5 = np.asarray(dv_turb, dtype=np.float64) / 1e5
    tmp = np.sqrt(1.0 + x)
    f = 0.965 / (1.0 + x / b5) ** 2 + np.float64(np.float32(0.035)) / tmp * np.exp(-8.5e-4 * tmp)
    return np.minimum(1.0, f)


def lya_self_shielding(N_col, cross_section):
    """min(1, exp(-N sigma)): the H2O and OH factors of update_params_above_alt (reference src/disk.f90:1847-1859) with
    LYA_CROSS_H2O / LYA_CROSS_OH."""
    return np.minimum(1.0, np.exp(-(np.asarray(N_col, dtype=np.float64) * cross_section)))


def co_shielding(table, N_H2, N_12CO):
    """12CO shielding by H2 and by itself from a caller-supplied table, with the algorithm of get_12CO_shielding (reference
    src/load_Visser_CO_selfshielding.f90:271-309): log10 of the column densities (floored at 1 cm^-2), the enclosing table
    cell (the last one beyond the table, the first one below it), four-point linear interpolation of ln f
    (calc_four_point_linear_interpol, src/sub_trivials.f90:803-821), clamped to [0, 1] as update_params_above_alt does.
    table = (logN_H2 [nrow], logN_12CO [ncol], f [ncol, nrow]), both axes ascending.  The reference's own table (Visser et
    al. 2009) is compiled into it; here it is the caller's data."""
    lh, lc, f = (np.asarray(a, dtype=np.float64) for a in table)
    x = np.log10(np.maximum(np.asarray(N_12CO, dtype=np.float64), 1.0))
    y = np.log10(np.maximum(np.asarray(N_H2, dtype=np.float64), 1.0))
    i1 = np.clip(np.searchsorted(lh, y, side="left") - 1, 0, lh.size - 2)
    j1 = np.clip(np.searchsorted(lc, x, side="left") - 1, 0, lc.size - 2)
    x1, x2, y1, y2 = lc[j1], lc[j1 + 1], lh[i1], lh[i1 + 1]
    z11, z12, z21, z22 = np.log(f[j1, i1]), np.log(f[j1, i1 + 1]), np.log(f[j1 + 1, i1]), np.log(f[j1 + 1, i1 + 1])
    k1 = (z12 - z11) / (y2 - y1)
    k2 = (z22 - z21) / (y2 - y1)
    v = ((k2 - k1) / (x2 - x1) * (x - x1) + k1) * (y - y1) + (z21 - z11) / (x2 - x1) * (x - x1) + z11
    return np.minimum(1.0, np.maximum(0.0, np.exp(v)))


def _load_tables(path):
    arrs = {}
    name, shape, vals = None, None, []
    for line in open(path):
        if line.startswith("!") or not line.strip():
            continue
        if line.startswith("#"):
            if name:
                arrs[name] = (shape, vals)
            t = line[1:].split()
            name, shape, vals = t[0], tuple(int(v) for v in t[1:]), []
        else:
            vals.extend(float(v) for v in line.split())
    if name:
        arrs[name] = (shape, vals)
    return {k: np.array(v, dtype=np.float64).reshape(sh, order="F") for k, (sh, v) in arrs.items()}  # (column-major, as Fortran holds them)


def load_xray_cross_sections(path):
    """data/bethell2011_xray_cross.dat (Bethell & Bergin 2011, Table 2, as the reference tabulates it in src/load_Bethell_Xray.f90):
    dict(E_r [2, 16] keV, c_g [3, 16], c_d [3, 16]) for sigma_xray_bethell."""
    return _load_tables(path)


def sigma_xray_bethell(table, E_keV, dust_depletion, d2h, grain_radius_cm):
    """sigma_Xray_Bethell (reference src/load_Bethell_Xray.f90:70-96): X-ray photoabsorption cross section per H nucleus [cm^2] at photon
    energy E [keV] -- gas plus dust, the dust part reduced by grain self-shielding f(tau), tau = sigma_dust / G * 3 / (2 pi) / a^2.
    The band is the first whose range holds E (the bands share their end points), the first / last one outside the table."""
    E = np.atleast_1d(np.asarray(E_keV, dtype=np.float64))
    Er, cg, cd = table["E_r"], table["c_g"], table["c_d"]
    inside = (E[:, None] >= Er[0][None, :]) & (E[:, None] <= Er[1][None, :])
    i0 = np.where(inside.any(1), inside.argmax(1), np.where(E < Er[0, 0], 0, Er.shape[1] - 1))
    sd = 1e-24 / (E * E * E) * (cd[0, i0] + (cd[1, i0] + cd[2, i0] * E) * E) * dust_depletion
    sg = 1e-24 / (E * E * E) * (cg[0, i0] + (cg[1, i0] + cg[2, i0] * E) * E)
    if dust_depletion <= 1e-30 or d2h <= 1e-30:
        f = np.ones_like(E)
    else:
        tau = sd / d2h * (3.0 / (2.0 * math.pi)) / (grain_radius_cm * grain_radius_cm)
        f = 1.5 / tau * (1.0 - 2.0 / tau / tau * (1.0 - (tau + 1.0) * np.exp(-tau)))
    out = sg + f * sd
    return out if np.ndim(E_keV) else float(out[0])


def xray_ionization_rate(table, lam_angstrom, local_flux, dust_depletion, d2h, grain_radius_cm, en_per_ion_eV=37.0):
    """calc_Xray_ionization_rate (reference src/disk.f90:1969-2010) for one cell: zeta_Xray_H2 [s^-1 per H] = sum over the wavelength bins
    of the X-ray range of local_flux_i / E_i * sigma(E_i) * (E_i / 37 eV), E_i = h c / lambda_i.  lam_angstrom [n] are the bins the
    caller's loop runs over (i1..i2 of dust_0%lam, its own range selection), local_flux [n] the photon ENERGY flux in each bin
    [erg s^-1 cm^-2] -- the cell's radiation field (c%optical%flux), or with calc_zetaXray_from_Ncol the star's spectrum attenuated by
    exp(-sigma N_toStar) / (4 pi r^2).  Both are inputs from the radiative transfer: zeta_Xray does not depend on the chemistry, so it is a
    per-cell preprocessing of the record (RACGPU_P_ZETA_X), not part of the sweep."""
    h, c, eV = 6.62606896e-27, 2.99792458e10, 1.60217657e-12   # phy_hPlanck_CGS, phy_SpeedOfLight_CGS, phy_eV2erg (src/sub_global_variables.f90)
    lam = np.asarray(lam_angstrom, dtype=np.float64)
    en = h * c / (lam * 1e-8) / eV / 1e3                        # keV
    sig = sigma_xray_bethell(table, en, dust_depletion, d2h, grain_radius_cm)
    z = 0.0
    for e, s, fl in zip(en, sig, np.asarray(local_flux, dtype=np.float64)):   # (the reference's running sum, in bin order)
        z = z + fl / (e * 1e3 * eV) * s * (e * 1e3 / en_per_ion_eV)
    return z


def load_co_shielding_table(path):
    """The reference's own 12CO shielding table (Visser, van Dishoeck & Black 2009) as shipped in data/visser2009_co_shielding.dat
    (written by tools/extract_reference_tables.py from the DATA statements of src/load_Visser_CO_selfshielding.f90): returns the
    (logN_H2 [nrow], logN_12CO [ncol], f [ncol, nrow]) triple co_shielding and Network.set_co_shielding_table take."""
    out = _load_tables(path)
    return out["logN_H2"], out["logN_12CO"], out["f_12CO"]


def andrews_columns(ncol=200, nz=100, rmin=0.1, rmax=200.0, zr_max=0.6):
    """The column structure of andrews_grid as racgpu_column_sweep / racgpu_set_star_rays take it: cells [c * nz + k], k = 0 the
    surface cell.  Returns dict(col_ptr, col_cells, dz, inner, ds, column, layer):
      dz[cell]    vertical extent of the cell [cm]: r_c * zr_max / nz
      inner[cell] the cell a ray from `cell` to the star enters next: the cells of one layer share z / r, so it is the cell of the
                  same layer in the next column inwards (-1 in the innermost column)
      ds[cell]    path length of such a ray through `cell` [cm]: (r_out - r_in) sqrt(1 + (z / r)^2)"""
    r = rmin * (rmax / rmin) ** ((np.arange(ncol) + 0.5) / ncol)
    edges = rmin * (rmax / rmin) ** (np.arange(ncol + 1) / ncol)
    mu = zr_max * (nz - 0.5 - np.arange(nz)) / nz
    column = np.repeat(np.arange(ncol), nz); layer = np.tile(np.arange(nz), ncol)
    dz = np.repeat(r * zr_max / nz * AU_CM, nz)
    ds = (np.diff(edges)[:, None] * np.sqrt(1.0 + mu[None, :] ** 2) * AU_CM).reshape(-1)
    inner = np.where(column > 0, (column - 1) * nz + layer, -1).astype(np.int32)
    return dict(col_ptr=(np.arange(ncol + 1) * nz).astype(np.int32), col_cells=np.arange(ncol * nz, dtype=np.int32), dz=dz, inner=inner,
                ds=np.ascontiguousarray(ds), column=column, layer=layer)


def wavefronts(col_ptr, col_cells, inner=None):
    """Dependency level of every cell of a column sweep: a cell comes after the cell above it in its column and, with star rays,
    after inner[cell].  Level 0 has no predecessor.  (sweep.solve_by_layers with these as `layer` is the host-driven sweep.)"""
    col_ptr = np.asarray(col_ptr); col_cells = np.asarray(col_cells)
    ncell = col_cells.size
    lev = np.zeros(ncell, dtype=np.int64)
    for c in range(col_ptr.size - 1):      # columns in index order: inner[cell] lies in an earlier column
        prev = -1
        for q in range(col_ptr[c], col_ptr[c + 1]):
            cell = col_cells[q]
            l = 0 if prev < 0 else lev[prev] + 1
            if inner is not None and inner[cell] >= 0:
                l = max(l, lev[inner[cell]] + 1)
            lev[cell] = l
            prev = cell
    return lev


def shielding_update(table=None, dv_turb=1e5, species=None, col_ptr=None, col_cells=None, dz=None, inner=None, ds=None):
    """The `update` callback of sweep.solve_by_layers that does on the host what racgpu_column_sweep does on the device (the same
    sums in the same order): toISM slots from the cells above in the column, and with inner/ds the toStar slots from
    N_toStar(cell) = N_toStar(inner) + n_gas(inner) X(inner) ds(inner).  species = dict(H2=, H2O=, OH=, CO=) 0-based indices."""
    col_ptr = np.asarray(col_ptr); col_cells = np.asarray(col_cells)
    ncell = col_cells.size
    above = np.full(ncell, -1, dtype=np.int64)
    for c in range(col_ptr.size - 1):
        q = col_cells[col_ptr[c]:col_ptr[c + 1]]
        above[q[1:]] = q[:-1]
    names = ("H2", "H2O", "OH", "CO")
    N_ism = np.zeros((ncell, 4)); N_star = np.zeros((ncell, 4))   # at the FAR side of the cell (own contribution included) once it is done
    isdone = np.zeros(ncell, dtype=bool)

    def slots(cells_, idx, N, s_H2, s_CO, s_H2O, s_OH):
        cells_[idx, s_H2] = h2_self_shielding(N[:, 0], dv_turb)
        cells_[idx, s_H2O] = lya_self_shielding(N[:, 1], LYA_CROSS_H2O)
        cells_[idx, s_OH] = lya_self_shielding(N[:, 2], LYA_CROSS_OH)
        if table is not None:
            cells_[idx, s_CO] = co_shielding(table, N[:, 0], N[:, 3])

    def update(k, idx, cells_, y_done, done):
        new = done[~isdone[done]] if len(done) else done
        for cell in new:   # what the cells solved since the last call add, in dependency order (a level never feeds itself)
            w = cells_[cell, P_NGAS]
            base_i = N_ism[above[cell]] if above[cell] >= 0 else np.zeros(4)
            N_ism[cell] = [base_i[j] + (w * dz[cell]) * y_done[cell, species[nm]] if species.get(nm, -1) >= 0 else base_i[j] for j, nm in enumerate(names)]
            if inner is not None:
                base_s = N_in_star[cell]
                N_star[cell] = [base_s[j] + (w * ds[cell]) * y_done[cell, species[nm]] if species.get(nm, -1) >= 0 else base_s[j] for j, nm in enumerate(names)]
            isdone[cell] = True
        Ni = np.array([N_ism[above[c]] if above[c] >= 0 else np.zeros(4) for c in idx])
        slots(cells_, idx, Ni, P_FSS_ISM_H2, P_FSS_ISM_CO, P_FSS_ISM_H2O, P_FSS_ISM_OH)
        if inner is not None:
            Ns = np.array([N_star[inner[c]] if inner[c] >= 0 else np.zeros(4) for c in idx])
            N_in_star[idx] = Ns
            slots(cells_, idx, Ns, P_FSS_STAR_H2, P_FSS_STAR_CO, P_FSS_STAR_H2O, P_FSS_STAR_OH)

    N_in_star = np.zeros((ncell, 4))
    return update


def column_density_above(n_species, dz, column, layer):
    """Column density [cm^-2] from the TOP of every cell of a regular column grid to the surface: the sum of n dz over the cells
    of the same column in the layers above it (layer 0 = the top; the cell itself is not counted, as with the reference's
    fromCellCenter = .false., src/disk.f90:2532-2537).  n_species [ncell] = n_gas * abundance."""
    n_species = np.asarray(n_species, dtype=np.float64)
    out = np.zeros_like(n_species)
    order = np.lexsort((np.asarray(layer), np.asarray(column)))
    col_sorted = np.asarray(column)[order]
    contrib = (n_species * np.asarray(dz, dtype=np.float64))[order]
    csum = np.cumsum(contrib) - contrib  # exclusive
    first = np.r_[True, col_sorted[1:] != col_sorted[:-1]]
    base = np.maximum.accumulate(np.where(first, np.arange(order.size), 0))
    out[order] = csum - csum[base]
    return out
