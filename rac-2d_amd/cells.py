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
