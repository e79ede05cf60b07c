"""rac-2d_amd -- host-side Python mirror of the reference's chemistry interface over the racgpu C ABI.

Everything numerical happens in ``libracgpu.so`` (hand-written gfx950 kernels, see csrc/).  This module
is plumbing: ctypes signatures for every symbol of ``include/racgpu.h`` plus thin wrappers whose names
follow the reference's own vocabulary (``chem_cal_rates`` -> :meth:`Network.cal_rates`,
``chem_evol_solve`` over the cell sweep -> :meth:`Network.evol_solve_batch`, ``chemsol_params`` ->
:class:`ChemsolParams`).  There is no CPU fallback: without the built library, or without a GPU for the
compute calls, the calls raise.

The package directory name contains a hyphen (``rac-2d_amd``), so import it with
``importlib.import_module("rac-2d_amd")``.
"""
import ctypes as C
import os

import numpy as np

from . import cells  # noqa: F401  (synthetic cell records)
from . import analysis  # noqa: F401  (rate dump, contributions, elemental reservoirs)
from . import sweep  # noqa: F401  (sharding over ranks, layer-by-layer sweeps)
from .cells import NPAR

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RACGPU_LIB") or os.path.join(_HERE, "libracgpu.so")  # RACGPU_LIB: developer builds
NSTAT = 20
NOUT = 6
NHC = 28
NHCTERMS = 29
MEM_HOST, MEM_DEVICE = 0, 1
F_RECTIFY = 1
(S_NST, S_NFE, S_NJE, S_NLU, S_NERR, S_NREC_REAL, S_QSUM, S_NCFAIL_ETFAIL, S_CYC_TOTAL, S_CYC_RHS, S_CYC_JAC, S_CYC_LU,
 S_CYC_SOLVE, S_CYC_LU_SCATTER, S_CYC_LU_LDS, S_CYC_LU_REG, S_ISAV, S_NITER, S_NREC, S_ERRCODES) = range(NSTAT)
O_R_H2_FORM, O_N_MOL_ON_GRAIN, O_T_END, O_TGAS, O_EVOLT_END, O_TFREEZE_REC = range(NOUT)
# the 29 values of type_heating_cooling_rates_list, in its order (reference src/data_struct.f90:489-520)
HC_TERM_NAMES = ("hc_net_rate", "heating_photoelectric_small_grain", "heating_formation_H2", "heating_cosmic_ray", "heating_vibrational_H2",
                 "heating_ionization_CI", "heating_photodissociation_H2", "heating_photodissociation_H2O", "heating_photodissociation_OH",
                 "heating_Xray_Bethell", "heating_viscosity", "heating_chem", "cooling_photoelectric_small_grain", "cooling_vibrational_H2",
                 "cooling_gas_grain_collision", "cooling_OI", "cooling_CII", "cooling_Neufeld_H2O_rot", "cooling_Neufeld_H2O_vib",
                 "cooling_Neufeld_CO_rot", "cooling_Neufeld_CO_vib", "cooling_Neufeld_H2_rot", "cooling_LymanAlpha", "cooling_free_bound",
                 "cooling_free_free", "cooling_NII", "cooling_SiII", "cooling_FeII", "cooling_OH_rot")

# every extern "C" symbol include/racgpu.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "racgpu_last_error", "racgpu_device_count", "racgpu_network_load", "racgpu_network_destroy",
    "racgpu_network_dims", "racgpu_network_set_reference_lenrw", "racgpu_network_reference_lenrw", "racgpu_species_name", "racgpu_species_index", "racgpu_reactions",
    "racgpu_lu_ordering", "racgpu_species_attrs", "racgpu_species_elements", "racgpu_reaction_rows", "racgpu_jac_pattern", "racgpu_load_initial_abundances", "racgpu_params_default",
    "racgpu_n_record", "racgpu_set_tolerances", "racgpu_init_abundances", "racgpu_set_device",
    "racgpu_set_stream", "racgpu_rates", "racgpu_rhs", "racgpu_jac_csc", "racgpu_newton_solve",
    "racgpu_solve_batch", "racgpu_evol_solve_batch", "racgpu_calc_cells", "racgpu_column_sweep", "racgpu_set_co_shielding_table", "racgpu_set_star_rays", "racgpu_star_ray_timeouts", "racgpu_rectify_abundances",
    "racgpu_hc_config_default", "racgpu_heating_cooling_load", "racgpu_heat_reactions", "racgpu_evolT_hooks", "racgpu_evolT_solve_batch",
    "racgpu_multi_create", "racgpu_multi_destroy", "racgpu_multi_ndev", "racgpu_multi_network", "racgpu_multi_last_error", "racgpu_multi_deal",
    "racgpu_multi_calc_cells",
    "racgpu_set_cost_hints", "racgpu_set_team_threshold", "racgpu_last_team_cells", "racgpu_last_parked_cells", "racgpu_workspace_bytes_per_cell", "racgpu_last_kernel_ms",
]


class ChemsolParams(C.Structure):
    """``racgpu_params``: the scalars of the reference's ``chemsol_params`` namelist variable
    (reference src/chemistry.f90:107-135) that the path reads."""
    _fields_ = [
        ("RTOL", C.c_double), ("ATOL", C.c_double), ("t_max", C.c_double), ("dt_first_step", C.c_double),
        ("ratio_tstep", C.c_double), ("max_runtime_allowed", C.c_double), ("Diff2DesorRatio", C.c_double),
        ("special_gH_E_diff", C.c_double),
        ("mxstep_per_interval", C.c_int32), ("steps_reset_solver", C.c_int32), ("H2_form_use_moeq", C.c_int32),
        ("evol_dust_size", C.c_int32), ("use_special_gH_mobi", C.c_int32), ("tol_policy_j", C.c_int32),
        ("max_steps_per_cell", C.c_int64),
        ("rt_cost_f", C.c_double), ("rt_cost_jac", C.c_double), ("rt_cost_lu", C.c_double),
    ]


class HcConfig(C.Structure):
    """``racgpu_hc_config``: heating_cooling_config (reference src/heating_cooling.f90:16-38) as far as the implemented branches read it,
    a_disk%base_alpha and chemsol_params%maySwitchT."""
    _fields_ = [("heating_eff_chem", C.c_double), ("heating_eff_H2form", C.c_double), ("heating_eff_phd_H2", C.c_double),
                ("heating_eff_phd_H2O", C.c_double), ("heating_eff_phd_OH", C.c_double), ("cooling_gg_coeff", C.c_double),
                ("base_alpha", C.c_double),
                ("use_chemicalheatingcooling", C.c_int32), ("use_Xray_heating", C.c_int32), ("use_phdheating_H2", C.c_int32),
                ("use_phdheating_H2OOH", C.c_int32), ("use_mygasgraincooling", C.c_int32), ("may_switch_T", C.c_int32)]


_lib = None


def lib():
    """Load libracgpu.so (built in-tree by ``__graft_entry__.build()`` / ``make -C rac-2d_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the racgpu path)" % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own libamdhip64; a process must end up with ONE HIP runtime.  If torch is
    # installed, load it first so that libracgpu.so binds to the runtime torch brought (loading them in the other
    # order leaves torch unable to see the GPU).  RACGPU_NO_TORCH=1 skips this (plain ROCm runtime).
    if not os.environ.get("RACGPU_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, dp, ip, lp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    pp = C.POINTER(ChemsolParams)
    L.racgpu_last_error.restype = C.c_char_p
    L.racgpu_network_load.restype = vp
    L.racgpu_network_load.argtypes = [C.c_char_p]
    L.racgpu_network_destroy.argtypes = [vp]
    L.racgpu_network_dims.argtypes = [vp, ip, ip, ip, ip, ip]
    L.racgpu_species_name.argtypes = [vp, C.c_int32, C.c_char_p, C.c_int32]
    L.racgpu_network_set_reference_lenrw.argtypes = [vp, C.c_int32]
    L.racgpu_network_reference_lenrw.argtypes = [vp]
    L.racgpu_species_index.argtypes = [vp, C.c_char_p]
    L.racgpu_reactions.argtypes = [vp, ip, ip, ip, ip, ip, ip]
    L.racgpu_species_attrs.argtypes = [vp, dp, dp, dp, ip, ip]
    L.racgpu_lu_ordering.argtypes = [vp, ip, ip, ip]
    L.racgpu_jac_pattern.argtypes = [vp, ip, ip]
    L.racgpu_species_elements.argtypes = [vp, ip]
    L.racgpu_reaction_rows.argtypes = [vp, dp, dp, C.c_char_p, C.c_char_p, C.c_char_p]
    L.racgpu_load_initial_abundances.argtypes = [vp, C.c_char_p, dp]
    L.racgpu_params_default.argtypes = [pp]
    L.racgpu_n_record.argtypes = [pp, C.c_double, C.c_double]
    L.racgpu_set_tolerances.argtypes = [vp, pp, C.c_int32, C.c_double, dp, dp]
    L.racgpu_init_abundances.argtypes = [vp, dp, dp, C.c_int64, dp]
    L.racgpu_set_device.argtypes = [C.c_int]
    L.racgpu_set_stream.argtypes = [vp, vp]
    L.racgpu_rates.argtypes = [vp, pp, dp, C.c_int64, dp]
    L.racgpu_rhs.argtypes = [vp, pp, dp, C.c_int64, dp, dp]
    L.racgpu_jac_csc.argtypes = [vp, pp, dp, C.c_int64, dp, dp]
    L.racgpu_newton_solve.argtypes = [vp, pp, dp, C.c_int64, dp, C.c_double, dp]
    L.racgpu_solve_batch.argtypes = [vp, pp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.racgpu_evol_solve_batch.argtypes = [vp, pp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int]
    L.racgpu_calc_cells.argtypes = [vp, pp, C.c_int32, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int]
    L.racgpu_rectify_abundances.argtypes = [vp, C.c_int64, dp]
    L.racgpu_workspace_bytes_per_cell.restype = C.c_int64
    L.racgpu_workspace_bytes_per_cell.argtypes = [vp]
    L.racgpu_set_cost_hints.restype = C.c_int
    L.racgpu_set_cost_hints.argtypes = [vp, dp, C.c_int64]
    L.racgpu_last_kernel_ms.restype = C.c_double
    L.racgpu_last_kernel_ms.argtypes = [vp]
    L.racgpu_column_sweep.restype = C.c_int
    L.racgpu_column_sweep.argtypes = [vp, C.POINTER(ChemsolParams), C.c_int64, vp, vp, C.c_int64, vp, vp, vp, C.c_double, vp, vp, vp, vp, C.c_int]
    L.racgpu_set_star_rays.restype = C.c_int
    L.racgpu_set_star_rays.argtypes = [vp, C.c_int64, vp, vp]
    L.racgpu_star_ray_timeouts.restype = C.c_int
    L.racgpu_star_ray_timeouts.argtypes = [vp]
    L.racgpu_set_co_shielding_table.restype = C.c_int
    L.racgpu_set_co_shielding_table.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp]
    L.racgpu_hc_config_default.argtypes = [C.POINTER(HcConfig)]
    L.racgpu_hc_config_default.restype = None
    L.racgpu_heating_cooling_load.argtypes = [vp, C.POINTER(HcConfig), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
    L.racgpu_heat_reactions.argtypes = [vp, ip, ip, dp]
    L.racgpu_evolT_hooks.argtypes = [vp, pp, dp, dp, C.c_int64, dp, dp, dp, vp, vp]
    L.racgpu_evolT_solve_batch.argtypes = [vp, pp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int]
    L.racgpu_multi_create.restype = vp
    L.racgpu_multi_create.argtypes = [C.c_char_p, C.c_int, ip]
    L.racgpu_multi_destroy.argtypes = [vp]
    L.racgpu_multi_ndev.argtypes = [vp]
    L.racgpu_multi_network.restype = vp
    L.racgpu_multi_network.argtypes = [vp, C.c_int]
    L.racgpu_multi_last_error.restype = C.c_char_p
    L.racgpu_multi_deal.argtypes = [C.c_int, C.c_int64, dp, ip, ip]
    L.racgpu_multi_calc_cells.argtypes = [vp, pp, C.c_int32, C.c_int64, dp, dp, dp, ip, lp, dp, dp]
    L.racgpu_set_team_threshold.restype = C.c_int
    L.racgpu_set_team_threshold.argtypes = [vp, C.c_double]
    L.racgpu_last_team_cells.restype = C.c_int64
    L.racgpu_last_team_cells.argtypes = [vp]
    L.racgpu_last_parked_cells.restype = C.c_int64
    L.racgpu_last_parked_cells.argtypes = [vp]
    _lib = L
    return L


class RacgpuError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise RacgpuError(lib().racgpu_last_error().decode())


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def default_params():
    p = ChemsolParams()
    lib().racgpu_params_default(C.byref(p))
    return p


def default_hc_config():
    c = HcConfig()
    lib().racgpu_hc_config_default(C.byref(c))
    return c


def multi_deal(ndev, ncell, cost=None):
    """racgpu_multi_deal: (owner [ncell], position [ncell]) of the single-process multi-GPU entry point's dealing rule (host only)."""
    owner = np.zeros(ncell, np.int32); pos = np.zeros(ncell, np.int32)
    c = None if cost is None else np.ascontiguousarray(cost, np.float64)
    _check(lib().racgpu_multi_deal(ndev, ncell, None if c is None else _dp(c), _ip(owner), _ip(pos)))
    return owner, pos


class MultiGPU:
    """racgpu_multi: one process, ndev GPUs of one node, one RCCL all-gather of the results (include/racgpu.h)."""

    def __init__(self, path, ndev, devices=None):
        d = None if devices is None else np.ascontiguousarray(devices, np.int32)
        self._m = lib().racgpu_multi_create(os.fsencode(path), ndev, None if d is None else _ip(d))
        if not self._m:
            raise RacgpuError(lib().racgpu_multi_last_error().decode())
        self.ndev = lib().racgpu_multi_ndev(self._m)

    def close(self):
        if getattr(self, "_m", None):
            lib().racgpu_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def calc_cells(self, params, cell_records, y, nlocal_iter=1, cost=None):
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        n = cr.shape[0]
        nS = y.shape[-1]
        y = np.array(y, np.float64).reshape(n, nS).copy()
        tf = np.zeros(n); q = np.zeros(n, np.int32); st = np.zeros((n, NSTAT), np.int64); co = np.zeros((n, NOUT))
        c = None if cost is None else np.ascontiguousarray(cost, np.float64)
        rc = lib().racgpu_multi_calc_cells(self._m, C.byref(params), nlocal_iter, n, _dp(cr), _dp(y), _dp(tf), _ip(q),
                                           st.ctypes.data_as(C.POINTER(C.c_int64)), _dp(co), None if c is None else _dp(c))
        if rc != 0:
            raise RacgpuError(lib().racgpu_multi_last_error().decode())
        return dict(y=y, t_final=tf, quality=q, stats=st, cell_out=co)


def device_count():
    return lib().racgpu_device_count()


def set_device(dev):
    _check(lib().racgpu_set_device(dev))


class Network:
    """A parsed reaction network + everything cell-independent (``chem_net``, ``chem_species``,
    sparsity, symbolic LU).  Mirrors the reference's one-time setup sequence src/disk.f90:1566-1581."""

    def __init__(self, path):
        self._h = lib().racgpu_network_load(os.fsencode(path))
        if not self._h:
            raise RacgpuError(lib().racgpu_last_error().decode())
        d = [C.c_int32() for _ in range(5)]
        _check(lib().racgpu_network_dims(self._h, *[C.byref(x) for x in d]))
        self.nSpecies, self.nReactions, self.nnzJ, self.nzl, self.nzu = [x.value for x in d]
        buf = C.create_string_buffer(32)
        self.names = []
        for i in range(1, self.nSpecies + 1):
            _check(lib().racgpu_species_name(self._h, i, buf, 32))
            self.names.append(buf.value.decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().racgpu_network_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_reference_lenrw(self, lenrw):
        """IWORK(17) of the reference's DLSODES for this network (0: unknown, the saved P survives ISTATE = 3); see racgpu.h."""
        _check(lib().racgpu_network_set_reference_lenrw(self._h, int(lenrw)))

    def reference_lenrw(self):
        return lib().racgpu_network_reference_lenrw(self._h)

    # ---- host-only queries -------------------------------------------------------------------------
    def species_index(self, name):
        return lib().racgpu_species_index(self._h, name.encode())

    def reactions(self):
        nR = self.nReactions
        reac = np.zeros((nR, 3), np.int32); prod = np.zeros((nR, 4), np.int32)
        nre = np.zeros(nR, np.int32); npr = np.zeros(nR, np.int32); it = np.zeros(nR, np.int32); nd = np.zeros(nR, np.int32)
        _check(lib().racgpu_reactions(self._h, _ip(reac), _ip(prod), _ip(nre), _ip(npr), _ip(it), _ip(nd)))
        return dict(reac=reac, prod=prod, n_reac=nre, n_prod=npr, itype=it, n_dupli=nd)

    def species_attrs(self):
        nS = self.nSpecies
        m = np.zeros(nS); v = np.zeros(nS); e = np.zeros(nS); cp = np.zeros(nS, np.int32); ch = np.zeros(nS, np.int32)
        _check(lib().racgpu_species_attrs(self._h, _dp(m), _dp(v), _dp(e), _ip(cp), _ip(ch)))
        return dict(mass_num=m, vib_freq=v, Edesorb=e, counterpart=cp, charge=ch)

    def species_elements(self):
        """chem_species%elements: [nS, 20] (column 0 = charge)."""
        el = np.zeros((self.nSpecies, 20), np.int32)
        _check(lib().racgpu_species_elements(self._h, _ip(el)))
        return el

    def reaction_rows(self):
        """ABC, T_range, ctype, reliability and the seven name fields of every reaction row as read from the network file."""
        nR = self.nReactions
        abc = np.zeros((nR, 3)); tr = np.zeros((nR, 2))
        ct = C.create_string_buffer(2 * nR); rl = C.create_string_buffer(nR); nm = C.create_string_buffer(nR * 84)
        _check(lib().racgpu_reaction_rows(self._h, _dp(abc), _dp(tr), ct, rl, nm))
        names = [[nm.raw[(r * 7 + k) * 12:(r * 7 + k + 1) * 12].decode() for k in range(7)] for r in range(nR)]
        return dict(ABC=abc, T_range=tr, ctype=[ct.raw[2 * r:2 * r + 2].decode() for r in range(nR)],
                    reliability=[rl.raw[r:r + 1].decode() for r in range(nR)], names=names)

    def jac_pattern(self):
        colptr = np.zeros(self.nSpecies + 1, np.int32); rowidx = np.zeros(self.nnzJ, np.int32)
        _check(lib().racgpu_jac_pattern(self._h, _ip(colptr), _ip(rowidx)))
        return colptr, rowidx

    def lu_ordering(self):
        """(perm [nS] 1-based: perm[new] = old, first position (1-based) of the dense trailing block)"""
        perm = np.zeros(self.nSpecies, np.int32); fd = C.c_int32()
        _check(lib().racgpu_lu_ordering(self._h, _ip(perm), C.byref(fd), None))
        return perm, fd.value

    def p_storage(self):
        """entry of jac_pattern() (1-based) held at each position of the engine's storage of the Newton matrix"""
        ps = np.zeros(self.nnzJ, np.int32)
        _check(lib().racgpu_lu_ordering(self._h, None, None, _ip(ps)))
        return ps

    def load_initial_abundances(self, path):
        y0 = np.zeros(self.nSpecies)
        _check(lib().racgpu_load_initial_abundances(self._h, os.fsencode(path), _dp(y0)))
        return y0

    def set_solver_flags_alt(self, params, j, d2h):
        rtol = np.zeros(self.nSpecies + 1); atol = np.zeros(self.nSpecies + 1)
        _check(lib().racgpu_set_tolerances(self._h, C.byref(params), j, d2h, _dp(rtol), _dp(atol)))
        return rtol, atol

    def init_abundances(self, y0, cell_records):
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        y = np.zeros((cr.shape[0], self.nSpecies))
        _check(lib().racgpu_init_abundances(self._h, _dp(np.ascontiguousarray(y0, np.float64)), _dp(cr), cr.shape[0], _dp(y)))
        return y

    def workspace_bytes_per_cell(self):
        return lib().racgpu_workspace_bytes_per_cell(self._h)

    # ---- GPU compute --------------------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        _check(lib().racgpu_set_stream(self._h, C.c_void_p(stream_ptr)))

    def cal_rates(self, params, cell_records):
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        out = np.zeros((cr.shape[0], self.nReactions))
        _check(lib().racgpu_rates(self._h, C.byref(params), _dp(cr), cr.shape[0], _dp(out)))
        return out

    def ode_f(self, params, cell_records, y):
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        y = np.ascontiguousarray(y, np.float64).reshape(cr.shape[0], self.nSpecies)
        out = np.zeros_like(y)
        _check(lib().racgpu_rhs(self._h, C.byref(params), _dp(cr), cr.shape[0], _dp(y), _dp(out)))
        return out

    def ode_jac(self, params, cell_records, y):
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        y = np.ascontiguousarray(y, np.float64).reshape(cr.shape[0], self.nSpecies)
        out = np.zeros((cr.shape[0], self.nnzJ))
        _check(lib().racgpu_jac_csc(self._h, C.byref(params), _dp(cr), cr.shape[0], _dp(y), _dp(out)))
        return out

    def newton_solve(self, params, cell_records, y, gamma, b):
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        y = np.ascontiguousarray(y, np.float64).reshape(cr.shape[0], self.nSpecies)
        x = np.array(b, np.float64).reshape(cr.shape[0], self.nSpecies).copy()
        _check(lib().racgpu_newton_solve(self._h, C.byref(params), _dp(cr), cr.shape[0], _dp(y), gamma, _dp(x)))
        return x

    def evol_solve_batch(self, params, cell_records, y, record=False, t0=None, tol_j=None, rectify=False):
        """chem_cal_rates + chem_set_solver_flags_alt + chem_evol_solve for every cell (host arrays).
        t0 [ncell]: chemsol_params%t0 per cell (continue runs: first step max(dt0, 1e-3 t0)); tol_j [ncell]: policy j per cell;
        rectify: apply rectify_abundances first (set_initial_condition_4solver_continue).  y/t_final out follow the
        hand-off rule (last record without NaN), stats[:, S_ISAV] tells which record; cell_out = RACGPU_O_* values."""
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        n = cr.shape[0]
        y = np.array(y, np.float64).reshape(n, self.nSpecies).copy()
        tf = np.zeros(n); q = np.zeros(n, np.int32); st = np.zeros((n, NSTAT), np.int64); co = np.full((n, NOUT), np.nan)
        t0a = None if t0 is None else np.ascontiguousarray(np.broadcast_to(np.asarray(t0, np.float64), (n,)))
        tja = None if tol_j is None else np.ascontiguousarray(np.broadcast_to(np.asarray(tol_j, np.int32), (n,)))
        rec = tos = None
        if record:
            nrec = lib().racgpu_n_record(C.byref(params), 0.0, params.t_max)
            rec = np.zeros((n, nrec, self.nSpecies + 1)); tos = np.zeros((n, nrec))
        _check(lib().racgpu_evol_solve_batch(self._h, C.byref(params), n, cr.ctypes.data, y.ctypes.data,
                                             None if t0a is None else t0a.ctypes.data, None if tja is None else tja.ctypes.data,
                                             tf.ctypes.data, q.ctypes.data, st.ctypes.data, rec.ctypes.data if record else None,
                                             tos.ctypes.data if record else None, co.ctypes.data, F_RECTIFY if rectify else 0, MEM_HOST))
        return dict(y=y, t_final=tf, quality=q, stats=st, record=rec, touts=tos, cell_out=co,
                    kernel_ms=lib().racgpu_last_kernel_ms(self._h))

    # ---- gas temperature co-evolving with the chemistry (chemsol_params%evolT) -------------------------------------------------
    def load_heating_cooling(self, data_dir, config=None):
        """heating_cooling_prepare + the reaction heats: reads Species_enthalpy.dat, neufeld_cooling_tables.dat and the three ion
        look-up tables from data_dir (data/README.md).  config: an HcConfig (default: the reference's template values)."""
        cfg = config or default_hc_config()
        j = lambda f: os.fsencode(os.path.join(data_dir, f))
        _check(lib().racgpu_heating_cooling_load(self._h, C.byref(cfg), j("Species_enthalpy.dat"), j("neufeld_cooling_tables.dat"),
                                                 j("N+_LUT.bin"), j("Si+_LUT.bin"), j("Fe+_LUT.bin")))

    def heat_reactions(self):
        """chem_net%iReacWithHeat (1-based) and %heat [erg] (chem_get_reaction_heat)."""
        n = C.c_int32()
        _check(lib().racgpu_heat_reactions(self._h, C.byref(n), None, None))
        rx = np.zeros(n.value, np.int32); ht = np.zeros(n.value)
        _check(lib().racgpu_heat_reactions(self._h, C.byref(n), _ip(rx), _dp(ht)))
        return rx, ht

    def ode_f_evolT(self, params, cell_records, hc_records, y, jac_border=False, full_rows=False):
        """chem_ode_f with T evolving at y [ncell, nS+1] (last entry Tgas): dict(ydot [ncell, nS+1], terms [ncell, 29]); with
        jac_border also chem_ode_jac's finite-difference T column [ncell, nS+1] and T row [ncell, 10].  full_rows (developer switch,
        RACGPU_DEBUG_FULL_TROW): every T-row entry by a full evaluation of the 28 terms instead of the blocks its species enters."""
        if full_rows:
            os.environ["RACGPU_DEBUG_FULL_TROW"] = "1"
        else:
            os.environ.pop("RACGPU_DEBUG_FULL_TROW", None)
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        n = cr.shape[0]
        hr = np.ascontiguousarray(hc_records, np.float64).reshape(n, NHC)
        y = np.ascontiguousarray(y, np.float64).reshape(n, self.nSpecies + 1)
        yd = np.zeros_like(y); tm = np.zeros((n, NHCTERMS))
        tc = np.zeros_like(y) if jac_border else None
        tr = np.zeros((n, 10)) if jac_border else None
        _check(lib().racgpu_evolT_hooks(self._h, C.byref(params), _dp(cr), _dp(hr), n, _dp(y), _dp(yd), _dp(tm),
                                        tc.ctypes.data if jac_border else None, tr.ctypes.data if jac_border else None))
        return dict(ydot=yd, terms=tm, tcol=tc, trow=tr)

    def evolT_solve_batch(self, params, cell_records, hc_records, y, record=False, t0=None, tol_j=None, rectify=False):
        """evol_solve_batch with the gas temperature co-evolving in every cell whose hc record has en_gain_tot > 0; the cell record's
        Tgas is the initial temperature, cell_out[:, O_TGAS] the one handed back, record[..., nS] T(t)."""
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        n = cr.shape[0]
        hr = np.ascontiguousarray(hc_records, np.float64).reshape(n, NHC)
        y = np.array(y, np.float64).reshape(n, self.nSpecies).copy()
        tf = np.zeros(n); q = np.zeros(n, np.int32); st = np.zeros((n, NSTAT), np.int64); co = np.full((n, NOUT), np.nan)
        t0a = None if t0 is None else np.ascontiguousarray(np.broadcast_to(np.asarray(t0, np.float64), (n,)))
        tja = None if tol_j is None else np.ascontiguousarray(np.broadcast_to(np.asarray(tol_j, np.int32), (n,)))
        rec = tos = None
        if record:
            nrec = lib().racgpu_n_record(C.byref(params), 0.0, params.t_max)
            rec = np.zeros((n, nrec, self.nSpecies + 1)); tos = np.zeros((n, nrec))
        _check(lib().racgpu_evolT_solve_batch(self._h, C.byref(params), n, cr.ctypes.data, hr.ctypes.data, y.ctypes.data,
                                              None if t0a is None else t0a.ctypes.data, None if tja is None else tja.ctypes.data,
                                              tf.ctypes.data, q.ctypes.data, st.ctypes.data, rec.ctypes.data if record else None,
                                              tos.ctypes.data if record else None, co.ctypes.data, F_RECTIFY if rectify else 0, MEM_HOST))
        return dict(y=y, t_final=tf, quality=q, stats=st, record=rec, touts=tos, cell_out=co, kernel_ms=lib().racgpu_last_kernel_ms(self._h))

    def evolT_calc_cells(self, params, cell_records, hc_records, y, nlocal_iter=4):
        """calc_this_cell's local-iteration loop (reference src/disk.f90:1651-1791) with the gas temperature co-evolving, driven from the host
        on top of racgpu_evolT_solve_batch (racgpu_calc_cells is the fixed-T loop on the device side).  Iteration j = 2.. takes the cells
        that ended flagged before half of their t_max and got past the previous iteration: abundances and Tgas of the hand-off record
        (set_initial_condition_4solver_continue, :2100-2146), rectify_abundances, t0 = t_final, tolerance policy j; a cell whose T moved by
        less than 1 % over the last iteration, with t_final >= 0.2 t_max and net heating below 1 % of its largest term
        (heating_cooling_is_very_slow, src/heating_cooling.f90:1492-1498), goes on at fixed T (:1662-1667).  The reference applies that
        last test to the terms of whichever heating_minus_cooling call came last; here they are evaluated at the hand-off state.
        Returns dict(y, t_final, quality, tgas, niter [ncell], stats (summed work counters))."""
        from . import cells as Cc
        cr = np.array(cell_records, np.float64).reshape(-1, NPAR).copy()
        n = cr.shape[0]
        hr = np.array(hc_records, np.float64).reshape(n, NHC).copy()
        y = np.array(y, np.float64).reshape(n, self.nSpecies).copy()
        tmax = np.where(cr[:, Cc.P_TMAX] > 0.0, cr[:, Cc.P_TMAX], params.t_max)
        tf = np.zeros(n); q = np.zeros(n, np.int32); niter = np.zeros(n, np.int32); st = np.zeros((n, NSTAT), np.int64)
        act = np.arange(n)
        for j in range(1, nlocal_iter + 1):
            if act.size == 0:
                break
            T0 = cr[act, Cc.P_TGAS].copy()
            o = self.evolT_solve_batch(params, cr[act], hr[act], y[act], t0=None if j == 1 else tf[act], tol_j=j, rectify=j > 1)
            t_end = o["cell_out"][:, O_T_END]
            proceeds = (j == 1) | (t_end > tf[act])               # "Local iteration does not proceed": nothing is taken over
            useful = proceeds & (o["stats"][:, S_ISAV] > 1)
            idx = act[useful]
            y[idx] = o["y"][useful]; tf[idx] = o["t_final"][useful]; cr[idx, Cc.P_TGAS] = o["cell_out"][useful, O_TGAS]
            q[act[proceeds]] = o["quality"][proceeds]; niter[act[proceeds]] = j
            st[act, :8] += o["stats"][:, :8]
            go = useful & (o["quality"] != 0) & (o["t_final"] < 0.5 * tmax[act])
            nxt = act[go]
            if nxt.size and j < nlocal_iter:                      # who goes on at fixed T (src/disk.f90:1662-1667)
                yT = np.hstack([y[nxt], cr[nxt, Cc.P_TGAS][:, None]])
                terms = self.ode_f_evolT(params, cr[nxt], hr[nxt], yT)["terms"]
                slow = terms[:, 0] < 1e-2 * np.maximum(np.max(np.c_[terms[:, 1:12], -terms[:, 12]], axis=1), np.max(np.c_[terms[:, 12:29], -terms[:, 11]], axis=1))
                same_T = np.abs(T0[go] - cr[nxt, Cc.P_TGAS]) <= 1e-2 * T0[go]
                off = same_T & (tf[nxt] >= 0.2 * tmax[nxt]) & slow
                hr[nxt[off], Cc.H_EN_GAIN_TOT] = 0.0
            act = nxt
        return dict(y=y, t_final=tf, quality=q, tgas=cr[:, Cc.P_TGAS].copy(), niter=niter, stats=st)

    def evolT_solve_batch_device(self, params, ncell, cells_ptr, hc_ptr, y_ptr, t_final_ptr=None, quality_ptr=None, stats_ptr=None, cell_out_ptr=None):
        _check(lib().racgpu_evolT_solve_batch(self._h, C.byref(params), ncell, cells_ptr, hc_ptr, y_ptr, None, None, t_final_ptr, quality_ptr,
                                              stats_ptr, None, None, cell_out_ptr, 0, MEM_DEVICE))

    def calc_cells(self, params, cell_records, y, nlocal_iter=4):
        """calc_this_cell's chemistry for every cell (reference src/disk.f90:1651-1791): up to nlocal_iter local iterations,
        each iteration one batched launch over the cells that still need it (host arrays)."""
        cr = np.ascontiguousarray(cell_records, np.float64).reshape(-1, NPAR)
        n = cr.shape[0]
        y = np.array(y, np.float64).reshape(n, self.nSpecies).copy()
        tf = np.zeros(n); q = np.zeros(n, np.int32); st = np.zeros((n, NSTAT), np.int64); co = np.full((n, NOUT), np.nan)
        _check(lib().racgpu_calc_cells(self._h, C.byref(params), nlocal_iter, n, cr.ctypes.data, y.ctypes.data, tf.ctypes.data,
                                       q.ctypes.data, st.ctypes.data, co.ctypes.data, MEM_HOST))
        return dict(y=y, t_final=tf, quality=q, stats=st, cell_out=co, kernel_ms=lib().racgpu_last_kernel_ms(self._h))

    def rectify_abundances(self, y):
        y = np.array(y, np.float64).reshape(-1, self.nSpecies).copy()
        _check(lib().racgpu_rectify_abundances(self._h, y.shape[0], _dp(y)))
        return y

    def evol_solve_batch_device(self, params, ncell, cells_ptr, y_ptr, t_final_ptr=None, quality_ptr=None, stats_ptr=None,
                                t0_ptr=None, tol_j_ptr=None, cell_out_ptr=None, flags=0):
        """Same, on device pointers (e.g. torch tensors' data_ptr()); asynchronous on the handle's stream."""
        _check(lib().racgpu_evol_solve_batch(self._h, C.byref(params), ncell, cells_ptr, y_ptr, t0_ptr, tol_j_ptr, t_final_ptr,
                                             quality_ptr, stats_ptr, None, None, cell_out_ptr, flags, MEM_DEVICE))

    def calc_cells_device(self, params, nlocal_iter, ncell, cells_ptr, y_ptr, t_final_ptr=None, quality_ptr=None, stats_ptr=None,
                          cell_out_ptr=None):
        """racgpu_calc_cells on device pointers (synchronises the handle's stream between local iterations)."""
        _check(lib().racgpu_calc_cells(self._h, C.byref(params), nlocal_iter, ncell, cells_ptr, y_ptr, t_final_ptr, quality_ptr,
                                       stats_ptr, cell_out_ptr, MEM_DEVICE))

    def set_co_shielding_table(self, table=None):
        """12CO shielding table for column_sweep: (logN_H2 [nrow], logN_12CO [ncol], f [ncol, nrow]) as cells.co_shielding takes
        it; None clears it (the CO slot of the records then stays as given)."""
        if table is None:
            _check(lib().racgpu_set_co_shielding_table(self._h, 0, 0, None, None, None))
            return
        lh, lc, f = (np.ascontiguousarray(a, dtype=np.float64) for a in table)
        if f.shape != (lc.size, lh.size):
            raise ValueError("f must be [ncol, nrow]")
        _check(lib().racgpu_set_co_shielding_table(self._h, lh.size, lc.size, lh.ctypes.data, lc.ctypes.data, f.ctypes.data))

    def set_star_rays(self, inner=None, ds=None):
        """racgpu_set_star_rays: inner[cell] = the cell a ray from `cell` to the star enters next (-1: none), ds[cell] = path length
        of such a ray through `cell` [cm]; column_sweep then rewrites the toStar shielding slots as well.  None clears."""
        if inner is None:
            _check(lib().racgpu_set_star_rays(self._h, 0, None, None))
            return
        inn = np.ascontiguousarray(inner, dtype=np.int32); d = np.ascontiguousarray(ds, dtype=np.float64)
        if inn.size != d.size:
            raise ValueError("inner and ds must have one entry per cell")
        _check(lib().racgpu_set_star_rays(self._h, inn.size, inn.ctypes.data, d.ctypes.data))

    def column_sweep(self, params, cell_records, y, col_ptr, col_cells, dz, dv_turb=1e5):
        """racgpu_column_sweep: the cells column by column, each column top down on one team of four waves, the toISM
        self-shielding slots (H2, H2O, OH) of every record rewritten from the cells above it before it is solved.
        col_ptr [ncolumn + 1], col_cells [ncell] (cell indices, surface first), dz [ncell] path lengths in cm.
        Returns dict(cells (updated copy), y, t_final, quality, stats, cell_out)."""
        cr = np.array(cell_records, dtype=np.float64, copy=True).reshape(-1, NPAR)
        ncell = cr.shape[0]
        y = np.array(y, dtype=np.float64, copy=True).reshape(ncell, self.nSpecies)
        cp = np.ascontiguousarray(col_ptr, dtype=np.int32); cc = np.ascontiguousarray(col_cells, dtype=np.int32)
        dz = np.ascontiguousarray(dz, dtype=np.float64)
        if cc.size != ncell or dz.size != ncell:
            raise ValueError("col_cells and dz must have one entry per cell")
        tf = np.zeros(ncell); q = np.zeros(ncell, np.int32); st = np.zeros((ncell, NSTAT), np.int64); co = np.zeros((ncell, NOUT))
        _check(lib().racgpu_column_sweep(self._h, C.byref(params), cp.size - 1, cp.ctypes.data, cc.ctypes.data, ncell, cr.ctypes.data,
                                         y.ctypes.data, dz.ctypes.data, float(dv_turb), tf.ctypes.data, q.ctypes.data, st.ctypes.data,
                                         co.ctypes.data, MEM_HOST))
        return dict(cells=cr, y=y, t_final=tf, quality=q, stats=st, cell_out=co)

    def set_cost_hints(self, cost=None):
        """Per-cell expected work (e.g. stats[:, S_NST] of the previous global iteration) for the following
        evol_solve_batch calls: waves take the costliest cells first.  None clears the hint."""
        if cost is None:
            _check(lib().racgpu_set_cost_hints(self._h, None, 0))
            return
        cost = np.ascontiguousarray(cost, dtype=np.float64).ravel()
        _check(lib().racgpu_set_cost_hints(self._h, cost.ctypes.data_as(C.POINTER(C.c_double)), cost.size))

    def set_team_threshold(self, frac):
        """With cost hints: cells expected to cost more than frac x (sum of costs / wave slots) are solved by four waves each
        (racgpu_set_team_threshold; default 0.5, <= 0 never; < 0 also switches off the hand-over of the last running cells to teams at
        the end of a pass).  Results do not depend on it."""
        _check(lib().racgpu_set_team_threshold(self._h, float(frac)))

    def last_team_cells(self):
        """cells the last solve pass gave to four-wave teams"""
        return int(lib().racgpu_last_team_cells(self._h))

    def last_parked_cells(self):
        """cells the last solve pass handed over to teams at its end (between two output times, once the queue was empty)"""
        return int(lib().racgpu_last_parked_cells(self._h))

    def last_kernel_ms(self):
        return lib().racgpu_last_kernel_ms(self._h)
