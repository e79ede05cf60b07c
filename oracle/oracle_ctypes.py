"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (rac-2d_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
NPAR = 28


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


class _Net(C.Structure):
    _fields_ = [
        ("nS", C.c_int), ("nR", C.c_int), ("NEQ", C.c_int),
        ("names", C.c_void_p), ("reac", C.POINTER(C.c_int)), ("prod", C.POINTER(C.c_int)),
        ("n_reac", C.POINTER(C.c_int)), ("n_prod", C.POINTER(C.c_int)), ("itype", C.POINTER(C.c_int)),
        ("ABC", C.POINTER(C.c_double)), ("Trange", C.POINTER(C.c_double)),
        ("ctype", C.c_void_p), ("reac_name1", C.c_void_p),
        ("dupli_ptr", C.POINTER(C.c_int)), ("dupli_list", C.POINTER(C.c_int)),
        ("elements", C.POINTER(C.c_int)), ("mass_num", C.POINTER(C.c_double)),
        ("vib_freq", C.POINTER(C.c_double)), ("Edesorb", C.POINTER(C.c_double)),
        ("counterpart", C.POINTER(C.c_int)), ("nGrain", C.c_int), ("idxGrain", C.POINTER(C.c_int)),
        ("idx10", C.c_int * 10),
        ("i_Grain0", C.c_int), ("i_GrainM", C.c_int), ("i_GrainP", C.c_int),
        ("i_gH", C.c_int), ("i_gH2", C.c_int), ("i_gH2O", C.c_int),
        ("NNZ", C.c_int), ("IA", C.POINTER(C.c_int)), ("JA", C.POINTER(C.c_int)),
    ]


class Params(C.Structure):
    _fields_ = [
        ("RTOL", C.c_double), ("ATOL", C.c_double), ("t_max", C.c_double),
        ("dt_first_step", C.c_double), ("ratio_tstep", C.c_double),
        ("mxstep_per_interval", C.c_int), ("steps_reset_solver", C.c_int), ("H2_form_use_moeq", C.c_int),
        ("Diff2DesorRatio", C.c_double), ("special_gH_E_diff", C.c_double),
        ("use_special_gH_mobi", C.c_int), ("update_gH_params_realtime", C.c_int),
        ("max_runtime_allowed", C.c_double),
        ("rt_cost_f", C.c_double), ("rt_cost_jac", C.c_double), ("rt_cost_lu", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [("nst", C.c_long), ("nfe", C.c_long), ("nje", C.c_long), ("nlu", C.c_long),
                ("nnz", C.c_int), ("nzl", C.c_int), ("nzu", C.c_int),
                ("nst_last", C.c_long), ("nfe_last", C.c_long), ("nje_last", C.c_long), ("nlu_last", C.c_long)]


class IterInfo(C.Structure):
    _fields_ = [("t0", C.c_double), ("dt_first", C.c_double), ("t_end", C.c_double), ("t_final", C.c_double),
                ("n_mol_on_grain", C.c_double), ("n_record", C.c_int), ("quality", C.c_int), ("nerr", C.c_int),
                ("isav", C.c_int), ("proceeds", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_network_load.restype = C.POINTER(_Net)
        L.orc_network_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.orc_network_free.argtypes = [C.POINTER(_Net)]
        L.orc_species_index.argtypes = [C.POINTER(_Net), C.c_char_p]
        L.orc_n_record.argtypes = [C.POINTER(Params), C.c_double, C.c_double]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Network:
    def __init__(self, path):
        err = C.create_string_buffer(256)
        self._p = lib().orc_network_load(path.encode(), err, 256)
        if not self._p:
            raise RuntimeError(err.value.decode())
        n = self._p.contents
        self.nS, self.nR, self.NEQ, self.NNZ = n.nS, n.nR, n.NEQ, n.NNZ
        raw = C.string_at(n.names, self.nS * 13)
        self.names = [raw[i * 13:(i + 1) * 13].split(b"\0")[0].decode() for i in range(self.nS)]
        self.reac = np.ctypeslib.as_array(n.reac, (self.nR, 3)).copy()
        self.prod = np.ctypeslib.as_array(n.prod, (self.nR, 4)).copy()
        self.n_reac = np.ctypeslib.as_array(n.n_reac, (self.nR,)).copy()
        self.n_prod = np.ctypeslib.as_array(n.n_prod, (self.nR,)).copy()
        self.itype = np.ctypeslib.as_array(n.itype, (self.nR,)).copy()
        self.ABC = np.ctypeslib.as_array(n.ABC, (self.nR, 3)).copy()
        self.Trange = np.ctypeslib.as_array(n.Trange, (self.nR, 2)).copy()
        self.dupli_ptr = np.ctypeslib.as_array(n.dupli_ptr, (self.nR + 1,)).copy()
        self.mass_num = np.ctypeslib.as_array(n.mass_num, (self.nS,)).copy()
        self.vib_freq = np.ctypeslib.as_array(n.vib_freq, (self.nS,)).copy()
        self.Edesorb = np.ctypeslib.as_array(n.Edesorb, (self.nS,)).copy()
        self.counterpart = np.ctypeslib.as_array(n.counterpart, (self.nS,)).copy()
        self.elements = np.ctypeslib.as_array(n.elements, (self.nS, 20)).copy()  # getElements: column 0 = charge, then the 19 elements
        self.charge = self.elements[:, 0].copy()
        self.IA = np.ctypeslib.as_array(n.IA, (self.NEQ + 1,)).copy()
        self.JA = np.ctypeslib.as_array(n.JA, (self.NNZ,)).copy()
        self.i_Grain0 = n.i_Grain0

    def __del__(self):
        try:
            lib().orc_network_free(self._p)
        except Exception:
            pass

    def index(self, name):
        return lib().orc_species_index(self._p, name.encode())

    def initial_abundances(self, path):
        y0 = np.zeros(self.nS)
        rc = lib().orc_load_initial_abundances(self._p, path.encode(), _dp(y0))
        if rc:
            raise RuntimeError("orc_load_initial_abundances rc=%d" % rc)
        return y0

    def tolerances(self, params, j, d2h):
        r = np.zeros(self.NEQ); a = np.zeros(self.NEQ)
        lib().orc_set_tolerances(self._p, C.byref(params), C.c_int(j), C.c_double(d2h), _dp(r), _dp(a))
        return r, a

    def rates(self, params, cell):
        cell = np.ascontiguousarray(cell, dtype=np.float64)
        out = np.zeros(self.nR)
        rc = lib().orc_cal_rates(self._p, C.byref(params), _dp(cell), _dp(out), None)
        if rc:
            raise RuntimeError("orc_cal_rates rc=%d" % rc)
        return out

    def rhs(self, params, cell, rates, y):
        cell = np.ascontiguousarray(cell, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        out = np.zeros(self.NEQ)
        lib().orc_ode_f(self._p, C.byref(params), _dp(cell), _dp(rates), _dp(y), _dp(out))
        return out

    def jac_col(self, params, cell, rates, y, j):
        cell = np.ascontiguousarray(cell, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        out = np.zeros(self.NEQ)
        lib().orc_ode_jac_col(self._p, C.byref(params), _dp(cell), _dp(rates), _dp(y), C.c_int(j), _dp(out))
        return out

    def jac_csc(self, params, cell, rates, y):
        cell = np.ascontiguousarray(cell, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        out = np.zeros(self.NNZ)
        lib().orc_jac_csc(self._p, C.byref(params), _dp(cell), _dp(rates), _dp(y), _dp(out))
        return out

    def initial_state(self, y0, cell):
        y = np.zeros(self.NEQ)
        y[:self.nS] = y0
        if self.i_Grain0 > 0:
            y[self.i_Grain0 - 1] = cell[6]
        y[self.nS] = cell[0]
        return y

    def calc_cell(self, params, cell, y0, nlocal_iter=4, y_init=None):
        """The local-iteration loop of calc_this_cell (reference src/disk.f90:1651-1791) for one cell.  y_init [nS] overrides
        the initial abundances (default: y0 with the Grain0 slot set)."""
        cell = np.ascontiguousarray(cell, dtype=np.float64)
        yi = self.initial_state(y0, cell)[:self.nS].copy() if y_init is None else np.ascontiguousarray(y_init, dtype=np.float64)
        ab = np.zeros(self.nS)
        tf = C.c_double(); q = C.c_int(); st = Stats()
        info = (IterInfo * nlocal_iter)()
        n = lib().orc_calc_cell(self._p, C.byref(params), _dp(cell), _dp(yi), C.c_int(nlocal_iter), _dp(ab), C.byref(tf), C.byref(q), info, C.byref(st))
        iters = [dict(t0=i.t0, dt_first=i.dt_first, t_end=i.t_end, t_final=i.t_final, n_mol_on_grain=i.n_mol_on_grain, n_record=i.n_record,
                      quality=i.quality, nerr=i.nerr, isav=i.isav, proceeds=i.proceeds) for i in info[:max(n, 0)]]
        return dict(rc=n, y=ab, t_final=tf.value, quality=q.value, iters=iters, nst=st.nst, nfe=st.nfe, nje=st.nje, nlu=st.nlu)

    def solve_cell(self, params, cell, y0, record=False, j=1, y_init=None, t0=0.0, rectify=False):
        """One chem_evol_solve run of one cell (tolerance policy j, start time t0 with the caller's continue rule for the first
        step, optional rectify_abundances); returns dict."""
        cell = np.ascontiguousarray(cell, dtype=np.float64)
        y = self.initial_state(y0, cell)
        if y_init is not None:
            y[:self.nS] = y_init
        if rectify:
            lib().orc_rectify_abundances(self._p, _dp(y))
        t_max = cell[27] if cell[27] > 0 else params.t_max
        if t0 > 0.0:
            pp = Params(); C.memmove(C.byref(pp), C.byref(params), C.sizeof(Params))
            pp.dt_first_step = max(params.dt_first_step, 1e-3 * t0)
            params = pp
        rtol, atol = self.tolerances(params, j, cell[6])
        rates = self.rates(params, cell)
        nrec = lib().orc_n_record(C.byref(params), t0, t_max)
        rec = np.zeros((nrec, self.NEQ)) if record else None
        touts = np.zeros(nrec)
        tf = C.c_double(); q = C.c_int(); ne = C.c_int(); nrr = C.c_int(); st = Stats()
        rc = lib().orc_evol_solve(self._p, C.byref(params), _dp(cell), _dp(rates), _dp(rtol), _dp(atol), _dp(y),
                                  C.c_double(t0), C.c_double(t_max), C.byref(tf), C.byref(q), C.byref(ne),
                                  C.byref(nrr), _dp(rec) if record else None, _dp(touts), C.byref(st))
        return dict(rc=rc, y=y, t_final=tf.value, quality=q.value, nerr=ne.value, n_record_real=nrr.value,
                    record=rec, touts=touts, nst=st.nst, nfe=st.nfe, nje=st.nje, nlu=st.nlu,
                    nnz=st.nnz, nzl=st.nzl, nzu=st.nzu, nst_last=st.nst_last, nfe_last=st.nfe_last,
                    nje_last=st.nje_last, nlu_last=st.nlu_last)


def default_params():
    p = Params()
    lib().orc_params_default(C.byref(p))
    return p
