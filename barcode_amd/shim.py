"""ctypes view of the compiled C++ host layer (``barcode_amd/shim/hmc_hip_shim.cc``, ``include/bchmc_shim.hpp``).

The C++ layer is the product's host side in the reference's language: ``Hamiltonian_EoM``, ``delta_Hamiltonian``,
``gradient_psi``, ``measure_spectrum`` on a view of ``HAMIL_DATA`` / ``HAMIL_NUMERICAL``, throwing
``std::runtime_error`` like the reference.  This module only exists so that the test-suite can drive that compiled
code (through its ``extern "C"`` hooks) with the same seeded cases it uses everywhere else.
"""
import ctypes as C
import os

import numpy as np

from . import engine as _engine

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

SHIM_EXPORTS = ("bchmc_shim_Hamiltonian_EoM", "bchmc_shim_delta_Hamiltonian", "bchmc_shim_gradient_psi",
                "bchmc_shim_measure_spectrum", "bchmc_shim_chain_set_state", "bchmc_shim_chain_get_state",
                "bchmc_shim_HamiltonianMC", "bchmc_shim_release", "bchmc_shim_sizeof_view",
                "bchmc_shim_sizeof_numerical", "bchmc_shim_sizeof_attempt_log",
                "bchmc_shim_HamiltonianMC_scripted", "bchmc_shim_kinetic_term", "bchmc_shim_psi",
                "bchmc_shim_eps_create", "bchmc_shim_eps_destroy", "bchmc_shim_eps_append", "bchmc_shim_eps_records",
                "bchmc_shim_eps_acceptance_rate", "bchmc_shim_update_eps_fac", "bchmc_shim_update_tables",
                "bchmc_shim_comm_bootstrap_file", "bchmc_shim_comm_attach", "bchmc_shim_comm_release",
                "bchmc_shim_inputs_changed", "bchmc_shim_mass_changed", "bchmc_shim_bootstrap_exchange_id",
                "bchmc_shim_bootstrap_cleanup")

_dp = C.POINTER(C.c_double)


class HamilNumericalView(C.Structure):
    """bchmc_shim::HamilNumericalView (members of HAMIL_NUMERICAL, struct_hamil.h:51-144)."""
    _fields_ = [
        ("N1", C.c_uint), ("N", C.c_ulong),
        ("L1", C.c_double), ("min1", C.c_double), ("min2", C.c_double), ("min3", C.c_double),
        ("xobs", C.c_double), ("yobs", C.c_double), ("zobs", C.c_double),
        ("planepar", C.c_bool), ("periodic", C.c_bool),
        ("mk", C.c_int), ("calc_h", C.c_int), ("mass_type", C.c_int),
        ("correct_delta", C.c_bool), ("div_dH_by_N", C.c_bool),
        ("particle_kernel_h", C.c_double), ("kth", C.c_double),
        ("grad_psi_prior_factor", C.c_double), ("grad_psi_likeli_factor", C.c_double), ("deltaQ_factor", C.c_double),
        ("N_eps_fac", C.c_double), ("eps_fac", C.c_double), ("epsilon", C.c_double), ("Neps", C.c_ulong),
        ("dH", C.c_double), ("dK", C.c_double), ("dE", C.c_double), ("dprior", C.c_double), ("dlikeli", C.c_double),
        ("psi_prior", C.c_double), ("psi_likeli", C.c_double),
        ("psi_prior_i", C.c_double), ("psi_prior_f", C.c_double), ("psi_likeli_i", C.c_double),
        ("psi_likeli_f", C.c_double), ("H_kin_i", C.c_double), ("H_kin_f", C.c_double),
        ("iGibbs", C.c_ulong), ("rejections", C.c_ulong), ("accepted", C.c_bool),
    ]


class EomEnergies(C.Structure):
    """bchmc_shim::HamilView::EomEnergies: the six terms Hamiltonian_EoM keeps for the delta_Hamiltonian that follows."""
    _fields_ = [("valid", C.c_bool), ("ptr", _dp * 4), ("hash", C.c_uint64 * 4), ("terms", C.c_double * 6),
                ("dH", C.c_double), ("inputs_generation", C.c_ulong), ("mass_generation", C.c_ulong)]


class HamilView(C.Structure):
    """bchmc_shim::HamilView (members of HAMIL_DATA, struct_hamil.h:146-222)."""
    _fields_ = [
        ("numerical", C.POINTER(HamilNumericalView)),
        ("likelihood", C.c_int), ("sfmodel", C.c_int), ("rsd_model", C.c_bool),
        ("rho_c", C.c_double), ("delta_min", C.c_double), ("biasP", C.c_double), ("biasE", C.c_double),
        ("ascale", C.c_double), ("D1", C.c_double), ("D2", C.c_double), ("OM", C.c_double), ("OL", C.c_double),
        ("signal_PS", _dp), ("mass_f", _dp), ("mass_r", _dp), ("nobs", _dp), ("noise", _dp), ("window", _dp),
        ("gradpsi", _dp), ("deltaX", _dp), ("posx", _dp), ("posy", _dp), ("posz", _dp),
        ("device", C.c_int), ("engine", C.c_void_p),
        ("eps", C.c_void_p), ("comm", C.c_void_p), ("comm_rank", C.c_int),
        ("inputs_generation", C.c_ulong), ("uploaded_generation", C.c_ulong), ("deterministic", C.c_int),
        ("mass_generation", C.c_ulong), ("mass_uploaded_generation", C.c_ulong),
        ("reuse_eom_energies", C.c_int), ("eom", EomEnergies),
    ]


class AttemptLog(C.Structure):
    """bchmc_shim::AttemptLog: one row of performance_log.txt (HMC.cc:40-60)."""
    _fields_ = [("accepted", C.c_bool), ("epsilon", C.c_double), ("Neps", C.c_ulong), ("steps_done", C.c_ulong),
                ("dH", C.c_double), ("dK", C.c_double), ("dE", C.c_double), ("dprior", C.c_double),
                ("dlikeli", C.c_double), ("psi_prior_i", C.c_double), ("psi_prior_f", C.c_double),
                ("psi_likeli_i", C.c_double), ("psi_likeli_f", C.c_double), ("H_kin_i", C.c_double),
                ("H_kin_f", C.c_double)]


UNIFORM_FN = C.CFUNCTYPE(C.c_double, C.c_void_p)


def load():
    """Load libbarcode_shim.so (after libbarcode_hip.so, which it links).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _engine.load()
    path = os.path.join(_HERE, "libbarcode_shim.so")
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `make -C barcode_amd/shim` (after barcode_amd/csrc)" % path)
    lib = C.CDLL(path)
    for s in SHIM_EXPORTS:
        getattr(lib, s)
    hv, sz, ul = C.POINTER(HamilView), C.c_size_t, C.c_ulong
    lib.bchmc_shim_sizeof_view.restype = sz
    lib.bchmc_shim_sizeof_numerical.restype = sz
    lib.bchmc_shim_Hamiltonian_EoM.argtypes = [hv, _dp, _dp, _dp, _dp, UNIFORM_FN, C.c_void_p, C.POINTER(ul),
                                               C.POINTER(ul), C.c_char_p, sz]
    lib.bchmc_shim_delta_Hamiltonian.argtypes = [hv, _dp, _dp, _dp, _dp, _dp, C.c_char_p, sz]
    lib.bchmc_shim_gradient_psi.argtypes = [hv, _dp, C.c_char_p, sz]
    lib.bchmc_shim_measure_spectrum.argtypes = [hv, _dp, _dp, _dp, ul, C.c_char_p, sz]
    lib.bchmc_shim_chain_set_state.argtypes = [hv, _dp, C.c_char_p, sz]
    lib.bchmc_shim_chain_get_state.argtypes = [hv, _dp, C.c_char_p, sz]
    lib.bchmc_shim_HamiltonianMC.argtypes = [hv, UNIFORM_FN, C.c_void_p, C.c_uint64, ul, C.POINTER(ul),
                                             C.POINTER(AttemptLog), ul, C.POINTER(ul), C.c_char_p, sz]
    lib.bchmc_shim_HamiltonianMC_scripted.argtypes = [hv, _dp, ul, UNIFORM_FN, C.c_void_p, ul, C.POINTER(ul),
                                                      C.POINTER(AttemptLog), ul, C.POINTER(ul), C.c_char_p, sz]
    lib.bchmc_shim_kinetic_term.argtypes = [hv, _dp, _dp, C.c_char_p, sz]
    lib.bchmc_shim_psi.argtypes = [hv, _dp, _dp, C.c_char_p, sz]
    lib.bchmc_shim_eps_create.argtypes = [C.c_int, C.c_uint, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                          C.c_double, ul]
    lib.bchmc_shim_eps_create.restype = C.c_void_p
    lib.bchmc_shim_eps_destroy.argtypes = [C.c_void_p]
    lib.bchmc_shim_eps_destroy.restype = None
    lib.bchmc_shim_eps_append.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.bchmc_shim_eps_append.restype = None
    lib.bchmc_shim_eps_records.argtypes = [C.c_void_p]
    lib.bchmc_shim_eps_records.restype = ul
    lib.bchmc_shim_eps_acceptance_rate.argtypes = [C.c_void_p]
    lib.bchmc_shim_eps_acceptance_rate.restype = C.c_double
    lib.bchmc_shim_update_eps_fac.argtypes = [hv, C.c_char_p, sz, C.c_char_p, sz]
    lib.bchmc_shim_update_tables.argtypes = [hv, C.c_char_p, sz]
    lib.bchmc_shim_comm_bootstrap_file.argtypes = [hv, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_char_p, sz]
    lib.bchmc_shim_bootstrap_exchange_id.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_ubyte),
                                                     C.c_char_p, sz]
    lib.bchmc_shim_bootstrap_cleanup.argtypes = [C.c_char_p, C.c_int]
    lib.bchmc_shim_bootstrap_cleanup.restype = None
    lib.bchmc_shim_comm_attach.argtypes = [hv, C.c_void_p, C.c_int]
    lib.bchmc_shim_comm_release.argtypes = [hv]
    lib.bchmc_shim_comm_release.restype = None
    lib.bchmc_shim_inputs_changed.argtypes = [hv]
    lib.bchmc_shim_inputs_changed.restype = None
    lib.bchmc_shim_mass_changed.argtypes = [hv]
    lib.bchmc_shim_mass_changed.restype = None
    lib.bchmc_shim_sizeof_attempt_log.restype = sz
    lib.bchmc_shim_release.argtypes = [hv]
    lib.bchmc_shim_release.restype = None
    if lib.bchmc_shim_sizeof_view() != C.sizeof(HamilView) or \
            lib.bchmc_shim_sizeof_numerical() != C.sizeof(HamilNumericalView) or \
            lib.bchmc_shim_sizeof_attempt_log() != C.sizeof(AttemptLog):
        raise ImportError("bchmc_shim.hpp and barcode_amd/shim.py disagree on the struct layouts")
    _lib = lib
    return lib


class ShimError(RuntimeError):
    """The C++ layer threw std::runtime_error."""


def bootstrap_exchange_id(path, rank, world, unique_id=None, timeout_s=30.0):
    """bchmc_shim::bootstrap_exchange_id: the file protocol that carries rank 0's 128-byte id to the other ranks
    (host only).  Rank 0 passes ``unique_id``; every rank gets the id back."""
    lib = load()
    buf = (C.c_ubyte * _engine.UNIQUE_ID_BYTES)()
    if rank == 0:
        buf[:] = bytes(unique_id)
    err = C.create_string_buffer(512)
    if lib.bchmc_shim_bootstrap_exchange_id(str(path).encode(), int(rank), int(world), float(timeout_s), buf, err, len(err)):
        raise ShimError(err.value.decode())
    return bytes(buf)


def bootstrap_cleanup(path, rank):
    load().bchmc_shim_bootstrap_cleanup(str(path).encode(), int(rank))


def _p(a):
    return a.ctypes.data_as(_dp)


class ShimHamil:
    """A HAMIL_DATA view filled from HamilParams + arrays, driving the compiled C++ functions."""

    def __init__(self, params, N_eps_fac=8.0, eps_fac=None, device=0, **arrays):
        self.lib = load()
        p = params
        self.N = p.N
        n = HamilNumericalView()
        n.N1, n.N, n.L1 = p.Nx, p.N, p.L
        for k in ("min1", "min2", "min3", "xobs", "yobs", "zobs", "mk", "calc_h", "mass_type", "particle_kernel_h",
                  "kth", "grad_psi_prior_factor", "grad_psi_likeli_factor", "deltaQ_factor"):
            setattr(n, k, getattr(p, k))
        n.planepar, n.periodic = bool(p.planepar), bool(p.periodic)
        n.correct_delta, n.div_dH_by_N = bool(p.correct_delta), bool(p.div_dH_by_N)
        n.N_eps_fac = N_eps_fac
        n.eps_fac = p.eps_heuristic() if eps_fac is None else eps_fac
        self.numerical = n
        hd = HamilView()
        hd.numerical = C.pointer(n)
        hd.likelihood, hd.sfmodel, hd.rsd_model = p.likelihood, p.sfmodel, bool(p.rsd_model)
        for k in ("rho_c", "delta_min", "biasP", "biasE", "ascale", "D1", "D2", "OM", "OL"):
            setattr(hd, k, getattr(p, k))
        self._keep = {}
        for k in ("signal_PS", "mass_f", "mass_r", "nobs", "noise", "window"):
            if arrays.get(k) is not None:
                a = np.ascontiguousarray(arrays[k], dtype=np.float64).reshape(-1)
                self._keep[k] = a
                setattr(hd, k, _p(a))
        for k in ("gradpsi", "deltaX", "posx", "posy", "posz"):
            a = np.zeros(p.N)
            self._keep[k] = a
            setattr(hd, k, _p(a))
        hd.device = device
        hd.reuse_eom_energies = 1
        self.hd = hd
        self.count_attempts = C.c_ulong(0)
        self._err = C.create_string_buffer(512)

    def _chk(self, rc):
        if rc:
            raise ShimError(self._err.value.decode())

    def _in(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        assert a.size == self.N
        return a

    def Hamiltonian_EoM(self, signali, momentai, uniform, out=None):
        qf, pf = out if out is not None else (np.empty(self.N), np.empty(self.N))
        done = C.c_ulong(0)
        cb = UNIFORM_FN(lambda _state: float(uniform()))
        self._chk(self.lib.bchmc_shim_Hamiltonian_EoM(C.byref(self.hd), _p(self._in(signali)), _p(self._in(momentai)),
                                                      _p(qf), _p(pf), cb, None, C.byref(self.count_attempts),
                                                      C.byref(done), self._err, len(self._err)))
        return qf, pf, int(done.value)

    def delta_Hamiltonian(self, signali, momentai, signalf, momentaf):
        dH = C.c_double(0)
        self._chk(self.lib.bchmc_shim_delta_Hamiltonian(C.byref(self.hd), _p(self._in(signali)), _p(self._in(momentai)),
                                                        _p(self._in(signalf)), _p(self._in(momentaf)), C.byref(dH),
                                                        self._err, len(self._err)))
        return dH.value

    def gradient_psi(self, signal):
        self._chk(self.lib.bchmc_shim_gradient_psi(C.byref(self.hd), _p(self._in(signal)), self._err, len(self._err)))
        return self._keep["gradpsi"]

    def measure_spectrum(self, signal, N_bin=200):
        km, pw = np.empty(N_bin), np.empty(N_bin)
        self._chk(self.lib.bchmc_shim_measure_spectrum(C.byref(self.hd), _p(self._in(signal)), _p(km), _p(pw), N_bin,
                                                       self._err, len(self._err)))
        return km, pw

    def chain_set_state(self, x):
        self._chk(self.lib.bchmc_shim_chain_set_state(C.byref(self.hd), _p(self._in(x)), self._err, len(self._err)))

    def chain_get_state(self):
        x = np.empty(self.N)
        self._chk(self.lib.bchmc_shim_chain_get_state(C.byref(self.hd), _p(x), self._err, len(self._err)))
        return x

    def HamiltonianMC(self, uniform, seed=1, itmax=2000):
        """One sample of the C++ HamiltonianMC loop (device momentum draw).  Returns the list of attempt records."""
        log = (AttemptLog * itmax)()
        n = C.c_ulong(0)
        cb = UNIFORM_FN(lambda _state: float(uniform()))
        self._chk(self.lib.bchmc_shim_HamiltonianMC(C.byref(self.hd), cb, None, int(seed), int(itmax),
                                                    C.byref(self.count_attempts), log, int(itmax), C.byref(n),
                                                    self._err, len(self._err)))
        return [{k: getattr(log[i], k) for k, _ in AttemptLog._fields_} for i in range(n.value)]

    def HamiltonianMC_scripted(self, script_dH, uniform, itmax=2000, log_cap=None):
        """The same C++ loop on a scripted engine (attempt k returns dH = script_dH[k]): bookkeeping tests, no GPU."""
        cap = itmax if log_cap is None else int(log_cap)
        log = (AttemptLog * max(cap, 1))()
        n = C.c_ulong(0)
        cb = UNIFORM_FN(lambda _state: float(uniform()))
        sc = np.ascontiguousarray(script_dH, dtype=np.float64)
        self._chk(self.lib.bchmc_shim_HamiltonianMC_scripted(C.byref(self.hd), _p(sc), sc.size, cb, None, int(itmax),
                                                             C.byref(self.count_attempts),
                                                             log if cap > 0 else None, cap, C.byref(n), self._err,
                                                             len(self._err)))
        return n.value, [{k: getattr(log[i], k) for k, _ in AttemptLog._fields_} for i in range(min(n.value, cap))]

    def kinetic_term(self, momenta):
        out = C.c_double(0)
        self._chk(self.lib.bchmc_shim_kinetic_term(C.byref(self.hd), _p(self._in(momenta)), C.byref(out), self._err,
                                                   len(self._err)))
        return out.value

    def psi(self, signal):
        out = C.c_double(0)
        self._chk(self.lib.bchmc_shim_psi(C.byref(self.hd), _p(self._in(signal)), C.byref(out), self._err,
                                          len(self._err)))
        return out.value

    # ---- step-size adaptation (time_step.cpp) -------------------------------------------------------------
    def eps_attach(self, cfg):
        """Create the C++ EpsAdapt from a barcode_amd.time_step.EpsConfig and hang it on hd->eps."""
        e = self.lib.bchmc_shim_eps_create(cfg.eps_fac_update_type, cfg.N_a_eps_update, cfg.acc_min, cfg.acc_max,
                                           cfg.eps_down_smooth, cfg.eps_up_fac, cfg.eps_fac_target, cfg.eps_fac_power,
                                           cfg.s_eps_total)
        if not e:
            raise ShimError("eps_adapt_create failed")
        self.hd.eps = e
        return e

    def eps_append(self, accepted, epsilon):
        self.lib.bchmc_shim_eps_append(self.hd.eps, int(bool(accepted)), float(epsilon))

    def eps_records(self):
        return int(self.lib.bchmc_shim_eps_records(self.hd.eps))

    def eps_acceptance_rate(self):
        return float(self.lib.bchmc_shim_eps_acceptance_rate(self.hd.eps))

    def update_eps_fac(self):
        msg = C.create_string_buffer(256)
        self._chk(self.lib.bchmc_shim_update_eps_fac(C.byref(self.hd), msg, len(msg), self._err, len(self._err)))
        return msg.value.decode()

    def update_epsilon_acc_rate_tables(self):
        self._chk(self.lib.bchmc_shim_update_tables(C.byref(self.hd), self._err, len(self._err)))

    def comm_attach(self, comm, rank):
        self.lib.bchmc_shim_comm_attach(C.byref(self.hd), comm, int(rank))

    def comm_bootstrap_file(self, path, rank, world, timeout_s=60.0):
        self._chk(self.lib.bchmc_shim_comm_bootstrap_file(C.byref(self.hd), path.encode(), int(rank), int(world),
                                                          float(timeout_s), self._err, len(self._err)))

    def comm_release(self):
        self.lib.bchmc_shim_comm_release(C.byref(self.hd))

    def inputs_changed(self):
        self.lib.bchmc_shim_inputs_changed(C.byref(self.hd))

    def mass_changed(self):
        """Only mass_f / mass_r were rewritten (HMC.cc:400-423): the resident chain keeps its carried gradient."""
        self.lib.bchmc_shim_mass_changed(C.byref(self.hd))

    def out(self, name):
        """hd->gradpsi / deltaX / posx / posy / posz as the C++ layer left them."""
        return self._keep[name]

    def close(self):
        self.lib.bchmc_shim_release(C.byref(self.hd))
        if self.hd.eps:
            self.lib.bchmc_shim_eps_destroy(self.hd.eps)
            self.hd.eps = None
