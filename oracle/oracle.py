"""ctypes loader for the C oracle (``liboracle.so`` / ``liboracle_omp.so``) -- TEST INFRASTRUCTURE ONLY.

The oracle is a plain-C restatement of the reference hot path (``oracle/bchmc_oracle.c``).  PARITY
UNPINNED: the reference holds no golden vectors for this path and cannot be built in this image.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

FIELDS = dict(signal_PS=0, mass_f=1, mass_r=2, nobs=3, noise=4, window=5,
              gradpsi=6, deltaX=7, posx=8, posy=9, posz=10)

ERRORS = {1: "bad argument", 2: "masskernel must be 3 (SPH) for calc_h 2/3",
          3: "non-plane-parallel RSD not implemented", 4: "invalid mass_type", 5: "unsupported in oracle"}


class OrcConfig(C.Structure):
    _fields_ = [
        ("N1", C.c_uint32), ("L1", C.c_double),
        ("min1", C.c_double), ("min2", C.c_double), ("min3", C.c_double),
        ("xobs", C.c_double), ("yobs", C.c_double), ("zobs", C.c_double),
        ("planepar", C.c_int32), ("periodic", C.c_int32),
        ("mk", C.c_int32), ("calc_h", C.c_int32), ("likelihood", C.c_int32), ("sfmodel", C.c_int32),
        ("rsd_model", C.c_int32), ("mass_type", C.c_int32), ("correct_delta", C.c_int32),
        ("div_dH_by_N", C.c_int32),
        ("particle_kernel_h", C.c_double),
        ("grad_psi_prior_factor", C.c_double), ("grad_psi_likeli_factor", C.c_double),
        ("deltaQ_factor", C.c_double),
        ("rho_c", C.c_double), ("delta_min", C.c_double), ("biasP", C.c_double), ("biasE", C.c_double),
        ("ascale", C.c_double), ("D1", C.c_double), ("D2", C.c_double), ("OM", C.c_double), ("OL", C.c_double),
        ("kth", C.c_double),
    ]


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__("oracle error %d: %s" % (code, ERRORS.get(code, "?")))
        self.code = code


def build(force=False):
    """Compile the oracle's two shared objects (serial + OpenMP) with gcc."""
    targets = [os.path.join(_HERE, n) for n in ("liboracle.so", "liboracle_omp.so")]
    if force or not all(os.path.exists(t) for t in targets):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return targets


_libs = {}


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota.  A GPU box shows all
    256 hardware threads of its host but grants about 16 of them; sizing OpenMP by the visible count makes the threads
    spin against the quota (measured: 18 s instead of 0.3 s per 64^3 leapfrog step)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse is not None:
                quota, period = parse(txt)
            else:
                quota, period = txt, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1") and int(period) > 0:
                n = min(n, max(1, int(quota) // int(period)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def _lib(omp):
    if omp not in _libs:
        path = os.path.join(_HERE, "liboracle_omp.so" if omp else "liboracle.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        dp, vp = C.POINTER(C.c_double), C.c_void_p
        lib.orc_sizeof_config.restype = C.c_size_t
        lib.orc_create.argtypes = [C.POINTER(OrcConfig), C.POINTER(vp)]
        lib.orc_destroy.argtypes = [vp]
        lib.orc_destroy.restype = None
        lib.orc_set_array.argtypes = [vp, C.c_int, dp]
        lib.orc_get_array.argtypes = [vp, C.c_int]
        lib.orc_get_array.restype = dp
        lib.orc_last_grad_prior.argtypes = [vp]
        lib.orc_last_grad_prior.restype = dp
        lib.orc_last_grad_like.argtypes = [vp]
        lib.orc_last_grad_like.restype = dp
        lib.orc_stencil.argtypes = [vp, C.POINTER(C.c_int)] + [C.POINTER(C.POINTER(C.c_int))] * 3
        lib.orc_fgrow.restype = C.c_double
        lib.orc_fgrow.argtypes = [C.c_double] * 3 + [C.c_int]
        lib.orc_c_pecvel.restype = C.c_double
        lib.orc_c_pecvel.argtypes = [C.c_double] * 3 + [C.c_int]
        for name, args in {
            "orc_convolveInvCorrFuncWithSignal": [vp, dp, dp, dp],
            "orc_theta2vel": [vp, dp, dp, dp, dp],
            "orc_Lag2Eul": [vp, dp, dp, dp, dp, dp, C.c_int],
            "orc_getDensity": [vp, C.c_int, dp, dp, dp, dp],
            "orc_partial_f_delta_x_log_like": [vp, dp, dp],
            "orc_likelihood_calc_V_SPH": [vp] + [dp] * 7,
            "orc_likelihood_calc_V_SPH_fourier_TSC": [vp] + [dp] * 4,
            "orc_likelihood_calc_h_SPH": [vp, dp, dp],
            "orc_likelihood_grad_log_like": [vp, dp, dp],
            "orc_grad_log_prior": [vp, dp, dp],
            "orc_log_prior": [vp, dp, dp],
            "orc_log_like": [vp, dp, dp],
            "orc_gradient_psi": [vp, dp],
            "orc_kinetic_term": [vp, dp, dp],
            "orc_psi": [vp, dp, dp, dp],
            "orc_delta_Hamiltonian": [vp, dp, dp, dp, dp, dp, dp],
            "orc_Hamiltonian_EoM": [vp, dp, dp, dp, dp, C.c_double, C.c_uint64, C.POINTER(C.c_uint64)],
            "orc_PoissonSolver": [vp, dp, dp],
            "orc_calc_m2v_mem": [vp, dp, dp],
            "orc_kernelcomp": [vp, C.c_double, dp],
            "orc_convcomp": [vp, dp, dp, C.c_double],
            "orc_theta2velcomp": [vp, dp, dp, C.c_int],
            "orc_cellboundcomp": [vp, dp],
            "orc_alpt_displacement": [vp, dp, dp, dp, dp],
            "orc_measure_spectrum": [vp, dp, dp, dp, C.c_uint64],
        }.items():
            getattr(lib, name).argtypes = args
            getattr(lib, name).restype = C.c_int
        lib.orc_create_GARFIELD.argtypes = [C.c_uint, C.c_double, dp, C.c_ulong, dp]
        lib.orc_create_GARFIELD.restype = C.c_int
        lib.orc_draw_momenta.argtypes = [C.c_uint, C.c_double, C.c_int, C.c_int, dp, dp, C.c_ulong, dp]
        lib.orc_draw_momenta.restype = C.c_int
        lib.orc_mt19937_stream.argtypes = [C.c_ulong, C.POINTER(C.c_uint32), C.c_size_t]
        lib.orc_mt19937_stream.restype = None
        lib.orc_ugaussian_stream.argtypes = [C.c_ulong, dp, C.c_size_t]
        lib.orc_ugaussian_stream.restype = None
        lib.orc_overdens.argtypes = [vp, dp, dp]
        lib.orc_overdens.restype = None
        assert lib.orc_sizeof_config() == C.sizeof(OrcConfig)
        lib.orc_set_threads.argtypes = [C.c_int]
        lib.orc_set_threads.restype = None
        lib.orc_get_max_threads.restype = C.c_int
        if omp:
            lib.orc_set_threads(int(os.environ.get("BCHMC_ORACLE_THREADS", host_cpu_share())))
        _libs[omp] = lib
    return _libs[omp]


def _p(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_double))


def fgrow(a, OM, OL, term=1):
    return _lib(False).orc_fgrow(a, OM, OL, term)


def c_pecvel(a, OM, OL, term=1):
    return _lib(False).orc_c_pecvel(a, OM, OL, term)


class Oracle:
    """One HAMIL_DATA-like context.  ``params`` is any object with the attributes of ``OrcConfig`` (plus Nx, L)."""

    def __init__(self, params, omp=False):
        self.lib = _lib(omp)
        cfg = OrcConfig()
        for name, _ in OrcConfig._fields_:
            if name == "N1":
                cfg.N1 = int(params.Nx)
            elif name == "L1":
                cfg.L1 = float(params.L)
            else:
                setattr(cfg, name, getattr(params, name))
        self.cfg = cfg
        self.Nx = int(params.Nx)
        self.N = self.Nx ** 3
        self.shape = (self.Nx,) * 3
        self.h = C.c_void_p()
        rc = self.lib.orc_create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise OracleError(rc)

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise OracleError(rc)

    def _in(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        assert a.size == self.N
        return a

    def _new(self):
        return np.empty(self.N, dtype=np.float64)

    # ---- arrays -------------------------------------------------------------------------------
    def set(self, **arrays):
        for k, v in arrays.items():
            self._chk(self.lib.orc_set_array(self.h, FIELDS[k], _p(self._in(v))))

    def get(self, name):
        ptr = self.lib.orc_get_array(self.h, FIELDS[name])
        return np.ctypeslib.as_array(ptr, shape=(self.N,)).copy()

    def stencil(self):
        n = C.c_int()
        ci, cj, ck = (C.POINTER(C.c_int)() for _ in range(3))
        self.lib.orc_stencil(self.h, C.byref(n), C.byref(ci), C.byref(cj), C.byref(ck))
        return np.array([[ci[m], cj[m], ck[m]] for m in range(n.value)])

    # ---- path functions (same names as the reference) -----------------------------------------
    def convolveInvCorrFuncWithSignal(self, signal, corr):
        out = self._new()
        self._chk(self.lib.orc_convolveInvCorrFuncWithSignal(self.h, _p(self._in(signal)), _p(out),
                                                             _p(self._in(corr))))
        return out

    def theta2vel(self, delta):
        vx, vy, vz = self._new(), self._new(), self._new()
        self._chk(self.lib.orc_theta2vel(self.h, _p(self._in(delta)), _p(vx), _p(vy), _p(vz)))
        return vx, vy, vz

    # ---- f-3: pieces of Lag2Eul_non_zeldovich (Lag2Eul.cc:138-312) -----------------------------------------
    def PoissonSolver(self, delta):
        out = self._new()
        self._chk(self.lib.orc_PoissonSolver(self.h, _p(self._in(delta)), _p(out)))
        return out

    def calc_m2v_mem(self, phiv):
        out = self._new()
        self._chk(self.lib.orc_calc_m2v_mem(self.h, _p(self._in(phiv)), _p(out)))
        return out

    def kernelcomp(self, smol):
        out = self._new()
        self._chk(self.lib.orc_kernelcomp(self.h, float(smol), _p(out)))
        return out

    def convcomp(self, a, smol):
        out = self._new()
        self._chk(self.lib.orc_convcomp(self.h, _p(self._in(a)), _p(out), float(smol)))
        return out

    def theta2velcomp(self, delta, comp):
        out = self._new()
        self._chk(self.lib.orc_theta2velcomp(self.h, _p(self._in(delta)), _p(out), int(comp)))
        return out

    def cellboundcomp(self, v):
        out = self._in(v).copy()
        self._chk(self.lib.orc_cellboundcomp(self.h, _p(out)))
        return out

    def alpt_displacement(self, delta):
        px, py, pz = self._new(), self._new(), self._new()
        self._chk(self.lib.orc_alpt_displacement(self.h, _p(self._in(delta)), _p(px), _p(py), _p(pz)))
        return px, py, pz

    def measure_spectrum(self, signal, N_bin=200):
        """field_statistics.cpp:20-90 -> (kmode, power), both N_bin long."""
        kmode, power = np.empty(N_bin), np.empty(N_bin)
        self._chk(self.lib.orc_measure_spectrum(self.h, _p(self._in(signal)), _p(kmode), _p(power), int(N_bin)))
        return kmode, power

    def Lag2Eul(self, delta, rsd=None):
        out, px, py, pz = (self._new() for _ in range(4))
        use_rsd = int(self.cfg.rsd_model if rsd is None else rsd)
        self._chk(self.lib.orc_Lag2Eul(self.h, _p(self._in(delta)), _p(out), _p(px), _p(py), _p(pz), use_rsd))
        return out, px, py, pz

    def getDensity(self, mk, px, py, pz):
        rho = self._new()
        self._chk(self.lib.orc_getDensity(self.h, mk, _p(self._in(px)), _p(self._in(py)), _p(self._in(pz)), _p(rho)))
        return rho

    def overdens(self, rho):
        out = self._new()
        self.lib.orc_overdens(self.h, _p(self._in(rho)), _p(out))
        return out

    def partial_f_delta_x_log_like(self, deltaX):
        out = np.zeros(self.N)
        self._chk(self.lib.orc_partial_f_delta_x_log_like(self.h, _p(self._in(deltaX)), _p(out)))
        return out

    def likelihood_calc_V_SPH(self, part_like, px, py, pz):
        vx, vy, vz = self._new(), self._new(), self._new()
        self._chk(self.lib.orc_likelihood_calc_V_SPH(self.h, _p(self._in(part_like)), _p(self._in(px)),
                                                     _p(self._in(py)), _p(self._in(pz)), _p(vx), _p(vy), _p(vz)))
        return vx, vy, vz

    def likelihood_calc_V_SPH_fourier_TSC(self, part_like, px, py, pz):
        self.set(posx=px, posy=py, posz=pz)
        vx, vy, vz = self._new(), self._new(), self._new()
        self._chk(self.lib.orc_likelihood_calc_V_SPH_fourier_TSC(self.h, _p(self._in(part_like)), _p(vx), _p(vy),
                                                                 _p(vz)))
        return vx, vy, vz

    def likelihood_grad_log_like(self, delta):
        out = self._new()
        self._chk(self.lib.orc_likelihood_grad_log_like(self.h, _p(self._in(delta)), _p(out)))
        return out

    def grad_log_prior(self, signal):
        out = self._new()
        self._chk(self.lib.orc_grad_log_prior(self.h, _p(self._in(signal)), _p(out)))
        return out

    def log_prior(self, signal):
        v = C.c_double()
        self._chk(self.lib.orc_log_prior(self.h, _p(self._in(signal)), C.byref(v)))
        return v.value

    def log_like(self, signal):
        v = C.c_double()
        self._chk(self.lib.orc_log_like(self.h, _p(self._in(signal)), C.byref(v)))
        return v.value

    def gradient_psi(self, signal):
        self._chk(self.lib.orc_gradient_psi(self.h, _p(self._in(signal))))
        gp = np.ctypeslib.as_array(self.lib.orc_last_grad_prior(self.h), shape=(self.N,)).copy()
        gl = np.ctypeslib.as_array(self.lib.orc_last_grad_like(self.h), shape=(self.N,)).copy()
        return self.get("gradpsi"), gp, gl

    def kinetic_term(self, momenta):
        v = C.c_double()
        self._chk(self.lib.orc_kinetic_term(self.h, _p(self._in(momenta)), C.byref(v)))
        return v.value

    def psi(self, signal):
        a, b = C.c_double(), C.c_double()
        self._chk(self.lib.orc_psi(self.h, _p(self._in(signal)), C.byref(a), C.byref(b)))
        return a.value, b.value

    def delta_Hamiltonian(self, qi, pi, qf, pf):
        dH = C.c_double()
        out = np.zeros(6)
        self._chk(self.lib.orc_delta_Hamiltonian(self.h, _p(self._in(qi)), _p(self._in(pi)), _p(self._in(qf)),
                                                 _p(self._in(pf)), C.byref(dH), _p(out)))
        return dH.value, out

    def Hamiltonian_EoM(self, qi, pi, epsilon, Neps):
        qf, pf = self._new(), self._new()
        done = C.c_uint64()
        self._chk(self.lib.orc_Hamiltonian_EoM(self.h, _p(self._in(qi)), _p(self._in(pi)), _p(qf), _p(pf),
                                               float(epsilon), int(Neps), C.byref(done)))
        return qf, pf, done.value


# ---- the reference's momentum draw (oracle/orc_random.c; GSL's MT19937 + polar Box-Muller stream restated) ----
def mt19937_stream(seed, n):
    out = np.zeros(n, dtype=np.uint32)
    _lib(False).orc_mt19937_stream(int(seed), out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    return out


def ugaussian_stream(seed, n):
    out = np.zeros(n)
    _lib(False).orc_ugaussian_stream(int(seed), _p(out), n)
    return out


def create_GARFIELD(n, L, power, seed):
    """random.cpp:48-511: real Gaussian field with <|FFT|^2> = N^2 / V * power, from gsl_rng seed ``seed``."""
    power = np.ascontiguousarray(power, dtype=np.float64).reshape(-1)
    out = np.zeros(n ** 3)
    rc = _lib(False).orc_create_GARFIELD(n, float(L), _p(power), int(seed), _p(out))
    if rc:
        raise OracleError(rc)
    return out


def draw_momenta(params, mass_f, mass_r, seed):
    """HMC_momenta.cc:42-94 for the mass type of ``params``."""
    mt = params.mass_type
    fs, rs = int(mt in (1, 2, 3, 4, 5)), int(mt in (0, 5, 6, 60))
    mf = None if mass_f is None else np.ascontiguousarray(mass_f, dtype=np.float64).reshape(-1)
    mr = None if mass_r is None else np.ascontiguousarray(mass_r, dtype=np.float64).reshape(-1)
    out = np.zeros(params.N)
    rc = _lib(False).orc_draw_momenta(params.Nx, float(params.L), fs, rs, _p(mf) if mf is not None else None,
                                      _p(mr) if mr is not None else None, int(seed), _p(out))
    if rc:
        raise OracleError(rc)
    return out
