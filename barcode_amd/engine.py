"""ctypes binding of ``libbarcode_hip.so`` (C ABI: ``include/bchmc.h``).

There is no CPU fallback: if the HIP library is missing or fails to load this module raises.
``torch`` is imported first on purpose: it brings the process-wide HIP runtime (``libamdhip64.so.7``) and
rocFFT that our library then binds to, so device pointers of torch tensors and the engine's stream live
in the same runtime.
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  (must precede loading libbarcode_hip.so, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# BCHMC_LIB: an alternative build of the same library (A/B runs of kernel variants on one box)
LIB_PATH = os.environ.get("BCHMC_LIB") or os.path.join(_HERE, "libbarcode_hip.so")
ABI_VERSION = 4

FIELDS = dict(signal_PS=0, mass_f=1, mass_r=2, nobs=3, noise=4, window=5, deltaX=6, posx=7, posy=8, posz=9,
              rho=10, part_like=11, Vx=12, Vy=13, Vz=14, psix=15, psiy=16, psiz=17, grad_prior=18, grad_like=19)
INPUT_FIELDS = ("signal_PS", "mass_f", "mass_r", "nobs", "noise", "window")
K_COUNT = 9


class BchmcConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("Nx", C.c_uint32), ("L", C.c_double),
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
        ("precision", C.c_int32), ("device", C.c_int32), ("deterministic", C.c_int32), ("reserved0", C.c_int32),
    ]


class EpsRecord(C.Structure):
    _fields_ = [("epsilon", C.c_double), ("accepted", C.c_int32), ("neps", C.c_int32)]


EPS_BATCH = 32            # BCHMC_EPS_BATCH
UNIQUE_ID_BYTES = 128     # BCHMC_UNIQUE_ID_BYTES
PACKET_BYTES = 8 + 16 * EPS_BATCH
# bchmc_allgather_fn: int (*)(void *ctx, const void *send, void *recv, size_t bytes_per_rank)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class BchmcError(RuntimeError):
    """Raised for any non-zero return code; mirrors the reference's std::runtime_error."""

    def __init__(self, code, text, detail=""):
        super().__init__("bchmc error %d (%s)%s" % (code, text, (": " + detail) if detail else ""))
        self.code = code


_lib = None

# every symbol include/bchmc.h declares; tests check that the library exports all of them
EXPORTS = ("bchmc_create", "bchmc_destroy", "bchmc_strerror", "bchmc_last_error", "bchmc_upload", "bchmc_fetch",
           "bchmc_leapfrog", "bchmc_leapfrog_dh", "bchmc_energies", "bchmc_delta_hamiltonian", "bchmc_gradient", "bchmc_forward",
           "bchmc_leapfrog_device", "bchmc_steps_done", "bchmc_energies_device", "bchmc_sync", "bchmc_stream",
           "bchmc_profile", "bchmc_profile_read", "bchmc_kernel_name", "bchmc_tile_info",
           "bchmc_chain_set_state", "bchmc_chain_get_state", "bchmc_chain_set_momenta", "bchmc_chain_get_momenta",
           "bchmc_chain_draw_momenta", "bchmc_chain_attempt", "bchmc_chain_get_proposal", "bchmc_chain_accept",
           "bchmc_measure_spectrum", "bchmc_philox_kat", "bchmc_kinetic_term", "bchmc_psi",
           "bchmc_comm_unique_id", "bchmc_comm_create", "bchmc_comm_create_custom", "bchmc_comm_destroy",
           "bchmc_comm_last_error", "bchmc_eps_exchange", "bchmc_comm_pending", "bchmc_comm_world", "bchmc_comm_rank",
           "bchmc_comm_transport")


def load():
    """Load the HIP library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, dp, u64 = C.c_void_p, C.POINTER(C.c_double), C.c_uint64
    lib.bchmc_create.argtypes = [C.POINTER(BchmcConfig), C.POINTER(vp)]
    lib.bchmc_destroy.argtypes = [vp]
    lib.bchmc_destroy.restype = None
    lib.bchmc_strerror.argtypes = [C.c_int]
    lib.bchmc_strerror.restype = C.c_char_p
    lib.bchmc_last_error.argtypes = [vp]
    lib.bchmc_last_error.restype = C.c_char_p
    lib.bchmc_upload.argtypes = [vp, C.c_int, dp, C.c_size_t]
    lib.bchmc_fetch.argtypes = [vp, C.c_int, dp, C.c_size_t]
    lib.bchmc_leapfrog.argtypes = [vp, dp, dp, dp, dp, C.c_double, u64, C.POINTER(u64)]
    lib.bchmc_leapfrog_dh.argtypes = [vp, dp, dp, dp, dp, C.c_double, u64, C.POINTER(u64), dp, dp]
    lib.bchmc_energies.argtypes = [vp, dp, dp, dp]
    lib.bchmc_delta_hamiltonian.argtypes = [vp, dp, dp, dp, dp, dp, dp]
    lib.bchmc_gradient.argtypes = [vp, dp, dp]
    lib.bchmc_forward.argtypes = [vp, dp, C.c_int]
    lib.bchmc_leapfrog_device.argtypes = [vp, vp, vp, vp, vp, C.c_double, u64]
    lib.bchmc_steps_done.argtypes = [vp, C.POINTER(u64)]
    lib.bchmc_energies_device.argtypes = [vp, vp, vp, dp]
    lib.bchmc_sync.argtypes = [vp]
    lib.bchmc_stream.argtypes = [vp]
    lib.bchmc_stream.restype = vp
    lib.bchmc_tile_info.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.bchmc_profile.argtypes = [vp, C.c_int]
    lib.bchmc_profile_read.argtypes = [vp, dp, C.POINTER(u64)]
    lib.bchmc_kernel_name.argtypes = [C.c_int]
    lib.bchmc_kernel_name.restype = C.c_char_p
    lib.bchmc_chain_set_state.argtypes = [vp, dp]
    lib.bchmc_chain_get_state.argtypes = [vp, dp]
    lib.bchmc_chain_set_momenta.argtypes = [vp, dp]
    lib.bchmc_chain_get_momenta.argtypes = [vp, dp]
    lib.bchmc_chain_draw_momenta.argtypes = [vp, u64, u64]
    lib.bchmc_chain_attempt.argtypes = [vp, C.c_double, u64, dp, dp, C.POINTER(u64)]
    lib.bchmc_chain_get_proposal.argtypes = [vp, dp, dp]
    lib.bchmc_chain_accept.argtypes = [vp, C.c_int]
    lib.bchmc_measure_spectrum.argtypes = [vp, dp, C.c_uint64, dp, dp]
    lib.bchmc_philox_kat.argtypes = [C.POINTER(C.c_uint32)] * 3
    lib.bchmc_kinetic_term.argtypes = [vp, dp, dp]
    lib.bchmc_psi.argtypes = [vp, dp, dp]
    lib.bchmc_comm_unique_id.argtypes = [C.POINTER(C.c_ubyte)]
    lib.bchmc_comm_create.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.bchmc_comm_create_custom.argtypes = [ALLGATHER_FN, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.bchmc_comm_destroy.argtypes = [vp]
    lib.bchmc_comm_destroy.restype = None
    lib.bchmc_comm_last_error.argtypes = [vp]
    lib.bchmc_comm_last_error.restype = C.c_char_p
    lib.bchmc_eps_exchange.argtypes = [vp, C.POINTER(EpsRecord), C.c_int, C.POINTER(EpsRecord), C.POINTER(C.c_int),
                                       C.c_int, C.POINTER(C.c_int)]
    lib.bchmc_comm_pending.argtypes = [vp]
    lib.bchmc_comm_world.argtypes = [vp]
    lib.bchmc_comm_rank.argtypes = [vp]
    lib.bchmc_comm_transport.argtypes = [vp]
    lib.bchmc_comm_transport.restype = C.c_char_p
    _lib = lib
    return lib


def philox_kat(ctr, key):
    """Philox4x32-10 block on the device (known-answer hook)."""
    lib = load()
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    rc = lib.bchmc_philox_kat(c, k, o)
    if rc:
        raise BchmcError(rc, lib.bchmc_strerror(rc).decode())
    return [int(x) for x in o]


def make_config(params, device=0, precision=0, deterministic=0):
    cfg = BchmcConfig()
    cfg.abi_version = ABI_VERSION
    cfg.Nx = int(params.Nx)
    cfg.L = float(params.L)
    for name, _ in BchmcConfig._fields_:
        if name in ("abi_version", "Nx", "L", "precision", "device", "deterministic", "reserved0"):
            continue
        setattr(cfg, name, getattr(params, name))
    cfg.precision = int(precision)
    cfg.device = int(device)
    cfg.deterministic = int(deterministic)
    return cfg


def _p(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Engine:
    """One chain on one GPU: owns a ``bchmc_handle``."""

    def __init__(self, params, device=0, precision=0, deterministic=0):
        """precision 0: fp64 field arrays (reference DOUBLE_PREC); 1: fp32 field arrays (BASELINE config 5).
        Host arrays are float64 either way.  deterministic 1: bitwise repeatable mass assignment (fixed point)."""
        self.lib = load()
        self.precision = int(precision)
        self.params = params
        self.Nx = int(params.Nx)
        self.N = self.Nx ** 3
        self.h = C.c_void_p()
        cfg = make_config(params, device, precision, deterministic)
        rc = self.lib.bchmc_create(C.byref(cfg), C.byref(self.h))
        if rc:
            detail = self.lib.bchmc_last_error(self.h).decode() if self.h else ""
            text = self.lib.bchmc_strerror(rc).decode()
            self.close()
            raise BchmcError(rc, text, detail)

    def close(self):
        if getattr(self, "h", None):
            self.lib.bchmc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise BchmcError(rc, self.lib.bchmc_strerror(rc).decode(), self.lib.bchmc_last_error(self.h).decode())

    def _in(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        if a.size != self.N:
            raise ValueError("expected %d elements, got %d" % (self.N, a.size))
        return a

    # ---- arrays --------------------------------------------------------------------------------
    def upload(self, **arrays):
        for k, v in arrays.items():
            self._chk(self.lib.bchmc_upload(self.h, FIELDS[k], _p(self._in(v)), self.N))

    def fetch(self, name):
        out = np.empty(self.N)
        self._chk(self.lib.bchmc_fetch(self.h, FIELDS[name], _p(out), self.N))
        return out

    # ---- host-array entry points (what the reference shim binds) ---------------------------------
    def leapfrog(self, q0, p0, eps, neps, out=None):
        """``out=(q1, p1)``: caller-owned result arrays (the reference allocates signalf / momentaf once per sample,
        HMC.cc:375); fresh arrays otherwise."""
        q1, p1 = out if out is not None else (np.empty(self.N), np.empty(self.N))
        done = C.c_uint64()
        self._chk(self.lib.bchmc_leapfrog(self.h, _p(self._in(q0)), _p(self._in(p0)), _p(q1), _p(p1), float(eps),
                                          int(neps), C.byref(done)))
        return q1, p1, done.value

    def leapfrog_dh(self, q0, p0, eps, neps, out=None):
        """Hamiltonian_EoM + delta_Hamiltonian of the same four arrays in one pass (bchmc_leapfrog_dh).
        Returns (q1, p1, steps_done, dH, terms[6])."""
        q1, p1 = out if out is not None else (np.empty(self.N), np.empty(self.N))
        done, dH = C.c_uint64(), C.c_double()
        terms = np.zeros(6)
        self._chk(self.lib.bchmc_leapfrog_dh(self.h, _p(self._in(q0)), _p(self._in(p0)), _p(q1), _p(p1), float(eps),
                                             int(neps), C.byref(done), C.byref(dH), _p(terms)))
        return q1, p1, done.value, dH.value, terms

    def kinetic_term(self, p):
        out = C.c_double()
        self._chk(self.lib.bchmc_kinetic_term(self.h, _p(self._in(p)), C.byref(out)))
        return out.value

    def psi(self, q):
        out = np.zeros(2)
        self._chk(self.lib.bchmc_psi(self.h, _p(self._in(q)), _p(out)))
        return out

    def energies(self, q, p):
        out = np.zeros(3)
        self._chk(self.lib.bchmc_energies(self.h, _p(self._in(q)), _p(self._in(p)), _p(out)))
        return out

    def delta_hamiltonian(self, qi, pi, qf, pf):
        dH = C.c_double()
        terms = np.zeros(6)
        self._chk(self.lib.bchmc_delta_hamiltonian(self.h, _p(self._in(qi)), _p(self._in(pi)), _p(self._in(qf)),
                                                   _p(self._in(pf)), C.byref(dH), _p(terms)))
        return dH.value, terms

    def gradient(self, q):
        g = np.empty(self.N)
        self._chk(self.lib.bchmc_gradient(self.h, _p(self._in(q)), _p(g)))
        return g

    def forward(self, q, rsd=-1):
        self._chk(self.lib.bchmc_forward(self.h, _p(self._in(q)), int(rsd)))

    # ---- device-resident entry points (torch tensors on the engine's device, float64, contiguous) ----
    @staticmethod
    def _dptr(t):
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        return C.c_void_p(t.data_ptr())

    def leapfrog_device(self, q0, p0, q1, p1, eps, neps):
        self._chk(self.lib.bchmc_leapfrog_device(self.h, self._dptr(q0), self._dptr(p0), self._dptr(q1),
                                                 self._dptr(p1), float(eps), int(neps)))

    def steps_done(self):
        v = C.c_uint64()
        self._chk(self.lib.bchmc_steps_done(self.h, C.byref(v)))
        return v.value

    def energies_device(self, q, p):
        out = np.zeros(3)
        self._chk(self.lib.bchmc_energies_device(self.h, self._dptr(q), self._dptr(p), _p(out)))
        return out

    def sync(self):
        self._chk(self.lib.bchmc_sync(self.h))

    @property
    def stream(self):
        return self.lib.bchmc_stream(self.h)

    # ---- device-resident chain ------------------------------------------------
    def chain_set_state(self, q):
        self._chk(self.lib.bchmc_chain_set_state(self.h, _p(self._in(q))))

    def chain_get_state(self):
        q = np.empty(self.N)
        self._chk(self.lib.bchmc_chain_get_state(self.h, _p(q)))
        return q

    def chain_set_momenta(self, p):
        self._chk(self.lib.bchmc_chain_set_momenta(self.h, _p(self._in(p))))

    def chain_get_momenta(self):
        p = np.empty(self.N)
        self._chk(self.lib.bchmc_chain_get_momenta(self.h, _p(p)))
        return p

    def chain_draw_momenta(self, seed, attempt):
        self._chk(self.lib.bchmc_chain_draw_momenta(self.h, int(seed), int(attempt)))

    def chain_attempt(self, eps, neps):
        dH, done = C.c_double(), C.c_uint64()
        terms = np.zeros(6)
        self._chk(self.lib.bchmc_chain_attempt(self.h, float(eps), int(neps), C.byref(dH), _p(terms), C.byref(done)))
        return dH.value, terms, done.value

    def chain_get_proposal(self):
        q1, p1 = np.empty(self.N), np.empty(self.N)
        self._chk(self.lib.bchmc_chain_get_proposal(self.h, _p(q1), _p(p1)))
        return q1, p1

    def chain_accept(self, accepted):
        self._chk(self.lib.bchmc_chain_accept(self.h, int(bool(accepted))))

    def measure_spectrum(self, signal=None, n_bin=200):
        """measure_spectrum (field_statistics.cpp:20-90) of a host field, or of the resident chain state when
        ``signal`` is None (nothing but 2 x n_bin doubles crosses PCIe).  Returns (kmode, power)."""
        kmode, power = np.empty(n_bin), np.empty(n_bin)
        sig = None if signal is None else _p(self._in(signal))
        self._chk(self.lib.bchmc_measure_spectrum(self.h, sig, int(n_bin), _p(kmode), _p(power)))
        return kmode, power

    def tile_info(self):
        out = (C.c_int32 * 8)()
        self._chk(self.lib.bchmc_tile_info(self.h, out))
        keys = ("tiled", "one_pass", "cap", "cap_alloc", "watch", "stage", "unrolled81", "alpt_planes")
        return dict(zip(keys, [int(v) for v in out]))

    # ---- measurement ---------------------------------------------------------------------------
    def profile(self, enable):
        self._chk(self.lib.bchmc_profile(self.h, int(bool(enable))))

    def profile_read(self):
        ms = np.zeros(K_COUNT)
        n = (C.c_uint64 * K_COUNT)()
        self._chk(self.lib.bchmc_profile_read(self.h, _p(ms), n))
        return {self.lib.bchmc_kernel_name(i).decode(): (float(ms[i]), int(n[i])) for i in range(K_COUNT)}


class Comm:
    """The cross-chain record exchange of include/bchmc.h (``bchmc_comm``): RCCL transport (``unique_id`` from rank 0)
    or a custom host all-gather (``allgather(send_bytes) -> bytes of all ranks``; tests, MPI, torch.distributed)."""

    def __init__(self, rank, world, device=0, unique_id=None, allgather=None):
        self.lib = load()
        self.rank, self.world = int(rank), int(world)
        self.h = C.c_void_p()
        self._cb = None
        if allgather is not None or world == 1 and unique_id is None:
            def _fn(_ctx, send, recv, nbytes):
                try:
                    out = allgather(C.string_at(send, nbytes))
                    if len(out) != nbytes * self.world:
                        return 2
                    C.memmove(recv, out, len(out))
                    return 0
                except Exception:  # a Python exception must not unwind through the C caller
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = ALLGATHER_FN(_fn) if allgather is not None else C.cast(None, ALLGATHER_FN)
            rc = self.lib.bchmc_comm_create_custom(self._cb, None, self.rank, self.world, C.byref(self.h))
        else:
            buf = (C.c_ubyte * UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
            rc = self.lib.bchmc_comm_create(buf, self.rank, self.world, int(device), C.byref(self.h))
        if rc:
            detail = self.lib.bchmc_comm_last_error(self.h).decode() if self.h else ""
            self.close()
            raise BchmcError(rc, self.lib.bchmc_strerror(rc).decode(), detail)

    @staticmethod
    def unique_id():
        lib = load()
        buf = (C.c_ubyte * UNIQUE_ID_BYTES)()
        rc = lib.bchmc_comm_unique_id(buf)
        if rc:
            raise BchmcError(rc, lib.bchmc_strerror(rc).decode(), "bchmc_comm_unique_id (is librccl loadable?)")
        return bytes(buf)

    def exchange(self, records):
        """records: list of (epsilon, accepted, neps) of this rank's finished attempts (may be empty).
        Returns [(rank, epsilon, accepted, neps), ...] of every rank's contribution to this exchange."""
        n = len(records)
        mine = (EpsRecord * max(n, 1))()
        for i, (eps, acc, neps) in enumerate(records):
            mine[i].epsilon, mine[i].accepted, mine[i].neps = float(eps), int(bool(acc)), int(neps)
        cap = self.world * EPS_BATCH
        out, who, got = (EpsRecord * cap)(), (C.c_int * cap)(), C.c_int(0)
        rc = self.lib.bchmc_eps_exchange(self.h, mine, n, out, who, cap, C.byref(got))
        if rc:
            raise BchmcError(rc, self.lib.bchmc_strerror(rc).decode(), self.lib.bchmc_comm_last_error(self.h).decode())
        return [(int(who[i]), float(out[i].epsilon), bool(out[i].accepted), int(out[i].neps)) for i in range(got.value)]

    def pending(self):
        return int(self.lib.bchmc_comm_pending(self.h))

    def info(self):
        """What the communicator itself reports (not what the caller asked for): world, rank, transport."""
        return dict(world=int(self.lib.bchmc_comm_world(self.h)), rank=int(self.lib.bchmc_comm_rank(self.h)),
                    transport=self.lib.bchmc_comm_transport(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.bchmc_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
