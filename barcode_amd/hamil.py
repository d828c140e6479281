"""Host-side mirror of the reference's interface for the leapfrog path, on top of the C ABI.

Same names, argument meaning and bookkeeping as ``barlib/src/HMC.cc`` so that callers (and tests) read
like the reference: ``Hamiltonian_EoM``, ``delta_Hamiltonian``, ``gradient_psi``, ``kinetic_term``, ``psi``
operate on a ``HamilData`` (the reference's ``HAMIL_DATA`` + ``HAMIL_NUMERICAL``).  The arithmetic all
happens in ``libbarcode_hip.so``; this layer only draws (Neps, epsilon), forwards arrays and stores the
scalars the reference keeps for its performance log (``HMC.cc:40-60``).
"""
from dataclasses import dataclass

import numpy as np

from .engine import Engine


@dataclass
class HamilNumerical:
    """The per-attempt scalars of HAMIL_NUMERICAL (struct_hamil.h:86-92, 101-105)."""
    N_eps_fac: float = 8.0
    eps_fac: float = 0.0
    Neps: int = 0
    epsilon: float = 0.0
    accepted: bool = False
    dH: float = 0.0
    dK: float = 0.0
    dE: float = 0.0
    dprior: float = 0.0
    dlikeli: float = 0.0
    psi_prior: float = 0.0
    psi_likeli: float = 0.0
    psi_prior_i: float = 0.0
    psi_prior_f: float = 0.0
    psi_likeli_i: float = 0.0
    psi_likeli_f: float = 0.0
    H_kin_i: float = 0.0
    H_kin_f: float = 0.0
    steps_done: int = 0
    count_attempts: int = 0
    iGibbs: int = 1        # sample number (struct_hamil.h:106): scheme 3's fast initial phase runs while it is 1
    rejections: int = 0    # rejected attempts of the current sample (HMC.cc:500-501)


class HamilData:
    """HAMIL_DATA: parameters + input arrays bound to one engine handle (one chain, one GPU)."""

    def __init__(self, params, device=0, N_eps_fac=8.0, eps_fac=None, **arrays):
        self.params = params
        self.engine = Engine(params, device=device)
        self.numerical = HamilNumerical(N_eps_fac=N_eps_fac,
                                        eps_fac=params.eps_heuristic() if eps_fac is None else eps_fac)
        self.gradpsi = None
        if arrays:
            self.engine.upload(**arrays)

    # hd->deltaX / hd->pos*: state of the last force or energy evaluation
    @property
    def deltaX(self):
        return self.engine.fetch("deltaX")

    def pos(self):
        return tuple(self.engine.fetch(k) for k in ("posx", "posy", "posz"))


def Hamiltonian_EoM(hd, signali, momentai, uniform):
    """HMC.cc:251-369.  ``uniform`` stands for ``gsl_rng_uniform(seed)``; it is called exactly twice, in the
    reference's order (Neps first, then epsilon; HMC.cc:260-261)."""
    n = hd.numerical
    n.Neps = int(n.N_eps_fac * uniform()) + 1
    n.epsilon = float(n.eps_fac * uniform())
    if n.epsilon > 2.0:
        n.epsilon = 2.0
    # one pass: the trajectory and the six energy terms of its four arrays (bchmc_leapfrog_dh), kept for the
    # delta_Hamiltonian that HamiltonianMC calls next about the same arrays (HMC.cc:455-459; the C++ shim does the same)
    signalf, momentaf, done, dH, t = hd.engine.leapfrog_dh(signali, momentai, n.epsilon, n.Neps)
    hd._eom = ((signali, momentai, signalf, momentaf), dH, t)
    n.steps_done = done
    n.count_attempts += 1  # data->numerical->count_attempts++ (HMC.cc:368)
    return signalf, momentaf


def gradient_psi(hd, signal):
    """HMC.cc:146-206: fills hd.gradpsi."""
    hd.gradpsi = hd.engine.gradient(signal)
    return hd.gradpsi


def kinetic_term(hd, momenta, signal=None):
    """HMC.cc:64-121.  The engine evaluates all three energy terms in one call; ``signal`` defaults to zeros."""
    q = np.zeros(hd.engine.N) if signal is None else signal
    return float(hd.engine.energies(q, momenta)[0])


def psi(hd, signal, momenta=None):
    """HMC.cc:124-143: returns psi_prior + psi_likelihood and stores both in hd.numerical."""
    p = np.zeros(hd.engine.N) if momenta is None else momenta
    e = hd.engine.energies(signal, p)
    hd.numerical.psi_prior, hd.numerical.psi_likeli = float(e[1]), float(e[2])
    return float(e[1] + e[2])


def delta_Hamiltonian(hd, signali, momentai, signalf, momentaf):
    """HMC.cc:209-248, including the performance-log bookkeeping."""
    n = hd.numerical
    kept, hd._eom = getattr(hd, "_eom", None), None  # one use, only as the next call about the very same arrays
    if kept is not None and all(a is b for a, b in zip(kept[0], (signali, momentai, signalf, momentaf))):
        dH, t = kept[1], kept[2]
    else:
        dH, t = hd.engine.delta_hamiltonian(signali, momentai, signalf, momentaf)
    n.H_kin_i, n.psi_prior_i, n.psi_likeli_i = (float(x) for x in t[:3])
    n.H_kin_f, n.psi_prior_f, n.psi_likeli_f = (float(x) for x in t[3:])
    n.psi_prior, n.psi_likeli = n.psi_prior_f, n.psi_likeli_f
    n.dprior = n.psi_prior_f - n.psi_prior_i
    n.dlikeli = n.psi_likeli_f - n.psi_likeli_i
    n.dK = n.H_kin_f - n.H_kin_i
    n.dE = (n.psi_prior_f + n.psi_likeli_f) - (n.psi_prior_i + n.psi_likeli_i)
    n.dH = float(dH)
    return n.dH


def measure_spectrum(hd, signal=None, N_bin=200):
    """field_statistics.cpp:20-90 -> (kmode, power).  ``signal`` None = the resident chain state (what
    barcoderunner.cc:530-533 measures after every sample, without moving the field off the device)."""
    return hd.engine.measure_spectrum(signal, N_bin)


def HamiltonianMC(hd, uniform, seed=1, itmax=2000, ring=None, group=None, momenta=None, eps_cfg=None):
    """One sample of the reference's HamiltonianMC loop (HMC.cc:431-511) on the device-resident chain:
    repeat { draw momenta; draw (Neps, epsilon); trajectory; dH; Metropolis test } until accepted.

    ``uniform`` stands for ``gsl_rng_uniform`` and is consumed in the reference's order per attempt: Neps, epsilon
    (HMC.cc:260-261), then the acceptance draw only if p_acc < 1 (HMC.cc:478-480).  Momenta come from the engine's
    counter-based device generator (``seed``, attempt index) unless ``momenta`` (a callable returning a host
    array, e.g. a port of the GSL draw) is given.  The chain state must have been set with
    ``hd.engine.chain_set_state``.  Returns the list of per-attempt records (the performance-log row, HMC.cc:40-60).

    Step-size bookkeeping as upstream: ``update_eps_fac`` before every trajectory (HMC.cc:453; needs ``ring`` and
    ``eps_cfg``, a ``time_step.EpsConfig``), ``rejections`` += 1 on a reject (500-501), the attempt into the ring
    (``update_epsilon_acc_rate_tables``, 506-507).  With a ``group``, ONE exchange after the loop -- the fixed point
    every chain reaches once per sample -- pools the other chains' records into the ring.
    """
    from . import time_step
    n = hd.numerical
    e = hd.engine
    log = []
    for _ in range(itmax):
        attempt = n.count_attempts
        if momenta is None:
            e.chain_draw_momenta(seed, attempt)
        else:
            e.chain_set_momenta(momenta())
        if ring is not None and eps_cfg is not None:
            n.eps_fac = time_step.update_eps_fac(n.eps_fac, ring, eps_cfg, iGibbs=n.iGibbs, rejections=n.rejections)
        n.Neps = int(n.N_eps_fac * uniform()) + 1
        n.epsilon = float(n.eps_fac * uniform())
        if n.epsilon > 2.0:
            n.epsilon = 2.0
        dH, t, done = e.chain_attempt(n.epsilon, n.Neps)
        n.count_attempts += 1
        n.steps_done = done
        n.H_kin_i, n.psi_prior_i, n.psi_likeli_i, n.H_kin_f, n.psi_prior_f, n.psi_likeli_f = (float(x) for x in t)
        n.dprior, n.dlikeli = n.psi_prior_f - n.psi_prior_i, n.psi_likeli_f - n.psi_likeli_i
        n.dK = n.H_kin_f - n.H_kin_i
        n.dE = n.dprior + n.dlikeli
        n.dH = float(dH)
        # HMC.cc:462-486, statement for statement: a uniform is consumed only when p_acceptance < 1.  (A NaN dH
        # leaves p_acceptance at 1 there; the reference build traps FE_INVALID before that, main.cc:66-78.)
        p_acceptance = 1.0
        if dH < 0.0:
            p_acceptance = 1.0
        elif np.exp(-dH) < 1.0:
            p_acceptance = float(np.exp(-dH))
        if p_acceptance >= 1.0:
            accepted = True
        else:
            accepted = uniform() < p_acceptance
        n.accepted = bool(accepted)
        e.chain_accept(accepted)
        if not accepted:
            n.rejections += 1
        if ring is not None:
            ring.record(accepted, n.epsilon)
        log.append(dict(accepted=n.accepted, epsilon=n.epsilon, Neps=n.Neps, dH=n.dH, dK=n.dK, dE=n.dE,
                        dprior=n.dprior, dlikeli=n.dlikeli, psi_prior_i=n.psi_prior_i, psi_prior_f=n.psi_prior_f,
                        psi_likeli_i=n.psi_likeli_i, psi_likeli_f=n.psi_likeli_f, H_kin_i=n.H_kin_i,
                        H_kin_f=n.H_kin_f, steps_done=done, eps_fac=n.eps_fac))
        if accepted:
            break
    if group is not None and ring is not None:
        group.pool_into(ring, [(r["epsilon"], r["accepted"], r["Neps"]) for r in log])
    return log
