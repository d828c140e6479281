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
    signalf, momentaf, done = hd.engine.leapfrog(signali, momentai, n.epsilon, n.Neps)
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
