"""Step-size adaptation of the reference, restated for the host side of the resident chain.

``update_eps_fac`` and its helpers follow ``barlib/src/hmc/leapfrog/time_step.cpp:40-185`` and the templates of
``barlib/include/hmc/leapfrog/time_step.hpp:23-75`` statement by statement; the tables they read are the
``EpsRing`` of ``barcode_amd.chains`` (optionally filled with every chain's records through the RCCL all-gather).
O(100) host work per call: nothing here touches the GPU.
"""
from dataclasses import dataclass

import numpy as np


def power_mean(x, y, p):
    """math_funcs.cc:36-44."""
    if p == 0:
        return float(np.sqrt(x * y))
    return float(((x ** p + y ** p) / 2.0) ** (1.0 / p))


def sort_vector_by_other(sortee, other):
    """time_step.hpp:36-49 (std::sort is not stable; ties in ``other`` have unspecified order upstream)."""
    idx = np.argsort(np.asarray(other), kind="stable")
    return np.asarray(sortee)[idx]


def cumulative_moving_average(a):
    """time_step.hpp:51-61: running mean of the first 1, 2, ... elements."""
    a = np.asarray(a, dtype=np.float64)
    return np.cumsum(a) / np.arange(1, a.size + 1)


def stl_smooth(a, smooth_size):
    """time_step.hpp:63-75: boxcar mean over [i - s, i + s], clipped at the ends."""
    a = np.asarray(a, dtype=np.float64)
    out = np.empty_like(a)
    for i in range(a.size):
        lo, hi = max(i - smooth_size, 0), min(i + smooth_size + 1, a.size)
        out[i] = a[lo:hi].sum() / (hi - lo)
    return out


@dataclass
class EpsConfig:
    """input.par keys of the step-size schemes (data/input.par:59-87) and their defaults."""
    eps_fac_update_type: int = 3
    N_a_eps_update: int = 100
    acc_min: float = 0.6
    acc_max: float = 0.7
    eps_down_smooth: int = 5
    eps_up_fac: float = 1.0
    eps_fac_target: float = 0.0
    eps_fac_power: float = 2.0
    s_eps_total: int = 10


def update_eps_fac_acceptance_rate_downwards(eps_fac, ring, cfg):
    """time_step.cpp:40-104.  Returns the new eps_fac."""
    alpha = ring.acceptance_rate()
    acc_target = (cfg.acc_max + cfg.acc_min) / 2.0
    a_sort = sort_vector_by_other(ring.acc_flag, ring.epsilon).astype(np.float64)
    a_sm = stl_smooth(cumulative_moving_average(a_sort), cfg.eps_down_smooth)
    ix_max = int(np.argmax(a_sm))  # std::max_element: first of the largest
    if a_sm[ix_max] > acc_target:
        below = np.nonzero(a_sm[ix_max:] < acc_target)[0]
        if below.size:  # else: "special" case, eps_fac stays
            eps_fac = float(np.sort(ring.epsilon)[ix_max + int(below[0])])
    else:
        if alpha == 0.0:
            eps_fac = float(ring.epsilon.min())
        else:
            eps_fac = eps_fac / 3.0
    if eps_fac == 0.0:
        raise RuntimeError("In update_eps_fac_acceptance_rate_downwards: epsilon became zero, shouldn't happen!")
    return eps_fac


def update_eps_fac_acceptance_rate(eps_fac, ring, cfg, due=None):
    """time_step.cpp:106-135.  ``due``: the "every N_a attempts" trigger, already evaluated by the caller (crossing
    form, see EpsRing.crossed); None evaluates the reference's equality (single chain)."""
    if due is None:
        due = ring.count_attempts % cfg.N_a_eps_update == 0 and ring.count_attempts > 0
    if due:
        alpha = ring.acceptance_rate()
        if alpha < cfg.acc_min:
            eps_fac = update_eps_fac_acceptance_rate_downwards(eps_fac, ring, cfg)
        elif alpha > cfg.acc_max:
            acc_target = (cfg.acc_max + cfg.acc_min) / 2.0
            eps_fac = eps_fac * cfg.eps_up_fac * (alpha / acc_target)
    return eps_fac


def update_eps_fac(eps_fac, ring, cfg, iGibbs=2, rejections=0):
    """time_step.cpp:151-185, called before every trajectory (HMC.cc:453).  ``iGibbs`` / ``rejections`` are
    HAMIL_NUMERICAL's (used by scheme 3 only).  The "every so many attempts" triggers use the ring's crossing test, so
    that pooled records of other chains (which advance the count by several between calls) cannot step over them;
    for a single chain it is the reference's ``count_attempts % n == 0 && count_attempts > 0``."""
    t = cfg.eps_fac_update_type
    every = cfg.s_eps_total if t == 1 else cfg.N_a_eps_update
    due = ring.crossed(every)  # consumed on every call, also in the fast initial phase (which skips the test upstream)
    if t == 1:
        if due:
            eps_fac = power_mean(eps_fac, cfg.eps_fac_target, cfg.eps_fac_power)
    elif t == 2:
        eps_fac = update_eps_fac_acceptance_rate(eps_fac, ring, cfg, due)
    elif t == 3:
        if iGibbs == 1 and rejections > 0:  # fast initial phase, time_step.cpp:137-149
            eps_fac = eps_fac / 2.0
        else:
            eps_fac = update_eps_fac_acceptance_rate(eps_fac, ring, cfg, due)
    return eps_fac
