#!/usr/bin/env python3
"""One-off robustness matrix: every configuration of tests/test_gpu_parity.py::CONFIGS at 32^3 with the padded
layout (so that planes mode, the fused boundary and the one-pass binning are all active where they apply) against
the oracle: gradient, 6-step trajectory, energies, resident-chain attempt.
    BCHMC_FFT_PAD=1 python scripts/matrix_check.py"""
import os
import sys

import numpy as np

os.environ.setdefault("BCHMC_FFT_PAD", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import CONFIGS  # noqa: E402
from tests.util import Case, rel_l2  # noqa: E402

worst = 0.0
for i, kw in enumerate(CONFIGS):
    c = Case(Nx=32, **kw)
    e = c.engine()
    g, _, _ = c.oracle.gradient_psi(c.q0)
    eg = rel_l2(e.gradient(c.q0), g)
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 6)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 6)
    et = max(rel_l2(q1, q1o), rel_l2(p1, p1o))
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    e.chain_set_state(c.q0)
    e.chain_set_momenta(c.p0)
    dH, t, _ = e.chain_attempt(c.eps, 6)
    ee = float(np.max(np.abs(t - to)) / np.abs(to).max())
    worst = max(worst, eg, et, ee)
    print("cfg%-2d grad %.1e  traj %.1e  attempt energies %.1e  %s" % (i, eg, et, ee, kw))
    e.close()
print("worst", worst)
assert worst < 1e-9
