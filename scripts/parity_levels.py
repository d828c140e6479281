#!/usr/bin/env python3
"""Print the actual error levels of the engine against the oracle (16^3 and 32^3 cases) next to the tolerances."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import Case, rel_l2  # noqa: E402

for nx, kw in [(16, dict(likelihood=1, rsd_model=1)), (16, dict(likelihood=0, rsd_model=0)), (32, dict(likelihood=1, rsd_model=1))]:
    c = Case(Nx=nx, **kw)
    e = c.engine()
    g, gp, gl = c.oracle.gradient_psi(c.q0)
    gg = e.gradient(c.q0)
    rho = c.oracle.getDensity(3, *[c.oracle.get(k) for k in ("posx", "posy", "posz")])
    pl = c.oracle.partial_f_delta_x_log_like(c.oracle.get("deltaX"))
    V = c.oracle.likelihood_calc_V_SPH(pl, *[c.oracle.get(k) for k in ("posx", "posy", "posz")])
    print("n=%d %s: rho %.1e (tol 1e-12)  V %.1e (1e-11)  grad %.1e (1e-11)" %
          (nx, kw, rel_l2(e.fetch("rho"), rho), rel_l2(e.fetch("Vx"), V[0]), rel_l2(gg, g)), end="")
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 10)
    q1, p1, _ = e.leapfrog(c.q0, c.p0, c.eps, 10)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    dH, t = e.delta_hamiltonian(c.q0, c.p0, q1o, p1o)
    print("  traj10 q %.1e p %.1e (1e-11)  energies %.1e (1e-10)" %
          (rel_l2(q1, q1o), rel_l2(p1, p1o), float(np.max(np.abs(t - to) / np.abs(to)))))
    e.close()
