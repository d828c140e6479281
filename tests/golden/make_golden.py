#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the CPU oracle (serial build, deterministic).

PARITY UNPINNED: these vectors come from OUR restatement of the reference (oracle/bchmc_oracle.c), not from
the reference itself -- it ships no golden vectors for this path and cannot be built in this image (FFTW3
and GSL absent).  They pin the oracle against regressions and give the GPU path a fixed target.

    python tests/golden/make_golden.py [name ...]        (no names: all of them)

Each <name>.npz holds the scalar parameters, every input array and the outputs of one force evaluation, one
trajectory (forced Neps, epsilon: SURVEY M5) and delta_Hamiltonian.
"""
import dataclasses
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.util import Case  # noqa: E402

CASES = {
    "gauss_zeld_8": dict(Nx=8, likelihood=1, rsd_model=0),
    "gauss_rsd_16": dict(Nx=16, likelihood=1, rsd_model=1, sfmodel=2),
    "poisson_zeld_16": dict(Nx=16, likelihood=0, rsd_model=0),
    "lognormal_zeld_8": dict(Nx=8, likelihood=2, rsd_model=0),
    "grf_8": dict(Nx=8, likelihood=3, rsd_model=0),
    "gauss_mass5_8": dict(Nx=8, likelihood=1, rsd_model=0, mass_type=5),
    "gauss_calch3_rsd_8": dict(Nx=8, likelihood=1, rsd_model=1, calc_h=3),
    "gauss_cic_calch1_8": dict(Nx=8, likelihood=1, rsd_model=0, calc_h=1, mk=1),
    "gauss_alpt_8": dict(Nx=8, likelihood=1, rsd_model=0, sfmodel=2),  # ALPT forward model
    "poisson_calch3_8": dict(Nx=8, likelihood=0, rsd_model=0, calc_h=3),
    "poisson_ngp_calch1_8": dict(Nx=8, likelihood=0, rsd_model=0, calc_h=1, mk=0),
    "gauss_tsc_calch1_rsd_8": dict(Nx=8, likelihood=1, rsd_model=1, calc_h=1, mk=2),
}
NEPS = 10


def make(name, kw):
    c = Case(**kw)
    o = c.oracle
    out = dict(params=json.dumps(dataclasses.asdict(c.p)), neps=NEPS, eps=c.eps,
               signal_PS=c.signal_PS, mass_f=c.mass_f, mass_r=c.mass_r, window=c.window, noise=c.noise, nobs=c.nobs,
               q0=c.q0, p0=c.p0)
    g, gp, gl = o.gradient_psi(c.q0)
    out.update(gradpsi=g, grad_prior=gp, grad_like=gl)
    if c.p.likelihood != 3:
        dX = o.get("deltaX")
        pos = [o.get(k) for k in ("posx", "posy", "posz")]
        pl = o.partial_f_delta_x_log_like(dX)
        out.update(deltaX=dX, posx=pos[0], posy=pos[1], posz=pos[2], part_like=pl)
        if c.p.calc_h == 2:
            V = o.likelihood_calc_V_SPH(pl, *pos)
            out.update(Vx=V[0], Vy=V[1], Vz=V[2])
        elif c.p.calc_h == 3:
            V = o.likelihood_calc_V_SPH_fourier_TSC(pl, *pos)
            out.update(Vx=V[0], Vy=V[1], Vz=V[2])
    q1, p1, done = o.Hamiltonian_EoM(c.q0, c.p0, c.eps, NEPS)
    dH, terms = o.delta_Hamiltonian(c.q0, c.p0, q1, p1)
    out.update(q1=q1, p1=p1, steps_done=done, dH=dH, energy_terms=terms)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "dH = %.6e" % dH, "terms", terms)


if __name__ == "__main__":
    for name, kw in CASES.items():
        if len(sys.argv) == 1 or name in sys.argv[1:]:
            make(name, kw)
