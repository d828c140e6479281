#!/usr/bin/env python3
"""BASELINE config 3 at its full length: 256^3, Zel'dovich + plane-parallel RSD, Gaussian likelihood, fp64, ONE
trajectory of 100 leapfrog steps through the C ABI against the oracle (OpenMP build on the box's host cores: a few
seconds per step, hence a script and a committed result -- profiles/r03_parity_config3_100steps.json -- not a test;
the suite checks 8 steps + energies at this size and 100 steps at 32^3).  Tolerance of SURVEY 8d at 100 steps: 1e-9.
Usage: python scripts/parity_config3_full.py [steps] > out.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from barcode_amd import inputs  # noqa: E402
from barcode_amd.engine import Engine  # noqa: E402
from barcode_amd.params import HamilParams  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
os.environ.setdefault("OMP_NUM_THREADS", str(inputs.host_cpu_share()))
p = HamilParams(Nx=256, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
f = inputs.make_fields(p)
e = Engine(p)
e.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], nobs=np.zeros(p.N), window=np.ones(p.N), noise=np.ones(p.N))
e.forward(f["truth"], 1)
dX = e.fetch("deltaX").reshape((p.Nx,) * 3)
window, noise, nobs = inputs.mock_observations(p, dX)
e.upload(window=window, noise=noise, nobs=nobs)
o = Oracle(p, omp=True)
o.set(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)
eps = 0.5 * p.eps_heuristic()  # the bench's step size


def rel(a, b):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


t0 = time.perf_counter()
q1, p1, done, dH, t = e.leapfrog_dh(f["q0"], f["p0"], eps, steps)
t1 = time.perf_counter()
print("engine: %d steps in %.2f s" % (done, t1 - t0), file=sys.stderr, flush=True)
q1o, p1o, done_o = o.Hamiltonian_EoM(f["q0"], f["p0"], eps, steps)
t2 = time.perf_counter()
print("oracle: %d steps in %.1f s" % (done_o, t2 - t1), file=sys.stderr, flush=True)
dHo, to = o.delta_Hamiltonian(f["q0"], f["p0"], q1o, p1o)
# conditioning of the trajectory itself: the oracle again from a start state perturbed by 1e-13
q1c, p1c, _ = o.Hamiltonian_EoM(f["q0"] * (1.0 + 1e-13), f["p0"], eps, steps) if steps <= 100 else (q1o, p1o, 0)
out = dict(grid=256, steps=steps, steps_done=[int(done), int(done_o)], eps=eps,
           rel_l2_q=rel(q1, q1o), rel_l2_p=rel(p1, p1o),
           energies_engine=[float(x) for x in t], energies_oracle=[float(x) for x in to],
           energies_max_rel=float(np.max(np.abs(np.asarray(t) - to) / np.abs(to))), dH_engine=float(dH), dH_oracle=float(dHo),
           oracle_sensitivity_to_a_1e13th_of_q0=dict(rel_l2_q=rel(q1c, q1o), rel_l2_p=rel(p1c, p1o)),
           tolerance=1e-9, ok=bool(rel(q1, q1o) < 1e-9 and rel(p1, p1o) < 1e-9 and done == done_o == steps),
           oracle="oracle/liboracle_omp.so (parity unpinned: no reference vectors exist)",
           oracle_seconds=round(t2 - t1, 1))
print(json.dumps(out, indent=1))
o.close()
e.close()
