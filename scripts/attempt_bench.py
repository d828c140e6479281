#!/usr/bin/env python3
"""Time one HamiltonianMC attempt (momenta + trajectory + delta_Hamiltonian) two ways on the GPU box:
  host-array path:  bchmc_leapfrog_dh with numpy arrays (what the reference shim binds: Hamiltonian_EoM keeps the six
                    terms for the delta_Hamiltonian that follows), and bchmc_leapfrog + bchmc_delta_hamiltonian (the
                    latter always evaluates) for comparison,
  resident chain:   bchmc_chain_draw_momenta + bchmc_chain_attempt + bchmc_chain_accept.
Usage: python scripts/attempt_bench.py [nx] [neps]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from barcode_amd import inputs  # noqa: E402
from barcode_amd.engine import Engine  # noqa: E402
from barcode_amd.params import HamilParams  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
neps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = HamilParams(Nx=nx, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
f = inputs.make_fields(p)
e = Engine(p)
e.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], nobs=np.zeros(p.N), window=np.ones(p.N), noise=np.ones(p.N))
e.forward(f["truth"], 1)
w, s, nobs = inputs.mock_observations(p, e.fetch("deltaX").reshape((nx,) * 3))
e.upload(window=w, noise=s, nobs=nobs)
eps = 0.5 * p.eps_heuristic()
q0, p0 = f["q0"].ravel().copy(), f["p0"].ravel().copy()
q1, p1 = np.zeros(p.N), np.zeros(p.N)   # signalf / momentaf: allocated once per sample upstream (HMC.cc:375)


def host_path(one_pass=True):
    if one_pass:
        return e.leapfrog_dh(q0, p0, eps, neps, out=(q1, p1))[3]
    e.leapfrog(q0, p0, eps, neps, out=(q1, p1))
    return e.delta_hamiltonian(q0, p0, q1, p1)[0]


def chain_path(i, accept=False):
    e.chain_draw_momenta(1, i)
    dH, _, _ = e.chain_attempt(eps, neps)
    e.chain_accept(accept)
    return dH


host_path()
e.chain_set_state(q0)
chain_path(0)
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    host_path()
t1 = time.perf_counter()
for i in range(reps):
    chain_path(i + 1)
t2 = time.perf_counter()
import os  # noqa: E402
# the same chain with every proposal accepted (the carried gradient is the proposal's), and without carrying it
for i in range(reps):
    chain_path(100 + i, True)
t2a = time.perf_counter()
os.environ["BCHMC_NO_FORCE_CARRY"] = "1"
chain_path(200)
t2b = time.perf_counter()
for i in range(reps):
    chain_path(201 + i, True)
t2c = time.perf_counter()
host_path(False)                          # the r01 protocol: plain trajectory, then two full energy evaluations
t3 = time.perf_counter()
for _ in range(reps):
    host_path(False)
t4 = time.perf_counter()
del os.environ["BCHMC_NO_FORCE_CARRY"]


def timed_host(**env):
    os.environ.update(env)
    host_path()
    ta = time.perf_counter()
    for _ in range(reps):
        host_path()
    tb = time.perf_counter()
    for k in env:
        del os.environ[k]
    return 1e3 * (tb - ta) / reps


# the two transfers that run beside compute, switched off one at a time on the same box (read per call)
no_down = timed_host(BCHMC_NO_DOWNLOAD_OVERLAP="1")
no_both = timed_host(BCHMC_NO_DOWNLOAD_OVERLAP="1", BCHMC_NO_UPLOAD_OVERLAP="1")
again = timed_host()
print(json.dumps(dict(grid=nx, neps=neps, host_array_ms_per_attempt=1e3 * (t1 - t0) / reps,
                      host_array_again_ms=again, host_array_without_download_overlap_ms=no_down,
                      host_array_without_either_overlap_ms=no_both,
                      resident_chain_ms_per_attempt=1e3 * (t2 - t1) / reps,
                      resident_chain_all_accepted_ms=1e3 * (t2a - t2) / reps,
                      resident_chain_gradient_recomputed_ms=1e3 * (t2c - t2b) / reps,
                      host_array_without_trajectory_reuse_ms=1e3 * (t4 - t3) / reps,
                      note="host path = bchmc_leapfrog_dh on caller arrays (pinned staging, energies taken from the "
                           "trajectory's own pass; bchmc_delta_hamiltonian itself always evaluates); excludes the host-side momentum draw the "
                           "reference does per attempt")))
