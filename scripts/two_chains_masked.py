#!/usr/bin/env python3
"""Experiment: two chains on ONE GPU, each restricted to half of the CUs (BCHMC_CU_MASK), so that the VALU-bound
particle-mesh kernels of one chain run next to the HBM-bound transforms of the other.
    python scripts/two_chains_masked.py [lo,hi | even,odd | evencu,oddcu | none]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcode_amd import inputs  # noqa: E402
from barcode_amd.engine import Engine  # noqa: E402
from barcode_amd.params import HamilParams  # noqa: E402

modes = (sys.argv[1] if len(sys.argv) > 1 else "lo,hi").split(",")
steps = 60
dev = torch.device("cuda", 0)
p = HamilParams(Nx=256, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
f = inputs.make_fields(p)
engines, obs = [], None
for m in modes:
    if m == "none":
        os.environ.pop("BCHMC_CU_MASK", None)
    else:
        os.environ["BCHMC_CU_MASK"] = m
    e = Engine(p, device=0)
    e.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], nobs=np.zeros(p.N), window=np.ones(p.N), noise=np.ones(p.N))
    if obs is None:
        e.forward(f["truth"], 1)
        obs = inputs.mock_observations(p, e.fetch("deltaX").reshape((p.Nx,) * 3), delta_lag=f["truth"])
    e.upload(window=obs[0], noise=obs[1], nobs=obs[2])
    engines.append(e)
eps = 0.5 * p.eps_heuristic()
q0 = torch.from_numpy(f["q0"].reshape(-1)).to(dev)
states = []
for c in range(len(engines)):
    p0 = torch.from_numpy(inputs.gaussian_random_field(p, f["mass_f"], inputs.SEED_P0 + c).reshape(-1)).to(dev)
    states.append((q0.clone(), p0, torch.empty_like(q0), torch.empty_like(q0)))


def run(active):
    for i in active:
        engines[i].leapfrog_device(*states[i], eps, 3)
    for i in active:
        engines[i].sync()
    t0 = time.perf_counter()
    for i in active:
        engines[i].leapfrog_device(*states[i], eps, steps)
    for i in active:
        engines[i].sync()
    return len(active) * steps / (time.perf_counter() - t0)


for i in range(len(engines)):
    print("chain %d alone (mask %s): %.1f steps/s" % (i, modes[i], run([i])))
print("both together: %.1f steps/s aggregate" % run(list(range(len(engines)))))
