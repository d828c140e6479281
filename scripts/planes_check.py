#!/usr/bin/env python3
"""One-off check at sizes the oracle cannot reach: planes mode (k_step_boundary_x) against the 3-D-plan path.
    python scripts/planes_check.py 512 [fp32]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcode_amd import inputs  # noqa: E402
from barcode_amd.engine import Engine  # noqa: E402
from barcode_amd.params import HamilParams  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = 1 if len(sys.argv) > 2 and sys.argv[2] == "fp32" else 0
p = HamilParams(Nx=nx, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
f = inputs.make_fields(p)
out = []
obs = None
for no_planes in ("0", "1"):
    os.environ["BCHMC_NO_PLANES"] = no_planes
    e = Engine(p, precision=prec)
    e.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], nobs=np.zeros(p.N), window=np.ones(p.N), noise=np.ones(p.N))
    if obs is None:
        e.forward(f["truth"], 1)
        obs = inputs.mock_observations(p, e.fetch("deltaX").reshape((nx,) * 3), delta_lag=f["truth"])
    e.upload(window=obs[0], noise=obs[1], nobs=obs[2])
    q1, p1, done = e.leapfrog(f["q0"], f["p0"], 0.5 * p.eps_heuristic(), 4)
    out.append((q1, p1, done))
    e.close()
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print("n=%d %s: steps %d/%d, rel-L2 planes vs 3-D plans: q %.2e p %.2e" %
      (nx, "fp32" if prec else "fp64", out[0][2], out[1][2], rel(out[0][0], out[1][0]), rel(out[0][1], out[1][1])))
