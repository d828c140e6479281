#!/usr/bin/env python3
"""Soak run on one GPU: a long trajectory at 256^3 and a HamiltonianMC loop at 128^3 on the resident chain.
Checks: finite state, no early stop, device memory stable, acceptance bookkeeping consistent."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcode_amd import hamil, inputs  # noqa: E402
from barcode_amd.chains import EpsRing  # noqa: E402
from barcode_amd.engine import Engine  # noqa: E402
from barcode_amd.params import HamilParams  # noqa: E402


def mock(p, e, f):
    e.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], nobs=np.zeros(p.N), window=np.ones(p.N), noise=np.ones(p.N))
    e.forward(f["truth"], p.rsd_model)
    dX = e.fetch("deltaX").reshape((p.Nx,) * 3)
    window, noise, nobs = inputs.mock_observations(p, dX, delta_lag=f["truth"])
    e.upload(window=window, noise=noise, nobs=nobs)
    return dict(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)


def main():
    dev = torch.device("cuda", 0)
    # ---- long trajectory
    p = HamilParams(Nx=256, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
    f = inputs.make_fields(p)
    e = Engine(p)
    mock(p, e, f)
    q0 = torch.from_numpy(f["q0"].reshape(-1)).to(dev)
    p0 = torch.from_numpy(f["p0"].reshape(-1)).to(dev)
    q1, p1 = torch.empty_like(q0), torch.empty_like(p0)
    eps = 0.5 * p.eps_heuristic()
    # first call: everything the engine allocates lazily (ping-pong state buffers, guard slots) and the re-partitioning
    # of the binning's record slots to this field's populations happen here, not inside the timed trajectory (r02's
    # soak timed its very first trajectory: 280.7 steps/s against 319 for the warmed-up 100-step bench)
    free_a = torch.cuda.mem_get_info()[0]
    e.leapfrog_device(q0, p0, q1, p1, eps, 10)
    e.steps_done()
    free0 = torch.cuda.mem_get_info()[0]
    info0 = e.tile_info()
    t0 = time.perf_counter()
    e.leapfrog_device(q0, p0, q1, p1, eps, 1000)
    done = e.steps_done()
    dt = time.perf_counter() - t0
    split = []
    for _ in range(5):   # the same length again as five 200-step trajectories (each carries ~1.1 steps of fixed cost)
        s0 = time.perf_counter()
        e.leapfrog_device(q0, p0, q1, p1, eps, 200)
        e.steps_done()
        split.append(200 / (time.perf_counter() - s0))
    e.leapfrog_device(q0, p0, q1, p1, eps, 1000)
    e.steps_done()
    en0, en1 = e.energies_device(q0, p0), e.energies_device(q1, p1)
    free1 = torch.cuda.mem_get_info()[0]
    print("256^3: 1000 steps in %.2f s (%.1f steps/s), done %d, finite %s, H %.6e -> %.6e (dH %.3e)"
          % (dt, 1000 / dt, done, bool(torch.isfinite(q1).all() and torch.isfinite(p1).all()), en0.sum(), en1.sum(),
             en1.sum() - en0.sum()))
    print("       200-step trajectories: %s steps/s; first (10-step) call allocated %.1f MB, the timed calls %.1f MB; "
          "record slots per tile %d of %d allocated" % (" ".join("%.1f" % v for v in split), (free_a - free0) / 1e6,
                                                         (free0 - free1) / 1e6, e.tile_info()["cap"], info0["cap_alloc"]))
    # device memory is stable once the first trajectory has run (allocator granularity: a few MB at most)
    assert abs(free0 - free1) < 64e6, "device memory moved by %.1f MB during the timed trajectories" % ((free0 - free1) / 1e6)
    assert e.tile_info()["cap_alloc"] == info0["cap_alloc"]
    assert done == 1000
    e.close()
    # ---- sampler loop
    p = HamilParams(Nx=128, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
    f = inputs.make_fields(p)
    hd = hamil.HamilData(p, N_eps_fac=8.0, eps_fac=2.0 * p.eps_heuristic())
    mock(p, hd.engine, f)
    hd.engine.chain_set_state(f["q0"])
    rng = np.random.default_rng(7)
    ring = EpsRing()
    n_acc = n_att = 0
    for sample in range(6):
        log = hamil.HamiltonianMC(hd, rng.random, seed=11, itmax=50, ring=ring)
        n_att += len(log)
        n_acc += sum(r["accepted"] for r in log)
        r = log[-1]
        print("sample %d: %d attempt(s), last dH %.3e, eps %.3e, Neps %d, accepted %s" %
              (sample, len(log), r["dH"], r["epsilon"], r["Neps"], r["accepted"]))
    km, pw = hd.engine.measure_spectrum(None, 50)
    print("acceptance %d/%d; resident-state spectrum bins with power: %d" % (n_acc, n_att, int((pw > 0).sum())))
    assert n_acc == 6 and np.isfinite(pw).all()
    hd.engine.close()
    # ---- the C++ HamiltonianMC with the reference's step-size adaptation (scheme 3), from a step size 40x too large:
    # the first sample halves eps_fac on every rejection (time_step.cpp:137-149), later samples adapt from the tables
    from barcode_amd import time_step as ts
    from barcode_amd.shim import ShimHamil
    from tests.util import Case
    c = Case(Nx=64, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
    c.oracle.close()
    sh = ShimHamil(c.p, N_eps_fac=8.0, eps_fac=40.0 * c.p.eps_heuristic(), **c.arrays())
    sh.eps_attach(ts.EpsConfig(eps_fac_update_type=3, N_a_eps_update=20, acc_min=0.6, acc_max=0.7))
    sh.chain_set_state(c.q0)
    rng = np.random.default_rng(3)
    tot = acc = 0
    for sample in range(1, 41):
        sh.numerical.iGibbs, sh.numerical.rejections = sample, 0
        log = sh.HamiltonianMC(rng.random, seed=5, itmax=200)
        tot += len(log)
        acc += 1 if log[-1]["accepted"] else 0
        if sample <= 3 or sample % 10 == 0:
            print("C++ sample %2d: %2d attempt(s), eps_fac %.3e, table acceptance %.2f, records %d"
                  % (sample, len(log), sh.numerical.eps_fac, sh.eps_acceptance_rate(), sh.eps_records()))
        assert log[-1]["accepted"], "a sample ran to itmax"
    print("C++ loop: %d samples in %d attempts (acceptance %.2f), final eps_fac / heuristic = %.2f"
          % (acc, tot, acc / tot, sh.numerical.eps_fac / c.p.eps_heuristic()))
    assert acc == 40 and sh.eps_records() == tot
    sh.close()


if __name__ == "__main__":
    main()
