"""The host-array drop-in path (what barcode/main.cc reaches through the shim): staged copies, the single-pass
bchmc_leapfrog whose energies bchmc_delta_hamiltonian reuses, the kinetic_term / psi entry points, mass re-upload by
generation, and the RCCL transport of the record exchange (world size 1: all a one-GPU box can run)."""
import numpy as np
import pytest

from tests.util import TOL_ENERGY, TOL_TRAJ_10, Case, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [dict(likelihood=1, rsd_model=1, sfmodel=2), dict(likelihood=0, rsd_model=0),
                                dict(likelihood=0, rsd_model=1), dict(likelihood=1, mass_type=5),
                                dict(likelihood=3)],
                         ids=["gauss_rsd_fast", "poisson_fast", "poisson_rsd_generic", "mass_rs_generic", "grf_generic"])
def test_delta_hamiltonian_after_leapfrog_reuses_the_trajectory_and_equals_the_explicit_evaluation(kw, monkeypatch):
    c = Case(Nx=16, **kw)
    e = c.engine()
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 6)
    dH_cached, t_cached = e.delta_hamiltonian(c.q0, c.p0, q1, p1)          # same arrays: answered from the trajectory
    dH_full, t_full = e.delta_hamiltonian(c.q0.copy(), c.p0.copy(), q1.copy(), p1.copy())   # other arrays: evaluated
    assert np.all(np.abs(t_cached - t_full) <= TOL_ENERGY * np.abs(t_full))
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 6)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    assert np.all(np.abs(t_cached - to) <= 10 * TOL_ENERGY * np.abs(to))
    # hd->deltaX after the pair of calls is psi(signalf)'s (HMC.cc:225), with or without the reuse
    if c.p.likelihood != 3:  # the GRF likelihood has no forward model
        dX = e.fetch("deltaX")
        c.oracle.psi(q1o)
        assert rel_l2(dX, c.oracle.get("deltaX")) < 1e-9
    # a changed array is noticed (content fingerprint), not answered from the cache
    q1b = q1.copy()
    e.leapfrog(c.q0, c.p0, c.eps, 6)
    dH2, t2 = e.delta_hamiltonian(c.q0, c.p0, q1b, p1)                     # q1b: different pointer
    assert np.all(np.abs(t2 - t_full) <= TOL_ENERGY * np.abs(t_full))
    monkeypatch.setenv("BCHMC_NO_DH_CACHE", "1")
    q3, p3, _ = e.leapfrog(c.q0, c.p0, c.eps, 6)                           # plain trajectory, nothing cached
    assert rel_l2(q3, q1) < 1e-13 and rel_l2(p3, p1) < 1e-13
    e.close()


def test_fingerprint_catches_in_place_modification():
    c = Case(Nx=16, likelihood=1)
    e = c.engine()
    q0, p0 = c.q0.reshape(-1).copy(), c.p0.reshape(-1).copy()
    q1, p1, _ = e.leapfrog(q0, p0, c.eps, 3)
    _, t = e.delta_hamiltonian(q0, p0, q1, p1)
    q1 *= 1.5                                                              # same pointer, new contents
    _, t_mod = e.delta_hamiltonian(q0, p0, q1, p1)
    _, t_ref = e.delta_hamiltonian(q0.copy(), p0.copy(), q1.copy(), p1.copy())
    assert np.allclose(t_mod, t_ref, rtol=1e-12) and not np.allclose(t_mod[3:], t[3:], rtol=1e-6)
    e.close()


def test_kinetic_term_and_psi_entry_points():
    for kw in (dict(likelihood=1, rsd_model=1), dict(likelihood=0), dict(likelihood=1, mass_type=5), dict(likelihood=3)):
        c = Case(Nx=16, **kw)
        e = c.engine()
        K, prior, like = e.energies(c.q0, c.p0)
        assert abs(e.kinetic_term(c.p0) - K) <= 1e-13 * abs(K)
        pr, li = e.psi(c.q0)
        assert abs(pr - prior) <= 1e-13 * abs(prior) and abs(li - like) <= 1e-12 * abs(like)
        assert abs(K - c.oracle.kinetic_term(c.p0)) <= TOL_ENERGY * abs(K)
        e.close()
    # kinetic_term needs the mass only (HMC.cc:64-121): no data arrays uploaded
    from barcode_amd.engine import Engine
    c = Case(Nx=16, likelihood=1)
    e2 = Engine(c.p)
    e2.upload(mass_f=c.mass_f)
    assert abs(e2.kinetic_term(c.p0) - c.oracle.kinetic_term(c.p0)) <= TOL_ENERGY * abs(c.oracle.kinetic_term(c.p0))
    e2.close()


def test_staged_copies_round_trip_large_and_odd_sizes(monkeypatch):
    """h2d / d2h chunking (2 x BCHMC_STAGE_MB pinned chunks, several host threads): bit-exact round trips, also when
    the array is not a multiple of the chunk."""
    from barcode_amd.engine import Engine
    from barcode_amd.params import HamilParams
    monkeypatch.setenv("BCHMC_STAGE_MB", "1")
    monkeypatch.setenv("BCHMC_STAGE_THREADS", "3")
    p = HamilParams(Nx=72, L=100.0)           # 2.98 MB per array: 3 chunks of 1 MiB, the last one partial
    e = Engine(p)
    rng = np.random.default_rng(3)
    a = rng.standard_normal(p.N)
    e.upload(nobs=a)
    assert np.array_equal(e.fetch("nobs"), a)
    e.close()


def test_shim_kinetic_psi_and_mass_generation():
    from barcode_amd.shim import ShimHamil
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    hd = ShimHamil(c.p, eps_fac=c.eps, **c.arrays())
    K = hd.kinetic_term(c.p0)
    assert abs(K - c.oracle.kinetic_term(c.p0)) <= TOL_ENERGY * abs(K)
    ps = hd.psi(c.q0)
    pr, li = c.oracle.psi(c.q0)
    assert abs(ps - (pr + li)) <= TOL_ENERGY * abs(pr + li)
    assert hd.numerical.psi_prior == pytest.approx(pr, rel=1e-10) and hd.numerical.psi_likeli == pytest.approx(li, rel=1e-10)
    assert rel_l2(hd.out("deltaX"), c.oracle.get("deltaX")) < 1e-11
    # the mass changes under the engine (HMC.cc:400-423 recomputes it per sample): seen only after inputs_changed()
    hd._keep["mass_f"] *= 2.0
    assert hd.kinetic_term(c.p0) == K
    hd.inputs_changed()
    assert hd.kinetic_term(c.p0) == pytest.approx(K / 2.0, rel=1e-12)
    assert hd.hd.uploaded_generation == hd.hd.inputs_generation == 1
    hd.close()


def test_rccl_transport_world_size_one(tmp_path):
    """ncclCommInitRank + ncclAllGather + destroy on this GPU through the C ABI and through the shim's file bootstrap.
    More ranks need more GPUs (RCCL refuses two ranks on one device): the multi-rank behaviour of the exchange is
    covered with the custom transport on CPU (tests/test_eps_host.py)."""
    from barcode_amd import engine as eng
    uid = eng.Comm.unique_id()
    assert len(uid) == eng.UNIQUE_ID_BYTES and any(uid)
    c = eng.Comm(rank=0, world=1, device=0, unique_id=uid)
    assert c.exchange([(0.5, True, 4), (0.25, False, 2)]) == [(0, 0.5, True, 4), (0, 0.25, False, 2)]
    big = [(0.001 * i, True, i) for i in range(40)]
    assert len(c.exchange(big)) == eng.EPS_BATCH and c.pending() == 8
    assert len(c.exchange([])) == 8
    c.close()
    from barcode_amd.shim import ShimHamil
    from barcode_amd import time_step as ts
    cs = Case(Nx=16, likelihood=1)
    hd = ShimHamil(cs.p, eps_fac=cs.eps, **cs.arrays())
    hd.eps_attach(ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=10))
    hd.comm_bootstrap_file(str(tmp_path / "bchmc_unique_id"), 0, 1)
    hd.chain_set_state(cs.q0)
    u = iter([0.3, 0.5, 0.9] * 50)
    log = hd.HamiltonianMC(lambda: next(u), seed=7, itmax=20)
    assert log and log[-1]["accepted"] and hd.eps_records() == len(log)   # a single rank pools nothing
    hd.comm_release()
    hd.close()
