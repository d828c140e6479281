"""The host-array drop-in path (what barcode/main.cc reaches through the shim): staged copies, the single-pass
bchmc_leapfrog_dh (trajectory + the six energy terms; bchmc_delta_hamiltonian itself always evaluates), the
kinetic_term / psi entry points, mass re-upload by generation, and the RCCL transport of the record exchange (world
size 1: all a one-GPU box can run)."""
import numpy as np
import pytest

from tests.util import TOL_ENERGY, TOL_TRAJ_10, Case, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [dict(likelihood=1, rsd_model=1, sfmodel=2), dict(likelihood=0, rsd_model=0),
                                dict(likelihood=0, rsd_model=1), dict(likelihood=1, mass_type=5),
                                dict(likelihood=3)],
                         ids=["gauss_rsd_fast", "poisson_fast", "poisson_rsd_generic", "mass_rs_generic", "grf_generic"])
def test_leapfrog_dh_is_leapfrog_plus_delta_hamiltonian(kw):
    """bchmc_leapfrog_dh: Hamiltonian_EoM and delta_Hamiltonian of the same four arrays in one pass (HMC.cc:455-459).
    Its six terms are the ones the always-evaluating bchmc_delta_hamiltonian gives, and the oracle's."""
    c = Case(Nx=16, **kw)
    e = c.engine()
    q1, p1, done, dH, t = e.leapfrog_dh(c.q0, c.p0, c.eps, 6)
    q1b, p1b, doneb = e.leapfrog(c.q0, c.p0, c.eps, 6)                     # the plain trajectory
    assert done == doneb == 6 and rel_l2(q1b, q1) < 1e-13 and rel_l2(p1b, p1) < 1e-13
    dH_full, t_full = e.delta_hamiltonian(c.q0, c.p0, q1, p1)
    assert np.all(np.abs(t - t_full) <= TOL_ENERGY * np.abs(t_full))
    assert abs(dH - dH_full) <= 10 * TOL_ENERGY * np.abs(t_full).max()
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 6)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    assert np.all(np.abs(t - to) <= 10 * TOL_ENERGY * np.abs(to))
    # hd->deltaX after bchmc_leapfrog_dh is psi(signalf)'s (HMC.cc:225), as after the pair of calls
    if c.p.likelihood != 3:  # the GRF likelihood has no forward model
        e.leapfrog_dh(c.q0, c.p0, c.eps, 6)
        dX = e.fetch("deltaX")
        c.oracle.psi(q1o)
        assert rel_l2(dX, c.oracle.get("deltaX")) < 1e-9
    # zero steps: the state comes back as it went in and both ends have the same energies
    q1z, p1z, donez, dHz, tz = e.leapfrog_dh(c.q0, c.p0, c.eps, 0)
    assert donez == 0 and rel_l2(q1z, c.q0) < 1e-13 and abs(dHz) <= 1e-9 * np.abs(tz).max()
    assert np.all(np.abs(tz[:3] - t_full[:3]) <= TOL_ENERGY * np.abs(t_full[:3]))
    e.close()


@pytest.mark.parametrize("nx", [16, 64])
def test_q1_crosses_pcie_beside_the_last_force_evaluation(monkeypatch, nx):
    """The last leapfrog step only kicks p (HMC.cc:343-352), so a host-array trajectory transforms its final q before
    the last force evaluation and downloads it on the copy stream while that runs.  Same arrays as with
    BCHMC_NO_DOWNLOAD_OVERLAP=1 (to the scatter's atomic-order noise) for the one-pass and the plain entry point, one
    and several steps, several staging chunks -- and for a trajectory the runaway guard stops, where the q sent early
    is NOT the state returned (HMC.cc:360-364) and has to be replaced."""
    c = Case(Nx=nx, likelihood=1, rsd_model=1)
    monkeypatch.setenv("BCHMC_STAGE_MB", "1")   # 64^3: two chunks per array
    p_bad = c.p0.copy().ravel()
    p_bad[0] = 1e60
    outs = []
    for off in ("0", "1"):
        monkeypatch.setenv("BCHMC_NO_DOWNLOAD_OVERLAP", off)
        e = c.engine()
        r = [e.leapfrog_dh(c.q0, c.p0, c.eps, 3)[:3], e.leapfrog_dh(c.q0, c.p0, c.eps, 1)[:3],
             e.leapfrog(c.q0, c.p0, c.eps, 4), e.leapfrog(c.q0, c.p0, c.eps, 1),
             e.leapfrog_dh(c.q0, p_bad, 1e-6, 4)[:3], e.leapfrog(c.q0, p_bad, 1e-6, 4),
             e.leapfrog_dh(c.q0, c.p0, c.eps, 2)[:3]]   # and a normal one after the stopped ones
        outs.append(r)
        e.close()
    for (qa, pa, da), (qb, pb, db) in zip(*outs):
        assert da == db
        assert rel_l2(qa, qb) < 1e-13 and rel_l2(pa, pb) < 1e-13
    assert outs[0][4][2] == outs[0][5][2] == 1
    if nx == 16:
        q1o, p1o, done_o = c.oracle.Hamiltonian_EoM(c.q0, p_bad, 1e-6, 4)
        assert done_o == 1
        for k in (4, 5):
            assert rel_l2(outs[0][k][0], q1o) < 1e-10 and rel_l2(outs[0][k][1], p1o) < 1e-10


def test_delta_hamiltonian_never_answers_from_an_earlier_call():
    """VERDICT r2 / ADVICE r2: the C ABI keeps no energy cache.  Whatever happened before, bchmc_delta_hamiltonian
    evaluates the arrays it is given against the inputs uploaded now."""
    c = Case(Nx=16, likelihood=1)
    e = c.engine()
    q0, p0 = c.q0.reshape(-1).copy(), c.p0.reshape(-1).copy()
    q1, p1, _, _, t = e.leapfrog_dh(q0, p0, c.eps, 3)
    # (1) a new mass between the trajectory and the question: both kinetic terms must follow it
    e.upload(mass_f=2.0 * c.mass_f)
    _, t_mass = e.delta_hamiltonian(q0, p0, q1, p1)
    assert np.allclose(t_mass[[0, 3]], 0.5 * t[[0, 3]], rtol=1e-12)
    assert np.allclose(t_mass[[1, 2, 4, 5]], t[[1, 2, 4, 5]], rtol=1e-12)
    e.upload(mass_f=c.mass_f)
    # (2) one element changed in place, away from every sampling stride a fingerprint could use
    q1, p1, _, _, t = e.leapfrog_dh(q0, p0, c.eps, 3)
    i = 1234
    assert i % max(q1.size // 509, 1) != 0
    q1[i] += 0.5
    _, t_mod = e.delta_hamiltonian(q0, p0, q1, p1)
    _, t_ref = e.delta_hamiltonian(q0.copy(), p0.copy(), q1.copy(), p1.copy())
    assert np.allclose(t_mod, t_ref, rtol=1e-12)
    assert abs(t_mod[4] - t[4]) > 1e-6 * abs(t[4]) and np.allclose(t_mod[:4], t[:4], rtol=1e-12)
    # (3) an in-place trajectory (q1 is q0, p1 is p0) followed by the question about those arrays: both ends are the
    # same arrays now, so dH is 0 -- not the trajectory's dH
    qa, pa = q0.copy(), p0.copy()
    e.leapfrog(qa, pa, c.eps, 3, out=(qa, pa))
    dH_same, t_same = e.delta_hamiltonian(qa, pa, qa, pa)
    assert np.allclose(t_same[:3], t_same[3:], rtol=1e-12) and abs(dH_same) <= 1e-10 * np.abs(t_same).max()
    assert abs(t[3] - t[0]) > 1e-6 * abs(t[0])                             # ... whereas the trajectory's dK is not 0
    e.close()


def test_host_calls_invalidate_a_pending_chain_proposal():
    """bchmc_psi / bchmc_kinetic_term / bchmc_energies / ... overwrite the k-space buffers that hold the
    resident chain's proposal; bchmc_chain_accept must then refuse instead of committing the wrong state."""
    from barcode_amd.engine import BchmcError
    c = Case(Nx=16, likelihood=1)
    e = c.engine()
    e.chain_set_state(c.q0)
    e.chain_set_momenta(c.p0)
    e.chain_attempt(c.eps, 3)
    e.psi(c.truth)
    with pytest.raises(BchmcError) as err:
        e.chain_accept(True)
    assert err.value.code == 9                                             # BCHMC_ERR_STATE
    with pytest.raises(BchmcError):
        e.chain_get_proposal()
    # the chain itself is intact: the next attempt proposes, accepts and matches a chain that was never disturbed
    dH, t, done = e.chain_attempt(c.eps, 3)
    e.chain_accept(True)
    x = e.chain_get_state()
    e2 = c.engine()
    e2.chain_set_state(c.q0)
    e2.chain_set_momenta(c.p0)
    dH2, t2, _ = e2.chain_attempt(c.eps, 3)
    e2.chain_accept(True)
    assert np.allclose(t, t2, rtol=1e-12) and rel_l2(x, e2.chain_get_state()) < 1e-13
    e.close()
    e2.close()


def test_shim_reuses_the_trajectory_energies_only_under_its_stated_contract():
    """bchmc_shim::Hamiltonian_EoM keeps bchmc_leapfrog_dh's six terms; delta_Hamiltonian answers from them only as the
    NEXT call, about the same four arrays, with unchanged inputs (HMC.cc:455-459).  Everything else evaluates."""
    from barcode_amd.shim import ShimHamil
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    arrays = c.arrays()
    arrays["mass_f"] = arrays["mass_f"].copy()
    hd = ShimHamil(c.p, N_eps_fac=8.0, eps_fac=c.eps * 2, **arrays)
    n = hd.numerical
    q0, p0 = c.q0.reshape(-1).copy(), c.p0.reshape(-1).copy()
    qf, pf = np.empty_like(q0), np.empty_like(p0)

    def eom():
        d = iter([0.3, 0.5])
        hd.Hamiltonian_EoM(q0, p0, lambda: next(d), out=(qf, pf))

    def terms():
        return np.array([n.H_kin_i, n.psi_prior_i, n.psi_likeli_i, n.H_kin_f, n.psi_prior_f, n.psi_likeli_f])

    eom()
    assert hd.hd.eom.valid
    dH = hd.delta_Hamiltonian(q0, p0, qf, pf)                               # reused
    t_reuse = terms()
    assert not hd.hd.eom.valid                                             # one use
    dH2 = hd.delta_Hamiltonian(q0, p0, qf, pf)                              # evaluated
    assert np.all(np.abs(terms() - t_reuse) <= TOL_ENERGY * np.abs(t_reuse)) and abs(dH - dH2) <= 1e-9 * np.abs(t_reuse).max()
    dX_eval = hd.out("deltaX").copy()
    eom()
    hd.delta_Hamiltonian(q0, p0, qf, pf)
    assert rel_l2(hd.out("deltaX"), dX_eval) < 1e-12                       # hd->deltaX is psi(signalf)'s either way
    # other arrays -> evaluated (and correct for them)
    eom()
    hd.delta_Hamiltonian(q0, p0, qf.copy(), pf)
    assert np.all(np.abs(terms() - t_reuse) <= TOL_ENERGY * np.abs(t_reuse))
    # the mass changes between the two calls (mass_changed): both kinetic terms follow the new mass
    eom()
    hd._keep["mass_f"] *= 2.0                                              # the caller-owned array hd->mass_f points at
    hd.mass_changed()
    hd.delta_Hamiltonian(q0, p0, qf, pf)
    assert np.allclose(terms()[[0, 3]], 0.5 * t_reuse[[0, 3]], rtol=1e-10)
    hd._keep["mass_f"] *= 0.5
    hd.mass_changed()
    # another call in between ends the validity
    eom()
    hd.psi(c.truth)
    assert not hd.hd.eom.valid
    # mode 2: the contents are verified, a single changed element is noticed
    hd.hd.reuse_eom_energies = 2
    eom()
    hd.delta_Hamiltonian(q0, p0, qf, pf)
    assert np.all(np.abs(terms() - t_reuse) <= TOL_ENERGY * np.abs(t_reuse))
    eom()
    qf[3211] += 0.5
    hd.delta_Hamiltonian(q0, p0, qf, pf)
    assert abs(terms()[4] - t_reuse[4]) > 1e-6 * abs(t_reuse[4])
    # mode 0: Hamiltonian_EoM is the plain trajectory, nothing is kept
    hd.hd.reuse_eom_energies = 0
    eom()
    assert not hd.hd.eom.valid
    hd.delta_Hamiltonian(q0, p0, qf, pf)
    assert np.all(np.abs(terms() - t_reuse) <= TOL_ENERGY * np.abs(t_reuse))
    hd.close()


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


def test_bench_self_launch_two_ranks_reports_what_the_collective_saw():
    """`python bench.py --gpus 2` without a launcher: the parent spawns two fresh rank processes (and asserts that it
    never initialised the GPU itself), the ranks meet over gloo on one device (a one-GPU box cannot host two RCCL
    ranks), and the JSON line says what the collective saw -- world, ranks, records -- so that a recorded N-GPU line
    needs no inference about the rank count."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-device", "--backend",
                        "gloo", "--nx", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-kernel-profile", "--force-rccl-probe"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["rehearsal_single_device"] is True
    c = d["collective"]
    assert c["backend"] == "gloo" and c["transport"] == "custom" and c["world_seen"] == 2
    assert c["ranks_seen"] == [0, 1] and c["records"] == 2 and c["consistent_across_ranks"] and c["ok"]
    assert d["config"]["steps_done"] == 3 and d["config"]["finite"]
    # the untimed probe of the library's own RCCL transport ran too (forced here: two ranks on one device, which RCCL
    # refuses) -- it reports the refusal and costs the measurement nothing; with one GPU per rank it reports world / ranks
    probe = c["rccl_native"]
    assert probe["ok"] is False and "ncclCommInitRank" in probe["error"] and not probe.get("hung")
