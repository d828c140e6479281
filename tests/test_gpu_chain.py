"""Device-resident chain: bchmc_chain_* keeps q and p in HBM across attempts, takes both
-log L values from the trajectory's own force evaluations, and can draw the momenta on the device."""
import numpy as np
import pytest

from tests.util import TOL_ENERGY, TOL_TRAJ_10, Case, rel_l2

pytestmark = pytest.mark.gpu


def test_philox_known_answers():
    """Random123 known-answer vectors for Philox4x32-10 (kat_vectors)."""
    from barcode_amd.engine import philox_kat
    assert philox_kat([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert philox_kat([f, f, f, f], [f, f]) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox_kat([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("kw", [dict(likelihood=1, rsd_model=1), dict(likelihood=0), dict(likelihood=3),
                                dict(likelihood=1, mass_type=5), dict(likelihood=1, deltaQ_factor=0.9),
                                dict(likelihood=2, deltaQ_factor=0.9), dict(likelihood=1, sfmodel=2),
                                dict(likelihood=0, rsd_model=1, sfmodel=2, eps_scale=0.01)],
                         ids=["gauss_rsd_fast", "poisson_fast", "grf_generic", "mass5_generic", "gauss_dq_fast",
                              "lognormal_dq_generic", "gauss_alpt_fast", "poisson_rsd_alpt_generic"])
def test_attempt_equals_leapfrog_plus_delta_hamiltonian(kw):
    """One attempt on the resident chain == Hamiltonian_EoM followed by delta_Hamiltonian on host arrays
    (and hence == the oracle), for the shared-forward-model fast path and the generic fallback."""
    c = Case(Nx=16, **kw)
    e = c.engine()
    e.chain_set_state(c.q0)
    e.chain_set_momenta(c.p0)
    assert rel_l2(e.chain_get_state(), c.q0) < 1e-14 and rel_l2(e.chain_get_momenta(), c.p0) < 1e-14
    dH, terms, done = e.chain_attempt(c.eps, 6)
    q1, p1 = e.chain_get_proposal()
    q1o, p1o, done_o = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 6)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    assert done == done_o == 6
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    assert np.all(np.abs(terms - to) <= 10 * TOL_ENERGY * np.abs(to)), (terms, to)
    assert abs(dH - dHo) <= 1e-9 * np.abs(to).max()
    # reject keeps the state, accept replaces it
    e.chain_accept(False)
    assert rel_l2(e.chain_get_state(), c.q0) < 1e-14
    e.chain_attempt(c.eps, 6)
    e.chain_accept(True)
    assert rel_l2(e.chain_get_state(), q1o) < TOL_TRAJ_10
    e.close()


@pytest.mark.parametrize("kw", [dict(likelihood=1, rsd_model=1), dict(likelihood=0)], ids=["gauss_rsd", "poisson"])
def test_gradient_carried_across_attempts(kw, monkeypatch):
    """The chain keeps gradient_psi and -log L of its state from one attempt to the next (the end of an accepted
    trajectory, or the start of a rejected one, is the next start: the evaluation of HMC.cc:279 is already known).
    Every attempt of an accept / reject sequence must still equal Hamiltonian_EoM + delta_Hamiltonian of the oracle
    from the same state, the carried chain must equal the recomputing one, and new inputs must drop the carry."""
    c = Case(Nx=16, **kw)
    rng = np.random.default_rng(5)
    pattern = [True, False, False, True, True]
    moms = [c.p0 * (1. + 0.1 * rng.standard_normal()) for _ in pattern]

    def run(engine):
        out = []
        engine.chain_set_state(c.q0)
        for acc, p in zip(pattern, moms):
            engine.chain_set_momenta(p)
            dH, terms, done = engine.chain_attempt(c.eps, 3)
            out.append((dH, terms.copy(), engine.chain_get_proposal()[0]))
            engine.chain_accept(acc)
        return out

    e = c.engine()
    carried = run(e)
    monkeypatch.setenv("BCHMC_NO_FORCE_CARRY", "1")
    recomputed = run(e)
    monkeypatch.delenv("BCHMC_NO_FORCE_CARRY")
    q = c.q0
    for acc, p, (dH, terms, q1), (dH2, terms2, q12) in zip(pattern, moms, carried, recomputed):
        q1o, p1o, _ = c.oracle.Hamiltonian_EoM(q, p, c.eps, 3)
        dHo, to = c.oracle.delta_Hamiltonian(q, p, q1o, p1o)
        assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(q12, q1o) < TOL_TRAJ_10
        assert np.all(np.abs(terms - to) <= 10 * TOL_ENERGY * np.abs(to)), (terms, to)
        assert np.all(np.abs(terms - terms2) <= 10 * TOL_ENERGY * np.abs(to))
        assert abs(dH - dHo) <= 1e-9 * np.abs(to).max() and abs(dH - dH2) <= 1e-9 * np.abs(to).max()
        if acc:
            q = q1o
    # new data: the next attempt must see it (same answer as an engine that never carried anything)
    nobs2 = c.nobs * 1.25 + 0.5
    e.upload(nobs=nobs2)
    e.chain_set_momenta(moms[0])
    dH_new, terms_new, _ = e.chain_attempt(c.eps, 3)
    e.close()
    c2 = Case(Nx=16, **kw)
    e2 = c2.engine()
    e2.upload(nobs=nobs2)
    e2.chain_set_state(q)
    e2.chain_set_momenta(moms[0])
    dH_ref, terms_ref, _ = e2.chain_attempt(c.eps, 3)
    e2.close()
    assert np.all(np.abs(terms_new - terms_ref) <= 1e-9 * np.abs(terms_ref)), (terms_new, terms_ref)
    assert abs(terms_new[2] - carried[-1][1][5]) > 1e-6 * abs(terms_new[2])  # and it is not the stale -log L


def test_attempt_after_runaway_guard():
    c = Case(Nx=16)
    e = c.engine()
    p0 = c.p0.copy().ravel()
    p0[0] = 1e60
    e.chain_set_state(c.q0)
    e.chain_set_momenta(p0)
    dH, terms, done = e.chain_attempt(1e-6, 5)
    assert done == 1
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, p0, 1e-6, 5)
    _, to = c.oracle.delta_Hamiltonian(c.q0, p0, q1o, p1o)
    # the final state has displacements ~1e50 cells: its positions (fmod of 1e50) and hence psi_likeli_f are
    # round-off noise in any implementation; everything else must agree
    assert np.all(np.abs(terms[:5] - to[:5]) <= 1e-8 * np.abs(to[:5]))
    assert np.isfinite(terms[5])
    e.close()


def test_device_momentum_draw_statistics_and_reproducibility():
    """p ~ N(0, M): K = 1/2 p^T M^-1 p has mean N_modes/2 and variance N_modes/2; the draw is a pure function
    of (seed, attempt); its spectrum follows mass_f (HMC_momenta.cc:42-74 / random.cpp:82,106 convention)."""
    c = Case(Nx=32, L=100.0)
    e = c.engine()
    e.chain_set_state(c.q0)
    e.chain_draw_momenta(1234, 0)
    pa = e.chain_get_momenta()
    e.chain_draw_momenta(1234, 0)
    assert np.array_equal(pa, e.chain_get_momenta())
    e.chain_draw_momenta(1234, 1)
    pb = e.chain_get_momenta()
    e.chain_draw_momenta(1235, 0)
    pc = e.chain_get_momenta()
    assert abs(np.corrcoef(pa, pb)[0, 1]) < 0.02 and abs(np.corrcoef(pa, pc)[0, 1]) < 0.02
    n_modes = c.p.N - 1  # mass_f(k = 0) = 0: that mode carries no momentum
    for p in (pa, pb, pc):
        K = e.energies(c.q0, p)[0]
        assert abs(K - n_modes / 2) < 5 * np.sqrt(n_modes / 2), K
    n = c.p.Nx
    pk = np.abs(np.fft.rfftn(pa.reshape(n, n, n))) ** 2
    expect = c.p.N ** 2 * c.mass_f[:, :, : n // 2 + 1] / c.p.L ** 3
    sel = expect > 0
    assert abs((pk[sel] / expect[sel]).mean() - 1) < 0.02
    assert abs(pa.mean()) < 1e-10 * np.abs(pa).max() + 1e-12  # no k = 0 power
    e.close()


def test_device_draw_has_the_spectrum_of_the_reference_draw():
    """The device draw against the ORACLE's restatement of the reference's draw (draw_momenta / create_GARFIELD with
    GSL's MT19937 + polar Box-Muller stream, oracle/orc_random.c), not against a formula of this test: the same
    estimator (measure_spectrum, field_statistics.cpp:20-90) on both, bin by bin.  Parity of the draw is statistical
    by construction: the reference's serial, shell-ordered stream is not reproduced on the device;
    a host-drawn reference momentum field goes in unchanged through bchmc_chain_set_momenta."""
    from oracle import oracle as orc
    c = Case(Nx=32, L=100.0)
    e = c.engine()
    e.chain_set_state(c.q0)
    nbin = 24
    pw_dev, pw_ref = np.zeros(nbin), np.zeros(nbin)
    for s in range(8):
        e.chain_draw_momenta(99, s)
        km, pw = e.measure_spectrum(e.chain_get_momenta(), nbin)
        pw_dev += pw
        km2, pw2 = e.measure_spectrum(orc.draw_momenta(c.p, c.mass_f, None, seed=1000 + s), nbin)
        pw_ref += pw2
    sel = (km > 0) & (pw_ref > 0)
    assert np.count_nonzero(sel) >= 20
    ratio = (pw_dev[sel] / pw_ref[sel])
    # two independent sets of reference draws scatter by up to 20 % in the first bins and in the corner bins beyond the
    # Nyquist sphere (few modes each) and by 1-3 % in between: the device draw must sit inside that band
    assert np.all(np.abs(ratio - 1.0) < 0.35)
    assert np.all(np.abs(ratio[6:20] - 1.0) < 0.08), ratio
    assert abs(ratio[6:20].mean() - 1.0) < 0.02
    # and the reference draw itself goes through the engine unchanged
    pr = orc.draw_momenta(c.p, c.mass_f, None, seed=5)
    e.chain_set_momenta(pr)
    assert rel_l2(e.chain_get_momenta(), pr) < 1e-13
    K_dev = e.energies(c.q0, pr)[0]
    o = c.oracle
    assert abs(K_dev - o.kinetic_term(pr)) <= TOL_ENERGY * abs(K_dev)
    e.close()


def test_real_space_mass_draw():
    c = Case(Nx=16, mass_type=0)
    e = c.engine()
    e.chain_set_state(c.q0)
    e.chain_draw_momenta(7, 3)
    p = e.chain_get_momenta()
    # p = sqrt(mass_r) * white (HMC_momenta.cc:76-94): whitened values are N(0,1)
    w = p / np.sqrt(c.mass_r.ravel())
    assert abs(w.mean()) < 5 / np.sqrt(w.size) and abs(w.std() - 1) < 0.05
    e.close()


def test_hamiltonian_mc_loop_on_resident_chain():
    """barcode_amd.hamil.HamiltonianMC: RNG consumption order and accept/reject bookkeeping of HMC.cc:431-511."""
    from barcode_amd import hamil
    from barcode_amd.chains import EpsRing
    c = Case(Nx=16, likelihood=1)
    hd = hamil.HamilData(c.p, N_eps_fac=4.0, eps_fac=4 * c.eps, **c.arrays())
    hd.engine.chain_set_state(c.q0)
    calls = []

    def uniform():
        v = [0.3, 0.6, 0.999999, 0.9, 0.1, 0.0][len(calls) % 6]
        calls.append(v)
        return v

    ring = EpsRing(8)
    log = hamil.HamiltonianMC(hd, uniform, seed=99, ring=ring, itmax=4)
    assert 1 <= len(log) <= 4 and ring.count_attempts == len(log)
    first = log[0]
    assert first["Neps"] == int(4.0 * 0.3) + 1 and np.isclose(first["epsilon"], 4 * c.eps * 0.6)
    assert np.isfinite(first["dH"]) and np.isclose(first["dH"], first["dK"] + first["dE"], rtol=1e-9, atol=1e-6)
    # consumed uniforms: 2 per attempt + 1 acceptance draw for every attempt with dH > 0 that is not exp(-dH) == 1
    expect = sum(2 + (1 if (r["dH"] >= 0 and np.exp(-r["dH"]) < 1.0) else 0) for r in log)
    assert len(calls) == expect
    if log[-1]["accepted"]:
        q1, _ = hd.engine.chain_get_proposal() if False else (hd.engine.chain_get_state(), None)
        assert rel_l2(q1, c.q0) > 0  # the state moved
    hd.engine.close()


@pytest.mark.parametrize("nx,precision", [(16, 0), (12, 0), (16, 1)])
def test_measure_spectrum_host_field_and_resident_state(nx, precision):
    """bchmc_measure_spectrum == measure_spectrum (field_statistics.cpp:20-90) for a host array and for the chain
    state kept on the device (whose R2C is already resident: no transform, no field transfer)."""
    c = Case(Nx=nx)
    e = c.engine(precision=precision)
    tol = 1e-12 if precision == 0 else 2e-5
    for nb in (20, 200):
        kmo, pwo = c.oracle.measure_spectrum(c.q0, nb)
        km, pw = e.measure_spectrum(c.q0, nb)
        atol = tol * pwo.max()  # the k = 0 bin of a zero-mean field is round-off on both sides
        assert np.allclose(km, kmo, rtol=1e-13, atol=0) and np.allclose(pw, pwo, rtol=tol, atol=atol)
        e.chain_set_state(c.q0)
        km2, pw2 = e.measure_spectrum(None, nb)
        assert np.allclose(km2, kmo, rtol=1e-13, atol=0) and np.allclose(pw2, pwo, rtol=tol, atol=atol)
    with pytest.raises(Exception):
        e.measure_spectrum(c.q0, 0)
    e.close()
