"""Deterministic mode (bchmc_config.deterministic; the reference announces the run-to-run noise of its
OpenMP atomics at barcode/main.cc:86-90): fixed-point mass assignment makes every result bitwise repeatable, and it
stays within the parity tolerances of the default mode."""
import numpy as np
import pytest

from tests.util import TOL_FIELD, TOL_TRAJ_10, Case, rel_l2

pytestmark = pytest.mark.gpu


def _engine(c, **kw):
    from barcode_amd.engine import Engine
    e = Engine(c.p, deterministic=1, **kw)
    e.upload(**c.arrays())
    return e


@pytest.mark.parametrize("kw", [dict(Nx=32, likelihood=1, rsd_model=1, sfmodel=2), dict(Nx=24, likelihood=0),
                                dict(Nx=16, likelihood=1, mk=1, calc_h=1), dict(Nx=16, likelihood=1, mk=2, calc_h=1),
                                dict(Nx=18, likelihood=2)],
                         ids=["tile81_rsd_32", "generic_tiles_24", "cic", "tsc", "direct_sph_18"])
def test_two_runs_are_bitwise_identical_and_match_the_oracle(kw):
    c = Case(**kw)
    runs = []
    for _ in range(2):
        e = _engine(c)   # a fresh handle each time: nothing carried over
        e.forward(c.q0)
        rho = e.fetch("rho")
        g = e.gradient(c.q0)
        q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 10)
        dH, terms = e.delta_hamiltonian(c.q0, c.p0, q1, p1)
        runs.append((rho, g, q1, p1, terms))
        e.close()
    for a, b in zip(*runs):
        assert np.array_equal(a, b)
    rho, g, q1, p1, terms = runs[0]
    dXo, px, py, pz = c.oracle.Lag2Eul(c.q0, rsd=c.p.rsd_model if c.p.likelihood == 1 else 0)
    mk = c.p.mk
    rho_o = c.oracle.getDensity(mk, px, py, pz)
    assert rel_l2(rho, rho_o) < TOL_FIELD
    go, _, _ = c.oracle.gradient_psi(c.q0)
    assert rel_l2(g, go) < 10 * TOL_FIELD
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 10)
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10


def test_deterministic_mode_fp32_fields_and_environment_switch(monkeypatch):
    c = Case(Nx=32, likelihood=1, rsd_model=1)
    outs = []
    for _ in range(2):
        e = _engine(c, precision=1)
        q1, p1, _ = e.leapfrog(c.q0, c.p0, c.eps, 6)
        outs.append((q1, p1))
        e.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    monkeypatch.setenv("BCHMC_DETERMINISTIC", "1")
    e1, e2 = c.engine(), c.engine()     # deterministic = 0 in the config: the environment switches it on
    assert np.array_equal(e1.gradient(c.q0), e2.gradient(c.q0))
    e1.close()
    e2.close()


def test_deterministic_mode_at_256_cubed():
    """The benchmarked size: two 3-step trajectories on two handles, bitwise equal."""
    from barcode_amd import inputs
    from barcode_amd.engine import Engine
    from barcode_amd.params import HamilParams
    p = HamilParams(Nx=256, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
    f = inputs.make_fields(p)
    res = []
    for _ in range(2):
        e = Engine(p, deterministic=1)
        e.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], nobs=np.ones(p.N), window=np.ones(p.N), noise=np.ones(p.N))
        q1, p1, done = e.leapfrog(f["q0"], f["p0"], 0.5 * p.eps_heuristic(), 3)
        res.append((q1, p1))
        e.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_carried_gradient_is_the_recomputed_one_bit_for_bit(monkeypatch):
    """In deterministic mode the gradient a chain carries from the end of an accepted trajectory IS the one the next
    trajectory would evaluate at its start (same state, same kernels, same order): the chain with the carry and the
    chain that re-evaluates (BCHMC_NO_FORCE_CARRY=1) produce identical bits, planes mode included (32^3 with padding)."""
    monkeypatch.setenv("BCHMC_FFT_PAD", "1")
    # the identity needs the trajectory's last force evaluation and a fresh first one to run the same kernels: true
    # with planes mode at the ends (the default) and without planes mode, not with planes mode for interior steps only
    monkeypatch.delenv("BCHMC_NO_PLANES_ENDS", raising=False)
    c = Case(Nx=32, likelihood=1, rsd_model=1)
    outs = []
    for no_carry in ("0", "1"):
        monkeypatch.setenv("BCHMC_NO_FORCE_CARRY", no_carry)
        e = _engine(c)
        e.chain_set_state(c.q0)
        seq = []
        for i, acc in enumerate([True, False, True, True]):
            e.chain_draw_momenta(99, i)
            dH, terms, done = e.chain_attempt(c.eps, 3)
            q1, p1 = e.chain_get_proposal()
            seq.append((dH, terms.copy(), q1, p1))
            e.chain_accept(acc)
        seq.append((0., np.zeros(6), e.chain_get_state(), e.chain_get_momenta()))
        outs.append(seq)
        e.close()
    for (dH_a, t_a, q_a, p_a), (dH_b, t_b, q_b, p_b) in zip(*outs):
        assert dH_a == dH_b and np.array_equal(t_a, t_b)
        assert np.array_equal(q_a, q_b) and np.array_equal(p_a, p_b)


def test_fixed_point_saturation_is_reported_not_hidden(monkeypatch):
    """a fixed-point density cell that came within a factor two of wrapping (or came out negative) must turn
    into BCHMC_ERR_STATE at the next synchronising call, not into a plausible density.  The limit is 2^62 (2^16 maximal
    contributions per cell); the test hook lowers it to 2^47 -- two maximal contributions -- so that an ordinary field
    trips it."""
    from barcode_amd.engine import BchmcError, Engine
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    monkeypatch.setenv("BCHMC_FIX_SAT_LOG2", "47")
    e = Engine(c.p, deterministic=1)
    e.upload(**c.arrays())
    with pytest.raises(BchmcError) as err:
        e.forward(c.truth, 1)
    assert err.value.code == 9 and "fixed-point range" in str(err.value)
    e.close()
    monkeypatch.delenv("BCHMC_FIX_SAT_LOG2")
    e = Engine(c.p, deterministic=1)                       # the real limit: an ordinary field is far below it
    e.upload(**c.arrays())
    e.forward(c.truth, 1)
    e.close()
