"""The compiled C++ host layer (include/bchmc_shim.hpp, barcode_amd/shim/hmc_hip_shim.cc): the reference's function
names on a view of HAMIL_DATA, checked against the oracle exactly like the reference's own functions would be."""
import numpy as np
import pytest

from tests.util import TOL_ENERGY, TOL_FIELD, TOL_TRAJ_10, Case, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [dict(likelihood=1, rsd_model=1, sfmodel=2), dict(likelihood=0, rsd_model=0),
                                dict(likelihood=1, rsd_model=0, sfmodel=2)],
                         ids=["gauss_rsd", "poisson", "gauss_alpt"])
def test_cpp_layer_matches_oracle(kw):
    from barcode_amd.shim import ShimHamil
    c = Case(Nx=16, **kw)
    hd = ShimHamil(c.p, N_eps_fac=8.0, eps_fac=c.eps * 2, **c.arrays())
    # HMC.cc:260-264: Neps = int(N_eps_fac * u1) + 1, epsilon = eps_fac * u2, in this order
    draws = iter([0.55, 0.5])
    qf, pf, done = hd.Hamiltonian_EoM(c.q0, c.p0, lambda: next(draws))
    n = hd.numerical
    assert n.Neps == 5 and np.isclose(n.epsilon, c.eps) and hd.count_attempts.value == 1 and done == 5
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 5)
    assert rel_l2(qf, q1o) < TOL_TRAJ_10 and rel_l2(pf, p1o) < TOL_TRAJ_10
    dH = hd.delta_Hamiltonian(c.q0, c.p0, qf, pf)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    got = np.array([n.H_kin_i, n.psi_prior_i, n.psi_likeli_i, n.H_kin_f, n.psi_prior_f, n.psi_likeli_f])
    assert np.all(np.abs(got - to) <= 100 * TOL_ENERGY * np.abs(to))
    assert abs(dH - dHo) <= 1e-8 * np.abs(to).max() and n.dH == dH
    assert np.isclose(n.dK, n.H_kin_f - n.H_kin_i) and np.isclose(n.dE, n.dprior + n.dlikeli)
    assert n.psi_prior == n.psi_prior_f and n.psi_likeli == n.psi_likeli_f
    # hd->deltaX holds the last evaluation's field (psi(signalf), HMC.cc:225)
    assert rel_l2(hd.out("deltaX"), c.oracle.get("deltaX")) < 1e-9
    g = hd.gradient_psi(c.q0)
    go, _, _ = c.oracle.gradient_psi(c.q0)
    assert rel_l2(g, go) < 10 * TOL_FIELD
    for k in ("deltaX", "posx", "posy", "posz"):
        assert rel_l2(hd.out(k), c.oracle.get(k)) < TOL_FIELD
    km, pw = hd.measure_spectrum(c.q0, 50)
    kmo, pwo = c.oracle.measure_spectrum(c.q0, 50)
    assert np.allclose(km, kmo, rtol=1e-13) and np.allclose(pw, pwo, rtol=1e-12, atol=1e-12 * pwo.max())
    hd.close()


def test_cpp_layer_throws_like_the_reference():
    """The reference's runtime_error sites arrive as C++ exceptions (here: caught by the extern "C" hook)."""
    from barcode_amd.shim import ShimError, ShimHamil
    c = Case(Nx=16, likelihood=1)
    hd = ShimHamil(c.p, **c.arrays())
    hd.numerical.mass_type = 7  # struct_hamil.h:309-312
    with pytest.raises(ShimError, match="mass_type"):
        hd.gradient_psi(c.q0)
    hd2 = ShimHamil(c.p, **c.arrays())
    hd2.numerical.mk = 1        # HMC_models.cc:316-319: calc_h 2 needs the SPH kernel
    with pytest.raises(ShimError, match="SPH"):
        hd2.gradient_psi(c.q0)
    hd.close()
    hd2.close()


def test_cpp_hamiltonian_mc_loop_matches_the_python_mirror():
    """bchmc_shim::HamiltonianMC (C++, HMC.cc:431-511 on the resident chain) consumes the uniform stream and takes the
    accept/reject decisions exactly like barcode_amd.hamil.HamiltonianMC; same device momentum draw (seed, attempt)."""
    from barcode_amd import hamil
    from barcode_amd.shim import ShimHamil
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    eps_fac = 12 * c.eps  # large steps: some attempts are rejected

    def stream():
        rng = np.random.default_rng(123)
        return lambda: float(rng.random())

    hd = ShimHamil(c.p, N_eps_fac=4.0, eps_fac=eps_fac, **c.arrays())
    hd.chain_set_state(c.q0)
    u = stream()
    logs_cpp = [hd.HamiltonianMC(u, seed=5, itmax=30) for _ in range(3)]
    x_cpp = hd.chain_get_state()
    hd.close()

    hp = hamil.HamilData(c.p, N_eps_fac=4.0, eps_fac=eps_fac, **c.arrays())
    hp.engine.chain_set_state(c.q0)
    u = stream()
    logs_py = [hamil.HamiltonianMC(hp, u, seed=5, itmax=30) for _ in range(3)]
    x_py = hp.engine.chain_get_state()
    hp.engine.close()

    assert [len(l) for l in logs_cpp] == [len(l) for l in logs_py]
    for lc, lp in zip(logs_cpp, logs_py):
        for a, b in zip(lc, lp):
            assert a["accepted"] == b["accepted"] and a["Neps"] == b["Neps"] and a["steps_done"] == b["steps_done"]
            assert np.isclose(a["epsilon"], b["epsilon"], rtol=1e-15)
            # rejected attempts with these large steps are unstable trajectories (dH ~ 1e5): compare relative to the
            # largest energy term involved
            scale = max(abs(b[k]) for k in ("H_kin_i", "psi_prior_i", "psi_likeli_i", "H_kin_f", "psi_prior_f",
                                             "psi_likeli_f", "dH"))
            assert abs(a["dH"] - b["dH"]) <= 1e-8 * scale
        assert lc[-1]["accepted"] or len(lc) == 30
    assert rel_l2(x_cpp, x_py) < 1e-11


def test_mass_changed_keeps_the_carried_gradient_and_uses_the_new_mass(monkeypatch):
    """HamiltonianMC rewrites mass_f every sample (HMC.cc:400-423): mass_changed() re-uploads the two mass arrays only,
    and the chain's carried gradient_psi / -log L (which do not involve the mass) survive it.  The samples must be the
    ones a chain that re-evaluates everything produces."""
    from barcode_amd.shim import ShimHamil
    c = Case(Nx=16, likelihood=1, rsd_model=1)

    def run():
        arrays = c.arrays()
        arrays["mass_f"] = arrays["mass_f"].copy()
        rng = np.random.default_rng(7)
        u = lambda: float(rng.random())  # noqa: E731
        hd = ShimHamil(c.p, N_eps_fac=3.0, eps_fac=4 * c.eps, **arrays)
        hd.chain_set_state(c.q0)
        logs = []
        for s in range(3):
            logs += hd.HamiltonianMC(u, seed=11, itmax=20)
            arrays["mass_f"] *= 1.5          # the caller-owned array HAMIL_DATA::mass_f points at
            hd.mass_changed()
        x = hd.chain_get_state()
        hd.close()
        return logs, x

    logs_a, x_a = run()
    monkeypatch.setenv("BCHMC_NO_FORCE_CARRY", "1")
    logs_b, x_b = run()
    assert len(logs_a) == len(logs_b) >= 3
    for a, b in zip(logs_a, logs_b):
        assert a["accepted"] == b["accepted"] and a["Neps"] == b["Neps"]
        scale = max(abs(b[k]) for k in ("H_kin_i", "psi_prior_i", "psi_likeli_i", "H_kin_f", "psi_prior_f", "psi_likeli_f"))
        assert abs(a["dH"] - b["dH"]) <= 1e-9 * scale
    assert rel_l2(x_a, x_b) < 1e-11
