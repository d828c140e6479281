"""BASELINE-size checks on the GPU.  Every BASELINE configuration is compared with the (OpenMP) oracle AT ITS OWN
SIZE -- 64^3 x 10 steps (config 1), 128^3 Poissonian x 50 steps (config 2), 256^3 RSD x 3 steps + energies (config 3:
the kernel instantiations bench.py times), 512^3 RSD with fp32 field arrays x 1 step (config 5) -- plus
size-independent properties at 256^3 / 512^3 (reversibility, planes mode vs 3-D plans, Parseval/real-space energy
agreement, mass conservation, linearity of the prior force, determinism of everything but the atomic scatter)."""
import numpy as np
import pytest

from barcode_amd import inputs
from barcode_amd.params import HamilParams
from tests.util import TOL_ENERGY, TOL_TRAJ_10, Case, rel_l2

pytestmark = pytest.mark.gpu


def test_config1_64cubed_ten_steps_against_oracle():
    """BASELINE config 1: 64^3, Gaussian prior + Zel'dovich, Gaussian likelihood, 10 leapfrog steps, fp64."""
    c = Case(Nx=64, L=200.0, likelihood=1, rsd_model=0)
    c.oracle.close()
    from oracle.oracle import Oracle
    o = Oracle(c.p, omp=True)
    o.set(**c.arrays())
    e = c.engine()
    q1o, p1o, _ = o.Hamiltonian_EoM(c.q0, c.p0, c.eps, 10)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 10)
    assert done == 10
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    dHo, to = o.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    dH, t = e.delta_hamiltonian(c.q0, c.p0, q1o, p1o)
    assert np.all(np.abs(t - to) <= TOL_ENERGY * np.abs(to))
    e.close()


@pytest.mark.parametrize("kw", [dict(likelihood=0, rsd_model=0), dict(likelihood=2, rsd_model=0),
                                dict(likelihood=3, rsd_model=0), dict(likelihood=1, rsd_model=1, calc_h=3),
                                dict(likelihood=1, rsd_model=0, sfmodel=2), dict(likelihood=1, rsd_model=1, mass_type=5)],
                         ids=["poisson", "lognormal", "grf", "calc_h3_rsd", "alpt", "mass5_rsd"])
def test_other_models_at_64_cubed_against_oracle(kw):
    """The non-default likelihoods, the Fourier+TSC force, the ALPT forward model and a real-space mass at 64^3 (256
    tiles, several work items per dense tile, planes mode for the default force path) instead of only at 16^3: a
    5-step trajectory and the energies against the OpenMP oracle."""
    from oracle.oracle import Oracle
    c = Case(Nx=64, L=200.0, **kw)
    c.oracle.close()
    o = Oracle(c.p, omp=True)
    o.set(**c.arrays())
    e = c.engine()
    q1o, p1o, done_o = o.Hamiltonian_EoM(c.q0, c.p0, c.eps, 5)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 5)
    assert done == done_o == 5
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    dHo, to = o.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    dH, t = e.delta_hamiltonian(c.q0, c.p0, q1o, p1o)
    assert np.all(np.abs(t - to) <= TOL_ENERGY * np.abs(to))
    o.close()
    e.close()


@pytest.mark.parametrize("precision", [0, 1], ids=["fp64", "fp32"])
def test_two_tile_step_boundary_at_128_cubed(monkeypatch, precision):
    """k_step_boundary_x2 -- both forward and both inverse x transforms at once on two LDS tiles, every operand of a
    phase requested before the previous phase's transforms run; the default for fp32 fields at 128^3 / 256^3,
    BCHMC_BX_V2=1 for fp64 -- against the one-tile kernel (BCHMC_BX_V1=1) and the 3-D-plan path (BCHMC_NO_PLANES=1):
    a 6-step trajectory (five interior boundaries) at the grid size whose instantiation (256 / 512 threads, 4 elements
    per thread, odd log2 n) the benchmark grids share."""
    c = Case(Nx=128, L=200.0, likelihood=1, rsd_model=1)
    c.oracle.close()
    outs = []
    for env in (dict(BCHMC_BX_V1="1"), dict(BCHMC_BX_V2="1"), dict(BCHMC_NO_PLANES="1")):
        for k in ("BCHMC_BX_V1", "BCHMC_BX_V2", "BCHMC_NO_PLANES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = c.engine(precision=precision)
        q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 6)
        assert done == 6
        outs.append((q1, p1))
        e.close()
    noise = 1e-13 if precision == 0 else 1e-5
    for q1, p1 in outs[1:]:
        assert rel_l2(q1, outs[0][0]) < noise and rel_l2(p1, outs[0][1]) < noise


@pytest.mark.parametrize("precision", [0, 1], ids=["fp64", "fp32"])
def test_128_z_pass_inside_the_binning(monkeypatch, precision):
    """k_ypass + k_zbin_direct<T, 128> (odd log2 n: a radix-2 stage first; two waves per workgroup) against rocFFT's 2-D
    C2R + k_bin_direct (BCHMC_NO_ZBIN=1), with and without a forced overflow of the record segments; the 256^3 and
    512^3 instantiations are covered by test_256_z_pass_inside_the_binning and the 512^3 planes-vs-3-D test."""
    c = Case(Nx=128, L=200.0, likelihood=1, rsd_model=1)
    c.oracle.close()
    monkeypatch.setenv("BCHMC_ZBIN_128", "1")  # 128^3 keeps rocFFT by default (2 % faster there)
    outs = []
    for env in (dict(BCHMC_NO_ZBIN="1"), dict(), dict(BCHMC_SORT_CAP="2048", BCHMC_SORT_CAP_FIXED="1")):
        for k in ("BCHMC_NO_ZBIN", "BCHMC_SORT_CAP", "BCHMC_SORT_CAP_FIXED"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = c.engine(precision=precision)
        q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 5)
        assert done == 5
        outs.append((q1, p1))
        e.close()
    noise = 1e-13 if precision == 0 else 1e-5
    for q1, p1 in outs[1:]:
        assert rel_l2(q1, outs[0][0]) < noise and rel_l2(p1, outs[0][1]) < 10 * noise


def test_config2_128cubed_poisson_fifty_steps_against_oracle():
    """BASELINE config 2 at its real length: 128^3, Gaussian prior + Zel'dovich, Poissonian likelihood, 50 leapfrog
    steps, fp64 (HMC.cc:251-369).  Tolerance re-stated for 50 steps: TOL_TRAJ_50 (see tests/util.py: the measured
    growth of a 1e-13 perturbation of q0 over these 50 steps in the oracle itself, times the 10-step tolerance's
    safety margin)."""
    from tests.util import TOL_TRAJ_50
    c = Case(Nx=128, L=200.0, likelihood=0, rsd_model=0)
    c.oracle.close()
    from oracle.oracle import Oracle
    o = Oracle(c.p, omp=True)
    o.set(**c.arrays())
    e = c.engine()
    q1o, p1o, done_o = o.Hamiltonian_EoM(c.q0, c.p0, c.eps, 50)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 50)
    assert done == done_o == 50
    assert rel_l2(q1, q1o) < TOL_TRAJ_50 and rel_l2(p1, p1o) < TOL_TRAJ_50
    dHo, to = o.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    dH, t = e.delta_hamiltonian(c.q0, c.p0, q1o, p1o)
    assert np.all(np.abs(t - to) <= TOL_ENERGY * np.abs(to))
    o.close()
    e.close()


@pytest.fixture(scope="module")
def big():
    """BASELINE config 3 at full size: 256^3, Zel'dovich + plane-parallel RSD, Gaussian likelihood."""
    from barcode_amd.engine import Engine
    p = HamilParams(Nx=256, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
    f = inputs.make_fields(p)
    e = Engine(p)
    e.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], nobs=np.zeros(p.N), window=np.ones(p.N),
             noise=np.ones(p.N))
    e.forward(f["truth"], 1)
    dX = e.fetch("deltaX").reshape((p.Nx,) * 3)
    window, noise, nobs = inputs.mock_observations(p, dX)
    e.upload(window=window, noise=noise, nobs=nobs)
    yield p, f, e, dX
    e.close()


def test_config3_256cubed_eight_steps_and_energies_against_oracle(big):
    """BASELINE config 3 at the benchmarked size, against the oracle (OpenMP build, ~20 s of host time): the
    instantiations bench.py times -- k_step_boundary_x<double, 512, 4>, chunk = 2048, padded rows nhp = 136, 16384
    tiles with 64-bit record offsets -- on an 8-step trajectory (planes mode throughout: the force evaluation before
    the first step and the first step run k_step_boundary_x<BX_FIRST> / <BX_LAST>, the six boundaries between steps the
    interior variant, the last step <BX_LAST>; HMC.cc:251-369; 8 = the reference's largest default Neps, HMC.cc:260)
    and delta_Hamiltonian (HMC.cc:209-248)."""
    from oracle.oracle import Oracle
    p, f, e, dX = big
    window, noise, nobs = inputs.mock_observations(p, dX)
    o = Oracle(p, omp=True)
    o.set(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)
    eps = 0.5 * p.eps_heuristic()  # the bench's step size
    q1o, p1o, done_o = o.Hamiltonian_EoM(f["q0"], f["p0"], eps, 8)
    q1, p1, done = e.leapfrog(f["q0"], f["p0"], eps, 8)
    assert done == done_o == 8
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    dHo, to = o.delta_Hamiltonian(f["q0"], f["p0"], q1o, p1o)
    dH, t = e.delta_hamiltonian(f["q0"], f["p0"], q1o, p1o)
    assert np.all(np.abs(t - to) <= TOL_ENERGY * np.abs(to))
    # the host-array pair as HamiltonianMC issues it (energies taken from the trajectory's own pass)
    dH2, t2 = e.delta_hamiltonian(f["q0"], f["p0"], q1, p1)
    assert np.all(np.abs(t2 - to) <= 10 * TOL_ENERGY * np.abs(to))
    # the force itself and the forward model's density at this size
    g = e.gradient(f["q0"])
    g_o = o.gradient_psi(f["q0"])[0]
    assert rel_l2(g, g_o) < 1e-11
    assert rel_l2(e.fetch("deltaX"), o.get("deltaX")) < 1e-12
    o.close()


def test_256_mass_conservation_and_overdensity(big):
    p, f, e, dX = big
    rho = e.fetch("rho")
    # every particle deposits sum_cells W ~ 1/d^3 (kernel sampled at cell centres): total within a per cent of N/d^3
    assert abs(rho.sum() * p.d ** 3 / p.N - 1.0) < 0.02
    assert rho.min() >= 0.0
    assert abs(dX.mean()) < 1e-12          # overdens() removes the mean exactly
    assert dX.min() >= -1.0


def test_256_reversibility(big):
    """Leapfrog is time-reversible: integrate, flip p, integrate back -> the start, to round-off
    (amplified by the trajectory's own sensitivity, hence 1e-8)."""
    p, f, e, _ = big
    eps = 0.25 * p.eps_heuristic()
    q1, p1, done = e.leapfrog(f["q0"], f["p0"], eps, 5)
    assert done == 5
    q2, p2, done = e.leapfrog(q1, -p1, eps, 5)
    assert rel_l2(q2, f["q0"]) < 1e-8
    assert rel_l2(-p2, f["p0"]) < 1e-8


def test_256_planes_mode_is_the_3d_plan_trajectory(big, monkeypatch):
    """At the BASELINE size the interior step boundaries run in planes mode (2-D rocFFT plans + k_step_boundary_x with
    512-thread workgroups); the same trajectory with the batched 3-D plans (BCHMC_NO_PLANES=1) must agree to the
    scatter's summation noise.  Beyond the oracle's reach; the small-grid variants of both paths are checked against
    the oracle in tests/test_gpu_parity.py."""
    from barcode_amd.engine import Engine
    p, f, e, dX = big
    eps = 0.5 * p.eps_heuristic()
    q1, p1, done = e.leapfrog(f["q0"], f["p0"], eps, 6)
    monkeypatch.setenv("BCHMC_NO_PLANES", "1")
    window, noise, nobs = inputs.mock_observations(p, dX)
    e2 = Engine(p)
    e2.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)
    q2, p2, done2 = e2.leapfrog(f["q0"], f["p0"], eps, 6)
    e2.close()
    assert done == done2 == 6
    assert rel_l2(q1, q2) < 1e-13 and rel_l2(p1, p2) < 1e-12


@pytest.mark.parametrize("precision", [0, 1], ids=["fp64", "fp32"])
def test_256_z_pass_inside_the_binning(big, monkeypatch, precision):
    """Interior steps at 256^3 run the engine's own y pass (k_ypass) and the z pass inside the binning kernel
    (k_zbin_direct: two real rows per complex LDS transform, pair-wise 64-bit counter atomics) instead of rocFFT's 2-D
    C2R + k_bin_direct.  Same trajectory as with BCHMC_NO_ZBIN=1 to the scatter's summation noise -- also when a record
    segment overflows inside an interior step (BCHMC_SORT_CAP=4096: 512 slots per segment for populations of ~900), where
    k_zbin_direct<PSI_ONLY> has to hand the displacements to the two-pass fallback sort; the fused kernel on the last
    step is not used (its positions may be fetched), which the deltaX comparison checks."""
    from barcode_amd.engine import Engine
    p, f, _, dX = big
    window, noise, nobs = inputs.mock_observations(p, dX)
    eps = 0.5 * p.eps_heuristic()
    outs = []
    for env in (dict(BCHMC_NO_ZBIN="1"), dict(), dict(BCHMC_SORT_CAP="4096", BCHMC_SORT_CAP_FIXED="1")):
        for k in ("BCHMC_NO_ZBIN", "BCHMC_SORT_CAP", "BCHMC_SORT_CAP_FIXED"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e2 = Engine(p, precision=precision)
        e2.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)
        q1, p1, done = e2.leapfrog(f["q0"], f["p0"], eps, 5)
        assert done == 5
        outs.append((q1, p1, e2.fetch("deltaX"), e2.fetch("posz")))
        e2.close()
    noise_q, noise_p = (1e-13, 1e-12) if precision == 0 else (1e-5, 1e-5)
    for q1, p1, dx, pz in outs[1:]:
        assert rel_l2(q1, outs[0][0]) < noise_q and rel_l2(p1, outs[0][1]) < noise_p
        assert rel_l2(dx, outs[0][2]) < (1e-11 if precision == 0 else 1e-4)
        assert rel_l2(pz, outs[0][3]) < (1e-13 if precision == 0 else 1e-5)


def test_256_z_pass_inside_the_binning_alpt_and_deterministic(big, monkeypatch):
    """The same comparison for the two other users of the planes-space C2R at 256^3: the ALPT forward model (its mix
    kernel leaves Psi^ in planes space like the Zel'dovich boundary does) and the deterministic mode (k_zbin_direct
    clears the fixed-point density instead of rho; results must be bit-identical to the rocFFT path's only up to the
    transform's own rounding, so the comparison is at round-off, not exact)."""
    from barcode_amd.engine import Engine
    p, f, _, dX = big
    window, noise, nobs = inputs.mock_observations(p, dX)
    cases = [(HamilParams(Nx=256, L=200.0, likelihood=1, rsd_model=0, sfmodel=2), {}, 0.2),
             (p, dict(BCHMC_DETERMINISTIC="1"), 0.5)]
    for pc, extra, scale in cases:
        eps = scale * pc.eps_heuristic()
        outs = []
        for nozbin in ("1", "0"):
            monkeypatch.setenv("BCHMC_NO_ZBIN", nozbin)
            for k, v in extra.items():
                monkeypatch.setenv(k, v)
            e2 = Engine(pc)
            e2.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)
            q1, p1, done = e2.leapfrog(f["q0"], f["p0"], eps, 4)
            assert done == 4
            outs.append((q1, p1))
            e2.close()
        for k in extra:
            monkeypatch.delenv(k, raising=False)
        assert rel_l2(outs[1][0], outs[0][0]) < 1e-13 and rel_l2(outs[1][1], outs[0][1]) < 1e-11


def test_256_runaway_guard_with_the_fused_z_pass(big, monkeypatch):
    """HMC.cc:360-364 at the benchmark size: a momentum of 1e60 trips the guard after the first step; the steps that were
    already enqueued run their y / z passes and the binning on whatever the stopped boundary kernels left in Ck (huge or
    non-finite displacements: positions are folded into the box or dropped, never used as indices) and must leave the
    returned state alone -- the same state as on the rocFFT path, and a normal trajectory afterwards is unaffected."""
    from barcode_amd.engine import Engine
    p, f, _, dX = big
    window, noise, nobs = inputs.mock_observations(p, dX)
    eps = 0.5 * p.eps_heuristic()
    p_bad = f["p0"].copy().ravel()
    p_bad[0] = 1e60
    outs = []
    for nozbin in ("1", "0"):
        monkeypatch.setenv("BCHMC_NO_ZBIN", nozbin)
        e2 = Engine(p)
        e2.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)
        qb, pb, doneb = e2.leapfrog(f["q0"], p_bad, 1e-6, 6)
        q1, p1, done = e2.leapfrog(f["q0"], f["p0"], eps, 4)
        assert doneb == 1 and done == 4
        assert np.all(np.isfinite(qb)) and np.all(np.isfinite(q1))
        outs.append((qb, pb, q1, p1))
        e2.close()
    for a, b in zip(outs[0], outs[1]):
        assert rel_l2(b, a) < 1e-12


def test_256_energy_terms_against_real_space_evaluation(big):
    """The engine evaluates 1/2 x^T A x by Parseval in k-space; compare with the reference's real-space form
    0.5 * sum(x * IFFT[w FFT x]) (HMC.cc:101-115, gaussian.cpp:24-32) computed with numpy, and the Gaussian
    -log L from the fetched deltaX (gaussian_independent.cpp:82-89)."""
    p, f, e, _ = big
    K, prior, like = e.energies(f["q0"], f["p0"])
    n = p.Nx
    normFS = p.L ** 3 / p.N

    def quad(x, spec):
        s = spec[:, :, : n // 2 + 1]
        w = np.zeros_like(s)
        np.divide(normFS, s, out=w, where=s > 0)
        y = np.fft.irfftn(np.fft.rfftn(x) * w, s=(n, n, n), axes=(0, 1, 2))
        return 0.5 * float(np.sum(x * y))

    assert abs(K - quad(f["p0"], f["mass_f"])) <= 1e-10 * abs(K)
    assert abs(prior - quad(f["q0"], f["signal_PS"])) <= 1e-10 * abs(prior)
    dX = e.fetch("deltaX")
    nobs, noise = e.fetch("nobs"), e.fetch("noise")
    lam = p.rho_c * (1.0 + dX)
    ref = float(np.sum(np.where(lam > 0, 0.5 * ((lam - nobs) / noise) ** 2, 0.0)))
    assert abs(like - ref) <= 1e-10 * abs(ref)


def test_256_prior_force_is_linear_and_likelihood_factor_scales(big):
    p, f, e, _ = big
    e.gradient(f["q0"])
    gp1, gl1 = e.fetch("grad_prior"), e.fetch("grad_like")
    e.gradient(2.0 * f["q0"])
    gp2 = e.fetch("grad_prior")
    assert rel_l2(gp2, 2.0 * gp1) < 1e-13
    # repeated evaluation: identical up to the atomic-add order of the scatter
    e.gradient(f["q0"])
    assert rel_l2(e.fetch("grad_like"), gl1) < 1e-12
    assert np.array_equal(e.fetch("grad_prior"), gp1)


def test_256_small_step_conserves_energy(big):
    p, f, e, _ = big
    eps = 0.02 * p.eps_heuristic()
    q1, p1, _ = e.leapfrog(f["q0"], f["p0"], eps, 4)
    dH, terms = e.delta_hamiltonian(f["q0"], f["p0"], q1, p1)
    assert abs(dH) < 1e-4 * abs(terms[:3].sum())


def test_hundred_step_trajectory_against_oracle():
    """rel-L2 of (q1, p1) <= 1e-9 at 100 leapfrog steps (BASELINE config 3 length), here on a 32^3 grid
    the oracle integrates in a few seconds, with Zel'dovich + plane-parallel RSD."""
    from oracle.oracle import Oracle
    c = Case(Nx=32, L=100.0, likelihood=1, rsd_model=1, sfmodel=2, eps_scale=0.05)
    c.oracle.close()
    o = Oracle(c.p, omp=True)
    o.set(**c.arrays())
    e = c.engine()
    q1o, p1o, done_o = o.Hamiltonian_EoM(c.q0, c.p0, c.eps, 100)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 100)
    assert done == done_o == 100
    assert rel_l2(q1, q1o) < 1e-9 and rel_l2(p1, p1o) < 1e-9
    dHo, to = o.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    dH, t = e.delta_hamiltonian(c.q0, c.p0, q1, p1)
    assert np.all(np.abs(t - to) <= 1e-8 * np.abs(to))
    e.close()


@pytest.fixture(scope="module")
def huge():
    """BASELINE config 5: 512^3, Zel'dovich + plane-parallel RSD, Gaussian likelihood, fp32 field arrays."""
    from barcode_amd.engine import Engine
    p = HamilParams(Nx=512, L=200.0, likelihood=1, rsd_model=1, sfmodel=2)
    f = inputs.make_fields(p)
    e = Engine(p, precision=1)
    e.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], nobs=np.zeros(p.N), window=np.ones(p.N),
             noise=np.ones(p.N))
    e.forward(f["truth"], 1)
    dX = e.fetch("deltaX").reshape((p.Nx,) * 3)
    window, noise, nobs = inputs.mock_observations(p, dX)
    del dX
    e.upload(window=window, noise=noise, nobs=nobs)
    yield p, f, e, (window, noise, nobs)
    e.close()


def test_config5_512cubed_fp32_one_step_against_fp64_oracle(huge):
    """Config 5 against the oracle (fp64, OpenMP build, two force evaluations at 512^3): one leapfrog step and the
    energies at the re-stated fp32 tolerance (rel-L2 <= 1e-4 on (q1, p1), energies <= 1e-5).  The precision
    switch upstream is define_opt.h:46-50."""
    from oracle.oracle import Oracle
    p, f, e, (window, noise, nobs) = huge
    o = Oracle(p, omp=True)
    o.set(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)
    eps = 0.5 * p.eps_heuristic()
    q1o, p1o, done_o = o.Hamiltonian_EoM(f["q0"], f["p0"], eps, 1)
    q1, p1, done = e.leapfrog(f["q0"], f["p0"], eps, 1)
    assert done == done_o == 1
    assert rel_l2(q1, q1o) < 1e-4 and rel_l2(p1, p1o) < 1e-4
    K, prior, like = e.energies(q1o, p1o)
    Ko = o.kinetic_term(p1o)
    prior_o, like_o = o.psi(q1o)
    o.close()
    for a, b in ((K, Ko), (prior, prior_o), (like, like_o)):
        assert abs(a - b) <= 1e-5 * abs(b)


def test_512_fp32_planes_mode_reversibility_and_mass(huge, monkeypatch):
    """Size-independent properties at 512^3 with fp32 fields: the planes-mode trajectory (k_step_boundary_x<float,
    1024, 8>) equals the 3-D-plan one to fp32 round-off, the integrator is reversible to the fp32 level, the scatter
    conserves mass."""
    from barcode_amd.engine import Engine
    p, f, e, (window, noise, nobs) = huge
    eps = 0.5 * p.eps_heuristic()
    q1, p1, done = e.leapfrog(f["q0"], f["p0"], eps, 4)
    assert done == 4
    rho = e.fetch("rho")
    assert abs(rho.sum() * p.d ** 3 / p.N - 1.0) < 0.02 and rho.min() >= 0.0
    del rho
    q2, p2, done = e.leapfrog(q1, -p1, eps, 4)
    assert rel_l2(q2, f["q0"]) < 1e-4 and rel_l2(-p2, f["p0"]) < 1e-4
    del q2, p2
    monkeypatch.setenv("BCHMC_NO_PLANES", "1")
    e2 = Engine(p, precision=1)
    e2.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], window=window, noise=noise, nobs=nobs)
    q3, p3, done3 = e2.leapfrog(f["q0"], f["p0"], eps, 4)
    e2.close()
    assert done3 == 4
    assert rel_l2(q1, q3) < 1e-5 and rel_l2(p1, p3) < 1e-4
