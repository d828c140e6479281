"""CPU tests of the oracle itself (no GPU): two independent restatements must agree, and the force must be
the derivative of the energy.  PARITY UNPINNED: the reference holds no golden vectors for this path and
cannot be built here, so these self-consistency checks plus tests/golden (oracle-generated) are what
anchors the oracle.  See oracle/README.md."""
import numpy as np
import pytest

from oracle.np_restatement import NpHamil
from tests.util import Case, rel_l2

CONFIGS = [
    dict(likelihood=1, rsd_model=0, calc_h=0),
    dict(likelihood=0, rsd_model=0, calc_h=0),
    dict(likelihood=2, rsd_model=0, calc_h=0),
    dict(likelihood=1, rsd_model=0),
    dict(likelihood=1, rsd_model=1),
    dict(likelihood=0, rsd_model=0),
    dict(likelihood=2, rsd_model=0),
    dict(likelihood=1, rsd_model=0, sfmodel=2),            # ALPT forward model (Lag2Eul_non_zeldovich)
    dict(likelihood=0, rsd_model=0, sfmodel=2, kth=2.0),
]


def _np(case):
    return NpHamil(case.p, case.signal_PS, case.mass_f, case.nobs, case.noise, case.window, mass_r=case.mass_r)


@pytest.mark.parametrize("kw", CONFIGS)
def test_c_oracle_matches_numpy_restatement(kw):
    c = Case(Nx=8, **kw)
    n = _np(c)
    o = c.oracle
    dX, px, py, pz = o.Lag2Eul(c.truth)
    dXn, posn = n.lag2eul(c.truth, bool(c.p.rsd_model))
    assert rel_l2(dX, dXn) < 1e-13
    for a, b in zip((px, py, pz), posn):
        assert rel_l2(a, b) < 1e-14
    g, gp, gl = o.gradient_psi(c.q0)
    gn, gpn, gln = n.gradient_psi(c.q0)
    assert rel_l2(gp, gpn) < 1e-13
    assert rel_l2(gl, gln) < 1e-12
    assert rel_l2(g, gn) < 1e-12
    q1, p1, done = o.Hamiltonian_EoM(c.q0, c.p0, c.eps, 3)
    q1n, p1n = n.leapfrog(c.q0, c.p0, c.eps, 3)
    assert done == 3
    assert rel_l2(q1, q1n) < 1e-11 and rel_l2(p1, p1n) < 1e-11
    dH, terms = o.delta_Hamiltonian(c.q0, c.p0, q1, p1)
    ref = [n.kinetic(c.p0), n.log_prior(c.q0), n.log_like(c.q0), n.kinetic(p1n), n.log_prior(q1n), n.log_like(q1n)]
    assert np.allclose(terms, ref, rtol=1e-10)
    assert np.isclose(dH, sum(ref[3:]) - sum(ref[:3]), rtol=1e-8, atol=1e-8)


def test_stencil_is_the_81_cell_hull():
    """SURVEY M2 / SPH_kernel.cpp:62-102: for h = d the stencil has 81 cells in 21 (i, j) columns."""
    c = Case(Nx=8)
    st = c.oracle.stencil()
    assert len(st) == 81
    assert len({(i, j) for i, j, _ in st}) == 21
    assert np.abs(st).max() == 2


def test_fft_roundtrip_and_against_numpy():
    c = Case(Nx=8)
    o = c.oracle
    # conv with corr = V/N everywhere is the identity: IFFT[FFT[x]] (HMC_help.cc:41-58 with C = normFS)
    corr = np.full(c.p.N, c.p.L ** 3 / c.p.N)
    assert rel_l2(o.convolveInvCorrFuncWithSignal(c.q0, corr), c.q0) < 1e-14
    vx, vy, vz = o.theta2vel(c.q0)
    nvx, nvy, nvz = _np(c).theta2vel(c.q0.reshape((8,) * 3))
    for a, b in zip((vx, vy, vz), (nvx, nvy, nvz)):
        assert rel_l2(a, b) < 1e-13


@pytest.mark.parametrize("n", [6, 12])
def test_non_power_of_two_grid(n):
    """The oracle's FFT falls back to a plain DFT for other sizes; it must still match numpy."""
    c = Case(Nx=n)
    g, _, _ = c.oracle.gradient_psi(c.q0)
    gn, _, _ = _np(c).gradient_psi(c.q0)
    assert rel_l2(g, gn) < 1e-12


@pytest.mark.parametrize("kw", [dict(likelihood=1), dict(likelihood=1, rsd_model=1), dict(likelihood=0),
                                dict(likelihood=2)])
def test_force_is_gradient_of_energy(kw):
    """d(-log L)/dq_i by central differences vs likelihood_grad_log_like (HMC_models.cc:377-471).
    The reference's force neglects the dependence of the mean density on q and zeroes Nyquist modes,
    so agreement is approximate; a sign or normalisation slip would be off by O(1).

    Reference quirk kept bug-for-bug: the Poissonian partial (poissonian.cpp:30) is +d(-log L)/d delta_x
    while the Gaussian and log-normal ones (gaussian_independent.cpp:37-38, lognormal_independent.cpp:51)
    are MINUS that, and all three go through the same `zeldovich_norm = -1` (HMC_models.cc:460).  So the
    reference's Poissonian force is minus the gradient of its own log_like; the test pins that sign."""
    c = Case(Nx=8, **kw)
    o = c.oracle
    q = 0.3 * c.q0.ravel()  # gentle field: no shell crossing pile-ups
    gl = o.likelihood_grad_log_like(q)
    idx = np.argsort(-np.abs(gl))[:6]
    hstep = 1e-5
    for i in idx:
        e = np.zeros_like(q)
        e[i] = hstep
        fd = (o.log_like(q + e) - o.log_like(q - e)) / (2 * hstep)
        if c.p.likelihood == 0:
            fd = -fd
        # log-normal: the reference's partial is d/d ln(1 + delta_x), not d/d delta_x (lognormal_independent.cpp:44),
        # so its force is only roughly the gradient; the sign and magnitude must still agree.
        tol = 0.3 if c.p.likelihood == 2 else 0.05
        assert abs(fd - gl[i]) <= tol * abs(gl[i]) + 1e-6, (i, fd, gl[i])


def test_prior_force_is_gradient_of_prior_energy():
    c = Case(Nx=8)
    o = c.oracle
    gp = o.grad_log_prior(c.q0)
    for i in (0, 17, 300):
        e = np.zeros(c.p.N)
        e[i] = 1e-4
        fd = (o.log_prior(c.q0.ravel() + e) - o.log_prior(c.q0.ravel() - e)) / 2e-4
        assert np.isclose(fd, gp[i], rtol=1e-7, atol=1e-9)


def test_leapfrog_is_reversible_and_conserves_energy():
    c = Case(Nx=8)
    o = c.oracle
    eps = 0.02 * c.p.eps_heuristic()
    q1, p1, _ = o.Hamiltonian_EoM(c.q0, c.p0, eps, 4)
    q2, p2, _ = o.Hamiltonian_EoM(q1, -p1, eps, 4)
    assert rel_l2(q2, c.q0) < 1e-10 and rel_l2(-p2, c.p0) < 1e-10
    dH, terms = o.delta_Hamiltonian(c.q0, c.p0, q1, p1)
    assert abs(dH) < 1e-3 * abs(sum(terms[:3]))


def test_mass_is_conserved_by_sph_assignment():
    """Sum of rho * d^3 stays within a few per cent of N (W_4 sampled at cell centres), and NGP/CIC/TSC exactly."""
    c = Case(Nx=8)
    o = c.oracle
    _, px, py, pz = o.Lag2Eul(c.truth)
    d3 = c.p.d ** 3
    assert abs(o.getDensity(3, px, py, pz).sum() * d3 / c.p.N - 1) < 0.05
    for mk in (0, 1, 2):
        assert abs(o.getDensity(mk, px, py, pz).sum() / c.p.N - 1) < 1e-12


def test_runaway_guard_stops_the_trajectory():
    """HMC.cc:360-364: |p[0]| > 1e50 ends the loop after the current step."""
    c = Case(Nx=8)
    p0 = c.p0.copy()
    p0[0] = 1e60
    _, _, done = c.oracle.Hamiltonian_EoM(c.q0, p0, 1e-6, 5)
    assert done == 1


def test_error_codes():
    from oracle.oracle import OracleError
    c = Case(Nx=8, mk=1)  # CIC with calc_h = 2 -> HMC_models.cc:316-319
    with pytest.raises(OracleError) as ei:
        c.oracle.gradient_psi(c.q0)
    assert ei.value.code == 2
    with pytest.raises(OracleError) as ei:
        Case(Nx=8, mass_type=7)
    assert ei.value.code == 4


# ---- f-3: the pieces of Lag2Eul_non_zeldovich (Lag2Eul.cc:138-312) ----------------------------------------
def test_alpt_pieces_against_independent_numpy_forms():
    c = Case(Nx=8, likelihood=1, rsd_model=0, sfmodel=2)
    n, o, p = _np(c), c.oracle, c.p
    q = c.q0.reshape(n.shape)
    # PoissonSolver: the spectral Laplacian of the potential gives back delta minus its mean
    phi = o.PoissonSolver(q).reshape(n.shape)
    lap = n.c2r(-n.ksq * n.r2c(phi))
    assert rel_l2(lap, q - q.mean()) < 1e-13
    # calc_m2v_mem on a plane wave: only the xx term survives, so delta(2) = 0
    x = (np.arange(p.Nx) + 0.5) * p.d
    wave = np.sin(2 * np.pi * x / p.L)[:, None, None] * np.ones(n.shape)
    assert np.abs(o.calc_m2v_mem(wave)).max() < 1e-14
    # ... and against the np.roll form on a random field
    g = [n.gradfindif(phi, a) for a in range(3)]
    xx, xy, xz = (n.gradfindif(g[0], a) for a in range(3))
    yy, yz, zz = n.gradfindif(g[1], 1), n.gradfindif(g[1], 2), n.gradfindif(g[2], 2)
    assert rel_l2(o.calc_m2v_mem(phi), xx * yy - xy * xy + xx * zz - xz * xz + yy * zz - yz * yz) < 1e-13
    # kernelcomp: Gaussian on the full grid, normalised so that the k = 0 mode is 1; convcomp keeps constants
    K = o.kernelcomp(p.kth).reshape(n.shape)
    assert abs(K[0, 0, 0] - 1.0) < 1e-14
    assert rel_l2(K[:, :, : p.Nx // 2 + 1], n.alpt_kernel(p.kth)) < 1e-14
    assert rel_l2(o.convcomp(np.full(p.N, 3.5), p.kth), np.full(p.N, 3.5)) < 1e-14
    assert rel_l2(o.convcomp(q, p.kth), n.c2r(n.r2c(q) * n.alpt_kernel(p.kth))) < 1e-13
    # theta2velcomp is theta2vel component by component
    v = o.theta2vel(q)
    for comp in (1, 2, 3):
        assert rel_l2(o.theta2velcomp(q, comp), v[comp - 1]) < 1e-14
    # cellboundcomp: mean with the (i-1, j-1, k-1) neighbour, periodic
    a = np.arange(p.N, dtype=np.float64).reshape(n.shape)
    assert np.array_equal(o.cellboundcomp(a).reshape(n.shape), 0.5 * (a + np.roll(a, (1, 1, 1), (0, 1, 2))))
    # the whole displacement, two independent restatements
    for ours, theirs in zip(n.alpt_displacement(q), o.alpt_displacement(q)):
        assert rel_l2(ours, theirs) < 1e-13


def test_alpt_displacement_has_the_reference_sign_and_zeldovich_limit():
    """Finding M11: Lag2Eul_non_zeldovich feeds +D1 delta (minus the divergence) to the velocity kernel where
    Lag2Eul_zeldovich feeds -D1 delta (Lag2Eul.cc:88 vs 199-200, 212-226), so for small delta the ALPT displacement
    is MINUS the Zel'dovich one (up to the cell-boundary average).  Restated as is."""
    c = Case(Nx=8, likelihood=1, rsd_model=0, sfmodel=2)
    o, p = c.oracle, c.p
    q = 1e-4 * c.q0
    za = o.theta2vel(-p.D1 * q.ravel())
    alpt = o.alpt_displacement(q)
    for a, z in zip(alpt, za):
        assert rel_l2(a, -o.cellboundcomp(z)) < 1e-3


def test_measure_spectrum_against_full_grid_numpy():
    """field_statistics.cpp:20-90 restated on the half-complex transform == the literal full-grid loop in numpy."""
    c = Case(Nx=8)
    n, L, nb = 8, c.p.L, 20
    km, pw = c.oracle.measure_spectrum(c.q0, nb)
    F = np.fft.fftn(c.q0.reshape(n, n, n))
    i = np.arange(n)
    k1 = np.where(i <= n // 2, 2 * np.pi / L * i, -2 * np.pi / L * (n - i))
    kt = np.sqrt(k1[:, None, None] ** 2 + k1[None, :, None] ** 2 + k1[None, None, :] ** 2)
    dk = np.sqrt(3.0) * 2 * np.pi / L * (n // 2) / nb
    b = (kt / dk).astype(np.int64).ravel()
    ok = b < nb
    cnt = np.bincount(b[ok], minlength=nb)
    ks = np.bincount(b[ok], weights=kt.ravel()[ok], minlength=nb)
    ps = np.bincount(b[ok], weights=(np.abs(F) ** 2).ravel()[ok], minlength=nb)
    has = cnt > 0
    assert np.allclose(km[has], ks[has] / cnt[has], rtol=1e-14) and np.all(km[~has] == 0)
    assert np.allclose(pw[has], ps[has] / cnt[has] * L ** 3 / n ** 6, rtol=1e-13) and np.all(pw[~has] == 0)
    assert cnt.sum() == n ** 3 - 1  # only the |k|max corner mode falls outside (field_statistics.cpp:49-51)
