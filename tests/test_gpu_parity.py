"""Parity of the HIP engine (through the C ABI) with the CPU oracle on identical seeded inputs.

fp64 tolerances (written out in tests/util.py): intermediates of one force evaluation
rel-L2 <= 1e-12, final (q, p) of a 10-step trajectory rel-L2 <= 1e-11, energies rel <= 1e-10.
The engine's scatter uses fp64 hardware atomics (order-dependent in the last bits), hence tolerances
instead of bit-exactness; the reference has the same property under OpenMP (main.cc:86-90).
"""
import numpy as np
import pytest

from tests.util import TOL_ENERGY, TOL_FIELD, TOL_TRAJ_10, Case, rel_l2

pytestmark = pytest.mark.gpu

CONFIGS = [
    dict(likelihood=1, rsd_model=0),                       # BASELINE config 1 (Gaussian, Zel'dovich)
    dict(likelihood=0, rsd_model=0),                       # config 2 (Poissonian)
    dict(likelihood=1, rsd_model=1, sfmodel=2),            # config 3 ("2LPT + RSD" == Zel'dovich + RSD)
    dict(likelihood=2, rsd_model=0),                       # log-normal likelihood
    dict(likelihood=3, rsd_model=0),                       # GRF likelihood
    dict(likelihood=1, rsd_model=0, mass_type=0),          # real-space mass only
    dict(likelihood=1, rsd_model=0, mass_type=5),          # Fourier + real-space mass
    dict(likelihood=1, rsd_model=0, calc_h=1),             # h = partial_f (HMC_models.cc:413-415)
    dict(likelihood=1, rsd_model=0, calc_h=0),             # legacy likelihood_calc_h, spectral gradient
    dict(likelihood=0, rsd_model=0, calc_h=0, mk=1, eps_scale=0.01),  # legacy, finite differences, CIC
    dict(likelihood=2, rsd_model=0, calc_h=0, eps_scale=1e-3),  # legacy, finite differences of log density (stiff)
    dict(likelihood=1, rsd_model=0, calc_h=3),             # Fourier + TSC variant of V (HMC_models_testing.cpp:54-188)
    dict(likelihood=1, rsd_model=1, calc_h=3, sfmodel=2),
    dict(likelihood=1, rsd_model=0, calc_h=1, mk=1),       # CIC forward model (massFunctions.cc:100-164)
    dict(likelihood=0, rsd_model=0, calc_h=1, mk=0),       # NGP (massFunctions.cc:49-98)
    dict(likelihood=1, rsd_model=1, calc_h=1, mk=2),       # TSC (massFunctions.cc:167-364)
    dict(likelihood=1, rsd_model=1, deltaQ_factor=0.9, grad_psi_prior_factor=0.5, grad_psi_likeli_factor=2.0,
         correct_delta=0),                                 # test factors (HMC.cc:170-173, HMC_models.cc:461-468)
    dict(likelihood=1, rsd_model=0, sfmodel=2),            # ALPT forward model in force and energies (Lag2Eul.cc:138-312)
    dict(likelihood=0, rsd_model=0, sfmodel=2, kth=2.0, deltaQ_factor=0.9),
    dict(likelihood=0, rsd_model=1, sfmodel=2, eps_scale=0.01),  # force: Zel'dovich + RSD; Poissonian log_like: ALPT, no RSD
                                                                 # (ill-conditioned at the default step: amplification 3e9)
]


@pytest.fixture(scope="module", params=range(len(CONFIGS)), ids=lambda i: "cfg%d" % i)
def case(request):
    kw = CONFIGS[request.param]
    c = Case(Nx=16, **kw)
    c.e = c.engine()
    yield c
    c.e.close()


def test_forward_model_intermediates(case):
    """theta2vel -> disp_part (-> calc_pos_rsd) -> getDensity_SPH -> overdens."""
    c = case
    if c.p.likelihood == 3:
        pytest.skip("GRF likelihood has no forward model")
    rsd = c.p.rsd_model
    dX, px, py, pz = c.oracle.Lag2Eul(c.truth, rsd=rsd)
    c.e.forward(c.truth, rsd)
    if c.p.sfmodel != 1 and not rsd:
        psi = c.oracle.alpt_displacement(c.truth)
    else:
        psi = c.oracle.theta2vel(-c.p.D1 * c.truth.ravel())
    for name, ref in zip(("psix", "psiy", "psiz"), psi):
        assert rel_l2(c.e.fetch(name), ref) < TOL_FIELD
    for name, ref in zip(("posx", "posy", "posz"), (px, py, pz)):
        assert rel_l2(c.e.fetch(name), ref) < TOL_FIELD
    rho = c.oracle.getDensity(c.p.mk, px, py, pz)
    assert rel_l2(c.e.fetch("rho"), rho) < TOL_FIELD
    assert rel_l2(c.e.fetch("deltaX"), dX) < TOL_FIELD


def test_zero_step_trajectory_returns_the_state():
    """Neps = 0 (the reference draws Neps >= 1, HMC.cc:260, but its loop at :284 handles 0): no kick, no drift."""
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    e = c.engine()
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 0)
    q1o, p1o, done_o = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 0)
    assert done == done_o == 0
    assert rel_l2(q1, q1o) < 1e-14 and rel_l2(p1, p1o) < 1e-14
    assert rel_l2(q1, c.q0) < 1e-14 and rel_l2(p1, c.p0) < 1e-14
    e.close()


def test_positions_wrap_far_outside_the_box():
    """pacman_coordinate (pacman.cpp:20-28) on displacements of many box lengths: the engine folds -L < x < 0 and
    L <= x < 2L with one exact add / subtract and leaves the rest to fmod; every branch must give the oracle's position
    (a wrong fold is off by a box length)."""
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    e = c.engine()
    for scale in (1.0, 40.0, 2000.0):   # displacements << L, ~ L, >> L
        big = c.truth * scale
        _, px, py, pz = c.oracle.Lag2Eul(big, rsd=1)
        e.forward(big, 1)
        L = c.p.L
        for name, ref in zip(("posx", "posy", "posz"), (px, py, pz)):
            got = e.fetch(name)
            assert got.min() >= 0. and got.max() <= L
            # compare on the circle: a position within round-off of a face may legitimately land on either side
            diff = np.abs(got - ref.ravel())
            diff = np.minimum(diff, L - diff)
            assert diff.max() < 1e-9 * L * max(scale, 1.), (name, scale, diff.max())
    e.close()


def test_gradient_psi_and_its_pieces(case):
    c = case
    g, gp, gl = c.oracle.gradient_psi(c.q0)
    gg = c.e.gradient(c.q0)
    assert rel_l2(c.e.fetch("grad_prior"), gp) < TOL_FIELD
    assert rel_l2(c.e.fetch("grad_like"), gl) < 10 * TOL_FIELD
    assert rel_l2(gg, g) < 10 * TOL_FIELD
    if c.p.likelihood != 3:
        dX = c.oracle.get("deltaX")
        assert rel_l2(c.e.fetch("deltaX"), dX) < TOL_FIELD
        pl = c.oracle.partial_f_delta_x_log_like(dX)
        assert rel_l2(c.e.fetch("part_like"), pl) < 10 * TOL_FIELD
        if c.p.calc_h in (2, 3):
            pos = [c.oracle.get(k) for k in ("posx", "posy", "posz")]
            if c.p.calc_h == 2:
                V = c.oracle.likelihood_calc_V_SPH(pl, *pos)
            else:
                V = c.oracle.likelihood_calc_V_SPH_fourier_TSC(pl, *pos)
            for name, ref in zip(("Vx", "Vy", "Vz"), V):
                assert rel_l2(c.e.fetch(name), ref) < 10 * TOL_FIELD


def test_ten_step_trajectory(case):
    """One leapfrog trajectory's final (q, p) on identical inputs (BASELINE north_star correctness gate)."""
    c = case
    q1o, p1o, done_o = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 10)
    q1, p1, done = c.e.leapfrog(c.q0, c.p0, c.eps, 10)
    assert done == done_o == 10
    assert rel_l2(q1, q1o) < TOL_TRAJ_10
    assert rel_l2(p1, p1o) < TOL_TRAJ_10


def test_delta_hamiltonian(case):
    c = case
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 3)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    dH, t = c.e.delta_hamiltonian(c.q0, c.p0, q1o, p1o)
    assert np.all(np.abs(t - to) <= TOL_ENERGY * np.abs(to))
    assert abs(dH - dHo) <= 1e-9 * max(abs(to).max(), 1.0)
    # hd->deltaX holds the LAST evaluation, i.e. psi(signalf) (HMC.cc:225)
    if c.p.likelihood != 3:
        assert rel_l2(c.e.fetch("deltaX"), c.oracle.get("deltaX")) < TOL_FIELD


def test_div_dH_by_N():
    c = Case(Nx=16, div_dH_by_N=1)
    e = c.engine()
    q1, p1, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 2)
    dHo, _ = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1, p1)
    dH, _ = e.delta_hamiltonian(c.q0, c.p0, q1, p1)
    assert abs(dH - dHo) < 1e-9 * max(abs(dHo), 1e-3)
    e.close()


@pytest.mark.parametrize("nx", [8, 32])
def test_other_grid_sizes(nx):
    c = Case(Nx=nx, likelihood=1, rsd_model=1)
    e = c.engine()
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 5)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 5)
    assert done == 5
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    e.close()


def test_window_with_holes_and_empty_cells():
    """Masked cells (window = 0) and zero counts contribute nothing (gaussian_independent.cpp:34-40)."""
    c = Case(Nx=16, likelihood=0, window_zero_fraction=0.4)
    e = c.engine()
    g, _, _ = c.oracle.gradient_psi(c.q0)
    assert rel_l2(e.gradient(c.q0), g) < 10 * TOL_FIELD
    e.close()


def test_runaway_guard_matches_reference_semantics():
    """HMC.cc:360-364: the loop ends after the first step whose |p[0]| exceeds 1e50."""
    c = Case(Nx=16)
    e = c.engine()
    p0 = c.p0.copy().ravel()
    p0[0] = 1e60
    q1o, p1o, done_o = c.oracle.Hamiltonian_EoM(c.q0, p0, 1e-6, 5)
    q1, p1, done = e.leapfrog(c.q0, p0, 1e-6, 5)
    assert done == done_o == 1
    assert rel_l2(p1, p1o) < 1e-10
    # and a normal trajectory afterwards is unaffected by the tripped flag
    _, _, done2 = e.leapfrog(c.q0, c.p0, c.eps, 3)
    assert done2 == 3
    e.close()


@pytest.mark.parametrize("neps", [1, 2, 5, 6])
def test_fused_step_boundary_is_the_unfused_trajectory(monkeypatch, neps):
    """k_step_boundary (second half kick | first half kick + drift + Zel'dovich in one pass, ping-pong buffers) is the
    same arithmetic as the separate k_assemble / k_kick_drift_za kernels (BCHMC_NO_FUSE=1): trajectories agree to the
    order-of-summation noise of the scatter's float atomics, for odd and even numbers of buffer swaps, including a
    trajectory stopped by the runaway guard."""
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    p_bad = c.p0.copy().ravel()
    p_bad[0] = 1e60
    out = []
    for nofuse in ("0", "1"):
        monkeypatch.setenv("BCHMC_NO_FUSE", nofuse)
        e = c.engine()
        q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, neps)
        qb, pb, doneb = e.leapfrog(c.q0, p_bad, 1e-6, neps + 3)
        q2, p2, done2 = e.leapfrog(q1, p1, c.eps, neps)  # state buffers were swapped an odd/even number of times
        out.append((q1, p1, done, qb, pb, doneb, q2, p2, done2))
        e.close()
    f, u = out
    assert f[2] == u[2] == neps and f[8] == u[8] == neps and f[5] == u[5] == 1
    noise = 1e-13
    assert rel_l2(f[0], u[0]) < noise and rel_l2(f[1], u[1]) < noise
    assert rel_l2(f[6], u[6]) < noise and rel_l2(f[7], u[7]) < noise
    # stopped trajectory: rolled back to the end of the step that tripped the guard
    assert rel_l2(f[3], u[3]) < noise and rel_l2(f[4], u[4]) < noise
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, neps)
    assert rel_l2(f[0], q1o) < TOL_TRAJ_10 and rel_l2(f[1], p1o) < TOL_TRAJ_10


def test_epsilon_is_clipped_at_two():
    """HMC.cc:263-264."""
    c = Case(Nx=8, likelihood=3)
    e = c.engine()
    a = e.leapfrog(c.q0, c.p0, 2.0, 1)
    b = e.leapfrog(c.q0, c.p0, 5.0, 1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    e.close()


def test_error_conventions():
    """Same failure conditions as the reference's runtime_errors, as return codes."""
    from barcode_amd.engine import BchmcError, Engine
    from barcode_amd.params import HamilParams
    c = Case(Nx=8, mk=1)  # CIC + calc_h 2: HMC_models.cc:316-319
    with pytest.raises(BchmcError) as ei:
        e = c.engine()
        e.gradient(c.q0)
    assert ei.value.code == 2
    with pytest.raises(BchmcError) as ei:
        Engine(HamilParams(Nx=8, mass_type=7))
    assert ei.value.code == 4
    with pytest.raises(BchmcError) as ei:
        Engine(HamilParams(Nx=8)).gradient(np.zeros(512))  # nothing uploaded
    assert ei.value.code == 9
    with pytest.raises(BchmcError) as ei:
        c2 = Case(Nx=8, rsd_model=1)
        c2.p.planepar = 0
        c2.engine().gradient(c2.q0)
    assert ei.value.code == 3


# ---- fp32 field arrays (BASELINE config 5): storage and particle-mesh arithmetic in float, k-space arithmetic and
# reductions in double.  Re-stated tolerances: rel-L2 <= 1e-4 on q1, p1; energies rel <= 1e-5.
TOL_F32_FIELD = 2e-5
TOL_F32_TRAJ = 1e-4
TOL_F32_ENERGY = 1e-5


@pytest.mark.parametrize("kw", [dict(likelihood=1, rsd_model=1, sfmodel=2), dict(likelihood=1, rsd_model=0),
                                dict(likelihood=3), dict(likelihood=1, mass_type=5)],
                         ids=["gauss_rsd", "gauss", "grf", "mass5"])
def test_fp32_field_mode(kw):
    c = Case(Nx=16, **kw)
    e = c.engine(precision=1)
    g, gp, gl = c.oracle.gradient_psi(c.q0)
    gg = e.gradient(c.q0)
    assert rel_l2(e.fetch("grad_prior"), gp) < TOL_F32_FIELD
    assert rel_l2(gg, g) < 10 * TOL_F32_FIELD
    if c.p.likelihood != 3:
        assert rel_l2(e.fetch("deltaX"), c.oracle.get("deltaX")) < 10 * TOL_F32_FIELD
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 10)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 10)
    assert done == 10
    assert rel_l2(q1, q1o) < TOL_F32_TRAJ and rel_l2(p1, p1o) < TOL_F32_TRAJ
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    dH, t = e.delta_hamiltonian(c.q0, c.p0, q1o, p1o)
    assert np.all(np.abs(t - to) <= TOL_F32_ENERGY * np.abs(to))
    e.close()


# ---- code paths that the default 16^3 cases do not reach ----------------------------------------------------
@pytest.mark.parametrize("nx,kw", [
    (10, dict(likelihood=1, rsd_model=1)),                       # no tile shape divides 10: direct (unsorted) kernels
    (12, dict(likelihood=1, rsd_model=0)),                       # 4 x 4 x 4 tiles
    (24, dict(likelihood=0, rsd_model=0)),                       # 8 x 8 x 8 tiles
    (16, dict(likelihood=1, rsd_model=1, particle_kernel_h_rel=1.3)),   # hull not exact: cube loop, halo = reach
    (16, dict(likelihood=1, rsd_model=0, particle_kernel_h_rel=0.8)),   # smaller kernel
    # the 81-cell hull holds for 0.83 d <= h < 1.06 d, the unrolled kernels' unroll-time choice of spline branch for
    # 0.866 d <= h only (home cell at q <= 1): just below that bound (generic tile kernel), and inside the range
    (16, dict(likelihood=1, rsd_model=1, particle_kernel_h_rel=0.85)),
    (16, dict(likelihood=1, rsd_model=1, particle_kernel_h_rel=0.9)),
    (16, dict(likelihood=1, rsd_model=0, particle_kernel_h_rel=1.04)),
    (16, dict(likelihood=1, rsd_model=0, min1=1.0, min2=2.0, min3=0.5)),  # xllc != 0: particles below min are dropped
], ids=["n10_direct", "n12_tile4", "n24_tile8", "h1.3", "h0.8", "h0.85", "h0.9", "h1.04", "xllc"])
def test_other_tilings_and_kernel_sizes(nx, kw):
    import warnings
    c = Case(Nx=nx, **kw)
    e = c.engine()
    g, gp, gl = c.oracle.gradient_psi(c.q0)
    gg = e.gradient(c.q0)
    assert rel_l2(e.fetch("rho"), c.oracle.getDensity(3, *[c.oracle.get(k) for k in ("posx", "posy", "posz")])) < TOL_FIELD
    assert rel_l2(gg, g) < 10 * TOL_FIELD
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 5)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 5)
    assert done == 5
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    e.close()


@pytest.mark.parametrize("cap", ["0", "8", "1100"], ids=["two_pass_only", "overflow_fallback", "tight_cap"])
def test_tile_sort_fallback_paths(monkeypatch, cap):
    """The one-pass tile binning reserves BCHMC_SORT_CAP record slots per tile (default 8x the mean occupancy, in eight
    octant segments); when a segment overflows, or with the one-pass path disabled (0), the two-pass counting sort
    produces the records and k_subsort orders them."""
    monkeypatch.setenv("BCHMC_SORT_CAP", cap)
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    e = c.engine()
    g, _, _ = c.oracle.gradient_psi(c.q0)
    assert rel_l2(e.gradient(c.q0), g) < 10 * TOL_FIELD
    assert rel_l2(e.fetch("rho"), c.oracle.getDensity(3, *[c.oracle.get(k) for k in ("posx", "posy", "posz")])) < TOL_FIELD
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 5)
    for _ in range(9):  # after an overflow the slots double before the next trajectory: 8 -> ... -> 2048 >= occupancy
        q1, p1, _ = e.leapfrog(c.q0, c.p0, c.eps, 5)
        assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    e.close()


def test_direct_kernels_when_tiling_is_disabled(monkeypatch):
    """BCHMC_NO_TILES=1 forces the unsorted scatter/gather kernels (IEEE sqrt/divide, global atomics)."""
    monkeypatch.setenv("BCHMC_NO_TILES", "1")
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    e = c.engine()
    g, _, _ = c.oracle.gradient_psi(c.q0)
    assert rel_l2(e.gradient(c.q0), g) < 10 * TOL_FIELD
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 5)
    q1, p1, _ = e.leapfrog(c.q0, c.p0, c.eps, 5)
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    e.close()


@pytest.mark.parametrize("kw", [
    dict(likelihood=1, rsd_model=1),
    dict(likelihood=1, rsd_model=0, calc_h=3),
    dict(likelihood=1, rsd_model=0, calc_h=0),
    dict(likelihood=0, rsd_model=1, mass_type=5),
    dict(likelihood=3, rsd_model=0),
    dict(likelihood=1, rsd_model=1, precision=1),
], ids=["gauss_rsd", "calc_h3", "calc_h0", "mass5", "grf", "fp32"])
def test_padded_half_complex_rows(monkeypatch, kw):
    """BCHMC_FFT_PAD=1 forces the padded row stride of the half-complex arrays (default only for n >= 128) at 16^3:
    every k-space kernel must skip the padding and the rocFFT plans must use the strided layout."""
    monkeypatch.setenv("BCHMC_FFT_PAD", "1")
    kw = dict(kw)
    f32 = kw.pop("precision", 0) == 1
    c = Case(Nx=16, **kw)
    e = c.engine(precision=1) if f32 else c.engine()
    tf, tt, te = (TOL_F32_FIELD * 10, TOL_F32_TRAJ, TOL_F32_ENERGY) if f32 else (10 * TOL_FIELD, TOL_TRAJ_10, TOL_ENERGY)
    g, _, _ = c.oracle.gradient_psi(c.q0)
    assert rel_l2(e.gradient(c.q0), g) < tf
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 5)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 5)
    assert done == 5
    assert rel_l2(q1, q1o) < tt and rel_l2(p1, p1o) < tt
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    dH, t = e.delta_hamiltonian(c.q0, c.p0, q1o, p1o)
    assert np.all(np.abs(t - to) <= te * np.abs(to))
    if not f32:
        # resident chain: tapped energies of the same trajectory
        e.chain_set_state(c.q0)
        e.chain_set_momenta(c.p0)
        dHc, tc, donec = e.chain_attempt(c.eps, 5)
        assert donec == 5 and np.all(np.abs(tc - to) <= 1e-8 * np.abs(to).max())
    e.close()


@pytest.mark.parametrize("kw", [dict(likelihood=0, rsd_model=0), dict(likelihood=2, rsd_model=0),
                                dict(likelihood=1, rsd_model=0, deltaQ_factor=0.9, grad_psi_prior_factor=0.5,
                                     grad_psi_likeli_factor=2.0, correct_delta=0)],
                         ids=["poisson", "lognormal", "factors"])
def test_planes_mode_other_likelihoods(monkeypatch, kw):
    """Planes mode only changes how V^ reaches the boundary arithmetic: other likelihoods and the test factors."""
    monkeypatch.setenv("BCHMC_FFT_PAD", "1")
    c = Case(Nx=32, **kw)
    e = c.engine()
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 5)
    q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 5)
    assert done == 5 and rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    e.chain_set_state(c.q0)
    e.chain_set_momenta(c.p0)
    dH, terms, _ = e.chain_attempt(c.eps, 5)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    assert np.all(np.abs(terms - to) <= 1e-8 * np.abs(to).max())
    e.close()


@pytest.mark.parametrize("nx,precision", [(32, 0), (32, 1), (64, 0)], ids=["n32_fp64", "n32_fp32", "n64_fp64"])
def test_planes_mode_step_boundary(monkeypatch, nx, precision):
    """Interior step boundaries in "planes" mode (2-D rocFFT transforms of the (y, z) planes, the x passes fused into
    k_step_boundary_x; its BX_FIRST / BX_LAST variants at the ends of the trajectory and for the force evaluation before
    the first step) against the oracle, against planes mode for interior steps only (BCHMC_NO_PLANES_ENDS=1) and
    against the 3-D-transform path (BCHMC_NO_PLANES=1); a one-step trajectory is first and last step at once."""
    monkeypatch.setenv("BCHMC_FFT_PAD", "1")  # whole 128-byte k-groups per row (default only for n >= 128)
    c = Case(Nx=nx, likelihood=1, rsd_model=1)
    neps = 4
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, neps)
    q1o_1, p1o_1, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 1)
    out = []
    tol = TOL_TRAJ_10 if precision == 0 else TOL_F32_TRAJ
    for no_planes, no_ends in (("0", "0"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("BCHMC_NO_PLANES", no_planes)
        monkeypatch.setenv("BCHMC_NO_PLANES_ENDS", no_ends)
        e = c.engine(precision=precision)
        qs, ps_, done = e.leapfrog(c.q0, c.p0, c.eps, 1)
        assert done == 1 and rel_l2(qs, q1o_1) < tol and rel_l2(ps_, p1o_1) < tol
        q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, neps)
        assert done == neps
        qb = pb = np.zeros(1)
        if precision == 0:  # 1e50 is not representable in fp32 storage: the guard is an fp64 feature
            p_bad = c.p0.copy().ravel()
            p_bad[0] = 1e60
            qb, pb, doneb = e.leapfrog(c.q0, p_bad, 1e-6, neps + 2)  # guard trips inside a planes-mode boundary
            assert doneb == 1
        out.append((q1, p1, qb, pb))
        e.close()
    for q1, p1, _, _ in out:
        assert rel_l2(q1, q1o) < tol and rel_l2(p1, p1o) < tol
    noise = 1e-13 if precision == 0 else 1e-5
    for other in out[1:]:
        assert rel_l2(out[0][0], other[0]) < noise and rel_l2(out[0][1], other[1]) < noise
        if precision == 0:
            assert rel_l2(out[0][2], other[2]) < noise and rel_l2(out[0][3], other[3]) < noise


def test_device_resident_entry_points():
    """bchmc_leapfrog_device / bchmc_energies_device on torch tensors give the host-array results."""
    import torch
    c = Case(Nx=16, likelihood=1, rsd_model=1)
    e = c.engine()
    dev = torch.device("cuda", 0)
    q0 = torch.from_numpy(c.q0.reshape(-1).copy()).to(dev)
    p0 = torch.from_numpy(c.p0.reshape(-1).copy()).to(dev)
    q1, p1 = torch.empty_like(q0), torch.empty_like(p0)
    e.leapfrog_device(q0, p0, q1, p1, c.eps, 6)
    assert e.steps_done() == 6
    qh, ph, _ = e.leapfrog(c.q0, c.p0, c.eps, 6)
    assert rel_l2(q1.cpu().numpy(), qh) < 1e-12 and rel_l2(p1.cpu().numpy(), ph) < 1e-12
    assert torch.equal(q0.cpu(), torch.from_numpy(c.q0.reshape(-1)))  # inputs untouched
    en_d = e.energies_device(q1, p1)
    en_h = e.energies(qh, ph)
    assert np.allclose(en_d, en_h, rtol=1e-11)
    e.close()


def test_host_side_mirror_of_the_reference_interface():
    """barcode_amd.hamil: Hamiltonian_EoM draws (Neps, epsilon) exactly as HMC.cc:260-264 and keeps the log scalars."""
    from barcode_amd import hamil
    c = Case(Nx=16, likelihood=1)
    hd = hamil.HamilData(c.p, N_eps_fac=8.0, eps_fac=c.eps * 2, **c.arrays())
    draws = iter([0.55, 0.5])  # -> Neps = int(8 * 0.55) + 1 = 5, epsilon = eps_fac * 0.5
    qf, pf = hamil.Hamiltonian_EoM(hd, c.q0, c.p0, lambda: next(draws))
    assert hd.numerical.Neps == 5 and np.isclose(hd.numerical.epsilon, c.eps) and hd.numerical.count_attempts == 1
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 5)
    assert rel_l2(qf, q1o) < TOL_TRAJ_10 and rel_l2(pf, p1o) < TOL_TRAJ_10
    dH = hamil.delta_Hamiltonian(hd, c.q0, c.p0, qf, pf)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    n = hd.numerical
    assert np.allclose([n.H_kin_i, n.psi_prior_i, n.psi_likeli_i, n.H_kin_f, n.psi_prior_f, n.psi_likeli_f], to, rtol=1e-9)
    assert abs(dH - dHo) <= 1e-8 * abs(to).max()
    assert rel_l2(hamil.gradient_psi(hd, c.q0), c.oracle.gradient_psi(c.q0)[0]) < 10 * TOL_FIELD
    hd.engine.close()
