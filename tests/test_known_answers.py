"""Hand-derived known answers for three pieces of the path, checked by BOTH the oracle (CPU tests) and the engine
(``-m gpu`` tests).  They do not lift "parity unpinned" (the reference holds no vectors for this path and cannot be
built here) but they anchor the oracle -- and the engine -- to closed forms derived from the reference's formulas
instead of to each other.

K1  SPH mass assignment on the undisplaced lattice (``getDensity_SPH``, massFunctions.cc:392-495 with the Monaghan W_4
    kernel of ``SPH_kernel_3D``, :366-384).  h = d (particle_kernel_h_rel = 1), every particle on its cell centre.
    A cell at integer offset (i, j, k) from a particle is at r = d sqrt(i^2 + j^2 + k^2), q = r / h = sqrt(i^2+j^2+k^2),
    and receives W = w (1 - 3/2 q^2 + 3/4 q^3) for q <= 1, w/4 (2 - q)^3 for 1 < q <= 2, with w = 1 / (pi h^3).
    Offsets with q <= 2:  1 at q = 0 (W = w),  6 at q = 1 (w/4),  12 at q = sqrt2 (w/4 (2 - sqrt2)^3),
    8 at q = sqrt3 (w/4 (2 - sqrt3)^3),  6 at q = 2 (exactly 0): 33 cells, 27 of them non-zero.
    One particle: those 33 values.  Full lattice: every cell receives the same sum
        rho = w [1 + 6/4 + 3 (2 - sqrt2)^3 + 2 (2 - sqrt3)^3],      delta_x = 0.

K2  Zel'dovich displacement of one plane wave (``theta2vel``, EqSolvers.cc:168-277, fed with -D1 delta as
    ``Lag2Eul_zeldovich`` does, Lag2Eul.cc:88-93).  delta(x) = A cos(k0 x), k0 = 2 pi m / L, sampled at x_i = i d.
    theta2vel multiplies the transform of phi = -D1 delta by -i k_j / k^2, i.e. Psi = grad (inverse-Laplacian phi):
    inverse-Laplacian phi = -phi / k0^2 = D1 A cos(k0 x) / k0^2, so
        Psi_x(x_i) = -(D1 A / k0) sin(k0 x_i),   Psi_y = Psi_z = 0,
    and the particle of cell i sits at (i + 1/2) d + Psi_x(x_i) (``disp_part``, disp_part.cc:55-107), wrapped to [0, L).

K3  SPH-kernel gradient gather for a single excited cell (``likelihood_calc_V_SPH``, HMC_models.cc:200-303, with
    ``grad_SPH_kernel_3D_h_units``, SPH_kernel.cpp:148-208).  On the undisplaced lattice with part_like = 1 in one cell
    c0 and 0 elsewhere, the particle at offset (i, j, k) cells from c0 gets, in units of h = d,
        V = m * partial(q) * (i, j, k),   m = rho_c L^3 / N = d^3,   norm = 1 / (pi h^4),
        partial(q) = (2.25 q - 3) norm            for q^2 <= 1,
                   = -0.75 (q - 2)^2 norm / q     for 1 < q^2 <= 4,   0 beyond,
    i.e. V = f(q) (i, j, k) / (pi d) with f(1) = -3/4, f(sqrt2) = -3/4 (sqrt2 - 2)^2 / sqrt2,
    f(sqrt3) = -3/4 (sqrt3 - 2)^2 / sqrt3, f(2) = 0 (and 0 at the excited cell itself: the offset vanishes).
    The engine is driven to this state through its data arrays: Gaussian likelihood, window = sigma = 1, rho_c = 1,
    undisplaced lattice (q = 0 => delta_x = 0 => Lambda = 1), so part_like = (nobs - Lambda) / sigma^2 = nobs - 1
    (gaussian_independent.cpp:24-42): nobs = 1 everywhere, 2 in c0.
"""
import itertools

import numpy as np
import pytest

from barcode_amd.params import HamilParams
from oracle.oracle import Oracle

N, L = 16, 40.0
D = L / N
W0 = 1.0 / (np.pi * D ** 3)
SQ2, SQ3 = np.sqrt(2.0), np.sqrt(3.0)


def w4(q):
    return W0 * (1 - 1.5 * q * q + 0.75 * q ** 3) if q <= 1 else (W0 * 0.25 * (2 - q) ** 3 if q <= 2 else 0.0)


def k1_single_particle(c0):
    """Expected rho for one particle on the centre of cell c0."""
    rho = np.zeros((N, N, N))
    for i, j, k in itertools.product(range(-2, 3), repeat=3):
        q = np.sqrt(i * i + j * j + k * k)
        if q <= 2:
            rho[(c0[0] + i) % N, (c0[1] + j) % N, (c0[2] + k) % N] += w4(q)
    return rho


K1_LATTICE = W0 * (1 + 6 / 4 + 3 * (2 - SQ2) ** 3 + 2 * (2 - SQ3) ** 3)


def f_grad(q):
    if q * q <= 1:
        return 2.25 * q - 3
    return -0.75 * (q - 2) ** 2 / q if q * q <= 4 else 0.0


def k3_expected(c0):
    V = np.zeros((3, N, N, N))
    for i, j, k in itertools.product(range(-2, 3), repeat=3):
        q = np.sqrt(i * i + j * j + k * k)
        if 0 < q <= 2:
            V[:, (c0[0] + i) % N, (c0[1] + j) % N, (c0[2] + k) % N] = f_grad(q) * np.array([i, j, k]) / (np.pi * D)
    return V


def lattice():
    c = (np.arange(N) + 0.5) * D
    px, py, pz = np.meshgrid(c, c, c, indexing="ij")
    return px.ravel().copy(), py.ravel().copy(), pz.ravel().copy()


def params(**kw):
    return HamilParams(Nx=N, L=L, likelihood=1, rsd_model=0, sfmodel=1, **kw)


# ---- K1 -------------------------------------------------------------------------------------------------------
def test_k1_oracle_single_particle_and_lattice():
    o = Oracle(params())
    c0 = (3, 0, 15)                                    # next to two periodic faces
    px, py, pz = (np.full(N ** 3, -1.0) for _ in range(3))   # outside [0, L): dropped (massFunctions.cc:426)
    p0 = c0[2] + N * (c0[1] + N * c0[0])
    px[p0], py[p0], pz[p0] = [(c + 0.5) * D for c in c0]
    rho = o.getDensity(3, px, py, pz).reshape(N, N, N)
    exp = k1_single_particle(c0)
    assert np.count_nonzero(exp) == 27 and np.count_nonzero(rho) == 27
    assert np.allclose(rho, exp, rtol=1e-14, atol=1e-16 * W0)
    assert rho[c0] == pytest.approx(W0, rel=1e-15) and rho[(c0[0] + 1) % N, c0[1], c0[2]] == pytest.approx(W0 / 4, rel=1e-15)
    rho_l = o.getDensity(3, *lattice())
    assert np.allclose(rho_l, K1_LATTICE, rtol=1e-14)
    o.close()


@pytest.mark.gpu
def test_k1_engine_lattice():
    from barcode_amd.engine import Engine
    e = Engine(params())
    one = np.ones(N ** 3)
    e.upload(signal_PS=one, mass_f=one, nobs=one, noise=one, window=one)
    e.forward(np.zeros(N ** 3))
    assert np.allclose(e.fetch("rho"), K1_LATTICE, rtol=1e-14)
    assert np.max(np.abs(e.fetch("deltaX"))) < 1e-14
    c = (np.arange(N) + 0.5) * D
    assert np.array_equal(e.fetch("posx").reshape(N, N, N)[:, 0, 0], c)   # disp_part with Psi = 0: cell centres
    e.close()


# ---- K2 -------------------------------------------------------------------------------------------------------
def plane_wave(m, A):
    k0 = 2 * np.pi * m / L
    x = np.arange(N) * D
    delta = np.broadcast_to((A * np.cos(k0 * x))[:, None, None], (N, N, N)).copy()
    psi_x = np.broadcast_to((-(A / k0) * np.sin(k0 * x))[:, None, None], (N, N, N)).copy()   # D1 = 1
    return delta, psi_x


@pytest.mark.parametrize("m", [1, 3, 7])
def test_k2_oracle_plane_wave_displacement(m):
    p = params()
    o = Oracle(p)
    delta, psi_x = plane_wave(m, 0.3)
    vx, vy, vz = o.theta2vel(-p.D1 * delta)
    scale = np.abs(psi_x).max()
    assert np.max(np.abs(vx.reshape(N, N, N) - psi_x)) < 1e-13 * scale
    assert np.max(np.abs(vy)) < 1e-13 * scale and np.max(np.abs(vz)) < 1e-13 * scale
    _, posx, posy, posz = o.Lag2Eul(delta, rsd=0)
    exp = ((np.arange(N) + 0.5) * D)[:, None, None] + psi_x
    exp = np.where(exp < 0, exp + L, np.where(exp >= L, exp - L, exp))
    assert np.max(np.abs(posx.reshape(N, N, N) - exp)) < 1e-12
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1, 3, 7])
def test_k2_engine_plane_wave_displacement(m):
    from barcode_amd.engine import Engine
    e = Engine(params())
    one = np.ones(N ** 3)
    e.upload(signal_PS=one, mass_f=one, nobs=one, noise=one, window=one)
    delta, psi_x = plane_wave(m, 0.3)
    e.forward(delta)
    scale = np.abs(psi_x).max()
    assert np.max(np.abs(e.fetch("psix").reshape(N, N, N) - psi_x)) < 1e-13 * scale
    assert np.max(np.abs(e.fetch("psiy"))) < 1e-13 * scale and np.max(np.abs(e.fetch("psiz"))) < 1e-13 * scale
    exp = ((np.arange(N) + 0.5) * D)[:, None, None] + psi_x
    exp = np.where(exp < 0, exp + L, np.where(exp >= L, exp - L, exp))
    assert np.max(np.abs(e.fetch("posx").reshape(N, N, N) - exp)) < 1e-12
    e.close()


# ---- K3 -------------------------------------------------------------------------------------------------------
def test_k3_oracle_gradient_gather_of_one_excited_cell():
    o = Oracle(params())
    c0 = (0, 7, 15)
    plike = np.zeros((N, N, N))
    plike[c0] = 1.0
    vx, vy, vz = o.likelihood_calc_V_SPH(plike, *lattice())
    exp = k3_expected(c0)
    scale = 0.75 / (np.pi * D)
    for got, want in zip((vx, vy, vz), exp):
        assert np.max(np.abs(got.reshape(N, N, N) - want)) < 1e-14 * scale
    assert np.count_nonzero(np.abs(exp).sum(axis=0)) == 26     # 6 + 12 + 8 neighbours; q = 2 gives exactly 0
    assert vx.reshape(N, N, N)[1, 7, 15] == pytest.approx(-0.75 / (np.pi * D), rel=1e-15)   # q = 1, offset (1, 0, 0)
    o.close()


@pytest.mark.gpu
def test_k3_engine_gradient_gather_of_one_excited_cell():
    from barcode_amd.engine import Engine
    e = Engine(params())
    one = np.ones(N ** 3)
    c0 = (0, 7, 15)
    nobs = np.ones((N, N, N))
    nobs[c0] = 2.0
    e.upload(signal_PS=one, mass_f=one, nobs=nobs, noise=one, window=one)
    e.gradient(np.zeros(N ** 3))
    plike = e.fetch("part_like").reshape(N, N, N)
    exp_pl = np.zeros((N, N, N))
    exp_pl[c0] = 1.0
    assert np.max(np.abs(plike - exp_pl)) < 1e-14
    exp = k3_expected(c0)
    scale = 0.75 / (np.pi * D)
    for name, want in zip(("Vx", "Vy", "Vz"), exp):
        # part_like is 1 + O(1e-16) in c0 and O(1e-16) elsewhere (delta_x = 0 to rounding): same bound
        assert np.max(np.abs(e.fetch(name).reshape(N, N, N) - want)) < 1e-13 * scale
    e.close()
