"""oracle/orc_random.c: the reference's momentum draw (HMC_momenta.cc:42-94, random.cpp:48-511, random.hpp:35-120) with
GSL's generator restated.  GSL is absent from the reference tree and from this image, so the stream is pinned to the
published MT19937 known answers and cross-checked against numpy's legacy MT19937 (same seeding); the field is checked
through the properties create_GARFIELD is built for: Hermitian symmetry, the Fourier amplitude convention, and the
resolution independence of the random grid."""
import numpy as np
import pytest

from barcode_amd import inputs
from barcode_amd.params import HamilParams
from oracle import oracle as orc


def test_mt19937_known_answers_and_numpy_cross_check():
    # Matsumoto & Nishimura's reference output for init_genrand(5489) (the generator's documented default seed)
    assert orc.mt19937_stream(5489, 5).tolist() == [3499211612, 581869302, 3890346734, 3586334585, 545404204]
    for seed in (1, 42, 20260410):
        raw = orc.mt19937_stream(seed, 2000).astype(np.uint64)
        # numpy's legacy RandomState seeds with the same init_genrand and builds doubles from (a >> 5, b >> 6)
        dbl = ((raw[0::2] >> np.uint64(5)) * 67108864.0 + (raw[1::2] >> np.uint64(6))) / 9007199254740992.0
        assert np.array_equal(dbl, np.random.RandomState(seed).random_sample(1000))
    assert np.array_equal(orc.mt19937_stream(0, 3), orc.mt19937_stream(4357, 3))   # gsl: seed 0 means 4357


def test_ugaussian_is_the_polar_method_on_that_stream():
    seed, n = 7, 500
    u = orc.mt19937_stream(seed, 8 * n) / 4294967296.0
    out, i = [], 0
    while len(out) < n:
        x, y = -1 + 2 * u[i], -1 + 2 * u[i + 1]    # (a zero uniform would be redrawn: none in this stretch)
        i += 2
        r2 = x * x + y * y
        if r2 > 1.0 or r2 == 0:
            continue
        out.append(y * np.sqrt(-2.0 * np.log(r2) / r2))
    assert np.all(u[:i] != 0)
    assert np.allclose(orc.ugaussian_stream(seed, n), out, rtol=1e-15, atol=0)
    g = orc.ugaussian_stream(3, 200000)
    assert abs(g.mean()) < 0.01 and abs(g.var() - 1.0) < 0.01


def _fields(n, L=100.0):
    p = HamilParams(Nx=n, L=L)
    P = inputs.power_grid(p)
    return p, P


def test_garfield_is_real_with_the_reference_amplitude_convention():
    n = 32
    p, P = _fields(n)
    f = orc.create_GARFIELD(n, p.L, P, seed=11).reshape(n, n, n)
    fk = np.fft.fftn(f)
    # Hermitian by construction; the zero mode is removed (random.cpp:351-357)
    assert abs(fk[0, 0, 0]) < 1e-9 * np.abs(fk).max()
    # <|FFT f|^2> = N^2 / V * P (random.cpp:88-90, 106): shell-averaged ratio of one realisation
    ratio = np.abs(fk) ** 2 / np.where(P > 0, p.N ** 2 / p.L ** 3 * P, np.inf)
    sel = P > 0
    assert abs(ratio[sel].mean() - 1.0) < 0.03
    # the seven real corner modes: re = sqrt(2) sigma g, so their variance is N^2 / V * P too; here just real
    h = n // 2
    for c in [(h, 0, 0), (0, h, 0), (0, 0, h), (h, h, 0), (h, 0, h), (0, h, h), (h, h, h)]:
        assert abs(fk[c].imag) < 1e-9 * abs(fk[c])


def test_random_grid_is_resolution_independent():
    """Same seed, grid 2 n: the low-k modes carry the same random numbers (random.hpp:35-44, random.cpp:51-56), so
    FFT[f] / sigma agrees mode by mode for |k_i| < n / 2."""
    n, seed, L = 8, 5, 100.0
    Ps, Pb = np.ones(n ** 3), np.ones((2 * n) ** 3)
    fs = np.fft.fftn(orc.create_GARFIELD(n, L, Ps, seed).reshape(n, n, n)) / n ** 3          # sigma ~ N: divide
    fb = np.fft.fftn(orc.create_GARFIELD(2 * n, L, Pb, seed).reshape(2 * n, 2 * n, 2 * n)) / (2 * n) ** 3
    idx_s = [0, 1, 2, 3, -3, -2, -1]
    for a in idx_s:
        for b in idx_s:
            for c in idx_s:
                if (a, b, c) == (0, 0, 0):
                    continue
                # layer structure of the fill: index i of the small grid is index i (or 2n - (n - i)) of the big one
                assert fs[a, b, c] == pytest.approx(fb[a, b, c], rel=1e-12, abs=1e-14), (a, b, c)


def test_draw_momenta_kinetic_energy_and_real_space_part():
    n = 16
    p = HamilParams(Nx=n, L=50.0, mass_type=5)
    P = inputs.power_grid(p)
    mass_f = inputs.inverse_power_mass(P)
    mass_r = np.full(p.N, 0.25)
    pm = orc.draw_momenta(p, mass_f, mass_r, seed=3)
    pf = orc.draw_momenta(HamilParams(Nx=n, L=50.0, mass_type=1), mass_f, None, seed=3)
    white = pm - pf                      # the real-space part comes from the same stream, after the Fourier part
    assert abs(white.var() - 0.25) < 0.02 and abs(white.mean()) < 0.02
    # K = 1/2 p^T M^-1 p has mean N_modes / 2 for the Fourier part
    o = orc.Oracle(HamilParams(Nx=n, L=50.0, mass_type=1))
    o.set(mass_f=mass_f, signal_PS=P)
    K = o.kinetic_term(pf)
    nmodes = np.count_nonzero(mass_f > 0)
    assert abs(K - nmodes / 2.0) < 5 * np.sqrt(nmodes / 2.0)
    o.close()
