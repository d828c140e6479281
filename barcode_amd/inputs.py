"""Synthetic inputs for the HMC leapfrog path.

Host-side numpy only; nothing here is on the measured path.  Conventions follow the reference:
  * P(k) table -> 3-D grid: ``readtab`` (``barlib/src/calc_power.cc:31-107``): table read as float32,
    linear interpolation in |k|, P(k=0) = 0, full N^3 real grid indexed ``k + N3*(j + N2*i)``.
  * Gaussian random field with spectrum S: <|FFT[f]|^2> = N^2 S / V (FOURIER_DEF_2; amplitude
    convention of ``barlib/src/random.cpp:82,106``).
  * mass_type 1: mass_f = 1/P, 0 where P <= 0 (``barlib/src/HMC_mass.cc:117-124,163-172``).
  * mock data: ``setup_random_test`` (``barlib/src/barcoderunner.cc:117-183``).
All random draws use numpy's counter-based Philox generator with documented seeds.
"""
import os

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
PLANCK_TABLE = os.path.join(DATA_DIR, "PLANCK_CAMB.dat")

SEED_TRUTH, SEED_NOBS, SEED_Q0, SEED_P0 = 1001, 1002, 1003, 1004


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (a GPU box shows all 256
    hardware threads of its host and grants about 16 of them; thread pools sized by the visible count spin against the
    quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _workers():
    """Threads for the big-grid FFTs."""
    return min(host_cpu_share(), 32)


def _rng(seed):
    return np.random.Generator(np.random.Philox(seed))


def read_power_table(path=PLANCK_TABLE):
    """Two-column text table k [h/Mpc], P(k); stored through float32 like readtab (calc_power.cc:42-68)."""
    tab = np.loadtxt(path, dtype=np.float32)
    return tab[:, 0].astype(np.float64), tab[:, 1].astype(np.float64)


def k_axis(n, L):
    """calc_ki (scale_space.cpp:41-51): 2*pi/L * i for i <= n/2, else -2*pi/L * (n - i)."""
    i = np.arange(n)
    kfac = 2.0 * np.pi / L
    return np.where(i <= n // 2, kfac * i, -kfac * (n - i))


def power_grid(params, table=None):
    """signal_PS on the full N^3 grid (calc_power.cc:91-104)."""
    ktab, ptab = table if table is not None else read_power_table()
    k1 = k_axis(params.Nx, params.L)
    ksq = k1[:, None, None] ** 2 + k1[None, :, None] ** 2 + k1[None, None, :] ** 2
    P = np.interp(np.sqrt(ksq), ktab, ptab)
    P[0, 0, 0] = 0.0
    return np.ascontiguousarray(P)


def inverse_power_mass(P):
    """mass_f for mass_type 1."""
    out = np.zeros_like(P)
    np.divide(1.0, P, out=out, where=P > 0)
    return out


def gaussian_random_field(params, spectrum, seed, scale=1.0):
    """Real field f with <|FFT f|^2> = N^2 * spectrum / V, Hermitian by construction."""
    n, N, V = params.Nx, params.N, params.L ** 3
    white = _rng(seed).standard_normal((n, n, n))
    amp = np.sqrt(np.maximum(spectrum[:, :, : n // 2 + 1], 0.0) * (N / V))
    if n >= 256:
        # BASELINE-size grids: threaded pocketfft (same algorithm; last-bit differences from numpy's serial one do
        # not matter, every consumer of a case reads the same arrays).  Small grids keep numpy: the golden fixtures
        # were generated with it.
        from scipy import fft as sfft
        nw = _workers()
        wk = sfft.rfftn(white, workers=nw)
        del white
        wk *= amp
        f = sfft.irfftn(wk, s=(n, n, n), axes=(0, 1, 2), workers=nw, overwrite_x=True)
    else:
        wk = np.fft.rfftn(white)
        f = np.fft.irfftn(wk * amp, s=(n, n, n), axes=(0, 1, 2))
    if scale != 1.0:
        f *= scale
    return np.ascontiguousarray(f)


def mock_observations(params, delta_eul, seed=SEED_NOBS, delta_lag=None, sigma_fac_lognormal=0.1):
    """window, noise, nobs for the configured likelihood (barcoderunner.cc:91-183, window_type 1)."""
    rng = _rng(seed)
    shape = delta_eul.shape
    window = np.ones(shape)
    noise = np.full(shape, params.sigma_min)
    lam = params.rho_c * (1.0 + delta_eul)
    if params.likelihood == 0:
        nobs = rng.poisson(np.maximum(lam, 0.0)).astype(np.float64)
    elif params.likelihood == 1:
        nobs = np.maximum(lam + params.sigma_min * rng.standard_normal(shape), 0.0)
    elif params.likelihood == 2:
        noise = np.full(shape, sigma_fac_lognormal)
        lam_ln = np.log(params.rho_c * (1.0 + np.maximum(delta_eul, params.delta_min)))
        nobs = lam_ln + sigma_fac_lognormal * rng.standard_normal(shape)
    elif params.likelihood == 3:
        nobs = delta_lag + params.sigma_min * rng.standard_normal(shape)
    else:
        raise ValueError("likelihood must be 0..3")
    return window, noise, np.ascontiguousarray(nobs)


def make_fields(params, table=None, q0_scale=0.5):
    """signal_PS, mass_f, truth delta_q, starting point q0 and momenta p0 (SURVEY 8d seeds 1001/1003/1004)."""
    P = power_grid(params, table)
    mass_f = inverse_power_mass(P)
    truth = gaussian_random_field(params, P, SEED_TRUTH)
    q0 = gaussian_random_field(params, P, SEED_Q0, scale=q0_scale)
    p0 = gaussian_random_field(params, mass_f, SEED_P0)
    return dict(signal_PS=P, mass_f=mass_f, truth=truth, q0=q0, p0=p0)
