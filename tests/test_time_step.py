"""Host-side step-size adaptation (barcode_amd/time_step.py) against hand-worked cases of the reference's rules
(time_step.cpp:40-185, time_step.hpp:23-75)."""
import numpy as np
import pytest

from barcode_amd import time_step as ts
from barcode_amd.chains import EpsRing


def test_helpers():
    assert np.allclose(ts.cumulative_moving_average([1, 0, 1, 1]), [1, 0.5, 2 / 3, 0.75])
    assert np.allclose(ts.stl_smooth([0, 3, 6, 9], 1), [1.5, 3, 6, 7.5])
    assert ts.sort_vector_by_other([True, False, True], [0.3, 0.1, 0.2]).tolist() == [False, True, True]
    assert np.isclose(ts.power_mean(1.0, 4.0, 0), 2.0) and np.isclose(ts.power_mean(1.0, 3.0, 2), np.sqrt(5.0))


def _ring(pairs, n=None):
    r = EpsRing(n or len(pairs))
    for acc, eps in pairs:
        r.record(acc, eps)
    return r


def test_no_update_off_schedule_and_in_band():
    cfg = ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=4)
    r = _ring([(1, .1), (1, .2), (0, .3)], n=4)           # 3 attempts: not a multiple of 4
    assert ts.update_eps_fac(1.0, r, cfg) == 1.0
    cfg2 = ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=20)
    r2 = _ring([(i % 20 < 13, 0.01 * i) for i in range(20)])   # acceptance 0.65: inside [0.6, 0.7]
    assert ts.update_eps_fac(1.0, r2, cfg2) == 1.0


def test_upwards():
    cfg = ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=4, eps_up_fac=1.0)
    r = _ring([(1, .1), (1, .2), (1, .3), (1, .4)])
    assert np.isclose(ts.update_eps_fac(2.0, r, cfg), 2.0 * (1.0 / 0.65))   # time_step.cpp:124-127


def test_downwards_threshold_crossing():
    # small steps accepted, large ones rejected: eps_fac drops to the epsilon where the smoothed running
    # acceptance (sorted by epsilon) first falls below the target 0.65
    cfg = ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=10, eps_down_smooth=0)
    pairs = [(1, .01), (1, .02), (1, .03), (0, .04), (0, .05), (0, .06), (0, .07), (0, .08), (0, .09), (0, .10)]
    r = _ring(pairs)
    # running acceptance by epsilon: 1, 1, 1, .75, .6, ... -> first below .65 at index 4 -> eps = .05
    assert np.isclose(ts.update_eps_fac(0.5, r, cfg), 0.05)


def test_downwards_fallbacks():
    cfg = ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=4, eps_down_smooth=0)
    none = _ring([(0, .4), (0, .2), (0, .3), (0, .5)])
    assert ts.update_eps_fac(1.0, none, cfg) == 0.2                  # no accepted step: lowest epsilon tried
    # acceptance 0.5 but the smoothed curve never exceeds the target: divide by three (time_step.cpp:95-99)
    low = _ring([(0, .1), (1, .2), (0, .3), (1, .4)])
    assert np.isclose(ts.update_eps_fac(0.9, low, cfg), 0.3)
    zero = _ring([(0, 0.0), (0, 0.0), (0, 0.0), (0, 0.0)])
    with pytest.raises(RuntimeError):
        ts.update_eps_fac(1.0, zero, cfg)


def test_scheme_1_and_3():
    cfg = ts.EpsConfig(eps_fac_update_type=1, s_eps_total=2, eps_fac_target=0.1, eps_fac_power=2)
    r = _ring([(1, .1), (1, .2)], n=100)
    assert np.isclose(ts.update_eps_fac(0.5, r, cfg), ts.power_mean(0.5, 0.1, 2))
    cfg3 = ts.EpsConfig(eps_fac_update_type=3, N_a_eps_update=100)
    assert ts.update_eps_fac(0.8, r, cfg3, iGibbs=1, rejections=2) == 0.4     # halving until the first acceptance
    assert ts.update_eps_fac(0.8, r, cfg3, iGibbs=5, rejections=2) == 0.8
    assert ts.update_eps_fac(0.8, r, ts.EpsConfig(eps_fac_update_type=0)) == 0.8
