"""NGP / CIC / TSC mass assignment and calc_h = 3's TSC interpolation on the tile-sorted records (tiles_low.hpp)
against the oracle (massFunctions.cc:49-364, interpolate_grid.cpp:134-202) and against their direct one-thread-per-
particle forms (variants.hpp, BCHMC_NO_TILES_LOW=1), which must give the same sums up to the order of the atomics."""
import numpy as np
import pytest

from tests.util import TOL_FIELD, TOL_TRAJ_10, Case, rel_l2

pytestmark = pytest.mark.gpu

LOW = [dict(likelihood=1, rsd_model=0, calc_h=1, mk=1), dict(likelihood=0, rsd_model=0, calc_h=1, mk=0),
       dict(likelihood=1, rsd_model=1, calc_h=1, mk=2), dict(likelihood=0, rsd_model=0, calc_h=0, mk=1, eps_scale=0.01)]


@pytest.mark.parametrize("nx", [16, 32, 48])
@pytest.mark.parametrize("kw", LOW, ids=["cic", "ngp", "tsc_rsd", "cic_calch0"])
def test_low_order_mass_assignment_on_tiles(kw, nx):
    """48^3: tiles of 8 x 8 x 16 with three tiles along z; 16^3: one tile spans the whole z axis (the halo wraps onto
    the tile itself)."""
    c = Case(Nx=nx, **kw)
    e = c.engine()
    rsd = c.p.rsd_model
    dX, px, py, pz = c.oracle.Lag2Eul(c.truth, rsd=rsd)
    e.forward(c.truth, rsd)
    rho = c.oracle.getDensity(c.p.mk, px, py, pz)
    assert rel_l2(e.fetch("rho"), rho) < TOL_FIELD
    assert rel_l2(e.fetch("deltaX"), dX) < TOL_FIELD
    assert abs(e.fetch("rho").sum() - rho.sum()) <= 1e-12 * rho.sum()      # the flushed partial sums are the mean
    g, _, _ = c.oracle.gradient_psi(c.q0)
    assert rel_l2(e.gradient(c.q0), g) < 10 * TOL_FIELD
    if nx == 16:
        q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 10)
        q1, p1, done = e.leapfrog(c.q0, c.p0, c.eps, 10)
        assert done == 10 and rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    e.close()


@pytest.mark.parametrize("kw", LOW[:3] + [dict(likelihood=1, rsd_model=1, calc_h=3, sfmodel=2)],
                         ids=["cic", "ngp", "tsc_rsd", "calc_h3_rsd"])
def test_tiled_and_direct_forms_agree(kw, monkeypatch):
    c = Case(Nx=32, **kw)
    e = c.engine()
    g_tiled = e.gradient(c.q0)
    rho_tiled, V_tiled = e.fetch("rho"), [e.fetch(k) for k in ("Vx", "Vy", "Vz")] if c.p.calc_h == 3 else None
    e.close()
    monkeypatch.setenv("BCHMC_NO_TILES_LOW", "1")
    e = c.engine()
    g_direct = e.gradient(c.q0)
    assert rel_l2(rho_tiled, e.fetch("rho")) < 1e-14
    assert rel_l2(g_tiled, g_direct) < 1e-12
    if V_tiled is not None:  # same sums in the same order, from densities that differ in the last bits (atomic order)
        for a, k in zip(V_tiled, ("Vx", "Vy", "Vz")):
            assert rel_l2(a, e.fetch(k)) < 1e-12
    e.close()


def test_calc_h3_interpolation_on_tiles_against_oracle():
    for kw in (dict(likelihood=1, rsd_model=0, calc_h=3), dict(likelihood=1, rsd_model=1, calc_h=3, sfmodel=2)):
        c = Case(Nx=32, **kw)
        e = c.engine()
        g, _, _ = c.oracle.gradient_psi(c.q0)
        assert rel_l2(e.gradient(c.q0), g) < 10 * TOL_FIELD
        pl = c.oracle.partial_f_delta_x_log_like(c.oracle.get("deltaX"))
        V = c.oracle.likelihood_calc_V_SPH_fourier_TSC(pl, *[c.oracle.get(k) for k in ("posx", "posy", "posz")])
        for k, ref in zip(("Vx", "Vy", "Vz"), V):
            assert rel_l2(e.fetch(k), ref) < 10 * TOL_FIELD
        e.close()


def test_low_order_tiles_fp32_and_deterministic():
    from barcode_amd.engine import Engine
    c = Case(Nx=32, likelihood=1, rsd_model=1, calc_h=1, mk=1)
    dX = c.oracle.Lag2Eul(c.truth, rsd=1)[0]
    e = c.engine(precision=1)
    e.forward(c.truth, 1)
    assert rel_l2(e.fetch("deltaX"), dX) < 2e-5
    e.close()
    runs = []
    for _ in range(2):
        e = Engine(c.p, deterministic=1)
        e.upload(**c.arrays())
        e.forward(c.truth, 1)
        runs.append(e.fetch("rho"))
        assert rel_l2(e.fetch("deltaX"), dX) < TOL_FIELD
        e.close()
    assert np.array_equal(runs[0], runs[1])


def test_nonzero_grid_origin_keeps_the_direct_kernels():
    """The binning keys on floor(x / d), the low-order kernels' cells on floor((x - min) / d): with xllc != 0 the engine
    must not take the tile path (and must still match the oracle, dropped particles included)."""
    c = Case(Nx=16, likelihood=1, rsd_model=0, calc_h=1, mk=1, min1=3.0, min2=0.0, min3=1.5)
    e = c.engine()
    dX, px, py, pz = c.oracle.Lag2Eul(c.truth, rsd=0)
    e.forward(c.truth, 0)
    assert rel_l2(e.fetch("rho"), c.oracle.getDensity(1, px, py, pz)) < TOL_FIELD
    e.close()
