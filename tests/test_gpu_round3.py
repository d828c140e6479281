"""Round-3 paths against the oracle and against their predecessors: the staged density flush (k_scatter_tile81<STAGE> +
k_stage_combine81), record slots sized from the measured populations (adapt_slots / poll_slots), and the ALPT model on
the 2-D plans (k_step_boundary_x<ALPT>, k_alpt_mix_x with the cell-boundary average as a k-space phase)."""
import numpy as np
import pytest

from tests.util import TOL_ENERGY, TOL_FIELD, TOL_TRAJ_10, Case, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nx,chunk", [(16, None), (32, None), (32, "64"), (48, "128")],
                         ids=["n16", "n32", "n32_many_items_per_tile", "n48_three_tiles_in_z"])
def test_staged_flush_against_oracle_and_atomic_flush(monkeypatch, nx, chunk):
    """The scatter writes its LDS images to the staging area, the combine pass sums the <= 8 images per cell (all work
    items of each tile: BCHMC_CHUNK=64 makes ~16 per tile) -- same density as the oracle's and as the atomic flush."""
    if chunk:
        monkeypatch.setenv("BCHMC_CHUNK", chunk)
    monkeypatch.setenv("BCHMC_STAGE", "1")                 # opt-in: measured a wash at 256^3
    c = Case(Nx=nx, likelihood=1, rsd_model=1)
    dX, px, py, pz = c.oracle.Lag2Eul(c.truth, rsd=1)
    rho_o = c.oracle.getDensity(3, px, py, pz)
    e = c.engine()
    info = e.tile_info()
    assert info["stage"] == 1 and info["unrolled81"] == 1
    e.forward(c.truth, 1)                                  # forward_rest combines on its own (no likelihood pass)
    rho = e.fetch("rho")
    assert rel_l2(rho, rho_o) < TOL_FIELD and rel_l2(e.fetch("deltaX"), dX) < TOL_FIELD
    assert abs(rho.sum() - rho_o.sum()) <= 1e-12 * rho_o.sum()
    g, _, gl = c.oracle.gradient_psi(c.q0)                 # like_force: combine fused with the likelihood partial
    assert rel_l2(e.gradient(c.q0), g) < 10 * TOL_FIELD
    pl = c.oracle.partial_f_delta_x_log_like(c.oracle.get("deltaX"))
    assert rel_l2(e.fetch("part_like"), pl) < 10 * TOL_FIELD and rel_l2(e.fetch("deltaX"), c.oracle.get("deltaX")) < TOL_FIELD
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 6)
    q1, p1, done, dH, t = e.leapfrog_dh(c.q0, c.p0, c.eps, 6)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    assert done == 6 and rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    assert np.all(np.abs(t - to) <= 10 * TOL_ENERGY * np.abs(to))
    e.close()
    monkeypatch.setenv("BCHMC_STAGE", "0")
    e2 = c.engine()
    assert e2.tile_info()["stage"] == 0
    e2.forward(c.truth, 1)
    assert rel_l2(e2.fetch("rho"), rho) < 1e-14
    q2, p2, _ = e2.leapfrog(c.q0, c.p0, c.eps, 6)
    assert rel_l2(q2, q1) < 1e-13 and rel_l2(p2, p1) < 1e-13
    e2.close()


def test_staged_flush_other_likelihoods_and_fp32(monkeypatch):
    monkeypatch.setenv("BCHMC_STAGE", "1")
    for kw, prec, tol in ((dict(likelihood=0, rsd_model=0), 0, 10 * TOL_FIELD), (dict(likelihood=2, rsd_model=0), 0, 10 * TOL_FIELD),
                          (dict(likelihood=1, rsd_model=1), 1, 2e-5)):
        c = Case(Nx=32, **kw)
        e = c.engine(precision=prec)
        assert e.tile_info()["stage"] == 1
        g, _, _ = c.oracle.gradient_psi(c.q0)
        assert rel_l2(e.gradient(c.q0), g) < tol
        assert rel_l2(e.fetch("deltaX"), c.oracle.get("deltaX")) < (TOL_FIELD if prec == 0 else 2e-5)
        e.close()


def _largest_population(c, field, rsd):
    """Largest (tile, sub-cell octant) population of the oracle's particle positions: 8 x 8 x 16 tiles of home cells."""
    _, px, py, pz = c.oracle.Lag2Eul(field, rsd=rsd)
    n, d = c.p.Nx, c.p.L / c.p.Nx
    f = [np.asarray(a) / d for a in (px, py, pz)]
    cell = [np.minimum(np.floor(a).astype(np.int64), n) % n for a in f]
    octant = sum(((a - np.floor(a)) >= 0.5).astype(np.int64) << s for a, s in zip(f, (2, 1, 0)))
    tile = (cell[2] // 16) + (n // 16) * ((cell[1] // 8) + (n // 8) * (cell[0] // 8))
    return int(np.bincount(tile * 8 + octant).max())


def test_record_slots_follow_the_measured_populations(monkeypatch):
    """The one-pass binning starts with whatever partition it is given (here: far too small), runs the exact two-pass sort
    for the evaluation that overflowed, and extends the partition to its whole allocation at the next synchronising
    call; a field whose 1.5x largest (tile, octant) population does not fit the allocation gets the array reallocated
    for it.  It never shrinks (measured: nothing to gain, overflow to lose)."""
    monkeypatch.setenv("BCHMC_SORT_CAP", "64")             # 8 slots per (tile, octant): every tile overflows
    c = Case(Nx=32, likelihood=1, rsd_model=1)
    e = c.engine()
    i0 = e.tile_info()
    assert i0["one_pass"] == 1 and i0["cap"] == 64 and i0["cap_alloc"] >= 16 * 1024
    fields = sorted((s * c.truth for s in (0.02, 0.3, 1.0, 4.0)), key=lambda fld: _largest_population(c, fld, 1))
    caps = []
    for fld in (fields[0], fields[0], fields[-1], fields[0]):
        dX = c.oracle.Lag2Eul(fld, rsd=1)[0]
        e.forward(fld, 1)                                  # exact whichever sort ran; synchronises: adapt_slots
        assert rel_l2(e.fetch("deltaX"), dX) < TOL_FIELD
        info = e.tile_info()
        caps.append(info["cap"])
        maxc = _largest_population(c, fld, 1)
        assert info["one_pass"] == 1 and info["cap"] % 8 == 0 and info["cap"] == info["cap_alloc"] - info["cap_alloc"] % 8
        assert 8 * maxc <= 7 * (info["cap"] // 8)          # room for the field just seen, with the 1/8 margin
    assert caps[0] >= i0["cap_alloc"] - 7 and caps[1] == caps[0]   # extended to the whole allocation, then stable
    assert caps[2] >= caps[1] and caps[3] == caps[2]       # never shrinks
    e.close()


def test_record_array_is_reallocated_from_the_measured_population(monkeypatch):
    """BCHMC_SORT_CAP above the default allocation is not needed to hold a strongly clustered field: the array grows
    to 1.5x its largest population (+25 %) at the synchronising call after the overflow."""
    c = Case(Nx=32, likelihood=1, rsd_model=1)
    e = c.engine()
    i0 = e.tile_info()
    fld = 40.0 * c.truth                                   # displacements of many cells: strongly clustered
    maxc = _largest_population(c, fld, 1)
    dX = c.oracle.Lag2Eul(fld, rsd=1)[0]
    e.forward(fld, 1)
    assert rel_l2(e.fetch("deltaX"), dX) < TOL_FIELD
    info = e.tile_info()
    if 8 * maxc > 7 * (i0["cap"] // 8):                    # it did not fit: reallocated for it
        want = ((3 * maxc) // 2 + 16 + 7) // 8 * 8 * 8
        assert info["cap_alloc"] == want + want // 4 and info["cap"] == info["cap_alloc"] - info["cap_alloc"] % 8
        e.forward(fld, 1)                                  # ... and the one-pass binning holds it now
        assert e.tile_info()["cap"] == info["cap"] and rel_l2(e.fetch("deltaX"), dX) < TOL_FIELD
    else:
        assert info["cap"] == i0["cap"]
    e.close()


def test_long_trajectory_polls_the_slot_words(monkeypatch):
    """A 20-step trajectory that starts on too small a partition: the overflow is noticed by the lagging poll inside the
    trajectory (every 4 steps), not only at its end; every step is exact either way."""
    monkeypatch.setenv("BCHMC_SORT_CAP", "64")
    monkeypatch.setenv("BCHMC_VERBOSE", "1")
    c = Case(Nx=32, likelihood=1, rsd_model=1)
    e = c.engine()
    assert e.tile_info()["watch"] == 1
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 20)
    import torch
    dev = torch.device("cuda", 0)
    q0, p0 = torch.from_numpy(c.q0.reshape(-1)).to(dev), torch.from_numpy(c.p0.reshape(-1)).to(dev)
    q1, p1 = torch.empty_like(q0), torch.empty_like(p0)
    e.leapfrog_device(q0, p0, q1, p1, c.eps, 20)           # no synchronising call before the trajectory is enqueued
    grown_inside = e.tile_info()["cap"]                    # read before bchmc_steps_done: set by poll_slots
    assert e.steps_done() == 20
    assert grown_inside > 64
    assert rel_l2(q1.cpu().numpy(), q1o) < 10 * TOL_TRAJ_10 and rel_l2(p1.cpu().numpy(), p1o) < 10 * TOL_TRAJ_10
    e.close()


@pytest.mark.parametrize("kw", [dict(likelihood=1, rsd_model=0, sfmodel=2),
                                dict(likelihood=0, rsd_model=0, sfmodel=2, kth=2.0, deltaQ_factor=0.9)],
                         ids=["gauss_alpt", "poisson_alpt_kth2"])
@pytest.mark.parametrize("nx", [32, 64])
def test_alpt_on_the_2d_plans(monkeypatch, kw, nx):
    """Lag2Eul_non_zeldovich through k_step_boundary_x<ALPT> / k_alpt_mix_x (cellboundcomp as a k-space phase) against
    the oracle's real-space pipeline and against the engine's 3-D-plan path."""
    monkeypatch.setenv("BCHMC_FFT_PAD", "1")               # planes mode needs whole 128-byte k-groups per row
    c = Case(Nx=nx, **kw)
    e = c.engine()
    assert e.tile_info()["alpt_planes"] == 1
    dX, px, py, pz = c.oracle.Lag2Eul(c.truth, rsd=0)
    e.forward(c.truth, 0)
    psi = c.oracle.alpt_displacement(c.truth)
    for name, ref in zip(("psix", "psiy", "psiz"), psi):
        assert rel_l2(e.fetch(name), ref) < TOL_FIELD
    for name, ref in zip(("posx", "posy", "posz"), (px, py, pz)):
        assert rel_l2(e.fetch(name), ref) < TOL_FIELD
    assert rel_l2(e.fetch("deltaX"), dX) < TOL_FIELD
    g, _, _ = c.oracle.gradient_psi(c.q0)
    assert rel_l2(e.gradient(c.q0), g) < 10 * TOL_FIELD
    neps = 5
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, neps)
    q1, p1, done, dH, t = e.leapfrog_dh(c.q0, c.p0, c.eps, neps)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    assert done == neps and rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    assert np.all(np.abs(t - to) <= 10 * TOL_ENERGY * np.abs(to))
    qs, ps_, _ = e.leapfrog(c.q0, c.p0, c.eps, 1)          # first and last step at once
    q1o1, p1o1, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 1)
    assert rel_l2(qs, q1o1) < TOL_TRAJ_10 and rel_l2(ps_, p1o1) < TOL_TRAJ_10
    e.close()
    monkeypatch.setenv("BCHMC_NO_ALPT_PLANES", "1")
    e2 = c.engine()
    q2, p2, _ = e2.leapfrog(c.q0, c.p0, c.eps, neps)
    assert rel_l2(q2, q1) < 1e-12 and rel_l2(p2, p1) < 1e-12
    e2.close()


def test_alpt_planes_resident_chain_and_guard(monkeypatch):
    monkeypatch.setenv("BCHMC_FFT_PAD", "1")
    c = Case(Nx=32, likelihood=1, rsd_model=0, sfmodel=2)
    e = c.engine()
    e.chain_set_state(c.q0)
    e.chain_set_momenta(c.p0)
    dH, t, done = e.chain_attempt(c.eps, 4)
    q1o, p1o, _ = c.oracle.Hamiltonian_EoM(c.q0, c.p0, c.eps, 4)
    dHo, to = c.oracle.delta_Hamiltonian(c.q0, c.p0, q1o, p1o)
    assert done == 4 and np.all(np.abs(t - to) <= 10 * TOL_ENERGY * np.abs(to))
    q1, p1 = e.chain_get_proposal()
    assert rel_l2(q1, q1o) < TOL_TRAJ_10 and rel_l2(p1, p1o) < TOL_TRAJ_10
    p_bad = c.p0.copy().ravel()
    p_bad[0] = 1e60
    qb, pb, doneb = e.leapfrog(c.q0, p_bad, 1e-6, 6)       # the guard trips inside an ALPT planes-mode boundary
    qbo, pbo, donebo = c.oracle.Hamiltonian_EoM(c.q0, p_bad, 1e-6, 6)
    assert doneb == donebo == 1 and rel_l2(qb, qbo) < 1e-11
    e.close()
