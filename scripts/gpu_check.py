"""Quick GPU-vs-oracle comparison used while developing (not a test)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from barcode_amd.params import HamilParams
from barcode_amd import inputs
from barcode_amd.engine import Engine
from oracle.oracle import Oracle


def rel(a, b):
    a = np.asarray(a).ravel(); b = np.asarray(b).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def run(Nx, L, lik, rsd, neps=3, mass_type=1):
    p = HamilParams(Nx=Nx, L=L, likelihood=lik, rsd_model=rsd, mass_type=mass_type)
    f = inputs.make_fields(p)
    o = Oracle(p)
    mass_r = np.abs(inputs.gaussian_random_field(p, f["signal_PS"], 77)) + 0.5
    o.set(signal_PS=f["signal_PS"], mass_f=f["mass_f"], mass_r=mass_r)
    dX, px, py, pz = o.Lag2Eul(f["truth"], rsd=rsd)
    win, noise, nobs = inputs.mock_observations(p, dX.reshape(Nx, Nx, Nx), delta_lag=f["truth"])
    o.set(window=win, noise=noise, nobs=nobs)
    e = Engine(p)
    e.upload(signal_PS=f["signal_PS"], mass_f=f["mass_f"], mass_r=mass_r, window=win, noise=noise, nobs=nobs)
    print(f"== Nx={Nx} lik={lik} rsd={rsd} mass_type={mass_type}")
    if lik != 3:
        e.forward(f["truth"], rsd)
        print("  deltaX", rel(e.fetch("deltaX"), dX), "posx", rel(e.fetch("posx"), px), "posz", rel(e.fetch("posz"), pz))
    g, gp, gl = o.gradient_psi(f["q0"])
    gg = e.gradient(f["q0"])
    print("  grad", rel(gg, g), "prior", rel(e.fetch("grad_prior"), gp), "like", rel(e.fetch("grad_like"), gl))
    eps = 0.1 * p.eps_heuristic()
    t = time.time(); q1, p1, done = o.Hamiltonian_EoM(f["q0"], f["p0"], eps, neps); t_o = time.time() - t
    t = time.time(); q1g, p1g, doneg = e.leapfrog(f["q0"], f["p0"], eps, neps); t_g = time.time() - t
    print("  traj q", rel(q1g, q1), "p", rel(p1g, p1), done, doneg, "t_oracle", round(t_o, 3), "t_gpu", round(t_g, 4))
    dH, terms = o.delta_Hamiltonian(f["q0"], f["p0"], q1, p1)
    dHg, termsg = e.delta_hamiltonian(f["q0"], f["p0"], q1, p1)
    print("  energies rel", np.abs(termsg - terms) / np.abs(terms), "dH", dH, dHg)


if __name__ == "__main__":
    for (lik, rsd) in [(1, 0), (1, 1), (0, 0), (2, 0), (3, 0)]:
        run(16, 50.0, lik, rsd)
    run(16, 50.0, 1, 0, mass_type=0)
    run(16, 50.0, 1, 0, mass_type=5)
    run(32, 100.0, 1, 1, neps=5)
