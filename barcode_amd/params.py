"""Scalars of the reference's ``HAMIL_NUMERICAL`` / ``HAMIL_DATA`` that the leapfrog path reads.

Field names, meanings and defaults follow ``/root/reference/barlib/include/struct_hamil.h:51-222`` and the
``input.par`` template ``/root/reference/data/input.par:4-168``; cosmology constants follow
``INIT_COSMOLOGY`` (``barlib/src/init_par.cc:430-532``, WMAP7 case 2, the compiled-in default).
"""
from dataclasses import dataclass, field


@dataclass
class HamilParams:
    # grid (cubic only: init_par.cc:116-118)
    Nx: int = 64
    L: float = 200.0
    min1: float = 0.0          # xllc
    min2: float = 0.0
    min3: float = 0.0
    # redshift-space distortions (input.par:128-133)
    xobs: float = 90.0
    yobs: float = 90.0
    zobs: float = 90.0
    planepar: int = 1
    periodic: int = 1
    # model switches
    mk: int = 3                # masskernel: 0 NGP, 1 CIC, 2 TSC, 3 SPH
    calc_h: int = 2
    likelihood: int = 1        # 0 Poisson, 1 Gaussian, 2 log-normal, 3 GRF
    prior: int = 0             # only the Gaussian prior exists upstream (init_par.cc:582-588)
    sfmodel: int = 1           # 1 Zel'dovich, anything else ALPT (Lag2Eul.cc:325-331); ignored when rsd_model is set
    kth: float = 4.0           # = slength (input.par:121, struct_hamil.h:259): ALPT split scale
    rsd_model: int = 0
    mass_type: int = 1
    correct_delta: int = 1
    div_dH_by_N: int = 0
    particle_kernel: int = 0
    particle_kernel_h_rel: float = 1.0
    # test factors (input.par:157-166)
    grad_psi_prior_factor: float = 1.0
    grad_psi_likeli_factor: float = 1.0
    deltaQ_factor: float = 1.0
    # observational scalars (INIT_OBSERVATIONAL, init_par.cc:574-578)
    rho_c: float = 1.0
    biasP: float = 1.0
    biasE: float = 1.0
    sigma_min: float = 1.0
    delta_min: float = -0.999
    # cosmology at z = 0 (WMAP7): D1 = 1 exactly, D2 = -3/7 D1^2 Omega^(-1/143) (init_par.cc:519-528)
    ascale: float = 1.0
    OM: float = 0.272
    OL: float = 0.728
    D1: float = 1.0
    D2: float = field(default=None)

    def __post_init__(self):
        if self.D2 is None:
            omega = self.OM / (self.ascale ** 3 * (self.OM / self.ascale ** 3 + self.OL
                                                   + (1.0 - self.OM - self.OL) / self.ascale ** 2))
            self.D2 = -3.0 / 7.0 * self.D1 * self.D1 * omega ** (-1.0 / 143.0)

    @property
    def N(self):
        return self.Nx ** 3

    @property
    def d(self):
        return self.L / self.Nx

    @property
    def particle_kernel_h(self):
        # init_par.cc:378-379: h = h_rel * average cell size
        return self.particle_kernel_h_rel * self.d

    def eps_heuristic(self):
        """Heuristic step size ``eps_fac_target`` of init_par.cc:259-261."""
        return 2.38902581 * float(self.N) ** (-0.57495347)
