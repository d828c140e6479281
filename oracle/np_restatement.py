"""Second, independent CPU restatement of the hot path in vectorised numpy -- TEST INFRASTRUCTURE ONLY.

Purpose: cross-check ``oracle/bchmc_oracle.c`` (loops, own FFT) against a differently structured
implementation (array expressions, ``numpy.fft``) so transcription errors in either show up as a
disagreement.  It follows the same reference lines (cited per function) but is only meant for small
grids (<= 32^3).  PARITY UNPINNED, like the C oracle: neither has been checked against reference output.
"""
import numpy as np


class NpHamil:
    def __init__(self, p, signal_PS, mass_f, nobs, noise, window, mass_r=None):
        self.p = p
        n = p.Nx
        self.n, self.N, self.L, self.d = n, n ** 3, p.L, p.L / n
        self.shape = (n, n, n)
        self.signal_PS = signal_PS.reshape(self.shape)
        self.mass_f = mass_f.reshape(self.shape)
        self.mass_r = None if mass_r is None else mass_r.reshape(self.shape)
        self.nobs, self.noise, self.window = (a.reshape(self.shape) for a in (nobs, noise, window))
        i = np.arange(n)
        kfac = 2 * np.pi / p.L
        k1 = np.where(i <= n // 2, kfac * i, -kfac * (n - i))
        self.kx = k1[:, None, None]
        self.ky = k1[None, :, None]
        self.kz = k1[None, None, : n // 2 + 1]
        self.ksq = self.kx ** 2 + self.ky ** 2 + self.kz ** 2
        nyq = np.zeros((n, n, n // 2 + 1), bool)
        nyq[n // 2, :, :] = True
        nyq[:, n // 2, :] = True
        nyq[:, :, n // 2] = True
        self.nyq = nyq
        self.mass_fs = p.mass_type in (1, 2, 3, 4, 5)
        self.mass_rs = p.mass_type in (0, 5, 6, 60)

    # fftwrapper.cc:88-119
    def r2c(self, a):
        return np.fft.rfftn(a.reshape(self.shape))

    def c2r(self, c):
        return np.fft.irfftn(c, s=self.shape, axes=(0, 1, 2))  # includes the 1/N

    # HMC_help.cc:16-64
    def conv_inv(self, signal, corr):
        c = corr[:, :, : self.n // 2 + 1]
        mult = np.zeros_like(c)
        np.divide(self.L ** 3 / self.N, c, out=mult, where=c > 0)
        return self.c2r(self.r2c(signal) * mult)

    # EqSolvers.cc:168-277
    def theta2vel(self, delta):
        dk = self.r2c(delta)
        inv = np.zeros_like(self.ksq)
        np.divide(1.0, self.ksq, out=inv, where=self.ksq > 1e-14)
        inv[self.nyq] = 0.0
        base = -1j * dk * inv  # (Im, -Re) = -i * (Re + i Im)
        return tuple(self.c2r(base * k) for k in (self.kx, self.ky, self.kz))

    @staticmethod
    def _pacman(x, L):  # pacman.cpp:20-28
        x = np.where(x < 0, np.fmod(x, L) + L, x)
        return np.where(x >= L, np.fmod(x, L), x)

    # ---- f-3: ALPT displacement (Lag2Eul_non_zeldovich, Lag2Eul.cc:160-267), written independently of the C
    # oracle: derivatives with np.roll, the three filters combined in k-space (everything there is linear) ----
    def gradfindif(self, a, axis):  # gradient.cpp:81-154: 4th-order central difference
        fac = self.n / (2.0 * self.L)
        l, r = np.roll(a, 1, axis), np.roll(a, -1, axis)
        ll, rr = np.roll(a, 2, axis), np.roll(a, -2, axis)
        return -(fac * ((4.0 / 3) * (l - r) - (1.0 / 6) * (ll - rr)))

    def alpt_kernel(self, smol):  # convolution.cpp:224-324, Gaussian; sum of its inverse transform = K(0) = 1
        return np.exp(-self.ksq * smol * smol / 2.0)

    def alpt_displacement(self, delta):
        p = self.p
        d1 = delta.reshape(self.shape)
        inv = np.zeros_like(self.ksq)
        np.divide(1.0, self.ksq, out=inv, where=self.ksq > 0)
        phi = self.c2r(-self.r2c(d1) * inv)                                   # PoissonSolver, EqSolvers.cc:29-64
        g = [self.gradfindif(phi, a) for a in range(3)]                       # calc_m2v_mem, EqSolvers.cc:373-422
        xx, xy, xz = (self.gradfindif(g[0], a) for a in range(3))
        yy, yz = self.gradfindif(g[1], 1), self.gradfindif(g[1], 2)
        zz = self.gradfindif(g[2], 2)
        d2 = xx * yy - xy * xy + xx * zz - xz * xz + yy * zz - yz * yz
        div_2lpt = p.D1 * d1 - p.D2 * d2
        psilin = -p.D1 * d1
        arg = 1.0 + 2.0 / 3.0 * psilin
        div_sc = -np.where(arg > 0, 3.0 * (np.sqrt(np.maximum(arg, 0.0)) - 1.0), -3.0)
        K = self.alpt_kernel(p.kth)
        mix = K * self.r2c(div_2lpt) + (1.0 - K) * self.r2c(div_sc)           # K o A + B - K o B
        inv_e = np.zeros_like(self.ksq)
        np.divide(1.0, self.ksq, out=inv_e, where=self.ksq > 1e-14)           # linearvel3d, EqSolvers.cc:156-160
        inv_e[self.nyq] = 0.0
        base = -1j * mix * inv_e
        psi = [self.c2r(base * k) for k in (self.kx, self.ky, self.kz)]
        shift = lambda a: 0.5 * (a + np.roll(a, (1, 1, 1), (0, 1, 2)))         # cellboundcomp, massFunctions.cc:588-658
        return tuple(shift(a) for a in psi)

    # Lag2Eul.cc:69-132 / 338-424 + disp_part.cc + rsd.cc
    def positions(self, delta, rsd):
        p = self.p
        if not rsd and p.sfmodel != 1:
            psi = self.alpt_displacement(delta)
        else:
            psi = self.theta2vel(-p.D1 * delta.reshape(self.shape))
        c = self.d * np.arange(self.n) + 0.5 * self.d
        pos = [self._pacman(c[:, None, None] + psi[0], self.L),
               self._pacman(c[None, :, None] + psi[1], self.L),
               self._pacman(c[None, None, :] + psi[2], self.L)]
        if rsd:
            E = np.sqrt(p.OM / p.ascale ** 3 + (1 - p.OM - p.OL) / p.ascale ** 2 + p.OL)
            f = (p.OM / (E * E * p.ascale ** 3)) ** (5.0 / 9.0)
            cpec = f * 100.0 * E * p.ascale
            pos[2] = self._pacman(pos[2] + (cpec * psi[2]) * (1.0 / (100.0 * E) / p.ascale), self.L)
        return pos

    # massFunctions.cc:392-495 + 366-384
    def density_sph(self, pos):
        n, d, h = self.n, self.d, self.p.particle_kernel_h
        reach = int(2 * h / d) + 1
        px, py, pz = (a.ravel() for a in pos)
        ix, iy, iz = ((a / d).astype(np.int64) for a in (px, py, pz))
        rho = np.zeros(self.N)
        for i1 in range(-reach, reach + 1):
            dx = px - ((ix + 0.5) * d + i1 * d)
            for i2 in range(-reach, reach + 1):
                dy = py - ((iy + 0.5) * d + i2 * d)
                for i3 in range(-reach, reach + 1):
                    dz = pz - ((iz + 0.5) * d + i3 * d)
                    q = np.sqrt(dx * dx + dy * dy + dz * dz) / h
                    w = np.where(q <= 1, 1 - 1.5 * q * q + 0.75 * q ** 3, np.where(q <= 2, 0.25 * (2 - q) ** 3, 0.0))
                    w *= 1.0 / np.pi / h ** 3
                    idx = ((iz + i3) % n) + n * (((iy + i2) % n) + n * ((ix + i1) % n))
                    np.add.at(rho, idx, np.where(q <= 2, w, 0.0))
        return rho.reshape(self.shape)

    def lag2eul(self, delta, rsd):
        pos = self.positions(delta, rsd)
        rho = self.density_sph(pos)
        return rho / rho.mean() - 1.0, pos

    # likelihood partials
    def partial_f(self, dX):
        p = self.p
        w, nobs, s = self.window, self.nobs, self.noise
        if p.likelihood == 1:
            lam = w * p.rho_c * (1 + p.biasP * dX) ** p.biasE
            return np.where((w > 0) & (lam > 0), (nobs - lam) / (s * s), 0.0)
        if p.likelihood == 0:
            dens = 1 + p.biasP * dX
            lam = w * p.rho_c * dens ** p.biasE
            with np.errstate(divide="ignore", invalid="ignore"):
                v = (1 - nobs / lam) * p.rho_c * p.biasE * p.biasP * dens ** (p.biasE - 1)
            return np.where((w > 0) & (dens > 0), v, 0.0)
        if p.likelihood == 2:
            with np.errstate(divide="ignore"):
                lam = np.log(p.rho_c * (1 + p.biasP * dX) ** p.biasE)
            return np.where(w > 0, (nobs - lam) / (s * s), 0.0)
        raise ValueError

    # HMC_models.cc:200-303 (+ 77-128, SPH_kernel.cpp:148-208)
    def calc_V(self, part_like, pos, rsd):
        p = self.p
        n, d, h = self.n, self.d, p.particle_kernel_h
        reach = int(2 * h / d) + 1
        normalize = p.rho_c * self.L ** 3 / self.N
        norm = 1.0 / (np.pi * h ** 4)
        px, py, pz = (a.ravel() for a in pos)
        ix, iy, iz = ((a / d).astype(np.int64) for a in (px, py, pz))
        cx, cy, cz = (a / h - (i + 0.5) * d / h for a, i in ((px, ix), (py, iy), (pz, iz)))
        pl = part_like.ravel()
        V = [np.zeros(self.N) for _ in range(3)]
        for i1 in range(-reach, reach + 1):
            for i2 in range(-reach, reach + 1):
                for i3 in range(-reach, reach + 1):
                    if (abs(i1) - 0.5) ** 2 + (abs(i2) - 0.5) ** 2 + (abs(i3) - 0.5) ** 2 > (2 * h / d) ** 2:
                        continue
                    x, y, z = cx - i1 * d / h, cy - i2 * d / h, cz - i3 * d / h
                    qs = x * x + y * y + z * z
                    q = np.sqrt(qs)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        part = np.where(qs > 4, 0.0, np.where(qs > 1, -0.75 * (q - 2) ** 2 * norm / q,
                                                              (2.25 * q - 3) * norm))
                    idx = ((iz + i3) % n) + n * (((iy + i2) % n) + n * ((ix + i1) % n))
                    c = pl[idx] * part
                    V[0] += c * x
                    V[1] += c * y
                    V[2] += c * z
        V = [normalize * v for v in V]
        if rsd:
            E = np.sqrt(p.OM / p.ascale ** 3 + (1 - p.OM - p.OL) / p.ascale ** 2 + p.OL)
            V[2] = V[2] + (p.OM / (E * E * p.ascale ** 3)) ** (5.0 / 9.0) * V[2]
        return [v.reshape(self.shape) for v in V]

    # HMC_models.cc:312-372 + gradient.cpp:157-211
    def calc_h(self, V):
        inv = np.zeros_like(self.ksq)
        np.divide(1.0, self.ksq, out=inv, where=self.ksq > 0)
        inv[self.nyq] = 0.0
        hk = sum(-1j * self.r2c(v) * k * inv for v, k in zip(V, (self.kx, self.ky, self.kz)))
        return self.c2r(hk)

    # HMC_models_testing.cpp:25-50 with gradfft (gradient.cpp:22-78) / gradfindif (gradient.cpp:81-154)
    def calc_h_legacy(self, dX):
        p = self.p
        part = self.partial_f(dX)
        if p.likelihood == 1:
            fk = self.r2c(dX)
            grads = [self.c2r(np.where(self.nyq, 0.0, 1j * k * fk)) for k in (self.kx, self.ky, self.kz)]
        else:
            f = dX if p.likelihood == 0 else np.log(p.rho_c * (1 + np.maximum(dX, p.delta_min)))
            fac = self.n / (2.0 * self.L)
            grads = [-(fac * ((4.0 / 3) * (np.roll(f, 1, a) - np.roll(f, -1, a))
                              - (1.0 / 6) * (np.roll(f, 2, a) - np.roll(f, -2, a)))) for a in range(3)]
        return self.calc_h([part * g for g in grads])

    # HMC_models.cc:377-471
    def grad_log_like(self, q):
        p = self.p
        rsd = bool(p.rsd_model)
        dX, pos = self.lag2eul(p.deltaQ_factor * q.reshape(self.shape), rsd)
        self.deltaX, self.pos = dX, pos
        if p.calc_h == 0:
            hfield = self.calc_h_legacy(dX)
        else:
            hfield = self.calc_h(self.calc_V(self.partial_f(dX), pos, rsd))
        norm = -1.0 * p.deltaQ_factor * (p.D1 if p.correct_delta else 1.0)
        return norm * hfield

    # HMC.cc:146-206
    def gradient_psi(self, q):
        p = self.p
        gp = p.grad_psi_prior_factor * self.conv_inv(q, self.signal_PS)
        gl = p.grad_psi_likeli_factor * self.grad_log_like(q)
        return gp + gl, gp, gl

    # energies: HMC.cc:64-143 and the *_log_like functions
    def kinetic(self, mom):
        mom = mom.reshape(self.shape)
        dummy = np.zeros(self.shape)
        if self.mass_fs:
            dummy = self.conv_inv(mom, self.mass_f)
        if self.mass_rs:
            inv = np.zeros(self.shape)
            np.divide(1.0, self.mass_r, out=inv, where=self.mass_r > 0)
            dummy = dummy + inv * mom
        return float(np.sum(0.5 * mom * dummy))

    def log_prior(self, q):
        q = q.reshape(self.shape)
        return float(np.sum(0.5 * q * self.conv_inv(q, self.signal_PS)))

    def log_like(self, q):
        p = self.p
        q = q.reshape(self.shape)
        w, nobs, s = self.window, self.nobs, self.noise
        if p.likelihood == 1:
            dX, _ = self.lag2eul(p.deltaQ_factor * q, bool(p.rsd_model))
            lam = w * p.rho_c * (1 + p.biasP * dX) ** p.biasE
            return float(np.sum(np.where((w > 0) & (lam > 0), 0.5 * ((lam - nobs) / s) ** 2, 0.0)))
        if p.likelihood == 0:
            dX, _ = self.lag2eul(q, False)
            lam = w * p.rho_c * (1 + p.biasP * dX) ** p.biasE
            with np.errstate(divide="ignore", invalid="ignore"):
                return float(np.sum(np.where((w > 0) & (lam > 0), lam - nobs * np.log(lam), 0.0)))
        if p.likelihood == 2:
            dX, _ = self.lag2eul(q, False)
            lam = np.log(p.rho_c * (1 + np.maximum(dX, p.delta_min)))
            return float(np.sum(np.where(w > 0, 0.5 * (lam - nobs) ** 2 / (s * s), 0.0)))
        raise ValueError

    # HMC.cc:251-369
    def leapfrog(self, q0, p0, eps, neps):
        q, mom = q0.reshape(self.shape).copy(), p0.reshape(self.shape).copy()
        g, _, _ = self.gradient_psi(q)
        for _ in range(neps):
            mom -= 0.5 * eps * g
            dummy = np.zeros(self.shape)
            if self.mass_fs:
                dummy = self.conv_inv(mom, self.mass_f)
            if self.mass_rs:
                inv = np.zeros(self.shape)
                np.divide(1.0, self.mass_r, out=inv, where=self.mass_r > 0)
                dummy = dummy + inv * mom
            q += eps * dummy
            g, _, _ = self.gradient_psi(q)
            mom -= 0.5 * eps * g
        return q, mom
