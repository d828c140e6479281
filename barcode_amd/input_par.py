"""``input.par`` reader with the reference's parsing rules, and the mapping of its keys onto ``HamilParams``.

``parameter_inifile`` follows ``barlib/src/ini_reader.cpp:14-44`` + ``barlib/include/ini_reader.hpp:16-28``: every
white-space character is removed from a line first, lines that are empty or start with ``#`` are skipped, a trailing
``# comment`` is cut, the rest is split at the first ``=``; ``find`` converts with the semantics of
``std::stringstream >> std::boolalpha >> value``.  ``hamil_params`` reads the keys of the leapfrog path as
``INIT_PARAMS`` does (``barlib/src/init_par.cc:52-186, 293-334``).
"""
from .params import HamilParams


class parameter_inifile:
    def __init__(self, filename):
        self.parameters = {}
        try:
            f = open(filename)
        except OSError:
            # the reference only prints "Couldn't open config file ..." and goes on with an empty map
            return
        with f:
            for line in f:
                line = "".join(ch for ch in line if not ch.isspace())
                if not line or line[0] == "#":
                    continue
                cut = line.find("#")
                if cut != -1:
                    line = line[:cut]
                pos = line.find("=")
                key = line[:pos] if pos != -1 else line
                value = line[pos + 1:] if pos != -1 else line  # substr(npos + 1) == substr(0) upstream
                self.parameters[key] = value

    def find(self, kind, key):
        """``params.find<T>(key)``: ``kind`` is ``bool``, ``int``, ``float`` or ``str``.  A missing key yields the
        value-initialised result of a failed stream extraction (``False`` / 0 / 0.0 / ``""``), as upstream."""
        text = self.parameters.get(key, "")
        if kind is str:
            return text
        if kind is bool:
            return text == "true"  # std::boolalpha accepts exactly "true" / "false"
        try:
            if kind is int:
                # operator>> for integers stops at the first character that cannot continue the number
                digits = ""
                for i, ch in enumerate(text):
                    if ch.isdigit() or (i == 0 and ch in "+-"):
                        digits += ch
                    else:
                        break
                return int(digits)
            return float(_leading_float(text))
        except ValueError:
            return kind()


def _leading_float(text):
    """Longest prefix strtod would accept (enough for input.par: digits, sign, point, exponent)."""
    best = ""
    for end in range(1, len(text) + 1):
        try:
            float(text[:end])
            best = text[:end]
        except ValueError:
            if text[:end][-1] not in "eE+-.":
                break
    if not best:
        raise ValueError(text)
    return best


def hamil_params(filename, **overrides):
    """HamilParams from an ``input.par`` (keys and meaning: init_par.cc:52-186, 293-334; cubic grid: Nx, Lx only)."""
    p = parameter_inifile(filename)
    kw = dict(
        Nx=p.find(int, "Nx"), L=p.find(float, "Lx"),
        min1=p.find(float, "xllc"), min2=p.find(float, "yllc"), min3=p.find(float, "zllc"),
        xobs=p.find(float, "xobs"), yobs=p.find(float, "yobs"), zobs=p.find(float, "zobs"),
        planepar=int(p.find(bool, "planepar")), periodic=int(p.find(bool, "periodic")),
        mk=p.find(int, "masskernel"), calc_h=p.find(int, "calc_h"),
        likelihood=p.find(int, "likelihood"), prior=p.find(int, "prior"),
        sfmodel=p.find(int, "sfmodel"), kth=p.find(float, "slength"),
        rsd_model=int(p.find(bool, "rsd_model")), mass_type=p.find(int, "mass_type"),
        correct_delta=int(p.find(bool, "correct_delta")), div_dH_by_N=int(p.find(bool, "div_dH_by_N")),
        particle_kernel=p.find(int, "particle_kernel"), particle_kernel_h_rel=p.find(float, "particle_kernel_h_rel"),
        grad_psi_prior_factor=p.find(float, "grad_psi_prior_factor"),
        grad_psi_likeli_factor=p.find(float, "grad_psi_likeli_factor"),
        deltaQ_factor=p.find(float, "deltaQ_factor"),
        sigma_min=p.find(float, "sigma_min"), delta_min=p.find(float, "delta_min"),
        ascale=1.0 / (1.0 + p.find(float, "z")),
    )
    kw.update(overrides)
    return HamilParams(**kw)
