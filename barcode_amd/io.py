"""On-disk formats next to the path.

* Field dumps: raw native-endian ``real_prec`` (double) arrays, no header -- ``write_array`` / ``read_array``
  (``barlib/src/IOfunctionsGen.cc:185-229``), including the ``.dat`` extension rule (`add_extension_if_missing`).
* ``performance_log.txt``: one tab-separated row of 14 columns per attempt (``HMC.cc:40-60``) under the header
  written by ``barcoderunner.cc:357-358``.
"""
import os

import numpy as np

PERFORMANCE_LOG_COLUMNS = ("accepted", "epsilon", "Neps", "dH", "dK", "dE", "dprior", "dlikeli",
                           "psi_prior_i", "psi_prior_f", "psi_likeli_i", "psi_likeli_f", "H_kin_i", "H_kin_f")


def add_extension_if_missing(fn, ext=".dat"):
    """IOfunctionsGen.cc:185-191: append ``ext`` only when the name has no '.' at all."""
    return fn + ext if fn.rfind(".") == -1 else fn


def write_array(fname, a):
    """IOfunctionsGen.cc:216-229: N * sizeof(real_prec) raw bytes."""
    np.ascontiguousarray(a, dtype=np.float64).tofile(add_extension_if_missing(fname))


def read_array(fname, n):
    """IOfunctionsGen.cc:194-203: read exactly ``n`` doubles (raises if the file is short or missing)."""
    path = add_extension_if_missing(fname)
    if not os.path.isfile(path):
        raise RuntimeError("In read_array: error opening file " + path)
    a = np.fromfile(path, dtype=np.float64, count=n)
    if a.size != n:
        raise RuntimeError("In read_array: %s holds %d values, %d requested" % (path, a.size, n))
    return a


def performance_log_header():
    return "\t".join(PERFORMANCE_LOG_COLUMNS) + "\n"


def performance_log_row(rec):
    """``rec``: one record of ``barcode_amd.hamil.HamiltonianMC`` (or any mapping with the 14 columns).
    C++ ``operator<<`` formatting: bool as 0/1, integers plain, doubles with 6 significant digits (%g)."""
    out = []
    for k in PERFORMANCE_LOG_COLUMNS:
        v = rec[k]
        if k == "accepted":
            out.append("1" if v else "0")
        elif k == "Neps":
            out.append(str(int(v)))
        else:
            out.append("%g" % float(v))
    return "\t".join(out) + "\n"


def dump_measured_spec(kmode, power, fname):
    """``dump_measured_spec`` / ``dump_ps_it`` (IOfunctions.cc:20-34, 37-82): one ``k   P(k)`` line per bin with
    ``k > 0`` and ``P > 0``, C++ default stream formatting (6 significant digits).  ``dump_ps_it`` names the file
    ``<dir>powSpecit<iGibbs>.dat`` (``power_spectrum_filename``)."""
    with open(fname, "w") as f:
        for x, y in zip(np.asarray(kmode).ravel(), np.asarray(power).ravel()):
            if y > 0.0 and x > 0.0:
                f.write("%g   %g\n" % (x, y))


def power_spectrum_filename(directory, iGibbs):
    return os.path.join(directory, "powSpecit%d.dat" % int(iGibbs)) if directory else "powSpecit%d.dat" % int(iGibbs)
