"""On-disk formats against the reference's OWN fixture: test/data/io_array.dat, the 64-byte file
its Catch2 known-answer test reads (test/io_array.cpp:81-98).  The file is committed as data under tests/golden/."""
import os

import numpy as np
import pytest

from barcode_amd import io as bio
from tests.util import GOLDEN_DIR

EXPECTED = [18012.18201, 280.22, 300021.850, 3.14, 2., 333888., 807520.20, 170412.0]  # test/io_array.cpp:67-74


def test_read_array_reads_the_reference_fixture():
    a = bio.read_array(os.path.join(GOLDEN_DIR, "reference_io_array"), 8)  # extension added like the reference does
    assert a.tolist() == EXPECTED


def test_write_read_round_trip_and_byte_identity(tmp_path):
    """test/io_array.cpp:34-59 (round trip), plus: our writer reproduces the reference file byte for byte."""
    fn = str(tmp_path / "arr")
    bio.write_array(fn, np.array(EXPECTED))
    assert os.path.exists(fn + ".dat")
    assert open(fn + ".dat", "rb").read() == open(os.path.join(GOLDEN_DIR, "reference_io_array.dat"), "rb").read()
    assert bio.read_array(fn, 8).tolist() == EXPECTED
    with pytest.raises(RuntimeError):
        bio.read_array(str(tmp_path / "missing"), 8)
    with pytest.raises(RuntimeError):
        bio.read_array(fn, 9)
    assert bio.add_extension_if_missing("a.b") == "a.b" and bio.add_extension_if_missing("a") == "a.dat"


def test_performance_log_schema():
    assert bio.performance_log_header().split("\t")[0] == "accepted" and len(bio.PERFORMANCE_LOG_COLUMNS) == 14
    rec = dict(accepted=True, epsilon=0.00123456789, Neps=5, dH=-1.5, dK=2.0, dE=-3.5, dprior=1e-7, dlikeli=-3.5,
               psi_prior_i=1234567.0, psi_prior_f=1.0, psi_likeli_i=2.0, psi_likeli_f=3.0, H_kin_i=4.0, H_kin_f=5.0)
    row = bio.performance_log_row(rec).rstrip("\n").split("\t")
    assert row[:4] == ["1", "0.00123457", "5", "-1.5"] and row[8] == "1.23457e+06" and len(row) == 14


def test_dump_measured_spec_format(tmp_path):
    """IOfunctions.cc:29-31 / 75-80: bins with k > 0 and P > 0 only, `k   P`, 6 significant digits."""
    from barcode_amd.io import dump_measured_spec, power_spectrum_filename
    fn = power_spectrum_filename(str(tmp_path), 12)
    assert fn.endswith("powSpecit12.dat")
    dump_measured_spec([0.0, 0.0314159265, 0.25, 1.5], [5.0, 1234.56789, 0.0, 1e-5], fn)
    assert open(fn).read() == "0.0314159   1234.57\n1.5   1e-05\n"


def test_input_par_reader_against_the_reference_fixture():
    """The REQUIREs of the reference's own test (test/parameter_input_file.cpp:21-46) on its own fixture
    (test/data/input.par, copied as tests/golden/reference_test_input.par)."""
    import os
    from barcode_amd.input_par import parameter_inifile
    from tests.util import GOLDEN_DIR
    params = parameter_inifile(os.path.join(GOLDEN_DIR, "reference_test_input.par"))
    assert params.find(bool, "bool_true") is True
    assert params.find(bool, "bool_false") is False
    assert params.find(bool, "bool_comment") is False
    for key in ("zero", "zero_comment"):
        assert params.find(int, key) == 0 and params.find(float, key) == 0.0
    assert params.find(float, "float_one") == 1.0
    assert params.find(float, "float_minus") == -1.2
    assert params.find(str, "string") == "hello_no_spaces_please"
    assert params.find(str, "string_comment") == "stuff"


def test_hamil_params_from_the_reference_input_par_template():
    """data/input.par (the template main.cc reads) -> the scalars of the leapfrog path (init_par.cc:52-186, 293-334)."""
    import os
    from barcode_amd.input_par import hamil_params
    from tests.util import GOLDEN_DIR
    p = hamil_params(os.path.join(GOLDEN_DIR, "reference_template_input.par"))
    assert (p.Nx, p.L, p.mk, p.calc_h, p.likelihood, p.prior, p.sfmodel, p.rsd_model, p.mass_type) == \
        (64, 200.0, 3, 2, 1, 0, 1, 0, 1)
    assert (p.xobs, p.yobs, p.zobs, p.planepar, p.periodic) == (90.0, 90.0, 90.0, 1, 1)
    assert (p.kth, p.correct_delta, p.div_dH_by_N, p.particle_kernel_h_rel) == (4.0, 1, 0, 1.0)
    assert (p.grad_psi_prior_factor, p.grad_psi_likeli_factor, p.deltaQ_factor) == (1.0, 1.0, 1.0)
    assert (p.sigma_min, p.delta_min, p.ascale) == (1.0, -0.999, 1.0)
    assert p.particle_kernel_h == 200.0 / 64
    # the dataclass defaults ARE this template
    from barcode_amd.params import HamilParams
    d = HamilParams()
    for k in ("Nx", "L", "mk", "calc_h", "likelihood", "sfmodel", "rsd_model", "mass_type", "kth", "xobs", "planepar",
              "periodic", "correct_delta", "div_dH_by_N", "deltaQ_factor", "sigma_min", "delta_min", "ascale"):
        assert getattr(d, k) == getattr(p, k), k
