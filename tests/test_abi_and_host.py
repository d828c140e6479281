"""CPU-only checks: the C-ABI library loads and exports every symbol include/bchmc.h declares (no compute
calls), struct layouts agree between C and Python, host-side input generation and chain bookkeeping work,
and the world_size-2 epsilon-statistics exchange runs over gloo."""
import ctypes as C
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from barcode_amd import inputs
from barcode_amd.params import HamilParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    path = os.path.join(ROOT, "barcode_amd", "libbarcode_hip.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "barcode_amd", "csrc")])
    return path


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bchmc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bchmc_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    from barcode_amd import engine
    lib = C.CDLL(built_lib)
    declared = _declared_symbols()
    assert len(declared) >= 19
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert sorted(engine.EXPORTS) == declared


def test_config_struct_layout_matches_header():
    """Field order of BchmcConfig must be the header's; compile a tiny C probe for sizeof/offsetof."""
    from barcode_amd.engine import BchmcConfig, EpsRecord
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "bchmc.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu\n", sizeof(bchmc_config), offsetof(bchmc_config, particle_kernel_h),
             offsetof(bchmc_config, OL), offsetof(bchmc_config, device), sizeof(bchmc_eps_record),
             offsetof(bchmc_config, mk));
      return 0;
    }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        cfile, exe = os.path.join(d, "probe.c"), os.path.join(d, "probe")
        open(cfile, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe])
        vals = [int(v) for v in subprocess.check_output([exe]).split()]
    assert vals == [C.sizeof(BchmcConfig), BchmcConfig.particle_kernel_h.offset, BchmcConfig.OL.offset,
                    BchmcConfig.device.offset, C.sizeof(EpsRecord), BchmcConfig.mk.offset]


def test_engine_fails_loudly_without_the_hip_library(monkeypatch, tmp_path):
    from barcode_amd import engine
    monkeypatch.setattr(engine, "_lib", None)
    monkeypatch.setattr(engine, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(ImportError):
        engine.load()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "barcode_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.replace("# no oracle", ""), os.path.join(dirpath, fn)


def test_power_grid_follows_readtab():
    p = HamilParams(Nx=8, L=100.0)
    P = inputs.power_grid(p)
    assert P.shape == (8, 8, 8) and P[0, 0, 0] == 0.0
    assert np.all(P.ravel()[1:] > 0)
    # depends on |k| only: symmetric under index reflection (what makes K2's full-grid indexing harmless)
    assert np.allclose(P[1, 2, 3], P[7, 6, 5])
    k = 2 * np.pi / 100.0 * np.sqrt(1 + 4 + 9)
    ktab, ptab = inputs.read_power_table()
    assert np.isclose(P[1, 2, 3], np.interp(k, ktab, ptab))
    m = inputs.inverse_power_mass(P)
    assert m[0, 0, 0] == 0.0 and np.isclose(m[1, 2, 3], 1.0 / P[1, 2, 3])


def test_gaussian_random_field_spectrum_convention():
    """<|FFT f|^2> = N^2 P / V (random.cpp:82,106): check the measured spectrum over many modes."""
    p = HamilParams(Nx=32, L=400.0)
    P = inputs.power_grid(p)
    f = inputs.gaussian_random_field(p, P, seed=5)
    fk = np.fft.rfftn(f)
    sel = P[:, :, :17] > 0
    ratio = (np.abs(fk) ** 2)[sel] / (p.N ** 2 * P[:, :, :17][sel] / p.L ** 3)
    assert abs(ratio.mean() - 1.0) < 0.05
    assert np.array_equal(f, inputs.gaussian_random_field(p, P, seed=5))  # Philox: reproducible


def test_mock_observations_follow_setup_random_test():
    p = HamilParams(Nx=8, L=100.0, likelihood=0)
    dX = np.full((8, 8, 8), 0.5)
    w, s, nobs = inputs.mock_observations(p, dX)
    assert np.all(w == 1) and np.all(nobs >= 0) and np.all(nobs == np.round(nobs))
    p.likelihood = 1
    w, s, nobs = inputs.mock_observations(p, dX)
    assert np.all(nobs >= 0) and np.all(s == p.sigma_min)


def test_params_defaults_match_input_par():
    p = HamilParams()
    assert (p.Nx, p.L, p.mk, p.calc_h, p.likelihood, p.sfmodel, p.mass_type) == (64, 200.0, 3, 2, 1, 1, 1)
    assert p.D1 == 1.0 and abs(p.D2 - (-3.0 / 7.0) * 0.272 ** (-1.0 / 143.0)) < 1e-15
    assert np.isclose(p.eps_heuristic(), 2.38902581 * 262144 ** -0.57495347)
    assert p.particle_kernel_h == p.d


def test_eps_ring_matches_reference_tables():
    from barcode_amd.chains import EpsRing
    r = EpsRing(4)
    for i, (acc, eps) in enumerate([(1, .1), (0, .2), (1, .3), (1, .4), (0, .5)]):
        r.record(acc, eps)
    # fifth attempt wraps to slot 0 (time_step.cpp:192-195)
    assert list(r.epsilon) == [.5, .2, .3, .4] and list(r.acc_flag) == [False, False, True, True]
    assert r.acceptance_rate() == 0.5 and not r.due_for_update()
    r2 = EpsRing(4)
    for _ in range(4):
        r2.record(True, 1.0)
    assert r2.due_for_update()
    # crossing form of the same trigger: fires once per multiple of N_a, also when several records arrive between calls
    r3 = EpsRing(4)
    fired = []
    for burst in (3, 3, 1, 1, 9):
        for _ in range(burst):
            r3.record(True, 1.0)
        fired.append(r3.crossed(4))
    assert fired == [False, True, False, True, True]   # counts 3, 6, 7, 8, 17


def test_cpp_host_layer_is_built_and_exports_its_hooks():
    """libbarcode_shim.so (g++, include/bchmc_shim.hpp) loads, exports every extern "C" hook and agrees with the
    ctypes mirrors on the struct layouts (no compute: no GPU needed)."""
    import ctypes as C
    from barcode_amd import shim
    lib = shim.load()
    for s in shim.SHIM_EXPORTS:
        assert hasattr(lib, s)
    assert lib.bchmc_shim_sizeof_view() == C.sizeof(shim.HamilView)
    assert lib.bchmc_shim_sizeof_numerical() == C.sizeof(shim.HamilNumericalView)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` starts its own ranks; with fewer GPUs than ranks (none in the build container) it must
    refuse with a non-zero exit code instead of recording an N-GPU number from fewer chains, and never touch a GPU in
    the parent (torch.cuda.device_count() only)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine could actually run two ranks")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 2
    assert b"refusing to run" in r.stderr and r.stdout.strip() == b""
